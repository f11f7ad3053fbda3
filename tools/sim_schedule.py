"""What would other ways of dealing the wave kernel's units to its waves buy?  Replays the measured per-unit durations of one
launch (the .npy written by tools/timeline_wave.py: eight stamps per unit) through a list scheduler with the kernel's own rule
-- 256 waves per XCD, the next unit in index order to the first free wave -- and through variants of it.
python tools/sim_schedule.py [timeline_wave_<config>.npy]"""
import heapq
import sys

import numpy as np

path = sys.argv[1] if len(sys.argv) > 1 else "profiles/timeline_wave_nuscenes_gs25600_solid_r03d.npy"
T = np.load(path)
T = T[T[:, 0] > 0]
nu = len(T)
per_xcd = (nu + 7) // 8
dur = (T[:, 6] - T[:, 0]) / 100.0                       # unit: claim to stored
cons = np.where(T[:, 4] > 0, (T[:, 5] - T[:, 4]) / 100.0, 0.0)   # groups only
fixed = dur - cons
print(f"{nu} units, {per_xcd} per XCD; unit {dur.mean():.2f} us = {fixed.mean():.2f} fixed + {cons.mean():.2f} in groups; "
      f"measured launch {(T[:, 6].max() - T[:, 0].min()) / 100.0:.2f} us; work per slot {dur.sum() / 2048:.2f} us")


def span(pieces, slots=256, t0=0.7):
    h = [t0] * slots
    heapq.heapify(h)
    end = 0.0
    for d in pieces:
        t = heapq.heappop(h)
        heapq.heappush(h, t + d)
        end = max(end, t + d)
    return end


def launch(policy):
    return max(span(policy(dur[x * per_xcd:(x + 1) * per_xcd], cons[x * per_xcd:(x + 1) * per_xcd])) for x in range(8))


print("as launched (index order, dynamic claims)        : %.1f us" % launch(lambda d, c: list(d)))
print("longest unit first (durations known in advance)  : %.1f us" % launch(lambda d, c: sorted(d, reverse=True)))
rem = per_xcd % 256


def split_tail(d, c, keep=0.9):
    """the units of the last, partial round as two single bricks each: all of the fixed cost (x keep), half of the groups"""
    out = list(d[:per_xcd - rem])
    for k in range(per_xcd - rem, len(d)):
        out += [(d[k] - c[k]) * keep + 0.5 * c[k]] * 2
    return out


print("last round's units split into single bricks      : %.1f us" % launch(split_tail))
for cut in (1.0, 2.0, 3.0):
    print(f"every unit {cut:.0f} us shorter (fixed cost)               : %.1f us" % launch(lambda d, c, cut=cut: list(np.maximum(d - cut, 1.0))))
print("perfect balance (work / slots + one start-up)    : %.1f us" % (dur.sum() / 2048 + 0.7))
