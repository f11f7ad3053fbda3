"""Development probe (GPU box): per-item wall-clock stamps of gf_daf_raccumulate_kernel (library built with -DGF_DAF_TL).
GF_LIB=.../libgf_hip_daftl.so python tools/timeline_daf.py [projected|uniform]"""
import ctypes
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
from gaussianformer_amd import _lib
from gaussianformer_amd.deformable_aggregation import deformable_aggregation_backward
from gaussianformer_amd.synthetic import make_daf_inputs

dev = torch.device("cuda:0")
dist = sys.argv[1] if len(sys.argv) > 1 else "projected"
pts = 230400
d = make_daf_inputs(num_pts=pts, seed=0)
feat, ss, st, loc, w = (torch.from_numpy(d[k]).to(dev) for k in ("mc_ms_feat", "spatial_shape", "scale_start_index", "sampling_location", "weights"))
if dist == "projected":
    import bench_frame
    from gaussianformer_amd.deformable_prepare import deformable_prepare
    g = torch.Generator(device="cpu").manual_seed(1)
    A = pts // 9
    lo = torch.tensor(bench_frame.PC_RANGE[:3]); hi = torch.tensor(bench_frame.PC_RANGE[3:])
    centre = lo + (hi - lo) * torch.rand(1, A, 3, generator=g)
    offs = torch.tensor(bench_frame.FIX_SCALE + [[0.3, 0.3, 0.0], [-0.3, 0.3, 0.0]]) * 0.35
    kp = (centre[:, :, None] + offs[None, None]).to(dev)
    pm, wh = bench_frame.cameras(dev)
    raw = torch.randn(1, A, 6, 4, 9, 4, generator=g).to(dev)
    loc, w = deformable_prepare(kp, pm, wh, raw)
    loc, w = loc.contiguous(), w.contiguous()
go = torch.randn(1, loc.shape[1], 128, device=dev)
for _ in range(3):
    gf, gl, gw = torch.zeros_like(feat), torch.zeros_like(loc), torch.zeros_like(w)
    deformable_aggregation_backward(feat, ss, st, loc, w, go, gf, gl, gw)
    torch.cuda.synchronize()
lib = _lib.load()
n = 8 * 16384
buf = np.zeros(n, dtype=np.uint64)
lib.gf_debug_daf_timeline.argtypes = [ctypes.c_void_p, ctypes.c_int]
rc = lib.gf_debug_daf_timeline(buf.ctypes.data, n)
t = buf.reshape(-1, 8)
t = t[t[:, 0] > 0]
ns = 10.0   # wall_clock64 ticks at 100 MHz
t0 = t[:, 0].min()
ni = (t[:, 4] & 0xffffffff).astype(np.int64); wg = (t[:, 4] >> 32).astype(np.int64)
print(f"{dist}: rc {rc}, items {len(t)}, samples {ni.sum()}, kernel span {(t[:, 3].max() - t0) * ns / 1e3:.1f} us")
head = (t[:, 1] - t[:, 0]) * ns / 1e3; body = (t[:, 2] - t[:, 1]) * ns / 1e3; tail = (t[:, 3] - t[:, 2]) * ns / 1e3
print(f"per item: header mean {head.mean():.2f} us, batches mean {body.mean():.2f} us (per batch of 64: {body.sum() / np.ceil(ni / 64).sum():.2f} us), row adds mean {tail.mean():.2f} us max {tail.max():.2f}")
ph = np.stack([(t[:, 6] >> (16 * i)) & 0xffff for i in range(4)] + [t[:, 7]], 1).astype(np.float64) * ns / 1e3
nb = np.ceil(ni / 64).sum()
print("per batch (wave 0's clock): wait+stage %.2f us, taps %.2f, scan %.2f, scatter %.2f, row walks %.2f" % tuple(ph.sum(0) / nb))
print("item size histogram:", np.histogram(ni, bins=[1, 2, 8, 32, 64, 128, 256, 511, 512, 513])[0].tolist())
busy = np.zeros(512); last = np.zeros(512); cnt = np.zeros(512, dtype=int)
for i in range(len(t)):
    busy[wg[i]] += (t[i, 3] - t[i, 0]) * ns / 1e3; last[wg[i]] = max(last[wg[i]], (t[i, 3] - t0) * ns / 1e3); cnt[wg[i]] += 1
print(f"workgroups: busy mean {busy.mean():.1f} us min {busy.min():.1f} max {busy.max():.1f}; last end mean {last.mean():.1f} min {last.min():.1f} max {last.max():.1f}; items per wg mean {cnt.mean():.1f} max {cnt.max()}")
first = np.array([t[wg == k, 0].min() if (wg == k).any() else t0 for k in range(512)])
print(f"first claim after kernel start: mean {(first - t0).mean() * ns / 1e3:.2f} us max {(first - t0).max() * ns / 1e3:.2f}")
order = np.argsort(t[:, 3])[-8:]
for i in order:
    print(f"  late item: wg {wg[i]} samples {ni[i]} region {t[i, 5]} start {(t[i, 0] - t0) * ns / 1e3:.1f} head {head[i]:.1f} body {body[i]:.1f} tail {tail[i]:.1f} end {(t[i, 3] - t0) * ns / 1e3:.1f}")
