"""Debug: per-unit timeline of the solo / fused render kernel (-DGF_TIMELINE=1 build).  python tools/timeline_solo.py [config] [flags]
flags 1 (GF_PTS_ASSUME_DENSE) = the fused single-launch forward; 0 = option "dev.splat_solo" = the two-launch solo kernel."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from gaussianformer_amd import build as _b
_tl = os.path.join(_b.CSRC, "libgf_hip_timeline_dev.so")
_deps = [os.path.join(_b.CSRC, f) for f in _b.SOURCES + _b.HEADERS]
if not os.path.exists(_tl) or any(os.path.getmtime(d) > os.path.getmtime(_tl) for d in _deps if os.path.exists(d)):
    _b.build(extra_flags=("-DGF_TIMELINE=1", "-DGF_DEV=1"), lib_name="libgf_hip_timeline_dev.so")
os.environ["GF_LIB"] = _tl
from gaussianformer_amd import _lib
from gaussianformer_amd.local_aggregate import SplatForwardPlan
from gaussianformer_amd.synthetic import make_splat_inputs
import oracle
config = sys.argv[1] if len(sys.argv) > 1 else "nuscenes_gs25600_solid"
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 1
_lib.set_option("dev.splat_solo" if not flags else "dev.splat_fused", 1)
dev = torch.device("cuda:0")
si = make_splat_inputs(config, seed=0)
pi, mi, radii, cov6 = oracle.prepare_splat_inputs(si.pts, si.means3D, si.scales, si.cov3D, si.pc_min, si.grid_size, si.scale_multiplier)
t = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (si.pts, pi, si.means3D, mi, si.opacities, si.semantics, radii, cov6)]
plan = SplatForwardPlan(0, *t, si.H, si.W, si.D, flags=flags)
lib = _lib.load()
for _ in range(5): plan.run()
torch.cuda.synchronize()
nunits = ((si.H + 7) // 8) * ((si.W + 7) // 8) * 4 * ((si.D + 7) // 8)
nw = 4096
tl = torch.zeros(12 * nunits + 4 * nw, dtype=torch.int64, device=dev)
lib.gf_debug_set_timeline.argtypes = [ctypes.c_void_p]
lib.gf_debug_set_timeline(tl.data_ptr())
plan.run(); torch.cuda.synchronize()
lib.gf_debug_set_timeline(None)
print("state", plan.state_words()[:5])
A = tl.cpu().numpy().astype(np.float64)
T = A[:12 * nunits].reshape(nunits, 12)
F = A[12 * nunits:].reshape(nw, 4)
F = F[F[:, 0] > 0]
T = T[T[:, 0] > 0]
t0 = min(T[:, 0].min(), F[:, 0].min() if len(F) else T[:, 0].min())
if len(F):
    G = F[F[:, 1] > 0]
    s_, a_, b_, c_ = (G[:, 0] - t0) / 100, (G[:, 1] - t0) / 100, (G[:, 2] - t0) / 100, (G[:, 3] - t0) / 100
    print(f"fused records pass: {len(F)} waves, {len(G)} with items; start p50 {np.median(s_):.2f} max {s_.max():.2f}; first inputs landed p50 {np.median(a_):.2f} max {a_.max():.2f}; "
          f"stores issued p50 {np.median(b_):.2f} max {b_.max():.2f}; stores acknowledged p50 {np.median(c_):.2f} max {c_.max():.2f}")
names = ["start", "row landed", "list built", "-", "filter done", "-", "first records landed", "groups done", "claim + next row", "stored"]
X = (T[:, :10] - t0) / 100.0
print("units", len(T), "kernel span us %.2f" % X[:, 9].max(), "groups per unit %.2f" % T[:, 10].mean(), "list length %.1f (max %d)" % (T[:, 11].mean(), T[:, 11].max()))
prev = X[:, 0]
for k in (1, 2, 4, 6, 7, 8, 9):
    cur = np.where(T[:, k] > 0, X[:, k], prev)
    d = cur - prev
    print(f"  {names[k]:24s} +{d.mean():6.2f} us (p50 {np.median(d):5.2f}, p90 {np.percentile(d, 90):5.2f}, max {d.max():5.2f})")
    prev = cur
tot = X[:, 9] - X[:, 0]
print("  unit total mean %.2f p50 %.2f p90 %.2f max %.2f" % (tot.mean(), np.median(tot), np.percentile(tot, 90), tot.max()))
st = np.sort(X[:, 0]); en = np.sort(X[:, 9])
print("first unit starts: min %.2f p50 of first 2048 %.2f; last unit starts %.2f; ends p50 %.2f p90 %.2f p99 %.2f max %.2f" % (st[0], np.median(st[:2048]), st[-1], np.median(en), np.percentile(en, 90), np.percentile(en, 99), en.max()))
print("sum of unit times / 2048 slots = %.2f us" % (tot.sum() / 2048))
