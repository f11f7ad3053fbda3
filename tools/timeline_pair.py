"""Debug: per-unit, per-wave timeline of the pair render kernel (-DGF_TIMELINE=1 build).  python tools/timeline_pair.py [config]"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from gaussianformer_amd import build as _b
_tl = os.path.join(_b.CSRC, "libgf_hip_timeline.so")
_deps = [os.path.join(_b.CSRC, f) for f in _b.SOURCES + _b.HEADERS]
if not os.path.exists(_tl) or any(os.path.getmtime(d) > os.path.getmtime(_tl) for d in _deps if os.path.exists(d)):
    _b.build(extra_flags=("-DGF_TIMELINE=1",), lib_name="libgf_hip_timeline.so")
os.environ["GF_LIB"] = _tl
from gaussianformer_amd import _lib
from gaussianformer_amd.local_aggregate import SplatForwardPlan
from gaussianformer_amd.synthetic import make_splat_inputs
import oracle
config = sys.argv[1] if len(sys.argv) > 1 else "nuscenes_gs25600_solid"
dev = torch.device("cuda:0")
si = make_splat_inputs(config, seed=0)
pi, mi, radii, cov6 = oracle.prepare_splat_inputs(si.pts, si.means3D, si.scales, si.cov3D, si.pc_min, si.grid_size, si.scale_multiplier)
t = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (si.pts, pi, si.means3D, mi, si.opacities, si.semantics, radii, cov6)]
plan = SplatForwardPlan(0, *t, si.H, si.W, si.D, flags=1)
lib = _lib.load()
for _ in range(5): plan.run()
torch.cuda.synchronize()
nu = 8 * ((si.H + 7) // 8) * ((si.W + 7) // 8) * 4 * ((si.D + 7) // 8) // 8 + 64
tl = torch.zeros(24 * nu, dtype=torch.int64, device=dev)
lib.gf_debug_set_timeline.argtypes = [ctypes.c_void_p]
lib.gf_debug_set_timeline(tl.data_ptr())
plan.run(); torch.cuda.synchronize()
lib.gf_debug_set_timeline(None)
T = tl.cpu().numpy().reshape(nu, 2, 12).astype(np.float64)
os.makedirs("gpurun_out", exist_ok=True)
np.save("gpurun_out/timeline_pair_%s.npy" % config, T)
T = T[T[:, 0, 0] > 0]
t0 = T[:, :, 0].min()
names = ["start", "row landed (w0)", "list built (w0)", "after A", "filter done", "after B", "first records landed", "groups done", "claim + next row (w0)", "stored"]
print("units", len(T), "kernel span us %.2f" % ((T[:, :, 9].max() - t0) / 100.0), "groups per brick %.2f / %.2f" % (T[:, 0, 10].mean(), T[:, 1, 10].mean()),
      "list length %.1f (max %d)" % (T[:, 0, 11].mean(), T[:, 0, 11].max()))
for w in range(2):
    X = (T[:, w, :10] - t0) / 100.0
    prev = X[:, 0]
    print("wave", w)
    for k in range(1, 10):
        cur = np.where(T[:, w, k] > 0, X[:, k], prev)
        d = cur - prev
        print(f"  {names[k]:24s} +{d.mean():6.2f} us (p50 {np.median(d):5.2f}, p90 {np.percentile(d, 90):5.2f}, max {d.max():5.2f})")
        prev = cur
    tot = X[:, 9] - X[:, 0]
    print("  unit total mean %.2f p50 %.2f p90 %.2f max %.2f" % (tot.mean(), np.median(tot), np.percentile(tot, 90), tot.max()))
X = (T[:, 0, :10] - t0) / 100.0
st = np.sort(X[:, 0]); en = np.sort(X[:, 9])
nslots = 2048
print("starts: first %d by %.2f us; last unit starts %.2f; ends p50 %.2f p90 %.2f p99 %.2f max %.2f" % (nslots, st[min(nslots - 1, len(st) - 1)], st[-1], np.median(en), np.percentile(en, 90), np.percentile(en, 99), en.max()))
print("sum of unit times (wave 0) / %d slots = %.2f us" % (nslots, (X[:, 9] - X[:, 0]).sum() / nslots))
