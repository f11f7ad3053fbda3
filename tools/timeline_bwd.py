"""Debug: per-unit timeline of the matrix-core backward kernel (-DGF_TIMELINE=1 build).  python tools/timeline_bwd.py [config]"""
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
from gaussianformer_amd.local_aggregate import splat_backward, splat_forward
from gaussianformer_amd.synthetic import make_splat_inputs
import oracle
config = sys.argv[1] if len(sys.argv) > 1 else "nuscenes_gs25600_solid"
dev = torch.device("cuda:0")
si = make_splat_inputs(config, seed=0)
pi, mi, radii, cov6 = oracle.prepare_splat_inputs(si.pts, si.means3D, si.scales, si.cov3D, si.pc_min, si.grid_size, si.scale_multiplier)
t = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (si.pts, pi, si.means3D, mi, si.opacities, si.semantics, radii, cov6)]
logits, _, _, _, state = splat_forward(0, *t, si.H, si.W, si.D, flags=_lib.GF_PREPARE_BACKWARD)
g = torch.randn(logits.shape, generator=torch.Generator().manual_seed(1)).to(dev)
lib = _lib.load()
run = lambda: splat_backward(0, *t, si.H, si.W, si.D, g, state=state, flags=_lib.GF_MFMA_SPLAT | _lib.GF_RECORDS_VALID)
for _ in range(5): run()
torch.cuda.synchronize()
nu = 8 * ((si.H + 7) // 8) * ((si.W + 7) // 8) * 4 * ((si.D + 7) // 8) // 8 + 64
tl = torch.zeros(8 * nu, dtype=torch.int64, device=dev)
lib.gf_debug_set_bwd_timeline.argtypes = [ctypes.c_void_p]
lib.gf_debug_set_bwd_timeline(tl.data_ptr())
run(); torch.cuda.synchronize()
lib.gf_debug_set_bwd_timeline(None)
T = tl.cpu().numpy().reshape(nu, 8).astype(np.float64)
os.makedirs("gpurun_out", exist_ok=True)
np.save(f"gpurun_out/timeline_bwd_{config}.npy", tl.cpu().numpy().reshape(nu, 8))
T = T[T[:, 0] > 0]
groups = T[:, 7].copy()
t0 = T[:, 0].min()
T = (T[:, :7] - t0) / 100.0
print("units", len(T), "kernel span us %.2f" % T[:, 6].max(), " groups per unit %.2f" % groups.mean())
names = ["row + dL landed", "list built", "boxes landed, dL staged", "first records landed", "groups done", "claim answered"]
prev = T[:, 0]
for k, nme in enumerate(names, 1):
    cur = np.where(T[:, k] > 0, T[:, k], prev)
    d = cur - prev
    print(f"{nme:26s} +{d.mean():6.2f} us (p50 {np.median(d):5.2f}, p90 {np.percentile(d, 90):5.2f}, max {d.max():5.2f})")
    prev = cur
tot = T[:, 6] - T[:, 0]
print("unit total mean %.2f p50 %.2f p90 %.2f max %.2f" % (tot.mean(), np.median(tot), np.percentile(tot, 90), tot.max()))
gd = (T[:, 5] - np.where(T[:, 4] > 0, T[:, 4], T[:, 5]))
print("per group (groups phase / groups): %.2f us" % (gd.sum() / max(groups.sum(), 1)))
st = np.sort(T[:, 0]); en = np.sort(T[:, 6])
print("starts: first 2048 by %.2f us; last unit starts %.2f; ends p50 %.2f p90 %.2f max %.2f" % (st[min(2047, len(st) - 1)], st[-1], np.median(en), np.percentile(en, 90), en.max()))
print("sum of unit times / 2048 slots = %.2f us" % (tot.sum() / 2048.0))
