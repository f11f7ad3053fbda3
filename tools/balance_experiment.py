"""Scheduling experiment for the render kernel (debug build): what would a cost-balanced workgroup -> tile
assignment buy?  Per-tile costs (wave iterations, from the boxes) are computed on the host, tiles are dealt to
the 256 CUs by longest-processing-time-first (workgroup b runs on CU slot b mod 256 -- measured), and the kernel is
timed with hipEvents with and without the permutation.  Not part of the product path."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from gaussianformer_amd import build as _b
os.environ["GF_LIB"] = _b.build(extra_flags=("-DGF_TIMELINE=1",), lib_name="libgf_hip_timeline.so")
from gaussianformer_amd import _lib
from gaussianformer_amd.local_aggregate import SplatForwardPlan
from gaussianformer_amd.synthetic import make_splat_inputs
import oracle

config = sys.argv[1] if len(sys.argv) > 1 else "nuscenes_gs25600_solid"
dev = torch.device("cuda:0")
si = make_splat_inputs(config, seed=0)
pi, mi, radii, cov6 = oracle.prepare_splat_inputs(si.pts, si.means3D, si.scales, si.cov3D, si.pc_min, si.grid_size, si.scale_multiplier)
H, W, D = si.H, si.W, si.D
r = radii.reshape(-1, 1) if radii.ndim == 1 else radii
lo = np.maximum(mi - r, 0); hi = np.minimum(mi + r + 1, [H, W, D])
ok = np.all(hi > lo, axis=1); lo, hi = lo[ok], hi[ok]
nsx, nsy = (H + 7) // 8, (W + 7) // 8
ntx, nty = nsx, nsy * 2
iters = np.zeros((ntx, nty), dtype=np.int64)
for (x0, y0, z0), (x1, y1, z1) in zip(lo, hi):
    x1 -= 1; y1 -= 1; z1 -= 1
    for xh in range(x0 // 4, x1 // 4 + 1):
        iters[xh // 2, y0 // 4:y1 // 4 + 1] += (z1 // 8) - (z0 // 8) + 1
ntiles = nsx * nsy * 2
grid = (ntiles + 7) // 8 * 8
cost = np.zeros(grid)
for logical in range(ntiles):
    s, t = divmod(logical, 2)
    cost[logical] = iters[s // nsy, (s % nsy) * 2 + t]
# LPT onto CU slots: block b -> slot b % 256; slot c holds blocks c, c + 256, ...
nslots = 256
cap = np.array([len(range(c, grid, nslots)) for c in range(nslots)])
load = np.zeros(nslots); used = np.zeros(nslots, dtype=int)
perm = np.full(grid, ntiles, dtype=np.int32)  # filler blocks get an out-of-range tile (they exit)
for logical in np.argsort(-cost[:ntiles], kind="stable"):
    c = int(np.argmin(load + (used >= cap) * 1e18))
    perm[c + nslots * used[c]] = logical
    load[c] += cost[logical]; used[c] += 1
print("tiles", ntiles, "grid", grid, "slot load max/mean: default %.3f  LPT %.3f" % (
    max(cost[[ (b & 7) * (grid // 8) + (b >> 3) for b in range(c, grid, nslots)]].sum() for c in range(nslots)) / (cost.sum() / nslots),
    load.max() / load.mean()))

t = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (si.pts, pi, si.means3D, mi, si.opacities, si.semantics, radii, cov6)]
plan = SplatForwardPlan(0, *t, H, W, D, flags=1)
lib = _lib.load()
lib.gf_debug_set_tile_perm.argtypes = [ctypes.c_void_p]
ref = None
perm_d = torch.from_numpy(perm).to(dev)
for name, ptr in (("default mapping", None), ("LPT permutation", perm_d.data_ptr()), ("default mapping", None), ("LPT permutation", perm_d.data_ptr())):
    lib.gf_debug_set_tile_perm(ptr)
    for _ in range(10): plan.run()
    torch.cuda.synchronize()
    _lib.check(lib.gf_profile_stride(1), "stride"); _lib.check(lib.gf_profile_enable(200), "enable")
    for _ in range(200): plan.run()
    torch.cuda.synchronize()
    buf = (ctypes.c_float * 200)(); n = lib.gf_profile_read(buf, 200); lib.gf_profile_enable(0)
    print(f"{name:18s}: render kernel {np.mean(buf[:n]) * 1e3:7.2f} us (hipEvents, {n} launches)")
    out = plan.logits.clone()
    if ref is None: ref = out
    else: print("   logits identical to the default mapping:", bool(torch.equal(out, ref)))
lib.gf_debug_set_tile_perm(None)
