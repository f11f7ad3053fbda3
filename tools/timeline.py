"""Debug: per-workgroup timeline of the TILE render kernels (start / list built / consumed / stored).
python tools/timeline.py [config] [exact]   -- default: the matrix-core tile kernel (forced with library option "splat.mfma_tile_kernel": rows of <= 618
bitmask words otherwise go to the wave-autonomous kernel, whose timeline is tools/timeline_wave.py); "exact": the exact-fp32 tile kernel"""
import ctypes, sys
import numpy as np, torch
sys.path.insert(0, ".")
import os
from gaussianformer_amd import build as _b
_tl = os.path.join(_b.CSRC, "libgf_hip_timeline.so")
_deps = [os.path.join(_b.CSRC, f) for f in _b.SOURCES + _b.HEADERS]
if not os.path.exists(_tl) or any(os.path.getmtime(d) > os.path.getmtime(_tl) for d in _deps if os.path.exists(d)):
    _b.build(extra_flags=("-DGF_TIMELINE=1",), lib_name="libgf_hip_timeline.so")   # prebuilt in-tree copies travel with gpurun
os.environ["GF_LIB"] = _tl
from gaussianformer_amd import _lib
_lib.set_option("splat.mfma_tile_kernel", 1)
from gaussianformer_amd.local_aggregate import SplatForwardPlan
from gaussianformer_amd.synthetic import make_splat_inputs
import oracle
config = sys.argv[1] if len(sys.argv) > 1 else "nuscenes_gs25600_solid"
extra_flags = _lib.GF_EXACT_FP32 if "exact" in sys.argv[2:] else 0   # default = the matrix-core kernel
dev = torch.device("cuda:0")
si = make_splat_inputs(config, seed=0)
pi, mi, radii, cov6 = oracle.prepare_splat_inputs(si.pts, si.means3D, si.scales, si.cov3D, si.pc_min, si.grid_size, si.scale_multiplier)
t = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (si.pts, pi, si.means3D, mi, si.opacities, si.semantics, radii, cov6)]
plan = SplatForwardPlan(0, *t, si.H, si.W, si.D, flags=1 | extra_flags)
lib = _lib.load()
for _ in range(5): plan.run()
torch.cuda.synchronize()
nb = 1256
tl = torch.zeros(nb * 5 + nb * 24, dtype=torch.int64, device=dev)
lib.gf_debug_set_timeline.argtypes = [ctypes.c_void_p]
lib.gf_debug_set_timeline(tl.data_ptr())
plan.run(); torch.cuda.synchronize()
lib.gf_debug_set_timeline(None)
raw = tl.cpu().numpy()
ids = raw[4 * nb:5 * nb]
extra = raw[5 * nb:].reshape(nb, 4, 6).astype(np.float64)   # per (tile, wave): cycles waiting for records, operand prep, blocks, groups, tile total
T = raw[:4 * nb].reshape(nb, 4).astype(np.float64)
valid = T[:, 0] > 0
ids = ids[valid]
extra = extra[valid]
T = T[T[:, 0] > 0]
t0 = T[:, 0].min()
T = (T - t0) / 100.0  # 100 MHz -> us
os.makedirs("gpurun_out", exist_ok=True)
np.savez("gpurun_out/timeline_blocks_%s.npz" % config, block=np.nonzero(valid)[0], ids=ids, T=T)
print("blocks", len(T), "kernel span us", T[:, 3].max())
for name, col in (("start", 0), ("list built", 1), ("consumed", 2), ("end", 3)):
    v = T[:, col]; print(f"{name:12s} min {v.min():7.2f} p50 {np.median(v):7.2f} p90 {np.percentile(v,90):7.2f} max {v.max():7.2f}")
g = extra[..., 3]
if g.sum() > 0:
    tot = extra[..., :3].sum(axis=(0, 1)); ng = g.sum()
    print("matrix-core kernel, per group of 32 hits (shader cycles incl. ~4 s_memtime reads): records wait %.0f  operand prep %.0f  blocks (exp + MFMA issue) %.0f  | groups %.0f, per (tile, wave) %.2f" %
          (tot[0] / ng, tot[1] / ng, tot[2] / ng, ng, ng / max((g > 0).sum(), 1)))
    tt = extra[..., 4][g > 0]
    print("   tile time per wave (cycles): mean %.0f;  in-group share %.2f;  wait for the output stores to be acknowledged: mean %.0f" % (tt.mean(), extra[..., :3].sum() / tt.sum(), extra[..., 5][g > 0].mean()))
d_prod = T[:, 1] - T[:, 0]; d_cons = T[:, 2] - T[:, 1]; d_epi = T[:, 3] - T[:, 2]
for name, v in (("produce", d_prod), ("consume", d_cons), ("epilogue", d_epi), ("total", T[:, 3] - T[:, 0])):
    print(f"dur {name:9s} mean {v.mean():7.2f} p50 {np.median(v):7.2f} p90 {np.percentile(v,90):7.2f} max {v.max():7.2f}")
order = np.argsort(T[:, 0])
first = T[order[:2048]]; late = T[order[2048:]]
print("first-wave blocks: mean total", (first[:, 3]-first[:, 0]).mean(), " late blocks:", len(late), "mean start", late[:, 0].mean() if len(late) else None, "mean total", (late[:, 3]-late[:, 0]).mean() if len(late) else None)
hist, edges = np.histogram(T[:, 0], bins=12); print("start hist", hist.tolist(), [round(e,1) for e in edges.tolist()])

hw = ids & 0xFFFFFFFF; xcc = (ids >> 32) & 0xF
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
uniq, cnt = np.unique(key, return_counts=True)
print("distinct CUs used", len(uniq), "blocks per CU: min", cnt.min(), "max", cnt.max(), "hist", np.bincount(cnt).tolist())
endt = T[:, 3]
per_cu_end = {k: endt[key == k].max() for k in uniq}
ends = np.array(list(per_cu_end.values()))
print("per-CU finish time us: min %.1f p50 %.1f max %.1f" % (ends.min(), np.median(ends), ends.max()))
for c in sorted(set(cnt.tolist())):
    sel = [per_cu_end[k] for k, n in zip(uniq, cnt) if n == c]
    print("  CUs with %d blocks: %d, mean finish %.1f us" % (c, len(sel), np.mean(sel)))
