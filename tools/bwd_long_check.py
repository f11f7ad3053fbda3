"""Development probe (GPU box): the matrix-core backward on long rows (P > 39 552) against the exact Gaussian-major kernels -- per-row
errors, which path ran, times.  python tools/bwd_long_check.py"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gaussianformer_amd import _lib
from gaussianformer_amd.local_aggregate import splat_backward, splat_forward
from gaussianformer_amd.synthetic import make_splat_inputs
from util import prep, to_dev, grad_row_errors, whole_grid_rows
dev = torch.device("cuda:0")
cases = [("nuscenes_gs144000", {}), ("nuscenes_gs144000", dict(P=50000, H=24, W=24, D=16)), ("nuscenes_gs144000", dict(P=40000, H=64, W=40, D=8)),
         ("nuscenes_gs25600_solid", dict(P=45000, H=16, W=16, D=8))]
for cfg, kw in cases:
    si = make_splat_inputs(cfg, seed=3, **kw)
    pi, mi, radii, cov6 = prep(si)
    t = to_dev(dev, si.pts, pi, si.means3D, mi, si.opacities, si.semantics, radii, cov6)
    N = si.pts.shape[0]
    g = torch.randn(N, 18, generator=torch.Generator().manual_seed(1)).to(dev)
    logits, _, _, _, state = splat_forward(_lib.GF_SPLAT_BASE, *t, si.H, si.W, si.D, flags=_lib.GF_PREPARE_BACKWARD)
    torch.cuda.synchronize()
    words = state.view(torch.int32)[:5].tolist()
    fast = words[0] == 0 and words[1] in _lib.GF_PATHS_MATRIX_CORE and (words[4] & 1)
    mc = splat_backward(_lib.GF_SPLAT_BASE, *t, si.H, si.W, si.D, g, state=state, flags=(_lib.GF_MFMA_SPLAT | _lib.GF_RECORDS_VALID) if fast else 0)
    mc = [x.clone() for x in mc]
    def timed(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
    # (timed BEFORE the exact pipeline runs: that one uses the workspace, after which a backward that vouches for the forward's records
    # -- GF_RECORDS_VALID -- stands down and returns NaN)
    t_mc = timed(lambda: splat_backward(_lib.GF_SPLAT_BASE, *t, si.H, si.W, si.D, g, state=state, flags=(_lib.GF_MFMA_SPLAT | _lib.GF_RECORDS_VALID) if fast else 0))
    ex = splat_backward(_lib.GF_SPLAT_BASE, *t, si.H, si.W, si.D, g, state=state, flags=_lib.GF_EXACT_FP32)
    torch.cuda.synchronize()
    whole = whole_grid_rows(mi, radii, si.H, si.W, si.D)
    errs = {}
    for name, a, b in zip(("means", "opacity", "semantics", "cov"), mc, ex):
        e = grad_row_errors(a.cpu().numpy(), b.cpu().numpy(), whole)
        errs[name] = (round(e["ordinary"], 7), round(e.get("whole_grid", 0.0) or 0.0, 7))
    def fb():
        lg, _, _, _, st = splat_forward(_lib.GF_SPLAT_BASE, *t, si.H, si.W, si.D, flags=_lib.GF_PREPARE_BACKWARD)
        return splat_backward(_lib.GF_SPLAT_BASE, *t, si.H, si.W, si.D, g, state=st, flags=(_lib.GF_MFMA_SPLAT | _lib.GF_RECORDS_VALID) if fast else 0)
    t_ex = timed(lambda: splat_backward(_lib.GF_SPLAT_BASE, *t, si.H, si.W, si.D, g, state=state, flags=_lib.GF_EXACT_FP32))
    t_fb = timed(fb)
    print(cfg, kw, "state", words, "fast", bool(fast), "row errors (ordinary, whole-grid) vs exact:", errs,
          f"backward {t_mc:.1f} us (exact {t_ex:.1f}), forward(prepared) + backward {t_fb:.1f} us", flush=True)
