"""Parity figures of the three BASELINE shapes against the reference's own kernels (oracle/_ref), in north_star's own terms:
ABSOLUTE error of the fp32 outputs (bound 1e-4) and gradients ROW BY ROW (every Gaussian against its own magnitude, floor = the
median ordinary row, the whole-grid "empty" Gaussian judged by itself; bound 1e-3) -- next to the scaled / tensor-wide
figures earlier rounds reported.   python tools/parity_report.py > profiles/parity_r06.txt      (GPU box, oracle/_ref built)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gaussianformer_amd import _lib
from gaussianformer_amd.synthetic import make_splat_inputs
from oracle import ref
from util import grad_row_errors, hip_splat_backward, hip_splat_forward, prep, whole_grid_rows

dev = torch.device("cuda:0")
assert ref.available(), "oracle/_ref is not built"
NAMES = ("means3D_grad", "opacity_grad", "semantics_grad", "cov3D_grad")


def report(config, per_axis=False, flags=0, tag=""):
    si = make_splat_inputs(config, seed=0)
    pi, mi, radii, cov6 = prep(si, per_axis)
    rng = np.random.default_rng(1)
    N = si.pts.shape[0]
    g = rng.standard_normal((N, 18)).astype(np.float32)
    gb = gd = None
    if si.variant == "prob":
        gb, gd = rng.standard_normal(N).astype(np.float32), rng.standard_normal(N).astype(np.float32)
    rf, rgrads, _ = ref.splat_forward_backward(si.variant, si.pts, pi, si.means3D, mi, si.opacities, si.semantics, radii, cov6,
                                               si.H, si.W, si.D, g, gb, gd)
    got, t, state, fwd_t = hip_splat_forward(dev, si, pi, mi, radii, cov6, flags=flags)
    words = state.view(torch.int32)[:3].tolist()
    print(f"== {config}{tag}: P = {si.means3D.shape[0]}, R = {rf['num_rendered']}, state words (general pts, path, verdicts) {words}")
    for k in (("logits", "bin_logits", "density", "probability") if si.variant == "prob" else ("logits",)):
        fin = np.isfinite(rf[k]) if rf[k].ndim == 1 else np.isfinite(rf[k]).all(axis=1)
        a, b = got[k][fin].astype(np.float64), rf[k][fin].astype(np.float64)
        err = np.abs(a - b)
        print(f"   {k:12s} max ABSOLUTE err {err.max():.3e}   scaled |err|/max(1,|ref|) {(err / np.maximum(1, np.abs(b))).max():.3e}"
              f"   max|ref| {np.abs(b).max():.3e}   reference non-finite voxels {int((~fin).sum())}")
    grads = hip_splat_backward(dev, si, t, state, fwd_t, g, gb, gd, flags=flags)
    whole = whole_grid_rows(mi, radii, si.H, si.W, si.D)
    for name, a, b in zip(NAMES, grads, rgrads):
        ok = np.isfinite(b) if b.ndim == 1 else np.isfinite(b).all(axis=1)
        e = grad_row_errors(a[ok], b[ok], whole[ok])
        print(f"   {name:15s} worst ordinary row {e['ordinary']:.3e} (floor {e['floor']:.3e}, absolute {e['abs']:.3e}, row {e['worst_row']})"
              f"   whole-grid row {e['whole_grid']:.3e}   tensor-wide {e['tensor']:.3e}   whole-grid rows {int(whole.sum())}")


report("nuscenes_gs25600_solid")
report("nuscenes_gs25600_solid", flags=_lib.GF_EXACT_FP32, tag=" (GF_EXACT_FP32)")
report("nuscenes_gs144000")
report("nuscenes_gs144000", flags=_lib.GF_EXACT_FP32, tag=" (GF_EXACT_FP32)")
report("prob_gs6400")
report("prob_gs6400", per_axis=True, tag=" (prob_fast radii)")
