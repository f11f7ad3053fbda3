"""Shared helpers for the parity tests: run the same seeded inputs through the CPU oracle
and through the HIP path (via the C ABI), and compare with the tolerances of SURVEY.md §8c /
north_star: integer path bit-exact; fp32 logits within 1e-4 -- ``assert_logits_close`` bounds the
scaled error |err| / max(1, |ref|), ``assert_logits_abs`` the ABSOLUTE error (what north_star says;
used at the BASELINE shapes against oracle/_ref); gradients <= 1e-3 -- ``assert_grad_close``
relative to the tensor's largest magnitude (small shapes), ``assert_grad_rows_close`` ROW BY ROW
(every Gaussian against its own magnitude, with a floor taken from the ordinary rows and the
whole-grid "empty" Gaussian, gaussian_head.py:90-102, judged separately: at nuscenes_gs25600_solid
its row is 1e4 times an ordinary one and would make a tensor-wide scale vacuous)."""
import numpy as np

import oracle
from gaussianformer_amd.synthetic import make_splat_inputs

LOGIT_TOL = 1e-4
GRAD_RTOL = 1e-3


def prep(si, per_axis=False):
    radii_min = 1 if si.variant == "prob" else None
    pi, mi, radii, cov6 = oracle.prepare_splat_inputs(si.pts, si.means3D, si.scales, si.cov3D, si.pc_min,
                                                      si.grid_size, si.scale_multiplier, per_axis=per_axis,
                                                      radii_min=radii_min)
    return pi, mi, radii, cov6


def assert_logits_close(got, ref, what="logits", tol=LOGIT_TOL):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert np.isfinite(got).all(), f"{what}: non-finite values"
    err = np.abs(got - ref) / np.maximum(1.0, np.abs(ref))
    assert err.max() <= tol, f"{what}: max scaled err {err.max():.3e} > {tol} at {np.unravel_index(err.argmax(), err.shape)}"


def assert_grad_close(got, ref, what="grad", rtol=GRAD_RTOL):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert np.isfinite(got).all(), f"{what}: non-finite values"
    scale = max(np.abs(ref).max(), 1e-6)
    err = np.abs(got - ref).max() / scale
    assert err <= rtol, f"{what}: max err / max|ref| = {err:.3e} > {rtol}"


def assert_logits_abs(got, ref, what="logits", tol=LOGIT_TOL):
    """north_star: "fp32 logits within 1e-4" -- the absolute error, no scaling by |ref|."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert np.isfinite(got).all(), f"{what}: non-finite values"
    err = np.abs(got - ref)
    assert err.max() <= tol, (f"{what}: max ABSOLUTE err {err.max():.3e} > {tol} at "
                              f"{np.unravel_index(err.argmax(), err.shape)} (max|ref| {np.abs(ref).max():.3e})")
    return float(err.max())


def whole_grid_rows(mi, radii, H, W, D):
    """Mask of the Gaussians whose integer box (auxiliary.h:8-20) is the whole grid -- the appended "empty"
    Gaussian of gaussian_head.py:90-102 (scale [100, 100, 8] -> radius 600)."""
    mi = np.asarray(mi, dtype=np.int64)
    r3 = np.asarray(radii, dtype=np.int64)
    if r3.ndim == 1:
        r3 = np.repeat(r3[:, None], 3, axis=1)
    dims = np.array([H, W, D], dtype=np.int64)
    lo = np.minimum(dims, np.maximum(0, mi - r3))
    hi = np.minimum(dims, np.maximum(0, mi + r3 + 1))
    return (lo == 0).all(axis=1) & (hi == dims).all(axis=1)


def grad_row_errors(got, ref, whole=None):
    """Per-row gradient errors.  Returns a dict: ``ordinary`` = max over the ordinary rows g of
    |err_g| / max(max|ref_g|, floor) with floor = the MEDIAN of the ordinary rows' max|ref_g| (so that a row whose
    terms cancel is held to the typical row's magnitude, not to zero, and never to the whole-grid row's);
    ``whole_grid`` = the same for the whole-grid rows against their own magnitude; ``floor``; ``abs`` = the largest
    absolute error of an ordinary row; ``tensor`` = the old tensor-wide figure max|err| / max|ref|."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    P = ref.shape[0]
    g2, r2 = got.reshape(P, -1), ref.reshape(P, -1)
    whole = np.zeros(P, bool) if whole is None else np.asarray(whole, bool)
    rowerr = np.abs(g2 - r2).max(axis=1) if P else np.zeros(0)
    rowmax = np.abs(r2).max(axis=1) if P else np.zeros(0)
    out = {"ordinary": 0.0, "whole_grid": 0.0, "floor": 0.0, "abs": 0.0, "worst_row": -1,
           "tensor": float(rowerr.max() / max(rowmax.max(), 1e-30)) if P else 0.0}
    o = ~whole
    if o.any():
        floor = max(float(np.median(rowmax[o])), 1e-30)
        rel = rowerr[o] / np.maximum(rowmax[o], floor)
        out.update(ordinary=float(rel.max()), floor=floor, abs=float(rowerr[o].max()),
                   worst_row=int(np.flatnonzero(o)[rel.argmax()]))
    if whole.any():
        out["whole_grid"] = float((rowerr[whole] / np.maximum(rowmax[whole], 1e-30)).max())
    return out


def assert_grad_rows_close(got, ref, whole=None, what="grad", rtol=GRAD_RTOL):
    """Row-by-row gradient bound (see the module docstring); returns the error record."""
    assert np.asarray(got).shape == np.asarray(ref).shape, (what, np.asarray(got).shape, np.asarray(ref).shape)
    assert np.isfinite(np.asarray(got)).all(), f"{what}: non-finite values"
    e = grad_row_errors(got, ref, whole)
    assert e["ordinary"] <= rtol, (f"{what}: row {e['worst_row']}: err / max(|ref row|, floor {e['floor']:.3e}) = "
                                   f"{e['ordinary']:.3e} > {rtol} (absolute {e['abs']:.3e})")
    assert e["whole_grid"] <= rtol, f"{what}: whole-grid row: err / max|ref row| = {e['whole_grid']:.3e} > {rtol}"
    return e


def to_dev(dev, *arrays):
    import torch
    return [None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in arrays]


def hip_splat_forward(dev, si, pi, mi, radii, cov6, flags=0):
    from gaussianformer_amd import _lib
    from gaussianformer_amd.local_aggregate import splat_forward
    variant = _lib.GF_SPLAT_PROB if si.variant == "prob" else _lib.GF_SPLAT_BASE
    t = to_dev(dev, si.pts, pi, si.means3D, mi, si.opacities, si.semantics, radii, cov6)
    logits, bl, de, pr, state = splat_forward(variant, *t, si.H, si.W, si.D, flags=flags)
    out = {"logits": logits.cpu().numpy()}
    if bl is not None:
        out.update(bin_logits=bl.cpu().numpy(), density=de.cpu().numpy(), probability=pr.cpu().numpy())
    return out, t, state, (logits, bl, de, pr)


def hip_splat_backward(dev, si, t, state, fwd_t, out_grad, bin_grad=None, dens_grad=None, flags=0):
    from gaussianformer_amd import _lib
    from gaussianformer_amd.local_aggregate import splat_backward
    variant = _lib.GF_SPLAT_PROB if si.variant == "prob" else _lib.GF_SPLAT_BASE
    g, bg, dg = to_dev(dev, out_grad, bin_grad, dens_grad)
    grads = splat_backward(variant, *t, si.H, si.W, si.D, g, fwd_outputs=fwd_t if variant else None,
                           bin_logits_grad=bg, density_grad=dg, state=state, flags=flags)
    return [x.cpu().numpy() for x in grads]
