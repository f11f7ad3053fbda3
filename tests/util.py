"""Shared helpers for the parity tests: run the same seeded inputs through the CPU oracle
and through the HIP path (via the C ABI), and compare with the tolerances of SURVEY.md §8c:
integer path bit-exact, fp32 logits |err| <= 1e-4 * max(1, |ref|), gradients <= 1e-3 relative
(to the largest reference magnitude of the tensor)."""
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
