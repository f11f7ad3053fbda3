"""The HIP path, through the C ABI, against the frozen fixtures of REFERENCE outputs (tests/golden/*.npz, produced by
tools/make_golden.py from oracle/_ref = the reference's own kernels on an MI355X).  Needs only the committed fixtures:
neither /root/reference nor oracle/_ref at run time."""
import os
from types import SimpleNamespace

import numpy as np
import pytest

from util import assert_grad_close, assert_logits_close, hip_splat_backward, hip_splat_forward, to_dev

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["splat_base", "splat_base_signed", "splat_prob", "splat_prob_fast"])
def test_splat_matches_reference_fixture(gpu, name):
    import torch
    from gaussianformer_amd import _lib
    from gaussianformer_amd.local_aggregate import splat_box_volumes
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    assert "oracle/_ref" in str(d["producer"])
    si = SimpleNamespace(variant=str(d["variant"]), pts=d["pts"], means3D=d["means3D"], opacities=d["opacities"],
                         semantics=d["semantics"], H=int(d["H"]), W=int(d["W"]), D=int(d["D"]))
    pi, mi, radii, cov6 = d["points_int"], d["means_int"], d["radii"], d["cov6"]
    prob = si.variant == "prob"
    # default flags (base variant on the dense grid: the matrix-core kernel, north_star's 1e-4 bound), the exact-fp32
    # tile kernel and the arbitrary-points kernel (both 1e-5)
    for flags, tol in ((0, 1e-5 if prob else 1e-4), (_lib.GF_EXACT_FP32, 1e-5), (_lib.GF_PTS_GENERAL, 1e-5)):
        got, t, state, fwd_t = hip_splat_forward(gpu, si, pi, mi, radii, cov6, flags=flags)
        for k in (("logits", "bin_logits", "density", "probability") if prob else ("logits",)):
            assert_logits_close(got[k], d[k], what=f"{name}: {k} vs reference fixture (flags {flags})", tol=tol)
        grads = hip_splat_backward(gpu, si, t, state, fwd_t, d["out_grad"], d["bin_grad"] if prob else None,
                                   d["density_grad"] if prob else None, flags=flags)
        for k, g in zip(("means3D_grad", "opacity_grad", "semantics_grad", "cov3D_grad"), grads):
            assert_grad_close(g, d[k], what=f"{name}: {k} vs reference fixture", rtol=1e-4)
    vols, R = splat_box_volumes(torch.from_numpy(mi).to(gpu), torch.from_numpy(radii).to(gpu), si.H, si.W, si.D)
    assert R == int(d["num_rendered"]) and np.array_equal(vols.cpu().numpy().astype(np.uint32), d["tiles_touched"])


def test_daf_matches_reference_fixture(gpu):
    import torch
    from gaussianformer_amd.deformable_aggregation import DeformableAggregationFunction as DAF
    d = np.load(os.path.join(GOLDEN, "daf.npz"))
    assert "oracle/_ref" in str(d["producer"])
    feat, ss, st, loc, w = to_dev(gpu, d["mc_ms_feat"], d["spatial_shape"], d["scale_start_index"],
                                  d["sampling_location"], d["weights"])
    feat.requires_grad_(True); loc.requires_grad_(True); w.requires_grad_(True)
    out = DAF.apply(feat, ss, st, loc, w)
    assert_logits_close(out.detach().cpu().numpy(), d["output"], what="daf output vs reference fixture", tol=1e-5)
    out.backward(torch.from_numpy(d["grad_output"]).to(gpu))
    assert_grad_close(feat.grad.cpu().numpy(), d["grad_mc_ms_feat"], "grad_mc_ms_feat vs reference fixture", rtol=1e-4)
    assert_grad_close(loc.grad.cpu().numpy(), d["grad_sampling_location"], "grad_sampling_location vs reference fixture", rtol=1e-4)
    assert_grad_close(w.grad.cpu().numpy(), d["grad_weights"], "grad_weights vs reference fixture", rtol=1e-4)
