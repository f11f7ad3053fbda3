"""Pins the CPU oracle of the deformable aggregation against an independent fp64
``grid_sample`` formulation (values and, through autograd, all three gradients), the
operator's edge rules, and the frozen fixture of reference outputs."""
import os

import numpy as np
import torch

import oracle
from oracle import dense_ref
from gaussianformer_amd.synthetic import make_daf_inputs

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _case():
    d = make_daf_inputs(num_pts=50, seed=2, B=2, cams=3, C=16, G=4, levels=((6, 9), (3, 5), (2, 2)))
    d["sampling_location"][0, 0, 0] = [0.0, 0.5]       # on the gate: camera skipped (strict 0 < loc < 1)
    d["sampling_location"][0, 1, 0] = [0.999, 0.001]   # taps outside the map are zero
    d["sampling_location"][0, 2, 1] = [0.03, 0.97]
    return d


def test_oracle_matches_grid_sample_fp64():
    d = _case()
    out = oracle.daf_forward(**d)
    t = lambda a, g=False: torch.tensor(a, dtype=torch.float64, requires_grad=g)
    f, l, w = t(d["mc_ms_feat"], True), t(d["sampling_location"], True), t(d["weights"], True)
    ref = dense_ref.daf_dense(f, d["spatial_shape"], d["scale_start_index"], l, w)
    assert np.abs(ref.detach().numpy() - out).max() < 5e-6
    g = np.random.default_rng(0).standard_normal(out.shape).astype(np.float32)
    (ref * torch.tensor(g)).sum().backward()
    gf, gl, gw = oracle.daf_backward(d["mc_ms_feat"], d["spatial_shape"], d["scale_start_index"],
                                     d["sampling_location"], d["weights"], g)
    for name, a, b in (("feat", gf, f.grad), ("loc", gl, l.grad), ("weights", gw, w.grad)):
        b = b.numpy()
        assert np.abs(a - b).max() < 2e-5 * max(1.0, np.abs(b).max()), name


def test_gate_and_multithreading():
    d = _case()
    out = oracle.daf_forward(**d, nthreads=1)
    assert np.array_equal(out, oracle.daf_forward(**d, nthreads=4))
    # all cameras out of view -> exactly zero output and zero gradients
    d["sampling_location"][:] = 1.5
    assert np.all(oracle.daf_forward(**d) == 0)
    gf, gl, gw = oracle.daf_backward(d["mc_ms_feat"], d["spatial_shape"], d["scale_start_index"],
                                     d["sampling_location"], d["weights"], np.ones_like(out))
    assert not gf.any() and not gl.any() and not gw.any()


def test_golden_fixture():
    """tests/golden/daf.npz holds the REFERENCE's outputs (oracle/_ref run on an MI355X by tools/make_golden.py):
    the C restatement must reproduce them to fp32 rounding (the gradients are float-atomic sums there)."""
    d = np.load(os.path.join(GOLDEN, "daf.npz"))
    assert "oracle/_ref" in str(d["producer"])
    ins = {k: d[k] for k in ("mc_ms_feat", "spatial_shape", "scale_start_index", "sampling_location", "weights")}
    np.testing.assert_allclose(oracle.daf_forward(**ins), d["output"], rtol=1e-5, atol=1e-6)
    gf, gl, gw = oracle.daf_backward(*ins.values(), d["grad_output"])
    np.testing.assert_allclose(gf, d["grad_mc_ms_feat"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(gl, d["grad_sampling_location"], rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(gw, d["grad_weights"], rtol=1e-5, atol=2e-6)
