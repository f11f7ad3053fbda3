"""Pins the CPU oracle of the splat (the reference has no tests or golden vectors to pin it
with -- SURVEY.md §4): against an independent dense fp64 formulation with autograd, against
the invariants of the operator, and against the frozen fixtures in tests/golden/."""
import os

import numpy as np
import pytest
import torch

import oracle
from oracle import dense_ref
from gaussianformer_amd.synthetic import make_splat_inputs

from util import prep

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _t(a, grad=False):
    return torch.tensor(a, dtype=torch.float64, requires_grad=grad)


def _fwd(si, pi, mi, radii, cov6, **kw):
    return oracle.splat_forward(si.variant, si.pts, pi, si.means3D, mi, si.opacities, si.semantics, radii, cov6,
                                si.H, si.W, si.D, **kw)


@pytest.mark.parametrize("config,per_axis,dense_pts", [
    ("nuscenes_gs25600_solid", False, True), ("nuscenes_gs144000", False, True),
    ("prob_gs6400", False, True), ("prob_gs6400", True, True), ("nuscenes_gs25600_solid", False, False),
])
def test_oracle_matches_dense_fp64(config, per_axis, dense_pts):
    si = make_splat_inputs(config, seed=1, P=40, H=12, W=10, D=8, dense_pts=dense_pts, N=400)
    pi, mi, radii, cov6 = prep(si, per_axis)
    fwd = _fwd(si, pi, mi, radii, cov6)
    if not dense_pts:
        # voxel2pts: only one point per voxel takes part in the backward; mirror that in the
        # dense formulation by masking the upstream gradient of the losers (highest index wins)
        key = (pi[:, 0] * si.W + pi[:, 1]) * si.D + pi[:, 2]
        winner = np.zeros(len(key), bool)
        last = {}
        for n, k in enumerate(key):
            last[k] = n
        winner[list(last.values())] = True
    else:
        winner = np.ones(si.pts.shape[0], bool)
    m, o, s, c = _t(si.means3D, True), _t(si.opacities, True), _t(si.semantics, True), _t(cov6, True)
    out = dense_ref.splat_dense(si.variant, _t(si.pts), torch.tensor(pi), m, torch.tensor(mi), o, s,
                                torch.tensor(radii), c, si.H, si.W, si.D)
    rng = np.random.default_rng(5)
    N = si.pts.shape[0]
    g = rng.standard_normal((N, 18)).astype(np.float32)
    wmask = torch.tensor(winner, dtype=torch.float64)
    if si.variant == "base":
        scale = max(1.0, np.abs(fwd["logits"]).max())
        assert np.abs(out.detach().numpy() - fwd["logits"]).max() < 2e-6 * scale
        (out * _t(g) * wmask[:, None]).sum().backward()
        grads = oracle.splat_backward("base", si.pts, pi, si.means3D, mi, si.opacities, si.semantics, radii, cov6,
                                      si.H, si.W, si.D, g)
    else:
        for name, ref in zip(("logits", "bin_logits", "density", "probability"), out):
            ref = ref.detach().numpy()
            assert np.abs(ref - fwd[name]).max() < 5e-5 * max(1.0, np.abs(ref).max()), name
        gb = rng.standard_normal(N).astype(np.float32)
        gd = rng.standard_normal(N).astype(np.float32)
        ((out[0] * _t(g)).sum() + (out[1] * _t(gb)).sum() + (out[2] * _t(gd)).sum()).backward()
        grads = oracle.splat_backward("prob", si.pts, pi, si.means3D, mi, si.opacities, si.semantics, radii, cov6,
                                      si.H, si.W, si.D, g, fwd=fwd, bin_grad=gb, density_grad=gd)
    # prob config: scales span 0.01..3.2 m, so the quadratic form cancels ~1e3 -> ~1e0 and
    # fp32 (the reference's arithmetic) is itself only good to ~1e-4 relative there.
    gtol = 5e-4 if si.variant == "prob" else 5e-5
    for name, a, b in zip(("means", "opa", "sem", "cov"), grads, (m.grad, o.grad, s.grad, c.grad)):
        b = b.numpy()
        assert np.abs(a - b).max() < gtol * max(1.0, np.abs(b).max()), name


def test_integer_path_matches_torch_semantics():
    """prepare_splat_inputs == the torch expressions of the reference wrapper, bit for bit
    (fp32 subtract, fp32 true division, truncation; ceil of fp32 products)."""
    si = make_splat_inputs("nuscenes_gs25600_solid", seed=2, P=5000, dense_pts=False, N=20000)
    pi, mi, radii, cov6 = oracle.prepare_splat_inputs(si.pts, si.means3D, si.scales, si.cov3D, si.pc_min,
                                                      si.grid_size, si.scale_multiplier)
    pc_min = torch.tensor(si.pc_min, dtype=torch.float).unsqueeze(0)
    pts, means, scales = torch.from_numpy(si.pts), torch.from_numpy(si.means3D), torch.from_numpy(si.scales)
    assert torch.equal(((pts - pc_min) / si.grid_size).to(torch.int), torch.from_numpy(pi))
    assert torch.equal(((means - pc_min) / si.grid_size).to(torch.int), torch.from_numpy(mi))
    assert torch.equal(torch.ceil(scales.max(dim=-1)[0] * si.scale_multiplier / si.grid_size).to(torch.int),
                       torch.from_numpy(radii))
    assert torch.equal(torch.from_numpy(si.cov3D).flatten(1)[:, [0, 4, 8, 1, 5, 2]], torch.from_numpy(cov6))
    _, _, r3, _ = oracle.prepare_splat_inputs(si.pts, si.means3D, si.scales, si.cov3D, si.pc_min, si.grid_size, 4,
                                              per_axis=True, radii_min=1)
    assert torch.equal(torch.ceil(scales * 4 / si.grid_size).to(torch.int).clamp(min=1), torch.from_numpy(r3))
    # dense grid: point n sits in voxel n
    d = make_splat_inputs("nuscenes_gs25600_solid", seed=2, P=1)
    pi, *_ = oracle.prepare_splat_inputs(d.pts, d.means3D, d.scales, d.cov3D, d.pc_min, d.grid_size, 3)
    key = (pi[:, 0].astype(np.int64) * d.W + pi[:, 1]) * d.D + pi[:, 2]
    assert np.array_equal(key, np.arange(d.H * d.W * d.D))


def test_invariants():
    si = make_splat_inputs("nuscenes_gs25600_solid", seed=3, P=60, H=12, W=12, D=8)
    pi, mi, radii, cov6 = prep(si)
    base = _fwd(si, pi, mi, radii, cov6)["logits"]
    # linear in opacity and in semantics (power-of-two scaling is exact in fp32)
    si2 = make_splat_inputs("nuscenes_gs25600_solid", seed=3, P=60, H=12, W=12, D=8)
    si2.opacities *= np.float32(0.5)
    assert np.array_equal(_fwd(si2, pi, mi, radii, cov6)["logits"], base * np.float32(0.5))
    # permutation of the Gaussians changes only the summation order
    perm = np.random.default_rng(0).permutation(si.means3D.shape[0])
    out_p = oracle.splat_forward("base", si.pts, pi, si.means3D[perm], mi[perm], si.opacities[perm],
                                 si.semantics[perm], radii[perm], cov6[perm], si.H, si.W, si.D)["logits"]
    assert np.abs(out_p - base).max() < 1e-5 * max(1, np.abs(base).max())
    # a voxel no box covers is exactly 0 (base) / uniform 1/17 with channel 17 = 0 (prob)
    one = make_splat_inputs("nuscenes_gs144000", seed=4, P=1, H=16, W=16, D=8)
    pi1, mi1, r1, c1 = prep(one)
    o = oracle.splat_forward("base", one.pts, pi1, one.means3D, mi1, one.opacities, one.semantics, r1, c1, 16, 16, 8)
    inside = np.all((pi1 >= mi1[0] - r1[0]) & (pi1 <= mi1[0] + r1[0]), axis=1)
    assert np.all(o["logits"][~inside] == 0) and np.all(np.abs(o["logits"][inside]).sum(1) > 0)
    assert o["num_rendered"] == inside.sum()
    op = oracle.splat_forward("prob", one.pts, pi1, one.means3D, mi1, one.opacities, np.abs(one.semantics), r1, c1, 16, 16, 8)
    assert np.all(op["logits"][~inside, :17] == np.float32(1.0 / 17)) and np.all(op["logits"][~inside, 17] == 0)
    assert np.all(op["bin_logits"][~inside] == 0) and np.all(op["density"][~inside] == 0)
    # scalar radius == per-axis radius when the three radii are equal
    sp = make_splat_inputs("prob_gs6400", seed=5, P=20, H=12, W=12, D=8)
    pi, mi, radii, cov6 = prep(sp)
    a = _fwd(sp, pi, mi, radii, cov6)
    b = _fwd(sp, pi, mi, np.repeat(radii[:, None], 3, axis=1), cov6)
    assert all(np.array_equal(a[k], b[k]) for k in ("logits", "bin_logits", "density", "probability"))


def test_whole_grid_gaussian_and_clipping():
    si = make_splat_inputs("nuscenes_gs25600_solid", seed=6, P=0, H=10, W=9, D=8)  # only the "empty" Gaussian
    pi, mi, radii, cov6 = prep(si)
    assert radii[0] == 600
    touched, offsets, R = oracle.box_offsets(mi, radii, si.H, si.W, si.D)
    assert R == si.H * si.W * si.D == touched[0] == offsets[-1]
    out = _fwd(si, pi, mi, radii, cov6)
    assert out["num_rendered"] == R and np.all(out["logits"][:, 17] > 0) and np.all(out["logits"][:, :17] == 0)
    # box clipping at the grid border: a corner Gaussian with radius 2 covers 3x3x3 voxels
    t, _, R = oracle.box_offsets(np.array([[0, 0, 0]], np.int32), np.array([2], np.int32), 10, 9, 8)
    assert t[0] == 27 and R == 27
    t, _, _ = oracle.box_offsets(np.array([[9, 8, 7]], np.int32), np.array([[1, 2, 30]], np.int32), 10, 9, 8)
    assert t[0] == 2 * 3 * 8


@pytest.mark.parametrize("name", ["splat_base", "splat_base_signed", "splat_prob", "splat_prob_fast"])
def test_golden_fixture(name):
    """tests/golden/splat_*.npz hold the REFERENCE's outputs (oracle/_ref = the reference's kernels compiled for gfx950,
    run on an MI355X by tools/make_golden.py).  The restatement must reproduce the integer outputs bit for bit and the
    floating-point ones to fp32 rounding (libm vs ocml exp/pow differ in the last place)."""
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    assert "oracle/_ref" in str(d["producer"])
    variant = str(d["variant"])
    pi, mi, radii, cov6 = oracle.prepare_splat_inputs(d["pts"], d["means3D"], d["scales"], d["cov3D"], d["pc_min"],
                                                      float(d["grid_size"]), float(d["scale_multiplier"]),
                                                      per_axis=bool(d["per_axis"]),
                                                      radii_min=1 if variant == "prob" else None)
    for k, v in (("points_int", pi), ("means_int", mi), ("radii", radii), ("cov6", cov6)):
        assert np.array_equal(d[k], v), k
    H, W, D = int(d["H"]), int(d["W"]), int(d["D"])
    fwd = oracle.splat_forward(variant, d["pts"], pi, d["means3D"], mi, d["opacities"], d["semantics"], radii, cov6, H, W, D)
    assert fwd["num_rendered"] == int(d["num_rendered"])
    touched, offsets, _ = oracle.box_offsets(mi, radii, H, W, D)
    assert np.array_equal(touched, d["tiles_touched"]) and np.array_equal(offsets, d["offsets"])
    keys = ["logits"] + (["bin_logits", "density", "probability"] if variant == "prob" else [])
    for k in keys:
        err = np.abs(fwd[k].astype(np.float64) - d[k]) / np.maximum(1.0, np.abs(d[k]))
        assert err.max() <= 2e-6, (k, err.max())
    grads = oracle.splat_backward(variant, d["pts"], pi, d["means3D"], mi, d["opacities"], d["semantics"], radii, cov6,
                                  H, W, D, d["out_grad"], fwd=fwd, bin_grad=d["bin_grad"], density_grad=d["density_grad"])
    for k, g in zip(("means3D_grad", "opacity_grad", "semantics_grad", "cov3D_grad"), grads):
        err = np.abs(g.astype(np.float64) - d[k]).max() / max(np.abs(d[k]).max(), 1e-6)
        assert err <= 2e-5, (k, err)
