"""gf_key_points against the torch restatement of SparseGaussian3DKeyPointsGenerator.forward
(oracle/daf_prepare_ref.key_points, itself pinned through tests/golden/caller_dfa.npz, which the reference's own
generator produced): values and every gradient, fp64 autograd as the truth."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
FIX = [[0, 0, 0], [0.45, 0, 0], [-0.45, 0, 0], [0, 0.45, 0], [0, -0.45, 0], [0, 0, 0.45], [0, 0, -0.45]]
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("bs,A,D,K", [(1, 300, 28, 2), (2, 77, 11, 6), (1, 64, 10, 0)])
def test_key_points_values_and_gradients(gpu, bs, A, D, K):
    import torch
    from oracle import daf_prepare_ref
    from gaussianformer_amd.key_points import key_points
    g = torch.Generator().manual_seed(3)
    anchor = torch.randn(bs, A, D, generator=g) * 2.0
    anchor[0, 0, 0] = 12.0      # beyond the sigmoid clamp: zero gradient there
    anchor[0, 1, 4] = -9.5
    learned = torch.randn(bs, A, K, 3, generator=g) * 3.0 if K else None
    pc_range, scale_range = [-50.0, -50.0, -5.0, 50.0, 50.0, 3.0], [0.08, 0.64]
    w = torch.randn(bs, A, len(FIX) + K, 3, generator=g)
    # fp64 truth
    a64 = anchor.double().requires_grad_(True)
    l64 = learned.double().requires_grad_(True) if K else None
    feat = torch.zeros(bs, A, 1, dtype=torch.float64)
    wfc = None
    if K:
        # key_points takes the learned offsets through learnable_fc: feed them as a bias-free identity
        ref = _restated(daf_prepare_ref, a64, l64, pc_range, scale_range)
    else:
        ref = daf_prepare_ref.key_points(a64, feat, FIX, None, None, pc_range, scale_range)
    (ref * w.double()).sum().backward()
    a = anchor.to(gpu).requires_grad_(True)
    l = learned.to(gpu).requires_grad_(True) if K else None
    out = key_points(a, l, torch.tensor(FIX, dtype=torch.float32, device=gpu), pc_range, scale_range)
    (out * w.to(gpu)).sum().backward()
    assert torch.allclose(out.detach().cpu().double(), ref.detach(), rtol=1e-5, atol=2e-5)
    ga, ra = a.grad.cpu().double(), a64.grad
    assert float((ga[..., 10:]).abs().max()) == 0.0 if D > 10 else True
    assert torch.allclose(ga, ra, rtol=1e-4, atol=1e-4 * float(ra.abs().max()))
    assert float(ga[0, 0, 0]) == 0.0 and float(ga[0, 1, 4]) == 0.0
    if K:
        assert torch.allclose(l.grad.cpu().double(), l64.grad, rtol=1e-4, atol=1e-4 * float(l64.grad.abs().max()))


@pytest.mark.parametrize("xyz_act,scale_act", [("identity", "sigmoid"), ("sigmoid", "none"), ("identity", "identity")])
def test_identity_activations(gpu, xyz_act, scale_act):
    """xyz_activation / scale_activation other than "sigmoid" (deformable_module.py:27-28, :66-67, :79-80): the columns
    are used as they are; values and gradients against fp64 autograd of the restatement, through the drop-in module."""
    import torch
    from oracle import daf_prepare_ref
    from gaussianformer_amd.key_points import SparseGaussian3DKeyPointsGenerator
    g = torch.Generator().manual_seed(7)
    bs, A, D, K, E = 1, 257, 12, 2, 16
    anchor = torch.randn(bs, A, D, generator=g)
    anchor[..., :6] = torch.rand(bs, A, 6, generator=g)          # un-activated columns hold fractions of the ranges
    feat = torch.randn(bs, A, E, generator=g)
    pc_range, scale_range = [-50.0, -50.0, -5.0, 50.0, 50.0, 3.0], [0.08, 0.64]
    gen = SparseGaussian3DKeyPointsGenerator(embed_dims=E, num_learnable_pts=K, fix_scale=FIX, pc_range=pc_range,
                                             scale_range=scale_range, xyz_activation=xyz_act, scale_activation=scale_act).to(gpu)
    w = torch.randn(bs, A, len(FIX) + K, 3, generator=g)
    a64 = anchor.double().requires_grad_(True)
    ref = daf_prepare_ref.key_points(a64, feat.double(), FIX, gen.learnable_fc.weight.detach().cpu().double(),
                                     gen.learnable_fc.bias.detach().cpu().double(), pc_range, scale_range,
                                     xyz_activation=xyz_act, scale_activation=scale_act)
    (ref * w.double()).sum().backward()
    a = anchor.to(gpu).requires_grad_(True)
    out = gen(a, feat.to(gpu))
    (out * w.to(gpu)).sum().backward()
    assert torch.allclose(out.detach().cpu().double(), ref.detach(), rtol=1e-5, atol=2e-5)
    assert torch.allclose(a.grad.cpu().double(), a64.grad, rtol=1e-4, atol=1e-4 * float(a64.grad.abs().max()))


def _restated(ref_mod, anchor, learned, pc_range, scale_range):
    """daf_prepare_ref.key_points with the learned offsets given directly (weight = identity on a flattened feature)."""
    import torch
    bs, A, K, _ = learned.shape
    eye = torch.eye(K * 3, dtype=anchor.dtype)
    return ref_mod.key_points(anchor, learned.reshape(bs, A, K * 3), FIX, eye, torch.zeros(K * 3, dtype=anchor.dtype), pc_range,
                              scale_range)


def test_generator_module_matches_the_reference_caller(gpu):
    """The drop-in generator, with the reference's parameters, reproduces the sampling locations the reference's
    generator + project_points handed to the op (tests/golden/caller_dfa.npz)."""
    import torch
    from oracle import daf_prepare_ref
    from gaussianformer_amd.key_points import SparseGaussian3DKeyPointsGenerator
    d = np.load(os.path.join(GOLDEN, "caller_dfa.npz"))
    gen = SparseGaussian3DKeyPointsGenerator(embed_dims=int(d["embed_dims"]), num_learnable_pts=2, fix_scale=FIX,
                                             pc_range=[-20.0, -20.0, -2.0, 20.0, 20.0, 4.0], scale_range=[0.08, 0.64]).to(gpu)
    with torch.no_grad():
        gen.learnable_fc.weight.copy_(torch.from_numpy(d["param_kps_generator.learnable_fc.weight"]))
        gen.learnable_fc.bias.copy_(torch.from_numpy(d["param_kps_generator.learnable_fc.bias"]))
    kp = gen(torch.from_numpy(d["anchor"]).to(gpu), torch.from_numpy(d["instance_feature"]).to(gpu)).detach()
    uv, _ = daf_prepare_ref.project_points(kp.cpu(), torch.from_numpy(d["projection_mat"]), torch.from_numpy(d["image_wh"]))
    bs, cams, A, K, _ = uv.shape
    loc = uv.permute(0, 2, 3, 1, 4).reshape(bs, A * K, cams, 2).numpy()
    # points behind a camera are divided by the 1e-5 depth clamp (|uv| ~ 1e5): an fp32 ulp of the key point moves them by
    # ~1e-2, so those are compared relatively; the visible ones (0 < uv < 1) must agree to 2e-5
    want = d["call_sampling_location"]
    assert np.allclose(loc, want, rtol=1e-4, atol=2e-5)
    vis = (want > 0).all(-1) & (want < 1).all(-1)
    assert vis.mean() > 0.1 and np.abs(loc - want)[vis].max() <= 2e-5
