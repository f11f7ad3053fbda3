"""Fused caller-side preparation of the deformable aggregation: the torch
restatement against the reference-generated fixture on the CPU, the HIP kernels against the
restatement (values and autograd gradients) on the GPU."""
import os

import numpy as np
import pytest
import torch

from oracle import daf_prepare_ref as ref

GOLD = os.path.join(os.path.dirname(__file__), "golden", "daf_prepare.npz")


@pytest.fixture(scope="module")
def gold():
    return {k: torch.from_numpy(v) for k, v in np.load(GOLD).items()}


def _raw_weights(bs, A, cams, L, pts, G, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(bs, A, cams, L, pts, G, generator=g)


# ------------------------------------------------------------------ CPU: pin the restatement
def test_project_points_matches_reference_fixture(gold):
    p, m = ref.project_points(gold["key_points"], gold["projection_mat"], gold["image_wh"])
    assert torch.equal(p, gold["points_2d"]) and torch.equal(m, gold["mask"])
    p, m = ref.project_points(gold["key_points"], gold["projection_mat"], None)
    assert torch.equal(p, gold["points_2d_nowh"]) and torch.equal(m, gold["mask_nowh"])
    assert 0.05 < gold["mask"].float().mean() < 0.6   # the fixture exercises both branches of the mask


def test_prepare_properties(gold):
    bs, A, pts = gold["key_points"].shape[:3]
    cams, L, G = 6, 4, 4
    raw = _raw_weights(bs, A, cams, L, pts, G, 1)
    p2d, w = ref.prepare(gold["key_points"], gold["projection_mat"], gold["image_wh"], raw)
    assert p2d.shape == (bs, A * pts, cams, 2) and w.shape == (bs, A * pts, cams, L, G)
    vis = gold["mask"].permute(0, 2, 3, 1).reshape(bs, A * pts, cams)          # [bs, A*pts, cams]
    assert (w[~vis] == 0).all()                                                # invisible pairs get no weight
    per_anchor = w.reshape(bs, A, pts * cams * L, G).sum(2)                    # softmax over (pts, cams, L)
    seen = vis.reshape(bs, A, -1).any(-1)
    assert torch.allclose(per_anchor[seen], torch.ones_like(per_anchor[seen]), atol=1e-5)
    assert (per_anchor[~seen] == 0).all()                                      # all_miss anchors are zeroed


# ------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


@pytest.mark.gpu
@pytest.mark.parametrize("with_wh,with_mask,G", [(True, False, 4), (False, True, 8), (True, True, 1)])
def test_prepare_forward_backward(gpu, gold, with_wh, with_mask, G):
    from gaussianformer_amd.deformable_prepare import deformable_prepare
    bs, A, pts = gold["key_points"].shape[:3]
    cams, L = 6, 4
    raw = _raw_weights(bs, A, cams, L, pts, G, 2)
    wh = gold["image_wh"] if with_wh else None
    wmask = (torch.rand(raw.shape, generator=torch.Generator().manual_seed(3)) > 0.15) if with_mask else None
    kp = gold["key_points"]
    if not with_wh:
        # without image_wh the un-normalised pixel coordinates are far outside (0,1): shrink the intrinsics instead
        pm = gold["projection_mat"].clone()
        pm[:, :, 0] /= 1600.0
        pm[:, :, 1] /= 900.0
    else:
        pm = gold["projection_mat"]
    # reference values and gradients (fp64 autograd of the restatement)
    kp64 = kp.double().requires_grad_(True)
    raw64 = raw.double().requires_grad_(True)
    p_ref, w_ref = ref.prepare(kp64, pm.double(), None if wh is None else wh.double(), raw64, wmask)
    gp = torch.randn(p_ref.shape, generator=torch.Generator().manual_seed(4))
    gw = torch.randn(w_ref.shape, generator=torch.Generator().manual_seed(5))
    ((p_ref * gp.double()).sum() + (w_ref * gw.double()).sum()).backward()

    kpd = kp.to(gpu).requires_grad_(True)
    rawd = raw.to(gpu).requires_grad_(True)
    p, w = deformable_prepare(kpd, pm.to(gpu), None if wh is None else wh.to(gpu), rawd,
                              None if wmask is None else wmask.to(gpu))
    # visibility decisions must agree except where a coordinate sits within rounding of the gate
    p32, w32 = ref.prepare(kp, pm, wh, raw, wmask)
    assert torch.allclose(p.cpu(), p32, rtol=1e-5, atol=1e-5 * float(p32.abs().max()))
    assert torch.allclose(w.cpu().double(), w_ref.detach(), rtol=1e-4, atol=2e-6)
    ((p * gp.to(gpu)).sum() + (w * gw.to(gpu)).sum()).backward()
    for got, want, name in ((rawd.grad, raw64.grad, "grad_raw_weights"), (kpd.grad, kp64.grad, "grad_key_points")):
        got = got.double().cpu()
        scale = float(want.abs().max())
        assert (got - want).abs().max() <= 2e-5 * scale, (name, float((got - want).abs().max()), scale)


@pytest.mark.gpu
def test_prepare_feeds_daf(gpu, gold):
    """End to end: the fused preparation followed by DAF.apply equals the restatement followed
    by the DAF oracle, including the gradient that reaches the key points."""
    import oracle
    from gaussianformer_amd.deformable_aggregation import DeformableAggregationFunction as DAF
    from gaussianformer_amd.deformable_prepare import deformable_prepare
    from gaussianformer_amd.synthetic import make_daf_inputs
    bs, A, pts = gold["key_points"].shape[:3]
    cams, L, G, C = 6, 2, 4, 32
    d = make_daf_inputs(num_pts=A * pts, seed=6, B=bs, cams=cams, C=C, G=G, levels=((12, 20), (6, 10)))
    raw = _raw_weights(bs, A, cams, L, pts, G, 7)
    feat = torch.from_numpy(d["mc_ms_feat"]).to(gpu)
    ss, st = torch.from_numpy(d["spatial_shape"]).to(gpu), torch.from_numpy(d["scale_start_index"]).to(gpu)
    kpd = gold["key_points"].to(gpu).requires_grad_(True)
    rawd = raw.to(gpu).requires_grad_(True)
    p, w = deformable_prepare(kpd, gold["projection_mat"].to(gpu), gold["image_wh"].to(gpu), rawd)
    out = DAF.apply(feat, ss, st, p, w)
    p_ref, w_ref = ref.prepare(gold["key_points"], gold["projection_mat"], gold["image_wh"], raw)
    want = oracle.daf_forward(d["mc_ms_feat"], d["spatial_shape"], d["scale_start_index"], p_ref.numpy(), w_ref.numpy())
    assert np.abs(out.detach().cpu().numpy() - want).max() <= 1e-4 * max(1.0, np.abs(want).max())
    out.sum().backward()
    assert torch.isfinite(kpd.grad).all() and torch.isfinite(rawd.grad).all()
    assert float(kpd.grad.abs().max()) > 0 and float(rawd.grad.abs().max()) > 0


@pytest.mark.gpu
def test_prepare_random_shapes_sweep(gpu):
    """Random anchors / points / cameras / levels / groups, with and without image_wh and keep
    mask, against fp64 autograd of the restatement."""
    import os
    from gaussianformer_amd.deformable_prepare import deformable_prepare
    rng = np.random.default_rng(int(os.environ.get("GF_SWEEP_SEED", "5")))
    for trial in range(int(os.environ.get("GF_SWEEP_TRIALS", "10"))):
        bs, A, pts = int(rng.integers(1, 3)), int(rng.integers(1, 70)), int(rng.integers(1, 14))
        cams, L, G = int(rng.integers(1, 7)), int(rng.integers(1, 5)), int(rng.choice([1, 2, 4, 8, 16]))
        g = torch.Generator().manual_seed(900 + trial)
        kp = torch.rand(bs, A, pts, 3, generator=g) * torch.tensor([60.0, 60.0, 5.0]) - torch.tensor([30.0, 30.0, 1.0])
        pm = torch.randn(bs, cams, 4, 4, generator=g)
        pm[:, :, 2, :3] = torch.nn.functional.normalize(pm[:, :, 2, :3], dim=-1)   # depth = unit direction . X + offset
        pm[:, :, 2, 3] = torch.rand(bs, cams, generator=g) * 20
        pm[:, :, :2] *= 0.3
        wh = (torch.rand(bs, cams, 2, generator=g) + 0.5) if trial % 2 else None
        raw = torch.randn(bs, A, cams, L, pts, G, generator=g)
        wmask = (torch.rand(raw.shape, generator=g) > 0.2) if trial % 3 == 0 else None
        try:
            kp64, raw64 = kp.double().requires_grad_(True), raw.double().requires_grad_(True)
            p_ref, w_ref = ref.prepare(kp64, pm.double(), None if wh is None else wh.double(), raw64, wmask)
            gp, gw = torch.randn(p_ref.shape, generator=g), torch.randn(w_ref.shape, generator=g)
            # sampling locations far outside the image are legal but make the projection gradient huge; weigh them down
            gp = gp * (p_ref.detach().abs().float() < 4).float()
            _ph = torch.matmul(pm.double()[:, :, None, None], torch.cat([kp.double(), torch.ones(bs, A, pts, 1, dtype=torch.float64)], -1)[:, None, ..., None]).squeeze(-1)
            gp = gp * (_ph[..., 2].permute(0, 2, 3, 1).reshape(bs, A * pts, cams) > 0.5).float()[..., None]  # and ill-conditioned ones out
            ((p_ref * gp.double()).sum() + (w_ref * gw.double()).sum()).backward()
            kpd, rawd = kp.to(gpu).requires_grad_(True), raw.to(gpu).requires_grad_(True)
            p, w = deformable_prepare(kpd, pm.to(gpu), None if wh is None else wh.to(gpu), rawd,
                                      None if wmask is None else wmask.to(gpu))
            # a coordinate within rounding of a gate (u, v at 0 or 1, depth at 1e-5) may flip its
            # visibility between two fp32 evaluations: leave out the anchors that have such a point
            whd = torch.ones(bs, cams, 2, dtype=torch.float64) if wh is None else wh.double()
            ph = torch.matmul(pm.double()[:, :, None, None], torch.cat([kp.double(), torch.ones(bs, A, pts, 1, dtype=torch.float64)], -1)[:, None, ..., None]).squeeze(-1)
            depth = ph[..., 2]                                                        # [bs, cams, A, pts]
            uv = ph[..., :2] / depth.clamp(min=1e-5)[..., None] / whd[:, :, None, None]
            tol = 1e-4 * (1 + uv.abs())
            near = ((uv.abs() < tol) | ((uv - 1).abs() < tol)).any(-1) | ((depth - 1e-5).abs() < 1e-4 * (1 + ph[..., :3].abs().amax(-1)))
            same = ~near.any(dim=1).any(dim=-1)                                       # [bs, A]
            wsel = same[:, :, None].expand(bs, A, pts).reshape(bs, A * pts)
            dw = (w.cpu().double()[wsel] - w_ref.detach()[wsel]).abs()
            assert dw.numel() == 0 or dw.max() <= 2e-6 + 1e-4 * w_ref.detach()[wsel].abs().max(), f"weights diff {dw.max():.3e}"
            # u = x / depth: a depth that cancels to ~0 amplifies fp32 rounding without bound; check the
            # well-conditioned projections (depth > 0.5)
            dep = depth.permute(0, 2, 3, 1).reshape(bs, A * pts, cams)
            pok = (p_ref.detach().abs() < 1e3) & (dep > 0.5)[..., None]
            dp = (p.cpu().double()[pok] - p_ref.detach()[pok]).abs()
            assert dp.max() <= 1e-4 + 1e-4 * p_ref.detach()[pok].abs().max(), f"points_2d diff {dp.max():.3e}"
            ((p * gp.to(gpu)).sum() + (w * gw.to(gpu)).sum()).backward()
            if same.any():
                graw, want = rawd.grad.double().cpu()[same], raw64.grad[same]
                assert (graw - want).abs().max() <= 2e-5 * max(float(want.abs().max()), 1e-6), f"grad_raw diff {(graw - want).abs().max():.3e} of {want.abs().max():.3e}"
            gk, wk = kpd.grad.double().cpu(), kp64.grad
            assert (gk - wk).abs().max() <= 2e-4 * max(float(wk.abs().max()), 1e-6), f"grad_key_points diff {(gk - wk).abs().max():.3e} of {wk.abs().max():.3e}"
        except AssertionError as e:
            raise AssertionError(f"trial {trial}: bs={bs} A={A} pts={pts} cams={cams} L={L} G={G} wh={wh is not None} mask={wmask is not None}: {e}")
