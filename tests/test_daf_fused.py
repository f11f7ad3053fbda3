"""Round 6: the inference-only fused deformable aggregation (gf_daf_fused_forward: project_points, mask / all_miss /
softmax, multi-scale bilinear sampling and the sum over the key points in one launch -- deformable_module.py:174-233,242) against
the three-step path it replaces (gf_daf_prepare -> gf_daf_forward -> features.sum(dim=2)), each step of which is held to the
reference by its own tests (tests/test_daf_prepare.py: fixture made with the reference's project_points; tests/test_daf_gpu.py:
oracle/_ref kernels).  Same products, another order of the sums: bound 1e-5 of the row's magnitude."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cameras(dev, cams, wh=(1600.0, 864.0)):
    """A ring of pinhole cameras looking outwards (what tools/bench_frame.py uses): a point is seen by one or two of them."""
    pm = torch.eye(4).repeat(1, cams, 1, 1)
    K = torch.tensor([[1260.0, 0, 800.0], [0, 1260.0, 432.0], [0, 0, 1.0]])
    for c in range(cams):
        yaw = 2 * np.pi * c / cams
        R = torch.tensor([[-np.sin(yaw), np.cos(yaw), 0.0], [0.0, 0.0, -1.0], [np.cos(yaw), np.sin(yaw), 0.0]], dtype=torch.float32)
        pm[0, c, :3, :3] = K @ R
        pm[0, c, :3, 3] = K @ torch.tensor([0.0, 1.5, 0.0])
    return pm.to(dev), torch.tensor([[list(wh)] * cams], device=dev)


def _pyramid(dev, cams, C, levels, g, B=1):
    ss = torch.tensor(levels, dtype=torch.int32)
    sizes = ss[:, 0] * ss[:, 1]
    st = torch.cat([torch.zeros(1, dtype=torch.int32), torch.cumsum(sizes, 0)[:-1].to(torch.int32)])
    feat = torch.randn(B, cams, int(sizes.sum()), C, generator=g)
    return feat.to(dev), ss.to(dev), st.to(dev)


def _three_steps(kp, pm, wh, raw, feat, ss, st):
    from gaussianformer_amd.deformable_aggregation import deformable_aggregation_forward
    from gaussianformer_amd.deformable_prepare import deformable_prepare
    B, A, pts = kp.shape[:3]
    loc, w = deformable_prepare(kp, pm, wh, raw)
    out = deformable_aggregation_forward(feat, ss, st, loc.contiguous(), w.contiguous())
    return out.reshape(B, A, pts, feat.shape[-1]).sum(dim=2)


@pytest.mark.parametrize("A,pts,cams,levels,G,C,B,split,with_wh", [
    (3000, 9, 6, [(64, 176), (32, 88), (16, 44), (8, 22)], 4, 128, 1, True, True),     # the encoder's block (bench_frame): camera-embedded logits
    (3000, 9, 6, [(64, 176), (32, 88), (16, 44), (8, 22)], 4, 128, 1, False, True),    # the same with the full logits tensor
    (1001, 13, 6, [(20, 30), (10, 15)], 8, 128, 2, True, True),                         # two batch elements (a workgroup straddles them), G = 8
    (500, 5, 4, [(12, 9), (6, 5), (3, 3)], 2, 64, 1, False, False),                     # no image_wh (pixel-normalised projection), C = 64
    (257, 7, 3, [(9, 9)], 1, 32, 1, True, True),                                         # one level, one group, C = 32
])
def test_fused_forward_equals_prepare_forward_sum(A, pts, cams, levels, G, C, B, split, with_wh):
    from gaussianformer_amd.deformable_prepare import deformable_fused_forward
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(A + pts)
    L = len(levels)
    pm, wh = _cameras(dev, cams)
    if B > 1:
        pm, wh = pm.repeat(B, 1, 1, 1).contiguous(), wh.repeat(B, 1, 1).contiguous()
        pm[1, :, :3, 3] += 0.5            # (the second batch element sees the scene from elsewhere)
    if not with_wh:
        # pixel coordinates already divided by the image size inside the projection matrix
        scale = torch.tensor([1.0 / 1600.0, 1.0 / 864.0, 1.0, 1.0], device=dev)
        pm = pm * scale[None, None, :, None]
        wh = None
    # key points around the rig: +-50 m, some behind every camera's near plane, some straight above (seen by nobody)
    kp = torch.empty(B, A, pts, 3).uniform_(-50.0, 50.0, generator=g)
    kp[..., 2] = torch.empty(B, A, pts).uniform_(-3.0, 5.0, generator=g)
    kp[:, :7] = torch.tensor([0.0, 0.0, 80.0])            # anchors no camera sees: all_miss -> a zero row
    kp = kp.to(dev)
    ra = torch.randn(B, A, L, pts, G, generator=g).to(dev)
    rc = (torch.randn(B, cams, L, pts, G, generator=g) * 0.7).to(dev)
    raw = (ra[:, :, None] + rc[:, None]).contiguous()
    feat, ss, st = _pyramid(dev, cams, C, levels, g, B)
    want = _three_steps(kp, pm, wh, raw, feat, ss, st)
    with torch.no_grad():
        got = (deformable_fused_forward(kp, pm, wh, feat, ss, st, raw_anchor=ra, raw_cam=rc) if split
               else deformable_fused_forward(kp, pm, wh, feat, ss, st, raw_weights=raw))
    assert got.shape == want.shape and bool(torch.isfinite(got).all())
    assert float(got[:, :7].abs().max()) == 0.0 and float(want[:, :7].abs().max()) == 0.0
    visible = (want.abs().amax(dim=-1) > 0).float().mean().item()
    assert visible > 0.5, visible                     # the case does exercise the sampling
    scale = want.abs().amax(dim=-1, keepdim=True).clamp(min=1e-3)
    err = ((got - want).abs() / scale).max().item()
    assert err <= 1e-5, err


def test_fused_forward_refuses_autograd():
    from gaussianformer_amd.deformable_prepare import deformable_fused_forward
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    pm, wh = _cameras(dev, 6)
    feat, ss, st = _pyramid(dev, 6, 128, [(8, 8)], g)
    kp = torch.randn(1, 10, 3, 3, device=dev)
    raw = torch.randn(1, 10, 6, 1, 3, 4, device=dev, requires_grad=True)
    with pytest.raises(RuntimeError, match="no gradients"):
        deformable_fused_forward(kp, pm, wh, feat, ss, st, raw_weights=raw)
    with torch.no_grad():
        assert deformable_fused_forward(kp, pm, wh, feat, ss, st, raw_weights=raw).shape == (1, 10, 128)
