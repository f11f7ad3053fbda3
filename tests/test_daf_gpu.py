"""GPU parity of the deformable aggregation (forward + backward) against the CPU oracle."""
import numpy as np
import pytest

import oracle
from gaussianformer_amd.synthetic import make_daf_inputs

from util import assert_grad_close, assert_logits_close, to_dev

pytestmark = pytest.mark.gpu

CASES = [
    # num_pts, B, cams, C, G, levels
    dict(num_pts=777, B=1, cams=6, C=128, G=4, levels=((27, 50), (14, 25), (7, 13), (4, 7))),  # reference layout, reduced maps
    dict(num_pts=300, B=2, cams=3, C=32, G=4, levels=((6, 9), (3, 5))),                        # 8 lanes per point
    dict(num_pts=200, B=1, cams=2, C=24, G=4, levels=((5, 4),)),                               # vec 2, non-pow2 lanes -> atomic fallback
    dict(num_pts=100, B=1, cams=2, C=12, G=4, levels=((5, 4), (3, 3))),                        # vec 1
    dict(num_pts=500, B=2, cams=6, C=64, G=4, levels=((9, 16), (5, 8), (2, 3))),               # pixel-major, 16 lanes per tap
    dict(num_pts=300, B=1, cams=3, C=256, G=8, levels=((6, 9), (3, 5))),                       # pixel-major, 64 lanes per tap
    dict(num_pts=4000, B=1, cams=2, C=128, G=4, levels=((2, 2), (1, 1))),                      # thousands of taps per pixel row
]


def _edge_locs(d):
    loc = d["sampling_location"]
    loc[0, 0, 0] = [0.0, 0.5]        # exactly on the gate -> camera skipped
    loc[0, 1, 0] = [0.999, 0.001]    # corner taps fall outside the map
    loc[0, 2, 0] = [1.0, 1.0]
    loc[0, 3, 0] = [1e-4, 0.9999]
    return d


@pytest.mark.parametrize("case", CASES)
def test_daf_forward_backward(gpu, case):
    import torch
    from gaussianformer_amd.deformable_aggregation import DeformableAggregationFunction as DAF
    d = _edge_locs(make_daf_inputs(seed=21, **case))
    ref = oracle.daf_forward(**d)
    feat, ss, st, loc, w = to_dev(gpu, d["mc_ms_feat"], d["spatial_shape"], d["scale_start_index"],
                                  d["sampling_location"], d["weights"])
    feat.requires_grad_(True); loc.requires_grad_(True); w.requires_grad_(True)
    out = DAF.apply(feat, ss.long(), st.long(), loc, w)   # int64 metadata as produced by feature_maps_format
    assert_logits_close(out.detach().cpu().numpy(), ref, what="daf output")
    g = np.random.default_rng(22).standard_normal(ref.shape).astype(np.float32)
    out.backward(torch.from_numpy(g).to(gpu))
    gf, gl, gw = oracle.daf_backward(d["mc_ms_feat"], d["spatial_shape"], d["scale_start_index"],
                                     d["sampling_location"], d["weights"], g)
    assert_grad_close(feat.grad.cpu().numpy(), gf, "grad_mc_ms_feat")
    assert_grad_close(loc.grad.cpu().numpy(), gl, "grad_sampling_location")
    assert_grad_close(w.grad.cpu().numpy(), gw, "grad_weights")


def test_daf_forward_pinned_channel_groups_is_bit_identical(gpu):
    """gf_daf_forward_pinned (channel groups pinned to XCDs) against gf_daf_forward: the same bits, at the nuScenes layout
    (odd point counts, two batches) and for layouts it has to decline (falls back to the plain kernel)."""
    import torch
    from gaussianformer_amd.deformable_aggregation import deformable_aggregation_forward
    for case in (dict(num_pts=20011, B=1), dict(num_pts=777, B=2), dict(num_pts=300, B=1, cams=3, C=32, G=4, levels=((6, 9), (3, 5))),
                 dict(num_pts=300, B=1, cams=3, C=256, G=8, levels=((6, 9), (3, 5)))):
        d = _edge_locs(make_daf_inputs(seed=27, **case))
        t = to_dev(gpu, d["mc_ms_feat"], d["spatial_shape"], d["scale_start_index"], d["sampling_location"], d["weights"])
        plain = deformable_aggregation_forward(*t)
        pinned = deformable_aggregation_forward(*t, pin_channel_groups=True)
        assert torch.equal(plain.view(torch.int32), pinned.view(torch.int32)), case


def test_daf_full_feature_pyramid(gpu):
    """nuScenes-shaped pyramid (108x200 .. 14x25, 6 cams, 128 ch, 4 groups), 20 000 sample
    points vs the oracle, plus linearity in the weights at the gs25600 size (230 400 pts)."""
    import torch
    from gaussianformer_amd.deformable_aggregation import DeformableAggregationFunction as DAF
    d = make_daf_inputs(num_pts=20000, seed=23)
    ref = oracle.daf_forward(**d)
    feat, ss, st, loc, w = to_dev(gpu, d["mc_ms_feat"], d["spatial_shape"], d["scale_start_index"],
                                  d["sampling_location"], d["weights"])
    out = DAF.apply(feat, ss, st, loc, w)
    assert_logits_close(out.cpu().numpy(), ref, what="daf output")
    d = make_daf_inputs(num_pts=230400, seed=24)
    feat, ss, st, loc, w = to_dev(gpu, d["mc_ms_feat"], d["spatial_shape"], d["scale_start_index"],
                                  d["sampling_location"], d["weights"])
    o1 = DAF.apply(feat, ss, st, loc, w)
    o2 = DAF.apply(feat, ss, st, loc, w * 2)
    assert torch.equal(o2, o1 * 2)
    assert torch.isfinite(o1).all()


def test_daf_backward_pixel_major_vs_scatter(gpu):
    """gs25600-sized backward (230 400 sample points, nuScenes pyramid): the pixel-major path
    (counting sort + gather) against the reference's atomic scatter formulation, and against
    the oracle on a 20 000-point subset."""
    import torch
    from gaussianformer_amd import _lib
    from gaussianformer_amd.deformable_aggregation import deformable_aggregation_backward as bwd
    d = make_daf_inputs(num_pts=230400, seed=25)
    feat, ss, st, loc, w = to_dev(gpu, d["mc_ms_feat"], d["spatial_shape"], d["scale_start_index"],
                                  d["sampling_location"], d["weights"])
    B, cams, num_feat, C = feat.shape
    assert _lib.load().gf_daf_backward_workspace_bytes(B, cams, num_feat, C, ss.shape[0], loc.shape[1], w.shape[4]) > 0
    go = torch.randn(1, 230400, 128, device=gpu, generator=torch.Generator(device=gpu).manual_seed(26))
    res = []
    for pixel_major in (True, False):
        gf, gl, gw = torch.zeros_like(feat), torch.zeros_like(loc), torch.zeros_like(w)
        bwd(feat, ss, st, loc, w, go, gf, gl, gw, pixel_major=pixel_major)
        res.append((gf, gl, gw))
    (gf1, gl1, gw1), (gf0, gl0, gw0) = res
    # same point-major kernel (two instantiations: fp contraction may differ in the last bit)
    assert torch.allclose(gl1, gl0, rtol=1e-5, atol=1e-5 * gl0.abs().max().item())
    assert torch.allclose(gw1, gw0, rtol=1e-5, atol=1e-5 * gw0.abs().max().item())
    scale = gf0.abs().max().item()
    assert (gf1 - gf0).abs().max().item() <= 2e-5 * scale          # fp32 summation order only
    # accumulate-into semantics: a second call doubles the result
    bwd(feat, ss, st, loc, w, go, gf1, gl1, gw1)
    assert (gf1 - 2 * gf0).abs().max().item() <= 4e-5 * scale
    # oracle on a subset
    n = 20000
    gfs, gls, gws = torch.zeros_like(feat), torch.zeros_like(loc[:, :n]), torch.zeros_like(w[:, :n])
    bwd(feat, ss, st, loc[:, :n].contiguous(), w[:, :n].contiguous(), go[:, :n].contiguous(), gfs, gls, gws)
    of, ol, ow = oracle.daf_backward(d["mc_ms_feat"], d["spatial_shape"], d["scale_start_index"],
                                     d["sampling_location"][:, :n], d["weights"][:, :n], go[:, :n].cpu().numpy())
    assert_grad_close(gfs.cpu().numpy(), of, "grad_mc_ms_feat")
    assert_grad_close(gls.cpu().numpy(), ol, "grad_sampling_location")
    assert_grad_close(gws.cpu().numpy(), ow, "grad_weights")


def test_feature_maps_format_roundtrip(gpu):
    import torch
    from gaussianformer_amd.deformable_aggregation import DeformableAggregationFunction as DAF
    maps = [torch.randn(1, 6, 16, h, w, device=gpu) for h, w in ((8, 12), (4, 6), (2, 3))]
    col, ss, st = DAF.feature_maps_format(maps)
    assert col.shape == (1, 6, 8 * 12 + 4 * 6 + 2 * 3, 16) and st.tolist() == [0, 96, 120]
    back = DAF.feature_maps_format([col, ss, st], inverse=True)
    for a, b in zip(maps, back):
        assert torch.equal(a, b)
    # the tiled-transpose kernel against the reference's cat + permute (deformable_aggregation.py:77-100),
    # odd sizes (partial 32x32 tiles), several batch elements, and its gradient (the inverse transpose)
    maps = [torch.randn(2, 3, 37, h, w, device=gpu, requires_grad=True) for h, w in ((9, 13), (5, 7), (1, 1))]
    col, ss, st = DAF.feature_maps_format(maps)
    want = torch.cat([f.reshape(2, 3, 37, -1) for f in maps], dim=-1).permute(0, 1, 3, 2)
    assert col.is_contiguous() and torch.equal(col, want)
    assert ss.tolist() == [[9, 13], [5, 7], [1, 1]] and st.tolist() == [0, 117, 152]
    g = torch.randn_like(col)
    got = torch.autograd.grad(col, maps, g)
    ref = torch.autograd.grad(want, maps, g)
    for a, b in zip(got, ref):
        assert torch.equal(a, b)


def test_daf_random_shapes_sweep(gpu):
    """Randomised sweep over batch, cameras, channels/groups, pyramid shapes and point counts
    (pixel-major and scatter backward paths, every lane layout) against the oracle."""
    import os
    import torch
    from gaussianformer_amd.deformable_aggregation import DeformableAggregationFunction as DAF
    rng = np.random.default_rng(int(os.environ.get("GF_SWEEP_SEED", "77")))
    for trial in range(int(os.environ.get("GF_SWEEP_TRIALS", "12"))):
        G = int(rng.choice([1, 2, 4, 8]))
        C = G * int(rng.choice([1, 2, 3, 4, 8, 16, 32]))
        if C > 256:
            C = 256
            G = int(rng.choice([4, 8]))
        nl = int(rng.integers(1, 5))
        levels = tuple((int(rng.integers(1, 30)), int(rng.integers(1, 40))) for _ in range(nl))
        case = dict(num_pts=int(rng.integers(1, 1500)), B=int(rng.integers(1, 4)), cams=int(rng.integers(1, 7)), C=C, G=G,
                    levels=levels)
        d = make_daf_inputs(seed=500 + trial, **case)
        try:
            ref = oracle.daf_forward(**d)
            feat, ss, st, loc, w = to_dev(gpu, d["mc_ms_feat"], d["spatial_shape"], d["scale_start_index"],
                                          d["sampling_location"], d["weights"])
            feat.requires_grad_(True); loc.requires_grad_(True); w.requires_grad_(True)
            out = DAF.apply(feat, ss, st, loc, w)
            assert_logits_close(out.detach().cpu().numpy(), ref, what="daf output")
            g = rng.standard_normal(ref.shape).astype(np.float32)
            out.backward(torch.from_numpy(g).to(gpu))
            gf, gl, gw = oracle.daf_backward(d["mc_ms_feat"], d["spatial_shape"], d["scale_start_index"],
                                             d["sampling_location"], d["weights"], g)
            assert_grad_close(feat.grad.cpu().numpy(), gf, "grad_mc_ms_feat")
            assert_grad_close(loc.grad.cpu().numpy(), gl, "grad_sampling_location")
            assert_grad_close(w.grad.cpu().numpy(), gw, "grad_weights")
        except AssertionError as e:
            raise AssertionError(f"trial {trial}: {case}: {e}")


# ---- round 5: the accumulation of grad_mc_ms_feat by image regions (gf_daf_raccumulate_kernel) against the tile formulation
# (library option "daf.backward_tiles") and the oracle, on the shapes its geometry has to get right

_REGION_CASES = [
    # the nuScenes layout: dyadic pyramid, 10 x 10 + 7 x 7 + 5 x 5 + 4 x 4 rows per region
    ("nuscenes pyramid", dict(num_pts=3000, B=1, cams=6, C=128, G=4, levels=((64, 176), (32, 88), (16, 44), (8, 22))), None),
    # levels that are NOT halves of each other: rectangles from the ratios, some capped -> taps outside take the per-tap path
    ("odd ratios", dict(num_pts=2500, B=1, cams=2, C=128, G=4, levels=((37, 53), (30, 41), (9, 17))), None),
    ("coarse levels larger than level 0's share", dict(num_pts=1500, B=1, cams=1, C=128, G=2, levels=((16, 16), (15, 15), (14, 14), (13, 13))), None),
    ("sixteen lanes per row, one group, two batch elements", dict(num_pts=1200, B=2, cams=3, C=64, G=1, levels=((20, 33), (10, 17))), None),
    ("one level", dict(num_pts=900, B=1, cams=2, C=128, G=4, levels=((24, 40),)), None),
    ("maps smaller than a region", dict(num_pts=700, B=1, cams=2, C=128, G=4, levels=((5, 7), (3, 4), (2, 2), (1, 1))), None),
    # everything lands in ONE region: many items of one region, their row adds meet in the same rows
    ("one crowded region", dict(num_pts=6000, B=1, cams=1, C=128, G=4, levels=((64, 176), (32, 88), (16, 44), (8, 22))), "crowd"),
    ("nothing visible", dict(num_pts=500, B=1, cams=2, C=128, G=4, levels=((16, 24), (8, 12))), "invisible"),
    ("on the image border", dict(num_pts=800, B=1, cams=2, C=128, G=4, levels=((16, 24), (8, 12), (4, 6))), "border"),
]


@pytest.mark.parametrize("name,case,special", _REGION_CASES, ids=[c[0] for c in _REGION_CASES])
def test_daf_backward_by_regions(gpu, name, case, special):
    import torch
    from gaussianformer_amd.deformable_aggregation import deformable_aggregation_backward as bwd
    d = make_daf_inputs(seed=31, **case)
    rng = np.random.default_rng(32)
    loc = d["sampling_location"]
    if special == "crowd":
        loc[...] = (0.40 + 0.02 * rng.random(loc.shape)).astype(np.float32)
    elif special == "invisible":
        loc[...] = (1.0 + rng.random(loc.shape)).astype(np.float32)
    elif special == "border":
        edge = rng.integers(0, 4, size=loc.shape[:-1])
        loc[..., 0] = np.where(edge == 0, 1e-6, np.where(edge == 1, 1 - 1e-6, loc[..., 0])).astype(np.float32)
        loc[..., 1] = np.where(edge == 2, 1e-6, np.where(edge == 3, 1 - 1e-6, loc[..., 1])).astype(np.float32)
    feat, ss, st, loc_t, w = to_dev(gpu, d["mc_ms_feat"], d["spatial_shape"], d["scale_start_index"], loc, d["weights"])
    B, pts, C = loc.shape[0], loc.shape[1], feat.shape[-1]
    g = rng.standard_normal((B, pts, C)).astype(np.float32)
    go = torch.from_numpy(g).to(gpu)
    res = {}
    from gaussianformer_amd import _lib
    for mode in ("regions", "tiles"):
        with _lib.option("daf.backward_tiles", 1 if mode == "tiles" else 0):
            gf, gl, gw = torch.zeros_like(feat), torch.zeros_like(loc_t), torch.zeros_like(w)
            bwd(feat, ss, st, loc_t, w, go, gf, gl, gw)
        res[mode] = (gf, gl, gw)
    (gf1, gl1, gw1), (gf0, gl0, gw0) = res["regions"], res["tiles"]
    assert torch.equal(gl1, gl0) and torch.equal(gw1, gw0)            # the gather side is the same kernel
    scale = max(gf0.abs().max().item(), 1e-30)
    assert torch.isfinite(gf1).all()
    assert (gf1 - gf0).abs().max().item() <= 2e-5 * scale, name       # fp32 summation order only
    if special == "invisible":
        assert gf1.abs().max().item() == 0.0
    ogf, ogl, ogw = oracle.daf_backward(d["mc_ms_feat"], d["spatial_shape"], d["scale_start_index"], loc, d["weights"], g)
    assert_grad_close(gf1.cpu().numpy(), ogf, "grad_mc_ms_feat")
    assert_grad_close(gl1.cpu().numpy(), ogl, "grad_sampling_location")
    assert_grad_close(gw1.cpu().numpy(), ogw, "grad_weights")
    # accumulation semantics: a second call adds to what is there
    bwd(feat, ss, st, loc_t, w, go, gf1, gl1, gw1)
    assert (gf1 - 2 * gf0).abs().max().item() <= 4e-5 * scale
