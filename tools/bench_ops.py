"""Secondary measurements (not the headline bench): device time of splat backward and of the
deformable aggregation forward/backward at the BASELINE shapes, with algorithmic-bytes
roofline fractions.  Prints one JSON line per op."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ".")
import oracle  # noqa: E402  (numpy pre-processing restatement only)
from gaussianformer_amd import _lib  # noqa: E402
from gaussianformer_amd.deformable_aggregation import (deformable_aggregation_backward,  # noqa: E402
                                                        deformable_aggregation_forward)
from gaussianformer_amd.local_aggregate import splat_backward, splat_forward  # noqa: E402
from gaussianformer_amd.synthetic import make_daf_inputs, make_splat_inputs  # noqa: E402

dev = torch.device("cuda:0")
PEAK = 8000.0


def timed(fn, warm=5, iters=30):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def report(op, config, seconds, abytes, extra=None):
    d = {"op": op, "config": config, "us": seconds * 1e6, "algorithmic_MB": abytes / 1e6,
         "achieved_GBs": abytes / seconds / 1e9, "frac_of_8TBs": abytes / seconds / 1e9 / PEAK}
    d.update(extra or {})
    print(json.dumps(d), flush=True)


# (the "clustered" rows: centres drawn as sigmoid(N(0,1)) of the range, the lifter's anchors -- denser in the middle of the scene, as
# real nuScenes Gaussians are; the unit-claim order has no load balancing beyond late claiming)
for config, clustered in (("nuscenes_gs25600_solid", False), ("nuscenes_gs25600_solid", True), ("nuscenes_gs144000", False),
                          ("nuscenes_gs144000", True), ("prob_gs6400", False)):
    si = make_splat_inputs(config, seed=0, clustered=clustered)
    config = config + (" (clustered centres)" if clustered else "")
    pi, mi, radii, cov6 = oracle.prepare_splat_inputs(si.pts, si.means3D, si.scales, si.cov3D, si.pc_min, si.grid_size,
                                                      si.scale_multiplier, radii_min=1 if si.variant == "prob" else None)
    t = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in
         (si.pts, pi, si.means3D, mi, si.opacities, si.semantics, radii, cov6)]
    variant = _lib.GF_SPLAT_PROB if si.variant == "prob" else _lib.GF_SPLAT_BASE
    P, N = si.means3D.shape[0], si.pts.shape[0]
    extra_out = 12 * N if variant else 0
    sec = timed(lambda: splat_forward(variant, *t, si.H, si.W, si.D))
    report("splat_forward(module-level call)", config, sec, 128 * P + 24 * N + 72 * N + extra_out, {"P": P, "Gaussians_per_s": P / sec})
    # as the autograd module calls the pair: the forward told of the backward (GF_PREPARE_BACKWARD), the backward told what the
    # forward's state block says (matrix cores, rows laid out -> GF_MFMA_SPLAT | GF_RECORDS_VALID)
    logits, bl, de, pr, state = splat_forward(variant, *t, si.H, si.W, si.D, flags=_lib.GF_PREPARE_BACKWARD)
    torch.cuda.synchronize()
    words = state.view(torch.int32)[:5].tolist()
    fast = words[0] == 0 and words[1] in _lib.GF_PATHS_MATRIX_CORE and (words[4] & 1)
    g = torch.randn(N, 18, device=dev)
    gb = torch.randn(N, device=dev) if variant else None
    sec = timed(lambda: splat_backward(variant, *t, si.H, si.W, si.D, g, fwd_outputs=(logits, bl, de, pr) if variant else None,
                                       bin_logits_grad=gb, density_grad=gb, state=state,
                                       flags=(_lib.GF_MFMA_SPLAT | _lib.GF_RECORDS_VALID) if fast else 0), iters=10)
    report("splat_backward", config, sec, 128 * P + 24 * N + 72 * N + 112 * P, {"P": P, "matrix_core_backward": bool(fast)})

from gaussianformer_amd.gaussian_prepare import gaussian_prepare  # noqa: E402
for P in (25601, 144000):
    m = torch.rand(P, 3, device=dev) * torch.tensor([79.9, 79.9, 6.3], device=dev) + torch.tensor([-40.0, -40.0, -1.0], device=dev)
    sc = 0.08 + 0.5 * torch.rand(P, 3, device=dev)
    q = torch.randn(P, 4, device=dev)
    sec = timed(lambda: gaussian_prepare(m, sc, q, [-40.0, -40.0, -1.0], 0.4, 3, 200, 200, 16))
    report("gaussian_prepare(module-level call)", f"P={P}", sec, 80 * P, {"P": P})

from gaussianformer_amd.deformable_prepare import deformable_prepare  # noqa: E402
from oracle import daf_prepare_ref  # noqa: E402  (torch-op restatement of the reference block, timed for comparison)
for A in (25600, 144000):
    pts_per, cams, L, G = 9, 6, 4, 4
    kp = torch.rand(1, A, pts_per, 3, device=dev) * torch.tensor([80.0, 80.0, 6.4], device=dev) + torch.tensor([-40.0, -40.0, -1.0], device=dev)
    pm = torch.eye(4, device=dev).repeat(1, cams, 1, 1)
    for c in range(cams):
        yaw = 2 * np.pi * c / cams
        R = torch.tensor([[-np.sin(yaw), np.cos(yaw), 0.0], [0.0, 0.0, -1.0], [np.cos(yaw), np.sin(yaw), 0.0]], dtype=torch.float32)
        K = torch.tensor([[1260.0, 0, 800.0], [0, 1260.0, 450.0], [0, 0, 1.0]])
        pm[0, c, :3, :3] = (K @ R).to(dev)
        pm[0, c, :3, 3] = (K @ torch.tensor([0.0, 1.5, 0.0])).to(dev)
    wh = torch.tensor([[[1600.0, 900.0]] * cams], device=dev)
    raw = torch.randn(1, A, cams, L, pts_per, G, device=dev)
    nbytes = 2 * raw.numel() * 4 + kp.numel() * 4 + A * pts_per * cams * 8
    sec = timed(lambda: deformable_prepare(kp, pm, wh, raw))
    report("deformable_prepare (fused, module-level call)", f"A={A}", sec, nbytes, {"anchors": A})
    sec = timed(lambda: daf_prepare_ref.prepare(kp, pm, wh, raw), iters=10)
    report("deformable_prepare (reference torch-op sequence on the same GPU)", f"A={A}", sec, nbytes, {"anchors": A})

from gaussianformer_amd.head import occupancy_labels  # noqa: E402
lg = torch.randn(640000, 18, device=dev)
sec = timed(lambda: occupancy_labels(lg))
report("head_labels (fused argmax, module-level call)", "N=640000", sec, 640000 * (72 + 8), {"N": 640000})
sec = timed(lambda: lg[None].transpose(1, 2).argmax(dim=1))
report("head_labels (reference: transposed-view torch argmax on the same GPU)", "N=640000", sec, 640000 * (72 + 8), {"N": 640000})

maps = [torch.randn(1, 6, 128, h, w, device=dev) for h, w in ((108, 200), (54, 100), (27, 50), (14, 25))]
from gaussianformer_amd.deformable_aggregation import DeformableAggregationFunction as _DAF  # noqa: E402
sec = timed(lambda: _DAF.feature_maps_format(maps)[0])
report("feature_maps_format (tiled transpose, module-level call)", "nuScenes pyramid", sec, 2 * 4 * sum(m.numel() for m in maps), {})
sec = timed(lambda: torch.cat([f.reshape(1, 6, 128, -1) for f in maps], dim=-1).permute(0, 1, 3, 2).contiguous())
report("feature_maps_format (reference: cat + permute + contiguous on the same GPU)", "nuScenes pyramid", sec, 2 * 4 * sum(m.numel() for m in maps), {})

from gaussianformer_amd.sparse_conv import Rulebook  # noqa: E402
for A in (25600, 144000):
    xyz = torch.rand(A, 3, device=dev) * torch.tensor([160.0, 160.0, 16.0], device=dev)
    idx = torch.cat([torch.zeros(A, 1, dtype=torch.int32, device=dev), xyz.to(torch.int32)], dim=1)
    feat_sc = torch.randn(A, 128, device=dev)
    w_sc = torch.randn(125, 128, 128, device=dev) * 0.05
    sec = timed(lambda: Rulebook(idx, 1, (160, 160, 16), 5), iters=10)
    rb = Rulebook(idx, 1, (160, 160, 16), 5)
    flops = 2.0 * rb.total * 128 * 128
    report("subm_conv rulebook (count + host read + fill)", f"A={A}", sec, 16 * A + 8 * rb.total, {"pairs": rb.total})
    sec = timed(lambda: rb.apply(feat_sc, w_sc), iters=10)
    # (the COMPULSORY bytes -- features in, rows out, weights, pair indices --, not the partial rows the formulation
    # writes and re-reads for itself; the op is compute-bound: its bound is the bf16 matrix rate, six MFMA products per fp32 product)
    report("subm_conv apply 5^3 128->128 (gather-GEMM + reduce)", f"A={A}", sec, 4 * (125 * 128 * 128 + 2 * A * 128) + 4 * rb.total,
           {"pairs": rb.total, "TFLOPs_fp32_equivalent": flops / sec / 1e12, "frac_of_dense_bf16_peak_2500TFs_at_six_products": 6 * flops / sec / 2.5e15,
            "self_inflicted_partial_row_MB": 2 * 4 * rb.total * 128 / 1e6})
    go_sc = torch.randn(A, 128, device=dev)
    sec = timed(lambda: rb.weight_grad(feat_sc, go_sc), iters=10)
    report("subm_conv weight gradient", f"A={A}", sec, 4 * (2 * rb.total * 128 + 125 * 128 * 128), {"pairs": rb.total, "TFLOPs": flops / sec / 1e12})

DAF_CASES = () if "--splat-only" in sys.argv else ((83200, "prob_gs6400"), (230400, "nuscenes_gs25600_solid"), (1296000, "nuscenes_gs144000"))


def projected_inputs(pts):
    """Sampling locations / weights as the encoder produces them: anchors uniform in the nuScenes range, nine key points each
    (seven fixed offsets + two more, 0.35 m scale), six pinhole cameras (tools/bench_frame.cameras), masked softmax weights."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_frame
    from gaussianformer_amd.deformable_prepare import deformable_prepare
    g = torch.Generator(device="cpu").manual_seed(1)
    A = pts // 9
    lo = torch.tensor(bench_frame.PC_RANGE[:3]); hi = torch.tensor(bench_frame.PC_RANGE[3:])
    centre = lo + (hi - lo) * torch.rand(1, A, 3, generator=g)
    offs = torch.tensor(bench_frame.FIX_SCALE + [[0.3, 0.3, 0.0], [-0.3, 0.3, 0.0]]) * 0.35
    kp = (centre[:, :, None] + offs[None, None]).to(dev)
    pm, wh = bench_frame.cameras(dev)
    raw = torch.randn(1, A, 6, 4, 9, 4, generator=g).to(dev)
    loc2, w2 = deformable_prepare(kp, pm, wh, raw)
    vis = float(((loc2 > 0) & (loc2 < 1)).all(-1).float().sum(-1).mean())
    return loc2.contiguous(), w2.contiguous(), vis


def fused_case(A):
    """Round 6: one encoder block's deformable aggregation for INFERENCE -- gf_daf_fused_forward (one launch, the logits in their
    anchor and camera parts) against what it replaces: the broadcast add of the two parts, gf_daf_prepare, gf_daf_forward and the
    sum over the key points (deformable_module.py:174-233,242)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_frame
    from gaussianformer_amd.deformable_prepare import deformable_fused_forward, deformable_prepare
    g = torch.Generator(device="cpu").manual_seed(1)
    lo = torch.tensor(bench_frame.PC_RANGE[:3]); hi = torch.tensor(bench_frame.PC_RANGE[3:])
    centre = lo + (hi - lo) * torch.rand(1, A, 3, generator=g)
    offs = torch.tensor(bench_frame.FIX_SCALE + [[0.3, 0.3, 0.0], [-0.3, 0.3, 0.0]]) * 0.35
    kp = (centre[:, :, None] + offs[None, None]).to(dev)
    pm, wh = bench_frame.cameras(dev)
    ra = torch.randn(1, A, 4, 9, 4, generator=g).to(dev)
    rc = torch.randn(1, 6, 4, 9, 4, generator=g).to(dev)
    d = make_daf_inputs(num_pts=9, seed=0)
    feat, ss, st = (torch.from_numpy(d[k]).to(dev) for k in ("mc_ms_feat", "spatial_shape", "scale_start_index"))

    def three():
        raw = (ra[:, :, None] + rc[:, None]).reshape(1, A, 6, 4, 9, 4)
        loc, w = deformable_prepare(kp, pm, wh, raw)
        return deformable_aggregation_forward(feat, ss, st, loc, w).reshape(1, A, 9, 128).sum(dim=2)
    with torch.no_grad():
        want = three()
        got = deformable_fused_forward(kp, pm, wh, feat, ss, st, raw_anchor=ra, raw_cam=rc)
        err = float(((got - want).abs() / want.abs().amax(dim=-1, keepdim=True).clamp(min=1e-3)).max())
        # compulsory bytes of the block: key points, the two logit parts, the rows of the pyramid it touches (at most all of it), one output row
        nbytes = 4 * (kp.numel() + ra.numel() + rc.numel() + feat.numel() + A * 128)
        sec = timed(lambda: deformable_fused_forward(kp, pm, wh, feat, ss, st, raw_anchor=ra, raw_cam=rc))
        report("daf_fused_forward (projection + softmax + sampling + key-point sum, one launch)", f"A={A}", sec, nbytes,
               {"anchors": A, "max_row_scaled_diff_vs_three_steps": err})
        sec = timed(three)
        report("... the same block in three steps (logit add, gf_daf_prepare, gf_daf_forward, torch sum)", f"A={A}", sec, nbytes, {"anchors": A})


for A_ in (25600, 144000):
    fused_case(A_)

for pts, name in DAF_CASES:
    d = make_daf_inputs(num_pts=pts, seed=0)
    feat, ss, st, loc, w = (torch.from_numpy(d[k]).to(dev) for k in
                            ("mc_ms_feat", "spatial_shape", "scale_start_index", "sampling_location", "weights"))
    fbytes = 4 * feat.numel() + pts * (8 * 6 + 4 * 6 * 4 * 4 + 4 * 128)
    bbytes = 2 * 4 * feat.numel() + 2 * pts * (8 * 6 + 4 * 6 * 4 * 4) + 4 * 128 * pts
    ploc, pw, vis = projected_inputs(pts - pts % 9)
    uvis = float(((loc > 0) & (loc < 1)).all(-1).float().sum(-1).mean())
    # both distributions, both forward variants: "uniform" = SURVEY.md section 8d's op-level input (locations uniform in
    # (-0.2, 1.2)^2, every camera independent: the adversarial case), "projected" = pinhole geometry as in the frame benchmark
    for dist_name, (l_, w_, v_) in (("uniform", (loc, w, uvis)), ("projected", (ploc, pw, vis))):
        n_ = l_.shape[1]
        for variant, pin in (("", False), (" [channel groups pinned to XCDs]", True)):
            sec = timed(lambda: deformable_aggregation_forward(feat, ss, st, l_, w_, pin_channel_groups=pin))
            report("daf_forward" + variant, f"{name}, {dist_name} locations", sec, fbytes, {"sample_points": n_, "visible_cameras_per_point": v_})
        go = torch.randn(1, n_, 128, device=dev)
        gf, gl, gw = torch.zeros_like(feat), torch.zeros_like(l_), torch.zeros_like(w_)
        sec = timed(lambda: deformable_aggregation_backward(feat, ss, st, l_, w_, go, gf, gl, gw), iters=10)
        report("daf_backward", f"{name}, {dist_name} locations", sec, bbytes, {"sample_points": n_, "visible_cameras_per_point": v_})
