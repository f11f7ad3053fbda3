#!/usr/bin/env python
"""frames/s of one INFERENCE frame of the GaussianFormer hot path (second clause of BASELINE.json's metric;
BASELINE configs [2]/[3] without the image backbone, which is third-party mmseg/mmcv code).

The frame follows the reference's op sequence (model/encoder/gaussian_encoder/gaussian_encoder.py:74-123 with the
``operation_order`` of config/nuscenes_gs25600_solid.py:161-173, then model/head/gaussian_head.py:122-197):

    feature pyramid -> feature_maps_format (ONCE per frame; the reference redoes it in every block)
    block 0:    deformable -> ffn -> norm -> refine
    blocks 1-3: spconv -> norm -> deformable -> ffn -> norm -> refine
    head:       prepare_gaussian_args (fused, device-side Sigma^-1) -> splat -> occupancy labels

Native (libgf_hip.so): feature_maps_format, SparseConv3D, key points, deformable_prepare (projection + masked softmax),
DAF, gaussian_prepare, the splat and the label epilogue.  Torch stand-ins with the reference's shapes and random
weights (there are no checkpoints offline): anchor encoder, learnable_fc, camera encoder + weights_fc, output_proj, AsymmetricFFN (256 -> 512 -> 128 + identity_fc), LayerNorm, the
refinement MLP.  Inputs are synthetic and resident in HBM; the timed region is the whole frame, bracketed by
synchronize().  Prints one JSON line per config.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

EMBED, CAMS, LEVELS, GROUPS = 128, 6, 4, 4
FIX_SCALE = [[0, 0, 0], [0.45, 0, 0], [-0.45, 0, 0], [0, 0.45, 0], [0, -0.45, 0], [0, 0, 0.45], [0, 0, -0.45]]
PC_RANGE = [-50.0, -50.0, -5.0, 50.0, 50.0, 3.0]
FRAME_CONFIGS = {
    # anchors, scale range, semantic dim carried by the anchor, appended empty Gaussian, opacity in the anchor
    "nuscenes_gs25600_solid": dict(anchors=25600, scale_range=(0.08, 0.64), sem_dim=17, with_empty=True, include_opa=True),
    "nuscenes_gs144000": dict(anchors=144000, scale_range=(0.08, 0.32), sem_dim=18, with_empty=False, include_opa=False),
}


def _sig(t):
    return torch.sigmoid(t.clamp(-9.21, 9.21))       # safe_sigmoid, model/utils/safe_ops.py:7-9


def _rotmat(q):
    """Rotation matrix of a (w, x, y, z) quaternion (model/utils/utils.py:20-69 produces the same matrix)."""
    q = F.normalize(q, dim=-1)
    w, x, y, z = q.unbind(-1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=-1).unflatten(-1, (3, 3))


def cameras(dev):
    pm = torch.eye(4).repeat(1, CAMS, 1, 1)
    K = torch.tensor([[1260.0, 0, 800.0], [0, 1260.0, 432.0], [0, 0, 1.0]])
    for c in range(CAMS):
        yaw = 2 * np.pi * c / CAMS
        R = torch.tensor([[-np.sin(yaw), np.cos(yaw), 0.0], [0.0, 0.0, -1.0], [np.cos(yaw), np.sin(yaw), 0.0]], dtype=torch.float32)
        pm[0, c, :3, :3] = K @ R
        pm[0, c, :3, 3] = K @ torch.tensor([0.0, 1.5, 0.0])
    return pm.to(dev), torch.tensor([[[1600.0, 864.0]] * CAMS], device=dev)


THREE_STEP_DAF = os.environ.get("GF_FRAME_THREE_STEP_DAF") == "1"   # (comparison: gf_daf_prepare + gf_daf_forward + torch sum, rounds 1 - 5)


class Deformable(nn.Module):
    """DeformableFeatureAggregation with residual_mode="cat" (config/_base_/model.py:72-100)."""

    def __init__(self, scale_range, learnable_pts):
        super().__init__()
        self.scale_range = scale_range
        self.num_pts = len(FIX_SCALE) + learnable_pts
        self.learnable_fc = nn.Linear(EMBED, learnable_pts * 3)
        self.camera_encoder = nn.Sequential(nn.Linear(12, EMBED), nn.ReLU(True), nn.LayerNorm(EMBED),
                                            nn.Linear(EMBED, EMBED), nn.ReLU(True), nn.LayerNorm(EMBED))
        self.weights_fc = nn.Linear(EMBED, GROUPS * LEVELS * self.num_pts)
        self.output_proj = nn.Linear(EMBED, EMBED)
        self.register_buffer("fix_scale", torch.tensor(FIX_SCALE, dtype=torch.float32))

    def key_points(self, anchor, feat):
        from gaussianformer_amd.key_points import key_points
        bs, A = anchor.shape[:2]
        return key_points(anchor, self.learnable_fc(feat).reshape(bs, A, -1, 3), self.fix_scale, PC_RANGE, self.scale_range)

    def forward(self, feat, anchor, anchor_embed, table, pm, wh):
        from gaussianformer_amd.deformable_aggregation import DeformableAggregationFunction as DAF
        from gaussianformer_amd.deformable_prepare import deformable_prepare
        bs, A = feat.shape[:2]
        kp = self.key_points(anchor, feat)
        cam = self.camera_encoder(pm[:, :, :3].reshape(bs, CAMS, -1))
        # weights_fc is linear: W (f + e + c_cam) + b = [W (f + e) + b] + W c_cam.  The reference applies it to the
        # broadcast sum (deformable_module.py:253-262: an [A*cams, 128] x [128, 144] GEMM, 3.0 ms at 25 600 anchors);
        # the anchor part and the camera part are multiplied separately here (1/6 of the GEMM) and added on the fly.
        per_anchor = self.weights_fc(feat + anchor_embed)                              # [bs, A, 144]
        per_cam = F.linear(cam, self.weights_fc.weight)                                # [bs, cams, 144]
        if not torch.is_grad_enabled() and not THREE_STEP_DAF:
            # inference (round 6): projection, softmax, sampling and the sum over the key points in ONE launch; the logits stay in
            # their two parts (gf_daf_fused_forward adds them as it reads them)
            from gaussianformer_amd.deformable_prepare import deformable_fused_forward
            out = deformable_fused_forward(kp, pm, wh, *table, raw_anchor=per_anchor.reshape(bs, A, LEVELS, self.num_pts, GROUPS),
                                           raw_cam=per_cam.reshape(bs, CAMS, LEVELS, self.num_pts, GROUPS))
        else:
            raw = (per_anchor[:, :, None] + per_cam[:, None]).reshape(bs, A, CAMS, LEVELS, self.num_pts, GROUPS)
            loc, weights = deformable_prepare(kp, pm, wh, raw)
            out = DAF.apply(*table, loc, weights).reshape(bs, A, self.num_pts, EMBED).sum(dim=2)
        return torch.cat([self.output_proj(out), feat], dim=-1)


class FFN(nn.Module):
    """AsymmetricFFN(in_channels=256, embed_dims=128, feedforward_channels=512), ffn_module.py:8-80."""

    def __init__(self):
        super().__init__()
        self.layers = nn.Sequential(nn.Linear(2 * EMBED, 4 * EMBED), nn.ReLU(True), nn.Linear(4 * EMBED, EMBED))
        self.identity_fc = nn.Linear(2 * EMBED, EMBED)

    def forward(self, x):
        return self.identity_fc(x) + self.layers(x)


class Refine(nn.Module):
    """Stand-in for SparseGaussian3DRefinementModule (refine_module.py:64-125): MLP on feature + embedding, residual
    update of the anchor, decoded Gaussian properties."""

    def __init__(self, anchor_dim, scale_range, sem_dim):
        super().__init__()
        self.layers = nn.Sequential(nn.Linear(EMBED, EMBED), nn.ReLU(True), nn.LayerNorm(EMBED),
                                    nn.Linear(EMBED, EMBED), nn.ReLU(True), nn.LayerNorm(EMBED), nn.Linear(EMBED, anchor_dim))
        self.scale_range = scale_range
        with torch.no_grad():
            self.layers[-1].weight.mul_(0.05)
            # semantic columns: full-size weights and a positive bias, so that with random weights the occupied classes win
            # against the empty Gaussian's scalar (10.0) where Gaussians are dense and different classes win in different
            # voxels -- the eager-vs-graph label comparison below is then a comparison of many labels, not of one
            self.layers[-1].weight[anchor_dim - sem_dim:].mul_(20.0 * 4.0)
            self.layers[-1].bias[anchor_dim - sem_dim:].add_(2.5)

    def forward(self, feat, anchor, anchor_embed):
        out = self.layers(feat + anchor_embed)
        anchor = torch.cat([out[..., :3] + anchor[..., :3], out[..., 3:6], F.normalize(out[..., 6:10], dim=-1), out[..., 10:]], dim=-1)
        return anchor


class Frame(nn.Module):
    def __init__(self, config):
        super().__init__()
        from gaussianformer_amd.local_aggregate import LocalAggregator
        from gaussianformer_amd.sparse_conv import SparseConv3D
        c = FRAME_CONFIGS[config]
        self.cfg = c
        self.anchor_dim = 10 + (1 if c["include_opa"] else 0) + c["sem_dim"]
        self.anchor_encoder = nn.Sequential(nn.Linear(self.anchor_dim, EMBED), nn.ReLU(True), nn.LayerNorm(EMBED), nn.Linear(EMBED, EMBED))
        order = ["deformable", "ffn", "norm", "refine"] + ["spconv", "norm", "deformable", "ffn", "norm", "refine"] * 3
        self.order = order
        layers = []
        for op in order:
            layers.append({"deformable": lambda: Deformable(c["scale_range"], 2), "ffn": FFN, "norm": lambda: nn.LayerNorm(EMBED),
                           "refine": lambda: Refine(self.anchor_dim, c["scale_range"], c["sem_dim"]),
                           "spconv": lambda: SparseConv3D(EMBED, EMBED, PC_RANGE, [0.5, 0.5, 0.5], use_out_proj=True,
                                                          pairs_per_point=64)}[op]())   # no host read per rulebook
        self.layers = nn.ModuleList(layers)
        for m in self.layers:
            if isinstance(m, SparseConv3D):
                nn.init.normal_(m.layer.weight, std=0.01)
        self.aggregator = LocalAggregator(3, 200, 200, 16, PC_RANGE[:3], 0.5, check_inputs=False)
        # constants as buffers (no host-to-device copy inside the frame: the frame can be captured as a HIP graph)
        self.register_buffer("pc_lo", torch.tensor(PC_RANGE[:3]), persistent=False)
        self.register_buffer("pc_span", torch.tensor(PC_RANGE[3:]) - torch.tensor(PC_RANGE[:3]), persistent=False)
        self.register_buffer("empty_scalar", torch.full((1,), 10.0), persistent=False)
        self.register_buffer("empty_mean", torch.tensor([[[0.0, 0.0, -1.0]]]), persistent=False)
        self.register_buffer("empty_scale", torch.tensor([[[100.0, 100.0, 8.0]]]), persistent=False)
        self.register_buffer("empty_rot", torch.tensor([[[1.0, 0.0, 0.0, 0.0]]]), persistent=False)

    def gaussians(self, anchor):
        c = self.cfg
        # keep the centres strictly inside the grid (the head asserts it in the reference)
        means = (0.001 + 0.998 * _sig(anchor[..., :3])) * self.pc_span + self.pc_lo
        lo, hi = c["scale_range"]
        scales = lo + (hi - lo) * _sig(anchor[..., 3:6])
        rots = F.normalize(anchor[..., 6:10], dim=-1)
        k = 10
        if c["include_opa"]:
            opa = _sig(anchor[..., k:k + 1]); k += 1
        else:
            opa = torch.ones_like(anchor[..., :1])
        sem = anchor[..., k:]
        if c["with_empty"]:   # gaussian_head.py:90-102: zero column + the appended empty Gaussian, one launch (gf_gaussian_pack)
            from gaussianformer_amd.gaussian_prepare import _GaussianPack
            means, scales, rots, sem, opa = _GaussianPack.apply(
                means, scales, rots, F.softplus(sem), opa, self.empty_scalar, [[0.0, 0.0, -1.0], [100.0, 100.0, 8.0], [1.0, 0.0, 0.0, 0.0]],
                18, False, True, False, 17)
        return means, scales, rots, opa, sem

    @torch.no_grad()
    def forward(self, anchor, feat, maps, pm, wh, pts):
        from gaussianformer_amd.deformable_aggregation import DeformableAggregationFunction as DAF
        from gaussianformer_amd.head import occupancy_labels
        table = DAF.feature_maps_format(maps)
        embed = self.anchor_encoder(anchor)
        for op, layer in zip(self.order, self.layers):
            if op == "deformable":
                feat = layer(feat, anchor, embed, table, pm, wh)
            elif op == "spconv":
                feat = layer(feat, anchor)
            elif op == "refine":
                anchor = layer(feat, anchor, embed)
                embed = self.anchor_encoder(anchor)
            else:
                feat = layer(feat)
        means, scales, rots, opa, sem = self.gaussians(anchor)
        logits = self.aggregator.forward_from_rotations(pts, means, opa.squeeze(-1), sem, scales, rots)
        return occupancy_labels(logits)

    @torch.no_grad()
    def forward_sharded(self, anchor, feat, maps, pm, wh, pts, rank, world, gather, reduce_sum, splat="slab"):
        """The same frame with the ANCHOR SET split over `world` ranks (the reference runs replicas only, train.py:41-43).
        Rank r owns anchors [lo, hi): key points, projection + masked softmax, deformable aggregation, FFN, norms and
        refinement are per-anchor work on its slice against the replicated feature pyramid -- no collective.  The sparse
        convolution needs every anchor's neighbours: features and anchors are all-gathered (`gather(t, dim)` -> the
        concatenation over ranks) and the convolution runs replicated on the whole set, each rank keeping its slice.  Head:
        the ranks' Gaussians are all-gathered (116 B each) and the grid is rendered either by x-slabs (`splat="slab"`:
        every rank its band of voxel rows from all Gaussians, labels all-gathered) or by Gaussian shards + one all-reduce
        of the logits (`splat="allreduce"`, `reduce_sum(t)`).  Returns the labels of the whole grid on every rank."""
        from gaussianformer_amd.deformable_aggregation import DeformableAggregationFunction as DAF
        from gaussianformer_amd.head import occupancy_labels
        from gaussianformer_amd.sharded import shard_bounds, slab_bounds
        A = anchor.shape[1]
        lo, hi = shard_bounds(A, rank, world)
        anchor, feat = anchor[:, lo:hi].contiguous(), feat[:, lo:hi].contiguous()
        table = DAF.feature_maps_format(maps)
        embed = self.anchor_encoder(anchor)
        for op, layer in zip(self.order, self.layers):
            if op == "deformable":
                feat = layer(feat, anchor, embed, table, pm, wh)
            elif op == "spconv":
                # the whole anchor set is the neighbourhood, the rank's own slice the output (round 5: the block used to run
                # replicated on every rank -- Amdahl's share of the sharded frame)
                feat = layer(gather(feat, 1), gather(anchor, 1), out_range=(lo, hi)).contiguous()
            elif op == "refine":
                anchor = layer(feat, anchor, embed)
                embed = self.anchor_encoder(anchor)
            else:
                feat = layer(feat)
        anchor_all = gather(anchor, 1)
        means, scales, rots, opa, sem = self.gaussians(anchor_all)     # (the appended empty Gaussian once, on the gathered set)
        if splat == "slab":
            x0, x1 = slab_bounds(self.aggregator.H, rank, world)
            plane = self.aggregator.W * self.aggregator.D
            from gaussianformer_amd.gaussian_prepare import covariance_inverse
            cov = covariance_inverse(scales, rots)
            logits = self.aggregator.forward_slab(x0, x1, pts, means, opa.squeeze(-1), sem, scales, cov)
            labels = occupancy_labels(logits)
            bounds = [slab_bounds(self.aggregator.H, r, world) for r in range(world)]
            tallest = max(b - a for a, b in bounds) * plane
            mine = labels.new_zeros(tallest)
            mine[:labels.shape[0]] = labels
            buf = gather(mine[None], 0).reshape(world, tallest)
            return torch.cat([buf[r, :(b - a) * plane] for r, (a, b) in enumerate(bounds)])
        P = means.shape[1]
        glo, ghi = shard_bounds(P, rank, world)
        logits = self.aggregator.forward_from_rotations(pts, means[:, glo:ghi], opa.squeeze(-1)[:, glo:ghi], sem[:, glo:ghi],
                                                         scales[:, glo:ghi], rots[:, glo:ghi]).contiguous()
        return occupancy_labels(reduce_sum(logits))

    def check(self):
        """After a synchronisation: did every rulebook of the last frame fit its capacity?"""
        for m in self.layers:
            if getattr(m, "last_rulebook", None) is not None:
                m.last_rulebook.check()


def run(config="nuscenes_gs25600_solid", frames=10, warmup=3, device="cuda:0", graph=False):
    from gaussianformer_amd.synthetic import DAF_LEVELS, voxel_centres
    if not torch.cuda.is_available():
        raise RuntimeError("bench_frame.py needs an MI355X")
    dev = torch.device(device)
    torch.manual_seed(0)
    model = Frame(config).to(dev).eval()
    A = model.cfg["anchors"]
    anchor = torch.randn(1, A, model.anchor_dim, device=dev)
    feat = torch.randn(1, A, EMBED, device=dev)
    maps = [torch.randn(1, CAMS, EMBED, h, w, device=dev) for h, w in DAF_LEVELS]
    pm, wh = cameras(dev)
    pts = torch.from_numpy(voxel_centres(200, 200, 16, 0.5, np.asarray(PC_RANGE[:3], dtype=np.float32))).to(dev)[None]
    for _ in range(warmup):
        labels = model(anchor, feat, maps, pm, wh, pts)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(frames):
        labels = model(anchor, feat, maps, pm, wh, pts)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / frames
    model.check()
    again = model(anchor, feat, maps, pm, wh, pts)     # the frame is reproducible bit for bit: same input, same labels
    torch.cuda.synchronize(dev)
    hist = torch.bincount(labels, minlength=18).tolist()
    out = {"config": config, "frames_per_s": 1.0 / dt, "ms_per_frame": dt * 1e3, "frames": frames, "anchors": A,
           "sample_points_per_block": A * (len(FIX_SCALE) + 2), "voxels": int(labels.numel()),
           "labels_used": int(sum(1 for h in hist if h)), "label_histogram": hist,
           "eager_labels_equal_eager": bool(torch.equal(again, labels)),
           "scope": "inference frame: feature_maps_format once, 4 encoder blocks (spconv / deformable / ffn / norm / refine in the "
                    "reference's order), fused Gaussian pre-processing, splat, occupancy labels; image backbone excluded; "
                    "FFN / LayerNorm / refine / anchor encoder / weights_fc (applied to the anchor and camera parts separately) are torch stand-ins with random weights"}
    if graph:
        # the same frame as ONE captured HIP graph: nothing in it synchronises (rulebooks are built without the host
        # read, the splat's range asserts are off) and every native op launches on torch's current stream
        try:
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                model(anchor, feat, maps, pm, wh, pts)
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                g_labels = model(anchor, feat, maps, pm, wh, pts)
            for _ in range(max(1, warmup)):
                g.replay()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(frames):
                g.replay()
            torch.cuda.synchronize(dev)
            dtg = (time.perf_counter() - t0) / frames
            model.check()
            out["frames_per_s_graph"] = 1.0 / dtg
            out["ms_per_frame_graph"] = dtg * 1e3
            out["graph_labels_equal_eager"] = bool(torch.equal(g_labels, labels))
        except Exception as exc:   # capture support is an extra; never lose the eager figure
            out["frames_per_s_graph"] = None
            out["graph_error"] = f"{type(exc).__name__}: {exc}"[:300]
    return out


def run_sharded(config, rank, world, gather, reduce_sum, barrier, max_over_ranks, frames=6, warmup=2, device="cuda:0", check=False):
    """ms per frame of the anchor-sharded frame (slowest rank), both head variants; with `check`, the fraction of voxels
    whose label equals the single-GPU frame's (GEMM tilings differ with the row count, so last-bit ties may flip)."""
    from gaussianformer_amd.synthetic import DAF_LEVELS, voxel_centres
    dev = torch.device(device)
    torch.manual_seed(0)
    model = Frame(config).to(dev).eval()
    A = model.cfg["anchors"]
    anchor = torch.randn(1, A, model.anchor_dim, device=dev)
    feat = torch.randn(1, A, EMBED, device=dev)
    maps = [torch.randn(1, CAMS, EMBED, h, w, device=dev) for h, w in DAF_LEVELS]
    pm, wh = cameras(dev)
    pts = torch.from_numpy(voxel_centres(200, 200, 16, 0.5, np.asarray(PC_RANGE[:3], dtype=np.float32))).to(dev)[None]
    out = {"config": config, "anchors": A, "anchors_per_rank": A // world, "world": world,
           "scope": "one inference frame, anchors sharded over the ranks: per-anchor encoder work on the shard (replicated pyramid, no "
                    "collective), sparse convolution replicated on the all-gathered set, head by x-slabs (labels all-gathered) or by "
                    "Gaussian shards + all-reduce of the logits"}
    for splat in ("slab", "allreduce"):
        step = lambda: model.forward_sharded(anchor, feat, maps, pm, wh, pts, rank, world, gather, reduce_sum, splat=splat)
        for _ in range(warmup):
            labels = step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(frames):
            labels = step()
        torch.cuda.synchronize(dev)
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0) / frames
        out[splat] = {"ms_per_frame": dt * 1e3, "frames_per_s": 1.0 / dt}
        if check:
            want = model(anchor, feat, maps, pm, wh, pts)
            out[splat]["labels_equal_single_gpu_fraction"] = float((labels == want).float().mean())
    model.check()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", nargs="*", default=list(FRAME_CONFIGS))
    ap.add_argument("--frames", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--graph", action="store_true", help="also replay the frame as one captured HIP graph")
    args = ap.parse_args()
    for c in args.configs:
        print(json.dumps(run(c, args.frames, args.warmup, graph=args.graph)), flush=True)
