"""Development aid (GPU box): the deformable aggregation forward and backward (pixel-major, what the Python op uses) a few
times on ONE location distribution, for rocprofv3 kernel traces and PMC passes.

    python tools/prof_daf2.py uniform|projected [iters] [bwd|fwd|both]

uniform   = SURVEY.md §8d's op-level input (locations uniform over the image, cameras independent)
projected = six pinhole cameras, key points around random anchors (the frame benchmark's geometry)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from gaussianformer_amd.deformable_aggregation import (deformable_aggregation_backward, deformable_aggregation_forward)
from gaussianformer_amd.synthetic import make_daf_inputs

dev = torch.device("cuda:0")
dist = sys.argv[1] if len(sys.argv) > 1 else "uniform"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
what = sys.argv[3] if len(sys.argv) > 3 else "both"
pts = 230400
d = make_daf_inputs(num_pts=pts, seed=0)
feat, ss, st, loc, w = (torch.from_numpy(d[k]).to(dev) for k in ("mc_ms_feat", "spatial_shape", "scale_start_index", "sampling_location", "weights"))
if dist == "projected":
    import bench_frame
    from gaussianformer_amd.deformable_prepare import deformable_prepare
    g = torch.Generator(device="cpu").manual_seed(1)
    A = pts // 9
    lo = torch.tensor(bench_frame.PC_RANGE[:3]); hi = torch.tensor(bench_frame.PC_RANGE[3:])
    centre = lo + (hi - lo) * torch.rand(1, A, 3, generator=g)
    if os.environ.get("GF_DAF_MORTON") == "1":
        # anchors in Morton order of their (x, y) cell (2 m cells): neighbouring anchors then sample neighbouring pixels
        cell = ((centre[0, :, :2] - lo[:2]) / 2.0).long().clamp(0, 63)
        def spread(v):
            v = (v | (v << 8)) & 0x00FF00FF; v = (v | (v << 4)) & 0x0F0F0F0F; v = (v | (v << 2)) & 0x33333333; v = (v | (v << 1)) & 0x55555555
            return v
        key = spread(cell[:, 0]) | (spread(cell[:, 1]) << 1)
        centre = centre[:, torch.argsort(key)]
    offs = torch.tensor(bench_frame.FIX_SCALE + [[0.3, 0.3, 0.0], [-0.3, 0.3, 0.0]]) * 0.35
    kp = (centre[:, :, None] + offs[None, None]).to(dev)
    pm, wh = bench_frame.cameras(dev)
    raw = torch.randn(1, A, 6, 4, 9, 4, generator=g).to(dev)
    loc, w = deformable_prepare(kp, pm, wh, raw)
    loc, w = loc.contiguous(), w.contiguous()
go = torch.randn(1, pts, 128, device=dev)
for _ in range(iters):
    if what in ("fwd", "both"):
        deformable_aggregation_forward(feat, ss, st, loc, w)
    if what in ("bwd", "both"):
        gf, gl, gw = torch.zeros_like(feat), torch.zeros_like(loc), torch.zeros_like(w)
        deformable_aggregation_backward(feat, ss, st, loc, w, go, gf, gl, gw)
torch.cuda.synchronize()
