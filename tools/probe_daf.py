"""Development probe (GPU box): deformable aggregation forward / backward timings, uniform and projected sampling locations."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from gaussianformer_amd.deformable_aggregation import deformable_aggregation_backward, deformable_aggregation_forward
from gaussianformer_amd.synthetic import make_daf_inputs
dev = torch.device("cuda:0")
pts = int(sys.argv[1]) if len(sys.argv) > 1 else 230400
what = sys.argv[2] if len(sys.argv) > 2 else "fwd"
d = make_daf_inputs(num_pts=pts, seed=0)
feat, ss, st, loc, w = (torch.from_numpy(d[k]).to(dev) for k in ("mc_ms_feat", "spatial_shape", "scale_start_index", "sampling_location", "weights"))
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for pin in (False, True):
    print(f"uniform U(-0.2,1.2): forward{' [pinned]' if pin else ''} {timed(lambda: deformable_aggregation_forward(feat, ss, st, loc, w, pin_channel_groups=pin)):.1f} us", flush=True)
if what == "bwd":
    go = torch.randn(1, pts, 128, device=dev)
    gf, gl, gw = torch.zeros_like(feat), torch.zeros_like(loc), torch.zeros_like(w)
    print(f"uniform: backward {timed(lambda: deformable_aggregation_backward(feat, ss, st, loc, w, go, gf, gl, gw), 10):.1f} us", flush=True)

# projected geometry: 25 600 anchors uniform in the nuScenes range, 9 key points each (fixed offsets x 0.3 m scale), six pinhole
# cameras (tools/bench_frame.cameras), masked softmax weights -- what the frame benchmark feeds the op
sys.path.insert(0, "tools")
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
vis = ((loc2 > 0) & (loc2 < 1)).all(-1).float().sum(-1).mean().item()
for pin in (False, True):
    print(f"projected geometry: {vis:.2f} visible cameras per point; forward{' [pinned]' if pin else ''} {timed(lambda: deformable_aggregation_forward(feat, ss, st, loc2.contiguous(), w2.contiguous(), pin_channel_groups=pin)):.1f} us", flush=True)
if what == "bwd":
    gf, gl, gw = torch.zeros_like(feat), torch.zeros_like(loc2), torch.zeros_like(w2)
    print(f"projected: backward {timed(lambda: deformable_aggregation_backward(feat, ss, st, loc2.contiguous(), w2.contiguous(), go, gf, gl, gw), 10):.1f} us", flush=True)
