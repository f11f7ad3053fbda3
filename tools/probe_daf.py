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
print(f"uniform U(-0.2,1.2): forward {timed(lambda: deformable_aggregation_forward(feat, ss, st, loc, w)):.1f} us", flush=True)
if what == "bwd":
    go = torch.randn(1, pts, 128, device=dev)
    gf, gl, gw = torch.zeros_like(feat), torch.zeros_like(loc), torch.zeros_like(w)
    print(f"uniform: backward {timed(lambda: deformable_aggregation_backward(feat, ss, st, loc, w, go, gf, gl, gw), 10):.1f} us", flush=True)
