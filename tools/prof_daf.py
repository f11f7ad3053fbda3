"""Development aid: run the deformable aggregation backward a few times (for rocprofv3)."""
import sys
import torch
sys.path.insert(0, ".")
from gaussianformer_amd.deformable_aggregation import deformable_aggregation_backward, deformable_aggregation_forward
from gaussianformer_amd.synthetic import make_daf_inputs
dev = torch.device("cuda:0")
pts = int(sys.argv[1]) if len(sys.argv) > 1 else 230400
d = make_daf_inputs(num_pts=pts, seed=0)
feat, ss, st, loc, w = (torch.from_numpy(d[k]).to(dev) for k in ("mc_ms_feat", "spatial_shape", "scale_start_index", "sampling_location", "weights"))
go = torch.randn(1, pts, 128, device=dev)
for _ in range(5):
    deformable_aggregation_forward(feat, ss, st, loc, w)
    gf, gl, gw = torch.zeros_like(feat), torch.zeros_like(loc), torch.zeros_like(w)
    deformable_aggregation_backward(feat, ss, st, loc, w, go, gf, gl, gw)
torch.cuda.synchronize()
