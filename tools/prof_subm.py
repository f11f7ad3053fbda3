"""Development aid: run the sparse-conv pieces a few times (for rocprofv3)."""
import sys
import torch
sys.path.insert(0, ".")
from gaussianformer_amd.sparse_conv import Rulebook
dev = torch.device("cuda:0")
A = int(sys.argv[1]) if len(sys.argv) > 1 else 25600
xyz = torch.rand(A, 3, device=dev) * torch.tensor([160.0, 160.0, 16.0], device=dev)
idx = torch.cat([torch.zeros(A, 1, dtype=torch.int32, device=dev), xyz.to(torch.int32)], dim=1)
feat = torch.randn(A, 128, device=dev)
w = torch.randn(125, 128, 128, device=dev) * 0.05
go = torch.randn(A, 128, device=dev)
for _ in range(5):
    rb = Rulebook(idx, 1, (160, 160, 16), 5)
    rb.apply(feat, w)
    rb.weight_grad(feat, go)
torch.cuda.synchronize()
