"""Runs the fused DAF preparation forward + backward a few times (for rocprofv3 runs)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from gaussianformer_amd.deformable_prepare import deformable_prepare
dev = torch.device("cuda:0")
A, pts, cams, L, G = 25600, 9, 6, 4, 4
torch.manual_seed(0)
kp = (torch.rand(1, A, pts, 3, device=dev) * torch.tensor([80.0, 80.0, 6.4], device=dev) + torch.tensor([-40.0, -40.0, -1.0], device=dev)).requires_grad_(True)
pm = torch.eye(4, device=dev).repeat(1, cams, 1, 1)
for c in range(cams):
    yaw = 2 * np.pi * c / cams
    R = torch.tensor([[-np.sin(yaw), np.cos(yaw), 0.0], [0.0, 0.0, -1.0], [np.cos(yaw), np.sin(yaw), 0.0]], dtype=torch.float32)
    K = torch.tensor([[1260.0, 0, 800.0], [0, 1260.0, 450.0], [0, 0, 1.0]])
    pm[0, c, :3, :3] = (K @ R).to(dev)
    pm[0, c, :3, 3] = (K @ torch.tensor([0.0, 1.5, 0.0])).to(dev)
wh = torch.tensor([[[1600.0, 900.0]] * cams], device=dev)
raw = torch.randn(1, A, cams, L, pts, G, device=dev, requires_grad=True)
for _ in range(5):
    p, w = deformable_prepare(kp, pm, wh, raw)
    (p.sum() + (w * w).sum()).backward()
torch.cuda.synchronize()
