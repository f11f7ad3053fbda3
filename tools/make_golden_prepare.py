"""Generates tests/golden/prepare.npz by running the REFERENCE's own get_rotation_matrix
(model/utils/utils.py:20-69, imported from /root/reference -- only available in the build
container) followed by the five lines of GaussianHead.prepare_gaussian_args
(model/head/gaussian_head.py:108-119) in fp32 on the CPU, exactly as the reference does.
The fixture pins oracle/prepare_ref.py.   Run:  python tools/make_golden_prepare.py"""
import importlib.util
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("ref_utils", "/root/reference/model/utils/utils.py")
ref_utils = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref_utils)

rng = np.random.default_rng(2024)
P = 96
means = torch.from_numpy((rng.random((1, P, 3)) * np.array([80.0, 80.0, 6.4]) + np.array([-40.0, -40.0, -1.0])).astype(np.float32))
scales = torch.from_numpy((0.08 + 0.56 * rng.random((1, P, 3))).astype(np.float32))
rotations = torch.from_numpy(rng.standard_normal((1, P, 4)).astype(np.float32))   # un-normalised, like the network's output

# --- the reference's lines, verbatim in behaviour (gaussian_head.py:108-119)
bs, g, _ = means.shape
S = torch.zeros(bs, g, 3, 3, dtype=means.dtype)
S[..., 0, 0] = scales[..., 0]
S[..., 1, 1] = scales[..., 1]
S[..., 2, 2] = scales[..., 2]
R = ref_utils.get_rotation_matrix(rotations)
M = torch.matmul(S, R)
Cov = torch.matmul(M.transpose(-1, -2), M)
CovInv = Cov.cpu().inverse()

# --- the wrapper's integer path (local_aggregate/__init__.py:139-143), nuScenes grid
pc_min = torch.tensor([[-40.0, -40.0, -1.0]])
grid_size, scale_multiplier = 0.4, 3
means3D_int = ((means[0] - pc_min) / grid_size).to(torch.int)
radii = torch.ceil(scales[0].max(dim=-1)[0] * scale_multiplier / grid_size).to(torch.int)
radii_axis = torch.ceil(scales[0] * scale_multiplier / grid_size).to(torch.int).clamp(min=1)
cov6 = CovInv[0].flatten(1)[:, [0, 4, 8, 1, 5, 2]]

np.savez_compressed(os.path.join(ROOT, "tests", "golden", "prepare.npz"),
                    means=means.numpy(), scales=scales.numpy(), rotations=rotations.numpy(), R=R.numpy(),
                    Cov=Cov.numpy(), CovInv=CovInv.numpy(), pc_min=pc_min.numpy(), grid_size=grid_size,
                    scale_multiplier=scale_multiplier, means3D_int=means3D_int.numpy(), radii=radii.numpy(),
                    radii_axis=radii_axis.numpy(), cov6=cov6.numpy())
print("wrote tests/golden/prepare.npz")
