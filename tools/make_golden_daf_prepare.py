"""Generates tests/golden/daf_prepare.npz by executing the REFERENCE's own
DeformableFeatureAggregation.project_points (model/encoder/gaussian_encoder/deformable_module.py:
268-285).  The module itself cannot be imported here (mmengine is absent), so the function's
source is cut out of the file with ``ast`` and executed unchanged.  Build container only.
Run:  python tools/make_golden_daf_prepare.py"""
import ast
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference/model/encoder/gaussian_encoder/deformable_module.py"
tree = ast.parse(open(SRC).read())
fn = None
for node in ast.walk(tree):
    if isinstance(node, ast.ClassDef) and node.name == "DeformableFeatureAggregation":
        for item in node.body:
            if isinstance(item, ast.FunctionDef) and item.name == "project_points":
                item.decorator_list = []
                fn = item
mod = ast.Module(body=[fn], type_ignores=[])
ns = {"torch": torch}
exec(compile(mod, SRC, "exec"), ns)
project_points = ns["project_points"]

rng = np.random.default_rng(77)
bs, A, pts, cams = 2, 40, 9, 6
key_points = torch.from_numpy((rng.random((bs, A, pts, 3)) * np.array([80.0, 80.0, 6.4]) + np.array([-40.0, -40.0, -1.0])).astype(np.float32))
# nuScenes-like pinhole cameras looking along six headings (lidar2img = K [R|t])
mats = []
for b in range(bs):
    per_cam = []
    for c in range(cams):
        yaw = 2 * np.pi * c / cams + 0.1 * rng.standard_normal()
        fwd = np.array([np.cos(yaw), np.sin(yaw), 0.0]); up = np.array([0.0, 0.0, 1.0]); right = np.cross(fwd, up)
        R = np.stack([right, -up, fwd])              # camera axes: x right, y down, z forward
        t = -R @ np.array([0.5 * rng.standard_normal(), 0.5 * rng.standard_normal(), 1.5])
        K = np.array([[1260.0, 0, 800.0], [0, 1260.0, 450.0], [0, 0, 1.0]])
        M = np.eye(4); M[:3, :4] = K @ np.concatenate([R, t[:, None]], axis=1)
        per_cam.append(M)
    mats.append(np.stack(per_cam))
projection_mat = torch.from_numpy(np.stack(mats).astype(np.float32))
image_wh = torch.tensor([[[1600.0, 900.0]] * cams] * bs)
points_2d, mask = project_points(key_points, projection_mat, image_wh)
points_2d_nowh, mask_nowh = project_points(key_points, projection_mat, None)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "daf_prepare.npz"), key_points=key_points.numpy(),
                    projection_mat=projection_mat.numpy(), image_wh=image_wh.numpy(), points_2d=points_2d.numpy(),
                    mask=mask.numpy(), points_2d_nowh=points_2d_nowh.numpy(), mask_nowh=mask_nowh.numpy())
print("wrote tests/golden/daf_prepare.npz; visible fraction", float(mask.float().mean()))
