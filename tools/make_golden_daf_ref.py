"""Generates tests/golden/daf_ref.npz by executing the REFERENCE's own pure-torch deformable-aggregation
fallback: ``DeformableFeatureAggregation.project_points`` / ``feature_sampling`` / ``multi_view_level_fusion``
(model/encoder/gaussian_encoder/deformable_module.py:287-353).  The module cannot be imported here (mmengine
is absent) and the fallback is dead code behind an assert (:119-120), so the three methods are cut out of the
file with ``ast`` and executed unchanged inside a bare class of the same name.  Build container only.

The fixture holds the inputs, the fallback's output ``[bs, A, K, C]`` and autograd gradients with respect to the
feature maps, the attention weights and the key points, in fp32 (what the reference would compute) and fp64.
Weights are masked the way the real forward masks them (:199-224: zero wherever the camera does not see the
point), because the fallback zero-pads out-of-view cameras while the CUDA kernel skips them.

Run:  python tools/make_golden_daf_ref.py
"""
import ast
import os
from typing import List, Optional  # noqa: F401  (names used by the reference's annotations)

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference/model/encoder/gaussian_encoder/deformable_module.py"
WANTED = ("project_points", "feature_sampling", "multi_view_level_fusion")

tree = ast.parse(open(SRC).read())
methods = []
for node in ast.walk(tree):
    if isinstance(node, ast.ClassDef) and node.name == "DeformableFeatureAggregation":
        methods = [item for item in node.body if isinstance(item, ast.FunctionDef) and item.name in WANTED]
assert len(methods) == 3
cls = ast.ClassDef(name="DeformableFeatureAggregation", bases=[], keywords=[], body=methods, decorator_list=[])
mod = ast.fix_missing_locations(ast.Module(body=[cls], type_ignores=[]))
ns = {"torch": torch, "List": List, "Optional": Optional}
exec(compile(mod, SRC, "exec"), ns)
DFA = ns["DeformableFeatureAggregation"]

rng = np.random.default_rng(91)
bs, A, K, cams, C, G = 2, 24, 5, 3, 16, 4
levels = ((9, 16), (5, 8), (3, 4))
L = len(levels)
key_points = (rng.random((bs, A, K, 3)) * np.array([60.0, 60.0, 5.0]) + np.array([-30.0, -30.0, -1.0])).astype(np.float32)
mats = []
for b in range(bs):
    per_cam = []
    for c in range(cams):
        yaw = 2 * np.pi * c / cams + 0.1 * rng.standard_normal()
        fwd = np.array([np.cos(yaw), np.sin(yaw), 0.0]); up = np.array([0.0, 0.0, 1.0]); right = np.cross(fwd, up)
        R = np.stack([right, -up, fwd])
        t = -R @ np.array([0.5 * rng.standard_normal(), 0.5 * rng.standard_normal(), 1.5])
        Kmat = np.array([[400.0, 0, 800.0], [0, 400.0, 450.0], [0, 0, 1.0]])   # wide lens: most points are seen by a camera
        M = np.eye(4); M[:3, :4] = Kmat @ np.concatenate([R, t[:, None]], axis=1)
        per_cam.append(M)
    mats.append(np.stack(per_cam))
projection_mat = np.stack(mats).astype(np.float32)
image_wh = np.array([[[1600.0, 900.0]] * cams] * bs, dtype=np.float32)
feature_maps = [rng.standard_normal((bs, cams, C, h, w)).astype(np.float32) for h, w in levels]
raw = rng.standard_normal((bs, A, cams, L, K, G)).astype(np.float32)
grad_out = rng.standard_normal((bs, A, K, C)).astype(np.float32)

out = {}
for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
    kp = torch.tensor(key_points, dtype=dt, requires_grad=True)
    pm = torch.tensor(projection_mat, dtype=dt)
    wh = torch.tensor(image_wh, dtype=dt)
    fms = [torch.tensor(f, dtype=dt, requires_grad=True) for f in feature_maps]
    with torch.no_grad():
        _, visible = DFA.project_points(kp, pm, wh)                      # [bs, cams, A, K]
    vis = visible.permute(0, 2, 1, 3)[:, :, :, None, :, None]            # -> [bs, A, cams, 1, K, 1]
    logits = torch.tensor(raw, dtype=dt).masked_fill(~vis.expand(bs, A, cams, L, K, G), float("-inf"))
    flat = logits.permute(0, 1, 5, 2, 3, 4).reshape(bs, A, G, -1)        # softmax over (cams, L, K) per group
    none = torch.isinf(flat).all(dim=-1, keepdim=True)
    soft = torch.where(none, torch.zeros_like(flat), flat.masked_fill(none, 0.0).softmax(dim=-1))
    weights = soft.reshape(bs, A, G, cams, L, K).permute(0, 1, 3, 4, 5, 2).contiguous().detach().requires_grad_(True)
    obj = object.__new__(DFA)
    obj.num_groups, obj.group_dims, obj.num_pts, obj.embed_dims = G, C // G, K, C
    feats = DFA.feature_sampling(fms, kp, pm, wh)                        # the reference's grid_sample path
    fused = obj.multi_view_level_fusion(feats, weights)                  # [bs, A, K, C]
    (fused * torch.tensor(grad_out, dtype=dt)).sum().backward()
    out[f"output_{tag}"] = fused.detach().numpy()
    out[f"grad_weights_{tag}"] = weights.grad.numpy()
    out[f"grad_key_points_{tag}"] = kp.grad.numpy()
    for i, f in enumerate(fms):
        out[f"grad_feature_map{i}_{tag}"] = f.grad.numpy()
    if tag == "f32":
        out["weights"] = weights.detach().numpy()
        out["visible"] = visible.numpy()

np.savez_compressed(os.path.join(ROOT, "tests", "golden", "daf_ref.npz"), key_points=key_points, projection_mat=projection_mat,
                    image_wh=image_wh, grad_output=grad_out, levels=np.array(levels, dtype=np.int32),
                    **{f"feature_map{i}": f for i, f in enumerate(feature_maps)}, **out)
print("wrote tests/golden/daf_ref.npz; visible fraction", float(out["visible"].mean()),
      "| max |f32 - f64| of the output", float(np.abs(out["output_f32"] - out["output_f64"]).max()))
