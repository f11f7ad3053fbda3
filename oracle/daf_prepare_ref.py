"""TEST INFRASTRUCTURE ONLY -- torch restatement of the caller-side preparation of the
deformable aggregation; never imported by the product.

``project_points`` states what DeformableFeatureAggregation.project_points computes
(model/encoder/gaussian_encoder/deformable_module.py:268-285); ``prepare`` states what the body of
``forward`` computes between ``_get_weights`` and ``DAF.apply`` (:174-214).  Everything is plain
differentiable torch, so autograd supplies the reference gradients.  ``project_points`` is pinned
bit for bit against tests/golden/daf_prepare.npz, which tools/make_golden_daf_prepare.py produced
by executing the reference's own function.
"""
import torch

DEPTH_EPS = 1e-5  # :277, :281


def safe_sigmoid(t):
    """model/utils/safe_ops.py:7-9"""
    return torch.sigmoid(torch.clamp(t, -9.21, 9.21))


def key_points(anchor, instance_feature, fix_scale, learnable_fc_weight, learnable_fc_bias, pc_range, scale_range,
               learnable_fixed_scale=1.0, xyz_activation="sigmoid", scale_activation="sigmoid"):
    """SparseGaussian3DKeyPointsGenerator.forward (deformable_module.py:51-90); an activation other than "sigmoid"
    leaves the columns as they are (:66-67, :79-80):
    ``[bs, A, 7 + k, 3]`` key points = Gaussian-frame offsets (fixed + learned) * scale, rotated by R(q)^T^T, plus the
    centre.  ``anchor [bs,A,>=10]`` is (xyz, scale, quaternion, ...) before activation.  Pinned through
    tests/golden/caller_dfa.npz (the sampling locations the reference caller handed to the op)."""
    from .prepare_ref import rotation_matrix
    bs, A = anchor.shape[:2]
    fix = anchor.new_tensor(fix_scale)
    scale = fix[None, None].tile([bs, A, 1, 1])
    if learnable_fc_weight is not None:
        k = learnable_fc_weight.shape[0] // 3
        learned = safe_sigmoid(torch.nn.functional.linear(instance_feature, learnable_fc_weight, learnable_fc_bias)
                               .reshape(bs, A, k, 3)) - 0.5
        scale = torch.cat([scale, learned * learnable_fixed_scale], dim=-2)
    gs = anchor[..., None, 3:6]
    if scale_activation == "sigmoid":
        gs = safe_sigmoid(gs)
    gs = scale_range[0] + (scale_range[1] - scale_range[0]) * gs
    kp = scale * gs
    rot = rotation_matrix(anchor[..., 6:10]).transpose(-1, -2)           # :72-73
    kp = torch.matmul(rot[:, :, None], kp[..., None]).squeeze(-1)
    xyz = anchor[..., :3]
    if xyz_activation == "sigmoid":
        xyz = safe_sigmoid(xyz)
    lo = anchor.new_tensor(pc_range[:3])
    hi = anchor.new_tensor(pc_range[3:])
    return kp + (xyz * (hi - lo) + lo).unsqueeze(2)


def project_points(key_points, projection_mat, image_wh=None):
    """key_points [b, A, p, 3], projection_mat [b, cams, 4, 4], image_wh [b, cams, 2] | None
    -> (uv [b, cams, A, p, 2], visible [b, cams, A, p])."""
    b, A, p, _ = key_points.shape
    homogeneous = torch.cat([key_points, key_points.new_ones(b, A, p, 1)], dim=-1)
    # every camera matrix applied to every point: [b, cams, 1, 1, 4, 4] @ [b, 1, A, p, 4, 1]
    cam_space = torch.matmul(projection_mat[:, :, None, None], homogeneous[:, None, :, :, :, None])[..., 0]
    depth = cam_space[..., 2]
    uv = cam_space[..., 0:2] / cam_space[..., 2:3].clamp(min=DEPTH_EPS)
    if image_wh is not None:
        uv = uv / image_wh[:, :, None, None, :]
    u, v = uv[..., 0], uv[..., 1]
    visible = (depth > DEPTH_EPS) & (u > 0) & (u < 1) & (v > 0) & (v < 1)
    return uv, visible


def prepare(key_points, projection_mat, image_wh, weights, weight_mask=None):
    """weights: raw attention logits [b, A, cams, L, p, G] (the layout `_get_weights` returns,
    :243-253); weight_mask: bool keep-mask of the same shape or None.
    -> (points_2d [b, A*p, cams, 2], weights [b, A*p, cams, L, G])."""
    b, A, cams, L, p, G = weights.shape
    # [b, A, cams, L, p, G] -> [b, A, p, cams, L, G]  (:176-193)
    logits = weights.permute(0, 1, 4, 2, 3, 5)
    keep = torch.ones_like(logits, dtype=torch.bool) if weight_mask is None else weight_mask.permute(0, 1, 4, 2, 3, 5)
    uv, visible = project_points(key_points, projection_mat, image_wh)
    points_2d = uv.permute(0, 2, 3, 1, 4).reshape(b, A * p, cams, 2)                     # :199-200
    usable = visible.permute(0, 2, 3, 1)[..., None, None] & keep                          # [b, A, p, cams, L, G] (:201-202)
    nothing = ~usable.flatten(2, 4).any(dim=2)                                            # [b, A, G]: no usable entry at all (:203)
    nothing6 = nothing[:, :, None, None, None, :].expand_as(usable)
    logits = logits.masked_fill(~usable, float("-inf")).masked_fill(nothing6, 0.0)        # :205-206
    soft = logits.flatten(2, 4).softmax(dim=2)                                            # over (p, cams, L) per group (:207)
    soft = soft * (~nothing)[:, :, None, :].to(soft.dtype)                                # :214
    return points_2d, soft.reshape(b, A, p, cams, L, G).reshape(b, A * p, cams, L, G)
