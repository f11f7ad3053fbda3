"""TEST INFRASTRUCTURE ONLY -- torch restatement of the caller-side preparation of the
deformable aggregation; never imported by the product.

``project_points`` follows DeformableFeatureAggregation.project_points
(model/encoder/gaussian_encoder/deformable_module.py:268-285); ``prepare`` follows the body of
``forward`` between ``_get_weights`` and ``DAF.apply`` (:174-214), op for op, so autograd gives
the reference gradients.  ``project_points`` is pinned against tests/golden/daf_prepare.npz,
produced by executing the reference's own function (tools/make_golden_daf_prepare.py).
"""
import torch


def project_points(key_points, projection_mat, image_wh=None):
    """:268-285"""
    pts_extend = torch.cat([key_points, torch.ones_like(key_points[..., :1])], dim=-1)
    points_2d = torch.matmul(projection_mat[:, :, None, None], pts_extend[:, None, ..., None]).squeeze(-1)
    depth = points_2d[..., 2]
    points_2d = points_2d[..., :2] / torch.clamp(points_2d[..., 2:3], min=1e-5)
    if image_wh is not None:
        points_2d = points_2d / image_wh[:, :, None, None]
    mask = (depth > 1e-5) & (points_2d[..., 0] > 0) & (points_2d[..., 0] < 1) & \
        (points_2d[..., 1] > 0) & (points_2d[..., 1] < 1)
    return points_2d, mask


def prepare(key_points, projection_mat, image_wh, weights, weight_mask=None):
    """:174-214.  weights: [bs, A, cams, L, pts, G] as returned by _get_weights (:243-253)."""
    bs, num_anchor, num_cams, num_levels, num_pts, num_groups = weights.shape
    if weight_mask is None:
        weight_mask = torch.ones_like(weights) > 0
    weights = weights.permute(0, 1, 4, 2, 3, 5).contiguous().reshape(bs, num_anchor, num_pts, num_cams, num_levels, num_groups)
    weight_mask = weight_mask.permute(0, 1, 4, 2, 3, 5).contiguous().reshape(weights.shape)
    points_2d, mask = project_points(key_points, projection_mat, image_wh)
    points_2d = points_2d.permute(0, 2, 3, 1, 4).reshape(bs, num_anchor * num_pts, num_cams, 2)
    mask = mask.permute(0, 2, 3, 1)
    mask = mask[..., None, None] & weight_mask
    all_miss = mask.sum(dim=[2, 3, 4], keepdim=True) == 0
    all_miss = all_miss.expand(-1, -1, num_pts, num_cams, num_levels, -1)
    weights = weights.masked_fill(~mask, -torch.inf)      # weights[~mask] = -inf
    weights = weights.masked_fill(all_miss, 0.0)          # weights[all_miss] = 0.
    weights = weights.flatten(2, 4).softmax(dim=-2).reshape(bs, num_anchor * num_pts, num_cams, num_levels, num_groups)
    weights = weights * (1 - all_miss.flatten(1, 2).float())
    return points_2d, weights
