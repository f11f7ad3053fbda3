"""Independent dense fp64 formulations of the two ops, differentiable by torch autograd.

TEST INFRASTRUCTURE ONLY.  These do NOT follow the reference's kernels; they state the
mathematics directly (dense [N,P] mask-and-sum for the splat, ``grid_sample`` for the
deformable aggregation) and exist to pin the C restatement (``gf_oracle.c``) -- forward
values and, through autograd, every gradient -- on grids small enough for O(N*P) work.
"""
import math

import torch
import torch.nn.functional as F


def _box_mask(points_int, means_int, radii, H, W, D):
    """[N,P] bool: voxel of point n lies inside Gaussian g's clipped integer box
    (box = [clamp(m-r,0,G), clamp(m+r+1,0,G)) per axis)."""
    pi = points_int.long()[:, None, :]          # N,1,3
    mi = means_int.long()[None, :, :]           # 1,P,3
    r = radii.long()
    r = r[None, :, None].expand(1, -1, 3) if r.dim() == 1 else r[None, :, :]
    G = torch.tensor([H, W, D], dtype=torch.long)
    lo = torch.minimum(torch.clamp(mi - r, min=0), G)
    hi = torch.minimum(torch.clamp(mi + r + 1, min=0), G)
    return ((pi >= lo) & (pi < hi)).all(-1)


def _power(pts, means3D, cov6):
    d = means3D[None, :, :] - pts[:, None, :]   # N,P,3  (mean - point)
    dx, dy, dz = d.unbind(-1)
    xx, yy, zz, xy, yz, xz = cov6.unbind(-1)
    return -0.5 * (xx * dx * dx + yy * dy * dy + zz * dz * dz) - (xy * dx * dy + yz * dy * dz + xz * dx * dz)


def splat_dense(variant, pts, points_int, means3D, means_int, opacity, semantics, radii, cov6, H, W, D):
    """fp64 dense evaluation.  Float inputs may require grad.  Returns logits (base) or
    (logits, bin_logits, density, probability) (prob)."""
    pts = pts.double()
    means3D, opacity, semantics, cov6 = (t.double() for t in (means3D, opacity, semantics, cov6))
    mask = _box_mask(points_int, means_int, radii, H, W, D).double()
    e = torch.exp(_power(pts, means3D, cov6)) * mask            # N,P
    if variant == "base":
        return (e * opacity[None, :]) @ semantics
    xx, yy, zz, xy, yz, xz = cov6.unbind(-1)
    det = xx * yy * zz + 2 * xy * yz * xz - xx * yz * yz - yy * xz * xz - zz * xy * xy
    prob = (2 * math.pi) ** -1.5 * torch.sqrt(det)[None, :] * e * opacity[None, :]
    prob_sum = prob.sum(1)
    C = semantics.shape[1]
    num = prob @ semantics
    has = prob_sum > 1e-9
    uniform = torch.full_like(num, 1.0 / (C - 1))
    uniform[:, C - 1] = 0.0
    logits = torch.where(has[:, None], num / torch.where(has, prob_sum, torch.ones_like(prob_sum))[:, None], uniform)
    bin_logits = 1 - torch.prod(1 - e, dim=1)
    density = e.sum(1)
    return logits, bin_logits, density, prob_sum


def daf_dense(mc_ms_feat, spatial_shape, scale_start_index, sampling_location, weights):
    """fp64 deformable aggregation via ``grid_sample`` (bilinear, zero padding,
    align_corners=False: pixel = loc*size - 0.5) with the per-camera strict (0,1) gate."""
    feat = mc_ms_feat.double()
    loc = sampling_location.double()
    w = weights.double()
    B, cams, _, C = feat.shape
    pts = loc.shape[1]
    L, G = w.shape[3], w.shape[4]
    vis = ((loc[..., 0] > 0) & (loc[..., 0] < 1) & (loc[..., 1] > 0) & (loc[..., 1] < 1)).double()  # B,pts,cams
    out = torch.zeros(B, pts, C, dtype=torch.float64)
    grid = (loc * 2 - 1).permute(0, 2, 1, 3).reshape(B * cams, pts, 1, 2)  # x=w, y=h
    for lvl in range(L):
        h, wd = int(spatial_shape[lvl, 0]), int(spatial_shape[lvl, 1])
        s = int(scale_start_index[lvl])
        fm = feat[:, :, s:s + h * wd, :].reshape(B * cams, h, wd, C).permute(0, 3, 1, 2)
        samp = F.grid_sample(fm, grid, mode="bilinear", padding_mode="zeros", align_corners=False)
        samp = samp.reshape(B, cams, C, pts).permute(0, 3, 1, 2)          # B,pts,cams,C
        wl = (w[:, :, :, lvl, :] * vis[..., None]).repeat_interleave(C // G, dim=-1)  # B,pts,cams,C
        out = out + (samp * wl).sum(2)
    return out
