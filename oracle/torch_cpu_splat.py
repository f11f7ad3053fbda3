"""TEST INFRASTRUCTURE / reported baseline -- a vectorised PyTorch-CPU formulation of the base splat forward:
"pair list -> index_add_" (BASELINE.md §2b).  The reference has no CPU path for the splat,
so this is what a pure-PyTorch fallback of model/head/localagg would look like: enumerate every (Gaussian, voxel) pair of
the integer boxes (src/auxiliary.h:8-20) in chunks, evaluate forward.cu:66-69 on the pairs, and scatter-add the 18-channel
contributions into the grid with ``index_add_``.  Summation order inside a voxel is index_add_'s (not ascending Gaussian
id), so results agree with the oracle to fp32 rounding only.  Never imported by the product.
"""
import torch


def splat_forward_torch(points_int, pts, means3D, means_int, opacity, semantics, radii, cov6, H, W, D, chunk_pairs=1 << 21):
    """All tensors CPU: pts [N,3] f32 must be the dense voxel-centre grid (point n in voxel n), means3D [P,3],
    means_int [P,3] i32, opacity [P], semantics [P,18], radii [P] i32, cov6 [P,6].  Returns logits [N,18]."""
    N = pts.shape[0]
    assert N == H * W * D
    logits = torch.zeros(N, semantics.shape[1], dtype=torch.float32)
    dims = torch.tensor([H, W, D], dtype=torch.int64)
    mi = means_int.to(torch.int64)
    r = radii.to(torch.int64)[:, None]
    lo = torch.minimum(dims, torch.clamp(mi - r, min=0))
    hi = torch.minimum(dims, torch.clamp(mi + r + 1, min=0))
    ext = (hi - lo).clamp(min=0)
    vol = ext.prod(dim=1)
    order = torch.argsort(vol)               # similar box sizes share a chunk -> little padding
    start = 0
    P = means3D.shape[0]
    while start < P:
        # grow the chunk until the padded pair count reaches the budget
        end = start + 1
        e_max = ext[order[start]].clone()
        while end < P:
            e_new = torch.maximum(e_max, ext[order[end]])
            if int(e_new.prod()) * (end - start + 1) > chunk_pairs:
                break
            e_max = e_new
            end += 1
        g = order[start:end]
        ex, ey, ez = (int(v) for v in e_max)
        if ex * ey * ez > 0:
            ox = torch.arange(ex)[None, :, None, None]
            oy = torch.arange(ey)[None, None, :, None]
            oz = torch.arange(ez)[None, None, None, :]
            x = lo[g, 0][:, None, None, None] + ox
            y = lo[g, 1][:, None, None, None] + oy
            z = lo[g, 2][:, None, None, None] + oz
            ok = (x < hi[g, 0][:, None, None, None]) & (y < hi[g, 1][:, None, None, None]) & (z < hi[g, 2][:, None, None, None])
            key = ((x * W + y) * D + z)[ok]                       # [R_chunk] voxel keys == point indices
            gi = g[:, None, None, None].expand(ok.shape)[ok]
            d = means3D[gi] - pts[key]
            c = cov6[gi]
            power = c[:, 0] * d[:, 0] * d[:, 0] + c[:, 1] * d[:, 1] * d[:, 1] + c[:, 2] * d[:, 2] * d[:, 2]
            power = -0.5 * power - (c[:, 3] * d[:, 0] * d[:, 1] + c[:, 4] * d[:, 1] * d[:, 2] + c[:, 5] * d[:, 0] * d[:, 2])
            w = opacity[gi] * torch.exp(power)
            logits.index_add_(0, key, semantics[gi] * w[:, None])
        start = end
    return logits
