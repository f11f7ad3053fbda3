"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the per-Gaussian pre-processing that
precedes the splat; never imported by the product.

Follows GaussianHead.prepare_gaussian_args (model/head/gaussian_head.py:108-119) and
get_rotation_matrix (model/utils/utils.py:20-69) step by step in torch so that autograd
provides the reference gradient: S, R, M = S R, Cov = M^T M, CovInv = inverse(Cov).
Pinned against tests/golden/prepare.npz, which tools/make_golden_prepare.py produced by
running the reference's own get_rotation_matrix + the same five lines in fp32.
"""
import torch

# (row, col, component, sign) of the two 4x4 factor matrices, model/utils/utils.py:24-64
_MAT1 = [(0, 0, 0, 1), (0, 1, 1, -1), (0, 2, 2, -1), (0, 3, 3, -1),
         (1, 0, 1, 1), (1, 1, 0, 1), (1, 2, 3, -1), (1, 3, 2, 1),
         (2, 0, 2, 1), (2, 1, 3, 1), (2, 2, 0, 1), (2, 3, 1, -1),
         (3, 0, 3, 1), (3, 1, 2, -1), (3, 2, 1, 1), (3, 3, 0, 1)]
_MAT2 = [(0, 0, 0, 1), (0, 1, 1, -1), (0, 2, 2, -1), (0, 3, 3, -1),
         (1, 0, 1, 1), (1, 1, 0, 1), (1, 2, 3, 1), (1, 3, 2, -1),
         (2, 0, 2, 1), (2, 1, 3, -1), (2, 2, 0, 1), (2, 3, 1, 1),
         (3, 0, 3, 1), (3, 1, 2, 1), (3, 2, 1, -1), (3, 3, 0, 1)]


def rotation_matrix(q):
    """utils.py:20-69: normalise, build the two 4x4 factors, multiply, drop row/column 0."""
    q = torch.nn.functional.normalize(q, dim=-1)
    rows1 = [[None] * 4 for _ in range(4)]
    rows2 = [[None] * 4 for _ in range(4)]
    for r, c, k, s in _MAT1:
        rows1[r][c] = s * q[..., k]
    for r, c, k, s in _MAT2:
        rows2[r][c] = s * q[..., k]
    m1 = torch.stack([torch.stack(r, dim=-1) for r in rows1], dim=-2)
    m2 = torch.stack([torch.stack(r, dim=-1) for r in rows2], dim=-2)
    return torch.matmul(m1, m2.transpose(-1, -2))[..., 1:, 1:]


def covariance_inverse(scales, rotations):
    """gaussian_head.py:108-119 (the host round trip of :119 is the identity for autograd)."""
    S = torch.diag_embed(scales)
    R = rotation_matrix(rotations)
    M = torch.matmul(S, R)
    cov = torch.matmul(M.transpose(-1, -2), M)
    return torch.linalg.inv(cov)


def pack6(cov):
    """cov3D.flatten(1)[:, [0,4,8,1,5,2]]   (local_aggregate/__init__.py:143)"""
    return cov.flatten(-2)[..., [0, 4, 8, 1, 5, 2]]
