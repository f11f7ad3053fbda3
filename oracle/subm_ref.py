"""Submanifold sparse 3-D convolution -- definition used as the checker of gf_subm_* (TEST INFRASTRUCTURE ONLY).

PARITY UNPINNED: the reference calls ``spconv.SubMConv3d`` (model/encoder/gaussian_encoder/spconv3d_module.py:28-44,
pip ``spconv-cu117``, version unpinned, docs/installation.md:26); spconv is neither in the reference tree nor
installable here, and the reference holds no test or golden vector for it.  What is restated is the published
operator: with stride 1 and padding K//2 the output sites are the input sites and

    out[i] = sum_k  sum_{j : cell(j) = cell(i) + offset_k}  feat[j] @ W[k]            (offsets in [K,K,K] order)

i.e. the dense K^3 cross-correlation of the scattered features read back at the active sites.  Where several
points share a cell spconv's hash table keeps one of them (which one is an insertion race); the well-defined
reading used here -- and by the kernels -- scatters with a SUM and gives every point of the cell its output.
Evaluated in whatever dtype the inputs carry (the tests use fp64) and differentiable through torch autograd.
"""
import torch


def subm_conv3d_dense(feat, idx, weight, batch, shape, K):
    """``feat [N,Cin]``, ``idx [N,4]`` int (batch, x, y, z), ``weight [K^3,Cin,Cout]`` -> ``out [N,Cout]``.
    Points outside the grid (or with a batch index outside [0, batch)) are inactive: they contribute
    nothing and receive zeros."""
    X, Y, Z = shape
    cout = weight.shape[2]
    cin = feat.shape[1]
    idx = idx.long()
    inside = (idx[:, 0] >= 0) & (idx[:, 0] < batch) & (idx[:, 1] >= 0) & (idx[:, 1] < X) & (idx[:, 2] >= 0) & \
        (idx[:, 2] < Y) & (idx[:, 3] >= 0) & (idx[:, 3] < Z)
    lin = ((idx[:, 0] * X + idx[:, 1]) * Y + idx[:, 2]) * Z + idx[:, 3]
    lin = torch.where(inside, lin, torch.zeros_like(lin))
    keep = inside[:, None].to(feat.dtype)
    dense = torch.zeros(batch * X * Y * Z, cin, dtype=feat.dtype).index_add(0, lin, feat * keep)
    dense = dense.view(batch, X, Y, Z, cin).permute(0, 4, 1, 2, 3)
    w = weight.view(K, K, K, cin, cout).permute(4, 3, 0, 1, 2)
    y = torch.nn.functional.conv3d(dense, w, padding=K // 2)            # cross-correlation: out[x] = sum_d in[x+d] w[d]
    y = y.permute(0, 2, 3, 4, 1).reshape(batch * X * Y * Z, cout)
    return y[lin] * keep
