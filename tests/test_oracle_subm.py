"""The sparse-convolution checker (oracle/subm_ref.py) against a literal evaluation of the operator it states:
out[i] = sum_k sum_{j: cell(j) = cell(i) + offset_k} feat[j] @ W[k] (CPU, no GPU needed)."""
import itertools

import numpy as np
import torch

from oracle.subm_ref import subm_conv3d_dense


def _literal(feat, idx, weight, batch, shape, K):
    N = feat.shape[0]
    out = torch.zeros(N, weight.shape[2], dtype=feat.dtype)
    r = K // 2
    offs = list(itertools.product(range(-r, r + 1), repeat=3))   # [K,K,K] order: x major, z fastest
    def active(p):
        return 0 <= p[0] < batch and all(0 <= p[a + 1] < shape[a] for a in range(3))
    for i in range(N):
        pi = idx[i].tolist()
        if not active(pi):
            continue
        for k, (dx, dy, dz) in enumerate(offs):
            target = [pi[0], pi[1] + dx, pi[2] + dy, pi[3] + dz]
            for j in range(N):
                if idx[j].tolist() == target and active(target):
                    out[i] += feat[j] @ weight[k]
    return out


def test_dense_definition_equals_literal_sum():
    rng = np.random.default_rng(3)
    for K, shape, batch, N in ((3, (4, 5, 3), 2, 40), (5, (6, 4, 4), 1, 30)):
        idx = np.stack([rng.integers(0, batch, N), rng.integers(0, shape[0], N), rng.integers(0, shape[1], N),
                        rng.integers(0, shape[2], N)], 1)
        idx[5] = idx[7]                      # two points in one cell
        idx[0, 1] = shape[0] + 2             # an inactive point
        idx = torch.from_numpy(idx)
        g = torch.Generator().manual_seed(K)
        feat = torch.randn(N, 6, dtype=torch.float64, generator=g)
        w = torch.randn(K ** 3, 6, 5, dtype=torch.float64, generator=g)
        got = subm_conv3d_dense(feat, idx, w, batch, shape, K)
        want = _literal(feat, idx, w, batch, shape, K)
        assert torch.allclose(got, want, rtol=1e-12, atol=1e-12)
        assert float(got[0].abs().max()) == 0.0
        assert torch.equal(got[5], got[7])
