"""Pins the CPU restatement of the deformable aggregation (``gfo_daf_forward/_backward`` in oracle/gf_oracle.c)
against outputs of the REFERENCE's own pure-torch fallback (``feature_sampling`` + ``multi_view_level_fusion``,
model/encoder/gaussian_encoder/deformable_module.py:307-353), frozen in tests/golden/daf_ref.npz by
tools/make_golden_daf_ref.py (which executes the reference's functions unchanged).  CPU only."""
import os

import numpy as np
import torch

import oracle
from oracle import daf_prepare_ref

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "daf_ref.npz")


def _load():
    d = np.load(GOLDEN)
    levels = [tuple(int(v) for v in row) for row in d["levels"]]
    maps = [d[f"feature_map{i}"] for i in range(len(levels))]
    bs, cams, C = maps[0].shape[:3]
    # feature_maps_format (ops/deformable_aggregation.py:77-100): concat over pixels, channels last
    feat = np.concatenate([m.reshape(bs, cams, C, -1) for m in maps], axis=-1).transpose(0, 1, 3, 2)
    sizes = np.array([h * w for h, w in levels])
    start = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)
    w6 = d["weights"]                                            # [bs, A, cams, L, K, G]
    bs, A, cams, L, K, G = w6.shape
    weights = np.ascontiguousarray(w6.transpose(0, 1, 4, 2, 3, 5).reshape(bs, A * K, cams, L, G))
    return d, levels, np.ascontiguousarray(feat), np.array(levels, np.int32), start, weights, (bs, A, K, cams, L, G, C)


def _points_2d(d, dtype=torch.float32, grad=False):
    kp = torch.tensor(d["key_points"], dtype=dtype, requires_grad=grad)
    uv, visible = daf_prepare_ref.project_points(kp, torch.tensor(d["projection_mat"], dtype=dtype),
                                                 torch.tensor(d["image_wh"], dtype=dtype))
    bs, cams, A, K, _ = uv.shape
    return kp, uv.permute(0, 2, 3, 1, 4).reshape(bs, A * K, cams, 2), visible       # deformable_module.py:204-205


def test_forward_matches_reference_fallback():
    d, levels, feat, ss, st, weights, (bs, A, K, cams, L, G, C) = _load()
    _, loc, visible = _points_2d(d)
    assert np.array_equal(visible.numpy(), d["visible"])
    out = oracle.daf_forward(feat, ss, st, loc.detach().numpy(), weights).reshape(bs, A, K, C)
    scale = np.abs(d["output_f64"]).max()
    assert np.abs(out - d["output_f64"]).max() <= 5e-6 * scale
    assert np.abs(out - d["output_f32"]).max() <= 1e-5 * scale
    assert np.abs(d["output_f64"]).max() > 0.1          # not a vacuous comparison


def test_backward_matches_reference_fallback():
    d, levels, feat, ss, st, weights, (bs, A, K, cams, L, G, C) = _load()
    kp, loc, visible = _points_2d(d, torch.float64, grad=True)
    g = d["grad_output"].reshape(bs, A * K, C)
    gf, gl, gw = oracle.daf_backward(feat, ss, st, loc.detach().numpy().astype(np.float32), weights, g)
    # feature maps: [bs, cams, num_feat, C] back to per-level [bs, cams, C, h, w]
    for i, (h, w) in enumerate(levels):
        part = gf[:, :, st[i]:st[i] + h * w].transpose(0, 1, 3, 2).reshape(bs, cams, C, h, w)
        want = d[f"grad_feature_map{i}_f64"]
        assert np.abs(part - want).max() <= 2e-5 * max(np.abs(want).max(), 1.0), i
    # weights: the kernel skips cameras outside (0,1)^2, the fallback zero-pads them -> compare where the gate is open
    want = d["grad_weights_f64"].transpose(0, 1, 4, 2, 3, 5).reshape(bs, A * K, cams, L, G)
    open_gate = visible.numpy().transpose(0, 2, 3, 1).reshape(bs, A * K, cams)
    assert open_gate.mean() > 0.2
    assert np.abs(gw - want)[open_gate].max() <= 2e-5 * np.abs(want).max()
    assert not gw[~open_gate].any()
    # sampling locations: chained to the key points through project_points (autograd of the pinned restatement)
    loc.backward(torch.tensor(gl, dtype=torch.float64))
    want = d["grad_key_points_f64"]
    assert np.abs(kp.grad.numpy() - want).max() <= 5e-5 * np.abs(want).max()
