"""Fused caller-side preparation of the deformable aggregation.

Host mirror of the block between the ``weights_fc`` GEMM and ``DAF.apply`` in
``DeformableFeatureAggregation.forward`` (model/encoder/gaussian_encoder/deformable_module.py:
174-214) and of ``project_points`` (:268-285), backed by ``gf_daf_prepare`` /
``gf_daf_prepare_backward`` (include/gf_hip.h).  One kernel reads the raw attention logits
once and writes the sampling locations and the softmaxed weights in the layouts ``DAF.apply``
takes; the reference runs ~15 permute / mask / softmax kernels over the 88-498 MB tensor.
"""
import torch
from torch.autograd.function import Function, once_differentiable

from . import _lib

f32 = torch.float32


def _c(t, dtype=f32):
    return None if t is None else t.detach().to(dtype).contiguous()


class DeformablePrepareFunction(Function):
    """``(points_2d [bs, A*pts, cams, 2], weights [bs, A*pts, cams, L, G]) =
    apply(key_points [bs,A,pts,3], projection_mat [bs,cams,4,4], image_wh [bs,cams,2] | None,
    raw_weights [bs,A,cams,L,pts,G], weight_mask bool same shape | None)``."""

    @staticmethod
    def forward(ctx, key_points, projection_mat, image_wh, raw_weights, weight_mask):
        _lib.require_gpu(key_points, projection_mat, image_wh, raw_weights, weight_mask)
        lib = _lib.load()
        kp, pm, wh, raw = _c(key_points), _c(projection_mat), _c(image_wh), _c(raw_weights)
        wm = None if weight_mask is None else weight_mask.detach().to(torch.uint8).contiguous()
        B, A, pts = kp.shape[:3]
        cams, L, G = raw.shape[2], raw.shape[3], raw.shape[5]
        assert raw.shape == (B, A, cams, L, pts, G) and pm.shape == (B, cams, 4, 4)
        points_2d = torch.empty(B, A * pts, cams, 2, dtype=f32, device=kp.device)
        weights = torch.empty(B, A * pts, cams, L, G, dtype=f32, device=kp.device)
        with torch.cuda.device(kp.device):
            rc = lib.gf_daf_prepare(B, A, pts, cams, L, G, _lib.ptr(kp), _lib.ptr(pm), _lib.ptr(wh), _lib.ptr(raw),
                                    _lib.ptr(wm), _lib.ptr(points_2d), _lib.ptr(weights), _lib.current_stream(kp.device))
        _lib.check(rc, "gf_daf_prepare")
        ctx.save_for_backward(kp, pm, wh if wh is not None else torch.empty(0, device=kp.device), weights)
        ctx.has_wh = wh is not None
        ctx.dims = (B, A, pts, cams, L, G)
        ctx.mark_non_differentiable()
        return points_2d, weights

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_points_2d, grad_weights):
        kp, pm, wh, weights = ctx.saved_tensors
        lib = _lib.load()
        B, A, pts, cams, L, G = ctx.dims
        need_kp, need_raw = ctx.needs_input_grad[0], ctx.needs_input_grad[3]
        g_kp = torch.empty_like(kp) if need_kp else None
        g_raw = torch.empty(B, A, cams, L, pts, G, dtype=f32, device=kp.device) if need_raw else None
        with torch.cuda.device(kp.device):
            rc = lib.gf_daf_prepare_backward(B, A, pts, cams, L, G, _lib.ptr(kp), _lib.ptr(pm),
                                             _lib.ptr(wh) if ctx.has_wh else None, _lib.ptr(weights),
                                             _lib.ptr(_c(grad_weights)), _lib.ptr(_c(grad_points_2d)), _lib.ptr(g_raw),
                                             _lib.ptr(g_kp), _lib.current_stream(kp.device))
        _lib.check(rc, "gf_daf_prepare_backward")
        return g_kp, None, None, g_raw, None


def deformable_prepare(key_points, projection_mat, image_wh, raw_weights, weight_mask=None):
    """Functional form; see :class:`DeformablePrepareFunction`."""
    return DeformablePrepareFunction.apply(key_points, projection_mat, image_wh, raw_weights, weight_mask)


def deformable_fused_forward(key_points, projection_mat, image_wh, mc_ms_feat, spatial_shape, scale_start_index,
                             raw_weights=None, raw_anchor=None, raw_cam=None):
    """INFERENCE only (no autograd): ``features [bs, A, C]`` of ``DeformableFeatureAggregation.forward`` -- project_points,
    mask / all_miss / softmax, ``DAF.apply`` and ``features.sum(dim=2)`` (deformable_module.py:174-233,242) -- in ONE launch
    (``gf_daf_fused_forward``): no ``[A*pts, cams, L, G]`` weights tensor, no ``[A*pts, C]`` sampled features.
    The attention logits come either as ``raw_weights [bs, A, cams, L, pts, G]`` (the ``weights_fc`` output as the reference
    reshapes it, :243-253) or -- ``use_camera_embed``, where ``weights_fc`` acts on ``feature + camera_embed`` and is linear
    (:253-262) -- as its two parts ``raw_anchor [bs, A, L, pts, G]`` and ``raw_cam [bs, cams, L, pts, G]``, added on the fly."""
    _lib.require_gpu(key_points, projection_mat, image_wh, mc_ms_feat, spatial_shape, scale_start_index, raw_weights, raw_anchor, raw_cam)
    if any(t is not None and t.requires_grad for t in (key_points, mc_ms_feat, raw_weights, raw_anchor, raw_cam)) and torch.is_grad_enabled():
        raise RuntimeError("deformable_fused_forward computes no gradients: use deformable_prepare + DeformableAggregationFunction "
                           "for training, or call it under torch.no_grad()")
    lib = _lib.load()
    kp, pm, wh = _c(key_points), _c(projection_mat), _c(image_wh)
    raw, ra, rc_ = _c(raw_weights), _c(raw_anchor), _c(raw_cam)
    feat = _c(mc_ms_feat)
    B, A, pts = kp.shape[:3]
    cams, num_feat, C = feat.shape[1], feat.shape[2], feat.shape[3]
    L = spatial_shape.shape[0]
    if raw is not None:
        G = raw.shape[5]
        assert raw.shape == (B, A, cams, L, pts, G) and ra is None and rc_ is None
    else:
        G = ra.shape[4]
        assert ra.shape == (B, A, L, pts, G) and rc_.shape == (B, cams, L, pts, G)
    assert pm.shape == (B, cams, 4, 4)
    ss, st = spatial_shape.to(torch.int32).contiguous(), scale_start_index.to(torch.int32).contiguous()
    out = torch.empty(B, A, C, dtype=f32, device=kp.device)
    with torch.cuda.device(kp.device):
        rc = lib.gf_daf_fused_forward(B, A, pts, cams, L, G, C, num_feat, _lib.ptr(kp), _lib.ptr(pm), _lib.ptr(wh), _lib.ptr(raw),
                                      _lib.ptr(ra), _lib.ptr(rc_), _lib.ptr(feat), _lib.ptr(ss), _lib.ptr(st), _lib.ptr(out),
                                      _lib.current_stream(kp.device))
    _lib.check(rc, "gf_daf_fused_forward")
    return out
