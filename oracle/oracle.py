"""ctypes front-end of the CPU oracle (``gf_oracle.c``) plus the numpy restatement of the
reference's host-side pre-processing.  TEST INFRASTRUCTURE ONLY (see package docstring).

All arrays are numpy, C-contiguous, fp32 / int32, laid out exactly as the reference op
expects them.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgf_oracle.so")
_lib = None

__all__ = [
    "build", "num_threads", "prepare_splat_inputs", "box_offsets", "splat_forward",
    "splat_backward", "daf_forward", "daf_backward",
]


def build(force=False):
    """Compile ``gf_oracle.c`` with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "gf_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B"], check=True, capture_output=True)
    return _LIB_PATH


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.gfo_splat_forward.restype = ctypes.c_int64
        _lib.gfo_box_offsets.restype = ctypes.c_int64
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def num_threads():
    return int(_load().gfo_num_threads())


def prepare_splat_inputs(pts, means3D, scales, cov3D, pc_min, grid_size, scale_multiplier,
                         per_axis=False, radii_min=None):
    """numpy restatement of ``LocalAggregator.forward``'s pre-processing.

    Follows model/head/localagg/local_aggregate/__init__.py:137-143 (base),
    model/head/localagg_prob/local_aggregate_prob/__init__.py:147-154 (``radii_min`` clamp)
    and model/head/localagg_prob_fast/local_aggregate_prob_fast/__init__.py:151 (per-axis
    radii).  All arithmetic is fp32; ``.to(torch.int)`` == C truncation toward zero.
    Inputs are already squeezed: pts [N,3], means3D [P,3], scales [P,3], cov3D [P,3,3].
    """
    pts = _f32(pts)
    means3D = _f32(means3D)
    scales = _f32(scales)
    pc_min = _f32(pc_min).reshape(1, 3)
    gs = np.float32(grid_size)
    sm = np.float32(scale_multiplier)
    points_int = ((pts - pc_min) / gs).astype(np.int32)          # :137
    means_int = ((means3D - pc_min) / gs).astype(np.int32)       # :139
    if per_axis:
        radii = np.ceil(scales * sm / gs).astype(np.int32)       # prob_fast :151
    else:
        radii = np.ceil(scales.max(axis=-1) * sm / gs).astype(np.int32)  # :141
    if radii_min is not None:
        radii = np.maximum(radii, np.int32(radii_min))           # prob :152
    cov6 = _f32(cov3D).reshape(-1, 9)[:, [0, 4, 8, 1, 5, 2]]     # :143
    return points_int, means_int, np.ascontiguousarray(radii), np.ascontiguousarray(cov6)


def box_offsets(means_int, radii, H, W, D):
    """(tiles_touched[P], offsets[P] inclusive scan, num_rendered) --
    src/forward.cu:9-28 + src/aggregator_impl.cu:193-197."""
    means_int = _i32(means_int)
    radii = _i32(radii)
    P = means_int.shape[0]
    per_axis = int(radii.ndim == 2)
    touched = np.zeros(P, dtype=np.uint32)
    offsets = np.zeros(P, dtype=np.uint32)
    R = _load().gfo_box_offsets(P, H, W, D, _p(means_int), _p(radii), per_axis, _p(touched), _p(offsets))
    return touched, offsets, int(R)


def splat_forward(variant, pts, points_int, means3D, means_int, opacity, semantics, radii, cov6,
                  H, W, D, nthreads=0):
    """variant: 'base' | 'prob'.  ``radii`` [P] or [P,3] (prob_fast).
    Returns dict(logits[N,C] (+ bin_logits, density, probability for prob), num_rendered)."""
    pts, means3D, opacity, semantics, cov6 = map(_f32, (pts, means3D, opacity, semantics, cov6))
    points_int, means_int, radii = map(_i32, (points_int, means_int, radii))
    N, P, C = pts.shape[0], means3D.shape[0], semantics.shape[1]
    per_axis = int(radii.ndim == 2)
    v = {"base": 0, "prob": 1}[variant]
    logits = np.zeros((N, C), dtype=np.float32)
    outs = [np.zeros(N, dtype=np.float32) for _ in range(3)] if v else [None] * 3
    R = _load().gfo_splat_forward(v, per_axis, P, N, C, H, W, D, _p(pts), _p(points_int), _p(means3D),
                                  _p(means_int), _p(opacity), _p(semantics), _p(radii), _p(cov6),
                                  _p(logits), _p(outs[0]), _p(outs[1]), _p(outs[2]), int(nthreads))
    if R < 0:
        raise RuntimeError(f"gfo_splat_forward failed: {R}")
    res = {"logits": logits, "num_rendered": int(R)}
    if v:
        res.update(bin_logits=outs[0], density=outs[1], probability=outs[2])
    return res


def splat_backward(variant, pts, points_int, means3D, means_int, opacity, semantics, radii, cov6,
                   H, W, D, out_grad, fwd=None, bin_grad=None, density_grad=None, nthreads=0):
    """Returns (means3D_grad[P,3], opacity_grad[P], semantics_grad[P,C], cov3D_grad[P,6]).
    For 'prob', ``fwd`` is the dict returned by :func:`splat_forward` and the three incoming
    grads are (out_grad=logits_grad, bin_grad, density_grad)."""
    pts, means3D, opacity, semantics, cov6 = map(_f32, (pts, means3D, opacity, semantics, cov6))
    points_int, means_int, radii = map(_i32, (points_int, means_int, radii))
    out_grad = _f32(out_grad)
    N, P, C = pts.shape[0], means3D.shape[0], semantics.shape[1]
    per_axis = int(radii.ndim == 2)
    mg = np.zeros((P, 3), np.float32)
    og = np.zeros(P, np.float32)
    sg = np.zeros((P, C), np.float32)
    cg = np.zeros((P, 6), np.float32)
    lib = _load()
    if variant == "base":
        rc = lib.gfo_splat_backward_base(per_axis, P, N, C, H, W, D, _p(pts), _p(points_int), _p(means3D),
                                         _p(means_int), _p(opacity), _p(semantics), _p(radii), _p(cov6),
                                         _p(out_grad), _p(mg), _p(og), _p(sg), _p(cg), int(nthreads))
    else:
        bin_grad, density_grad = _f32(bin_grad), _f32(density_grad)
        rc = lib.gfo_splat_backward_prob(per_axis, P, N, C, H, W, D, _p(pts), _p(points_int), _p(means3D),
                                         _p(means_int), _p(opacity), _p(semantics), _p(radii), _p(cov6),
                                         _p(_f32(fwd["logits"])), _p(_f32(fwd["bin_logits"])),
                                         _p(_f32(fwd["density"])), _p(_f32(fwd["probability"])),
                                         _p(out_grad), _p(bin_grad), _p(density_grad),
                                         _p(mg), _p(og), _p(sg), _p(cg), int(nthreads))
    if rc:
        raise RuntimeError(f"gfo_splat_backward failed: {rc}")
    return mg, og, sg, cg


def daf_forward(mc_ms_feat, spatial_shape, scale_start_index, sampling_location, weights, nthreads=0):
    """mc_ms_feat [B,cams,num_feat,C]; spatial_shape [L,2]; scale_start_index [L];
    sampling_location [B,pts,cams,2]; weights [B,pts,cams,L,G] -> [B,pts,C]."""
    feat, loc, w = _f32(mc_ms_feat), _f32(sampling_location), _f32(weights)
    ss, st = _i32(spatial_shape), _i32(scale_start_index)
    B, cams, num_feat, C = feat.shape
    L, pts, G = ss.shape[0], loc.shape[1], w.shape[4]
    out = np.zeros((B, pts, C), np.float32)
    _load().gfo_daf_forward(B, cams, num_feat, C, L, pts, G, _p(feat), _p(ss), _p(st), _p(loc), _p(w),
                            _p(out), int(nthreads))
    return out


def daf_backward(mc_ms_feat, spatial_shape, scale_start_index, sampling_location, weights, grad_output):
    """Returns (grad_mc_ms_feat, grad_sampling_location, grad_weights)."""
    feat, loc, w, go = _f32(mc_ms_feat), _f32(sampling_location), _f32(weights), _f32(grad_output)
    ss, st = _i32(spatial_shape), _i32(scale_start_index)
    B, cams, num_feat, C = feat.shape
    L, pts, G = ss.shape[0], loc.shape[1], w.shape[4]
    gf, gl, gw = np.zeros_like(feat), np.zeros_like(loc), np.zeros_like(w)
    _load().gfo_daf_backward(B, cams, num_feat, C, L, pts, G, _p(feat), _p(ss), _p(st), _p(loc), _p(w),
                             _p(go), _p(gf), _p(gl), _p(gw))
    return gf, gl, gw
