"""Key points of the deformable aggregation: host mirror of
``SparseGaussian3DKeyPointsGenerator`` (model/encoder/gaussian_encoder/deformable_module.py:17-90), backed by
``gf_key_points`` / ``gf_key_points_backward`` (include/gf_hip.h): one launch instead of ~25 torch kernels and a
230 400-problem batched GEMM per call."""
import ctypes

import numpy as np
import torch
import torch.nn as nn
from torch.autograd.function import Function, once_differentiable

from . import _lib

f32 = torch.float32


class KeyPointsFunction(Function):
    """``key_points [bs, A, F + K, 3] = apply(anchor [bs, A, D >= 10], learned [bs, A, K, 3] | None, fix_scale [F, 3],
    pc_range (6 floats), scale_range (2 floats), learnable_fixed_scale, identity_activations)``;
    ``identity_activations``: bit 0 = the centre columns are used without the sigmoid, bit 1 = the scale columns."""

    @staticmethod
    def forward(ctx, anchor, learned, fix_scale, pc_range, scale_range, learnable_fixed_scale, identity_activations=0):
        _lib.require_gpu(anchor, learned, fix_scale)
        lib = _lib.load()
        bs, A, D = anchor.shape
        a = anchor.detach().to(f32).contiguous()
        l = None if learned is None else learned.detach().to(f32).contiguous()
        fix = fix_scale.detach().to(f32).contiguous()
        F, K = fix.shape[0], 0 if l is None else l.shape[2]
        out = torch.empty(bs, A, F + K, 3, dtype=f32, device=a.device)
        pc = (ctypes.c_float * 6)(*[float(v) for v in pc_range])
        ctx.consts = (bs * A, D, F, K, pc, float(scale_range[0]), float(scale_range[1]), float(learnable_fixed_scale),
                      int(identity_activations))
        with torch.cuda.device(a.device):
            rc = lib.gf_key_points(bs * A, D, F, K, _lib.ptr(a), _lib.ptr(l), _lib.ptr(fix), ctypes.cast(pc, ctypes.c_void_p),
                                   ctx.consts[5], ctx.consts[6], ctx.consts[7], ctx.consts[8], _lib.ptr(out),
                                   _lib.current_stream(a.device))
        _lib.check(rc, "gf_key_points")
        ctx.save_for_backward(a, l if l is not None else torch.empty(0, device=a.device), fix)
        ctx.has_learned = l is not None
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_kp):
        a, l, fix = ctx.saved_tensors
        lib = _lib.load()
        n, D, F, K, pc, lo, hi, lfs, ident = ctx.consts
        g = grad_kp.detach().to(f32).contiguous()
        ga = torch.empty_like(a)
        gl = torch.empty_like(l) if ctx.has_learned else None
        with torch.cuda.device(a.device):
            rc = lib.gf_key_points_backward(n, D, F, K, _lib.ptr(a), _lib.ptr(l) if ctx.has_learned else None, _lib.ptr(fix),
                                            ctypes.cast(pc, ctypes.c_void_p), lo, hi, lfs, ident, _lib.ptr(g), _lib.ptr(ga), _lib.ptr(gl),
                                            _lib.current_stream(a.device))
        _lib.check(rc, "gf_key_points_backward")
        return ga, gl, None, None, None, None, None


def key_points(anchor, learned, fix_scale, pc_range, scale_range, learnable_fixed_scale=1.0, xyz_activation="sigmoid",
               scale_activation="sigmoid"):
    """Functional form; see :class:`KeyPointsFunction`.  Like the reference (deformable_module.py:66-67, :79-80), any
    activation name other than ``"sigmoid"`` means the columns are used as they are."""
    ident = (0 if xyz_activation == "sigmoid" else 1) | (0 if scale_activation == "sigmoid" else 2)
    return KeyPointsFunction.apply(anchor, learned, fix_scale, pc_range, scale_range, learnable_fixed_scale, ident)


class SparseGaussian3DKeyPointsGenerator(nn.Module):
    """Same constructor keys, parameters (``learnable_fc``) and ``forward(anchor, instance_feature)`` as the reference
    class (deformable_module.py:17-90), including the non-sigmoid (identity) activations of :66-67 and :79-80."""

    def __init__(self, embed_dims=256, num_learnable_pts=0, learnable_fixed_scale=1, fix_scale=None, pc_range=None,
                 scale_range=None, xyz_activation="sigmoid", scale_activation="sigmoid", **kwargs):
        super().__init__()
        self.xyz_act = xyz_activation
        self.scale_act = scale_activation
        self.embed_dims = embed_dims
        self.num_learnable_pts = num_learnable_pts
        self.learnable_fixed_scale = learnable_fixed_scale
        if fix_scale is None:
            fix_scale = ((0.0, 0.0, 0.0),)
        self.fix_scale = np.array(fix_scale)
        self.register_buffer("_fix", torch.tensor(self.fix_scale, dtype=f32), persistent=False)
        self.num_pts = len(self.fix_scale) + num_learnable_pts
        if num_learnable_pts > 0:
            self.learnable_fc = nn.Linear(self.embed_dims, num_learnable_pts * 3)
        self.pc_range = pc_range
        self.scale_range = scale_range

    def init_weight(self):
        if self.num_learnable_pts > 0:
            nn.init.xavier_uniform_(self.learnable_fc.weight)
            nn.init.constant_(self.learnable_fc.bias, 0.0)

    def forward(self, anchor, instance_feature=None):
        bs, A = anchor.shape[:2]
        learned = None
        if self.num_learnable_pts > 0 and instance_feature is not None:
            learned = self.learnable_fc(instance_feature).reshape(bs, A, self.num_learnable_pts, 3)
        return key_points(anchor, learned, self._fix, self.pc_range, self.scale_range, self.learnable_fixed_scale,
                          self.xyz_act, self.scale_act)
