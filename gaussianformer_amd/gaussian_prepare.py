"""Fused per-Gaussian pre-processing for the splat.

Host mirror of ``GaussianHead.prepare_gaussian_args`` (model/head/gaussian_head.py:82-120)
and of the integer path of ``LocalAggregator.forward``
(model/head/localagg/local_aggregate/__init__.py:137-143), backed by
``gf_gaussian_prepare`` / ``gf_gaussian_prepare_backward`` (include/gf_hip.h).  The reference
moves the covariances to the host for a LAPACK inverse and synchronises ~10 times per frame;
here one kernel does it on the device and nothing synchronises.
"""
import torch
import torch.nn as nn

from . import _lib

f32, i32 = torch.float32, torch.int32


def _c(t, dtype=f32):
    return t.detach().to(dtype).contiguous()


def gaussian_prepare(means3D, scales, rotations, pc_min, grid_size, scale_multiplier, H, W, D,
                     radii_mode=_lib.GF_RADII_SCALAR, radii_min=1, full_cov=False, status=None):
    """One launch: ``(means3D_int [P,3] i32, radii [P] | [P,3] i32, cov)`` with ``cov`` =
    Sigma^-1 packed ``[P,6]`` (xx,yy,zz,xy,yz,xz) or, with ``full_cov``, ``[P,3,3]``.
    ``status`` (optional zeroed int32[1] device tensor) collects the GF_PREPARE_* bits that
    stand in for the reference's host-side asserts; no gradient is recorded here."""
    _lib.require_gpu(means3D, scales, rotations, status)
    lib = _lib.load()
    means3D, scales, rotations = _c(means3D), _c(scales), _c(rotations)
    P = means3D.shape[0]
    assert means3D.shape == (P, 3) and scales.shape == (P, 3) and rotations.shape == (P, 4)
    dev = means3D.device
    means_int = torch.empty(P, 3, dtype=i32, device=dev)
    radii = torch.empty((P, 3) if radii_mode == _lib.GF_RADII_PER_AXIS else (P,), dtype=i32, device=dev)
    cov = torch.empty((P, 3, 3) if full_cov else (P, 6), dtype=f32, device=dev)
    import ctypes
    pc = (ctypes.c_float * 3)(*[float(v) for v in pc_min])
    with torch.cuda.device(dev):
        rc = lib.gf_gaussian_prepare(P, H, W, D, ctypes.cast(pc, ctypes.c_void_p), float(grid_size),
                                     float(scale_multiplier), int(radii_mode), int(radii_min), _lib.ptr(means3D),
                                     _lib.ptr(scales), _lib.ptr(rotations), _lib.ptr(means_int), _lib.ptr(radii),
                                     None if full_cov else _lib.ptr(cov), _lib.ptr(cov) if full_cov else None,
                                     _lib.ptr(status), _lib.current_stream(dev))
    _lib.check(rc, "gf_gaussian_prepare")
    return means_int, radii, cov


class _CovInverse(torch.autograd.Function):
    """Sigma^-1(scales, rotations), differentiable: replaces the autograd graph through
    S, R, M = S R, Cov = M^T M and ``Cov.cpu().inverse().cuda()`` (gaussian_head.py:108-119)."""

    @staticmethod
    def forward(ctx, scales, rotations, packed):
        _lib.require_gpu(scales, rotations)
        lib = _lib.load()
        s, q = _c(scales), _c(rotations)
        P = s.shape[0]
        cov = torch.empty((P, 6) if packed else (P, 3, 3), dtype=f32, device=s.device)
        import ctypes
        pc = (ctypes.c_float * 3)(0.0, 0.0, 0.0)
        with torch.cuda.device(s.device):
            rc = lib.gf_gaussian_prepare(P, 1, 1, 1, ctypes.cast(pc, ctypes.c_void_p), 1.0, 1.0, _lib.GF_RADII_SCALAR, 1,
                                         None, _lib.ptr(s), _lib.ptr(q), None, None,
                                         _lib.ptr(cov) if packed else None, None if packed else _lib.ptr(cov), None,
                                         _lib.current_stream(s.device))
        _lib.check(rc, "gf_gaussian_prepare")
        ctx.save_for_backward(s, q)
        ctx.packed = packed
        return cov

    @staticmethod
    def backward(ctx, cov_grad):
        s, q = ctx.saved_tensors
        lib = _lib.load()
        g = _c(cov_grad)
        P = s.shape[0]
        sg, qg = torch.empty_like(s), torch.empty_like(q)
        with torch.cuda.device(s.device):
            rc = lib.gf_gaussian_prepare_backward(P, 0 if ctx.packed else 1, _lib.ptr(s), _lib.ptr(q), _lib.ptr(g),
                                                  _lib.ptr(sg), _lib.ptr(qg), _lib.current_stream(s.device))
        _lib.check(rc, "gf_gaussian_prepare_backward")
        return sg, qg, None


class _GaussianPrepare(torch.autograd.Function):
    """``(means3D_int, radii, cov6)`` in one launch; only ``cov6`` carries a gradient (to
    ``scales`` and ``rotations``) -- the reference detaches means and scales on the integer
    path (local_aggregate/__init__.py:133,139)."""

    @staticmethod
    def forward(ctx, means3D, scales, rotations, pc_min, grid_size, scale_multiplier, H, W, D, radii_mode, radii_min,
                status):
        means_int, radii, cov6 = gaussian_prepare(means3D, scales, rotations, pc_min, grid_size, scale_multiplier,
                                                  H, W, D, radii_mode, radii_min, status=status)
        ctx.save_for_backward(_c(scales), _c(rotations))
        ctx.mark_non_differentiable(means_int, radii)
        return means_int, radii, cov6

    @staticmethod
    def backward(ctx, _gi, _gr, cov_grad):
        s, q = ctx.saved_tensors
        lib = _lib.load()
        g = _c(cov_grad)
        sg, qg = torch.empty_like(s), torch.empty_like(q)
        with torch.cuda.device(s.device):
            rc = lib.gf_gaussian_prepare_backward(s.shape[0], 0, _lib.ptr(s), _lib.ptr(q), _lib.ptr(g), _lib.ptr(sg),
                                                  _lib.ptr(qg), _lib.current_stream(s.device))
        _lib.check(rc, "gf_gaussian_prepare_backward")
        return (None, sg, qg) + (None,) * 9


def covariance_inverse(scales, rotations, packed=False):
    """Differentiable Sigma^-1 for ``[..., 3]`` scales and ``[..., 4]`` (w,x,y,z) rotations:
    ``[..., 3, 3]`` or, packed, ``[..., 6]`` in the rasteriser's (xx,yy,zz,xy,yz,xz) order."""
    lead = scales.shape[:-1]
    cov = _CovInverse.apply(scales.reshape(-1, 3), rotations.reshape(-1, 4), packed)
    return cov.reshape(*lead, 6) if packed else cov.reshape(*lead, 3, 3)


class _GaussianPack(torch.autograd.Function):
    """``gf_gaussian_pack``: zero column / appended empty Gaussian / softmax of ``prepare_gaussian_args``
    (gaussian_head.py:88-109) in one launch for batch size 1; the gradients are slices (and the softmax rule)."""

    @staticmethod
    def forward(ctx, means, scales, rotations, sem, opa, empty_scalar, empty_host, cout, zero_first, with_empty, softmax,
                empty_label):
        _lib.require_gpu(means, scales, rotations, sem, opa, empty_scalar)
        lib = _lib.load()
        import ctypes
        m, s_, q, se = _c(means[0]), _c(scales[0]), _c(rotations[0]), _c(sem[0])
        o = None if opa is None else _c(opa[0].reshape(-1))
        P, cin = m.shape[0], se.shape[1]
        Pout = P + (1 if with_empty else 0)
        dev = m.device
        outs = [torch.empty(Pout, k, dtype=f32, device=dev) for k in (3, 3, 4, cout)] + [torch.empty(Pout, dtype=f32, device=dev)]
        host = [None, None, None]
        if with_empty:
            host = [(ctypes.c_float * len(v))(*v) for v in empty_host]
        es = None if empty_scalar is None else _c(empty_scalar)
        with torch.cuda.device(dev):
            rc = lib.gf_gaussian_pack(P, cin, cout, int(zero_first), int(with_empty), int(softmax), int(empty_label),
                                      _lib.ptr(m), _lib.ptr(s_), _lib.ptr(q), _lib.ptr(se), _lib.ptr(o),
                                      *[None if h is None else ctypes.cast(h, ctypes.c_void_p) for h in host], _lib.ptr(es),
                                      *[_lib.ptr(t) for t in outs], _lib.current_stream(dev))
        _lib.check(rc, "gf_gaussian_pack")
        ctx.meta = (P, cin, cout, zero_first, with_empty, softmax, empty_label, opa is not None and opa.requires_grad,
                    None if opa is None else opa.shape)
        if softmax:
            ctx.save_for_backward(outs[3])
        res = tuple(t[None] for t in outs[:4]) + (outs[4][None, :, None],)
        if opa is None:
            ctx.mark_non_differentiable(res[4])   # all ones, as the reference's torch.ones_like
        return res

    @staticmethod
    def backward(ctx, gm, gs, gq, gsem, gopa):
        P, cin, cout, zero_first, with_empty, softmax, empty_label, opa_grad, opa_shape = ctx.meta
        shift = 1 if (cout > cin and zero_first) else 0
        g_sem_in = gsem[:, :P, shift:shift + cin]
        if softmax:
            (y_full,) = ctx.saved_tensors
            y = y_full[None, :P, shift:shift + cin]
            g_sem_in = y * (g_sem_in - (y * g_sem_in).sum(dim=-1, keepdim=True))
        g_es = gsem[0, P, empty_label].reshape(1) if with_empty else None
        g_opa = gopa[:, :P].reshape(opa_shape) if opa_grad else None
        return gm[:, :P], gs[:, :P], gq[:, :P], g_sem_in, g_opa, g_es, None, None, None, None, None, None


class GaussianArgs(nn.Module):
    """``GaussianHead.prepare_gaussian_args`` as a stand-alone module: same constructor keys,
    buffers and parameter (``empty_scalar``, ``empty_mean`` ... model/head/gaussian_head.py:41-51)
    and the same return tuple ``(means, origi_opa, opacities, scales, CovInv)``."""

    def __init__(self, num_classes=18, empty_args=None, with_empty=False, dataset_type='nusc', empty_label=17,
                 use_localaggprob=False):
        super().__init__()
        self.num_classes = num_classes
        self.use_localaggprob = use_localaggprob
        if with_empty:
            self.empty_scalar = nn.Parameter(torch.ones(1, dtype=torch.float) * 10.0)
            self.register_buffer('empty_mean', torch.tensor(empty_args['mean'])[None, None, :])
            self.register_buffer('empty_scale', torch.tensor(empty_args['scale'])[None, None, :])
            self.register_buffer('empty_rot', torch.tensor([1., 0., 0., 0.])[None, None, :])
            self.register_buffer('empty_sem', torch.zeros(self.num_classes)[None, None, :])
            self.register_buffer('empty_opa', torch.ones(1)[None, None, :])
        self._empty_host = None
        self.with_emtpy = with_empty  # (sic) attribute name of the reference
        self.dataset_type = dataset_type
        self.empty_label = empty_label

    def forward(self, means, scales, rotations, semantics, opacities=None):
        """Tensors are ``[b,g,3] [b,g,3] [b,g,4] [b,g,c] [b,g,1]|empty`` (the fields of the
        reference's ``gaussians`` namedtuple, gaussian_head.py:83-87)."""
        sem = semantics
        origi_opa = opacities
        kitti = 'kitti' in self.dataset_type
        if means.is_cuda and means.shape[0] == 1 and (self.with_emtpy or self.use_localaggprob):
            # one launch (gf_gaussian_pack) instead of the reference's zeros_like + six torch.cat (or softmax + cat)
            assert sem.shape[-1] == self.num_classes - 1
            opa_in = None if (origi_opa is None or origi_opa.numel() == 0) else origi_opa
            empty_host = None
            if self.with_emtpy:
                # the buffers are constants of the head: read once -- and again when one of them has been assigned or written in
                # place since (load_state_dict, .copy_(), .to(): storage pointer / version counter), ADVICE r4
                key = tuple((t.data_ptr(), t._version) for t in (self.empty_mean, self.empty_scale, self.empty_rot))
                if self._empty_host is None or self._empty_host[0] != key:
                    self._empty_host = (key, [self.empty_mean.flatten().tolist(), self.empty_scale.flatten().tolist(),
                                              self.empty_rot.flatten().tolist()])
                empty_host = self._empty_host[1]
            means, scales, rotations, sem, origi_opa = _GaussianPack.apply(
                means, scales, rotations, sem, opa_in, self.empty_scalar if self.with_emtpy else None, empty_host,
                self.num_classes, kitti, self.with_emtpy, not self.with_emtpy, self.empty_label)
            return means, origi_opa, sem, scales, covariance_inverse(scales, rotations)
        return self._forward_torch(means, scales, rotations, sem, origi_opa)

    def _forward_torch(self, means, scales, rotations, sem, origi_opa):
        """The reference's op sequence (gaussian_head.py:88-119 with the device-side Sigma^-1): batch sizes above 1, CPU
        tensors of the configuration tests, and the checker of the fused path (tests/test_prepare.py)."""
        kitti = 'kitti' in self.dataset_type
        if origi_opa is None or origi_opa.numel() == 0:
            origi_opa = torch.ones_like(sem[..., :1], requires_grad=False)
        if self.with_emtpy:
            assert sem.shape[-1] == self.num_classes - 1
            zero = torch.zeros_like(sem[..., :1])
            sem = torch.cat([zero, sem] if kitti else [sem, zero], dim=-1)
            b = means.shape[0]
            means = torch.cat([means, self.empty_mean.expand(b, -1, -1)], dim=1)
            scales = torch.cat([scales, self.empty_scale.expand(b, -1, -1)], dim=1)
            rotations = torch.cat([rotations, self.empty_rot.expand(b, -1, -1)], dim=1)
            empty_sem = self.empty_sem.clone()
            empty_sem[..., self.empty_label] += self.empty_scalar
            sem = torch.cat([sem, empty_sem.expand(b, -1, -1)], dim=1)
            origi_opa = torch.cat([origi_opa, self.empty_opa.expand(b, -1, -1)], dim=1)
        elif self.use_localaggprob:
            assert sem.shape[-1] == self.num_classes - 1
            sem = sem.softmax(dim=-1)
            zero = torch.zeros_like(sem[..., :1])
            sem = torch.cat([zero, sem] if kitti else [sem, zero], dim=-1)
        cov_inv = covariance_inverse(scales, rotations)  # b, g, 3, 3
        return means, origi_opa, sem, scales, cov_inv
