"""Submanifold sparse 3-D convolution over the Gaussian centres.

Host mirror of ``SparseConv3D`` (model/encoder/gaussian_encoder/spconv3d_module.py:10-83), whose
``spconv.SubMConv3d`` has no ROCm build: the rulebook and the gather-GEMM run in
``gf_subm_rulebook_* / gf_subm_conv_apply / gf_subm_conv_weight_grad`` (include/gf_hip.h).
"""
import math

import ctypes
import os

import torch
import torch.nn as nn
from torch.autograd.function import Function, once_differentiable

from . import _lib

f32, i32 = torch.float32, torch.int32


class Rulebook:
    """Neighbour pairs of one point set (reusable across layers and by the backward pass).

    ``pair_capacity=None``: the pair count is read back once to size the pair arrays (spconv reads its pair counts
    the same way).  ``pair_capacity=n``: nothing is read back -- the arrays hold ``n`` pairs, a point set with more
    pairs produces an empty rulebook (NaN output: a refused point set must not look like a healthy zero) and a
    refusal flag that :meth:`check` raises on; call it at a point where the stream is synchronised anyway, or use
    :meth:`poll` (non-blocking: the status is copied to pinned host memory behind the build and read once that copy
    has completed).  Note the memory side of a generous capacity: ``apply`` sizes its partial-row buffer
    ``[pair_capacity, Cout]`` by it (32 KB per point at ``pairs_per_point=64``, Cout = 128)."""

    def __init__(self, indices, batch_size, spatial_shape, kernel_size, pair_capacity=None, out_range=None):
        """``out_range=(lo, hi)``: pairs only for the output points ``[lo, hi)`` (every point stays a neighbour) -- the rows an
        anchor-sharded rank owns, at ``(hi - lo) / N`` of the gather-GEMM work; the other rows of ``apply`` are zeros."""
        _lib.require_gpu(indices)
        lib = _lib.load()
        self.indices = indices.detach().to(i32).contiguous()
        self.N = self.indices.shape[0]
        self.out_range = None if out_range is None else (int(out_range[0]), int(out_range[1]))
        rng = () if self.out_range is None else self.out_range
        count_fn = lib.gf_subm_rulebook_count if self.out_range is None else lib.gf_subm_rulebook_count_range
        fill_fn = lib.gf_subm_rulebook_fill if self.out_range is None else lib.gf_subm_rulebook_fill_range
        build_fn = lib.gf_subm_rulebook_build if self.out_range is None else lib.gf_subm_rulebook_build_range
        self.dims = (self.N, int(batch_size), int(spatial_shape[0]), int(spatial_shape[1]), int(spatial_shape[2]),
                     int(kernel_size))
        dev = self.indices.device
        nbytes = lib.gf_subm_tables_bytes(*self.dims)
        if nbytes == 0:
            raise RuntimeError(f"unsupported sparse-conv geometry {self.dims}")
        self.tables = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        self._status = self.tables[nbytes - 256:nbytes - 240].view(torch.int64)   # (pair count, refusal bits)
        self.checked = pair_capacity is None
        with torch.cuda.device(dev):
            if pair_capacity is not None:
                self.total = max(int(pair_capacity), 1)
                self.pair_in = torch.empty(self.total, dtype=i32, device=dev)
                self.pair_out = torch.empty(self.total, dtype=i32, device=dev)
                rc = build_fn(*self.dims, *rng, _lib.ptr(self.indices), _lib.ptr(self.tables), nbytes,
                                                _lib.ptr(self.pair_in), _lib.ptr(self.pair_out), self.total, _lib.current_stream(dev))
                _lib.check(rc, "gf_subm_rulebook_build")
                # deferred check: status -> pinned host memory on the same stream, event behind the copy
                # (not while a HIP graph is being captured: no host allocation there; check() still works after a replay)
                self._status_event = None
                if not torch.cuda.is_current_stream_capturing():
                    self._host_status = torch.empty(2, dtype=torch.int64, pin_memory=True)
                    self._host_status.copy_(self._status, non_blocking=True)
                    self._status_event = torch.cuda.Event()
                    self._status_event.record(torch.cuda.current_stream(dev))
                return
            rc = count_fn(*self.dims, *rng, _lib.ptr(self.indices), _lib.ptr(self.tables), nbytes, _lib.current_stream(dev))
            _lib.check(rc, "gf_subm_rulebook_count")
            total = self.check()   # the one host read
            self.total = int(total)
            self.pair_in = torch.empty(max(self.total, 1), dtype=i32, device=dev)
            self.pair_out = torch.empty(max(self.total, 1), dtype=i32, device=dev)
            rc = fill_fn(*self.dims, *rng, _lib.ptr(self.indices), _lib.ptr(self.tables),
                         _lib.ptr(self.pair_in), _lib.ptr(self.pair_out), _lib.current_stream(dev))
            _lib.check(rc, "gf_subm_rulebook_fill")

    def representative_mask(self):
        """``[N,1]`` fp32, 1 for the point with the LARGEST index of its cell (and for points outside the grid), 0 for the
        others: the ``duplicates="last"`` mode of :func:`subm_conv3d` (cached)."""
        if getattr(self, "_rep_mask", None) is None:
            N, B, X, Y, Z, _ = self.dims
            idx = self.indices.long()
            b, x, y, z = idx.unbind(1)
            valid = (b >= 0) & (b < B) & (x >= 0) & (x < X) & (y >= 0) & (y < Y) & (z >= 0) & (z < Z)
            key = (((b * X + x) * Y + y) * Z + z).clamp(min=0, max=B * X * Y * Z - 1)
            me = torch.arange(N, device=idx.device)
            last = torch.full((B * X * Y * Z,), -1, dtype=torch.long, device=idx.device)
            last.scatter_reduce_(0, key[valid], me[valid], reduce="amax")
            self._rep_mask = ((last[key] == me) | ~valid).to(f32)[:, None]
        return self._rep_mask

    def check(self):
        """Reads the device status (synchronises): returns the pair count, raises if the point set was refused."""
        total, refused = self._status.tolist()
        self.checked = True
        self.pair_count = int(total)
        if refused:
            self._raise(total, refused)
        return total

    def poll(self):
        """Non-blocking :meth:`check` of a ``pair_capacity`` rulebook: ``None`` while the status copy is still in
        flight, else the pair count (the device's count, the same number :meth:`check` returns -- not the capacity the
        arrays were sized with) -- raising like :meth:`check` if the point set was refused."""
        if self.checked:
            return self.pair_count
        if self._status_event is None or not self._status_event.query():
            return None
        total, refused = self._host_status.tolist()
        self.checked = True
        self.pair_count = int(total)
        if refused:
            self._raise(total, refused)
        return total

    def _raise(self, total, refused):
        raise RuntimeError("sparse-conv rulebook refused: " +
                           ("a cell holds more than 65535 points" if refused & 1 else
                            "more than 2^31 - 1 neighbour pairs" if refused & 2 else
                            f"{total} neighbour pairs exceed the pair_capacity of {self.total}"))

    def apply(self, features, weight):
        """``out[N, Cout]`` for ``features [N, Cin]`` and ``weight [K^3, Cin, Cout]`` (no autograd)."""
        lib = _lib.load()
        features, weight = features.detach().to(f32).contiguous(), weight.detach().to(f32).contiguous()
        cin, cout = weight.shape[1], weight.shape[2]
        dev = features.device
        out = torch.empty(self.N, cout, dtype=f32, device=dev)
        partial = torch.empty(max(self.total, 1), cout, dtype=f32, device=dev)
        # (the rows as two f16 terms under a power-of-two scale per row, split once per call: gf_subm_conv_apply_scratch)
        nscratch = int(lib.gf_subm_apply_scratch_bytes(self.N, cin))
        scratch = torch.empty(max(nscratch, 16), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            rc = lib.gf_subm_conv_apply_scratch(*self.dims, cin, cout, self.total, _lib.ptr(features), _lib.ptr(weight), _lib.ptr(self.tables),
                                                _lib.ptr(self.pair_in), _lib.ptr(partial), _lib.ptr(out), _lib.ptr(scratch), nscratch,
                                                _lib.current_stream(dev))
        _lib.check(rc, "gf_subm_conv_apply_scratch")
        return out

    def weight_grad(self, features, grad_out):
        lib = _lib.load()
        features, grad_out = features.detach().to(f32).contiguous(), grad_out.detach().to(f32).contiguous()
        cin, cout = features.shape[1], grad_out.shape[1]
        k3 = self.dims[5] ** 3
        gw = torch.empty(k3, cin, cout, dtype=f32, device=features.device)
        with torch.cuda.device(features.device):
            rc = lib.gf_subm_conv_weight_grad(*self.dims, cin, cout, self.total, _lib.ptr(features), _lib.ptr(grad_out),
                                              _lib.ptr(self.tables), _lib.ptr(self.pair_in), _lib.ptr(self.pair_out),
                                              _lib.ptr(gw), _lib.current_stream(features.device))
        _lib.check(rc, "gf_subm_conv_weight_grad")
        return gw


class _SubMConv(Function):
    @staticmethod
    def forward(ctx, features, weight, rulebook):
        ctx.rulebook = rulebook
        ctx.save_for_backward(features, weight)
        return rulebook.apply(features, weight)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        features, weight = ctx.saved_tensors
        rb = ctx.rulebook
        g_feat = g_w = None
        if ctx.needs_input_grad[0]:
            # the neighbour relation is symmetric: same rulebook, kernel mirrored and transposed
            g_feat = rb.apply(grad_out, weight.flip(0).transpose(1, 2))
        if ctx.needs_input_grad[1]:
            g_w = rb.weight_grad(features, grad_out)
        return g_feat, g_w, None


DUPLICATES = ("sum", "last")


def subm_conv3d(features, indices, weight, batch_size, spatial_shape, kernel_size, rulebook=None, duplicates="sum"):
    """Functional submanifold convolution; ``indices`` int ``[N,4]`` = (batch, x, y, z),
    ``weight [K^3, Cin, Cout]`` (offsets in [K,K,K] order).  Returns ``[N, Cout]``.

    ``duplicates`` says what several points in ONE cell mean.  ``"sum"`` (default): all of them are sources and each
    receives the cell's output -- the dense convolution of the scattered-and-summed features.  ``"last"``: only the point
    with the largest index of a cell is a source for its neighbours (every point still receives an output), which is one
    of the outcomes of spconv's hash insert, where the duplicate that ends up in the table is a race
    (spconv3d_module.py:73-77 hands the duplicates to ``spconv.SparseConvTensor`` as they are); the others get no
    gradient.  It is the same operator on features whose non-representative rows are zeroed."""
    if duplicates not in DUPLICATES:
        raise ValueError(f"duplicates must be one of {DUPLICATES}")
    rb = rulebook if rulebook is not None else Rulebook(indices, batch_size, spatial_shape, kernel_size)
    if rb.out_range is not None and torch.is_grad_enabled() and (features.requires_grad or weight.requires_grad):
        # (the backward relies on the neighbour relation being symmetric: a rulebook restricted to a range of output points is not)
        raise RuntimeError("a Rulebook with out_range is inference-only")
    if duplicates == "last":
        features = features * rb.representative_mask()
    return _SubMConv.apply(features, weight, rb)


class SubMConv3d(nn.Module):
    """``spconv.SubMConv3d(in_channels, out_channels, kernel_size, stride=1, padding=k//2, bias=...)``
    as used by the reference (spconv3d_module.py:28-44); weight ``[K^3, Cin, Cout]``."""

    def __init__(self, in_channels, out_channels, kernel_size=5, bias=True, duplicates="sum"):
        super().__init__()
        if duplicates not in DUPLICATES:
            raise ValueError(f"duplicates must be one of {DUPLICATES}")
        self.duplicates = duplicates
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        self.weight = nn.Parameter(torch.empty(kernel_size ** 3, in_channels, out_channels))
        nn.init.kaiming_uniform_(self.weight.view(-1, out_channels), a=math.sqrt(5))
        if bias:
            bound = 1.0 / math.sqrt(in_channels * kernel_size ** 3)
            self.bias = nn.Parameter(torch.empty(out_channels).uniform_(-bound, bound))
        else:
            self.register_parameter("bias", None)

    def forward(self, features, indices, batch_size, spatial_shape, rulebook=None):
        out = subm_conv3d(features, indices, self.weight, batch_size, spatial_shape, self.kernel_size, rulebook,
                          duplicates=self.duplicates)
        return out if self.bias is None else out + self.bias


class SparseConv3D(nn.Module):
    """Drop-in for the reference's ``SparseConv3D`` block (spconv3d_module.py:10-83): voxelise the
    anchor centres, submanifold conv (one bias-free layer, or with ``use_multi_layer`` three
    conv + LayerNorm + ReLU stages, :26-37), output projection.  The three stages share one
    rulebook -- a submanifold convolution keeps the active set."""

    def __init__(self, in_channels, embed_channels, pc_range, grid_size, xyz_activation="sigmoid", use_out_proj=False,
                 kernel_size=5, use_multi_layer=False, pairs_per_point=None, duplicates="sum", **kwargs):
        super().__init__()
        # extra keyword ``duplicates`` ("sum" | "last"): what several anchors in one cell mean, see subm_conv3d
        # extra keyword: with ``pairs_per_point=n`` the rulebook is built for at most n * points neighbour pairs without
        # reading the pair count back (no host synchronisation in the encoder loop); ``self.last_rulebook.check()``
        # reports a point set that did not fit.  None = exact size, one host read per rulebook (like spconv).
        self.pairs_per_point = pairs_per_point
        self.last_rulebook = None
        self._unchecked = []   # pair_capacity rulebooks whose status has not been seen yet (polled, never waited for)
        if use_multi_layer:
            self.layer = nn.ModuleList()
            for i in range(3):
                self.layer.append(SubMConv3d(in_channels if i == 0 else embed_channels, embed_channels, kernel_size,
                                             duplicates=duplicates))
                self.layer.append(nn.LayerNorm(embed_channels))
                self.layer.append(nn.ReLU(True))
        else:
            self.layer = SubMConv3d(in_channels, embed_channels, kernel_size, bias=False, duplicates=duplicates)
        self.kernel_size = kernel_size
        self.output_proj = nn.Linear(embed_channels, embed_channels) if use_out_proj else nn.Identity()
        self.use_sigmoid = xyz_activation == "sigmoid"
        self._range = [float(v) for v in pc_range]
        self.register_buffer('pc_range', torch.tensor(pc_range, dtype=torch.float))
        self.register_buffer('grid_size', torch.tensor(grid_size, dtype=torch.float))
        # spatial_shape = ((pc_range[3:] - pc_range[:3]) / grid_size).to(int32), evaluated once in fp32 like the reference (:70-71)
        self._spatial = ((self.pc_range[3:] - self.pc_range[:3]) / self.grid_size).to(torch.int32).tolist()

    def voxel_indices(self, anchor):
        """int32 ``[b*g, 4]`` (batch, x, y, z) of the anchor centres (spconv3d_module.py:56-66,
        ``cartesian`` model/encoder/gaussian_encoder/utils.py:26-36, ``safe_sigmoid`` model/utils/safe_ops.py:7-9)."""
        bs, g = anchor.shape[:2]
        # (integer indices: nothing to differentiate either way)
        if anchor.is_cuda and anchor.dtype == torch.float32 and anchor.shape[-1] >= 3 and os.environ.get("GF_SUBM_TORCH_VOXELIZE") is None:
            return self._voxel_indices_native(anchor)
        return self._voxel_indices_torch(anchor)

    def _voxel_indices_native(self, anchor):
        """One launch (``gf_subm_voxelize``) instead of the dozen elementwise ops of ``_voxel_indices_torch``: the same fp32
        operations in the same order, so the same indices (``tests/test_subm_conv.py``)."""
        import numpy as np
        bs, g = anchor.shape[:2]
        a2 = anchor.detach().reshape(bs * g, anchor.shape[-1])
        if not a2.is_contiguous():
            a2 = a2.contiguous()
        key = (self.pc_range.data_ptr(), self.pc_range._version, self.grid_size.data_ptr(), self.grid_size._version)
        host = getattr(self, "_voxel_host", None)
        if host is None or host[0] != key:   # (buffers can be reloaded: one host read per change)
            r = self._range
            pc, gs = self.pc_range.detach().cpu().numpy().astype(np.float32), self.grid_size.detach().cpu().numpy().astype(np.float32)
            arr = lambda v: (ctypes.c_float * 3)(*[float(np.float32(x)) for x in v])
            host = self._voxel_host = (key, arr([r[3 + a] - r[a] for a in range(3)]), arr([r[a] for a in range(3)]), arr(pc[:3]), arr(gs[:3]))
        out = torch.empty((bs * g, 4), dtype=torch.int32, device=anchor.device)
        lib = _lib.load()
        with torch.cuda.device(anchor.device):
            rc = lib.gf_subm_voxelize(bs * g, g, a2.shape[1], int(self.use_sigmoid), _lib.ptr(a2), host[1], host[2], host[3], host[4],
                                      _lib.ptr(out), _lib.current_stream(anchor.device))
        _lib.check(rc, "gf_subm_voxelize")
        return out

    def _voxel_indices_torch(self, anchor):
        bs, g = anchor.shape[:2]
        xyz = anchor[..., :3]
        xyz = xyz.clamp(-9.21, 9.21).sigmoid() if self.use_sigmoid else xyz.clamp(min=1e-6, max=1 - 1e-6)
        r = self._range
        xyz = torch.stack([xyz[..., a] * (r[3 + a] - r[a]) + r[a] for a in range(3)], dim=-1).flatten(0, 1)
        idx = ((xyz - self.pc_range[None, :3]) / self.grid_size[None, :]).to(torch.int32)
        bidx = torch.arange(bs, device=idx.device, dtype=torch.int32).repeat_interleave(g)[:, None]
        return torch.cat([bidx, idx], dim=-1)

    def forward(self, instance_feature, anchor, out_range=None):
        """``out_range=(lo, hi)`` (extra keyword, inference, batch size 1, single-layer blocks): ``instance_feature`` / ``anchor``
        are the WHOLE (all-gathered) anchor set, the block is evaluated for the anchors ``[lo, hi)`` only and returns
        ``[1, hi - lo, C]`` -- the rows an anchor-sharded rank owns, bit-identical to the same rows of the whole block, at
        ``(hi - lo) / g`` of the gather-GEMM and projection work."""
        bs, g, _ = instance_feature.shape
        indices = self.voxel_indices(anchor)
        feats = instance_feature.flatten(0, 1)
        # refusals of earlier calls surface here, without blocking: a refused rulebook already produced NaN features,
        # this names the cause as soon as its status has reached the host.  (Both paths below: the list keeps at most the
        # eight youngest rulebooks whose status is still in flight, so a long inference loop holds a bounded set of tables.)
        if self._unchecked and not torch.cuda.is_current_stream_capturing():
            still = []
            for old in self._unchecked:
                if old.poll() is None:
                    still.append(old)
            self._unchecked = still[-8:]
        if out_range is not None:
            if bs != 1 or not isinstance(self.layer, SubMConv3d):
                raise RuntimeError("out_range needs batch size 1 and a single-layer block")
            lo, hi = int(out_range[0]), int(out_range[1])
            cap = None if self.pairs_per_point is None else int(self.pairs_per_point) * max(hi - lo, 1)
            rb = self.last_rulebook = Rulebook(indices, bs, self._spatial, self.kernel_size, pair_capacity=cap, out_range=(lo, hi))
            if cap is not None and not rb.checked and rb._status_event is not None:
                self._unchecked.append(rb)
            out = self.layer(feats, indices, bs, self._spatial, rulebook=rb)
            return self.output_proj(out[lo:hi])[None]
        cap = None if self.pairs_per_point is None else int(self.pairs_per_point) * indices.shape[0]
        rb = self.last_rulebook = Rulebook(indices, bs, self._spatial, self.kernel_size, pair_capacity=cap)
        if cap is not None and not rb.checked and rb._status_event is not None:
            self._unchecked.append(rb)
        if isinstance(self.layer, SubMConv3d):
            out = self.layer(feats, indices, bs, self._spatial, rulebook=rb)
        else:
            out = feats
            for m in self.layer:
                out = m(out, indices, bs, self._spatial, rulebook=rb) if isinstance(m, SubMConv3d) else m(out)
        return self.output_proj(out.unflatten(0, (bs, g)))
