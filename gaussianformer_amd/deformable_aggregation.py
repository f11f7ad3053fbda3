"""Host-side mirror of ``model/encoder/gaussian_encoder/ops/deformable_aggregation.py``.

``DeformableAggregationFunction`` keeps the reference's interface (apply signature,
dtype coercions, ``once_differentiable`` backward that accumulates into three zeroed
buffers, ``feature_maps_format``) and calls the HIP kernels through the C ABI.
"""
import torch
from torch.autograd.function import Function, once_differentiable

from . import _lib


def deformable_aggregation_forward(mc_ms_feat, spatial_shape, scale_start_index, sampling_location, weights,
                                   pin_channel_groups=False):
    """Counterpart of ``deformable_aggregation_ext.deformable_aggregation_forward``
    (ops/src/deformable_aggregation.cpp:41-71): dims are read from the tensor sizes.  ``pin_channel_groups`` (extra
    keyword) takes ``gf_daf_forward_pinned``: channel groups pinned to XCDs, bit-identical output, faster when every
    point is seen by several cameras at unrelated places, slower on projected geometry (include/gf_hip.h)."""
    lib = _lib.load()
    _lib.require_gpu(mc_ms_feat, spatial_shape, scale_start_index, sampling_location, weights)
    B, cams, num_feat, C = mc_ms_feat.shape
    L, pts, G = spatial_shape.shape[0], sampling_location.shape[1], weights.shape[4]
    out = torch.empty((B, pts, C), dtype=torch.float32, device=mc_ms_feat.device)
    with torch.cuda.device(mc_ms_feat.device):
        fn = lib.gf_daf_forward_pinned if pin_channel_groups else lib.gf_daf_forward
        rc = fn(B, cams, num_feat, C, L, pts, G, _lib.ptr(mc_ms_feat), _lib.ptr(spatial_shape),
                _lib.ptr(scale_start_index), _lib.ptr(sampling_location), _lib.ptr(weights),
                _lib.ptr(out), _lib.current_stream(mc_ms_feat.device))
    _lib.check(rc, "gf_daf_forward")
    return out


_workspaces = {}


def _workspace(device, nbytes):
    """Grow-only scratch for the sorted backward (tap ids + counters), one per (device, stream): the kernels
    run on torch's current stream and two streams must not share it."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _workspaces.pop(key, None)
    while len(_workspaces) >= 8:             # bounded: least recently used entries go (a process that keeps creating streams)
        _workspaces.pop(next(iter(_workspaces)))
    if ws is not None:
        _workspaces[key] = ws
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def deformable_aggregation_backward(mc_ms_feat, spatial_shape, scale_start_index, sampling_location, weights,
                                    grad_output, grad_mc_ms_feat, grad_sampling_location, grad_weights,
                                    pixel_major=True):
    """Counterpart of ``deformable_aggregation_ext.deformable_aggregation_backward``
    (ops/src/deformable_aggregation.cpp:73-110): accumulates in place into the three
    caller-zeroed gradient tensors.  ``pixel_major`` selects ``gf_daf_backward_sorted`` (no
    per-channel atomic scatter) whenever the shape supports it; ``False`` forces the
    reference's scatter formulation (``gf_daf_backward``)."""
    lib = _lib.load()
    _lib.require_gpu(mc_ms_feat, grad_output, grad_mc_ms_feat, grad_sampling_location, grad_weights)
    B, cams, num_feat, C = mc_ms_feat.shape
    L, pts, G = spatial_shape.shape[0], sampling_location.shape[1], weights.shape[4]
    nbytes = lib.gf_daf_backward_workspace_bytes(B, cams, num_feat, C, L, pts, G) if pixel_major else 0
    if nbytes:
        with torch.cuda.device(mc_ms_feat.device):
            ws = _workspace(mc_ms_feat.device, nbytes)
            rc = lib.gf_daf_backward_sorted(B, cams, num_feat, C, L, pts, G, _lib.ptr(mc_ms_feat),
                                            _lib.ptr(spatial_shape), _lib.ptr(scale_start_index),
                                            _lib.ptr(sampling_location), _lib.ptr(weights), _lib.ptr(grad_output),
                                            _lib.ptr(grad_mc_ms_feat), _lib.ptr(grad_sampling_location),
                                            _lib.ptr(grad_weights), _lib.ptr(ws), nbytes,
                                            _lib.current_stream(mc_ms_feat.device))
        _lib.check(rc, "gf_daf_backward_sorted")
        return
    with torch.cuda.device(mc_ms_feat.device):
        rc = lib.gf_daf_backward(B, cams, num_feat, C, L, pts, G, _lib.ptr(mc_ms_feat), _lib.ptr(spatial_shape),
                                 _lib.ptr(scale_start_index), _lib.ptr(sampling_location), _lib.ptr(weights),
                                 _lib.ptr(grad_output), _lib.ptr(grad_mc_ms_feat),
                                 _lib.ptr(grad_sampling_location), _lib.ptr(grad_weights),
                                 _lib.current_stream(mc_ms_feat.device))
    _lib.check(rc, "gf_daf_backward")


_shape_tensors = {}   # (device, pyramid shapes) -> (spatial_shape, scale_start_index) int64 tensors


def _format_call(levels, table, inverse):
    """gf_feature_maps_format on contiguous fp32 CUDA tensors: levels [bs,cams,C,h,w], table [bs,cams,num_feat,C]."""
    import ctypes
    lib = _lib.load()
    bs, cams, C = levels[0].shape[:3]
    L = len(levels)
    hw = (ctypes.c_int * L)(*[int(f.shape[3] * f.shape[4]) for f in levels])
    ptrs = (ctypes.c_void_p * L)(*[f.data_ptr() for f in levels])
    with torch.cuda.device(table.device):
        rc = lib.gf_feature_maps_format(bs * cams, C, L, ctypes.cast(hw, ctypes.c_void_p), ctypes.cast(ptrs, ctypes.c_void_p),
                                        _lib.ptr(table), int(inverse), _lib.current_stream(table.device))
    _lib.check(rc, "gf_feature_maps_format")


class _FeatureMapsFormat(Function):
    """levels -> channels-last table in one tiled-transpose launch; backward = the inverse launch."""

    @staticmethod
    def forward(ctx, *feature_maps):
        levels = [f.contiguous().float() for f in feature_maps]
        bs, cams, C = levels[0].shape[:3]
        ctx.shapes = [tuple(f.shape) for f in levels]
        num_feat = sum(s[3] * s[4] for s in ctx.shapes)
        table = torch.empty(bs, cams, num_feat, C, dtype=torch.float32, device=levels[0].device)
        _format_call(levels, table, inverse=False)
        return table

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_table):
        grads = [torch.empty(s, dtype=torch.float32, device=grad_table.device) for s in ctx.shapes]
        _format_call(grads, grad_table.contiguous().float(), inverse=True)
        return tuple(grads)


class DeformableAggregationFunction(Function):
    """ops/deformable_aggregation.py:7-117."""

    @staticmethod
    def forward(ctx, mc_ms_feat, spatial_shape, scale_start_index, sampling_location, weights):
        # output: [bs, num_pts, num_embeds]
        mc_ms_feat = mc_ms_feat.contiguous().float()
        spatial_shape = spatial_shape.contiguous().int()
        scale_start_index = scale_start_index.contiguous().int()
        sampling_location = sampling_location.contiguous().float()
        weights = weights.contiguous().float()
        output = deformable_aggregation_forward(mc_ms_feat, spatial_shape, scale_start_index,
                                                sampling_location, weights)
        ctx.save_for_backward(mc_ms_feat, spatial_shape, scale_start_index, sampling_location, weights)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        mc_ms_feat, spatial_shape, scale_start_index, sampling_location, weights = ctx.saved_tensors
        grad_mc_ms_feat = torch.zeros_like(mc_ms_feat)
        grad_sampling_location = torch.zeros_like(sampling_location)
        grad_weights = torch.zeros_like(weights)
        deformable_aggregation_backward(mc_ms_feat, spatial_shape, scale_start_index, sampling_location, weights,
                                        grad_output.contiguous().float(), grad_mc_ms_feat, grad_sampling_location,
                                        grad_weights)
        return grad_mc_ms_feat, None, None, grad_sampling_location, grad_weights

    @staticmethod
    def feature_maps_format(feature_maps, inverse=False):
        """List of ``[bs, cams, C, h, w]`` maps <-> ``(col_feats [bs,cams,sum(hw),C],
        spatial_shape [L,2] int64, scale_start_index [L] int64)`` -- ops/deformable_aggregation.py:77-117."""
        if not inverse:
            bs, num_cams = feature_maps[0].shape[:2]
            shapes = [tuple(f.shape[-2:]) for f in feature_maps]
            starts, run = [], 0
            for h, w in shapes:
                starts.append(run)
                run += h * w
            if feature_maps[0].is_cuda and len(feature_maps) <= 8 and all(f.dtype == torch.float32 for f in feature_maps):
                # one tiled-transpose launch (gf_feature_maps_format), already contiguous for DAF.apply
                col_feats = _FeatureMapsFormat.apply(*feature_maps)
            else:
                flat = [f.reshape(bs, num_cams, f.shape[2], -1) for f in feature_maps]
                col_feats = torch.cat(flat, dim=-1).permute(0, 1, 3, 2)
            # the two index tensors depend on the pyramid's shape only: built once per (device, shapes) -- two
            # host-to-device copies less per call, and nothing left in the call that a HIP-graph capture refuses
            dev = col_feats.device
            key = (dev.type, dev.index, tuple(shapes))
            cached = _shape_tensors.get(key)
            if cached is None:
                cached = _shape_tensors[key] = (torch.tensor(shapes, dtype=torch.int64, device=dev),
                                                torch.tensor(starts, dtype=torch.int64, device=dev))
            return [col_feats, cached[0], cached[1]]
        spatial_shape = feature_maps[1].int()
        sizes = (spatial_shape[:, 0] * spatial_shape[:, 1]).tolist()
        maps = feature_maps[0].permute(0, 1, 3, 2)
        out = []
        for i, f in enumerate(torch.split(maps, sizes, dim=-1)):
            out.append(f.reshape(f.shape[:3] + (int(spatial_shape[i, 0]), int(spatial_shape[i, 1]))))
        return out
