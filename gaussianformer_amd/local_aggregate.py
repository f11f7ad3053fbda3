"""Host-side mirror of the reference's three splat packages
(``local_aggregate``, ``local_aggregate_prob``, ``local_aggregate_prob_fast``).

Same class names, constructor arguments, forward signature, outputs, gradient routing
and error behaviour as
  model/head/localagg/local_aggregate/__init__.py:18-161            (base)
  model/head/localagg_prob/local_aggregate_prob/__init__.py:18-169  (prob)
  model/head/localagg_prob_fast/local_aggregate_prob_fast/__init__.py:151 (per-axis radii)
so ``GaussianHead`` (model/head/gaussian_head.py:30-39) can import these unchanged.
The compute goes through the C ABI of ``libgf_hip.so``; there is no CPU path.
"""
import os
import threading

import torch
import torch.nn as nn

from . import _lib

_C18 = _lib.GF_NUM_CHANNELS


class _Workspace:
    """Scratch reused across calls, one grow-only buffer per (device, stream): the kernels of a call run on torch's
    current stream, so two calls on different streams must not share records / bitmask / verdict words.  A buffer
    is allocated while its stream is current, so the caching allocator frees it in that stream's order.  The cache is
    bounded (least recently used of ``MAX_STREAMS`` entries goes): a process that keeps creating streams does not
    accumulate a megabyte-plus per stream ever used, and a recycled stream handle meets at worst its own old buffer."""
    MAX_STREAMS = 8
    _cache = {}
    _uses = 0

    @classmethod
    def get(cls, device, nbytes):
        """The stream's buffer.  Its first 32 KB -- the library's flag section, the same words whatever the call's shape -- are zeroed
        when the buffer is allocated and afterwards only written by the library: what ``GF_WORKSPACE_ZEROED`` promises (the
        matrix-core forward then keeps its fall-back verdict in one word instead of one per wave of the records pass)."""
        key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
        buf = cls._cache.pop(key, None)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
            buf[:cls.FLAG_BYTES].zero_()
        cls._cache[key] = buf              # most recently used last
        while len(cls._cache) > cls.MAX_STREAMS:
            cls._cache.pop(next(iter(cls._cache)))
        cls._uses += 1
        return buf

    FLAG_BYTES = 32768
    _zeroed = {}

    @classmethod
    def get_zeroed(cls, device, nbytes, shape):
        """A buffer of its own per (stream, problem shape), zeroed when created and never handed to another shape: what
        the development build's fused forward needs (``dev.splat_fused``: its flags and counters are tagged per launch instead of
        being reset, which only holds in memory no other layout has written)."""
        key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream, nbytes, shape)
        buf = cls._zeroed.get(key)
        if buf is None:
            if len(cls._zeroed) >= cls.MAX_STREAMS:
                cls._zeroed.pop(next(iter(cls._zeroed)))
            buf = cls._zeroed[key] = torch.zeros(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        cls._uses += 1
        return buf

    @classmethod
    def stamp(cls, device):
        """(stream's buffer, number of hand-outs so far): equal stamps = nobody has been given a workspace in between, so the
        buffer still holds what the last call left in it (``GF_RECORDS_VALID``; the library checks it again on the device)."""
        key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
        buf = cls._cache.get(key)
        return (key, None if buf is None else buf.data_ptr(), cls._uses)


_lattice_cache = {}


def pts_is_exact_lattice(pts, H, W, D):
    """Host-side twin of the device lattice verdict (``gf_splat_prep_kernel``): is ``pts [N,3]`` exactly
    ``p0 + index * step`` per axis (fp64 comparison) in x-major / z-fastest order?  Evaluated once per tensor
    (data pointer, version counter, shape) and cached -- the voxel grid of a model does not change between frames.
    The modules use it to route grids whose centres are NOT exactly representable (e.g. a 0.4 m cell) to the
    exact-fp32 kernel instead of letting the matrix-core kernel's device verdict send every frame to the slow
    arbitrary-points body.  One host read on a cache miss."""
    key = (pts.device.index, pts.data_ptr(), pts._version, tuple(pts.shape), H, W, D)
    hit = _lattice_cache.get(key)
    if hit is not None:
        return hit
    ok = False
    if pts.dim() == 2 and pts.shape[0] == H * W * D and pts.shape[1] == 3:
        g = pts.detach().reshape(H, W, D, 3).double()
        p0 = g[0, 0, 0]
        step = torch.stack([g[1, 0, 0, 0] - p0[0] if H > 1 else p0.new_ones(()),
                            g[0, 1, 0, 1] - p0[1] if W > 1 else p0.new_ones(()),
                            g[0, 0, 1, 2] - p0[2] if D > 1 else p0.new_ones(())])
        ix = torch.arange(H, device=pts.device, dtype=torch.float64)[:, None, None]
        iy = torch.arange(W, device=pts.device, dtype=torch.float64)[None, :, None]
        iz = torch.arange(D, device=pts.device, dtype=torch.float64)[None, None, :]
        ok = bool(((g[..., 0] == p0[0] + ix * step[0]) & (g[..., 1] == p0[1] + iy * step[1]) &
                   (g[..., 2] == p0[2] + iz * step[2])).all().item())
    if len(_lattice_cache) > 16:
        _lattice_cache.clear()
    _lattice_cache[key] = ok
    return ok


def grid_is_exact_lattice(pc_min, grid_size, H, W, D):
    """Host-only (no device work, no synchronisation): are the fp32 voxel centres of a module's grid --
    ``arange(n) * grid_size + 0.5 * grid_size + pc_min`` in fp32, the arithmetic of the reference's ``get_meshgrid``
    (dataset/transform_3d.py:487-499) -- an exact affine lattice ``c[0] + i * (c[1] - c[0])``?  True for the nuScenes
    grid (multiples of 0.25 m), false e.g. for a 0.4 m cell.  The modules use it to choose the exact-fp32 kernel up front
    for grids the matrix-core kernel's device verdict would send to the slow arbitrary-points body every frame."""
    import numpy as np
    g = np.float32(grid_size)
    for n, lo in zip((H, W, D), pc_min):
        c = (np.arange(n, dtype=np.float32) * g + np.float32(0.5) * g + np.float32(lo)).astype(np.float64)
        step = c[1] - c[0] if n > 1 else 1.0
        if not np.array_equal(c, c[0] + np.arange(n, dtype=np.float64) * step):
            return False
    return True


def _contig(t, dtype):
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


def splat_forward(variant, pts, points_int, means3D, means3D_int, opacities, semantics, radii, cov3D,
                  H, W, D, flags=_lib.GF_PTS_AUTO):
    """Raw op: the counterpart of ``_C.local_aggregate``
    (model/head/localagg/local_aggregate.h:18-28; prob: localagg_prob/local_aggregate.cu:35-45).
    Returns ``(logits, bin_logits, density, probability, state)``; the last four are ``None``
    for the base variant except ``state`` (a small device block consumed by the backward)."""
    lib = _lib.load()
    _lib.require_gpu(pts, points_int, means3D, means3D_int, opacities, semantics, radii, cov3D)
    dev = pts.device
    f32, i32 = torch.float32, torch.int32
    pts, means3D, opacities, semantics, cov3D = (_contig(t, f32) for t in (pts, means3D, opacities, semantics, cov3D))
    points_int, means3D_int, radii = (_contig(t, i32) for t in (points_int, means3D_int, radii))
    N, P, C = pts.shape[0], means3D.shape[0], semantics.shape[1]
    if C != _C18:
        raise RuntimeError(f"semantics must have {_C18} channels (NUM_CHANNELS), got {C}")
    per_axis = int(radii.dim() == 2)
    prob = variant == _lib.GF_SPLAT_PROB
    logits = torch.empty((N, C), dtype=f32, device=dev)
    bin_logits = density = probability = None
    if prob:
        bin_logits = torch.empty(N, dtype=f32, device=dev)
        density = torch.empty(N, dtype=f32, device=dev)
        probability = torch.empty(N, dtype=f32, device=dev)
    state = torch.empty(lib.gf_splat_state_bytes(), dtype=torch.uint8, device=dev)
    nbytes = lib.gf_splat_workspace_bytes(P, N, H, W, D)
    if _lib.is_development_build() and _lib.get_option("dev.splat_fused"):   # (tools/ only: see _Workspace.get_zeroed)
        ws = _Workspace.get_zeroed(dev, nbytes, (P, N, H, W, D))
    else:
        ws = _Workspace.get(dev, nbytes)
    flags |= _lib.GF_WORKSPACE_ZEROED
    with torch.cuda.device(dev):
        rc = lib.gf_splat_forward(
            variant, per_axis, flags, P, N, C, H, W, D,
            _lib.ptr(pts), _lib.ptr(points_int), _lib.ptr(means3D), _lib.ptr(means3D_int), _lib.ptr(opacities),
            _lib.ptr(semantics), _lib.ptr(radii), _lib.ptr(cov3D),
            _lib.ptr(logits), _lib.ptr(bin_logits), _lib.ptr(density), _lib.ptr(probability), _lib.ptr(state),
            _lib.ptr(ws), ws.numel(), _lib.current_stream(dev))
    _lib.check(rc, "gf_splat_forward")
    return logits, bin_logits, density, probability, state


def splat_forward_labels(variant, pts, points_int, means3D, means3D_int, opacities, semantics, radii, cov3D,
                         H, W, D, threshold=0.5, empty_label=17, combine_geosem=False, keep_logits=False,
                         flags=_lib.GF_PTS_AUTO):
    """Forward splat with the head epilogue folded in (``gf_splat_forward_labels``): returns
    ``labels`` (int64 ``[N]``), or ``(labels, logits[, bin_logits, density, probability])`` with
    ``keep_logits``.  Without ``keep_logits`` the 46 MB of logits are never written."""
    lib = _lib.load()
    _lib.require_gpu(pts, points_int, means3D, means3D_int, opacities, semantics, radii, cov3D)
    f32, i32 = torch.float32, torch.int32
    pts, means3D, opacities, semantics, cov3D = (_contig(t, f32) for t in (pts, means3D, opacities, semantics, cov3D))
    points_int, means3D_int, radii = (_contig(t, i32) for t in (points_int, means3D_int, radii))
    N, P, C = pts.shape[0], means3D.shape[0], semantics.shape[1]
    dev = pts.device
    prob = variant == _lib.GF_SPLAT_PROB
    mode = _lib.GF_LABELS_ARGMAX if not prob else (
        _lib.GF_LABELS_PROB_GEOSEM if combine_geosem else _lib.GF_LABELS_PROB_THRESHOLD)
    labels = torch.empty(N, dtype=torch.int64, device=dev)
    logits = torch.empty((N, C), dtype=f32, device=dev) if keep_logits else None
    extra = [torch.empty(N, dtype=f32, device=dev) for _ in range(3)] if (prob and keep_logits) else [None] * 3
    state = torch.empty(lib.gf_splat_state_bytes(), dtype=torch.uint8, device=dev)
    nbytes = lib.gf_splat_workspace_bytes(P, N, H, W, D)
    ws = _Workspace.get(dev, nbytes)
    flags |= _lib.GF_WORKSPACE_ZEROED
    with torch.cuda.device(dev):
        rc = lib.gf_splat_forward_labels(variant, int(radii.dim() == 2), flags, P, N, C, H, W, D,
                                         _lib.ptr(pts), _lib.ptr(points_int), _lib.ptr(means3D), _lib.ptr(means3D_int),
                                         _lib.ptr(opacities), _lib.ptr(semantics), _lib.ptr(radii), _lib.ptr(cov3D),
                                         _lib.ptr(logits), *[_lib.ptr(e) for e in extra], mode, float(threshold),
                                         int(empty_label), _lib.ptr(labels), _lib.ptr(state), _lib.ptr(ws), nbytes,
                                         _lib.current_stream(dev))
    _lib.check(rc, "gf_splat_forward_labels")
    if not keep_logits:
        return labels
    return (labels, logits, *extra) if prob else (labels, logits)


class SplatForwardPlan:
    """Pre-bound forward call for a fixed set of device tensors: outputs, state and workspace
    are allocated once and ``run()`` is a single C-ABI call (no Python-side allocation).
    Used by bench.py and by callers that splat the same buffers every frame."""

    def __init__(self, variant, pts, points_int, means3D, means3D_int, opacities, semantics, radii, cov3D,
                 H, W, D, flags=_lib.GF_PTS_AUTO):
        self.lib = _lib.load()
        _lib.require_gpu(pts, points_int, means3D, means3D_int, opacities, semantics, radii, cov3D)
        f32, i32 = torch.float32, torch.int32
        self.inputs = [_contig(t, f32) if t.is_floating_point() else _contig(t, i32)
                       for t in (pts, points_int, means3D, means3D_int, opacities, semantics, radii, cov3D)]
        pts, points_int, means3D, means3D_int, opacities, semantics, radii, cov3D = self.inputs
        self.device = pts.device
        N, P, C = pts.shape[0], means3D.shape[0], semantics.shape[1]
        self.N, self.P = N, P
        prob = variant == _lib.GF_SPLAT_PROB
        self.logits = torch.empty((N, C), dtype=f32, device=self.device)
        self.bin_logits = torch.empty(N, dtype=f32, device=self.device) if prob else None
        self.density = torch.empty(N, dtype=f32, device=self.device) if prob else None
        self.probability = torch.empty(N, dtype=f32, device=self.device) if prob else None
        self.state = torch.empty(self.lib.gf_splat_state_bytes(), dtype=torch.uint8, device=self.device)
        self.workspace = torch.zeros(self.lib.gf_splat_workspace_bytes(P, N, H, W, D), dtype=torch.uint8,
                                     device=self.device)   # (zeroed once: GF_WORKSPACE_ZEROED)
        flags |= _lib.GF_WORKSPACE_ZEROED
        self.args = [variant, int(radii.dim() == 2), flags, P, N, C, H, W, D,
                     *[_lib.ptr(t) for t in self.inputs],
                     _lib.ptr(self.logits), _lib.ptr(self.bin_logits), _lib.ptr(self.density),
                     _lib.ptr(self.probability), _lib.ptr(self.state), _lib.ptr(self.workspace),
                     self.workspace.numel()]

    def run(self, stream=None):
        if stream is None:
            stream = _lib.current_stream(self.device)
        if torch.cuda.current_device() != self.device.index:
            with torch.cuda.device(self.device):   # the launch goes to the tensors' device, not the current one
                rc = self.lib.gf_splat_forward(*self.args, stream)
        else:
            rc = self.lib.gf_splat_forward(*self.args, stream)
        if rc:
            _lib.check(rc, "gf_splat_forward")
        return self.logits

    def state_words(self):
        """(not-dense flag, GF_PATH_* of the body that rendered the last call, verdict bits) -- synchronises."""
        return self.state[:12].view(torch.int32).tolist()


class SplatForwardPipeline:
    """``depth`` pre-bound plans (own outputs, state and workspace) served round-robin on ``depth``
    HIP streams: with two frames in flight the latency-bound prep kernel, the cold start and the
    tail of one frame overlap the render kernel of the other (38.6 vs 59 us per frame at
    gs25600 on one MI355X).  ``submit()`` enqueues the next frame and returns ``(logits, event)``;
    wait on the event (or synchronise) before reading that frame's logits, and before the same
    slot is submitted again ``depth`` frames later."""

    def __init__(self, variant, pts, points_int, means3D, means3D_int, opacities, semantics, radii, cov3D,
                 H, W, D, flags=_lib.GF_PTS_AUTO, depth=2):
        self.plans = [SplatForwardPlan(variant, pts, points_int, means3D, means3D_int, opacities, semantics, radii,
                                       cov3D, H, W, D, flags=flags) for _ in range(depth)]
        dev = self.plans[0].device
        self.streams = [torch.cuda.Stream(dev) for _ in range(depth)]
        self.events = [torch.cuda.Event() for _ in range(depth)]
        self.next = 0

    def submit(self):
        i = self.next
        self.next = (i + 1) % len(self.plans)
        stream = self.streams[i]
        stream.wait_stream(torch.cuda.current_stream(self.plans[i].device))  # inputs written on the caller's stream
        logits = self.plans[i].run(stream.cuda_stream)
        self.events[i].record(stream)
        return logits, self.events[i]


BACKWARD_MAX_GAUSSIANS = 262144   # gf_splat_backward's limit per call (include/gf_hip.h)


def splat_backward(variant, pts, points_int, means3D, means3D_int, opacities, semantics, radii, cov3D,
                   H, W, D, logits_grad, fwd_outputs=None, bin_logits_grad=None, density_grad=None,
                   state=None, flags=_lib.GF_PTS_AUTO):
    """Raw op: the counterpart of ``_C.local_aggregate_backward``
    (model/head/localagg/local_aggregate.h:30-43).  ``fwd_outputs`` =
    ``(logits, bin_logits, density, probability)`` for the prob variant.
    Returns ``(means3D_grad, opacity_grad, semantics_grad, cov3D_grad)``."""
    lib = _lib.load()
    _lib.require_gpu(pts, points_int, means3D, means3D_int, opacities, semantics, radii, cov3D, logits_grad)
    if means3D.shape[0] > BACKWARD_MAX_GAUSSIANS:
        # gf_splat_backward takes 262 144 Gaussians per call (include/gf_hip.h); the reference has no such limit, and gradients are
        # per Gaussian: a call per shard with the same out_grad (and the forward's per-point outputs) gives the same rows.  A shard
        # does not have the forward's records: the Gaussian-major pipeline, whatever the flags asked for.
        sflags = (flags & (_lib.GF_PTS_ASSUME_DENSE | _lib.GF_PTS_GENERAL)) | _lib.GF_EXACT_FP32
        parts = []
        for lo in range(0, means3D.shape[0], BACKWARD_MAX_GAUSSIANS):
            sl = slice(lo, lo + BACKWARD_MAX_GAUSSIANS)
            parts.append(splat_backward(variant, pts, points_int, means3D[sl], means3D_int[sl], opacities[sl], semantics[sl],
                                        radii[sl], cov3D[sl], H, W, D, logits_grad, fwd_outputs=fwd_outputs,
                                        bin_logits_grad=bin_logits_grad, density_grad=density_grad, state=state, flags=sflags))
        return tuple(torch.cat(col, dim=0) for col in zip(*parts))
    dev = pts.device
    f32, i32 = torch.float32, torch.int32
    # the same coercions as the forward (the reference calls .contiguous().data<float>() on every backward
    # argument, local_aggregate.cu:116-127): autograd saves the caller's tensors, which may be strided views
    # or another dtype
    pts, means3D, opacities, semantics, cov3D = (_contig(t, f32) for t in (pts, means3D, opacities, semantics, cov3D))
    points_int, means3D_int, radii = (_contig(t, i32) for t in (points_int, means3D_int, radii))
    N, P, C = pts.shape[0], means3D.shape[0], semantics.shape[1]
    per_axis = int(radii.dim() == 2)
    logits_grad = _contig(logits_grad, f32)
    lg = bl = de = pr = None
    if variant == _lib.GF_SPLAT_PROB:
        lg, bl, de, pr = (_contig(t, f32) for t in fwd_outputs)
        bin_logits_grad = None if bin_logits_grad is None else _contig(bin_logits_grad, f32)
        density_grad = None if density_grad is None else _contig(density_grad, f32)
    mg = torch.empty((P, 3), dtype=f32, device=dev)
    og = torch.empty(P, dtype=f32, device=dev)
    sg = torch.empty((P, C), dtype=f32, device=dev)
    cg = torch.empty((P, 6), dtype=f32, device=dev)
    nbytes = lib.gf_splat_workspace_bytes(P, N, H, W, D)
    ws = _Workspace.get(dev, nbytes)
    with torch.cuda.device(dev):
        rc = lib.gf_splat_backward(
            variant, per_axis, flags, P, N, C, H, W, D,
            _lib.ptr(pts), _lib.ptr(points_int), _lib.ptr(means3D), _lib.ptr(means3D_int), _lib.ptr(opacities),
            _lib.ptr(semantics), _lib.ptr(radii), _lib.ptr(cov3D),
            _lib.ptr(lg), _lib.ptr(bl), _lib.ptr(de), _lib.ptr(pr),
            _lib.ptr(logits_grad), _lib.ptr(bin_logits_grad), _lib.ptr(density_grad),
            _lib.ptr(mg), _lib.ptr(og), _lib.ptr(sg), _lib.ptr(cg), _lib.ptr(state),
            _lib.ptr(ws), ws.numel(), _lib.current_stream(dev))
    _lib.check(rc, "gf_splat_backward")
    return mg, og, sg, cg


def splat_box_volumes(means3D_int, radii, H, W, D):
    """(tiles_touched u32-as-int64 [P], num_rendered int) -- integer-exact counterpart of
    FORWARD::preprocessCUDA + the inclusive scan's last element
    (model/head/localagg/src/forward.cu:9-28, src/aggregator_impl.cu:193-197).
    Reading ``num_rendered`` synchronises the stream (the reference does too, :197)."""
    lib = _lib.load()
    _lib.require_gpu(means3D_int, radii)
    dev = means3D_int.device
    means3D_int = _contig(means3D_int, torch.int32)
    radii = _contig(radii, torch.int32)
    P = means3D_int.shape[0]
    touched = torch.zeros(P, dtype=torch.int32, device=dev)
    total = torch.zeros(1, dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        rc = lib.gf_splat_box_volumes(int(radii.dim() == 2), P, H, W, D, _lib.ptr(means3D_int), _lib.ptr(radii),
                                      _lib.ptr(touched), _lib.ptr(total), _lib.current_stream(dev))
    _lib.check(rc, "gf_splat_box_volumes")
    return touched.to(torch.int64) & 0xFFFFFFFF, int(total.item())


_tls = threading.local()


class _LocalAggregate(torch.autograd.Function):
    """model/head/localagg/local_aggregate/__init__.py:18-106."""

    @staticmethod
    def forward(ctx, pts, points_int, means3D, means3D_int, opacities, semantics, radii, cov3D, H, W, D, flags=_lib.GF_PTS_AUTO):
        # (a backward will follow: the forward lays out its partial-gradient rows on the way -- GF_PREPARE_BACKWARD)
        wants_grad = any(ctx.needs_input_grad)
        logits, _, _, _, state = splat_forward(_lib.GF_SPLAT_BASE, pts, points_int, means3D, means3D_int,
                                               opacities, semantics, radii, cov3D, H, W, D,
                                               flags=flags | (_lib.GF_PREPARE_BACKWARD if wants_grad else 0))
        ctx.dims = (H, W, D)
        ctx.fwd_flags = flags
        # Which body rendered the call is only known on the device (word 1 of the state block).  The backward has a kernel for
        # either answer; launched without knowing, both pipelines go out, each gated on that word.  So the three words are copied
        # to pinned host memory behind an event -- never waited for: by the time the backward runs the copy has usually landed
        # and exactly one pipeline is launched (not while a HIP graph is being captured: no host allocation there).
        ctx.state_host = ctx.state_event = None
        ctx.ws_stamp = _Workspace.stamp(pts.device) if state.is_cuda else None
        if any(ctx.needs_input_grad) and state.is_cuda and not torch.cuda.is_current_stream_capturing():
            ctx.state_host = torch.empty(5, dtype=torch.int32, pin_memory=True)
            ctx.state_host.copy_(state[:20].view(torch.int32), non_blocking=True)
            ctx.state_event = torch.cuda.Event()
            ctx.state_event.record(torch.cuda.current_stream(pts.device))
        ctx.save_for_backward(state, means3D, means3D_int, pts, points_int, cov3D, opacities, semantics, radii)
        _tls.last_state = state   # (this thread's last call: picked up by LocalAggregator._splat right after apply() returns)
        return logits

    @staticmethod
    def backward(ctx, out_grad):
        H, W, D = ctx.dims
        state, means3D, means3D_int, pts, points_int, cov3D, opacities, semantics, radii = ctx.saved_tensors
        # GF_EXACT_FP32: the Gaussian-major exact kernels (what an exact forward is paired with); GF_MFMA_SPLAT: the matrix-core
        # backward, asserted from the forward's own state words; neither: both pipelines, gated on the device
        if ctx.fwd_flags & _lib.GF_EXACT_FP32:
            bflags = _lib.GF_EXACT_FP32
        elif ctx.state_event is not None and ctx.state_event.query():
            words = ctx.state_host.tolist()
            on_matrix_cores = words[0] == 0 and words[1] in _lib.GF_PATHS_MATRIX_CORE
            # (word 4, bit 1: the forward took the matrix-core backward's row layout and the rows do not fit its buffer -- many
            # large Gaussians; what does not fit would be added with atomics, tens of times slower than the Gaussian-major kernels)
            bflags = _lib.GF_MFMA_SPLAT if on_matrix_cores and not (words[4] & 2) else _lib.GF_EXACT_FP32
            # the forward's records pass laid out the backward's rows as well (word 4); if nobody has been handed this stream's
            # workspace since, the backward does not repeat that pass
            if on_matrix_cores and (words[4] & 1) and ctx.ws_stamp == _Workspace.stamp(out_grad.device):
                bflags |= _lib.GF_RECORDS_VALID
        else:
            bflags = _lib.GF_PTS_AUTO
        mg, og, sg, cg = splat_backward(_lib.GF_SPLAT_BASE, pts, points_int, means3D, means3D_int, opacities,
                                        semantics, radii, cov3D, H, W, D, out_grad, state=state, flags=bflags)
        # grads for (means3D, opacities, semantics, cov3D) only -- :91-104.  The reference returns the opacity
        # gradient as [P] whatever the input's shape; a [P,1] opacity that requires grad would be rejected by
        # autograd there, so it is reshaped here.
        return None, None, mg, None, og.view_as(opacities), sg, None, cg, None, None, None, None


class _LocalAggregateProb(torch.autograd.Function):
    """model/head/localagg_prob/local_aggregate_prob/__init__.py:18-116."""

    @staticmethod
    def forward(ctx, pts, points_int, means3D, means3D_int, opas, semantics, radii, cov3D, H, W, D):
        logits, bin_logits, density, probability, state = splat_forward(
            _lib.GF_SPLAT_PROB, pts, points_int, means3D, means3D_int, opas, semantics, radii, cov3D, H, W, D)
        ctx.dims = (H, W, D)
        ctx.save_for_backward(state, means3D, means3D_int, pts, points_int, cov3D, opas, semantics, radii,
                              logits, bin_logits, density, probability)
        return logits, bin_logits, density

    @staticmethod
    def backward(ctx, logits_grad, bin_logits_grad, density_grad):
        H, W, D = ctx.dims
        (state, means3D, means3D_int, pts, points_int, cov3D, opas, semantics, radii,
         logits, bin_logits, density, probability) = ctx.saved_tensors
        mg, og, sg, cg = splat_backward(_lib.GF_SPLAT_PROB, pts, points_int, means3D, means3D_int, opas, semantics,
                                        radii, cov3D, H, W, D, logits_grad,
                                        fwd_outputs=(logits, bin_logits, density, probability),
                                        bin_logits_grad=bin_logits_grad, density_grad=density_grad, state=state)
        return None, None, mg, None, og.view_as(opas), sg, None, cg, None, None, None


class _AggregatorBase(nn.Module):
    """Shared pre-processing of the three ``LocalAggregator`` classes."""

    def _points_int(self, pts):
        """Voxel indices of the query points: fp32 subtract, fp32 true division, truncation (``.to(torch.int)``) --
        model/head/localagg/local_aggregate/__init__.py:137-141.  The voxel grid of a model does not change between frames: the
        result is kept for the LAST ``pts`` tensor seen (same storage, shape and version counter, same ``pc_min`` / cell; a strong
        reference, so the address cannot be recycled) -- three passes over 640 000 points and their launches per frame otherwise.
        A caller that builds a new tensor per frame simply recomputes.  Two caveats: a buffer rewritten behind autograd's back
        (raw pointer, ``.data``) does not move the version counter -- pass a new tensor then; and while a HIP graph is being
        captured the cache is neither read nor written (the indices are computed inside the capture, in the graph's own pool, so
        a replay never reads memory that a later eager call released)."""
        capturing = pts.is_cuda and torch.cuda.is_current_stream_capturing()
        key = (pts.data_ptr(), pts._version, tuple(pts.shape), pts.device, self.pc_min.data_ptr(), self.pc_min._version, float(self.grid_size))
        hit = getattr(self, "_points_int_cache", None)
        if not capturing and hit is not None and hit[0] == key and hit[1]._version == key[1]:
            return hit[2]
        points_int = ((pts - self.pc_min) / self.grid_size).to(torch.int)
        if pts.is_cuda and not capturing:
            self._points_int_cache = (key, pts, points_int)
        return points_int

    def _prepare(self, pts, means3D, opacities, semantics, scales, cov3D):
        assert pts.shape[0] == 1
        pts = pts.squeeze(0)
        assert not pts.requires_grad
        means3D = means3D.squeeze(0)
        opacities = opacities.squeeze(0)
        semantics = semantics.squeeze(0)
        scales = scales.detach().squeeze(0)
        cov3D = cov3D.squeeze(0)
        # integer path: fp32 subtract, fp32 true division, truncation (.to(torch.int)) --
        # model/head/localagg/local_aggregate/__init__.py:137-141
        points_int = self._points_int(pts)
        means3D_int = ((means3D.detach() - self.pc_min) / self.grid_size).to(torch.int)
        violations = None
        if self.check_inputs:
            # the reference's range asserts (:138-140), evaluated on the device and read back ONCE per call by
            # _raise_on_violation (the reference synchronises eight times here).  Returned, not stored on the module:
            # a module shared by several threads / streams stays re-entrant.
            hi = points_int.new_tensor([self.H, self.W, self.D])
            violations = torch.stack([
                (points_int < 0).any() | (points_int >= hi).any(),
                (means3D_int < 0).any() | (means3D_int >= hi).any()])
        return pts, points_int, means3D, means3D_int, opacities, semantics, scales, cov3D, violations

    _radii_mode = _lib.GF_RADII_SCALAR

    def _radii(self, scales):
        """Integer radii of the wrapper (local_aggregate/__init__.py:141; prob: :150-152; prob_fast: :151)."""
        raise NotImplementedError

    def forward_slab(self, x0, x1, pts, means3D, opacities, semantics, scales, cov3D):
        """The outputs of ``forward`` for the voxel rows ``x0 <= x < x1`` only -- the per-rank op of the SPATIAL (slab)
        partition (``sharded.slab_splat_forward``; an addition over the reference, which runs replicas only, train.py:41-43).
        ``pts [1, H*W*D, 3]`` must be the dense voxel-centre grid in the reference's x-major order; all P Gaussians
        are passed: the integer coordinates are taken on the FULL grid exactly as ``forward`` takes them and shifted by
        ``x0`` afterwards, so a Gaussian's box is clipped to the slab the way ``getRect`` clips it to the grid (a box
        that misses the slab comes out empty and costs nothing), every voxel of the slab sees the Gaussians it sees
        in the full call, in the same ascending order: the slab's rows equal the full call's rows bit for bit when the
        slab starts on a multiple of 8 rows (the binning granule; any start with ``matrix_cores=False``).  Differentiable
        like ``forward``."""
        assert 0 <= x0 < x1 <= self.H and pts.shape[0] == 1 and pts.shape[1] == self.H * self.W * self.D
        n0, n1 = x0 * self.W * self.D, x1 * self.W * self.D
        pts_s, points_int, means3D, means3D_int, opacities, semantics, scales, cov3D, violations = self._prepare(
            pts[:, n0:n1], means3D, opacities, semantics, scales, cov3D)
        radii = self._radii(scales)
        self._raise_on_violation(violations, radii)
        cov6 = cov3D.flatten(1)[:, [0, 4, 8, 1, 5, 2]]
        shift = points_int.new_tensor([x0, 0, 0])
        return self._splat(pts_s, points_int - shift, means3D, means3D_int - shift, opacities, semantics, radii, cov6, H=x1 - x0)

    def _raise_on_violation(self, violations, radii):
        """One host read for the three range conditions the reference asserts one by one
        (local_aggregate/__init__.py:138,140,142): same AssertionError, one synchronisation."""
        if not self.check_inputs:
            return
        bad = torch.cat([violations, (radii < 1).any().reshape(1)]).tolist()
        assert not bad[0], "points outside the voxel grid (points_int out of [0,H)x[0,W)x[0,D))"
        assert not bad[1], "Gaussian centres outside the voxel grid (means3D_int out of range)"
        assert not bad[2], "radii must be >= 1"

    def forward_from_rotations(self, pts, means3D, opacities, semantics, scales, rotations):
        """Fused entry: takes the Gaussians' ``rotations [1,g,4]`` instead of
        ``CovInv`` and computes Sigma^-1, ``means3D_int`` and ``radii`` in one kernel
        (``gf_gaussian_prepare``) -- no 3x3 tensors, no host inverse (gaussian_head.py:111-119),
        no per-call syncs.  Returns what ``forward`` returns; gradients reach ``scales`` and
        ``rotations`` through Sigma^-1 exactly as in the reference graph."""
        from .gaussian_prepare import _GaussianPrepare
        assert pts.shape[0] == 1
        pts = pts.squeeze(0)
        assert not pts.requires_grad
        means3D, opacities, semantics = means3D.squeeze(0), opacities.squeeze(0), semantics.squeeze(0)
        status = None
        if self.check_inputs:
            status = torch.zeros(1, dtype=torch.int32, device=pts.device)
        means3D_int, radii, cov6 = _GaussianPrepare.apply(
            means3D, scales.squeeze(0), rotations.squeeze(0), self._pc_min_host, self.grid_size,
            self.scale_multiplier, self.H, self.W, self.D, self._radii_mode, getattr(self, "radii_min", 1), status)
        points_int = self._points_int(pts)
        if self.check_inputs:
            assert int(status.item()) == 0, f"gaussian_prepare status {int(status.item())} (GF_PREPARE_* bits)"
        return self._splat(pts, points_int, means3D, means3D_int, opacities, semantics, radii, cov6)


class LocalAggregator(_AggregatorBase):
    """Drop-in for ``local_aggregate.LocalAggregator``
    (model/head/localagg/local_aggregate/__init__.py:108-161).  ``check_inputs`` (extra keyword, default on)
    keeps the reference's per-call range asserts -- evaluated on the device, one host read instead of eight;
    ``check_inputs=False`` makes the call fully asynchronous (boxes of out-of-grid centres are then clipped
    the way ``getRect`` clips them).

    ``matrix_cores`` (extra keyword) selects the forward kernel: ``None`` (default) = the library default, i.e. the
    split-f16 MFMA kernel (``GF_MFMA_SPLAT``: ~2e-5 from the reference, tolerance 1e-4) whenever ``pts`` is the dense
    grid of exactly representable voxel centres -- judged once per module from (pc_min, grid_size, H, W, D) on the host
    (``grid_is_exact_lattice``: no device work, no synchronisation), so that a grid that is not (e.g. a 0.4 m cell) goes
    straight to the exact-fp32 kernel; the device verdict still guards every call -- ``True`` = request it
    regardless of that check (the device verdict still guards every call), ``False`` = always the exact-fp32 kernel
    (``GF_EXACT_FP32``: ~2e-6 from the reference).  The backward follows the forward: after a matrix-core forward the
    voxel-major matrix-core backward (``splat_bwd_mfma.hip``, gradients ~1e-5 from the reference, bound 1e-3), after an exact
    forward the exact Gaussian-major kernels."""

    def __init__(self, scale_multiplier, H, W, D, pc_min, grid_size, inv_softmax=False, check_inputs=True, matrix_cores=None):
        super().__init__()
        self.matrix_cores = matrix_cores
        self.scale_multiplier = scale_multiplier
        self.H = H
        self.W = W
        self.D = D
        self.register_buffer('pc_min', torch.tensor(pc_min, dtype=torch.float).unsqueeze(0))
        self.grid_size = grid_size
        self.inv_softmax = inv_softmax
        self.check_inputs = check_inputs
        self._pc_min_host = [float(v) for v in pc_min]
        self._grid_exact = None     # grid_is_exact_lattice(...) of this module's grid, judged on first use
        self._registered = None     # register_grid()

    def register_grid(self, pts):
        """Optional: verify ONCE that ``pts [1, H*W*D, 3]`` (or ``[H*W*D, 3]``) is the dense grid -- point n in voxel n and an
        exact affine lattice: one host read -- and remember the tensor (a strong reference, so its address cannot be recycled
        by another tensor).  Later ``forward`` calls that pass this very tensor, unmodified (same storage address, shape and
        version counter), then tell the library so (``GF_PTS_ASSUME_DENSE``): the per-call scan of pts / points_int -- 15 MB
        and ~3.6 us at the nuScenes grid -- is skipped.  The range verdicts, which depend on each frame's Gaussians, still
        run in every call.  Returns whether the tensor qualified.  Callers that build a new ``pts`` tensor per frame (the
        reference's dataloader does) simply never hit it and keep the per-call device verdict."""
        flat = pts.detach().reshape(-1, 3)
        ok = flat.shape[0] == self.H * self.W * self.D and pts_is_exact_lattice(flat, self.H, self.W, self.D)
        if ok:
            pi = ((flat - self.pc_min.to(flat.device)) / self.grid_size).to(torch.int).long()
            key = (pi[:, 0] * self.W + pi[:, 1]) * self.D + pi[:, 2]
            ok = bool(((pi[:, 1] >= 0) & (pi[:, 1] < self.W) & (pi[:, 2] >= 0) & (pi[:, 2] < self.D) &
                       (key == torch.arange(flat.shape[0], device=flat.device))).all().item())
        self._registered = (pts, pts.data_ptr(), pts._version, flat.shape[0]) if ok else None
        return ok

    def _is_registered(self, pts):
        reg = getattr(self, "_registered", None)
        return (reg is not None and pts.data_ptr() == reg[1] and pts._version == reg[2] and reg[0]._version == reg[2]
                and pts.shape[0] == reg[3] and pts.is_contiguous())

    def _splat(self, pts, *args, H=None):
        H = self.H if H is None else H
        pts_flag = _lib.GF_PTS_ASSUME_DENSE if (H == self.H and self._is_registered(pts)) else _lib.GF_PTS_AUTO
        if self.matrix_cores is None:
            # the grid of this module, judged on the host once (no device work): not an exact lattice -> exact-fp32 kernel
            if self._grid_exact is None:
                self._grid_exact = grid_is_exact_lattice(self._pc_min_host, self.grid_size, self.H, self.W, self.D)
            flags = pts_flag | (0 if self._grid_exact else _lib.GF_EXACT_FP32)
        else:
            flags = pts_flag | (_lib.GF_MFMA_SPLAT if self.matrix_cores else _lib.GF_EXACT_FP32)
        out = _LocalAggregate.apply(pts, *args, H, self.W, self.D, flags)
        state, _tls.last_state = getattr(_tls, "last_state", None), None   # (the state block of THIS call, not of another module's or thread's)
        self.last_state = state
        if self.matrix_cores is None and self._grid_exact and pts.shape[0] == H * self.W * self.D:
            self._watch_path(pts.device, state)
        return out

    def _watch_path(self, device, state):
        """The module judged its grid an exact lattice from (pc_min, grid_size); the ``pts`` actually passed may be built
        differently (e.g. in fp64 and cast), fail the device's lattice verdict and be rendered by the arbitrary-points body in
        every frame, ~7x slower, with word 1 of the state block as the only sign.  So the word is copied to pinned host memory
        now and then (first calls, then every 64th; behind an event, never waited for); when a copy says GF_PATH_ARBITRARY the
        module warns once and uses the exact-fp32 tile kernel from then on."""
        import warnings
        w = getattr(self, "_path_watch", None)
        if w is None:
            w = self._path_watch = {"calls": 0, "host": None, "event": None}
        if w["event"] is not None and w["event"].query():
            if int(w["host"][1]) == _lib.GF_PATH_ARBITRARY:
                warnings.warn("LocalAggregator: pts is a dense grid but not the exact fp32 lattice of (pc_min, grid_size): the "
                              "matrix-core kernel's device verdict sends every frame to the arbitrary-points body; using the "
                              "exact-fp32 kernel from now on (matrix_cores=False selects it up front)")
                self._grid_exact = False
            w["host"] = w["event"] = None
        w["calls"] += 1
        if (w["event"] is None and state is not None and state.is_cuda and (w["calls"] <= 3 or w["calls"] % 64 == 0)
                and not torch.cuda.is_current_stream_capturing()):
            w["host"] = torch.empty(5, dtype=torch.int32, pin_memory=True)
            w["host"].copy_(state[:20].view(torch.int32), non_blocking=True)
            w["event"] = torch.cuda.Event()
            w["event"].record(torch.cuda.current_stream(device))

    def _radii(self, scales):
        return torch.ceil(scales.max(dim=-1)[0] * self.scale_multiplier / self.grid_size).to(torch.int)

    def forward(self, pts, means3D, opacities, semantics, scales, cov3D):
        pts, points_int, means3D, means3D_int, opacities, semantics, scales, cov3D, violations = self._prepare(
            pts, means3D, opacities, semantics, scales, cov3D)
        radii = self._radii(scales)
        self._raise_on_violation(violations, radii)
        cov3D = cov3D.flatten(1)[:, [0, 4, 8, 1, 5, 2]]   # (xx, yy, zz, xy, yz, xz) of the 3x3, :143
        logits = self._splat(pts, points_int, means3D, means3D_int, opacities, semantics, radii, cov3D)
        assert not self.inv_softmax, "inv_softmax=True is an `assert False` in the reference too (:158-161)"
        return logits


class LocalAggregatorProb(_AggregatorBase):
    """Drop-in for ``local_aggregate_prob.LocalAggregator``
    (model/head/localagg_prob/local_aggregate_prob/__init__.py:118-169); ``per_axis_radii``
    selects ``local_aggregate_prob_fast`` behaviour (…_prob_fast/__init__.py:151)."""
    per_axis_radii = False

    def __init__(self, scale_multiplier, H, W, D, pc_min, grid_size, radii_min=1, check_inputs=True):
        super().__init__()
        self.scale_multiplier = scale_multiplier
        self.H = H
        self.W = W
        self.D = D
        self.register_buffer('pc_min', torch.tensor(pc_min, dtype=torch.float).unsqueeze(0))
        self.grid_size = grid_size
        self.radii_min = radii_min
        self.check_inputs = check_inputs
        self._pc_min_host = [float(v) for v in pc_min]

    @property
    def _radii_mode(self):
        return _lib.GF_RADII_PER_AXIS if self.per_axis_radii else _lib.GF_RADII_SCALAR_CLAMPED

    def _splat(self, *args, H=None):
        return _LocalAggregateProb.apply(*args, self.H if H is None else H, self.W, self.D)

    def _radii(self, scales):
        if self.per_axis_radii:
            radii = torch.ceil(scales * self.scale_multiplier / self.grid_size).to(torch.int)
        else:
            radii = torch.ceil(scales.max(dim=-1)[0] * self.scale_multiplier / self.grid_size).to(torch.int)
        return radii.clamp(min=self.radii_min)

    def forward(self, pts, means3D, opas, semantics, scales, cov3D):
        pts, points_int, means3D, means3D_int, opas, semantics, scales, cov3D, violations = self._prepare(
            pts, means3D, opas, semantics, scales, cov3D)
        radii = self._radii(scales)
        self._raise_on_violation(violations, radii)
        cov3D = cov3D.flatten(1)[:, [0, 4, 8, 1, 5, 2]]
        return _LocalAggregateProb.apply(pts, points_int, means3D, means3D_int, opas, semantics, radii, cov3D,
                                         self.H, self.W, self.D)

    @torch.no_grad()
    def forward_pieces(self, pts, means3D, opas, semantics, scales, cov3D):
        """Inference-only: ``(numerator [n,18], bin_logits [n], density [n], probability [n])`` with the
        un-normalised numerator (``GF_PROB_NUMERATOR``) -- what the shards of
        ``sharded.sharded_splat_forward_prob`` exchange before normalising."""
        pts, points_int, means3D, means3D_int, opas, semantics, scales, cov3D, violations = self._prepare(
            pts, means3D, opas, semantics, scales, cov3D)
        radii = self._radii(scales)
        self._raise_on_violation(violations, radii)
        cov3D = cov3D.flatten(1)[:, [0, 4, 8, 1, 5, 2]]
        numerator, bin_logits, density, probability, _ = splat_forward(
            _lib.GF_SPLAT_PROB, pts, points_int, means3D, means3D_int, opas, semantics, radii, cov3D,
            self.H, self.W, self.D, flags=_lib.GF_PTS_AUTO | _lib.GF_PROB_NUMERATOR)
        return numerator, bin_logits, density, probability


class LocalAggregatorProbFast(LocalAggregatorProb):
    """Drop-in for ``local_aggregate_prob_fast.LocalAggregator``."""
    per_axis_radii = True
