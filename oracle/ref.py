"""ctypes front-end of ``oracle/_ref``: the REFERENCE's own CUDA kernels, compiled for gfx950 by
``oracle/ref_build.py`` and executed on the GPU.  TEST INFRASTRUCTURE ONLY.

Same call signatures as ``oracle.oracle`` (numpy in, numpy out), so a parity test can take either
checker.  ``variant`` selects the library: 'base' → model/head/localagg, 'prob' → localagg_prob when
``radii`` is ``[P]`` and localagg_prob_fast when it is ``[P,3]``.

Needs a GPU (the libraries are device code) and the prebuilt ``oracle/_ref/libref_*.so``; nothing here
reads ``/root/reference`` at run time.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_DIR = os.path.join(_HERE, "_ref")
_libs = {}

__all__ = ["available", "splat_forward", "splat_backward", "splat_forward_backward", "daf_forward", "daf_backward"]

_NAMES = {"base": "libref_localagg.so", "prob": "libref_localagg_prob.so", "prob_fast": "libref_localagg_prob_fast.so",
          "daf": "libref_daf.so"}


def available():
    """True when all four reference libraries are present (built here by ``__graft_entry__.build()``)."""
    return all(os.path.exists(os.path.join(_DIR, n)) for n in _NAMES.values())


def _load(key):
    if key not in _libs:
        path = os.path.join(_DIR, _NAMES[key])
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `python -m oracle.ref_build` where /root/reference exists")
        lib = ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL)   # the three splat libraries export the same C++ symbols
        if key != "daf":
            lib.ref_splat_forward.restype = ctypes.c_void_p
            lib.ref_splat_free.argtypes = [ctypes.c_void_p]
            lib.ref_splat_free.restype = None
            lib.ref_splat_backward.argtypes = [ctypes.c_void_p] * 9
            lib.ref_splat_binning.argtypes = [ctypes.c_void_p] * 6
            assert lib.ref_num_channels() == 18
            assert lib.ref_variant() == {"base": 0, "prob": 1, "prob_fast": 2}[key]
        _libs[key] = lib
    return _libs[key]


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _key(variant, radii):
    if variant == "base":
        assert radii.ndim == 1, "the base reference extension takes scalar radii"
        return "base"
    return "prob_fast" if radii.ndim == 2 else "prob"


class _Session:
    def __init__(self, lib, handle, P, N, H, W, D, R):
        self.lib, self.handle, self.P, self.N, self.H, self.W, self.D, self.R = lib, handle, P, N, H, W, D, R

    def binning(self):
        """dict(tiles_touched[P], point_offsets[P], ranges[HWD,2], point_list[R], keys_unsorted[R]) as the
        reference's own preprocess / scan / duplicateWithKeys / radix sort / identifyTileRanges left them."""
        tt = np.zeros(self.P, np.uint32)
        po = np.zeros(self.P, np.uint32)
        rg = np.zeros((self.H * self.W * self.D, 2), np.uint32)
        pl = np.zeros(self.R, np.uint32)
        ku = np.zeros(self.R, np.uint32)
        rc = self.lib.ref_splat_binning(self.handle, _p(tt), _p(po), _p(rg), _p(pl), _p(ku))
        if rc:
            raise RuntimeError("ref_splat_binning failed")
        return dict(tiles_touched=tt, point_offsets=po, ranges=rg, point_list=pl, keys_unsorted=ku)

    def close(self):
        if self.handle:
            self.lib.ref_splat_free(self.handle)
            self.handle = None

    def __del__(self):
        self.close()


def _forward(variant, pts, points_int, means3D, means_int, opacity, semantics, radii, cov6, H, W, D):
    pts, means3D, opacity, semantics, cov6 = map(_f32, (pts, means3D, opacity, semantics, cov6))
    points_int, means_int, radii = map(_i32, (points_int, means_int, radii))
    N, P, C = pts.shape[0], means3D.shape[0], semantics.shape[1]
    assert C == 18 and cov6.shape == (P, 6)
    key = _key(variant, radii)
    lib = _load(key)
    logits = np.zeros((N, C), np.float32)
    outs = [np.zeros(N, np.float32) for _ in range(3)] if variant != "base" else [None] * 3
    R = ctypes.c_int(0)
    h = lib.ref_splat_forward(P, N, _p(pts), _p(points_int), _p(means3D), _p(means_int), _p(opacity), _p(semantics),
                              _p(radii), _p(cov6), H, W, D, _p(logits), _p(outs[0]), _p(outs[1]), _p(outs[2]),
                              ctypes.byref(R))
    if not h:
        raise RuntimeError("reference splat forward failed (HIP error; see stderr)")
    res = {"logits": logits, "num_rendered": int(R.value)}
    if variant != "base":
        res.update(bin_logits=outs[0], density=outs[1], probability=outs[2])
    return res, _Session(lib, h, P, N, H, W, D, int(R.value))


def splat_forward(variant, pts, points_int, means3D, means_int, opacity, semantics, radii, cov6, H, W, D,
                  with_binning=False):
    """The reference's ``Aggregator::forward`` (src/aggregator_impl.cu:152-252).  Returns the same dict as
    ``oracle.splat_forward``; ``with_binning`` adds the sorted pair list / ranges / offsets."""
    res, s = _forward(variant, pts, points_int, means3D, means_int, opacity, semantics, radii, cov6, H, W, D)
    if with_binning:
        res.update(s.binning())
    s.close()
    return res


def splat_forward_backward(variant, pts, points_int, means3D, means_int, opacity, semantics, radii, cov6, H, W, D,
                           out_grad, bin_grad=None, density_grad=None):
    """Forward then backward through the reference (the backward consumes the forward's scratch blobs, as
    ``_LocalAggregate.backward`` does, local_aggregate/__init__.py:66-106).
    Returns (forward dict, (means3D_grad, opacity_grad, semantics_grad, cov3D_grad), voxel2pts)."""
    res, s = _forward(variant, pts, points_int, means3D, means_int, opacity, semantics, radii, cov6, H, W, D)
    P, C = s.P, 18
    mg, og, sg, cg = np.zeros((P, 3), np.float32), np.zeros(P, np.float32), np.zeros((P, C), np.float32), np.zeros((P, 6), np.float32)
    v2p = np.zeros(H * W * D, np.int32)
    out_grad = _f32(out_grad)
    assert out_grad.shape == (s.N, C)
    if variant != "base":
        bin_grad, density_grad = _f32(bin_grad), _f32(density_grad)
    rc = s.lib.ref_splat_backward(s.handle, _p(out_grad), _p(bin_grad), _p(density_grad), _p(mg), _p(og), _p(sg), _p(cg),
                                  _p(v2p))
    s.close()
    if rc:
        raise RuntimeError("reference splat backward failed")
    return res, (mg, og, sg, cg), v2p


def splat_backward(variant, pts, points_int, means3D, means_int, opacity, semantics, radii, cov6, H, W, D, out_grad,
                   fwd=None, bin_grad=None, density_grad=None):
    """Signature of ``oracle.splat_backward`` (``fwd`` is ignored: the reference recomputes its own forward)."""
    return splat_forward_backward(variant, pts, points_int, means3D, means_int, opacity, semantics, radii, cov6,
                                  H, W, D, out_grad, bin_grad, density_grad)[1]


def _daf_dims(feat, ss, loc, w):
    B, cams, num_feat, C = feat.shape
    return B, cams, num_feat, C, ss.shape[0], loc.shape[1], w.shape[4]


def daf_forward(mc_ms_feat, spatial_shape, scale_start_index, sampling_location, weights):
    """The reference's ``deformable_aggregation`` launcher (…_cuda.cu:262-285)."""
    feat, loc, w = _f32(mc_ms_feat), _f32(sampling_location), _f32(weights)
    ss, st = _i32(spatial_shape), _i32(scale_start_index)
    B, cams, num_feat, C, L, pts, G = _daf_dims(feat, ss, loc, w)
    out = np.zeros((B, pts, C), np.float32)
    rc = _load("daf").ref_daf_forward(_p(feat), _p(ss), _p(st), _p(loc), _p(w), B, cams, num_feat, C, L, pts, G, _p(out))
    if rc:
        raise RuntimeError(f"reference deformable aggregation forward failed ({rc})")
    return out


def daf_backward(mc_ms_feat, spatial_shape, scale_start_index, sampling_location, weights, grad_output):
    """The reference's ``deformable_aggregation_grad`` launcher (…_cuda.cu:288-313).  Float atomics: the
    result is order-dependent in the last bits."""
    feat, loc, w, go = _f32(mc_ms_feat), _f32(sampling_location), _f32(weights), _f32(grad_output)
    ss, st = _i32(spatial_shape), _i32(scale_start_index)
    B, cams, num_feat, C, L, pts, G = _daf_dims(feat, ss, loc, w)
    gf, gl, gw = np.zeros_like(feat), np.zeros_like(loc), np.zeros_like(w)
    rc = _load("daf").ref_daf_backward(_p(feat), _p(ss), _p(st), _p(loc), _p(w), _p(go), B, cams, num_feat, C, L, pts, G,
                                       _p(gf), _p(gl), _p(gw))
    if rc:
        raise RuntimeError(f"reference deformable aggregation backward failed ({rc})")
    return gf, gl, gw
