"""ctypes binding of ``libgf_hip.so`` (C ABI declared in ``include/gf_hip.h``).

There is NO fallback: if the library is missing or a call fails, a ``RuntimeError`` is
raised.  Tensors cross the boundary as raw device pointers + sizes; the stream is torch's
current HIP stream.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GF_LIB selects a development variant built by build.build(lib_name=...) (tools/ only)
LIB_PATH = os.environ.get("GF_LIB") or os.path.join(_HERE, "csrc", "libgf_hip.so")

GF_ABI_VERSION = 4
GF_SPLAT_BASE, GF_SPLAT_PROB = 0, 1
GF_NUM_CHANNELS = 18
GF_LABELS_ARGMAX, GF_LABELS_PROB_THRESHOLD, GF_LABELS_PROB_GEOSEM = 0, 1, 2
GF_RADII_SCALAR, GF_RADII_SCALAR_CLAMPED, GF_RADII_PER_AXIS = 0, 1, 2
GF_PREPARE_MEAN_OUT_OF_GRID, GF_PREPARE_RADIUS_BELOW_ONE = 1, 2
GF_PTS_AUTO, GF_PTS_ASSUME_DENSE, GF_PTS_GENERAL, GF_FAST_EXP, GF_LIBM_EXP, GF_COMP_EXP = 0, 1, 2, 4, 8, 16
GF_PROB_NUMERATOR = 32
GF_PROB_EXACT_DET = 64
GF_MFMA_SPLAT = 128
GF_EXACT_FP32 = 256
GF_RECORDS_VALID = 512
GF_PREPARE_BACKWARD = 1024
GF_WORKSPACE_ZEROED = 2048
GF_PATH_EXACT_TILE, GF_PATH_MATRIX_CORE, GF_PATH_ARBITRARY, GF_PATH_MATRIX_CORE_WAVE, GF_PATH_MATRIX_CORE_PAIR, GF_PATH_MATRIX_CORE_SOLO = 0, 1, 2, 3, 4, 5
GF_PATHS_MATRIX_CORE = (GF_PATH_MATRIX_CORE, GF_PATH_MATRIX_CORE_WAVE, GF_PATH_MATRIX_CORE_PAIR, GF_PATH_MATRIX_CORE_SOLO)

_vp, _i, _sz, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_float

# name -> (restype, argtypes); must list every symbol include/gf_hip.h declares
SIGNATURES = {
    "gf_abi_version": (_i, []),
    "gf_last_error": (ctypes.c_char_p, []),
    "gf_set_option": (_i, [ctypes.c_char_p, _i]),
    "gf_get_option": (_i, [ctypes.c_char_p, _vp]),
    "gf_is_development_build": (_i, []),
    "gf_splat_workspace_bytes": (_sz, [_i] * 5),
    "gf_splat_state_bytes": (_sz, []),
    "gf_splat_forward": (_i, [_i] * 9 + [_vp] * 13 + [_vp, _sz, _vp]),
    "gf_splat_forward_labels": (_i, [_i] * 9 + [_vp] * 12 + [_i, _f, _i, _vp] + [_vp, _vp, _sz, _vp]),
    "gf_splat_backward": (_i, [_i] * 9 + [_vp] * 20 + [_vp, _sz, _vp]),
    "gf_splat_box_volumes": (_i, [_i] * 5 + [_vp] * 5),
    "gf_daf_forward": (_i, [_i] * 7 + [_vp] * 7),
    "gf_daf_forward_pinned": (_i, [_i] * 7 + [_vp] * 7),
    "gf_daf_backward": (_i, [_i] * 7 + [_vp] * 10),
    "gf_daf_backward_workspace_bytes": (_sz, [_i] * 7),
    "gf_daf_backward_sorted": (_i, [_i] * 7 + [_vp] * 9 + [_vp, _sz, _vp]),
    "gf_subm_voxelize": (_i, [ctypes.c_longlong, _i, _i, _i] + [_vp] * 6 + [_vp]),
    "gf_subm_tables_bytes": (_sz, [_i] * 6),
    "gf_subm_rulebook_count": (_i, [_i] * 6 + [_vp, _vp, _sz, _vp]),
    "gf_subm_rulebook_fill": (_i, [_i] * 6 + [_vp] * 4 + [_vp]),
    "gf_subm_rulebook_build": (_i, [_i] * 6 + [_vp, _vp, _sz, _vp, _vp, ctypes.c_longlong, _vp]),
    "gf_subm_rulebook_count_range": (_i, [_i] * 8 + [_vp, _vp, _sz, _vp]),
    "gf_subm_rulebook_fill_range": (_i, [_i] * 8 + [_vp] * 4 + [_vp]),
    "gf_subm_rulebook_build_range": (_i, [_i] * 8 + [_vp, _vp, _sz, _vp, _vp, ctypes.c_longlong, _vp]),
    "gf_subm_conv_apply": (_i, [_i] * 8 + [ctypes.c_longlong] + [_vp] * 6 + [_vp]),
    "gf_subm_apply_scratch_bytes": (_sz, [_i, _i]),
    "gf_subm_conv_apply_scratch": (_i, [_i] * 8 + [ctypes.c_longlong] + [_vp] * 6 + [_vp, _sz, _vp]),
    "gf_subm_conv_weight_grad": (_i, [_i] * 8 + [ctypes.c_longlong] + [_vp] * 6 + [_vp]),
    "gf_feature_maps_format": (_i, [_i, _i, _i, _vp, _vp, _vp, _i, _vp]),
    "gf_head_labels": (_i, [ctypes.c_longlong, _i, _i, _vp, _vp, _f, _i, _vp, _vp]),
    "gf_daf_fused_forward": (_i, [_i] * 8 + [_vp] * 11),
    "gf_daf_prepare": (_i, [_i] * 6 + [_vp] * 7 + [_vp]),
    "gf_daf_prepare_backward": (_i, [_i] * 6 + [_vp] * 8 + [_vp]),
    "gf_gaussian_prepare": (_i, [_i] * 4 + [_vp, _f, _f, _i, _i] + [_vp] * 8 + [_vp]),
    "gf_gaussian_prepare_backward": (_i, [_i] * 2 + [_vp] * 5 + [_vp]),
    "gf_gaussian_pack": (_i, [_i] * 7 + [_vp] * 14 + [_vp]),
    "gf_key_points": (_i, [_i] * 4 + [_vp] * 4 + [_f] * 3 + [_i, _vp, _vp]),
    "gf_key_points_backward": (_i, [_i] * 4 + [_vp] * 4 + [_f] * 3 + [_i] + [_vp] * 3 + [_vp]),
    "gf_profile_enable": (_i, [_i]),
    "gf_profile_stride": (_i, [_i]),
    "gf_profile_read": (_i, [_vp, _i]),
}

_lib = None


def load():
    """Load the library (once).  Raises RuntimeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension has not been built. "
            "Run `python -m gaussianformer_amd.build` (needs hipcc; cross-compiles gfx950 without a GPU). "
            "There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.gf_abi_version() != GF_ABI_VERSION:
        raise RuntimeError(f"libgf_hip.so ABI {lib.gf_abi_version()} != expected {GF_ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().gf_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


def set_option(name, value):
    """``gf_set_option``: a process-wide library option (include/gf_hip.h); explicit calls -- the library never reads the
    environment.  Returns the previous value.  ``dev.*`` names exist in the development build only (tools/)."""
    lib = load()
    old = ctypes.c_int(0)
    check(lib.gf_get_option(name.encode(), ctypes.addressof(old)), f"gf_get_option({name})")
    check(lib.gf_set_option(name.encode(), int(value)), f"gf_set_option({name})")
    return old.value


def get_option(name):
    lib = load()
    v = ctypes.c_int(0)
    check(lib.gf_get_option(name.encode(), ctypes.addressof(v)), f"gf_get_option({name})")
    return v.value


class option:
    """``with _lib.option("splat.mfma_tile_kernel", 1): ...`` -- sets a library option for the block and restores it."""

    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        self.old = set_option(self.name, self.value)
        return self

    def __exit__(self, *exc):
        set_option(self.name, self.old)
        return False


def is_development_build():
    return bool(load().gf_is_development_build())


def ptr(t):
    """Device pointer of a tensor (``None`` -> NULL)."""
    return None if t is None else t.data_ptr()


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "gaussianformer_amd ops run on MI355X only (HIP kernels); got a CPU tensor. "
                "There is no CPU fallback -- move the inputs to the GPU.")


def current_stream(device):
    import torch
    return torch.cuda.current_stream(device).cuda_stream
