"""Builds ``libgf_hip.so`` (the C-ABI library, ``include/gf_hip.h``) with hipcc for gfx950.

In-tree build: the ``.so`` lands next to the sources in ``gaussianformer_amd/csrc`` so it
travels with a repo snapshot to the GPU box.  hipcc cross-compiles without a GPU.
"""
import os
import shutil
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_NAME = "libgf_hip.so"
SOURCES = ["gf_api.hip", "splat_fwd.hip", "splat_bwd.hip", "splat_bwd_mfma.hip", "daf.hip", "gaussian_prepare.hip", "daf_prepare.hip", "daf_fused.hip", "head_labels.hip", "feature_format.hip", "subm_conv.hip", "key_points.hip"]
HEADERS = ["gf_common.hpp", "splat_fwd_pair.inc", "splat_fwd_solo.inc", os.path.join("..", "..", "include", "gf_hip.h")]
ARCH = "gfx950"
# -munsafe-fp-atomics only for the translation units that issue float atomics (hardware fp32 adds without a CAS loop); the
# others are built without it
FP_ATOMICS = {"daf.hip", "splat_bwd.hip", "splat_bwd_mfma.hip", "subm_conv.hip"}


def lib_path():
    return os.path.join(CSRC, LIB_NAME)


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _stale():
    lib = lib_path()
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra_flags=(), lib_name=None):
    """Compile every HIP source into one shared library.  Returns its path.
    ``extra_flags`` / ``lib_name`` build a development variant (e.g. ``-DGF_TIMELINE=1``)
    next to the product library without touching it."""
    if lib_name is None and not force and not _stale():
        return lib_path()
    out_lib = lib_path() if lib_name is None else os.path.join(CSRC, lib_name)
    tag = "" if lib_name is None else "." + os.path.splitext(lib_name)[0]
    objs = []
    # -Os, not -O3 (round 6, measured: tools/gpu/os_compare.sh): the persistent splat kernels are instruction streams, and -O3's
    # unrolling and peeling make them 6 % longer (the wave kernel: 17 048 -> 15 984 bytes) for nothing -- forward 41.0 -> 40.0 us per
    # step, 83.9 -> 81.2 us at P = 144 000, matrix-core backward 75.2 -> 73.7 us; the gather-bound kernels (deformable aggregation,
    # sparse convolution) are indifferent; same bits everywhere
    common = [_hipcc(), f"--offload-arch={ARCH}", "-Os", "-std=c++17", "-fPIC",
              "-Wall", "-Wno-unused-function", "-Wno-inline-asm", *extra_flags]
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", tag + ".o"))
        objs.append(obj)
        cmd = common + (["-munsafe-fp-atomics"] if src in FP_ATOMICS else []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out, file=sys.stderr)
    link = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", out_lib] + objs
    r = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc link failed:\n{r.stdout}")
    return out_lib


DEV_LIB_NAME = "libgf_hip_dev.so"


def build_dev(verbose=False):
    """The development build tools/ use (``-DGF_DEV=1``: the measured-and-not-kept kernels of earlier rounds and the ``dev.*``
    options of ``gf_set_option``), next to the product library and never in its place.  Select it with ``GF_LIB=<path>``."""
    return build(force=True, verbose=verbose, extra_flags=("-DGF_DEV=1",), lib_name=DEV_LIB_NAME)


if __name__ == "__main__":
    if "--dev" in sys.argv:
        print(build_dev(verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
