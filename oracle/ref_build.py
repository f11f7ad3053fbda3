"""Build recipe for ``oracle/_ref``: the REFERENCE's own CUDA kernels compiled with hipcc for gfx950.

TEST INFRASTRUCTURE.  What this does, and what it does not:

* The reference's kernel sources are read **where they lie** under ``/root/reference``
  (``model/head/localagg{,_prob,_prob_fast}/src/{aggregator_impl,forward,backward}.cu`` and
  ``model/encoder/gaussian_encoder/ops/src/deformable_aggregation_cuda.cu``).  They are never copied
  into this repository; the only outputs are the shared libraries ``oracle/_ref/libref_*.so``
  (git-ignored, shipped to the GPU box by gpurun like every other built ``.so``).
* hipcc (clang) does not accept the reference's spaced launch chevrons ``<< <`` / ``>> >``
  (forward.cu:98,120; backward.cu:122,145; aggregator_impl.cu:205,230).  Each translation unit is
  therefore passed through a two-substitution text filter into a scratch directory under ``$TMPDIR``
  (deleted after the build) and compiled from there with ``-I <reference src dir>``; nothing else in
  the sources is altered (the filter asserts it changes only lines holding a kernel launch).
* CUDA-toolkit / torch headers the sources include but do not need (``device_launch_parameters.h``,
  ``cooperative_groups/reduce.h``, ``THC/THCAtomics.cuh``, ``torch/extension.h``, ``ATen/…``) and the
  CUDA runtime names (``cudaMemcpy`` …, ``cub::`` → hipCUB) are provided by the include-path stand-ins
  in ``oracle/ref_shims/``.
* ``oracle/ref_wrap_splat.hip`` / ``ref_wrap_daf.hip`` are the C entry points (the role of the
  reference's torch bindings): no arithmetic of their own.
* The reference's own build system (setup.py / CMake, nvcc flags) is not run.
* Compile flags: ``-O3 -ffp-contract=on`` is hipcc's default, nvcc's default is ``--fmad=true`` — both
  contract ``a*b+c`` within a statement, so the floating-point evaluation is the one the reference gets
  from its own ``setup.py`` (no fast-math flag there: model/head/localagg/setup.py:32).

Usage: ``python -m oracle.ref_build`` (from the repo root).  Needs ``/root/reference``; on the GPU box
only the prebuilt libraries are used.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
SHIMS = os.path.join(HERE, "ref_shims")
REFERENCE = os.environ.get("GF_REFERENCE_ROOT", "/root/reference")
ARCH = "gfx950"

SPLAT_VARIANTS = {
    # library name -> (reference directory, wrapper defines)
    "libref_localagg.so": ("model/head/localagg", ["-DREF_PROB=0"]),
    "libref_localagg_prob.so": ("model/head/localagg_prob", ["-DREF_PROB=1"]),
    "libref_localagg_prob_fast.so": ("model/head/localagg_prob_fast", ["-DREF_PROB=1", "-DREF_RADII_PER_AXIS=1"]),
}
SPLAT_UNITS = ["aggregator_impl.cu", "forward.cu", "backward.cu"]
DAF_DIR = "model/encoder/gaussian_encoder/ops/src"
DAF_UNIT = "deformable_aggregation_cuda.cu"

_LAUNCH = re.compile(r"<<\s+<|>>\s+>")


def available():
    return os.path.isdir(os.path.join(REFERENCE, "model", "head", "localagg", "src"))


def libraries():
    return [os.path.join(OUT, n) for n in list(SPLAT_VARIANTS) + ["libref_daf.so"]]


def built():
    return all(os.path.exists(p) for p in libraries())


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _filter_unit(src, dst):
    """Write ``src`` to ``dst`` with the spaced chevrons closed up.  Only lines that open or close a
    kernel launch may change."""
    changed = 0
    with open(src) as f, open(dst, "w") as g:
        for line in f:
            new = line.replace("<< <", "<<<").replace(">> >", ">>>")
            if new != line:
                changed += 1
                assert ("<<<" in new) or (">>>" in new and "(" in new), (src, line)
            g.write(new)
    return changed


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


def _newest_mtime(paths):
    return max(os.path.getmtime(p) for p in paths)


def build(force=False, verbose=False):
    """Compile the four reference libraries.  Returns the list of paths; raises if the reference tree
    is absent (callers check ``available()`` first)."""
    if not available():
        raise RuntimeError(f"reference tree not found at {REFERENCE}")
    os.makedirs(OUT, exist_ok=True)
    own = [os.path.join(HERE, "ref_wrap_splat.hip"), os.path.join(HERE, "ref_wrap_daf.hip"), os.path.abspath(__file__)]
    own += [os.path.join(dp, f) for dp, _, fs in os.walk(SHIMS) for f in fs]
    common = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-w", "-Wno-c++11-narrowing",
              "-I", SHIMS]
    scratch = tempfile.mkdtemp(prefix="gf_ref_build_")
    try:
        for lib, (rel, defines) in SPLAT_VARIANTS.items():
            src_dir = os.path.join(REFERENCE, rel, "src")
            units = [os.path.join(src_dir, u) for u in SPLAT_UNITS]
            target = os.path.join(OUT, lib)
            if not force and os.path.exists(target) and os.path.getmtime(target) > _newest_mtime(units + own):
                continue
            work = os.path.join(scratch, lib)
            os.makedirs(work)
            objs = []
            procs = []
            for u, path in zip(SPLAT_UNITS, units):
                filtered = os.path.join(work, u.replace(".cu", ".hip"))
                n = _filter_unit(path, filtered)
                assert n >= 2, (path, n)
                obj = os.path.join(work, u + ".o")
                objs.append(obj)
                procs.append(subprocess.Popen(common + ["-I", src_dir, "-c", filtered, "-o", obj],
                                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
            wobj = os.path.join(work, "wrap.o")
            objs.append(wobj)
            procs.append(subprocess.Popen(common + defines + ["-I", src_dir, "-c", os.path.join(HERE, "ref_wrap_splat.hip"),
                                                              "-o", wobj],
                                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
            for p in procs:
                out, _ = p.communicate()
                if p.returncode != 0:
                    raise RuntimeError(f"hipcc failed while building {lib}:\n{out}")
            _run([_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", target] + objs)
            if verbose:
                print("built", target, file=sys.stderr)
        # deformable aggregation
        target = os.path.join(OUT, "libref_daf.so")
        unit = os.path.join(REFERENCE, DAF_DIR, DAF_UNIT)
        if force or not os.path.exists(target) or os.path.getmtime(target) <= _newest_mtime([unit] + own):
            work = os.path.join(scratch, "daf")
            os.makedirs(work)
            filtered = os.path.join(work, "deformable_aggregation_cuda.hip")
            shutil.copyfile(unit, filtered)  # launches are already written <<<...>>>; compiled as is
            o1, o2 = os.path.join(work, "k.o"), os.path.join(work, "w.o")
            _run(common + ["-c", filtered, "-o", o1])
            _run(common + ["-c", os.path.join(HERE, "ref_wrap_daf.hip"), "-o", o2])
            _run([_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", target, o1, o2])
            if verbose:
                print("built", target, file=sys.stderr)
    finally:
        shutil.rmtree(scratch, ignore_errors=True)
    return libraries()


if __name__ == "__main__":
    for p in build(force="--force" in sys.argv, verbose=True):
        print(p)
