"""Development build only (GF_LIB=gaussianformer_amd/csrc/libgf_hip_dev.so, built by `python -m gaussianformer_amd.build --dev`):
round 5's three other organisations of the matrix-core forward -- pair, solo (two / three waves per SIMD), fused records pass --
against the CPU oracle (small shapes) and the wave kernel (full shape), and bit-reproducible.  They fold the opacity into the
exponent, so they agree with the wave kernel to rounding, not bit for bit.  Exit code 0 = all good.  (tests/ runs this in a
subprocess when the development build exists.)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import oracle
from gaussianformer_amd import _lib
from gaussianformer_amd.synthetic import make_splat_inputs
from util import assert_logits_close, hip_splat_forward, prep

assert _lib.is_development_build(), "not a development build: set GF_LIB to libgf_hip_dev.so"
gpu = torch.device("cuda:0")
CASES = [
    ("pair", {"dev.splat_pair": 1}, 0, _lib.GF_PATH_MATRIX_CORE_PAIR),
    ("solo", {"dev.splat_solo": 1}, 0, _lib.GF_PATH_MATRIX_CORE_SOLO),
    ("solo, three waves per SIMD", {"dev.splat_solo": 1, "dev.splat_solo_waves": 3}, 0, _lib.GF_PATH_MATRIX_CORE_SOLO),
    ("fused records pass", {"dev.splat_fused": 1}, 1, _lib.GF_PATH_MATRIX_CORE_SOLO),
]
SHAPES = [dict(), dict(P=300, H=16, W=16, D=8), dict(P=2000, H=40, W=40, D=16), dict(P=777, H=20, W=36, D=12),
          dict(P=64, H=8, W=8, D=4), dict(P=1, H=8, W=8, D=8), dict(P=3000, H=30, W=50, D=16)]


def set_all(opts, on):
    for k, v in opts.items():
        _lib.set_option(k, v if on else 0)


for name, opts, assume_dense, path in CASES:
    flags = _lib.GF_MFMA_SPLAT | (_lib.GF_PTS_ASSUME_DENSE if assume_dense else 0)
    for seed, kw in enumerate(SHAPES):
        si = make_splat_inputs("nuscenes_gs25600_solid", seed=seed + 1, **kw)
        pi, mi, radii, cov6 = prep(si)
        set_all(opts, True)
        got, _, state, _ = hip_splat_forward(gpu, si, pi, mi, radii, cov6, flags=flags)
        words = state[:12].view(torch.int32).cpu().tolist()
        assert words[1] == path, (name, kw, words[:3])
        assert np.isfinite(got["logits"]).all()
        again, *_ = hip_splat_forward(gpu, si, pi, mi, radii, cov6, flags=flags)
        assert np.array_equal(got["logits"], again["logits"]), (name, kw)
        set_all(opts, False)
        if kw:
            ref = oracle.splat_forward(si.variant, si.pts, pi, si.means3D, mi, si.opacities, si.semantics, radii, cov6, si.H, si.W, si.D)["logits"]
            assert_logits_close(got["logits"], ref, tol=1e-4)
        else:   # the full shape: the wave kernel (itself held to the oracle by tests/) stands in for the CPU oracle
            ref = hip_splat_forward(gpu, si, pi, mi, radii, cov6, flags=flags)[0]["logits"]
            assert_logits_close(got["logits"], ref, tol=5e-5)
    print(f"{name}: ok", flush=True)
print("development kernels OK")
