#!/bin/bash
# round 4, call R: full GPU suite after the label epilogue in the wave kernel; label-only timing
R=r04r
OUT=gpurun_out/profiles_$R
mkdir -p $OUT; export TMPDIR=/tmp
timeout 800 python -m pytest tests -m gpu -q -x --tb=short --timeout 300 2>&1 | tail -15 > $OUT/pytest_gpu_$R.log; cat $OUT/pytest_gpu_$R.log
python - <<'PY' 2>&1 | tee gpurun_out/profiles_r04r/labels_probe_r04r.txt
import sys, time, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from gaussianformer_amd import _lib
from gaussianformer_amd.local_aggregate import splat_forward, splat_forward_labels
from gaussianformer_amd.head import occupancy_labels
from gaussianformer_amd.synthetic import make_splat_inputs
from util import prep, to_dev
dev = torch.device("cuda:0")
si = make_splat_inputs("nuscenes_gs25600_solid", seed=0)
pi, mi, radii, cov6 = prep(si)
t = to_dev(dev, si.pts, pi, si.means3D, mi, si.opacities, si.semantics, radii, cov6)
def timed(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for name, fl in (("matrix cores (default)", 0), ("exact fp32", _lib.GF_EXACT_FP32)):
    a = timed(lambda: splat_forward_labels(0, *t, si.H, si.W, si.D, flags=fl))
    b = timed(lambda: splat_forward_labels(0, *t, si.H, si.W, si.D, keep_logits=True, flags=fl))
    c = timed(lambda: occupancy_labels(splat_forward(0, *t, si.H, si.W, si.D, flags=fl)[0]))
    print(f"gs25600 {name}: labels only {a:.1f} us, labels + logits {b:.1f} us, forward + gf_head_labels {c:.1f} us (module-level calls)")
PY
