#!/bin/bash
# round 4, call B: A/B of the forward against the round-3 library on the same box; the new parity tests in full
R=r04b
OUT=gpurun_out/profiles_$R
mkdir -p $OUT; export TMPDIR=/tmp
for i in 1 2; do
  GF_LIB=$PWD/gaussianformer_amd/csrc/libgf_hip_r03.so timeout 300 python tools/mfma_probe.py nuscenes_gs25600_solid nuscenes_gs144000 2>&1 | grep "us per step" | sed 's/^/r03lib  /' | tee -a $OUT/ab_forward_$R.txt
  timeout 300 python tools/mfma_probe.py nuscenes_gs25600_solid nuscenes_gs144000 2>&1 | grep "us per step" | sed 's/^/current /' | tee -a $OUT/ab_forward_$R.txt
done
timeout 900 python -m pytest tests/test_splat_mfma_gpu.py tests/test_splat_gpu.py tests/test_slab_gpu.py -m gpu -q -x --tb=short 2>&1 | tail -80 > $OUT/pytest_new_$R.log; cat $OUT/pytest_new_$R.log
