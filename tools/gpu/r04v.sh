#!/bin/bash
# records reuse in the backward: tests, probe, forward A/B against the round-3 library
R=r04v
OUT=gpurun_out/profiles_$R
mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_splat_mfma_gpu.py tests/test_splat_gpu.py -m gpu -q -x --tb=short --timeout 120 2>&1 | tail -40 > $OUT/pytest_new_$R.log; cat $OUT/pytest_new_$R.log
timeout 200 python tools/bwd_probe.py full nuscenes_gs25600_solid > $OUT/bwd_probe_$R.txt 2>&1; grep "us per call\|vs oracle" $OUT/bwd_probe_$R.txt | cut -c1-220
for i in 1 2; do
  GF_LIB=$PWD/gaussianformer_amd/csrc/libgf_hip_r03.so timeout 300 python tools/mfma_probe.py nuscenes_gs25600_solid 2>&1 | grep "us per step" | sed 's/^/r03lib  /' | tee -a $OUT/ab_forward_$R.txt
  timeout 300 python tools/mfma_probe.py nuscenes_gs25600_solid 2>&1 | grep "us per step" | sed 's/^/current /' | tee -a $OUT/ab_forward_$R.txt
done
