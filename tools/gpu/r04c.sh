#!/bin/bash
# round 4, call C: first run of the matrix-core backward (parity vs the exact kernels), forward A/B, forward tests
R=r04c
OUT=gpurun_out/profiles_$R
mkdir -p $OUT; export TMPDIR=/tmp
timeout 120 python tools/bwd_probe.py small > $OUT/bwd_probe_small_$R.txt 2>&1; echo "rc $?"; cut -c1-700 $OUT/bwd_probe_small_$R.txt | tail -70
timeout 200 python tools/bwd_probe.py full nuscenes_gs25600_solid > $OUT/bwd_probe_full_$R.txt 2>&1; echo "rc $?"; cut -c1-400 $OUT/bwd_probe_full_$R.txt | tail -20
GF_LIB=$PWD/gaussianformer_amd/csrc/libgf_hip_r03.so timeout 300 python tools/mfma_probe.py nuscenes_gs25600_solid nuscenes_gs144000 2>&1 | grep "us per step" | sed 's/^/r03lib  /' | tee -a $OUT/ab_forward_$R.txt
timeout 300 python tools/mfma_probe.py nuscenes_gs25600_solid nuscenes_gs144000 2>&1 | grep "us per step\|oracle/_ref" | sed 's/^/current /' | tee -a $OUT/ab_forward_$R.txt
timeout 900 python -m pytest tests/test_splat_mfma_gpu.py -m gpu -q --tb=short 2>&1 | tail -40 > $OUT/pytest_mfma_$R.log; cat $OUT/pytest_mfma_$R.log
