#!/bin/bash
# round 4, call D: kernel trace of the matrix-core backward; forward A/B; forward tests
R=r04d
OUT=gpurun_out/profiles_$R
mkdir -p $OUT; export TMPDIR=/tmp
rm -rf gpurun_out/kt_bwd; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_bwd -- python tools/prof_bwd.py nuscenes_gs25600_solid 20 128 > gpurun_out/kt_bwd.log 2>&1; cp $(find gpurun_out/kt_bwd -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_bwd_mfma_$R.csv; cut -c1-150 $OUT/kernel_stats_bwd_mfma_$R.csv | head -12
GF_LIB=$PWD/gaussianformer_amd/csrc/libgf_hip_r03.so timeout 300 python tools/mfma_probe.py nuscenes_gs25600_solid nuscenes_gs144000 2>&1 | grep "us per step" | sed 's/^/r03lib  /' | tee -a $OUT/ab_forward_$R.txt
timeout 300 python tools/mfma_probe.py nuscenes_gs25600_solid nuscenes_gs144000 2>&1 | grep "us per step\|oracle/_ref" | sed 's/^/current /' | tee -a $OUT/ab_forward_$R.txt
timeout 900 python -m pytest tests/test_splat_mfma_gpu.py -m gpu -q --tb=short 2>&1 | tail -40 > $OUT/pytest_mfma_$R.log; cat $OUT/pytest_mfma_$R.log
