#!/bin/bash
# round 4, call G: after the hang fix (crowded chunks), the lean range loop, no scratch in the group loop
R=r04g
OUT=gpurun_out/profiles_$R
mkdir -p $OUT; export TMPDIR=/tmp
timeout 100 python tools/bwd_probe.py small > $OUT/bwd_probe_small_$R.txt 2>&1; echo "rc $?"; grep -c "finite True" $OUT/bwd_probe_small_$R.txt; grep "finite False\|worst" $OUT/bwd_probe_small_$R.txt | cut -c1-300 | head
timeout 150 python tools/bwd_probe.py full nuscenes_gs25600_solid > $OUT/bwd_probe_full_$R.txt 2>&1; echo "rc $?"; grep "vs oracle\|us per call\|worst\|False" $OUT/bwd_probe_full_$R.txt | cut -c1-300
timeout 100 python tools/timeline_bwd.py nuscenes_gs25600_solid > $OUT/timeline_bwd_$R.txt 2>&1; tail -13 $OUT/timeline_bwd_$R.txt
rm -rf gpurun_out/kt_bwd; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_bwd -- python tools/prof_bwd.py nuscenes_gs25600_solid 20 128 > gpurun_out/kt_bwd.log 2>&1; cp $(find gpurun_out/kt_bwd -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_bwd_mfma_$R.csv; cut -c1-150 $OUT/kernel_stats_bwd_mfma_$R.csv | head -8
GF_LIB=$PWD/gaussianformer_amd/csrc/libgf_hip_r03.so timeout 200 python tools/mfma_probe.py nuscenes_gs25600_solid nuscenes_gs144000 2>&1 | grep "us per step" | sed 's/^/r03lib  /' | tee -a $OUT/ab_forward_$R.txt
timeout 200 python tools/mfma_probe.py nuscenes_gs25600_solid nuscenes_gs144000 2>&1 | grep "us per step\|oracle/_ref" | sed 's/^/current /' | tee -a $OUT/ab_forward_$R.txt
timeout 400 python -m pytest tests/test_splat_mfma_gpu.py tests/test_splat_gpu.py -m gpu -q -x --tb=short --timeout 60 2>&1 | tail -30 > $OUT/pytest_$R.log; cat $OUT/pytest_$R.log
