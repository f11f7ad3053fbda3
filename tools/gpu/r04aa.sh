#!/bin/bash
R=r04aa
OUT=gpurun_out/profiles_$R
mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_splat_mfma_gpu.py tests/test_splat_gpu.py tests/test_golden_gpu.py tests/test_hot_path_chain.py tests/test_head.py tests/test_slab_gpu.py -m gpu -q --tb=short --timeout 200 2>&1 | tail -30 > $OUT/pytest_new_$R.log; cat $OUT/pytest_new_$R.log
rm -rf gpurun_out/kt_bwd; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_bwd -- python tools/prof_bwd.py nuscenes_gs25600_solid 20 640 > gpurun_out/kt_bwd.log 2>&1; cp $(find gpurun_out/kt_bwd -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_bwd_mfma_$R.csv; cut -d, -f1-4 $OUT/kernel_stats_bwd_mfma_$R.csv | head -8
