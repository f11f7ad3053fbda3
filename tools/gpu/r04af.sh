#!/bin/bash
R=r04af
OUT=gpurun_out/profiles_$R
mkdir -p $OUT; export TMPDIR=/tmp
timeout 200 python tools/bwd_probe.py small > $OUT/bwd_probe_small_$R.txt 2>&1; grep -c "finite True" $OUT/bwd_probe_small_$R.txt; grep "finite False\|e-0[0-3] \|e+0" $OUT/bwd_probe_small_$R.txt | head -20
timeout 600 python -m pytest tests/test_splat_mfma_gpu.py tests/test_splat_gpu.py -m gpu -q -x --tb=short --timeout 120 2>&1 | tail -8 > $OUT/pytest_new_$R.log; cat $OUT/pytest_new_$R.log
timeout 200 python tools/bwd_probe.py full nuscenes_gs25600_solid > $OUT/bwd_probe_$R.txt 2>&1; grep "us per call\|vs oracle" $OUT/bwd_probe_$R.txt | cut -c1-220
GF_BWD_NO_LISTS=1 timeout 200 python tools/bwd_probe.py full nuscenes_gs25600_solid 2>&1 | grep "GF_RECORDS_VALID" | sed 's/^/no lists: /' | cut -c1-200
rm -rf gpurun_out/kt_bwd; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_bwd -- python tools/prof_bwd.py nuscenes_gs25600_solid 20 640 > gpurun_out/kt_bwd.log 2>&1; cp $(find gpurun_out/kt_bwd -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_bwd_mfma_$R.csv; cut -d, -f1-4 $OUT/kernel_stats_bwd_mfma_$R.csv | head -8
timeout 200 python tools/timeline_bwd.py nuscenes_gs25600_solid > $OUT/timeline_bwd_$R.txt 2>&1; tail -12 $OUT/timeline_bwd_$R.txt
for i in 1 2; do timeout 300 python tools/mfma_probe.py nuscenes_gs25600_solid 2>&1 | grep "us per step" | sed "s/^/current /" | tee -a $OUT/ab_forward_$R.txt; done
