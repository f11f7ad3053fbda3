#!/bin/bash
R=${1:-r04h}
OUT=gpurun_out/profiles_$R
mkdir -p $OUT; export TMPDIR=/tmp
timeout 100 python tools/bwd_probe.py small > $OUT/bwd_probe_small_$R.txt 2>&1; echo "rc $?"; grep -c "finite True" $OUT/bwd_probe_small_$R.txt; grep "finite False\|worst" $OUT/bwd_probe_small_$R.txt | cut -c1-300 | head -5
timeout 100 python tools/timeline_bwd.py nuscenes_gs25600_solid > $OUT/timeline_bwd_$R.txt 2>&1; tail -13 $OUT/timeline_bwd_$R.txt
rm -rf gpurun_out/kt_bwd; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_bwd -- python tools/prof_bwd.py nuscenes_gs25600_solid 20 128 > gpurun_out/kt_bwd.log 2>&1; cp $(find gpurun_out/kt_bwd -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_bwd_mfma_$R.csv; cut -c1-150 $OUT/kernel_stats_bwd_mfma_$R.csv | head -6
