#!/bin/bash
R=r04q
OUT=gpurun_out/profiles_$R
mkdir -p $OUT; export TMPDIR=/tmp
for m in 0 1; do
rm -rf gpurun_out/kt_daf3; GF_DAF_MORTON=$m timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_daf3 -- python tools/prof_daf2.py projected 5 both > gpurun_out/kt_daf3.log 2>&1; cp $(find gpurun_out/kt_daf3 -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_daf_projected_morton${m}_$R.csv; echo "== morton $m"; cut -c1-120 $OUT/kernel_stats_daf_projected_morton${m}_$R.csv | grep "gf_daf" | head -8
done
