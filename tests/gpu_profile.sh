#!/bin/bash
# tests + rocprofv3 kernel trace of the quick timing script
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -30 > gpurun_out/pytest_gpu.log
cat gpurun_out/pytest_gpu.log
rm -rf gpurun_out/prof1
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof1 -o qt -- python tools/quick_time.py nuscenes_gs25600_solid > gpurun_out/prof1.log 2>&1
tail -8 gpurun_out/prof1.log
find gpurun_out/prof1 -name "*stats*" | head; 
f=$(find gpurun_out/prof1 -name "*kernel_stats.csv" | head -1); cat "$f" | head -20
