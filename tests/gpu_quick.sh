#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -30 > gpurun_out/pytest_gpu.log
cat gpurun_out/pytest_gpu.log
timeout 300 python tools/timeline.py > gpurun_out/timeline.log 2>&1; tail -18 gpurun_out/timeline.log
rm -rf gpurun_out/kt; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -- python tools/prof_fwd.py nuscenes_gs25600_solid 30 0 > gpurun_out/kt.log 2>&1
cat $(find gpurun_out/kt -name "*kernel_stats.csv" | head -1)
timeout 300 python tools/quick_time.py > gpurun_out/quick_time.log 2>&1; cat gpurun_out/quick_time.log
