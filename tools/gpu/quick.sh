#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -30 > gpurun_out/pytest_gpu.log
cat gpurun_out/pytest_gpu.log
timeout 600 python tools/bench_ops.py 2>/dev/null | grep splat_backward
rm -rf gpurun_out/kt; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -- python tools/quick_time.py nuscenes_gs25600_solid > gpurun_out/kt.log 2>&1
cat $(find gpurun_out/kt -name "*kernel_stats.csv" | head -1)
