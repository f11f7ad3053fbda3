#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -30 > gpurun_out/pytest_gpu.log
cat gpurun_out/pytest_gpu.log
python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/bench.json 2> gpurun_out/bench.err; cat gpurun_out/bench.json
timeout 300 python tools/timeline.py > gpurun_out/timeline.log 2>&1; tail -16 gpurun_out/timeline.log
timeout 300 python tools/quick_time.py > gpurun_out/quick_time.log 2>&1; cat gpurun_out/quick_time.log
