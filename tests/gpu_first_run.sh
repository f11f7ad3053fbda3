#!/bin/bash
# first GPU bring-up: tests + a quick timing of the splat forward
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -60 > gpurun_out/pytest_gpu.log
cat gpurun_out/pytest_gpu.log
timeout 300 python tools/quick_time.py > gpurun_out/quick_time.log 2>&1
cat gpurun_out/quick_time.log
