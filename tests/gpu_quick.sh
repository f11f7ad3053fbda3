#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -30 > gpurun_out/pytest_gpu.log
cat gpurun_out/pytest_gpu.log
python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/bench.json 2> gpurun_out/bench.err; cat gpurun_out/bench.json
timeout 600 python tools/bench_ops.py > gpurun_out/bench_ops.jsonl 2> gpurun_out/bench_ops.err; cat gpurun_out/bench_ops.jsonl; tail -3 gpurun_out/bench_ops.err
