#!/bin/bash
# round-3 GPU check A: full GPU test suite, both render kernels timed (with and without longest-first order), timelines, bench line
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_bench_gpu.py 2>&1 | tail -30 > gpurun_out/pytest_a.log; cat gpurun_out/pytest_a.log
timeout 300 python tools/mfma_probe.py > gpurun_out/probe_lpt.txt 2>&1; grep -v amdgpu.ids gpurun_out/probe_lpt.txt
GF_NO_LPT=1 timeout 300 python tools/mfma_probe.py > gpurun_out/probe_nolpt.txt 2>&1; grep "us per step" gpurun_out/probe_nolpt.txt
timeout 300 python tools/timeline.py > gpurun_out/timeline_mfma.txt 2>&1; tail -18 gpurun_out/timeline_mfma.txt
timeout 300 python tools/timeline.py nuscenes_gs144000 > gpurun_out/timeline_mfma_144.txt 2>&1; tail -18 gpurun_out/timeline_mfma_144.txt
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err; tail -c 6000 gpurun_out/bench_a.json; tail -5 gpurun_out/bench_a.err
