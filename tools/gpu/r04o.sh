#!/bin/bash
# round 4, call O: bench (N = 1), the 2-rank shared-GPU bench test (frame_sharded), forward A/B
R=r04o
OUT=gpurun_out/profiles_$R
mkdir -p $OUT; export TMPDIR=/tmp
GF_LIB=$PWD/gaussianformer_amd/csrc/libgf_hip_r03.so timeout 200 python tools/mfma_probe.py nuscenes_gs25600_solid nuscenes_gs144000 2>&1 | grep "us per step" | sed 's/^/r03lib  /' | tee -a $OUT/ab_forward_$R.txt
timeout 200 python tools/mfma_probe.py nuscenes_gs25600_solid nuscenes_gs144000 2>&1 | grep "us per step" | sed 's/^/current /' | tee -a $OUT/ab_forward_$R.txt
timeout 600 python -m pytest tests/test_bench_gpu.py tests/test_splat_mfma_gpu.py -m gpu -q -x --tb=short --timeout 400 2>&1 | tail -25 > $OUT/pytest_$R.log; cat $OUT/pytest_$R.log
timeout 400 python bench.py --steps 200 --warmup 20 > $OUT/bench_$R.json 2> $OUT/bench_$R.err; cut -c1-1200 $OUT/bench_$R.json; tail -3 $OUT/bench_$R.err
