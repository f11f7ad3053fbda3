#!/bin/bash
# round 4, call A: the new parity tests, the bench line, DAF / splat-backward counters
R=r04a
OUT=gpurun_out/profiles_$R
mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_splat_mfma_gpu.py tests/test_subm_conv.py tests/test_splat_gpu.py -m gpu -q 2>&1 | tail -40 > $OUT/pytest_new_$R.log; cat $OUT/pytest_new_$R.log
timeout 600 python bench.py --steps 200 --warmup 20 > $OUT/bench_$R.json 2> $OUT/bench_$R.err; cut -c1-3000 $OUT/bench_$R.json; tail -3 $OUT/bench_$R.err
timeout 900 bash tools/gpu/pmc_daf.sh $R > $OUT/pmc_daf_$R.log 2>&1; tail -5 $OUT/pmc_daf_$R.log
