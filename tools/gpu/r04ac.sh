#!/bin/bash
R=r04ac
OUT=gpurun_out/profiles_$R
mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_splat_mfma_gpu.py tests/test_splat_gpu.py tests/test_golden_gpu.py -m gpu -q -x --tb=short --timeout 120 2>&1 | tail -8 > $OUT/pytest_new_$R.log; cat $OUT/pytest_new_$R.log
bash tools/gpu/kernel_pair.sh > $OUT/kernel_pair_$R.txt 2>&1; grep "mean\|per step" $OUT/kernel_pair_$R.txt
