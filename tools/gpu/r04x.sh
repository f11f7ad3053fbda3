#!/bin/bash
R=r04x
OUT=gpurun_out/profiles_$R
mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_splat_mfma_gpu.py tests/test_splat_gpu.py tests/test_golden_gpu.py tests/test_hot_path_chain.py -m gpu -q --tb=short --timeout 200 2>&1 | tail -30 > $OUT/pytest_new_$R.log; cat $OUT/pytest_new_$R.log
