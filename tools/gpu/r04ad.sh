#!/bin/bash
R=r04ad
OUT=gpurun_out/profiles_$R
mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_subm_conv.py tests/test_head.py tests/test_prepare.py tests/test_daf_prepare.py tests/test_ref_callers.py tests/test_hot_path_chain.py -m gpu -q -x --tb=short --timeout 200 2>&1 | tail -8 > $OUT/pytest_new_$R.log; cat $OUT/pytest_new_$R.log
timeout 300 python tools/bench_ops.py 2>/dev/null | grep "rulebook\|head_labels\|gaussian_prepare\|deformable_prepare (fused" | cut -c1-160
timeout 300 python tools/bench_frame.py --frames 20 --graph 2>/dev/null | cut -c1-160
