#!/bin/bash
R=r04ae
OUT=gpurun_out/profiles_$R
mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_subm_conv.py tests/test_hot_path_chain.py -m gpu -q -x --tb=short --timeout 200 2>&1 | tail -4 > $OUT/pytest_new_$R.log; cat $OUT/pytest_new_$R.log
timeout 300 python tools/bench_ops.py 2>/dev/null | grep "subm_conv" | cut -c1-200
timeout 300 python tools/bench_frame.py --frames 20 --graph > $OUT/bench_frame_$R.jsonl 2>/dev/null; grep -o '"config": "[a-z0-9_]*"\|"frames_per_s[a-z_]*": [0-9.]*' $OUT/bench_frame_$R.jsonl
timeout 300 python tools/bench_step.py 2>/dev/null | cut -c1-260
