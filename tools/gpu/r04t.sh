#!/bin/bash
# pack kernel: parity tests + the frame with it
OUT=gpurun_out/profiles_r04t; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_prepare.py tests/test_ref_callers.py tests/test_head.py -m gpu -x -q --timeout 120 2>&1 | tail -15 > $OUT/pytest_pack.log; cat $OUT/pytest_pack.log
timeout 300 python tools/bench_frame.py --frames 20 --graph > $OUT/bench_frame_r04t.jsonl 2> gpurun_out/bench_frame.err; cut -c1-400 $OUT/bench_frame_r04t.jsonl; tail -3 gpurun_out/bench_frame.err
