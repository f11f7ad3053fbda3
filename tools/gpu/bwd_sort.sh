#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python tools/bwd_sort_experiment.py "noxcd:-DGF_BWD_XCD=0" "xcd1:-DGF_BWD_PHASES=1" "xcd2:-DGF_BWD_PHASES=2" "xcd4:-DGF_BWD_PHASES=4" > gpurun_out/bwd_sort.log 2>&1; cat gpurun_out/bwd_sort.log
