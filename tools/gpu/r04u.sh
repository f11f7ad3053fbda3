#!/bin/bash
OUT=gpurun_out/profiles_r04u; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_prepare.py -m gpu -x -q --timeout 120 2>&1 | tail -5
timeout 200 python tools/timeline_bwd.py nuscenes_gs25600_solid > $OUT/timeline_bwd.txt 2>&1; tail -12 $OUT/timeline_bwd.txt
