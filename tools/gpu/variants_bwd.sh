#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python tools/variants_bwd.py "base:" "p2:-DGF_BWD_OCC_PROB=2" "p3d:-DGF_BWD_LG_STAGE=0" "p2d:-DGF_BWD_OCC_PROB=2 -DGF_BWD_LG_STAGE=0" "b5:-DGF_BWD_OCC_BASE=5" "b3:-DGF_BWD_OCC_BASE=3" > gpurun_out/variants_bwd.log 2>&1; cat gpurun_out/variants_bwd.log
