#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/pytest_gpu.log; cat gpurun_out/pytest_gpu.log
timeout 900 python tools/variants.py "base:" "fixedrec:-DGF_DIAG=1" "vmemrec:-DGF_DIAG=2 -DGF_OCC=4" "vmemrec6:-DGF_DIAG=2 -DGF_OCC=6" > gpurun_out/variants.log 2>&1; cat gpurun_out/variants.log
timeout 300 python tools/quick_time.py nuscenes_gs25600_solid > gpurun_out/quick_time.log 2>&1; cat gpurun_out/quick_time.log
