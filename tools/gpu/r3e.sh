#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/timeline.py > gpurun_out/timeline_mfma.txt 2>&1; grep "dur \|span\|per-CU\|matrix-core kernel\|tile time per wave" gpurun_out/timeline_mfma.txt
timeout 300 python tools/timeline.py nuscenes_gs144000 > gpurun_out/timeline_mfma_144.txt 2>&1; grep "dur \|span\|per-CU\|matrix-core kernel\|tile time per wave" gpurun_out/timeline_mfma_144.txt
