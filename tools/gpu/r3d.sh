#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_splat_mfma_gpu.py tests/test_splat_gpu.py tests/test_ref_parity.py -m gpu -q -x 2>&1 | tail -5
timeout 300 python tools/mfma_probe.py > gpurun_out/probe.txt 2>&1; grep "us per step\|_ref" gpurun_out/probe.txt
timeout 300 python tools/timeline.py > gpurun_out/timeline_mfma.txt 2>&1; grep "dur \|span\|per-CU\|matrix-core\|tile time" gpurun_out/timeline_mfma.txt
