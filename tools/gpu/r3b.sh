#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/mfma_probe.py > gpurun_out/probe_lpt.txt 2>&1; grep "us per step" gpurun_out/probe_lpt.txt
GF_NO_LPT=1 timeout 300 python tools/mfma_probe.py > gpurun_out/probe_nolpt.txt 2>&1; grep "us per step" gpurun_out/probe_nolpt.txt
rm -rf gpurun_out/kt; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -- python tools/mfma_probe.py nuscenes_gs25600_solid > gpurun_out/kt.log 2>&1
cut -d, -f1-4 $(find gpurun_out/kt -name "*kernel_stats.csv" | head -1) | head -8
