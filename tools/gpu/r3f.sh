#!/bin/bash
export TMPDIR=/tmp
echo "baseline"; timeout 200 python tools/mfma_probe.py 2>&1 | grep "mfma: .*us per step"
for e in 10; do echo "experiment $e"; GF_LIB=$PWD/gaussianformer_amd/csrc/libgf_hip_exp$e.so timeout 200 python tools/mfma_probe.py 2>&1 | grep "mfma: .*us per step\|mfma vs"; done
