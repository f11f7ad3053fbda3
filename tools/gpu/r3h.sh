#!/bin/bash
export TMPDIR=/tmp
for w in 64 58 53 48 40; do echo "WGs per XCD $w"; GF_MFMA_WGS_PER_XCD=$w timeout 200 python tools/mfma_probe.py 2>&1 | grep "mfma: .*us per step"; done
echo default; timeout 200 python tools/mfma_probe.py 2>&1 | grep "mfma: .*us per step"
