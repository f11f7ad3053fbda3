#!/bin/bash
timeout 120 python -m pytest tests/test_splat_mfma_gpu.py tests/test_golden_gpu.py -m gpu -q -x 2>&1 | tail -2
for rep in 1 2 3; do
timeout 60 python tools/mfma_probe.py nuscenes_gs25600_solid 2>&1 | grep "mfma: .*us per"
GF_MFMA_TILE=1 timeout 60 python tools/mfma_probe.py nuscenes_gs25600_solid 2>&1 | grep "mfma: .*us per"
done
