#!/bin/bash
for rep in 1 2; do
python tools/mfma_probe.py nuscenes_gs25600_solid 2>&1 | grep "mfma: .*us per"
echo tile; GF_MFMA_TILE=1 python tools/mfma_probe.py nuscenes_gs25600_solid 2>&1 | grep "mfma: .*us per"
done
