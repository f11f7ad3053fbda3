#!/bin/bash
timeout 900 python -m pytest tests/test_splat_mfma_gpu.py tests/test_golden_gpu.py tests/test_slab_gpu.py -m gpu -q -x 2>&1 | tail -3
python tools/timeline_wave.py 2>&1 | grep -v amdgpu.ids
for rep in 1 2; do
python tools/mfma_probe.py nuscenes_gs25600_solid 2>&1 | grep "mfma: .*us per"
echo tile; GF_MFMA_TILE=1 python tools/mfma_probe.py nuscenes_gs25600_solid 2>&1 | grep "mfma: .*us per"
done
