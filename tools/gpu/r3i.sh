#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_splat_mfma_gpu.py -m gpu -q -x 2>&1 | tail -3
timeout 300 python tools/mfma_probe.py 2>&1 | grep "us per step\|mfma vs"
for pass in "C:FETCH_SIZE" "D:WRITE_SIZE"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rm -rf gpurun_out/pmc_$name
  rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/pmc_$name -- python tools/prof_fwd.py nuscenes_gs25600_solid 8 0 > gpurun_out/pmc_$name.log 2>&1
done
python tools/make_traffic.py gpurun_out/pmc_C gpurun_out/pmc_D gpurun_out/traffic_now.json now | grep "bytes\|KiB"
