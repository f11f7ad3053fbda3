#!/bin/bash
# gradient rows by coalesced DMA + split once per unit
R=r04w
OUT=gpurun_out/profiles_$R
mkdir -p $OUT; export TMPDIR=/tmp
timeout 200 python tools/bwd_probe.py small > $OUT/bwd_probe_small_$R.txt 2>&1; grep -c "finite True" $OUT/bwd_probe_small_$R.txt; grep "finite False\|e-0[0-3] \|e+0" $OUT/bwd_probe_small_$R.txt | head -20
timeout 600 python -m pytest tests/test_splat_mfma_gpu.py tests/test_splat_gpu.py -m gpu -q -x --tb=short --timeout 120 2>&1 | tail -30 > $OUT/pytest_new_$R.log; cat $OUT/pytest_new_$R.log
timeout 200 python tools/bwd_probe.py full nuscenes_gs25600_solid > $OUT/bwd_probe_$R.txt 2>&1; grep "us per call\|vs oracle" $OUT/bwd_probe_$R.txt | cut -c1-220
timeout 200 python tools/timeline_bwd.py nuscenes_gs25600_solid > $OUT/timeline_bwd_$R.txt 2>&1; tail -12 $OUT/timeline_bwd_$R.txt
