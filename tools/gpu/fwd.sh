#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_splat_gpu.py -m gpu -q -x 2>&1 | tail -8
python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench us/step %.1f  render kernel us %.1f  value %.3f G/s' % (j['ms_per_step']*1e3, j['roofline']['kernel_us'], j['value']/1e9))"
rm -rf gpurun_out/kt; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -- python tools/quick_time.py nuscenes_gs25600_solid > gpurun_out/kt.log 2>&1
grep -E "render_kernel<0, 2>|prep_kernel" $(find gpurun_out/kt -name "*kernel_stats.csv" | head -1) | cut -d, -f1-4
timeout 300 python tools/bench_ops.py --splat-only 2>/dev/null | grep "splat_forward" | cut -c1-110
