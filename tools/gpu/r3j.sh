#!/bin/bash
# does the headline depend on how long the GPU has been busy before the timed region?
for w in 20 200 2000; do
python bench.py --steps 200 --warmup $w --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('warmup $w: us/step %.2f  kernel us %.2f' % (j['ms_per_step']*1e3, j['roofline']['kernel_us']))"
done
python bench.py --steps 2000 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps 2000 warmup 20: us/step %.2f  kernel us %.2f' % (j['ms_per_step']*1e3, j['roofline']['kernel_us']))"
python tools/mfma_probe.py nuscenes_gs25600_solid 2>&1 | grep "us per step"
