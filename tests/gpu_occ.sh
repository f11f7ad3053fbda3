#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for dyn in 0 3500 7500 12000 20000 34000; do
  echo "== dyn LDS $dyn"; GF_RENDER_DYN_LDS=$dyn python bench.py --steps 300 --warmup 30 --no-cpu-baseline | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('us/step %.1f render us %.1f' % (j['ms_per_step']*1e3, j['roofline']['kernel_us']))"
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/occ.log
