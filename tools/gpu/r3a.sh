#!/bin/bash
# round-3 GPU check: full GPU test suite, bench line (driver's command), probes
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -30 > gpurun_out/pytest_a.log; tail -12 gpurun_out/pytest_a.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err; python - <<'PY'
import json
b=json.loads(open('gpurun_out/bench_a.json').read().strip().splitlines()[-1])
print('value %.3f G/s  ms_per_step %.4f  roofline %s' % (b['value']/1e9, b['ms_per_step'], {k:b['roofline'][k] for k in ('frac','op_frac','kernel_us','kernel','verdict_bits')}))
for k in ('two_stream','hip_graph','exact_fp32_kernel'):
    print(k, b.get(k,{}).get('ms_per_step'), b.get(k,{}).get('error'))
f=b.get('frames_per_s',{})
for c in ('nuscenes_gs25600_solid','nuscenes_gs144000'):
    x=f.get(c,{}); print(c, x.get('frames_per_s'), x.get('frames_per_s_graph'), x.get('graph_labels_equal_eager'), (x.get('graph_error') or '')[:200])
print('train_step', {k:b.get('train_step',{}).get(k) for k in ('forward_ms','forward_backward_ms','error')})
PY
tail -3 gpurun_out/bench_a.err
