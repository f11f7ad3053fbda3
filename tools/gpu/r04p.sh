#!/bin/bash
# round 4, call P: 2-D pixel tiles in the DAF backward: parity tests, op timings (both distributions), kernel trace
R=r04p
OUT=gpurun_out/profiles_$R
mkdir -p $OUT; export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_daf_gpu.py tests/test_ref_parity.py -m gpu -q -x --tb=short --timeout 200 -k "daf or deformable" 2>&1 | tail -15 > $OUT/pytest_$R.log; cat $OUT/pytest_$R.log
timeout 400 python tools/bench_ops.py > $OUT/bench_ops_$R.jsonl 2> $OUT/bench_ops_$R.err; python - <<'PY'
import json
for l in open('gpurun_out/profiles_r04p/bench_ops_r04p.jsonl'):
    try: d=json.loads(l)
    except Exception: continue
    print(d.get("op"), "|", d.get("config","")[:60], "|", round(d.get("us",0),1), "us")
PY
rm -rf gpurun_out/kt_daf2; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_daf2 -- python tools/prof_daf2.py uniform 5 bwd > gpurun_out/kt_daf2.log 2>&1; cp $(find gpurun_out/kt_daf2 -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_daf_bwd_uniform_$R.csv; cut -c1-140 $OUT/kernel_stats_daf_bwd_uniform_$R.csv | head -9
rm -rf gpurun_out/kt_daf3; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_daf3 -- python tools/prof_daf2.py projected 5 bwd > gpurun_out/kt_daf3.log 2>&1; cp $(find gpurun_out/kt_daf3 -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_daf_bwd_projected_$R.csv; cut -c1-140 $OUT/kernel_stats_daf_bwd_projected_$R.csv | head -9
timeout 100 python tools/probe_dense.py 2>&1 | tail -6
