#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_daf_gpu.py -m gpu -q -x 2>&1 | tail -15
rm -rf gpurun_out/kt_daf; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_daf -- python tools/prof_daf.py > gpurun_out/kt_daf.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/kt_daf/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "gf" in r["Name"] or float(r["Percentage"]) > 2: print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:9.1f} us  {r["Percentage"]}%')
PY
timeout 300 python tools/bench_ops.py 2>/dev/null | grep daf_backward | cut -c1-120
