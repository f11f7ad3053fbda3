#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python bench.py --steps 200 --warmup 20 > gpurun_out/bench.json 2> gpurun_out/bench.err; cat gpurun_out/bench.json; tail -3 gpurun_out/bench.err
rm -rf gpurun_out/kt gpurun_out/pmcA gpurun_out/pmcB
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/kt.log 2>&1
f=$(find gpurun_out/kt -name "*kernel_stats.csv" | head -1); echo "== $f"; head -8 "$f"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmcA -- python tools/prof_fwd.py nuscenes_gs25600_solid 6 0 > gpurun_out/pmcA.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/pmcB -- python tools/prof_fwd.py nuscenes_gs25600_solid 6 0 > gpurun_out/pmcB.log 2>&1
for x in A B; do echo "== pass $x"; python tools/pmc_summary.py gpurun_out/pmc$x | grep -A9 "render_kernel\|prep_kernel"; done
