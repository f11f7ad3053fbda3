#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/pmcA gpurun_out/pmcB gpurun_out/pmcC gpurun_out/pmcD
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmcA -- python tools/prof_fwd.py nuscenes_gs25600_solid 6 1 > gpurun_out/pmcA.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/pmcB -- python tools/prof_fwd.py nuscenes_gs25600_solid 6 1 > gpurun_out/pmcB.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcC -- python tools/prof_fwd.py nuscenes_gs25600_solid 6 1 > gpurun_out/pmcC.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmcD -- python tools/prof_fwd.py nuscenes_gs25600_solid 6 1 > gpurun_out/pmcD.log 2>&1
for x in A B C D; do echo "== pass $x"; tail -2 gpurun_out/pmc$x.log; python tools/pmc_summary.py gpurun_out/pmc$x | grep -A12 "render_dense\|prep_kernel"; done
