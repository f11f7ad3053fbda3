#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for pass in "A:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" "B:SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" "E:SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rm -rf gpurun_out/pmc_$name
  rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/pmc_$name -- python tools/prof_fwd.py nuscenes_gs25600_solid 8 0 > gpurun_out/pmc_$name.log 2>&1
  echo "== PMC pass $name: $ctrs"
  python tools/pmc_summary.py gpurun_out/pmc_$name | grep -A12 "mfma_kernel"
done
timeout 300 python tools/bench_frame.py --frames 10 --graph 2>&1 | cut -c1-400
