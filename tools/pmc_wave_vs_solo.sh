export TMPDIR=/tmp
mkdir -p gpurun_out/pmc5
for mode in wave solo; do
  if [ $mode = solo ]; then export GF_MFMA_SOLO=1; else unset GF_MFMA_SOLO; fi
  for pass in "A:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" "B:SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" "C:SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_FLAT"; do
    name=${pass%%:*}; ctrs=${pass#*:}
    rm -rf gpurun_out/pmc5/$mode$name
    rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/pmc5/$mode$name -- python tools/prof_fwd.py nuscenes_gs25600_solid 8 0 > gpurun_out/pmc5/$mode$name.log 2>&1
    echo "== $mode pass $name" >> gpurun_out/pmc5/summary.txt
    python tools/pmc_summary.py gpurun_out/pmc5/$mode$name | grep -A9 "render" >> gpurun_out/pmc5/summary.txt
  done
done
cat gpurun_out/pmc5/summary.txt
