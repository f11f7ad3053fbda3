#!/bin/bash
# round 4, call E: where the matrix-core backward spends its time (per-unit timeline, SQ counters); forward kernel pair for both libraries
R=r04e
OUT=gpurun_out/profiles_$R
mkdir -p $OUT; export TMPDIR=/tmp
timeout 200 python tools/timeline_bwd.py nuscenes_gs25600_solid > $OUT/timeline_bwd_$R.txt 2>&1; tail -14 $OUT/timeline_bwd_$R.txt
for pass in "A:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" "B:SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" "C:SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_FLAT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "D:FETCH_SIZE" "E:WRITE_SIZE"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rm -rf gpurun_out/pmcb_$name
  timeout 200 rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/pmcb_$name -- python tools/prof_bwd.py nuscenes_gs25600_solid 3 128 > gpurun_out/pmcb_$name.log 2>&1
  echo "== PMC pass $name: $ctrs" >> $OUT/pmc_bwd_mfma_$R.txt
  python tools/pmc_summary.py gpurun_out/pmcb_$name | grep -A10 "gf_splat_bwd" >> $OUT/pmc_bwd_mfma_$R.txt
done
cat $OUT/pmc_bwd_mfma_$R.txt
bash tools/gpu/kernel_pair.sh > $OUT/kernel_pair_$R.txt 2>&1; tail -8 $OUT/kernel_pair_$R.txt
GF_LIB=$PWD/gaussianformer_amd/csrc/libgf_hip_r03.so bash tools/gpu/kernel_pair.sh > $OUT/kernel_pair_r03lib_$R.txt 2>&1; tail -8 $OUT/kernel_pair_r03lib_$R.txt
