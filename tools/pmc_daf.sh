# Development (GPU box): instruction and wait counters of the deformable aggregation backward's kernels, uniform and projected samples
export TMPDIR=/tmp
out=gpurun_out/pmc_daf
mkdir -p $out; rm -f $out/summary.txt
for dist in ${DISTS:-uniform projected}; do
  rm -rf $out/kt_$dist
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$dist -- python tools/daf_region_probe.py $dist > $out/kt_$dist.log 2>&1
  echo "== $dist kernel stats" >> $out/summary.txt
  f=$(find $out/kt_$dist -name '*kernel_stats.csv' | head -1); head -12 $f | cut -c1-200 >> $out/summary.txt
  for pass in "A:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" "B:SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" "C:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_FLAT"; do
    name=${pass%%:*}; ctrs=${pass#*:}
    rm -rf $out/$dist$name
    rocprofv3 --pmc $ctrs --output-format csv -d $out/$dist$name -- python tools/daf_region_probe.py $dist > $out/$dist$name.log 2>&1
    echo "== $dist pass $name" >> $out/summary.txt
    python tools/pmc_summary.py $out/$dist$name | grep -A9 "raccumulate\|gf_daf_bwd_kernel\|gf_daf_accumulate" >> $out/summary.txt
  done
done
cat $out/summary.txt
