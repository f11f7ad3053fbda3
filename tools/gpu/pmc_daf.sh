#!/bin/bash
# counter evidence for the deformable aggregation and splat-backward kernels -- one rocprofv3 --pmc pass per
# counter group (counters only: no trace domains beside --pmc), both location distributions.
#   bash tools/gpu/pmc_daf.sh r04      -> gpurun_out/profiles_r04/pmc_daf_{uniform,projected}_r04.txt, pmc_splat_bwd_r04.txt
R=${1:-r04}
OUT=gpurun_out/profiles_$R
mkdir -p $OUT; export TMPDIR=/tmp
for dist in uniform projected; do
  : > $OUT/pmc_daf_${dist}_$R.txt
  for pass in "C:FETCH_SIZE" "D:WRITE_SIZE" "E:TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "F:TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum" "A:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
    name=${pass%%:*}; ctrs=${pass#*:}
    rm -rf gpurun_out/pmcdaf_${dist}_$name
    timeout 300 rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/pmcdaf_${dist}_$name -- python tools/prof_daf2.py $dist 3 both > gpurun_out/pmcdaf_${dist}_$name.log 2>&1
    echo "== PMC pass $name ($dist): $ctrs" >> $OUT/pmc_daf_${dist}_$R.txt
    python tools/pmc_summary.py gpurun_out/pmcdaf_${dist}_$name | grep -A12 "gf_daf" >> $OUT/pmc_daf_${dist}_$R.txt
  done
done
: > $OUT/pmc_splat_bwd_$R.txt
for pass in "C:FETCH_SIZE" "D:WRITE_SIZE" "E:TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "F:TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum" "A:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rm -rf gpurun_out/pmcbwd_$name
  timeout 300 rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/pmcbwd_$name -- python tools/prof_bwd.py nuscenes_gs25600_solid 4 > gpurun_out/pmcbwd_$name.log 2>&1
  echo "== PMC pass $name: $ctrs" >> $OUT/pmc_splat_bwd_$R.txt
  python tools/pmc_summary.py gpurun_out/pmcbwd_$name | grep -A12 "gf_splat_bwd\|gf_bwd" >> $OUT/pmc_splat_bwd_$R.txt
done
python tools/make_traffic_daf.py $OUT $R > $OUT/traffic_daf_$R.json
cat $OUT/traffic_daf_$R.json
