#!/bin/bash
# Collects the round's evidence on the GPU box into gpurun_out/profiles_rNN/ (copy into profiles/ afterwards):
#   PMC passes (separate runs, counters only) -> HBM traffic per launch; bench JSON (reads that traffic figure);
#   rocprofv3 --kernel-trace --stats of the same bench command.
R=${1:-r01}
OUT=gpurun_out/profiles_$R
mkdir -p $OUT; export TMPDIR=/tmp
for pass in "A:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" "B:SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" "C:FETCH_SIZE" "D:WRITE_SIZE"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rm -rf gpurun_out/pmc_${R}_$name
  rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/pmc_${R}_$name -- python tools/prof_fwd.py nuscenes_gs25600_solid 8 0 > gpurun_out/pmc_${R}_$name.log 2>&1
  echo "== PMC pass $name: $ctrs" >> $OUT/pmc_$R.txt
  python tools/pmc_summary.py gpurun_out/pmc_${R}_$name >> $OUT/pmc_$R.txt
done
python tools/make_traffic.py gpurun_out/pmc_${R}_C gpurun_out/pmc_${R}_D $OUT/traffic_$R.json $R
cp $OUT/traffic_$R.json profiles/traffic_$R.json   # bench.py reports the newest profiles/traffic_r*.json
# the same two TCC passes for BASELINE config [3] (P = 144 000)
for pass in "C:FETCH_SIZE" "D:WRITE_SIZE"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rm -rf gpurun_out/pmc144_${R}_$name
  rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/pmc144_${R}_$name -- python tools/prof_fwd.py nuscenes_gs144000 8 0 > gpurun_out/pmc144_${R}_$name.log 2>&1
done
python tools/make_traffic.py gpurun_out/pmc144_${R}_C gpurun_out/pmc144_${R}_D $OUT/traffic_gs144000_$R.json $R nuscenes_gs144000 144000
cp $OUT/traffic_gs144000_$R.json profiles/traffic_gs144000_$R.json
python bench.py --steps 200 --warmup 20 > $OUT/bench_$R.json 2> $OUT/bench_$R.err
cat $OUT/bench_$R.json
rm -rf gpurun_out/kt_$R
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_$R -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras > $OUT/kernel_trace_bench_$R.log 2>&1
cp $(find gpurun_out/kt_$R -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_bench_$R.csv
cat $OUT/kernel_stats_bench_$R.csv
cat $OUT/pmc_$R.txt
