#!/bin/bash
# round 4 evidence run: tests, smoke, bench, profiles (copy gpurun_out/profiles_$R/* into profiles/ afterwards)
R=${1:-r04}
OUT=gpurun_out/profiles_$R
mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q --timeout 300 2>&1 | tail -6 > $OUT/pytest_gpu_$R.log; cat $OUT/pytest_gpu_$R.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke_$R.log 2>&1; tail -1 $OUT/smoke_$R.log
# PMC passes of the forward (traffic json read by bench.py), then the bench line and its kernel trace
timeout 900 bash tools/collect_profiles.sh $R > gpurun_out/collect_$R.log 2>&1; head -c 600 $OUT/bench_$R.json; echo; cat $OUT/kernel_stats_bench_$R.csv | cut -c1-120
timeout 300 python tools/bench_ops.py > $OUT/bench_ops_$R.jsonl 2> gpurun_out/bench_ops.err
timeout 200 python tools/bwd_probe.py full > $OUT/bwd_probe_full_$R.txt 2>&1; grep "us per call\|vs oracle" $OUT/bwd_probe_full_$R.txt | cut -c1-200
timeout 100 python tools/timeline_bwd.py nuscenes_gs25600_solid > $OUT/timeline_bwd_$R.txt 2>&1
rm -rf gpurun_out/kt_bwd; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_bwd -- python tools/prof_bwd.py nuscenes_gs25600_solid 20 640 > gpurun_out/kt_bwd.log 2>&1; cp $(find gpurun_out/kt_bwd -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_bwd_mfma_$R.csv
rm -rf gpurun_out/kt_bwd2; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_bwd2 -- python tools/prof_bwd.py nuscenes_gs25600_solid 20 256 > gpurun_out/kt_bwd2.log 2>&1; cp $(find gpurun_out/kt_bwd2 -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_bwd_exact_$R.csv
# HBM-side traffic of the matrix-core backward (separate counter passes)
: > $OUT/pmc_splat_bwd_mfma_$R.txt
for pass in "C:FETCH_SIZE" "D:WRITE_SIZE" "E:TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "A:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rm -rf gpurun_out/pmcbwdm_$name
  timeout 200 rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/pmcbwdm_$name -- python tools/prof_bwd.py nuscenes_gs25600_solid 4 640 > gpurun_out/pmcbwdm_$name.log 2>&1
  echo "== PMC pass $name: $ctrs" >> $OUT/pmc_splat_bwd_mfma_$R.txt
  python tools/pmc_summary.py gpurun_out/pmcbwdm_$name | grep -A12 "gf_splat_bwd" >> $OUT/pmc_splat_bwd_mfma_$R.txt
done
bash tools/gpu/kernel_pair.sh > $OUT/kernel_pair_$R.txt 2>&1; grep "mean\|per step" $OUT/kernel_pair_$R.txt
timeout 300 python tools/bench_frame.py --frames 20 --graph > $OUT/bench_frame_$R.jsonl 2> gpurun_out/bench_frame.err; cut -c1-200 $OUT/bench_frame_$R.jsonl
timeout 300 python tools/bench_step.py > $OUT/bench_step_$R.json 2> gpurun_out/bench_step.err; cat $OUT/bench_step_$R.json | cut -c1-300
rm -rf gpurun_out/kt_step; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_step -- python tools/bench_step.py --steps 5 --warmup 2 > gpurun_out/kt_step.log 2>&1; cp $(find gpurun_out/kt_step -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_step_$R.csv
timeout 300 python bench.py --config nuscenes_gs144000 --no-cpu-baseline --no-extras > $OUT/bench_gs144000_$R.json 2> gpurun_out/bench_gs144000.err; head -c 400 $OUT/bench_gs144000_$R.json
cp $OUT/traffic_$R.json $OUT/traffic_gs144000_$R.json gpurun_out/ 2>/dev/null
