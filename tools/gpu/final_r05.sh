#!/bin/bash
# round 5 evidence run: tests, smoke, bench, profiles (copy gpurun_out/profiles_$R/* into profiles/ afterwards)
R=${1:-r05}
OUT=gpurun_out/profiles_$R
mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 300 2>&1 | tail -6 > $OUT/pytest_gpu_$R.log; cat $OUT/pytest_gpu_$R.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke_$R.log 2>&1; tail -1 $OUT/smoke_$R.log
# PMC passes of the forward (traffic json read by bench.py), then the bench line and its kernel trace
timeout 900 bash tools/collect_profiles.sh $R > gpurun_out/collect_$R.log 2>&1; head -c 600 $OUT/bench_$R.json; echo; cat $OUT/kernel_stats_bench_$R.csv | cut -c1-120
timeout 400 python tools/bench_ops.py > $OUT/bench_ops_$R.jsonl 2> gpurun_out/bench_ops.err
# deformable aggregation backward: regions against tiles, kernel stats, counters -> traffic_daf_$R.json
timeout 200 python tools/daf_region_probe.py > $OUT/daf_region_$R.txt 2>&1; cat $OUT/daf_region_$R.txt | grep backward
for dist in projected uniform; do
  rm -rf gpurun_out/kt_dafb_$dist; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_dafb_$dist -- python tools/prof_daf2.py $dist 10 both > gpurun_out/kt_dafb_$dist.log 2>&1
  cp $(find gpurun_out/kt_dafb_$dist -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_daf_${dist}_$R.csv
done
timeout 900 bash tools/gpu/pmc_daf.sh $R > gpurun_out/pmc_daf_$R.log 2>&1
# splat backward
timeout 200 python tools/bwd_probe.py full > $OUT/bwd_probe_full_$R.txt 2>&1; grep "us per call\|vs oracle" $OUT/bwd_probe_full_$R.txt | cut -c1-200
rm -rf gpurun_out/kt_bwd; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_bwd -- python tools/prof_bwd.py nuscenes_gs25600_solid 20 640 > gpurun_out/kt_bwd.log 2>&1; cp $(find gpurun_out/kt_bwd -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_bwd_mfma_$R.csv
# forward: the development kernels next to the default one; prep / render / gap of the default path
timeout 300 python tools/pair_probe.py > $OUT/fwd_kernels_$R.txt 2>&1; cat $OUT/fwd_kernels_$R.txt | tail -4
bash tools/gpu/kernel_pair.sh > $OUT/kernel_pair_$R.txt 2>&1; grep "mean\|per step" $OUT/kernel_pair_$R.txt
(timeout 300 python tools/interleave_probe.py nuscenes_gs25600_solid nuscenes_gs144000; timeout 200 python tools/interleave_probe_exact.py) 2>&1 | grep -v amdgpu.ids > $OUT/interleave_$R.txt; cat $OUT/interleave_$R.txt | cut -c1-200
timeout 100 python tools/subm_range_probe.py > $OUT/subm_range_$R.txt 2>&1; cat $OUT/subm_range_$R.txt | grep anchors
timeout 300 python tools/bench_frame.py --frames 20 --graph > $OUT/bench_frame_$R.jsonl 2> gpurun_out/bench_frame.err; cut -c1-200 $OUT/bench_frame_$R.jsonl
timeout 300 python tools/bench_step.py > $OUT/bench_step_$R.json 2> gpurun_out/bench_step.err; cat $OUT/bench_step_$R.json | cut -c1-300
rm -rf gpurun_out/kt_step; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_step -- python tools/bench_step.py --steps 5 --warmup 2 > gpurun_out/kt_step.log 2>&1; cp $(find gpurun_out/kt_step -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_step_$R.csv
timeout 300 python bench.py --config nuscenes_gs144000 --no-cpu-baseline --no-extras > $OUT/bench_gs144000_$R.json 2> gpurun_out/bench_gs144000.err; head -c 400 $OUT/bench_gs144000_$R.json
cp $OUT/traffic_$R.json $OUT/traffic_gs144000_$R.json gpurun_out/ 2>/dev/null
