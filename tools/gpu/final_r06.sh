#!/bin/bash
# round 6 evidence run: tests, smoke, bench, profiles (copy gpurun_out/profiles_$R/* into profiles/ afterwards)
R=${1:-r06}
OUT=gpurun_out/profiles_$R
mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -6 > $OUT/pytest_gpu_$R.log; cat $OUT/pytest_gpu_$R.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke_$R.log 2>&1; tail -1 $OUT/smoke_$R.log
# PMC passes of the forward (traffic json read by bench.py), then the bench line and its kernel trace
timeout 1200 bash tools/collect_profiles.sh $R > gpurun_out/collect_$R.log 2>&1; head -c 400 $OUT/bench_$R.json; echo; cat $OUT/kernel_stats_bench_$R.csv | cut -c1-120 | head -4
cp profiles/traffic_$R.json profiles/traffic_gs144000_$R.json $OUT/ 2>/dev/null
timeout 500 python tools/bench_ops.py > $OUT/bench_ops_$R.jsonl 2> gpurun_out/bench_ops.err
# long rows (P = 144 000): the wave kernel's long-row instantiation against the tile kernel -- equal bits, repeatability, step times
(timeout 300 python tools/long_rows_check.py; timeout 300 python tools/long_rows_repro.py
 for i in 1 2; do timeout 200 python tools/fwd_time.py nuscenes_gs25600_solid nuscenes_gs144000; timeout 200 python tools/fwd_time.py nuscenes_gs144000 --tile; done) 2>&1 | grep -v amdgpu.ids > $OUT/long_rows_$R.txt; cat $OUT/long_rows_$R.txt | cut -c1-220
# sparse convolution at 144 000 anchors: kernel trace, runs of tiles
rm -rf gpurun_out/kt_subm; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_subm -- python tools/prof_subm.py 144000 > gpurun_out/kt_subm.log 2>&1; cp $(find gpurun_out/kt_subm -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_subm_$R.csv; head -4 $OUT/kernel_stats_subm_$R.csv | cut -c1-140
# splat backward
timeout 200 python tools/bwd_probe.py full > $OUT/bwd_probe_full_$R.txt 2>&1; grep "us per call\|vs oracle" $OUT/bwd_probe_full_$R.txt | cut -c1-200
timeout 300 python tools/parity_report.py > $OUT/parity_$R.txt 2>&1; tail -12 $OUT/parity_$R.txt | cut -c1-200
# frames: fused inference DAF (default) against the three-step path; training step
timeout 300 python tools/bench_frame.py --frames 20 --graph > $OUT/bench_frame_$R.jsonl 2> gpurun_out/bench_frame.err; cut -c1-200 $OUT/bench_frame_$R.jsonl
GF_FRAME_THREE_STEP_DAF=1 timeout 300 python tools/bench_frame.py --frames 20 --graph > $OUT/bench_frame_three_step_daf_$R.jsonl 2> /dev/null; cut -c1-200 $OUT/bench_frame_three_step_daf_$R.jsonl
timeout 300 python tools/bench_step.py > $OUT/bench_step_$R.json 2> gpurun_out/bench_step.err; cat $OUT/bench_step_$R.json | cut -c1-300
timeout 300 python bench.py --config nuscenes_gs144000 --no-cpu-baseline --no-extras > $OUT/bench_gs144000_$R.json 2> gpurun_out/bench_gs144000.err; head -c 400 $OUT/bench_gs144000_$R.json
# the development build's kernels stay honest (pair / solo / fused against the oracle)
if [ -f gaussianformer_amd/csrc/libgf_hip_dev.so ]; then GF_LIB=$PWD/gaussianformer_amd/csrc/libgf_hip_dev.so timeout 600 python tools/dev_kernels_check.py 2>&1 | grep -v amdgpu.ids | tail -6 > $OUT/dev_kernels_$R.txt; cat $OUT/dev_kernels_$R.txt; fi
# round 6, second half: the matrix-core backward on long rows (pair timings + kernel trace), the long-row sweep, the sparse convolution's
# three gather-GEMM arithmetics side by side
(timeout 200 python tools/prof_fb.py nuscenes_gs144000; timeout 200 python tools/prof_fb.py nuscenes_gs25600_solid; timeout 300 python tools/bwd_long_check.py) 2>&1 | grep -v amdgpu.ids > $OUT/bwd_long_rows_$R.txt; cut -c1-250 $OUT/bwd_long_rows_$R.txt
rm -rf gpurun_out/kt_fb; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_fb -- python tools/prof_fb.py nuscenes_gs144000 30 > gpurun_out/kt_fb.log 2>&1; cp $(find gpurun_out/kt_fb -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_fwd_bwd_gs144000_$R.csv; head -5 $OUT/kernel_stats_fwd_bwd_gs144000_$R.csv | cut -c1-140
timeout 600 python tools/long_rows_sweep.py 16 2>&1 | grep -v amdgpu.ids > $OUT/long_rows_sweep_$R.txt; tail -1 $OUT/long_rows_sweep_$R.txt
timeout 300 python tools/subm_f16_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/subm_f16_probe_$R.txt; cat $OUT/subm_f16_probe_$R.txt
