#!/bin/bash
# full GPU evidence run: tests, smoke, bench, secondary ops, profiles
R=${1:-r01}
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/pytest_gpu_$R.log; cat gpurun_out/pytest_gpu_$R.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$R.log 2>&1; tail -2 gpurun_out/smoke_$R.log
bash tools/collect_profiles.sh $R > gpurun_out/collect_$R.log 2>&1; head -3 gpurun_out/collect_$R.log | cut -c1-600
timeout 600 python tools/bench_ops.py > gpurun_out/profiles_$R/bench_ops_$R.jsonl 2> gpurun_out/bench_ops.err; cat gpurun_out/profiles_$R/bench_ops_$R.jsonl
timeout 300 python tools/timeline.py nuscenes_gs25600_solid exact > gpurun_out/profiles_$R/timeline_render_$R.txt 2>&1
timeout 300 python tools/timeline_wave.py nuscenes_gs25600_solid > gpurun_out/profiles_$R/timeline_render_mfma_wave_$R.txt 2>&1; tail -12 gpurun_out/profiles_$R/timeline_render_mfma_wave_$R.txt
timeout 300 python tools/timeline.py nuscenes_gs25600_solid > gpurun_out/profiles_$R/timeline_render_mfma_tile_$R.txt 2>&1
timeout 300 python tools/timeline.py nuscenes_gs144000 > gpurun_out/profiles_$R/timeline_render_mfma_tile_gs144000_$R.txt 2>&1
timeout 300 python tools/probe_dense.py > gpurun_out/profiles_$R/probe_assume_dense_$R.txt 2>&1
bash tools/gpu/kernel_pair.sh > gpurun_out/profiles_$R/kernel_pair_$R.txt 2>&1; cat gpurun_out/profiles_$R/kernel_pair_$R.txt
timeout 300 python tools/mfma_probe.py > gpurun_out/profiles_$R/mfma_probe_$R.txt 2>&1; grep "us per step\|oracle/_ref" gpurun_out/profiles_$R/mfma_probe_$R.txt
echo "--- GF_MFMA_TILE=1 (the tile matrix-core kernel at every P)" >> gpurun_out/profiles_$R/mfma_probe_$R.txt; GF_MFMA_TILE=1 timeout 300 python tools/mfma_probe.py 2>&1 | grep "mfma" >> gpurun_out/profiles_$R/mfma_probe_$R.txt
timeout 300 python tools/bench_frame.py --frames 20 --graph > gpurun_out/profiles_$R/bench_frame_$R.jsonl 2> gpurun_out/bench_frame.err; cut -c1-140 gpurun_out/profiles_$R/bench_frame_$R.jsonl
rm -rf gpurun_out/kt_frame; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_frame -- python tools/bench_frame.py --configs nuscenes_gs25600_solid --frames 10 > gpurun_out/kt_frame.log 2>&1; cp $(find gpurun_out/kt_frame -name '*kernel_stats.csv' | head -1) gpurun_out/profiles_$R/kernel_stats_frame_gs25600_$R.csv
timeout 300 python tools/uniform_experiment.py > gpurun_out/profiles_$R/uniform_experiment_$R.txt 2>&1
[ -x tools/microbench/mfma4x4 ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/microbench/mfma4x4 tools/microbench/mfma4x4.hip 2>/dev/null
./tools/microbench/mfma4x4 > gpurun_out/profiles_$R/microbench_mfma4x4_$R.txt 2>&1
for m in dense_block dense_proto; do [ -x tools/microbench/$m ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o tools/microbench/$m tools/microbench/$m.hip 2>/dev/null; timeout 120 ./tools/microbench/$m > gpurun_out/profiles_$R/microbench_${m}_$R.txt 2>&1; done
for m in valu mfma lanes; do [ -x tools/microbench/$m ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/microbench/$m tools/microbench/$m.hip 2>/dev/null; done
./tools/microbench/valu > gpurun_out/profiles_$R/microbench_valu_$R.txt 2>&1
./tools/microbench/mfma > gpurun_out/profiles_$R/microbench_mfma_$R.txt 2>&1
./tools/microbench/lanes > gpurun_out/profiles_$R/microbench_lanes_$R.txt 2>&1
rm -rf gpurun_out/kt_subm; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_subm -- python tools/prof_subm.py > gpurun_out/kt_subm.log 2>&1; cp $(find gpurun_out/kt_subm -name '*kernel_stats.csv' | head -1) gpurun_out/profiles_$R/kernel_stats_subm_$R.csv
rm -rf gpurun_out/kt_daf; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_daf -- python tools/prof_daf.py > gpurun_out/kt_daf.log 2>&1; cp $(find gpurun_out/kt_daf -name '*kernel_stats.csv' | head -1) gpurun_out/profiles_$R/kernel_stats_daf_$R.csv
rm -rf gpurun_out/kt_q; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_q -- python tools/quick_time.py nuscenes_gs25600_solid > gpurun_out/profiles_$R/quick_time_gs25600_$R.log 2>&1; cp $(find gpurun_out/kt_q -name '*kernel_stats.csv' | head -1) gpurun_out/profiles_$R/kernel_stats_ops_gs25600_$R.csv
# BASELINE config [2]: the native ops of one training step, chained
timeout 300 python tools/bench_step.py > gpurun_out/profiles_$R/bench_step_$R.json 2> gpurun_out/bench_step.err
rm -rf gpurun_out/kt_step; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_step -- python tools/bench_step.py --steps 5 --warmup 2 > gpurun_out/kt_step.log 2>&1; cp $(find gpurun_out/kt_step -name '*kernel_stats.csv' | head -1) gpurun_out/profiles_$R/kernel_stats_step_$R.csv
# BASELINE config [3]: nuscenes_gs144000 inference
python bench.py --config nuscenes_gs144000 --no-cpu-baseline > gpurun_out/profiles_$R/bench_gs144000_$R.json 2> gpurun_out/bench_gs144000.err
rm -rf gpurun_out/kt_144; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_144 -- python bench.py --config nuscenes_gs144000 --no-cpu-baseline --no-extras > gpurun_out/kt_144.log 2>&1; cp $(find gpurun_out/kt_144 -name '*kernel_stats.csv' | head -1) gpurun_out/profiles_$R/kernel_stats_bench_gs144000_$R.csv
cp gpurun_out/pytest_gpu_$R.log gpurun_out/smoke_$R.log gpurun_out/profiles_$R/
# the reference's own kernels beside ours (same inputs, kernel time by rocprofv3)
timeout 900 bash tools/gpu/ref_compare.sh $R > gpurun_out/ref_compare.log 2>&1; grep "==" gpurun_out/profiles_$R/ref_vs_hip_$R.txt
