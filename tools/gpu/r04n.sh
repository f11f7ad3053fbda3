#!/bin/bash
# round 4, call N: regression run of the splat GPU tests (all backward paths), forward A/B
R=r04n
OUT=gpurun_out/profiles_$R
mkdir -p $OUT; export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_splat_gpu.py tests/test_splat_mfma_gpu.py tests/test_ref_parity.py tests/test_golden_gpu.py tests/test_head.py tests/test_slab_gpu.py tests/test_prepare.py tests/test_hot_path_chain.py tests/test_ref_callers.py -m gpu -q -x --tb=short --timeout 120 2>&1 | tail -25 > $OUT/pytest_$R.log; cat $OUT/pytest_$R.log
timeout 150 python tools/bwd_probe.py full > $OUT/bwd_probe_full_$R.txt 2>&1; grep "vs oracle\|us per call\|worst\|False" $OUT/bwd_probe_full_$R.txt | cut -c1-300
GF_LIB=$PWD/gaussianformer_amd/csrc/libgf_hip_r03.so timeout 200 python tools/mfma_probe.py nuscenes_gs25600_solid nuscenes_gs144000 2>&1 | grep "us per step" | sed 's/^/r03lib  /' | tee -a $OUT/ab_forward_$R.txt
timeout 200 python tools/mfma_probe.py nuscenes_gs25600_solid nuscenes_gs144000 2>&1 | grep "us per step" | sed 's/^/current /' | tee -a $OUT/ab_forward_$R.txt
