for l in base os; do
  if [ $l = base ]; then unset GF_LIB; else export GF_LIB=$PWD/gaussianformer_amd/csrc/libgf_hip_$l.so; fi
  echo "== $l"
  python tools/fwd_time.py nuscenes_gs25600_solid nuscenes_gs144000 2>&1 | grep -v amdgpu | cut -c1-110
  python tools/prof_fb.py nuscenes_gs25600_solid 2>&1 | grep -v amdgpu | head -1 | cut -c1-160
  python tools/prof_fb.py nuscenes_gs144000 2>&1 | grep -v amdgpu | head -1 | cut -c1-160
  python tools/daf_fused_time.py 2>&1 | grep -v amdgpu | cut -c1-100
  python tools/daf_fwd_time.py projected 2>&1 | grep -v amdgpu | cut -c1-120
  python tools/subm_f16_probe.py 2>&1 | grep -v amdgpu | cut -c1-160
done
