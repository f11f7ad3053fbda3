#!/bin/bash
# Development: variant libraries of splat_fwd.hip only (the other objects are the product's), e.g.  tools/xbuild.sh x1 -DGF_X=1
# -> gaussianformer_amd/csrc/libgf_hip_x1.so  (select with GF_LIB=...)
set -e
name=$1; shift
C=gaussianformer_amd/csrc
hipcc --offload-arch=gfx950 -Os -std=c++17 -fPIC -Wno-inline-asm "$@" -c $C/splat_fwd.hip -o $C/splat_fwd.$name.o
objs=""
for f in gf_api splat_bwd splat_bwd_mfma daf daf_fused gaussian_prepare daf_prepare head_labels feature_format subm_conv key_points; do objs="$objs $C/$f.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libgf_hip_$name.so $C/splat_fwd.$name.o $objs
echo $C/libgf_hip_$name.so
