#!/bin/bash
# Development: variant libraries of splat_bwd_mfma.hip only (the other objects are the product's), e.g.  tools/xbuild_bwd.sh xb1 -DGF_XB=1
set -e
name=$1; shift
C=gaussianformer_amd/csrc
hipcc --offload-arch=gfx950 -Os -std=c++17 -fPIC -munsafe-fp-atomics -Wno-inline-asm "$@" -c $C/splat_bwd_mfma.hip -o $C/splat_bwd_mfma.$name.o
objs=""
for f in gf_api splat_fwd splat_bwd daf gaussian_prepare daf_prepare head_labels feature_format subm_conv key_points; do objs="$objs $C/$f.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libgf_hip_$name.so $C/splat_bwd_mfma.$name.o $objs
echo $C/libgf_hip_$name.so
