#!/bin/bash
# Development: the records pass at P = 144 000 with parts compiled out (tools/xbuild.sh xpN -DGF_XP=N: 1 bitmask stores, 2 record
# stores, 8 no verification waves, 16 verification waves only) -- prep kernel durations from rocprofv3 --kernel-trace
mkdir -p gpurun_out/s7; O=gpurun_out/s7; export TMPDIR=/tmp
for x in base xp1 xp2 xp3 xp8 xp16; do
  if [ $x = base ]; then unset GF_LIB; else export GF_LIB=$PWD/gaussianformer_amd/csrc/libgf_hip_$x.so; fi
  rocprofv3 --kernel-trace --stats -d $O/$x -o p --output-format csv -- python tools/prof_fwd.py nuscenes_gs144000 60 > $O/$x.log 2>&1
  echo "$x: $(grep prep_kernel $O/$x/p_kernel_stats.csv | cut -d, -f1-4,6,7 | cut -c1-160)"
done
