#!/bin/bash
# GPU box: kernel time of the reference's own kernels (oracle/_ref) and of libgf_hip.so on the same inputs.
#   bash tools/gpu/ref_compare.sh <tag>      -> gpurun_out/profiles_<tag>/ref_vs_hip_<tag>.txt
tag=${1:-r02}
repo=$(pwd)
out=$repo/gpurun_out/profiles_$tag
work=$repo/gpurun_out/refcmp
mkdir -p "$out" "$work"
cd /tmp && export TMPDIR=/tmp
reps=3
for case in base_gs25600 base_gs144000 prob_gs6400 prob_fast_gs6400 daf_gs6400 daf_gs25600; do
    for which in ref hip; do
        d=$work/${case}_$which
        rm -rf "$d"; mkdir -p "$d"
        echo $reps > "$d/reps"
        (cd "$repo" && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -o k -- python tools/bench_ref.py $case $which $reps) > "$d/log.txt" 2>&1 \
            || echo "$case $which failed (see $d/log.txt)"
    done
done
cd "$repo" && python tools/bench_ref.py --summarise "$work" "$out/ref_vs_hip_$tag.txt"
