#!/bin/bash
# usage (on the GPU box): bash tools/sweep_variants.sh [frames]
# Every experimental build exp/librtgpu_<name>.so (same ABI, other compile-time constants) renders the benchmark frame
# with its kernels one after the other; prints the last frame's phase times next to the shipped library's.
R=$GRAFT_REPO_ROOT; F=${1:-5}
export RTGPU_SERIAL=1
echo -n "shipped: "; timeout -k 10 100 python3 $R/tools/prof_frame.py $F | cut -d' ' -f1-11 || exit 1
for L in $R/exp/librtgpu_*.so; do
    N=$(basename $L .so); echo -n "${N#librtgpu_}: "
    RTGPU_LIB=$L timeout -k 10 100 python3 $R/tools/prof_frame.py $F | cut -d' ' -f1-11 || exit 1
done
