#!/bin/bash
# usage (on the GPU box): bash tools/sweep_variants.sh [frames]
# Every experimental build exp/librtgpu_<name>.so (same ABI, other compile-time constants; built beforehand, e.g.
#   hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DRT_SPARSE=12 -c -o /tmp/v.o raytracer_amd/csrc/rtgpu.hip &&
#   hipcc -shared -fPIC --offload-arch=gfx950 -o exp/librtgpu_sp12.so /tmp/v.o build/obj/bvh_build.o
# exp/ is git-ignored and travels to the GPU box) renders the benchmark frame
# with its kernels one after the other; prints the last frame's phase times next to the shipped library's.
R=$GRAFT_REPO_ROOT; F=${1:-5}
export RTGPU_SERIAL=1
echo -n "shipped: "; timeout -k 10 100 python3 $R/tools/prof_frame.py $F | cut -d' ' -f1-11 || exit 1
for L in $R/exp/librtgpu_*.so; do
    N=$(basename $L .so); echo -n "${N#librtgpu_}: "
    RTGPU_LIB=$L timeout -k 10 100 python3 $R/tools/prof_frame.py $F | cut -d' ' -f1-11 || exit 1
done
