#!/bin/bash
# usage: tools/ktrace.sh <outdir-name> [prof_frame args...]
# rocprofv3 --kernel-trace --stats over tools/prof_frame.py with the kernels of a frame run one after the other
# (RTGPU_SERIAL=1): exclusive per-kernel durations.  Run on the GPU box through gpurun; writes gpurun_out/<name>/.
R=$GRAFT_REPO_ROOT; N=$1; shift
cd /tmp && export TMPDIR=/tmp
export RTGPU_SERIAL=${RTGPU_SERIAL-1}
timeout -s KILL 180 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$N -o kt -- python3 $R/tools/${PROG:-prof_frame.py} "$@" > $R/gpurun_out/$N.log 2>&1 || exit 1
