#!/bin/bash
# usage: tools/traffic.sh <outdir-name> [prof_frame args...]
# HBM-side traffic of every kernel: FETCH_SIZE and WRITE_SIZE in two separate PMC passes (they do not fit one pass)
# over tools/prof_frame.py.  Run on the GPU box through gpurun; writes gpurun_out/<name>/{fetch,write}/.
R=$GRAFT_REPO_ROOT; N=$1; shift
mkdir -p $(dirname $R/gpurun_out/$N)
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$N/fetch -o p -- python3 $R/tools/prof_frame.py "$@" > $R/gpurun_out/$N.fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/$N/write -o p -- python3 $R/tools/prof_frame.py "$@" > $R/gpurun_out/$N.write.log 2>&1 || exit 1
