#!/bin/bash
# usage: tools/pmc_sqc.sh <outdir-name> [prof_frame args...]   -- scalar-cache / instruction-cache PMC passes over tools/prof_frame.py
R=$GRAFT_REPO_ROOT; N=$1; shift
cd /tmp && export TMPDIR=/tmp
P1="SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_BUSY_CYCLES SQC_DCACHE_INPUT_VALID_READYB SQC_TC_STALL SQC_TC_DATA_READ_REQ SQ_INST_LEVEL_SMEM"
P2="SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $P --output-format csv -d $R/gpurun_out/$N/p$i -o p -- python3 $R/tools/prof_frame.py "$@" > $R/gpurun_out/$N.p$i.log 2>&1 || exit 1
done
