#!/bin/bash
# usage: tools/pmc.sh <outdir-name> [prof_frame args...]   -- five PMC passes over tools/prof_frame.py
# (run on the GPU box through gpurun; writes gpurun_out/<name>/*)
R=$GRAFT_REPO_ROOT; N=$1; shift
mkdir -p $(dirname $R/gpurun_out/$N)
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
P2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM"
P3="TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TD_TD_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum GRBM_GUI_ACTIVE"
P4="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
# lane-weighted VALU work (thread-cycles against instruction-cycles x 64 = the share of busy VALU cycles whose lanes are live)
# and the scalar unit's busy cycles
P5="SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $P --output-format csv -d $R/gpurun_out/$N/p$i -o p -- python3 $R/tools/${PROG:-prof_frame.py} "$@" > $R/gpurun_out/$N.p$i.log 2>&1 || exit 1
done
