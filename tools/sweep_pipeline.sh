#!/bin/bash
# usage: tools/sweep_pipeline.sh <n_shards> "<depths>"  -- shard_pipeline.py over hardware queues x side streams x level sizing
N=${1:-8}; D=${2:-4,8}
for Q in 4 24; do for S in 1 4; do for SY in 0 1; do
  echo "== queues $Q side-streams $S sync-levels $SY"
  RT_HW_QUEUES=$Q RTGPU_SIDE_STREAMS=$S RTGPU_SYNC_LEVELS=$SY DEPTHS=$D timeout -k 10 120 python3 tools/shard_pipeline.py $N 12 || exit 1
done; done; done
