#!/bin/bash
# usage: tools/sweep_batch.sh <n_shards>  -- shard_pipeline.py over batch sizes and hardware queues
N=${1:-8}
for Q in 4 24; do for B in 1 2 4 8; do
  echo "== queues $Q batch $B"
  RT_HW_QUEUES=$Q BATCH=$B DEPTHS=${DEPTHS:-2,4} timeout -k 10 120 python3 tools/shard_pipeline.py $N 6 || exit 1
done; done
