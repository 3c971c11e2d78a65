#!/bin/bash
# usage (on the GPU box, through gpurun): bash tools/round_profiles.sh <outdir-name>
# Everything profiles/ holds for a round, from one box: the bench line, rocprofv3 kernel statistics of the same command
# and of a run whose kernels do not overlap, counter passes, traffic, the other configurations, shard batches, tree builds.
N=${1:-rfinal}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$N
mkdir -p $O && cd $R || exit 1
timeout -k 10 300 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err || exit 1
for C in 2 3 3b 5a 5b; do timeout -k 10 300 python bench.py --config $C --cpu-seconds 5 > $O/bench_config$C.json 2> $O/bench_config$C.err || exit 1; done
(cd /tmp && export TMPDIR=/tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_bench -o kt -- python3 $R/bench.py --cpu-seconds 2 > $O/bench_n1_under_rocprof.json 2> $O/kt_bench.err) || exit 1
(cd /tmp && export TMPDIR=/tmp && RTGPU_SERIAL=1 timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_serial -o kt -- python3 $R/bench.py --frames-in-flight 1 --batch 1 --cpu-seconds 2 > $O/bench_n1_serial_under_rocprof.json 2> $O/kt_serial.err) || exit 1
bash tools/pmc.sh $N/pmc 3 || exit 1
bash tools/traffic.sh $N/traffic 3 || exit 1
for S in 0 3 5; do SHARD=$S BATCH=8 DEPTHS=2 RT_HW_QUEUES=4 timeout -k 10 120 python3 tools/shard_pipeline.py 8 6 2>&1 | grep shard >> $O/shard_pipeline.txt || exit 1; done
for cfg in "8 1 8" "4 4 2" "2 2 2" "1 2 3" "1 1 4"; do set -- $cfg; BATCH=$2 DEPTHS=$3 RT_HW_QUEUES=4 timeout -k 10 120 python3 tools/shard_pipeline.py $1 6 2>&1 | grep shard >> $O/shard_pipeline.txt || exit 1; done
timeout -k 10 300 python tools/build_time.py 2>&1 | grep tree > $O/build_time.txt || exit 1
echo profiles-done
