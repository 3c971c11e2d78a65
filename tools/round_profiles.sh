#!/bin/bash
# usage (on the GPU box, through gpurun): bash tools/round_profiles.sh <outdir-name> [quick]
# Everything profiles/ holds for a round, from one box, in the order the figures depend on each other:
#   1. counter passes (tools/pmc.sh: five --pmc passes; tools/traffic.sh: FETCH_SIZE and WRITE_SIZE) over
#      tools/prof_frame.py, which renders the frame of the camera path that bench.py's roofline legs render;
#   2. tools/roofline_inputs.py turns them into r3_roofline_inputs.json (with the hash of the kernel sources), copied
#      into profiles/ of this tree so that the bench lines that follow use it;
#   3. the bench line, rocprofv3 kernel statistics of the same command and of a run whose kernels do not overlap;
#   4. (not with "quick") the other configurations, shard batches, tree builds, the multi-device context.
# Results: gpurun_out/<name>/ (copy what is to be kept into profiles/).
N=${1:-rfinal}; QUICK=$2; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$N
mkdir -p $O && cd $R || exit 1
bash tools/pmc.sh $N/pmc 3 || exit 1
bash tools/traffic.sh $N/traffic 3 || exit 1
python3 tools/pmc_report.py $N/pmc > $O/pmc_summary.txt || exit 1
python3 tools/traffic_report.py $N/traffic 3 > $O/traffic.json || exit 1
python3 tools/roofline_inputs.py $N/pmc $N/traffic > $O/r3_roofline_inputs.json || exit 1
cp $O/r3_roofline_inputs.json $R/profiles/r3_roofline_inputs.json || exit 1
timeout -k 10 300 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err || exit 1
timeout -k 10 300 python bench.py --still --cpu-seconds 3 > $O/bench_n1_still.json 2> $O/bench_n1_still.err || exit 1
(cd /tmp && export TMPDIR=/tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_bench -o kt -- python3 $R/bench.py --cpu-seconds 2 > $O/bench_n1_under_rocprof.json 2> $O/kt_bench.err) || exit 1
(cd /tmp && export TMPDIR=/tmp && RTGPU_SERIAL=1 timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_serial -o kt -- python3 $R/bench.py --frames-in-flight 1 --batch 1 --cpu-seconds 2 > $O/bench_n1_serial_under_rocprof.json 2> $O/kt_serial.err) || exit 1
if [ "$QUICK" = "quick" ]; then echo profiles-done-quick; exit 0; fi
for C in 1 2 3 3b 5a 5b; do timeout -k 10 300 python bench.py --config $C --cpu-seconds 5 > $O/bench_config$C.json 2> $O/bench_config$C.err || exit 1; done
for S in 0 3 5; do SHARD=$S BATCH=8 DEPTHS=2 RT_HW_QUEUES=4 timeout -k 10 120 python3 tools/shard_pipeline.py 8 6 2>&1 | grep shard >> $O/shard_pipeline.txt || exit 1; done
for cfg in "8 1 8" "4 4 2" "2 2 2" "1 2 3" "1 1 4"; do set -- $cfg; BATCH=$2 DEPTHS=$3 RT_HW_QUEUES=4 timeout -k 10 120 python3 tools/shard_pipeline.py $1 6 2>&1 | grep shard >> $O/shard_pipeline.txt || exit 1; done
timeout -k 10 300 python tools/build_time.py 2>&1 | grep tree > $O/build_time.txt || exit 1
timeout -k 10 200 python tools/multi_bench.py 1 8 3 > $O/multi_bench.txt 2> $O/multi_bench.err || exit 1
DEVICES=0,0 timeout -k 10 200 python tools/multi_bench.py 2 8 3 >> $O/multi_bench.txt 2>> $O/multi_bench.err || exit 1
DEVICES=0,0,0,0,0,0,0,0 timeout -k 10 200 python tools/multi_bench.py 8 8 3 >> $O/multi_bench.txt 2>> $O/multi_bench.err || exit 1
RTGPU_MULTI_FORCE=1 timeout -k 10 200 python tools/multi_bench.py 1 8 3 >> $O/multi_bench.txt 2>> $O/multi_bench.err || exit 1
echo profiles-done
