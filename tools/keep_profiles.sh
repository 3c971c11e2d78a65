#!/bin/bash
# usage (in the development container): bash tools/keep_profiles.sh <outdir-name> <round-prefix>
# Copies what tools/round_profiles.sh left under gpurun_out/<name>/ into profiles/ under the round's names.
N=$1; P=${2:-r3}; S=gpurun_out/$N; D=profiles
set -e
jsonline() { grep -m1 '^{' "$1" > "$2"; }
jsonline $S/bench_n1.json $D/${P}_bench_n1.json
jsonline $S/bench_n1_still.json $D/${P}_bench_n1_still.json
jsonline $S/bench_n1_under_rocprof.json $D/${P}_bench_n1_under_rocprof.json
jsonline $S/bench_n1_serial_under_rocprof.json $D/${P}_bench_n1_serial_under_rocprof.json
cp $S/kt_bench/kt_kernel_stats.csv $D/${P}_kernel_stats.csv
cp $S/kt_serial/kt_kernel_stats.csv $D/${P}_kernel_stats_serial.csv
cp $S/pmc_summary.txt $D/${P}_pmc_summary.txt
cp $S/traffic.json $D/${P}_traffic.json
cp $S/r3_roofline_inputs.json $D/r3_roofline_inputs.json
for C in 1 2 3 3b 5a 5b; do [ -f $S/bench_config$C.json ] && jsonline $S/bench_config$C.json $D/${P}_bench_config$C.json; done
for F in build_time shard_pipeline; do [ -f $S/$F.txt ] && cp $S/$F.txt $D/${P}_$F.txt; done
[ -f $S/multi_bench.txt ] && grep '^{' $S/multi_bench.txt > $D/${P}_multi_bench.txt
echo kept
