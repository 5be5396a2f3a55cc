#!/bin/bash
# usage (build container, after a gpurun of tools/final_evidence.sh): tools/store_evidence.sh  -> copies gpurun_out/final/* into profiles/r02_*
set -e
cd "$(dirname "$0")/.."
f=gpurun_out/final
cp $f/prof/runc_kernel_stats.csv profiles/r02_b_kernel_stats_0p70ms.csv
cp $f/bench.json profiles/r02_b_bench_0p70ms.json
cp $f/sq_counters.csv profiles/r02_b_sq_counters.csv
cp $f/sq_counters.csv profiles/r02_sq_counters.csv
cp $f/traffic.json profiles/r02_traffic.json
sed -i 's#gpurun_out/final/sq_counters.csv#profiles/r02_sq_counters.csv#' profiles/r02_traffic.json
cp $f/prof_chfak5/runc_kernel_stats.csv profiles/r02_generic_chfak5_kernel_stats.csv
cat $f/bench_chfak5_train.json $f/bench_chfak5_infer.json $f/bench_chfak5_infer_f16.json > profiles/r02_generic_chfak5_bench.json
cat $f/bench_infer2048.json $f/bench_infer2048_f16head.json $f/bench_infer2048_f16all.json $f/bench_cli_train.json $f/bench_phase1.json > profiles/r02_b_side_benches.json
echo stored
