#!/bin/bash
# usage (build container, after a gpurun of tools/final_evidence.sh): tools/store_evidence.sh TAG  -> copies gpurun_out/final/* into
# profiles/r03_TAG_* (TAG e.g. "b")
set -e
cd "$(dirname "$0")/.."
f=gpurun_out/final
t=r03_${1:-b}
cp $f/prof/runc_kernel_stats.csv profiles/${t}_kernel_stats.csv
python tools/kernel_stats.py $f/prof 35 > profiles/${t}_kernels_per_step.txt
cp $f/bench.json profiles/${t}_bench.json
cp $f/bench_driver_cmd.json profiles/${t}_bench_driver_cmd.json
cp $f/sq_counters.csv profiles/${t}_sq_counters.csv
cp $f/traffic.json profiles/${t}_traffic.json
sed -i "s#gpurun_out/final/sq_counters.csv#profiles/${t}_sq_counters.csv#" profiles/${t}_traffic.json
cp $f/prof_chfak5/runc_kernel_stats.csv profiles/${t}_generic_chfak5_kernel_stats.csv
cat $f/bench_chfak5_train.json $f/bench_chfak5_infer.json $f/bench_chfak5_infer_f16.json > profiles/${t}_generic_chfak5_bench.json
cat $f/bench_infer2048.json $f/bench_infer2048_f16head.json $f/bench_infer2048_f16all.json $f/bench_config5.json $f/bench_cli_train.json $f/bench_phase1.json $f/bench_force_pg.json > profiles/${t}_side_benches.json
echo stored
