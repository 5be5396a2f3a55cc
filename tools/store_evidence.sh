#!/bin/bash
# usage (build container, after a gpurun of tools/final_evidence.sh): tools/store_evidence.sh TAG  -> copies gpurun_out/final/* into
# profiles/r05_TAG_* (TAG e.g. "b")
set -e
cd "$(dirname "$0")/.."
f=gpurun_out/final
t=r05_${1:-b}
ms=$(python -c "import json;print(json.loads(open('$f/bench.json').read().strip().splitlines()[-1])['ms_per_step'])")
cp $f/prof/runc_kernel_stats.csv profiles/${t}_kernel_stats.csv
python tools/kernel_stats.py $f/prof 0 30 $ms > profiles/${t}_kernels_per_step.txt
tail -1 $f/bench.json > profiles/${t}_bench.json
tail -1 $f/bench_driver_cmd.json > profiles/${t}_bench_driver_cmd.json
cp $f/sq_counters.csv profiles/${t}_sq_counters.csv
cp $f/traffic.json profiles/${t}_traffic.json
sed -i "s#gpurun_out/final/sq_counters.csv#profiles/${t}_sq_counters.csv#" profiles/${t}_traffic.json
[ -f $f/side_traffic.json ] && cp $f/side_traffic.json profiles/${t}_side_traffic.json
cp $f/scaling.json profiles/${t}_scaling.json
cp $f/prof_chfak5/runc_kernel_stats.csv profiles/${t}_generic_chfak5_kernel_stats.csv
cp $f/prof_infer_f16/runc_kernel_stats.csv profiles/${t}_infer2048_fp16_kernel_stats.csv
cp $f/prof_config5_train/runc_kernel_stats.csv profiles/${t}_config5_train_kernel_stats.csv 2>/dev/null || cp $f/prof_config5_train/*/runc_kernel_stats.csv profiles/${t}_config5_train_kernel_stats.csv
grep -h "^{" $f/bench_chfak5_infer.json $f/bench_chfak5_infer_f16.json $f/bench_infer2048.json $f/bench_infer2048_f16head.json $f/bench_cli_train.json $f/bench_phase1.json $f/bench_force_pg.json $f/bench_force_pg_eager.json > profiles/${t}_side_benches.json
echo stored
