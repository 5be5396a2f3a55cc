#!/bin/bash
# usage (build container, after a gpurun of tools/final_evidence.sh TAG a|b): tools/store_evidence.sh TAG a|b  -> copies gpurun_out/final/* into
# profiles/r06_TAG_* (a: counters + kernel profiles -- stored BEFORE part b runs, bench.py reads the newest profiles/r*_traffic.json; b: bench lines)
set -e
cd "$(dirname "$0")/.."
f=gpurun_out/final
t=r06_${1:-a}
part=${2:-a}
if [ "$part" = "a" ]; then
  cp $f/prof/runc_kernel_stats.csv profiles/${t}_kernel_stats.csv
  cp $f/sq_counters.csv profiles/${t}_sq_counters.csv
  cp $f/traffic.json profiles/${t}_traffic.json
  sed -i "s#gpurun_out/final/sq_counters.csv#profiles/${t}_sq_counters.csv#" profiles/${t}_traffic.json
  [ -f $f/side_traffic.json ] && cp $f/side_traffic.json profiles/${t}_side_traffic.json
  cp $f/prof_chfak5/runc_kernel_stats.csv profiles/${t}_generic_chfak5_kernel_stats.csv
  cp $f/prof_infer_f16/runc_kernel_stats.csv profiles/${t}_infer2048_fp16_kernel_stats.csv
  cp $f/prof_config5_train/runc_kernel_stats.csv profiles/${t}_config5_train_kernel_stats.csv 2>/dev/null || cp $f/prof_config5_train/*/runc_kernel_stats.csv profiles/${t}_config5_train_kernel_stats.csv
  cp $f/sq_chfak5.txt profiles/${t}_generic_chfak5_sq_counters.txt
  cp $f/sq_config4.txt profiles/${t}_infer2048_fp16_sq_counters.txt
else
  ms=$(python -c "import json;print(json.loads(open('$f/bench.json').read().strip().splitlines()[-1])['ms_per_step'])")
  python tools/kernel_stats.py $f/prof 0 30 $ms > profiles/${t}_kernels_per_step.txt
  tail -1 $f/bench.json > profiles/${t}_bench.json
  tail -1 $f/bench_driver_cmd.json > profiles/${t}_bench_driver_cmd.json
  tail -1 $f/bench_torchrun1.json > profiles/${t}_bench_torchrun_1rank.json
  cp $f/scaling.json profiles/${t}_scaling.json
  grep -h "^{" $f/bench_chfak5_infer.json $f/bench_chfak5_infer_f16.json $f/bench_infer2048.json $f/bench_infer2048_f16head.json $f/bench_cli_train.json $f/bench_phase1.json $f/bench_force_pg.json $f/bench_force_pg_eager.json > profiles/${t}_side_benches.json
fi
echo stored
