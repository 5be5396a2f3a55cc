#!/bin/bash
# usage (GPU box): tools/final_evidence.sh   -> everything the round's profiles/ files are made from, under gpurun_out/final/
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/final
mkdir -p $out
cd $root
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1 || { tail -5 $out/smoke.log; exit 1; }
python bench.py > $out/bench.json 2> $out/bench.err || exit 1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_driver_cmd.json 2>/dev/null || exit 1
python bench.py --force-pg --no-cpu-baseline > $out/bench_force_pg.json 2>/dev/null || exit 1
python bench.py --force-pg --dp-eager-allreduce --no-cpu-baseline > $out/bench_force_pg_eager.json 2>/dev/null || exit 1
python bench.py --chfak 5 --steps 20 --warmup 3 > $out/bench_chfak5_train.json 2>/dev/null || exit 1
python bench.py --chfak 5 --mode infer --steps 20 --warmup 3 > $out/bench_chfak5_infer.json 2>/dev/null || exit 1
python bench.py --chfak 5 --mode infer --fp16 --steps 20 --warmup 3 > $out/bench_chfak5_infer_f16.json 2>/dev/null || exit 1
python bench.py --mode infer --batch 2048 --steps 50 --warmup 5 > $out/bench_infer2048.json 2>/dev/null || exit 1
python bench.py --mode infer --batch 2048 --steps 50 --warmup 5 --fp16-mask-head > $out/bench_infer2048_f16head.json 2>/dev/null || exit 1
python bench.py --mode infer --batch 2048 --steps 50 --warmup 5 --fp16 > $out/bench_infer2048_f16.json 2>/dev/null || exit 1
python bench.py --config 5 --mode infer > $out/bench_config5.json 2>/dev/null || exit 1
python bench.py --config 5 --mode train --steps 50 --warmup 5 > $out/bench_config5_train.json 2>/dev/null || exit 1
python bench.py --mode cli-train > $out/bench_cli_train.json 2>/dev/null || exit 1
python bench.py --mode phase1 > $out/bench_phase1.json 2>/dev/null || exit 1
tools/prof.sh final/prof || exit 1
tools/prof_generic.sh final/prof_chfak5 > $out/prof_chfak5.txt 2>&1 || exit 1
tools/prof_infer.sh final/prof_infer_f16 --fp16 > $out/prof_infer_f16.txt 2>&1 || exit 1
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_config5_train -o runc -- python3 $root/bench.py --config 5 --mode train --steps 10 --warmup 2 > $out/prof_config5_train.log 2>&1 ) || exit 1
tools/sq_counters.sh final/pmc || exit 1
tools/sq_counters.sh final/pmc_c5 --config 5 --mode train || exit 1
python tools/sq_counters.py gpurun_out/final/pmc_c5 gpurun_out/final/sq_counters_config5.csv
python tools/sq_counters.py gpurun_out/final/pmc gpurun_out/final/sq_counters.csv && python tools/traffic_from_counters.py gpurun_out/final/sq_counters.csv gpurun_out/final/traffic.json 512
tail -1 $out/bench.json | cut -c1-300
