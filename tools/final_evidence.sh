#!/bin/bash
# usage (GPU box): tools/final_evidence.sh TAG [a|b|all]   -> everything the round's profiles/ files are made from, under gpurun_out/final/
# (a = smoke + counter passes + kernel profiles, b = the bench lines + scaling; two gpurun calls of <= 20 min each; b expects a's counter
#  summaries already stored under profiles/ by tools/store_evidence.sh TAG a)
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/final
tag=r06_${1:-x}
part=${2:-all}
mkdir -p $out
cd $root
if [ "$part" != "b" ]; then
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1 || { tail -5 $out/smoke.log; exit 1; }
tools/sq_counters.sh final/pmc || exit 1
python tools/sq_counters.py gpurun_out/final/pmc gpurun_out/final/sq_counters.csv && python tools/traffic_from_counters.py gpurun_out/final/sq_counters.csv gpurun_out/final/traffic.json 512
# counter traffic of the side workloads (FETCH_SIZE / WRITE_SIZE passes only)
tools/side_pmc.sh final/pmc_c4 --mode infer --fp16 --batch 2048 && python tools/side_traffic.py gpurun_out/final/pmc_c4 mask_infer config4_fp16_infer_batch2048 $out/side_traffic.json
tools/side_pmc.sh final/pmc_c5t --config 5 --mode train && python tools/side_traffic.py gpurun_out/final/pmc_c5t adam_kernel config5_train_batch256 $out/side_traffic.json
tools/side_pmc.sh final/pmc_c5i --config 5 --mode infer && python tools/side_traffic.py gpurun_out/final/pmc_c5i "tail_infer_h16_kernel" config5_infer_batch256 $out/side_traffic.json
# the counter summaries go into THIS copy's profiles/ before the bench lines are taken: bench.py reads the newest profiles/r*_traffic.json and says whether
# its source hash is the build's (tools/store_evidence.sh stores the same files in the build container afterwards)
cp $out/traffic.json profiles/${tag}_traffic.json && sed -i "s#gpurun_out/final/sq_counters.csv#profiles/${tag}_sq_counters.csv#" profiles/${tag}_traffic.json
cp $out/sq_counters.csv profiles/${tag}_sq_counters.csv
cp $out/side_traffic.json profiles/${tag}_side_traffic.json
tools/prof.sh final/prof || exit 1
tools/prof_generic.sh final/prof_chfak5 > $out/prof_chfak5.txt 2>&1 || exit 1
tools/prof_infer.sh final/prof_infer_f16 --fp16 > $out/prof_infer_f16.txt 2>&1 || exit 1
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_config5_train -o runc -- python3 $root/bench.py --config 5 --mode train --steps 10 --warmup 2 --prime-s 0 > $out/prof_config5_train.log 2>&1 ) || exit 1
# SQ counters of the chfak-5 step and of config 4
tools/sq_quick.sh final/sq_c5 --chfak 5 --mode train > $out/sq_c5.log 2>&1 && python tools/sq_any.py gpurun_out/final/sq_c5 > $out/sq_chfak5.txt 2>&1
tools/sq_quick.sh final/sq_c4 --mode infer --fp16 --batch 2048 > $out/sq_c4.log 2>&1 && python tools/sq_any.py gpurun_out/final/sq_c4 > $out/sq_config4.txt 2>&1
fi
if [ "$part" != "a" ]; then
# the default line (200 steps) and the driver's command, both with the side block and the CPU baseline
python bench.py > $out/bench.json 2> $out/bench.err || exit 1
python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_cmd.json 2>/dev/null || exit 1
python bench.py --force-pg --no-cpu-baseline > $out/bench_force_pg.json 2>/dev/null || exit 1
python bench.py --force-pg --dp-eager-allreduce --no-cpu-baseline > $out/bench_force_pg_eager.json 2>/dev/null || exit 1
python bench.py --chfak 5 --mode infer --steps 20 --warmup 3 > $out/bench_chfak5_infer.json 2>/dev/null || exit 1
python bench.py --chfak 5 --mode infer --fp16 --steps 20 --warmup 3 > $out/bench_chfak5_infer_f16.json 2>/dev/null || exit 1
python bench.py --mode infer --batch 2048 --steps 50 --warmup 5 > $out/bench_infer2048.json 2>/dev/null || exit 1
python bench.py --mode infer --batch 2048 --steps 50 --warmup 5 --fp16-mask-head > $out/bench_infer2048_f16head.json 2>/dev/null || exit 1
python bench.py --mode cli-train > $out/bench_cli_train.json 2>/dev/null || exit 1
python bench.py --mode phase1 > $out/bench_phase1.json 2>/dev/null || exit 1
# the driver's N > 1 launch form, rehearsed on ONE rank (RCCL group of one)
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_torchrun1.json 2>$out/bench_torchrun1.err || exit 1
python tools/scaling.py --out $out/scaling.json 128 256 512 1024 > $out/scaling.txt 2>&1 || exit 1
tail -1 $out/bench.json | cut -c1-300
fi
