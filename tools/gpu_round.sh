#!/bin/bash
# usage (on the GPU box): tools/gpu_round.sh NAME [pytest -k expression]  -> tests, bench lines (default, driver's command, 1-rank RCCL
# rehearsal), kernel stats under gpurun_out/
name=$1; kexpr=$2
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
if [ "$kexpr" = "none" ]; then echo "tests skipped";
elif [ -n "$kexpr" ]; then python -m pytest tests -m gpu -x -q -k "$kexpr" > gpurun_out/$name.tests.log 2>&1; rc=$?; tail -3 gpurun_out/$name.tests.log; [ $rc -ne 0 ] && exit $rc;
else python -m pytest tests -m gpu -x -q > gpurun_out/$name.tests.log 2>&1; rc=$?; tail -3 gpurun_out/$name.tests.log; [ $rc -ne 0 ] && exit $rc; fi
python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/$name.bench.json 2> gpurun_out/$name.bench.err || { tail -5 gpurun_out/$name.bench.err; exit 1; }
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/$name.bench_drv.json 2>> gpurun_out/$name.bench.err || exit 1
python - <<PY
import json
for f in ("bench", "bench_drv"):
    d = json.loads(open("gpurun_out/$name.%s.json" % f).read().strip().splitlines()[-1])
    print(f, "ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], "losses", d["final_losses"]["total"])
PY
tools/prof.sh $name.prof && python tools/kernel_stats.py gpurun_out/$name.prof 0 22 > gpurun_out/$name.kernels.txt && cat gpurun_out/$name.kernels.txt
