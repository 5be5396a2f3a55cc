#!/bin/bash
# usage (on the GPU box): tools/gpu_round.sh NAME [pytest -k expression]  -> tests, one bench line, kernel stats under gpurun_out/
name=$1; kexpr=$2
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
if [ -n "$kexpr" ]; then python -m pytest tests -m gpu -x -q -k "$kexpr" > gpurun_out/$name.tests.log 2>&1; else python -m pytest tests -m gpu -x -q > gpurun_out/$name.tests.log 2>&1; fi
rc=$?
tail -3 gpurun_out/$name.tests.log
[ $rc -ne 0 ] && exit $rc
python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/$name.bench.json 2> gpurun_out/$name.bench.err || exit 1
python - <<PY
import json
d = json.loads(open("gpurun_out/$name.bench.json").read().strip().splitlines()[-1])
print("ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"])
PY
tools/prof.sh $name.prof && python tools/kernel_stats.py gpurun_out/$name.prof > gpurun_out/$name.kernels.txt && cat gpurun_out/$name.kernels.txt
