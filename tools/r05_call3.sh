#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
o=gpurun_out/r05d; mkdir -p $o
python -m pytest tests/test_gpu_modules.py -m gpu -x -q -s -k "side_block" > $o/tests.log 2>&1; rc=$?; grep "^side\|passed\|failed\|^E " $o/tests.log | tail; [ $rc -ne 0 ] && { tail -30 $o/tests.log; exit $rc; }
( time python bench.py --steps 20 --warmup 5 > $o/bench_drv_full.json 2> $o/bench_drv_full.err ) 2> $o/time.txt; cat $o/time.txt; python - <<PY
import json
d=json.loads(open("$o/bench_drv_full.json").read().strip().splitlines()[-1])
print("headline", d["ms_per_step"], d["roofline"]["frac"])
for k,v in d["side"].items(): print(k, v.get("value"), v.get("ms_per_step"), v.get("wall_s"), v.get("error"))
PY
