#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
o=gpurun_out/r05e; mkdir -p $o
python -m pytest tests/test_gpu_engine.py -m gpu -x -q > $o/tests.log 2>&1; rc=$?; tail -5 $o/tests.log; [ $rc -ne 0 ] && { grep -B5 -A25 "Error\|assert" $o/tests.log | head -80; exit $rc; }
for i in 1 2 3; do python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused  ', round(d['ms_per_step'],4), round(d['roofline']['frac'],4), d['final_losses']['total'])"
CGS_ENC1_TAIL_BWD_FUSED=0 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('unfused', round(d['ms_per_step'],4), round(d['roofline']['frac'],4), d['final_losses']['total'])"; done
tools/prof.sh r05e/prof && python tools/kernel_stats.py gpurun_out/r05e/prof 0 14
