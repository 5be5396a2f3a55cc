#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
one() { label=$1; shift; env "$@" python bench.py --mode infer --batch 2048 --fp16 --steps 200 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', round(d['ms_per_step'],4), round(d['value']/1e6,2))"; }
for i in 1 2 3; do one h16tails CGS_F16_TAILS=1; one f32tails CGS_F16_TAILS=0; done
