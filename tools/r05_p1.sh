#!/bin/bash
one() { label=$1; shift; env "$@" python bench.py --mode phase1 --steps 300 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', round(d['ms_per_step'],4), round(d['value']/1e6,3))"; }
for i in 1 2 3; do
  one fused CGS_X=0
  one old CGS_PHASE1_FUSED_TAIL=0
done
