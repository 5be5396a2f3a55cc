#!/bin/bash
one() { label=$1; shift; env "$@" python bench.py --mode infer --batch 2048 --steps 100 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', round(d['ms_per_step'],4), round(d['value']/1e6,3))"; }
for i in 1 2 3; do
  one mfma CGS_X=0
  one tile CGS_MASK_INFER_KERNEL=tile
done
