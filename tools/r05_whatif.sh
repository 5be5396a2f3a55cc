#!/bin/bash
one() { label=$1; shift; python bench.py --steps 300 --warmup 20 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', round(d['ms_per_step'],4))"; }
for i in 1 2 3; do
  one drop0.3
  one drop0.0 --dropout 0.0
done
