#!/bin/bash
one() { label=$1; shift; env "$@" python bench.py --mode cli-train 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', round(d['value']))"; }
for i in 1 2 3; do
  one one_launch CGS_X=0
  one five CGS_GATHER_ONE_LAUNCH=0
done
