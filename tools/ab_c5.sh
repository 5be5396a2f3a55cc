#!/bin/bash
# usage (GPU box): tools/ab_c5.sh ROUNDS VARIANT...  interleaved `bench.py --chfak 5 --mode train` runs of the product library and libcgs_hip_VARIANT.so
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root; pkg=$(ls -d *_amd)
rounds=$1; shift
one() { label=$1; shift; env "$@" python bench.py --chfak 5 --mode train --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', round(d['ms_per_step'],4), round(d['roofline']['frac'],4))"; }
for i in $(seq 1 $rounds); do
  one prod CGS_X=0
  for v in "$@"; do one $v CGS_LIB_PATH=$root/$pkg/libcgs_hip_$v.so; done
done
