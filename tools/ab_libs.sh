#!/bin/bash
# usage (GPU box): tools/ab_libs.sh ROUNDS VARIANT...   interleaved headline runs (bench.py --steps 300) of the product library and libcgs_hip_VARIANT.so (tools/build_variant.py / build_rev.py)
root=$PWD; pkg=$(ls -d *_amd)
one() { label=$1; shift; env "$@" python bench.py --steps 300 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', round(d['ms_per_step'],4))"; }
rounds=$1; shift
for i in $(seq 1 $rounds); do
  one prod CGS_X=0
  for v in "$@"; do one $v CGS_LIB_PATH=$root/$pkg/libcgs_hip_$v.so; done
done
