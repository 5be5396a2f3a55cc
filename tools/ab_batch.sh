#!/bin/bash
# usage (GPU box): tools/ab_batch.sh VARIANT BATCH...   the headline step of the product library and libcgs_hip_VARIANT.so at several batch sizes (two rounds)
root=$PWD; pkg=$(ls -d *_amd); v=$1; shift
one() { label=$1; n=$2; shift 2; env "$@" python bench.py --batch $n --steps 300 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=$n $label', round(d['ms_per_step'],4))"; }
for n in "$@"; do for i in 1 2; do one prod $n CGS_X=0; one $v $n CGS_LIB_PATH=$root/$pkg/libcgs_hip_$v.so; done; done
