#!/bin/bash
# usage (GPU box): tools/ab_train.sh VARIANT...  -> headline bench with the product library and each libcgs_hip_VARIANT.so, interleaved twice
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
pkg=$(ls -d *_amd)
run() { python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],4), 'ms')"; }
for rep in $(seq 1 ${REPS:-2}); do
  unset CGS_LIB_PATH; run product
  for v in "$@"; do export CGS_LIB_PATH=$root/$pkg/libcgs_hip_$v.so; run $v; done
done
