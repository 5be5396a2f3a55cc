#!/bin/bash
# usage (GPU box): tools/ab_infer.sh VARIANT...  -> bench.py --mode infer --batch 2048 --fp16 with the product library and each libcgs_hip_VARIANT.so
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
pkg=$(ls -d *_amd)
run() { python bench.py --mode infer --batch 2048 --steps 100 --warmup 10 --fp16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],4), 'ms', round(d['value']/1e6,2), 'M img/s')"; }
for rep in 1 2; do
  unset CGS_LIB_PATH; run product
  for v in "$@"; do export CGS_LIB_PATH=$root/$pkg/libcgs_hip_$v.so; run $v; done
done
