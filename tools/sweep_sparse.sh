#!/bin/bash
# usage (GPU box): tools/sweep_sparse.sh  -> ms/step for a few workgroup caps of the sparse weight-gradient kernels
cd ${GRAFT_REPO_ROOT:-/root/repo}
for b1 in 96 128 192 256 320; do
  r=$(CGS_SPARSE_BOTH_BLOCKS=$b1 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "enc1-both $b1: $r ms"
done
