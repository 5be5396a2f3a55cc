#!/bin/bash
# usage (GPU box): tools/sweep_sparse.sh  -> ms/step for a few workgroup caps of the sparse weight-gradient kernels
cd ${GRAFT_REPO_ROOT:-/root/repo}
for b0 in 384 512 768 1024; do for b1 in 192 256 384; do
  r=$(CGS_SPARSE_BOTH0_BLOCKS=$b0 CGS_SPARSE_BOTH_BLOCKS=$b1 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "enc0-both $b0 enc1-both $b1: $r ms"
done; done
