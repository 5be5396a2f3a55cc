#!/bin/bash
# usage (GPU box): tools/sweep_sparse.sh  -> ms/step for a few workgroup caps of the sparse weight-gradient kernels
cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "384 192" "512 256" "768 256" "768 384" "1024 384" "1024 512" "1536 512" "2048 768"; do
  set -- $cfg
  r=$(CGS_SPARSE_BOTH0_BLOCKS=$1 CGS_SPARSE_BOTH_BLOCKS=$2 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "enc0-both $1 enc1-both $2: $r ms"
done
