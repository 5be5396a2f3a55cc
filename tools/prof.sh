#!/bin/bash
# usage (on the GPU box): tools/prof.sh NAME [bench args...]   -> gpurun_out/NAME/runc/*_kernel_stats.csv
name=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/$name -o runc -- python3 $root/bench.py --steps 30 --warmup 5 --prime-s 0 --no-cpu-baseline "$@" > $root/gpurun_out/$name.log 2>&1
