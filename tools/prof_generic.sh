#!/bin/bash
# usage (on the GPU box): tools/prof_generic.sh NAME [bench args...]  -> kernel stats of bench.py --chfak 5 under gpurun_out/NAME
name=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/$name -o runc -- python3 $root/bench.py --chfak 5 --steps 5 --warmup 1 --prime-s 0 "$@" > $root/gpurun_out/$name.log 2>&1
cd $root && python tools/kernel_stats.py gpurun_out/$name 7 40
