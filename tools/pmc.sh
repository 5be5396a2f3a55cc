#!/bin/bash
# usage (GPU box): tools/pmc.sh NAME "COUNTER1 COUNTER2 ..." [bench args]  -> gpurun_out/NAME/*counter_collection.csv
name=$1; ctrs=$2; shift; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $root/gpurun_out/$name -o pmc -- python3 $root/bench.py --steps ${PMC_STEPS:-3} --warmup 2 --no-cpu-baseline --no-graph "$@" > $root/gpurun_out/$name.log 2>&1
