#!/bin/bash
# usage (GPU box): tools/prof_c5.sh NAME [train|infer]  -> per-launch timeline + kernel stats of one config-5 step under gpurun_out/NAME
name=$1; mode=${2:-train}
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/$name -o c5 -- python3 $root/bench.py --config 5 --mode $mode --no-cpu-baseline --steps 40 --warmup 5 > $root/gpurun_out/$name.json 2> $root/gpurun_out/$name.err || exit 1
cd $root && python tools/c5_timeline.py gpurun_out/$name/c5_kernel_trace.csv > gpurun_out/$name.timeline.txt
grep -h '^{' gpurun_out/$name.json | cut -c1-260
