#!/bin/bash
# usage (GPU box): tools/side_pmc.sh TAG [bench args]   two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over one side command of bench.py
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $root/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
run() {
  rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $root/gpurun_out/$tag/$1 -o pmc -- python3 $root/bench.py --steps 3 --warmup 2 --prime-s 0 --no-cpu-baseline --no-graph "${@:3}" > $root/gpurun_out/$tag.$1.log 2>&1
}
run fetch "FETCH_SIZE" "$@" && run write "WRITE_SIZE" "$@"
