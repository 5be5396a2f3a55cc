#!/bin/bash
# usage (GPU box): tools/prof_g5.sh NAME [VARIANT]  -> rocprofv3 kernel stats of the chfak-5 training step (product library or libcgs_hip_VARIANT.so) under gpurun_out/NAME
name=$1; v=$2
root=${GRAFT_REPO_ROOT:-/root/repo}; pkg=$(cd $root; ls -d *_amd)
[ -n "$v" ] && export CGS_LIB_PATH=$root/$pkg/libcgs_hip_$v.so
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/$name -o g5 -- python3 $root/bench.py --chfak 5 --mode train --no-cpu-baseline --steps 10 --warmup 3 > $root/gpurun_out/$name.json 2> $root/gpurun_out/$name.err || exit 1
cd $root && python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/$name/g5_kernel_stats.csv")))
for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"]))[:26]:
    print(f'{r["Name"][:70]:70s} {int(r["Calls"]):5d} {float(r["AverageNs"])/1e3:9.1f} {float(r["TotalDurationNs"])/1e3:10.0f}')
PY
