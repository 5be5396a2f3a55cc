#!/bin/bash
# usage (GPU box): tools/prof_ab.sh NAME VARIANT -> per-kernel average durations (rocprofv3 --stats, 36 steps each) of the product library and
# libcgs_hip_VARIANT.so side by side
name=$1; var=$2
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
pkg=$(ls -d *_amd)
tools/prof.sh $name.a || exit 1
export CGS_LIB_PATH=$root/$pkg/libcgs_hip_$var.so
tools/prof.sh $name.b || exit 1
python - <<PY
import csv, glob
def load(d):
    f = glob.glob("gpurun_out/%s/*kernel_stats.csv" % d) + glob.glob("gpurun_out/%s/*/*kernel_stats.csv" % d)
    return {r["Name"].replace("void ", "")[:48]: (float(r["AverageNs"]) / 1e3, int(r["Calls"])) for r in csv.DictReader(open(f[0]))}
a, b = load("$name.a"), load("$name.b")
ta = tb = 0.0
for k in sorted(a, key=lambda k: -a[k][0] * a[k][1]):
    if k in b and "at::" not in k and "rocclr" not in k:
        ta += a[k][0] * a[k][1] / 36; tb += b[k][0] * b[k][1] / 36
        print(f"{k:48s} product {a[k][0]:7.1f} us   $var {b[k][0]:7.1f} us   {b[k][0]-a[k][0]:+6.1f}  x{a[k][1]//36}")
print(f"per-step sums: product {ta:.1f} us, $var {tb:.1f} us")
PY
