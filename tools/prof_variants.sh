#!/bin/bash
# usage (GPU box): tools/prof_variants.sh NAME KERNEL_SUBSTRING VARIANT...  -> average duration of the matching kernels under the product library and each variant
name=$1; pat=$2; shift; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
pkg=$(ls -d *_amd)
one() {
  tools/prof.sh $name.$1 > /dev/null 2>&1
  python - <<PY
import csv, glob
f = glob.glob("gpurun_out/$name.$1/*kernel_stats.csv") + glob.glob("gpurun_out/$name.$1/*/*kernel_stats.csv")
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows if "at::" not in r["Name"] and "rocclr" not in r["Name"]) / 36e3
print("$1".ljust(10), " ".join(f'{r["Name"].replace("void ","")[:24]}={float(r["AverageNs"])/1e3:.1f}' for r in rows if "$pat" in r["Name"]), f"step sum {tot:.1f} us")
PY
}
unset CGS_LIB_PATH; one product
for v in "$@"; do export CGS_LIB_PATH=$root/$pkg/libcgs_hip_$v.so; one $v; done
unset CGS_LIB_PATH; one product2
