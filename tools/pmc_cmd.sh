#!/bin/bash
# usage (GPU box): tools/pmc_cmd.sh TAG script.py [args]  -> two SQ counter passes over `python3 script.py args`, per-kernel sums printed
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAIT_ANY --kernel-trace --output-format csv -d $root/gpurun_out/$tag/a -o pmc -- python3 $root/"$@" > $root/gpurun_out/$tag.a.log 2>&1 &&
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SALU --kernel-trace --output-format csv -d $root/gpurun_out/$tag/b -o pmc -- python3 $root/"$@" > $root/gpurun_out/$tag.b.log 2>&1
cd $root && python3 - <<PY
import csv, glob, collections
for sub in "ab":
    f = glob.glob("gpurun_out/$tag/%s/*counter_collection.csv" % sub) + glob.glob("gpurun_out/$tag/%s/*/*counter_collection.csv" % sub)
    if not f: print("no counter file for pass", sub); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"][:50]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"])
        if key not in seen: seen.add(key); calls[k] += 1
    for k, d in agg.items():
        if "wgrad" in k or "enc0" in k.lower():
            print(sub, k, "calls", calls[k], {c: round(v / calls[k]) for c, v in d.items()})
PY
