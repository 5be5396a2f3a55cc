"""Per-kernel mean of rocprofv3 --pmc counters.  Usage: python tools/pmc_kernel.py DIR [name-substring]"""
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/*counter_collection.csv") + glob.glob(sys.argv[1] + "/*/*counter_collection.csv"))[-1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if sub in k:
        acc[k.replace("void ", "")[:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
