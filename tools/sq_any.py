"""Per-kernel means of the two SQ passes of tools/sq_counters.sh for ANY bench command (no step marker needed: averages over all
dispatches of a kernel).  Usage: python tools/sq_any.py gpurun_out/TAG [name-substring ...]"""
import collections, csv, glob, os, sys
root, pats = sys.argv[1], sys.argv[2:]
for sub in ("sq1", "sq2"):
    found = glob.glob(f"{root}/{sub}/**/*counter_collection.csv", recursive=True)
    if not found:
        continue
    agg = collections.OrderedDict()
    seen = collections.defaultdict(set)
    for r in csv.DictReader(open(max(found, key=os.path.getmtime))):
        k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")[:90]
        if pats and not any(p in k for p in pats):
            continue
        a = agg.setdefault(k, collections.defaultdict(float))
        a[r["Counter_Name"]] += float(r["Counter_Value"])
        seen[k].add(r["Dispatch_Id"])
        a["_vgpr"] = float(r.get("VGPR_Count", 0) or 0); a["_lds"] = float(r.get("LDS_Block_Size", 0) or 0)
    for k, a in agg.items():
        n = len(seen[k])
        print(f"== {k}  dispatches={n} vgpr={a['_vgpr']:.0f} lds={a['_lds']:.0f}")
        for c, v in a.items():
            if not c.startswith("_"):
                print(f"   {c:34s} {v / n:16.0f}")
