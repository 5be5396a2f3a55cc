"""Counter traffic of a SIDE workload (BASELINE configs 4 / 5): python tools/side_traffic.py DIR MARKER KEY OUT.json
DIR holds the two rocprofv3 --pmc passes fetch/ and write/ of one bench.py side command (tools/side_pmc.sh); MARKER = substring of a kernel
that runs exactly once per invocation of the workload (the divisor); KEY = the entry of bench.py's SIDE_RUNS this belongs to.  FETCH_SIZE
(KB) is doubled for the kernels whose reads are 16-byte-per-lane streaming loads (the gfx950 rule of MI355X_MICROARCH.md; the same list as
tools/sq_counters.py), WRITE_SIZE is exact.  The result is merged into OUT.json under KEY (bench.py's side block reads that file)."""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
root, marker, key, out = sys.argv[1:5]
STREAM16 = ("mask_head_kernel", "wgrad_kernel", "wgrad_any_kernel<WDec", "wgrad_any_kernel<WMask", "conv_bwd_both", "enc0_bwd_mix", "reduce_slabs",
            "reduce_adam", "tail_", "h5conv_kernel", "hwgrad_kernel", "mask_infer")


def load(sub, counter):
    f = max(glob.glob(f"{root}/{sub}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    per, calls = collections.defaultdict(float), collections.Counter()
    seen = set()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].replace("void ", "")
        if any(s in k for s in ("at::native", "__amd_rocclr", "ncclDevKernel")):
            continue
        per[k] += float(r["Counter_Value"]) * 1024.0
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            calls[k] += 1
    return per, calls


fe, calls = load("fetch", "FETCH_SIZE")
wr, _ = load("write", "WRITE_SIZE")
n_inv = sum(c for k, c in calls.items() if marker in k)
if n_inv == 0:
    raise SystemExit(f"marker {marker!r} matches no kernel: {list(calls)[:8]}")
raw = sum(fe.values()) / n_inv
cor = sum(v * (2.0 if any(s in k for s in STREAM16) else 1.0) for k, v in fe.items()) / n_inv
w = sum(wr.values()) / n_inv
import bench
entry = {"traffic_bytes_per_step": cor + w, "fetch_bytes_raw": raw, "fetch_bytes_corrected": cor, "write_bytes": w, "invocations": n_inv,
         "marker_kernel": marker, "csrc_sha16": bench.csrc_sha16(),
         "kernels": {k[:90]: {"calls_per_invocation": round(calls[k] / n_inv, 2), "fetch_raw": round(fe[k] / n_inv), "write": round(wr.get(k, 0.0) / n_inv)}
                     for k in sorted(fe, key=lambda k: -fe[k])[:12]},
         "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of the side command (tools/side_pmc.sh), KB x 1024, FETCH doubled for the "
                   "16-B/lane streaming kernels (gfx950 rule), divided by the calls of the once-per-invocation marker kernel"}
data = {}
if os.path.exists(out):
    data = json.load(open(out))
data[key] = entry
json.dump(data, open(out, "w"), indent=1)
print(key, json.dumps({k: v for k, v in entry.items() if k not in ("kernels", "method")}))
