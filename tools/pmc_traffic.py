"""Sums rocprofv3 FETCH_SIZE / WRITE_SIZE (two separate --pmc passes, MI355X_MICROARCH.md section HBM) over the
kernels of ONE training step and writes profiles/<tag>_traffic.{json,txt}.
gfx950 correction: FETCH_SIZE counts wide coalesced reads at half their bytes -> doubled; WRITE_SIZE is exact.
Usage: python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write r01 512"""
import collections, csv, glob, json, sys

def load(d, name):
    import os
    f = max(glob.glob(f"{d}/*/*_counter_collection.csv") + glob.glob(f"{d}/*counter_collection.csv"), key=os.path.getmtime)
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == name]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    idx = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
    a, b = idx[-2] + 1, idx[-1] + 1          # the kernels of the last complete step
    by = collections.OrderedDict()
    for r in rows[a:b]:
        k = r["Kernel_Name"].replace("void ", "")
        by[k] = by.get(k, 0.0) + float(r["Counter_Value"]) * 1024.0
    return by, b - a

fetch, nk = load(sys.argv[1], "FETCH_SIZE")
write, _ = load(sys.argv[2], "WRITE_SIZE")
tag, n = sys.argv[3], int(sys.argv[4])
tot_f = 2.0 * sum(fetch.values())
tot_w = sum(write.values())
out = {"n_images": n, "kernels_per_step": nk, "fetch_bytes_raw": sum(fetch.values()), "fetch_bytes_corrected_x2": tot_f,
       "write_bytes": tot_w, "traffic_bytes_per_step": tot_f + tot_w, "algorithmic_bytes_per_step": 4.841e6 * n,
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of bench.py --no-graph; values are KB; "
                 "FETCH_SIZE doubled (gfx950 counts 128-B requests at 64 B); summed over the kernels of one step"}
json.dump(out, open(f"profiles/{tag}_traffic.json", "w"), indent=1)
with open(f"profiles/{tag}_traffic.txt", "w") as fp:
    fp.write(json.dumps({k: v for k, v in out.items() if k != "method"}) + "\n")
    fp.write(f"{'kernel':64s} {'fetch x2 MB':>12s} {'write MB':>10s}\n")
    for k in sorted(fetch, key=lambda k: -(2 * fetch[k] + write.get(k, 0))):
        fp.write(f"{k[:64]:64s} {2 * fetch[k] / 1e6:12.1f} {write.get(k, 0) / 1e6:10.1f}\n")
print(json.dumps(out)[:400])
