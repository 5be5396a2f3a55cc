"""Compact per-kernel table of the two SQ counter passes of tools/sq_quick.sh for ANY bench command (no step marker needed):
  python tools/sq_table.py gpurun_out/TAG [min_us_per_dispatch]
columns: dispatches, mean us per dispatch (kernel trace of pass sq1), matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES)
(the two counters come from different passes of the same command), wait_any / wait_inst = SQ_WAIT_* / SQ_WAVE_CYCLES, LDS bank-conflict
ratio = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE."""
import collections, csv, glob, os, sys
root = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0


def norm(k):
    return k.replace("void ", "").replace("(anonymous namespace)::", "")


agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for sub in ("sq1", "sq2"):
    found = glob.glob(f"{root}/{sub}/**/*counter_collection.csv", recursive=True)
    if not found:
        continue
    for r in csv.DictReader(open(max(found, key=os.path.getmtime))):
        k = norm(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[(k, sub)].add(r["Dispatch_Id"])
dur = collections.defaultdict(list)
tf = glob.glob(f"{root}/sq1/**/*kernel_trace.csv", recursive=True)
if tf:
    for r in csv.DictReader(open(max(tf, key=os.path.getmtime))):
        dur[norm(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = []
for k, c in agg.items():
    n = len(disp[(k, "sq1")]) or len(disp[(k, "sq2")])
    us = sum(dur[k]) / len(dur[k]) if dur[k] else 0.0
    if us < min_us:
        continue
    wc, bc = max(c["SQ_WAVE_CYCLES"], 1.0), max(c["SQ_BUSY_CU_CYCLES"], 1.0)
    rows.append((us * n, k, n, us, c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * bc), c["SQ_WAIT_ANY"] / wc, c["SQ_WAIT_INST_ANY"] / wc,
                 c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1.0)))
print(f"{'kernel':72s} {'calls':>5s} {'us':>8s} {'mfma':>5s} {'w_any':>5s} {'w_inst':>6s} {'ldsconf':>7s}")
for _, k, n, us, mf, wa, wi, lc in sorted(rows, reverse=True):
    print(f"{k[:72]:72s} {n:5d} {us:8.1f} {mf:5.2f} {wa:5.2f} {wi:6.2f} {lc:7.2f}")
