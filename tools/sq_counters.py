"""Per-kernel table of the rocprofv3 --pmc passes collected by tools/sq_counters.sh (one training step, eager launches).
Usage: python tools/sq_counters.py gpurun_out/TAG profiles/rNN_sq_counters.csv
Columns: calls per step, mean duration (from the same pass's kernel trace), SQ counters summed over the step's launches of that
kernel, FETCH_SIZE / WRITE_SIZE in bytes (KB counters x 1024; FETCH_SIZE raw -- the gfx950 x2 rule applies to 16-B/lane streaming
reads only, see the column `fetch_x2_applies`)."""
import collections, csv, glob, os, sys

root, out = sys.argv[1], sys.argv[2]


def load(sub):
    found = glob.glob(f"{root}/{sub}/**/*counter_collection.csv", recursive=True)
    if not found:          # pass not collected (tools/sq_quick.sh runs the two SQ passes only)
        return {}
    f = max(found, key=os.path.getmtime)
    rows = list(csv.DictReader(open(f)))
    disp = collections.OrderedDict()
    for r in rows:
        d = disp.setdefault(int(r["Dispatch_Id"]), {"k": r["Kernel_Name"].replace("void ", ""), "c": {}})
        d["c"][r["Counter_Name"]] = d["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    ids = sorted(disp)
    adam = [i for i in ids if "adam_kernel" in disp[i]["k"]]
    lo, hi = adam[-2], adam[-1]                       # the launches of the last complete step
    step = [disp[i] for i in ids if lo < i <= hi]
    dur = {}
    tf = glob.glob(f"{root}/{sub}/**/*kernel_trace.csv", recursive=True)
    if tf:
        for r in csv.DictReader(open(max(tf, key=os.path.getmtime))):
            i = int(r["Dispatch_Id"])
            if lo < i <= hi:
                dur[i] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg = collections.OrderedDict()
    for i in ids:
        if not (lo < i <= hi):
            continue
        d = disp[i]
        a = agg.setdefault(d["k"], {"calls": 0, "us": 0.0, "c": collections.defaultdict(float)})
        a["calls"] += 1
        a["us"] += dur.get(i, 0.0)
        for k, v in d["c"].items():
            a["c"][k] += v
    return agg


sq1, sq2, fe, wr = load("sq1"), load("sq2"), load("fetch"), load("write")
cols1 = ["SQ_WAVES", "SQ_BUSY_CU_CYCLES", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_INSTS_VALU"]
cols2 = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_LDS", "SQ_WAIT_INST_LDS", "GRBM_GUI_ACTIVE"]
STREAM16 = ("mask_head_kernel", "mask0_fwd", "FMask2", "wgrad_kernel", "wgrad_any_kernel<WDec", "wgrad_any_kernel<WMask", "conv_bwd_both", "enc0_bwd_mix", "FEnc1", "FDec", "DDec", "reduce_slabs", "reduce_adam", "tail_",
            "h5conv_kernel", "hwgrad_kernel")      # (config 5: 16-byte staging loads; the uint8 frames of three of them are 3 of ~39 B / pixel)
with open(out, "w", newline="") as fp:
    w = csv.writer(fp)
    w.writerow(["kernel", "calls_per_step", "us_per_step"] + cols1 + cols2 + ["FETCH_bytes_raw", "WRITE_bytes", "fetch_x2_applies",
               "valu_active_frac_of_wave_cycles", "wait_inst_frac", "wait_any_frac"])
    for k, a in sorted(sq1.items(), key=lambda kv: -kv[1]["us"]):
        c1, c2 = a["c"], sq2.get(k, {"c": {}})["c"]
        wc = max(c1.get("SQ_WAVE_CYCLES", 0.0), 1.0)
        w.writerow([k.replace("(anonymous namespace)::", "")[:150], a["calls"], round(a["us"], 1)] + [int(c1.get(c, 0)) for c in cols1] + [int(c2.get(c, 0)) for c in cols2] +
                   [int(fe.get(k, {"c": {}})["c"].get("FETCH_SIZE", 0) * 1024), int(wr.get(k, {"c": {}})["c"].get("WRITE_SIZE", 0) * 1024),
                    int(any(s in k for s in STREAM16)),
                    round(c1.get("SQ_ACTIVE_INST_VALU", 0) / wc, 3), round(c1.get("SQ_WAIT_INST_ANY", 0) / wc, 3),
                    round(c1.get("SQ_WAIT_ANY", 0) / wc, 3)])
print("wrote", out)
