"""Per-launch timeline of the LAST optimiser step in a rocprofv3 kernel trace: tools/trace_step.py DIR [min_us]"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/*kernel_trace.csv") + glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[-1]
mn = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith(("adam_kernel", "reduce_adam_kernel"))]
a, b = idx[-2], idx[-1]
tot = 0.0
for r in rows[a + 1:b + 1]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    if d > mn:
        nm = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:44]
        print(f"{nm:44s} {d:9.1f} us  wgs={int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']):7d} lds={r['LDS_Block_Size']:>6s} vgpr={r['VGPR_Count']}+{r['Accum_VGPR_Count']}")
print(f"sum of kernel durations in the step: {tot:.1f} us")
