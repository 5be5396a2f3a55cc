"""Per-launch timeline of the LAST headline step in a rocprofv3 kernel trace: python tools/step_timeline.py TRACE.csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "reduce_adam" in r["Kernel_Name"]]
a, b = idx[-2] + 1, idx[-1] + 1
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f} "
          f"grid={int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']):5d}x{r['Workgroup_Size_X']} vgpr={r['VGPR_Count']:>3s} {r['Kernel_Name'][:80]}")
print(f"span {(int(rows[b - 1]['End_Timestamp']) - t0) / 1e3:.1f} us")
