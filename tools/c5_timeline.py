"""Per-launch timeline of the LAST step in a rocprofv3 kernel trace of bench.py --config 5: python tools/c5_timeline.py TRACE.csv
(start us, duration us, grid, kernel) + the sum per kernel name."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
marks = [i for i, n in enumerate(names) if n.startswith("adam_kernel")]
if len(marks) >= 2:
    a, b = marks[-2] + 1, marks[-1] + 1
else:       # inference: one step = the kernels between two launches of the first kernel of the pass
    first = names[-1]
    ends = [i for i, n in enumerate(names) if n == first]
    a, b = ends[-2] + 1, ends[-1] + 1
t0 = int(rows[a]["Start_Timestamp"])
per = collections.OrderedDict()
for r in rows[a:b]:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} {d:7.1f} grid={r['Grid_Size_X']:>9s} {n[:90]}")
    k = n[:60]
    per[k] = (per.get(k, (0, 0.0))[0] + 1, per.get(k, (0, 0.0))[1] + d)
span = (int(rows[b - 1]["End_Timestamp"]) - t0) / 1e3
print(f"span {span:.1f} us, kernel sum {sum(v[1] for v in per.values()):.1f} us, {b - a} launches")
for k, (c, d) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print(f"{d:8.1f} us {c:3d} x  {k}")
