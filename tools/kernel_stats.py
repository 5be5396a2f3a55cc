import csv, sys, glob
f = sorted(glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv") + glob.glob(sys.argv[1] + "/*_kernel_stats.csv"))[-1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 36
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{f}: total per step {tot/steps/1e3:.1f} us")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    name = r["Name"].replace("void ", "")[:60]
    per_step = float(r["TotalDurationNs"]) / steps / 1e3
    print(f'{name:60s} calls/step={int(r["Calls"])/steps:4.1f} avg_us={float(r["AverageNs"])/1e3:8.1f} us/step={per_step:7.1f} {100*float(r["TotalDurationNs"])/tot:5.1f}%')
