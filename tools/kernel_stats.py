"""Per-step view of a rocprofv3 --stats kernel summary: python tools/kernel_stats.py DIR [steps|0] [rows] [ms_per_step of the
un-profiled run].  steps = 0 (default): derived from the Calls column of a once-per-step kernel (reduce_adam / mask_head / reduce_slabs)."""
import csv, sys, glob
f = sorted(glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv") + glob.glob(sys.argv[1] + "/*_kernel_stats.csv"))[-1]
rows = list(csv.DictReader(open(f)))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if steps <= 0:
    once = [int(r["Calls"]) for r in rows if any(k in r["Name"] for k in ("reduce_adam_kernel", "mask_head_kernel", "reduce_slabs"))]
    steps = min(once) if once else 36
prod = [r for r in rows if not any(k in r["Name"] for k in ("at::native", "__amd_rocclr", "ncclDevKernel", "rccl"))]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
tot_prod = sum(float(r["TotalDurationNs"]) for r in prod)
print(f"{f}: {steps} steps in the profile (from a once-per-step kernel's Calls); all kernels {tot/steps/1e3:.1f} us/step, "
      f"the step's own kernels {tot_prod/steps/1e3:.1f} us/step")
if len(sys.argv) > 4:
    ms = float(sys.argv[4])
    print(f"un-profiled ms_per_step {ms:.4f}: per-step kernel sum under the profiler / ms_per_step = {tot_prod/steps/1e3/(ms*1e3):.3f}")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    name = r["Name"].replace("void ", "")[:60]
    per_step = float(r["TotalDurationNs"]) / steps / 1e3
    print(f'{name:60s} calls/step={int(r["Calls"])/steps:4.1f} avg_us={float(r["AverageNs"])/1e3:8.1f} us/step={per_step:7.1f} {100*float(r["TotalDurationNs"])/tot:5.1f}%')
