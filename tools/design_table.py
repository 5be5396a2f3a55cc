"""Markdown rows of DESIGN.md section 4's kernel table from ONE collection: python tools/design_table.py profiles/r06_a
(PREFIX_sq_counters.csv: the four rocprofv3 --pmc passes, eager launches; PREFIX_kernel_stats.csv: the --kernel-trace --stats profile of the graph step)."""
import csv, sys
pre = sys.argv[1]
stats = {}
for r in csv.DictReader(open(pre + "_kernel_stats.csv")):
    stats[r["Name"].replace("void ", "")[:40]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
steps = None
for k, (c, us) in stats.items():
    if k.startswith("reduce_adam_kernel"):
        steps = c
for r in csv.DictReader(open(pre + "_sq_counters.csv")):
    f = lambda k: float(r[k] or 0)
    us = f("us_per_step")
    if us < 5:
        continue
    cyc = 1024 * us * 2.1e3
    key = r["kernel"][:40]
    st = stats.get(key)
    prof = f"{st[1] * st[0] / steps:.1f}" if st and steps else "?"
    print(f"{r['kernel'][:52]:52s} | {us:.1f} / {prof} | {f('SQ_VALU_MFMA_BUSY_CYCLES') / cyc:.2f} | {f('wait_any_frac'):.2f} | "
          f"{f('SQ_LDS_BANK_CONFLICT') / max(1, f('SQ_LDS_IDX_ACTIVE')):.2f} | fetch {f('FETCH_bytes_raw') / 1e6:.0f} MB, write {f('WRITE_bytes') / 1e6:.0f} MB")
