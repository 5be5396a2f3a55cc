"""Static census of one kernel's ISA by source line: python tools/isa_lines.py FILE.s KERNEL_SUBSTR [top]
FILE.s from `hipcc -S -gline-tables-only --cuda-device-only`.  Counts VALU / MFMA / LDS / VMEM / SALU instructions per (file:line) of the
innermost .loc; for kernels whose loops are fully unrolled the static count is the dynamic count per wave."""
import collections, re, sys
path, pat = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
files = {}
cur = None
inside = False
cnt = collections.defaultdict(collections.Counter)
ops = collections.defaultdict(collections.Counter)
tot = collections.Counter()
for l in open(path, errors="replace"):
    s = l.strip()
    m = re.match(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', s)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
        continue
    m = re.match(r"^(_Z\S+):", s)
    if m:
        inside = pat in m.group(1)
        continue
    if not inside:
        continue
    if s.startswith(".loc"):
        p = s.split()
        cur = (files.get(int(p[1]), p[1]), int(p[2]))
        continue
    if s.startswith(".Lfunc_end"):
        inside = False
        continue
    if not s or s[0] in ".;":
        continue
    op = s.split()[0]
    kind = ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else
            "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "salu" if op.startswith("s_") else "other")
    cnt[cur][kind] += 1
    tot[kind] += 1
    if kind == "valu":
        ops[cur][op.replace("_e32", "").replace("_e64", "")] += 1
print("totals", dict(tot))
for k, c in sorted(cnt.items(), key=lambda kv: -kv[1]["valu"])[:top]:
    print(f"{k[0]}:{k[1]:<5d} valu {c['valu']:5d} mfma {c['mfma']:4d} lds {c['lds']:4d} vmem {c['vmem']:4d} salu {c['salu']:4d}  ", dict(ops[k].most_common(5)))
