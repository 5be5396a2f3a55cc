"""Compact per-kernel view of a tools/sq_counters.py CSV: python tools/sq_summary.py gpurun_out/TAG.sq.csv"""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    w = float(r["SQ_WAVES"])
    if w == 0:
        continue
    us = lambda c, div: float(r[c]) / div / 2.1e3
    print(f"{r['kernel'][:40]:40s} us={float(r['us_per_step']):6.1f} mfma_us={us('SQ_VALU_MFMA_BUSY_CYCLES', 1024):5.1f} "
          f"valu/w={float(r['SQ_INSTS_VALU']) / w:6.0f} mops/w={float(r['SQ_INSTS_VALU_MFMA_MOPS_F32']) / w:6.0f} lds/w={float(r['SQ_INSTS_LDS']) / w:5.0f} "
          f"ldsact_us={us('SQ_LDS_IDX_ACTIVE', 256):5.1f} confl={us('SQ_LDS_BANK_CONFLICT', 256):5.1f} wait_any={r['wait_any_frac']} "
          f"wait_inst={r['wait_inst_frac']} busycu_us={us('SQ_BUSY_CU_CYCLES', 256 * 4):.1f}")
