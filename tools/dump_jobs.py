"""Prints the slab-reduction job table of the headline step (rows x columns per destination) -- what reduce_adam reads."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
from cgs_amd import engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
eng = engine.HourglassEngine(n, device="cuda:0", dropout=0.3, use_graph=False)
eng.load_state(*bench.g1_weights())
A, B, Y = bench.synthetic(n, 0, torch.device("cuda:0"))
eng.phase2_step(A, B, Y)
tot = 0
for slab, nsl, cnt, off in eng._plans["p2"].jobs:
    tot += nsl * cnt * 4
    print(f"rows {nsl:5d} x cols {cnt:6d} -> grad[{off}:{off + cnt}]  {nsl * cnt * 4 / 1e6:.2f} MB")
print(f"total {tot / 1e6:.1f} MB read by the reduction, {len(eng._plans['p2'].jobs)} jobs")
