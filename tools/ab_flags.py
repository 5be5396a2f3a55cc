"""In-process A/B of a boolean switch of cgs_amd.hourglass (e.g. EARLY_REDUCE): python tools/ab_flags.py NAME [rounds]
Builds one engine per setting (N = 512, dropout 0.3, HIP graph) and times them alternately."""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from cgs_amd import engine, hourglass as hg  # noqa: E402

name = sys.argv[1]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n, dev = 512, torch.device("cuda:0")
A, B, Y = bench.synthetic(n, 0, dev)
engs = {}
for val in (True, False):
    setattr(hg, name, val)
    e = engine.HourglassEngine(n, device=dev, dropout=0.3, use_graph=True)
    e.load_state(*bench.g1_weights())
    for _ in range(3):
        e.phase2_step(A, B, Y)
    engs[val] = e
torch.cuda.synchronize()


def timed(e, reps=300):
    for _ in range(50):
        e.phase2_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        e.phase2_step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for e in engs.values():
    timed(e, 600)
for r in range(rounds):
    print(f"{name}: True {timed(engs[True]):.4f} ms/step   False {timed(engs[False]):.4f} ms/step", flush=True)
print("losses True ", engs[True].losses.cpu().tolist()[:6])
print("losses False", engs[False].losses.cpu().tolist()[:6])
