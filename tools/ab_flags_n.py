"""In-process A/B of boolean switches of cgs_amd.hourglass at a batch size: python tools/ab_flags_n.py N FLAG [FLAG ...]
One engine with every switch at its default, one per FLAG (A+B = both) with that switch off (HIP graph, dropout 0.3); timed alternately, three rounds."""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from cgs_amd import engine, hourglass as hg  # noqa: E402

n, names = int(sys.argv[1]), sys.argv[2:]
dev = torch.device("cuda:0")
A, B, Y = bench.synthetic(n, 0, dev)
flat = [f for k in names for f in k.split("+")]
defaults = {k: getattr(hg, k) for k in flat}


def make():
    e = engine.HourglassEngine(n, device=dev, dropout=0.3, use_graph=True)
    e.load_state(*bench.g1_weights())
    for _ in range(3):
        e.phase2_step(A, B, Y)
    return e


engs = {"default": make()}
for k in names:
    for f in k.split("+"):
        setattr(hg, f, False)
    engs["no " + k] = make()
    for f in k.split("+"):
        setattr(hg, f, defaults[f])
torch.cuda.synchronize()


def timed(e, reps=400):
    for _ in range(50):
        e.phase2_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        e.phase2_step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for e in engs.values():
    timed(e, 800)
for r in range(3):
    print(f"N={n}  " + "   ".join(f"{k}: {timed(e):.4f}" for k, e in engs.items()), flush=True)
