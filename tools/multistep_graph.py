"""How much of a step is the gap between two graph launches?  Times the engine's one-step HIP graph against a graph that holds
K consecutive steps (same kernels, same resident batch).  Usage (GPU box): python tools/multistep_graph.py [K]"""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from cgs_amd import engine  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = 512
dev = torch.device("cuda:0")
eng = engine.HourglassEngine(n, device=dev, dropout=0.3, use_graph=True)
eng.load_state(*bench.g1_weights())
A, B, Y = bench.synthetic(n, 0, dev)
for _ in range(3):
    eng.phase2_step(A, B, Y)
torch.cuda.synchronize()


def timed(fn, reps):
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for _ in range(3):      # clocks
    timed(lambda: eng.phase2_step(), 300)
t1 = timed(lambda: eng.phase2_step(), 400)
gk = torch.cuda.CUDAGraph()
with torch.cuda.graph(gk):
    for _ in range(K):
        eng._phase2_fwd_bwd()
        eng._adam()
tk = timed(gk.replay, 400 // K) / K
t1b = timed(lambda: eng.phase2_step(), 400)
print(f"one step per graph: {t1 * 1e3:.4f} ms/step (again: {t1b * 1e3:.4f});  {K} steps per graph: {tk * 1e3:.4f} ms/step")
