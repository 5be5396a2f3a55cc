"""Soak: 8000 phase-2 steps + inference + 2000 phase-1 steps at N = 512, 64 and 200 (graph replay); prints the losses and whether every parameter is finite."""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cgs_amd import engine
dev = torch.device("cuda:0")
for n in (512, 64, 200):
    A, B, Y = bench.synthetic(n, 0, dev)
    e = engine.HourglassEngine(n, device=dev, dropout=0.3, use_graph=True)
    e.load_state(*bench.g1_weights())
    for i in range(8000):
        l = e.phase2_step(A, B, Y) if i < 3 else e.phase2_step()
    torch.cuda.synchronize()
    pred, Z = e.infer(A)
    print(n, "losses", [round(float(x), 5) for x in l[:6].tolist()], "finite", bool(torch.isfinite(e.flat).all()), "Z", float(Z.min()), float(Z.max()), flush=True)
    for i in range(2000):
        e.phase1_step(A, Y) if i < 3 else e.phase1_step()
    torch.cuda.synchronize()
    print(n, "phase1 loss", float(e.losses[0]), "finite", bool(torch.isfinite(e.flat).all()), flush=True)
