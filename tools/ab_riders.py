"""In-process sweep of hourglass.DEC0_RIDERS (rider workgroups of dec_model.0's weight gradient in the A pass of the fused backward kernel) at N = 512:
python tools/ab_riders.py  (r05: 256 = one per CU is a sharp optimum: 128 +19 us, 192 +9, 320 / 384 +14)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cgs_amd import engine, hourglass as hg
n, dev = 512, torch.device("cuda:0")
A, B, Y = bench.synthetic(n, 0, dev)
engs = {}
for r in (256, 128, 192, 320, 384):
    hg.DEC0_RIDERS = r
    e = engine.HourglassEngine(n, device=dev, dropout=0.3, use_graph=True)
    e.load_state(*bench.g1_weights())
    for _ in range(3): e.phase2_step(A, B, Y)
    engs[r] = e
hg.DEC0_RIDERS = 256
torch.cuda.synchronize()
def timed(e, reps=400):
    for _ in range(50): e.phase2_step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): e.phase2_step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for e in engs.values(): timed(e, 800)
for _ in range(3): print("  ".join(f"riders {r}: {timed(e):.4f}" for r, e in engs.items()), flush=True)
