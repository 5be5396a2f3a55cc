"""One headline engine (N = 512, dropout 0.3, HIP graph) with module switches of cgs_amd.hourglass overridden, for a profiler:
rocprofv3 --kernel-trace --stats --output-format csv -d OUT -o runc -- python3 tools/prof_flags.py NAME=VALUE [NAME=VALUE ...]
(VALUE: int; booleans as 0 / 1).  Prints the un-profiled-clock ms per step of 200 graph replays."""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from cgs_amd import engine, hourglass as hg  # noqa: E402

for a in sys.argv[1:]:
    k, v = a.split("=")
    old = getattr(hg, k)
    setattr(hg, k, bool(int(v)) if isinstance(old, bool) else int(v))
n, dev = 512, torch.device("cuda:0")
A, B, Y = bench.synthetic(n, 0, dev)
e = engine.HourglassEngine(n, device=dev, dropout=0.3, use_graph=True)
e.load_state(*bench.g1_weights())
for _ in range(3):
    e.phase2_step(A, B, Y)
for _ in range(300):
    e.phase2_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    e.phase2_step()
torch.cuda.synchronize()
print(" ".join(sys.argv[1:]) or "(product settings)", f"{(time.perf_counter() - t0) / 200 * 1e3:.4f} ms/step")
