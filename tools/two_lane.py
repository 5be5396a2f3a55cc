"""Would two half-batch lanes on two streams inside one HIP graph beat the single-lane step?  (The latency-bound tail kernels of one
lane could run under the issue-bound convolutions of the other.)  Two independent engines of n/2 images each -- the gradient
plumbing of a real two-lane step is NOT done here, this only measures what the overlap would give.
Usage (GPU box): python tools/two_lane.py [stagger_launches ...]   stagger k: lane 2 starts after lane 1's k-th launch."""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from cgs_amd import engine, _lib  # noqa: E402

n = 512
dev = torch.device("cuda:0")


def timed(fn, reps):
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


full = engine.HourglassEngine(n, device=dev, dropout=0.3, use_graph=True)
full.load_state(*bench.g1_weights())
A, B, Y = bench.synthetic(n, 0, dev)
for _ in range(3):
    full.phase2_step(A, B, Y)
for _ in range(3):
    timed(lambda: full.phase2_step(), 300)
print(f"single lane, n = {n}: {timed(lambda: full.phase2_step(), 400) * 1e3:.4f} ms/step")

h = n // 2
lanes = []
for i in range(2):
    e = engine.HourglassEngine(h, device=dev, dropout=0.3, use_graph=False)
    e.load_state(*bench.g1_weights())
    e.phase2_step(A[i * h:(i + 1) * h].contiguous(), B[i * h:(i + 1) * h].contiguous(), Y[i * h:(i + 1) * h].contiguous())
    lanes.append(e)
torch.cuda.synchronize()
one = engine.HourglassEngine(h, device=dev, dropout=0.3, use_graph=True)
one.load_state(*bench.g1_weights())
one.phase2_step(A[:h].contiguous(), B[:h].contiguous(), Y[:h].contiguous())
print(f"single lane, n = {h}: {timed(lambda: one.phase2_step(), 400) * 1e3:.4f} ms/step")

real_call = _lib.call
for stagger in [int(a) for a in sys.argv[1:]] or [0, 3, 6]:
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur)
        ev = torch.cuda.Event()
        count = [0]

        def counting_call(name, *args):
            real_call(name, *args)
            count[0] += 1
            if count[0] == stagger:
                ev.record(torch.cuda.current_stream())

        with torch.cuda.stream(s1):
            _lib.call = counting_call
            engine._lib.call = counting_call
            engine.hg._lib.call = counting_call
            lanes[0]._phase2_fwd_bwd()
            _lib.call = engine._lib.call = engine.hg._lib.call = real_call
        if stagger > 0:
            s2.wait_event(ev)
        else:
            s2.wait_stream(cur)
        with torch.cuda.stream(s2):
            lanes[1]._phase2_fwd_bwd()
        cur.wait_stream(s1)
        cur.wait_stream(s2)
    for _ in range(2):
        timed(g.replay, 300)
    print(f"two lanes of {h} on two streams, lane 2 starts after lane 1's launch {stagger} of {count[0]}: {timed(g.replay, 400) * 1e3:.4f} ms per pair of half steps")
