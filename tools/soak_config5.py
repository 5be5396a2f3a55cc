"""Soak: 3000 config-5 training steps (batch 64) through the HIP graph on one fixed batch -- losses finite, the critic loss goes down, the step
counter and parameters stay sane.  python tools/soak_config5.py (GPU box)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cgs_amd
from cgs_amd import hourglass128
from oracle import hourglass_ref as orc
pc = orc.seeded_params(orc.critic128_shapes(), 31); pm = orc.seeded_params(orc.masker128_shapes(), 32)
n = 64
g = torch.Generator().manual_seed(0)
A = torch.randint(0, 256, (n, 128, 128, 3), dtype=torch.uint8, generator=g).cuda()
B = torch.randint(0, 256, (n, 128, 128, 3), dtype=torch.uint8, generator=g).cuda()
A[: n // 2] = (A[: n // 2].float() * 0.4).to(torch.uint8)
Y = torch.cat([torch.full((n // 2,), 0.9), torch.full((n // 2,), 0.1)]).cuda()
net = hourglass128.Hourglass128(pc, pm)
hist = []
for s in range(3000):
    l = net.phase2_step(A, B, Y) if s == 0 else net.phase2_step()
    if s % 500 == 0 or s == 2999:
        torch.cuda.synchronize(); hist.append([round(float(v), 5) for v in l[:6].cpu()])
print("losses (critic, replace, inject, l1, l2, total) every 500 steps:")
for h in hist: print(h)
assert all(np.isfinite(h).all() for h in hist)
assert hist[-1][0] < hist[0][0], "the critic loss did not go down"
print("step counter", int(net.step_t.item()), "finite parameters:", bool(torch.isfinite(net.flat).all()))
