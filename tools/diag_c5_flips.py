"""Diagnostic (GPU box): are the chfak-5 n=128 gradient differences of features.0 / features.3 argmax flips of near-tied pool cells?"""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch, torch.nn.functional as F
import test_gpu_generic_train as T
from oracle import hourglass_ref as orc
chfak, neck, n = 5, 32, 128
rs = np.random.RandomState(11)
dev = torch.device("cuda:0")
A = rs.randint(0, 256, (n, 64, 64, 3)).astype(np.uint8)
B = rs.randint(0, 256, (n, 64, 64, 3)).astype(np.uint8)
Y = rs.rand(n).astype(np.float32)
e, pc, pm = T.make_generic_engine(chfak, n, neck=neck, dropout=0.3, use_graph=False)
e.phase2_step(torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev), torch.from_numpy(Y).to(dev))
torch.cuda.synchronize()
torch.set_num_threads(16)
for li, key, src in ((0, "features.0", e.x3), (1, "features.3", e.cbuf["e0"][n:4 * n])):
    x = src.cpu().double().permute(0, 3, 1, 2)
    w, b = pc[key + ".weight"].double(), pc[key + ".bias"].double()
    y = F.relu(F.conv2d(x, w, b, padding=1))
    N, Cc, H, W = y.shape
    cells = y.reshape(N, Cc, H // 2, 2, W // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(N, Cc, H // 2, W // 2, 4)
    top, idx = cells.max(-1)
    srt = cells.sort(-1, descending=True).values
    gap = (srt[..., 0] - srt[..., 1]) / srt[..., 0].clamp_min(1e-30)
    am = e.cbuf[f"am{li}"][n:4 * n].cpu().permute(0, 3, 1, 2).long()
    live = top > 0
    mism = live & (am != idx)
    dead_mism = (~live) & (am < 4)
    live_dead = live & (am >= 4)
    print(key, "cells", live.numel(), "live", int(live.sum()), "argmax mismatches", int(mism.sum()), "gpu-live/ref-dead", int(dead_mism.sum()),
          "gpu-dead/ref-live", int(live_dead.sum()))
    if mism.any():
        print("  relative gaps of mismatched cells:", np.sort(gap[mism].numpy())[-10:], " top values:", top[mism][:10].numpy())
    print("  cells with rel gap < 1e-6:", int(((gap < 1e-6) & live).sum()), " < 1e-7:", int(((gap < 1e-7) & live).sum()))
