"""One layer shape of the generic convolution, a few launches: the target of rocprofv3 --pmc runs.
Usage: python tools/gen4_one.py [hw ca cb co n pool]"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from cgs_amd import generic  # noqa: E402

hw, ca, cb, co, n, pool = (int(v) for v in sys.argv[1:7]) if len(sys.argv) > 6 else (32, 40, 0, 40, 1024, 1)
dev = torch.device("cuda:0")
a = torch.randn(n, hw, hw, ca, device=dev)
b = torch.randn(n, hw // 2, hw // 2, cb, device=dev) if cb else None
w = torch.randn(9, ca + cb, co, device=dev) * 0.05
bias = torch.zeros(co, device=dev)
for _ in range(5):
    generic.conv3x3(a, b, w.data_ptr(), bias.data_ptr(), co, act="relu", pool=bool(pool), ups=2, want_argmax=bool(pool))
torch.cuda.synchronize()
