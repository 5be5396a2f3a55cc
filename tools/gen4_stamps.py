# needs the debug build of the library: python tools/build_variant.py stamps -DCGS_DEBUG_STAMPS ; CGS_LIB_PATH=<pkg>/libcgs_hip_stamps.so (the product library exports no dbg_* hooks)
"""Phase timing of gen4_conv3x3_kernel (debug hook dbg_gen4_stamps): mean s_memtime deltas between the phase boundaries of the
first 4096 workgroups for one layer shape.  Usage (GPU box): python tools/gen4_stamps.py [hw ca cb co n [u8] [pool]]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from cgs_amd import _lib, generic  # noqa: E402

hw, ca, cb, co, n = (int(v) for v in sys.argv[1:6]) if len(sys.argv) > 5 else (32, 40, 0, 40, 1024)
dev = torch.device("cuda:0")
lib = _lib.load()
lib.dbg_gen4_stamps.argtypes = [C.c_void_p]
u8, pool = "u8" in sys.argv, "pool" in sys.argv
a = torch.randint(0, 256, (n, hw, hw, ca), device=dev, dtype=torch.uint8) if u8 else torch.randn(n, hw, hw, ca, device=dev)
b = torch.randn(n, hw // 2, hw // 2, cb, device=dev) if cb else None
w = torch.randn(9, ca + cb, co, device=dev) * 0.05
bias = torch.zeros(co, device=dev)
run = lambda: generic.conv3x3(a, b, w.data_ptr(), bias.data_ptr(), co, act="relu", pool=pool, ups=2, want_argmax=pool)
for _ in range(3):
    run()
torch.cuda.synchronize()
buf = torch.zeros(4096 * 32, dtype=torch.int64, device=dev)
lib.dbg_gen4_stamps(C.c_void_p(buf.data_ptr()))
run()
torch.cuda.synchronize()
lib.dbg_gen4_stamps(C.c_void_p(0))
st = buf.cpu().numpy().reshape(4096, 32).astype(np.float64)
live = st[:, 0] != 0
st = st[live]
cols = [c for c in range(32) if (st[:, c] != 0).all()]
names = {0: "start"}
for ch in range(3):
    names.update({1 + 5 * ch: f"c{ch}:tile staged", 2 + 5 * ch: f"c{ch}:weights staged", 3 + 5 * ch: f"c{ch}:barrier", 4 + 5 * ch: f"c{ch}:mfma done", 5 + 5 * ch: f"c{ch}:barrier2"})
names[31] = "end"
print(f"{live.sum()} workgroups; shape hw={hw} ca={ca} cb={cb} co={co} n={n}")
prev = cols[0]
for c in cols[1:]:
    print(f"  {str(names.get(prev, prev)):>22s} -> {str(names.get(c, c)):<22s} {np.mean(st[:, c] - st[:, prev]):9.0f} ticks")
    prev = c
print("  total", np.mean(st[:, cols[-1]] - st[:, cols[0]]))
