# needs the debug build of the library: python tools/build_variant.py stamps -DCGS_DEBUG_STAMPS ; CGS_LIB_PATH=<pkg>/libcgs_hip_stamps.so (the product library exports no dbg_* hooks)
"""Phase time stamps (s_memtime, thread 0 of every workgroup) of the one-kernel mask head forward: tools/maskfwd_stamps.py [n]"""
import ctypes as C, os, sys, torch
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cgs_amd import _lib
dev = torch.device("cuda:0")
P = lambda t: C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
x = torch.randint(0, 256, (n, 64, 64, 3), dtype=torch.uint8, device=dev)
o0 = torch.randn(n, 32, 32, 8, device=dev)
w0, b0, w2, b2 = torch.randn(9 * 11 * 16, device=dev) * 0.1, torch.randn(16, device=dev), torch.randn(144, device=dev) * 0.1, torch.randn(1, device=dev)
h, z, zp = torch.empty(n, 64, 64, 16, device=dev), torch.empty(n, 64, 64, device=dev), torch.empty(n, 2, device=dev)
call = lambda: _lib.call("cgs_mask_train_fwd", n, _lib.SRC_U8, P(x), P(o0), P(w0), P(b0), P(w2), P(b2), P(h), P(z), P(zp), st())
for _ in range(20): call()
torch.cuda.synchronize()
stamps = torch.zeros(n, 64, dtype=torch.int64, device=dev)
lib = _lib.load()
lib.dbg_maskfwd_stamps(P(stamps))
call(); torch.cuda.synchronize()
lib.dbg_maskfwd_stamps(None)
s = stamps.cpu().numpy().astype("int64")
t0 = s[:, 0].min()
names = {0: "start", 1: "weights ready", 2: "strip top", 3: "committed+barrier", 4: "image MFMAs done", 5: "ups MFMAs + lrelu done", 6: "barrier",
         7: "h staged + stores issued", 8: "masker.2 MFMAs done", 9: "windows + barrier", 10: "finalised", 11: "barrier"}
print(f"n={n}: kernel span {(s[:, 41].max() - t0)} ticks of s_memtime")
for wg in (0, n // 2, n - 1):
    print(f"workgroup {wg}: start +{s[wg, 0] - t0}")
prev = s[:, 0]
for k in [1] + [2 + 10 * st_ + j for st_ in range(4) for j in range(10)]:
    cur = s[:, k]
    d = cur - prev
    kk = k if k < 2 else 2 + (k - 2) % 10
    print(f"stamp {k:2d} {names.get(kk, ''):28s} median delta {int(np.median(d)):7d}  max {int(d.max()):7d}   median since start {int(np.median(cur - s[:, 0])):8d}")
    prev = cur
