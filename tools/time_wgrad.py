"""Times cgs_conv3x3_bwd_weight for features.0 (uint8 frames) / features.3 at a few batch sizes: tools/time_wgrad.py"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cgs_amd import _lib, hourglass as hg
dev = torch.device("cuda:0")
P = lambda t: C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
nd = _lib.Dropout(0.0, 0, 0, None, 0, 0)
for name, hw, ca, u8 in (("features.0/u8", 64, 3, True), ("features.3", 32, 8, False)):
    for n in (128, 512, 1024, 2048):
        x = torch.randint(0, 256, (n, hw, hw, ca), dtype=torch.uint8, device=dev) if u8 else torch.randn(n, hw, hw, ca, device=dev)
        dE = torch.randn(n, hw // 2, hw // 2, 8, device=dev)
        am = torch.randint(0, 2**31 - 1, (n, hw // 2, hw // 2, 1), dtype=torch.int32, device=dev) & 0x33333333
        d = hg.conv_desc(n, hw, ca, 0, 8, u8, 2, "relu", 1, nd)
        nsl = _lib.load().cgs_conv3x3_bwd_weight_slabs(C.byref(d))
        slab = torch.empty(nsl, 9 * ca * 8 + 8, device=dev)
        call = lambda: _lib.call("cgs_conv3x3_bwd_weight", C.byref(d), P(x), None, P(dE), P(am), P(slab), st())
        for _ in range(5): call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): call()
        e1.record(); torch.cuda.synchronize()
        print(f"{name:14s} n={n:5d} slabs={nsl:5d}: {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us")
