"""Times the shape-generic 3x3 convolution (gen4.hip: v_mfma_f32_4x4x1, lane = pixel) on the layer shapes of the chfak-5
phase-2 step and checks it against torch's conv2d in float64.  The round-2 16x16x4 implicit GEMM it replaced measured, on the
same shapes (us, chfak 5, n = 512): 308 / 461 / 126 / 102 / 216 / 102 / 141 / 459 / 457 / 788 / 403.
Usage (GPU box): python tools/time_gen.py [chfak] [n]"""
import ctypes as C
import os
import sys

import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from cgs_amd import _lib, generic  # noqa: E402

chfak = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = torch.device("cuda:0")
lib = _lib.load()
c = 8 * chfak
# (name, images, hw, ca, cb, ups, co, act, pool, u8)
shapes = [
    ("features.0  u8 3->c  @64 pool", 2 * n, 64, 3, 0, 1, c, "relu", True, True),
    ("features.3  c->c     @32 pool", 2 * n, 32, c, 0, 1, c, "relu", True, False),
    ("features.6  c->c     @16 pool", 2 * n, 16, c, 0, 1, c, "relu", True, False),
    ("features.10 c->2c    @8  pool", 2 * n, 8, c, 0, 1, 2 * c, "relu", True, False),
    ("dec_model.3 2c+4c->2c @4     ", n, 4, 2 * c, 4 * c, 4, 2 * c, "none", False, False),
    ("dec_model.2 c+2c->c  @8      ", n, 8, c, 2 * c, 2, c, "none", False, False),
    ("dec_model.1 c+c->c   @16     ", n, 16, c, c, 2, c, "none", False, False),
    ("dec_model.0 c+c->c   @32     ", n, 32, c, c, 2, c, "none", False, False),
    ("masker.0 u8 3+c->16  @64     ", n, 64, 3, c, 2, 16, "lrelu", False, True),
    ("dgrad image c->3     @64     ", 2 * n, 64, c, 0, 1, 3, "none", False, False),
    ("dgrad 16->3+c        @64     ", n, 64, 16, 0, 1, 3 + c, "none", False, False),
]


def timed(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


torch.manual_seed(0)
tot4 = tot16 = 0.0
for name, ni, hw, ca, cb, ups, co, act, pool, u8 in shapes:
    a = torch.randint(0, 256, (ni, hw, hw, ca), device=dev, dtype=torch.uint8) if u8 else torch.randn(ni, hw, hw, ca, device=dev)
    b = torch.randn(ni, hw // ups, hw // ups, cb, device=dev) if cb else None
    w = torch.randn(9, ca + cb, co, device=dev) * (1.0 / (3.0 * (ca + cb) ** 0.5))
    bias = torch.randn(co, device=dev) * 0.1
    out = {}
    res = {}
    for form in (4,):
        run = lambda: generic.conv3x3(a, b, w.data_ptr(), bias.data_ptr(), co, act=act, pool=pool, ups=ups, want_argmax=pool)
        r = run()
        out[form] = r
        res[form] = timed(run)
    o4 = out[4][0] if pool else out[4]
    # torch reference (GPU fp32; cudnn-free direct conv)
    xa = (a.float() / 255.0 if u8 else a).permute(0, 3, 1, 2)
    xin = xa if b is None else torch.cat([xa, F.interpolate(b.permute(0, 3, 1, 2), scale_factor=ups, mode="nearest")], 1)
    wt = w.view(3, 3, ca + cb, co).permute(3, 2, 0, 1).contiguous()
    nchk = min(ni, 64)
    ref = F.conv2d(xin[:nchk].double(), wt.double(), bias.double(), padding=1)
    ref = F.relu(ref) if act == "relu" else (F.leaky_relu(ref, 0.01) if act == "lrelu" else ref)
    if pool:
        ref = F.max_pool2d(ref, 2)
    ref = ref.permute(0, 2, 3, 1).float()
    err4 = (o4[:nchk] - ref).abs().max().item() / max(ref.abs().max().item(), 1e-9)
    am_ok = ""
    flops = 2.0 * ni * hw * hw * 9 * (ca + cb) * co
    print(f"{name}: gen4 {res[4]:8.1f} us ({flops / res[4] / 1e6:6.1f} TF)  | rel err vs fp64 conv2d: {err4:.2e}{am_ok}")
    tot4 += res[4]
print(f"sum: gen4 {tot4:.0f} us")
