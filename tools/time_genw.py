"""Times the shape-generic 3x3 weight gradient (gen_train.hip: gen_conv3x3_wgrad_kernel) on the layer shapes of the chfak-5
phase-2 step and checks it against torch autograd (float64) on the first images.
Usage (GPU box): python tools/time_genw.py [chfak] [n]"""
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
# (name, images, hw, ca, cb, ups, co, pooled dy, u8)
shapes = [
    ("features.0  u8 3->c  @64 pool", 3 * n, 64, 3, 0, 1, c, True, True),
    ("features.3  c->c     @32 pool", 3 * n, 32, c, 0, 1, c, True, False),
    ("features.6  c->c     @16 pool", 3 * n, 16, c, 0, 1, c, True, False),
    ("features.10 c->2c    @8  pool", 3 * n, 8, c, 0, 1, 2 * c, True, False),
    ("dec_model.3 2c+4c->2c @4     ", n, 4, 2 * c, 4 * c, 4, 2 * c, False, False),
    ("dec_model.2 c+2c->c  @8      ", n, 8, c, 2 * c, 2, c, False, False),
    ("dec_model.1 c+c->c   @16     ", n, 16, c, c, 2, c, False, False),
    ("dec_model.0 c+c->c   @32     ", n, 32, c, c, 2, c, False, False),
    ("masker.0 u8 3+c->16  @64     ", n, 64, 3, c, 2, 16, False, True),
]
_p = generic._p


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
tot = 0.0
for name, ni, hw, ca, cb, ups, co, pooled, u8 in shapes:
    a = torch.randint(0, 256, (ni, hw, hw, ca), device=dev, dtype=torch.uint8) if u8 else torch.randn(ni, hw, hw, ca, device=dev)
    b = torch.randn(ni, hw // ups, hw // ups, cb, device=dev) if cb else None
    if pooled:
        dy = torch.randn(ni, hw // 2, hw // 2, co, device=dev)
        am = torch.randint(0, 5, (ni, hw // 2, hw // 2, co), device=dev, dtype=torch.uint8)
    else:
        dy = torch.randn(ni, hw, hw, co, device=dev)
        am = None
    nsl = lib.cgs_gen_conv3x3_bwd_weight_slabs(ni, ca, cb, co)
    cnt = 9 * (ca + cb) * co + co
    slab = torch.zeros(nsl, cnt, device=dev)
    run = lambda: _lib.call("cgs_gen_conv3x3_bwd_weight", ni, hw, ca, cb, co, int(u8), ups, _p(a), _p(b), _p(dy), _p(am), _p(slab),
                            generic._s())
    us = timed(run)
    got = slab.double().sum(0)
    # float64 reference
    xa = (a.double() / 255.0 if u8 else a.double()).permute(0, 3, 1, 2)
    xin = xa if b is None else torch.cat([xa, F.interpolate(b.double().permute(0, 3, 1, 2), scale_factor=ups, mode="nearest")], 1)
    if pooled:
        pos = torch.arange(4, device=dev).view(1, 2, 2, 1, 1, 1)          # (dy, dx) position inside the 2x2 cell
        full = torch.zeros(ni, hw // 2, 2, hw // 2, 2, co, device=dev, dtype=torch.float64)
        for py in range(2):
            for px in range(2):
                full[:, :, py, :, px, :] = torch.where(am == 2 * py + px, dy.double(), torch.zeros((), device=dev, dtype=torch.float64))
        dyf = full.view(ni, hw, hw, co)
    else:
        dyf = dy.double()
    w = torch.zeros(co, ca + cb, 3, 3, device=dev, dtype=torch.float64, requires_grad=True)
    bias = torch.zeros(co, device=dev, dtype=torch.float64, requires_grad=True)
    chunk = 128
    gw = torch.zeros_like(w)
    gb = torch.zeros_like(bias)
    for i in range(0, ni, chunk):
        o = F.conv2d(xin[i:i + chunk], w, bias, padding=1)
        g1, g2 = torch.autograd.grad(o, (w, bias), dyf[i:i + chunk].permute(0, 3, 1, 2))
        gw += g1
        gb += g2
    ref = torch.cat([gw.permute(2, 3, 1, 0).reshape(-1), gb])
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    flops = 2.0 * ni * hw * hw * 9 * (ca + cb) * co
    print(f"{name}: {us:8.1f} us ({flops / us / 1e6:6.1f} TF)  slabs {nsl:5d}  rel err vs fp64 autograd: {err:.2e}", flush=True)
    tot += us
print(f"sum: {tot:.0f} us")
