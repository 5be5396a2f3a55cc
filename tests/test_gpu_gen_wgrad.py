"""GPU parity of the shape-generic 3x3 weight gradient in its row-block form (csrc/gen_wgrad_rows.h, behind
cgs_gen_conv3x3_bwd_weight) and of the split-K batch reduction (cgs_gen_gemm_ex_splitk) on their own: every chunk geometry
(hw 4 .. 64, several images per chunk with a ragged last chunk, image counts below the number of workgroup shares), every
source kind (fp32 whole quads / odd-width fp32 / uint8, with and without a nearest-upsampled second source at ups 2 and 4),
pooled gradients (dE + argmax bytes, incl. the no-gradient code) and plain ones, one / two input-channel slices, one / two
output-channel slices, 1 / 2 / 4 row groups -- against torch autograd on the CPU in float64 (the reference's own backward of
nets.py:170-183 / 480-489 under loss.backward(), main.py:461)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref_wgrad(a, b, ups, dy_full, co):
    """float64 autograd of conv2d(cat(a, up(b))) w.r.t. (weight, bias) with the output gradient dy_full [n,hw,hw,co]."""
    xa = (a.double() / 255.0 if a.dtype == torch.uint8 else a.double()).permute(0, 3, 1, 2)
    xin = xa if b is None else torch.cat([xa, F.interpolate(b.double().permute(0, 3, 1, 2), scale_factor=ups, mode="nearest")], 1)
    ci = xin.shape[1]
    w = torch.zeros(co, ci, 3, 3, dtype=torch.float64, requires_grad=True)
    bias = torch.zeros(co, dtype=torch.float64, requires_grad=True)
    o = F.conv2d(xin, w, bias, padding=1)
    gw, gb = torch.autograd.grad(o, (w, bias), dy_full.double().permute(0, 3, 1, 2))
    return torch.cat([gw.permute(2, 3, 1, 0).reshape(-1), gb])          # HWIO dW | dbias: the slab row's order


@pytest.mark.parametrize("n,hw,ca,cb,ups,co,pooled,u8", [
    (5, 64, 3, 0, 1, 40, True, True),          # features.0 at chfak 5: uint8 frames, one row group x 8 pixel phases, pooled dY
    (3, 64, 3, 0, 1, 40, True, False),         # odd-width fp32 frames (the step's x3 buffer)
    (5, 32, 40, 0, 1, 40, True, False),        # 23 row blocks, 4 row groups, 256-pixel chunks
    (3, 16, 40, 0, 1, 40, True, False),
    (7, 8, 40, 0, 1, 80, True, False),         # two output-channel slices, two images per chunk + ragged last chunk
    (19, 4, 80, 160, 4, 80, False, False),     # dec_model.3: six input slices (A | up4(B)), eight images per chunk, ragged
    (6, 8, 40, 80, 2, 40, False, False),       # dec_model.2: A slice + two B slices
    (3, 16, 40, 40, 2, 40, False, False),
    (2, 32, 40, 40, 2, 40, False, False),
    (2, 64, 3, 40, 2, 16, False, True),        # masker.0: uint8 frames + upsampled decoder map in ONE slice, one column block
    (1, 32, 16, 0, 1, 16, True, False),        # chfak 2: 9 row blocks -> 2 row groups
    (2, 16, 24, 24, 2, 24, False, False),      # chfak 3: slice of 48
    (3, 8, 12, 0, 1, 20, False, False),        # 7 row blocks -> 1 row group; 20 output channels: partial second column block
    (300, 4, 8, 8, 4, 8, False, False),        # more images than workgroup shares on the smallest map
])
def test_wgrad_rows_vs_float64_autograd(n, hw, ca, cb, ups, co, pooled, u8):
    from cgs_amd import _lib, generic
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1000 * hw + 10 * ca + co + n)
    a = torch.randint(0, 256, (n, hw, hw, ca), dtype=torch.uint8, generator=g) if u8 else torch.randn(n, hw, hw, ca, generator=g)
    b = torch.randn(n, hw // ups, hw // ups, cb, generator=g) if cb else None
    if pooled:
        dE = torch.randn(n, hw // 2, hw // 2, co, generator=g)
        am = torch.randint(0, 5, (n, hw // 2, hw // 2, co), dtype=torch.uint8, generator=g)       # 4: no gradient (pooled value <= 0)
        am = torch.where(torch.rand(am.shape, generator=g) < 0.1, am | 4, am)                     # bit 2 set on a valid position too
        full = torch.zeros(n, hw // 2, 2, hw // 2, 2, co)
        for py in range(2):
            for px in range(2):
                full[:, :, py, :, px, :] = torch.where(am == 2 * py + px, dE, torch.zeros(()))
        dy_full = full.view(n, hw, hw, co)
        dy_d, am_d = dE.to(dev), am.to(dev)
    else:
        dy_full = torch.randn(n, hw, hw, co, generator=g)
        dy_d, am_d = dy_full.to(dev), None
    a_d, b_d = a.to(dev), (b.to(dev) if b is not None else None)
    lib = _lib.load()
    nsl = lib.cgs_gen_conv3x3_bwd_weight_slabs(n, ca, cb, co)
    assert nsl >= 1
    cnt = 9 * (ca + cb) * co + co
    slab = torch.full((nsl, cnt), float("nan"), device=dev)            # every element of every row must be written
    _lib.call("cgs_gen_conv3x3_bwd_weight", n, hw, ca, cb, co, int(u8), ups, generic._p(a_d), generic._p(b_d), generic._p(dy_d),
              generic._p(am_d), generic._p(slab), generic._s())
    torch.cuda.synchronize()
    got = slab.double().sum(0).cpu()
    assert torch.isfinite(got).all(), "an element of a slab row was not written"
    ref = _ref_wgrad(a, b, ups, dy_full, co)
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    assert err < 3e-6, f"weight gradient off by {err:.2e} of its maximum"
    # the same call again gives the same bits (fixed summation order, no atomics)
    slab2 = torch.zeros_like(slab)
    _lib.call("cgs_gen_conv3x3_bwd_weight", n, hw, ca, cb, co, int(u8), ups, generic._p(a_d), generic._p(b_d), generic._p(dy_d),
              generic._p(am_d), generic._p(slab2), generic._s())
    torch.cuda.synchronize()
    assert torch.equal(slab, slab2)


@pytest.mark.parametrize("n,hw,ca,cb,co,u8", [
    (3, 32, 40, 40, 40, False),        # dec_model.0 at chfak 5: ten row blocks per parity class x three column blocks
    (2, 64, 3, 40, 16, True),          # masker.0: uint8 frames (A by the row-block kernel's narrow form) + 40 folded channels, one column block
    (5, 16, 40, 40, 40, False),        # dec_model.1: one chunk per image
    (2, 32, 16, 16, 16, False),        # chfak 2
    (3, 16, 24, 24, 24, False),        # chfak 3
    (2, 64, 3, 32, 16, False),         # chfak 4 masker.0 with fp32 frames
    (700, 16, 8, 16, 8, False),        # more images than workgroups; a column block of 8
])
def test_wgrad_folded_upsample_vs_float64_autograd(n, hw, ca, cb, co, u8):
    """cgs_gen_conv3x3_bwd_weight_folded (A's rows + bias by the row-block kernel, B's rows folded at B's resolution) == float64 autograd of
    conv2d over the explicit cat(A, nearest-up_2(B)) (nets.py:480-489, 501-513); every element of every slab row written; same bits twice."""
    from cgs_amd import _lib, generic
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(77 * hw + ca + cb + co + n)
    a = torch.randint(0, 256, (n, hw, hw, ca), dtype=torch.uint8, generator=g) if u8 else torch.randn(n, hw, hw, ca, generator=g)
    b = torch.randn(n, hw // 2, hw // 2, cb, generator=g)
    dy = torch.randn(n, hw, hw, co, generator=g)
    lib = _lib.load()
    nsl = lib.cgs_gen_conv3x3_bwd_weight_folded_slabs(n, hw, ca, cb, co)
    assert nsl >= 1
    cnt = 9 * (ca + cb) * co + co
    a_d, b_d, dy_d = a.to(dev), b.to(dev), dy.to(dev)
    slab = torch.full((nsl, cnt), float("nan"), device=dev)
    _lib.call("cgs_gen_conv3x3_bwd_weight_folded", n, hw, ca, cb, co, int(u8), generic._p(a_d), generic._p(b_d), generic._p(dy_d),
              generic._p(slab), generic._s())
    torch.cuda.synchronize()
    got = slab.double().sum(0).cpu()
    assert torch.isfinite(got).all(), "an element of a slab row was not written"
    nref = min(n, 64)                                  # (float64 autograd on the CPU: the first images; the rest by linearity below)
    if n > nref:
        slab_h = torch.zeros_like(slab)
        _lib.call("cgs_gen_conv3x3_bwd_weight_folded", nref, hw, ca, cb, co, int(u8), generic._p(a_d), generic._p(b_d), generic._p(dy_d),
                  generic._p(slab_h), generic._s())
        torch.cuda.synchronize()
        nh = lib.cgs_gen_conv3x3_bwd_weight_folded_slabs(nref, hw, ca, cb, co)
        got = slab_h[:nh].double().sum(0).cpu()
        # the rest of the batch through torch on the device in float32 (coarser bound)
        xin = torch.cat([a_d.float().permute(0, 3, 1, 2), F.interpolate(b_d.permute(0, 3, 1, 2), scale_factor=2, mode="nearest")], 1).double()
        w = torch.zeros(co, ca + cb, 3, 3, dtype=torch.float64, device=dev, requires_grad=True)
        bias = torch.zeros(co, dtype=torch.float64, device=dev, requires_grad=True)
        gw, gb = torch.autograd.grad(F.conv2d(xin, w, bias, padding=1), (w, bias), dy_d.double().permute(0, 3, 1, 2))
        full = torch.cat([gw.permute(2, 3, 1, 0).reshape(-1), gb]).cpu()
        assert (slab.double().sum(0).cpu() - full).abs().max().item() / full.abs().max().item() < 3e-6
    ref = _ref_wgrad(a[:nref], b[:nref], 2, dy[:nref], co)
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    assert err < 3e-6, f"weight gradient off by {err:.2e} of its maximum"
    slab2 = torch.zeros_like(slab)
    _lib.call("cgs_gen_conv3x3_bwd_weight_folded", n, hw, ca, cb, co, int(u8), generic._p(a_d), generic._p(b_d), generic._p(dy_d),
              generic._p(slab2), generic._s())
    torch.cuda.synchronize()
    assert torch.equal(slab, slab2)


def test_wgrad_single_channel_output_stays_on_the_16x16x4_kernel():
    """co = 1 (masker.2) has no row-block form: the slab count and the result come from gen_conv3x3_wgrad_kernel."""
    from cgs_amd import _lib, generic
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(7)
    n, hw, ca, co = 3, 64, 16, 1
    a = torch.randn(n, hw, hw, ca, generator=g)
    dy = torch.randn(n, hw, hw, co, generator=g)
    lib = _lib.load()
    nsl = lib.cgs_gen_conv3x3_bwd_weight_slabs(n, ca, 0, co)
    slab = torch.zeros(nsl, 9 * ca * co + co, device=dev)
    a_d, dy_d = a.to(dev), dy.to(dev)           # (kept alive: the call takes raw addresses)
    _lib.call("cgs_gen_conv3x3_bwd_weight", n, hw, ca, 0, co, 0, 1, generic._p(a_d), None, generic._p(dy_d), None,
              generic._p(slab), generic._s())
    ref = _ref_wgrad(a, None, 1, dy, co)
    got = slab.double().sum(0).cpu()
    assert (got - ref).abs().max().item() / ref.abs().max().item() < 3e-6


@pytest.mark.parametrize("m,k,n", [(32, 1536, 1), (1, 1536, 32), (1280, 700, 32), (32, 63, 32), (5, 4097, 7)])
def test_gemm_ex_splitk_vs_float64(m, k, n):
    """X^T dY over the batch (k = images) through cgs_gen_gemm_ex_splitk + a row sum == the float64 product; the slab rows are exactly
    the partial products of the K shares."""
    from cgs_amd import _lib, generic
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(m + k + n)
    x = torch.randn(k, m, generator=g)          # A(m, k) = x[k][m]: sxm = 1, sxk = m
    w = torch.randn(k, n, generator=g)          # B(k, n) = w[k][n]: swk = n, swn = 1
    nsplit = max(1, min(32, k // 64))
    slab = torch.full((nsplit, m * n), float("nan"), device=dev)
    x_d, w_d = x.to(dev), w.to(dev)
    _lib.call("cgs_gen_gemm_ex_splitk", m, k, n, generic._p(x_d), 1, m, generic._p(w_d), n, 1, nsplit, generic._p(slab),
              generic._s())
    torch.cuda.synchronize()
    got = slab.double().sum(0).cpu().view(m, n)
    ref = x.double().t() @ w.double()
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() / ref.abs().max().item() < 2e-6
    kper = (k + nsplit - 1) // nsplit
    part0 = x[:kper].double().t() @ w[:kper].double()
    assert (slab[0].double().cpu().view(m, n) - part0).abs().max().item() / max(part0.abs().max().item(), 1e-9) < 2e-6


def test_gemm_ex_splitk_bad_arguments():
    from cgs_amd import _lib
    lib = _lib.load()
    assert lib.cgs_gen_gemm_ex_splitk(4, 4, 4, None, 1, 4, None, 4, 1, 2, None, None) != 0
    x = torch.zeros(16, device="cuda:0")
    p = C.c_void_p(x.data_ptr())
    assert lib.cgs_gen_gemm_ex_splitk(4, 4, 4, p, 1, 4, p, 4, 1, 0, p, None) != 0


@pytest.mark.parametrize("n,co,u8", [(5, 40, True), (3, 40, False), (200, 16, False), (2, 24, True), (2, 32, False), (130, 40, False)])
def test_gen_enc0_dedicated_weight_gradient_vs_float64_autograd(n, co, u8):
    """cgs_gen_enc0_bwd_weight (features.0 of chfak 2 .. 5 on its own kernel: 4x4x1 outer products, block = pixel, pooled dY + argmax bytes
    selected on the fly) against float64 autograd; every slab element written; same bits on a second call; n = 200 / 130: more strips than
    persistent workgroups."""
    from cgs_amd import _lib, generic
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(77 * co + n)
    a = torch.randint(0, 256, (n, 64, 64, 3), dtype=torch.uint8, generator=g) if u8 else torch.randn(n, 64, 64, 3, generator=g)
    dE = torch.randn(n, 32, 32, co, generator=g)
    am = torch.randint(0, 5, (n, 32, 32, co), dtype=torch.uint8, generator=g)
    am = torch.where(torch.rand(am.shape, generator=g) < 0.1, am | 4, am)
    full = torch.zeros(n, 32, 2, 32, 2, co)
    for py in range(2):
        for px in range(2):
            full[:, :, py, :, px, :] = torch.where(am == 2 * py + px, dE, torch.zeros(()))
    ref = _ref_wgrad(a, None, 1, full.view(n, 64, 64, co), co)
    lib = _lib.load()
    nsl = lib.cgs_gen_enc0_bwd_weight_slabs(n, co)
    assert nsl >= 1 and lib.cgs_gen_enc0_bwd_weight_slabs(n, 8) == 0
    cnt = 27 * co + co
    a_d, de_d, am_d = a.to(dev), dE.to(dev), am.to(dev)
    slab = torch.full((nsl, cnt), float("nan"), device=dev)
    _lib.call("cgs_gen_enc0_bwd_weight", n, co, int(u8), generic._p(a_d), generic._p(de_d), generic._p(am_d), generic._p(slab), generic._s())
    torch.cuda.synchronize()
    got = slab.double().sum(0).cpu()
    assert torch.isfinite(got).all(), "an element of a slab row was not written"
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    assert err < 3e-6, f"weight gradient off by {err:.2e} of its maximum"
    slab2 = torch.zeros_like(slab)
    _lib.call("cgs_gen_enc0_bwd_weight", n, co, int(u8), generic._p(a_d), generic._p(de_d), generic._p(am_d), generic._p(slab2), generic._s())
    torch.cuda.synchronize()
    assert torch.equal(slab, slab2)
