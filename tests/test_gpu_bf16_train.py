"""bf16 training kernels of BASELINE config 5 (the build-defined 128x128 variant; csrc/gen_bf16_train.hip, hourglass128.py).
PARITY UNPINNED by construction (no reference counterpart: nets.py:184,189-190 cannot take 128x128 frames): the kernels are checked
one by one against float64 torch ops on the SAME bf16-rounded operands (so only the accumulation order differs), the training step
against the autograd of the build's own fp32 restatement (oracle.hourglass128_phase2_loss) with the bf16 tolerance stated there."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import hourglass_ref as orc


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def S():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def bf(t):          # round to bf16 and back (float64 values that are exactly representable in bf16)
    return t.to(torch.bfloat16).to(torch.float64)


def cat_up(a, b, ups):      # NCHW float64
    if b is None:
        return a
    return torch.cat((a, F.interpolate(b, scale_factor=ups, mode="nearest")), 1)


WG_SHAPES = [  # (n, hw, ca, cb, ups, co, a_kind, dyc)
    (3, 128, 3, 0, 2, 8, 1, 8), (2, 128, 3, 8, 2, 16, 1, 16), (2, 128, 16, 0, 2, 1, 0, 4), (3, 64, 8, 8, 2, 8, 0, 8), (5, 32, 8, 0, 2, 8, 0, 8),
    (5, 16, 8, 8, 2, 8, 0, 8), (11, 8, 8, 16, 2, 8, 0, 8), (9, 8, 8, 0, 2, 16, 0, 16), (19, 4, 16, 32, 4, 16, 0, 16), (2, 128, 3, 0, 2, 8, 2, 8)]


@pytest.mark.parametrize("shape", WG_SHAPES)
def test_bf16_weight_gradient_vs_float64_autograd(shape):
    """dW / db of conv3x3(cat(A, up(B))) on v_mfma_f32_16x16x32_bf16 (K = 32 pixels through the transposing LDS read) vs float64 autograd on
    the same bf16-rounded operands: every slab element written, relative error of the sums <= 2e-5 of the tensor's maximum."""
    from cgs_amd import _lib
    n, hw, ca, cb, ups, co, a_kind, dyc = shape
    lib = _lib.load()
    g = torch.Generator().manual_seed(hw * 100 + ca)
    if a_kind == 1:
        a_dev = torch.randint(0, 256, (n, hw, hw, ca), dtype=torch.uint8, generator=g)
        a_ref = bf(a_dev.double() / 255.0)      # the loader rounds frame / 255 to bf16
    elif a_kind == 2:
        a_dev = torch.rand((n, hw, hw, ca), generator=g)
        a_ref = bf(a_dev.double())
    else:
        a_dev = (torch.randn((n, hw, hw, ca), generator=g)).to(torch.bfloat16)
        a_ref = a_dev.double()
    b_dev = torch.randn((n, hw // ups, hw // ups, cb), generator=g).to(torch.bfloat16) if cb else None
    dy_dev = torch.zeros((n, hw, hw, dyc), dtype=torch.bfloat16)
    dy_dev[..., :co] = (torch.randn((n, hw, hw, co), generator=g) * 0.1).to(torch.bfloat16)
    x = cat_up(a_ref.permute(0, 3, 1, 2), b_dev.double().permute(0, 3, 1, 2) if cb else None, ups)
    w = torch.zeros((co, ca + cb, 3, 3), dtype=torch.float64, requires_grad=True)
    bias = torch.zeros(co, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x, w, bias, padding=1)
    (y * dy_dev[..., :co].double().permute(0, 3, 1, 2)).sum().backward()
    nsl = lib.cgs_bf16_conv3x3_bwd_weight_slabs(n, hw, ca, cb)
    cnt = 9 * (ca + cb) * co + co
    slab = torch.full((nsl, cnt), float("nan"), device="cuda")
    a_g, b_g, dy_g = a_dev.cuda(), (b_dev.cuda() if cb else None), dy_dev.cuda()      # (kept alive: P() takes raw addresses)
    _lib.call("cgs_bf16_conv3x3_bwd_weight", n, hw, ca, cb, co, dyc, a_kind, ups, P(a_g), P(b_g), P(dy_g), P(slab), S())
    torch.cuda.synchronize()
    assert torch.isfinite(slab).all(), "a slab element was not written"
    got = slab.double().sum(0).cpu()
    ref = torch.cat((w.grad.permute(2, 3, 1, 0).reshape(-1), bias.grad))
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    print(f"{shape}: max err / max |ref| = {err:.2e}")
    assert err < 2e-5


HWG_SHAPES = [  # (n, hw, ca, cb, co, a_kind): the dedicated large-map shapes (csrc/hwgrad.hip); n ragged against the persistent grid
    (3, 128, 3, 0, 8, 1), (5, 128, 3, 0, 8, 2), (3, 128, 3, 8, 16, 1), (3, 128, 16, 0, 1, 0), (7, 64, 8, 0, 8, 0), (5, 64, 8, 8, 8, 0),
    (300, 64, 8, 0, 8, 0), (9, 32, 8, 0, 8, 0), (1100, 32, 8, 8, 8, 0)]


@pytest.mark.parametrize("shape", HWG_SHAPES)
def test_bf16_large_map_weight_gradient_vs_float64_autograd(shape):
    """cgs_bf16_hwgrad (the dedicated kernels of the 128x128 / 64x64 layers) vs float64 autograd on the same bf16-rounded operands:
    every slab element written, relative error of the sums <= 2e-5 of the tensor's maximum (n = 300 at 64x64: more strips than
    persistent workgroups, every workgroup walks several)."""
    from cgs_amd import _lib
    n, hw, ca, cb, co, a_kind = shape
    lib = _lib.load()
    g = torch.Generator().manual_seed(hw * 7 + ca + 31 * cb)
    if a_kind == 1:
        a_dev = torch.randint(0, 256, (n, hw, hw, ca), dtype=torch.uint8, generator=g)
        a_ref = bf(a_dev.double() / 255.0)
    elif a_kind == 2:
        a_dev = torch.rand((n, hw, hw, ca), generator=g)
        a_ref = bf(a_dev.double())
    else:
        a_dev = torch.randn((n, hw, hw, ca), generator=g).to(torch.bfloat16)
        a_ref = a_dev.double()
    b_dev = torch.randn((n, hw // 2, hw // 2, cb), generator=g).to(torch.bfloat16) if cb else None
    if co == 1:
        dy_dev = torch.randn((n, hw, hw), generator=g) * 0.1            # fp32 single-channel gradient, rounded to bf16 by the loader
        dy_ref = bf(dy_dev.double()).reshape(n, 1, hw, hw)
    else:
        dy_dev = (torch.randn((n, hw, hw, co), generator=g) * 0.1).to(torch.bfloat16)
        dy_ref = dy_dev.double().permute(0, 3, 1, 2)
    x = cat_up(a_ref.permute(0, 3, 1, 2), b_dev.double().permute(0, 3, 1, 2) if cb else None, 2)
    w = torch.zeros((co, ca + cb, 3, 3), dtype=torch.float64, requires_grad=True)
    bias = torch.zeros(co, dtype=torch.float64, requires_grad=True)
    (F.conv2d(x, w, bias, padding=1) * dy_ref).sum().backward()
    nsl = lib.cgs_bf16_hwgrad_slabs(n, hw, ca, cb, co)
    assert nsl > 0
    cnt = 9 * (ca + cb) * co + co
    slab = torch.full((nsl, cnt), float("nan"), device="cuda")
    a_g, b_g, dy_g = a_dev.cuda(), (b_dev.cuda() if cb else None), dy_dev.cuda()
    _lib.call("cgs_bf16_hwgrad", n, hw, ca, cb, co, a_kind, P(a_g), P(b_g), P(dy_g), P(slab), S())
    torch.cuda.synchronize()
    assert torch.isfinite(slab).all(), "a slab element was not written"
    got = slab.double().sum(0).cpu()
    ref = torch.cat((w.grad.permute(2, 3, 1, 0).reshape(-1), bias.grad))
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    print(f"{shape}: max err / max |ref| = {err:.2e}")
    assert err < 2e-5
    assert lib.cgs_bf16_hwgrad_slabs(n, 16, ca, cb, co) == 0            # not a dedicated shape: the caller takes the generic kernel


@pytest.mark.parametrize("shape", [(2, 128, 3, 8, 16), (3, 64, 8, 8, 8), (4, 8, 8, 0, 16), (3, 128, 16, 0, 1), (5, 4, 16, 32, 16)])
def test_bf16_data_gradient_is_forward_kernel_on_transposed_pack(shape):
    """d cat(A, up(B)) = conv3x3(dY) with flipped / transposed weights (cgs_genbf16_pack_weights_t + the forward kernel) vs float64
    autograd with bf16-rounded weights and dY: bf16 output -> 2^-8 relative + accumulation."""
    from cgs_amd import _lib
    n, hw, ca, cb, co = shape
    lib = _lib.load()
    ci = ca + cb
    g = torch.Generator().manual_seed(hw + ci)
    w = torch.randn((co, ci, 3, 3), generator=g) * 0.2
    dyc = (co + 3) // 4 * 4
    dy = torch.zeros((n, hw, hw, dyc), dtype=torch.bfloat16)
    dy[..., :co] = torch.randn((n, hw, hw, co), generator=g).to(torch.bfloat16)
    x = torch.zeros((n, ci, hw, hw), dtype=torch.float64, requires_grad=True)
    (F.conv2d(x, bf(w), None, padding=1) * dy[..., :co].double().permute(0, 3, 1, 2)).sum().backward()
    w_hwio = w.permute(2, 3, 1, 0).contiguous().cuda()      # (kept alive until the synchronize below)
    w16 = torch.empty(lib.cgs_gen16_packed_weight_halves(co, 0, ci), device="cuda", dtype=torch.bfloat16)
    _lib.call("cgs_genbf16_pack_weights_t", ci, co, P(w_hwio), P(w16), S())
    out = torch.empty((n, hw, hw, ci), device="cuda", dtype=torch.bfloat16)
    zb = torch.zeros(64, device="cuda")
    dy_g = dy.cuda()
    _lib.call("cgs_genbf16_conv3x3_fwd_train", n, hw, dyc, 0, ci, 0, 1, _lib.ACT_NONE, 0.01, 0, 0, P(dy_g), None, P(w16), P(zb), P(out), None, S())
    torch.cuda.synchronize()
    ref = x.grad.permute(0, 2, 3, 1)
    err = (out.double().cpu() - ref).abs().max().item() / ref.abs().max().item()
    print(f"{shape}: max err / max |ref| = {err:.2e}")
    assert err < 6e-3


def test_bf16_forward_codes_pool_expand_cat_split_lrelu_convert():
    """The forward kernel's argmax bytes == max_pool2d's indices (first maximum wins, 4 where the pooled value is <= 0), and the
    element-wise backward steps against their torch forms (bit-exact up to the bf16 rounding of sums)."""
    from cgs_amd import _lib
    lib = _lib.load()
    n, hw, ca, co = 3, 32, 8, 8
    g = torch.Generator().manual_seed(7)
    x = torch.randn((n, hw, hw, ca), generator=g).to(torch.bfloat16)
    x[0, :8, :8] = 0.5                    # a flat patch: ties
    w = torch.randn((co, ca, 3, 3), generator=g) * 0.2
    b = torch.randn(co, generator=g) * 0.1
    w16 = torch.empty(lib.cgs_gen16_packed_weight_halves(ca, 0, co), device="cuda", dtype=torch.bfloat16)
    w_g, x_g, b_g = w.permute(2, 3, 1, 0).contiguous().cuda(), x.cuda(), b.cuda()
    _lib.call("cgs_genbf16_pack_weights", ca, 0, co, P(w_g), P(w16), S())
    out = torch.empty((n, hw // 2, hw // 2, co), device="cuda", dtype=torch.bfloat16)
    codes = torch.full((n, hw // 2, hw // 2, co), 99, device="cuda", dtype=torch.uint8)
    _lib.call("cgs_genbf16_conv3x3_fwd_train", n, hw, ca, 0, co, 0, 1, _lib.ACT_RELU, 0.01, 1, 0, P(x_g), None, P(w16), P(b_g), P(out), P(codes), S())
    torch.cuda.synchronize()
    pre = F.relu(F.conv2d(x.double().permute(0, 3, 1, 2), bf(w), b.double(), padding=1))
    pooled, idx = F.max_pool2d(pre, 2, return_indices=True)
    yy, xx = idx // hw, idx % hw
    ref_code = ((yy % 2) * 2 + (xx % 2)).permute(0, 2, 3, 1)
    ref_code = torch.where(pooled.permute(0, 2, 3, 1) > 0, ref_code, torch.full_like(ref_code, 4))
    got_code = codes.cpu().long()
    # a code may differ where two window values are equal up to the fp32 summation order: the flat patch (exact ties) must agree
    agree = (got_code == ref_code).double().mean().item()
    assert agree > 0.999 and torch.equal(got_code[0, :3, :3], ref_code[0, :3, :3]), agree
    assert (out.double().cpu() - pooled.permute(0, 2, 3, 1)).abs().max().item() < 2e-2 * pooled.abs().max().item()
    # pool_expand
    dp = torch.randn((n, hw // 2, hw // 2, co), generator=g).to(torch.bfloat16)
    add = torch.randn((n, hw // 2, hw // 2, co), generator=g).to(torch.bfloat16)
    full = torch.empty((n, hw, hw, co), device="cuda", dtype=torch.bfloat16)
    dp_g, add_g = dp.cuda(), add.cuda()
    _lib.call("cgs_bf16_pool_expand", n, hw // 2, co, P(dp_g), P(add_g), P(codes), P(full), S())
    torch.cuda.synchronize()
    tot = (dp.float() + add.float()).to(torch.bfloat16)
    ref = torch.zeros((n, hw, hw, co), dtype=torch.bfloat16)
    for pos in range(4):
        ref[:, pos // 2::2, pos % 2::2] = torch.where(got_code == pos, tot, torch.zeros_like(tot))
    assert torch.equal(full.cpu(), ref)
    # cat_split (ups 2 and 4, bf16 and fp32 low)
    for hw2, ca2, cb2, ups, lowf in ((16, 3, 8, 2, 0), (4, 16, 32, 4, 1), (8, 8, 16, 2, 0)):
        dcat = torch.randn((n, hw2, hw2, ca2 + cb2), generator=g).to(torch.bfloat16)
        dskip = torch.empty((n, hw2, hw2, ca2), device="cuda", dtype=torch.bfloat16)
        dlow = torch.empty((n, hw2 // ups, hw2 // ups, cb2), device="cuda", dtype=torch.float32 if lowf else torch.bfloat16)
        dcat_g = dcat.cuda()
        _lib.call("cgs_bf16_cat_split", n, hw2, ca2, cb2, ups, P(dcat_g), P(dskip), P(dlow), lowf, S())
        torch.cuda.synchronize()
        assert torch.equal(dskip.cpu(), dcat[..., :ca2].contiguous())
        rl = dcat[..., ca2:].double().reshape(n, hw2 // ups, ups, hw2 // ups, ups, cb2).sum((2, 4))
        assert (dlow.double().cpu() - rl).abs().max().item() <= (1e-5 if lowf else 1.6e-2) * rl.abs().max().item()
    # lrelu_bwd, convert
    d = torch.randn(1000, generator=g).to(torch.bfloat16)
    h = torch.randn(1000, generator=g).to(torch.bfloat16)
    dd = d.cuda().clone()
    h_g = h.cuda()
    _lib.call("cgs_bf16_lrelu_bwd", 1000, P(dd), P(h_g), 0.01, S())
    assert torch.equal(dd.cpu(), torch.where(h.float() > 0, d.float(), d.float() * 0.01).to(torch.bfloat16))
    src = torch.randn((50, 3), generator=g)
    dst = torch.empty((50, 4), device="cuda", dtype=torch.bfloat16)
    src_g = src.cuda()
    _lib.call("cgs_bf16_convert", 50, 3, 4, 0, P(src_g), P(dst), S())
    back = torch.empty((50, 4), device="cuda")
    _lib.call("cgs_bf16_convert", 50, 4, 4, 1, P(dst), P(back), S())
    torch.cuda.synchronize()
    ref4 = torch.zeros((50, 4))
    ref4[:, :3] = src
    assert torch.equal(dst.cpu(), ref4.to(torch.bfloat16)) and torch.equal(back.cpu(), ref4.to(torch.bfloat16).float())


@pytest.mark.parametrize("src", ["u8", "f32"])
def test_bf16_enc0_whole_strip_kernel_vs_float64_conv(src):
    """cgs_bf16_enc0_fwd (features.0 of config 5 on the whole-strip kernel, csrc/hconv.hip) vs the float64 convolution of the same
    bf16-rounded operands: pooled output within one bf16 rounding, argmax bytes == max_pool2d's indices; n = 9 is ragged against
    the persistent grid's strip loop."""
    from cgs_amd import _lib
    _lib.load()
    n, hw = 9, 128
    g = torch.Generator().manual_seed(128)
    if src == "u8":
        x = torch.randint(0, 256, (n, hw, hw, 3), dtype=torch.uint8, generator=g)
        x[0, :16, :16] = 77               # a flat patch: exact ties, the first maximum wins
        xr = bf(x.double() / 255.0)
    else:
        x = torch.rand((n, hw, hw, 3), generator=g)
        x[0, :16, :16] = 0.25
        xr = bf(x.double())
    w = torch.randn((8, 3, 3, 3), generator=g) * 0.3
    b = torch.randn(8, generator=g) * 0.1
    b[0] = 0.5                            # a channel whose flat patch is positive, so the tie rule is exercised
    w_g, x_g, b_g = w.permute(2, 3, 1, 0).contiguous().cuda(), x.cuda(), b.cuda()
    out = torch.full((n, hw // 2, hw // 2, 8), 7.0, device="cuda", dtype=torch.bfloat16)
    codes = torch.full((n, hw // 2, hw // 2, 8), 99, device="cuda", dtype=torch.uint8)
    _lib.call("cgs_bf16_enc0_fwd", n, P(x_g), int(src == "f32"), P(w_g), P(b_g), P(out), P(codes), S())
    torch.cuda.synchronize()
    pre = F.relu(F.conv2d(xr.permute(0, 3, 1, 2), bf(w), b.double(), padding=1))
    pooled, idx = F.max_pool2d(pre, 2, return_indices=True)
    pooled = pooled.permute(0, 2, 3, 1)
    ref_code = (((idx // hw) % 2) * 2 + (idx % hw) % 2).permute(0, 2, 3, 1)
    ref_code = torch.where(pooled > 0, ref_code, torch.full_like(ref_code, 4))
    got_code = codes.cpu().long()
    agree = (got_code == ref_code).double().mean().item()
    assert agree > 0.999 and torch.equal(got_code[0, 1:7, 1:7], ref_code[0, 1:7, 1:7]), agree
    err = (out.double().cpu() - pooled).abs().max().item()
    assert err <= 2.0 ** -8 * pooled.abs().max().item(), err       # half a bf16 ulp of the largest value (+ fp32 summation order)
    # and without the codes pointer (the inference form): same output
    out2 = torch.empty_like(out)
    _lib.call("cgs_bf16_enc0_fwd", n, P(x_g), int(src == "f32"), P(w_g), P(b_g), P(out2), None, S())
    torch.cuda.synchronize()
    assert torch.equal(out, out2)


@pytest.mark.parametrize("n", [3, 70])
def test_h5conv_large_map_layers_vs_float64(n):
    """The five 128x128 kernels of config 5's training step (h5conv_kernel, csrc/hconv.hip) vs float64 torch ops on the same bf16-rounded
    operands: forward layers with their activations, data gradients with the fused LeakyReLU' factor / 2x2 cell sums.  bf16 outputs:
    within half a bf16 ulp of the largest value (2^-8 relative to it); fp32 outputs: 1e-5.  n = 70: more strips than persistent
    workgroups (every workgroup walks several)."""
    from cgs_amd import _lib
    _lib.load()
    g = torch.Generator().manual_seed(500 + n)
    hwio = lambda w: w.permute(2, 3, 1, 0).contiguous().cuda()
    relmax = lambda got, ref: (got.double().cpu() - ref).abs().max().item() / ref.abs().max().item()
    # masker.0 forward
    x = torch.randint(0, 256, (n, 128, 128, 3), dtype=torch.uint8, generator=g)
    o0 = torch.randn((n, 64, 64, 8), generator=g).to(torch.bfloat16)
    w0 = torch.randn((16, 11, 3, 3), generator=g) * 0.2
    b0 = torch.randn(16, generator=g) * 0.1
    x_g, o0_g, w0_g, b0_g = x.cuda(), o0.cuda(), hwio(w0), b0.cuda()
    hm_g = torch.full((n, 128, 128, 16), 7.0, device="cuda", dtype=torch.bfloat16)
    _lib.call("cgs_bf16_mask0_fwd", n, P(x_g), P(o0_g), P(w0_g), P(b0_g), P(hm_g), S())
    cat = cat_up(bf(x.double() / 255.0).permute(0, 3, 1, 2), o0.double().permute(0, 3, 1, 2), 2)
    ref = F.leaky_relu(F.conv2d(cat, bf(w0), b0.double(), padding=1), 0.01).permute(0, 2, 3, 1)
    assert relmax(hm_g, ref) <= 2.0 ** -8
    # masker.2 forward (on the kernel's own hm)
    w2 = torch.randn((1, 16, 3, 3), generator=g) * 0.1
    b2 = torch.randn(1, generator=g) * 0.1
    w2_g, b2_g = hwio(w2), b2.cuda()
    z_g = torch.full((n, 128, 128), 7.0, device="cuda")
    _lib.call("cgs_bf16_mask2_fwd", n, P(hm_g), P(w2_g), P(b2_g), P(z_g), S())
    hm = hm_g.cpu()
    refz = torch.sigmoid(F.conv2d(hm.double().permute(0, 3, 1, 2), bf(w2), b2.double(), padding=1))[:, 0]
    assert (z_g.double().cpu() - refz).abs().max().item() <= 1e-5
    # features.0 image gradient
    dy = (torch.randn((n, 128, 128, 8), generator=g) * 0.1).to(torch.bfloat16)
    we = torch.randn((8, 3, 3, 3), generator=g) * 0.3
    dy_g, we_g = dy.cuda(), hwio(we)
    dx_g = torch.full((n, 128, 128, 3), 7.0, device="cuda")
    _lib.call("cgs_bf16_enc0_bwd_data", n, P(dy_g), P(we_g), P(dx_g), S())
    refdx = F.conv_transpose2d(dy.double().permute(0, 3, 1, 2), bf(we), padding=1).permute(0, 2, 3, 1)
    assert relmax(dx_g, refdx) <= 1e-5
    # masker.2 data gradient x LeakyReLU'(hm)
    dz = torch.randn((n, 128, 128), generator=g) * 0.1
    dz_g = dz.cuda()
    dhm_g = torch.full((n, 128, 128, 16), 7.0, device="cuda", dtype=torch.bfloat16)
    _lib.call("cgs_bf16_mask2_bwd_data", n, P(dz_g), P(hm_g), P(w2_g), P(dhm_g), S())
    refdh = F.conv_transpose2d(bf(dz.double()).unsqueeze(1), bf(w2), padding=1).permute(0, 2, 3, 1)
    refdh = torch.where(hm.double() > 0, refdh, 0.01 * refdh)
    assert relmax(dhm_g, refdh) <= 2.0 ** -8
    # masker.0 data gradient of the upsampled source, summed over the 2x2 cells
    do_g = torch.full((n, 64, 64, 8), 7.0, device="cuda", dtype=torch.bfloat16)
    _lib.call("cgs_bf16_mask0_bwd_data", n, P(dhm_g), P(w0_g), P(do_g), S())
    torch.cuda.synchronize()
    dcat = F.conv_transpose2d(dhm_g.cpu().double().permute(0, 3, 1, 2), bf(w0), padding=1)[:, 3:]
    refdo = dcat.reshape(n, 8, 64, 2, 64, 2).sum((3, 5)).permute(0, 2, 3, 1)
    assert relmax(do_g, refdo) <= 2.0 ** -8


def test_h5conv_64x64_layers_vs_float64():
    """cgs_bf16_h5conv: features.3 forward (+ argmax bytes) / data gradient and dec_model.0 forward / its two data gradients vs float64
    torch ops on the same bf16-rounded operands (bf16 outputs: 2^-8 of the largest value); n = 150: strips > persistent workgroups."""
    from cgs_amd import _lib
    _lib.load()
    n = 150
    g = torch.Generator().manual_seed(64)
    hwio = lambda w: w.permute(2, 3, 1, 0).contiguous().cuda()
    relmax = lambda got, ref: (got.double().cpu() - ref).abs().max().item() / ref.abs().max().item()
    e0 = torch.randn((n, 64, 64, 8), generator=g).to(torch.bfloat16)
    e0[0, :8, :8] = 0.5
    w = torch.randn((8, 8, 3, 3), generator=g) * 0.2
    b = torch.randn(8, generator=g) * 0.1
    e0_g, w_g, b_g = e0.cuda(), hwio(w), b.cuda()
    e1_g = torch.full((n, 32, 32, 8), 7.0, device="cuda", dtype=torch.bfloat16)
    codes = torch.full((n, 32, 32, 8), 99, device="cuda", dtype=torch.uint8)
    _lib.call("cgs_bf16_h5conv", _lib.H5_ENC1_FWD, n, P(e0_g), None, P(w_g), P(b_g), P(e1_g), P(codes), S())
    pre = F.relu(F.conv2d(e0.double().permute(0, 3, 1, 2), bf(w), b.double(), padding=1))
    pooled, idx = F.max_pool2d(pre, 2, return_indices=True)
    pooled = pooled.permute(0, 2, 3, 1)
    ref_code = (((idx // 64) % 2) * 2 + (idx % 64) % 2).permute(0, 2, 3, 1)
    ref_code = torch.where(pooled > 0, ref_code, torch.full_like(ref_code, 4))
    got_code = codes.cpu().long()
    assert (got_code == ref_code).double().mean().item() > 0.999 and torch.equal(got_code[0, 1:3, 1:3], ref_code[0, 1:3, 1:3])
    assert relmax(e1_g, pooled) <= 2.0 ** -8
    dy = (torch.randn((n, 64, 64, 8), generator=g) * 0.1).to(torch.bfloat16)
    dy_g = dy.cuda()
    de0_g = torch.full((n, 64, 64, 8), 7.0, device="cuda", dtype=torch.bfloat16)
    _lib.call("cgs_bf16_h5conv", _lib.H5_ENC1_BWD_DATA, n, P(dy_g), None, P(w_g), None, P(de0_g), None, S())
    assert relmax(de0_g, F.conv_transpose2d(dy.double().permute(0, 3, 1, 2), bf(w), padding=1).permute(0, 2, 3, 1)) <= 2.0 ** -8
    o1 = torch.randn((n, 32, 32, 8), generator=g).to(torch.bfloat16)
    wd = torch.randn((8, 16, 3, 3), generator=g) * 0.2
    o1_g, wd_g = o1.cuda(), hwio(wd)
    o0_g = torch.full((n, 64, 64, 8), 7.0, device="cuda", dtype=torch.bfloat16)
    _lib.call("cgs_bf16_h5conv", _lib.H5_DEC0_FWD, n, P(e0_g), P(o1_g), P(wd_g), P(b_g), P(o0_g), None, S())
    cat = cat_up(e0.double().permute(0, 3, 1, 2), o1.double().permute(0, 3, 1, 2), 2)
    assert relmax(o0_g, F.conv2d(cat, bf(wd), b.double(), padding=1).permute(0, 2, 3, 1)) <= 2.0 ** -8
    ds_g = torch.full((n, 64, 64, 8), 7.0, device="cuda", dtype=torch.bfloat16)
    dl_g = torch.full((n, 32, 32, 8), 7.0, device="cuda", dtype=torch.bfloat16)
    _lib.call("cgs_bf16_h5conv", _lib.H5_DEC0_BWD_SKIP, n, P(dy_g), None, P(wd_g), None, P(ds_g), None, S())
    _lib.call("cgs_bf16_h5conv", _lib.H5_DEC0_BWD_LOW, n, P(dy_g), None, P(wd_g), None, P(dl_g), None, S())
    torch.cuda.synchronize()
    dcat = F.conv_transpose2d(dy.double().permute(0, 3, 1, 2), bf(wd), padding=1)
    assert relmax(ds_g, dcat[:, :8].permute(0, 2, 3, 1)) <= 2.0 ** -8
    assert relmax(dl_g, dcat[:, 8:].reshape(n, 8, 32, 2, 32, 2).sum((3, 5)).permute(0, 2, 3, 1)) <= 2.0 ** -8


@pytest.mark.parametrize("with_addend", [False, True])
def test_pooled_gradient_consumers_are_bit_identical_to_pool_expand_then_kernel(with_addend):
    """cgs_bf16_hwgrad_pooled / cgs_bf16_enc0_bwd_data_pooled / CGS_H5_ENC1_BWD_DATA_POOLED re-expand the pooled gradient while they stage
    it: same bits as cgs_bf16_pool_expand followed by the kernel on the expanded tensor (which the tests above pin to float64)."""
    from cgs_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(99)
    for hw, ca, a_kind, n in ((128, 3, 1, 5), (128, 3, 2, 3), (64, 8, 0, 37), (32, 8, 0, 41)):
        hp = hw // 2
        if a_kind == 1:
            a = torch.randint(0, 256, (n, hw, hw, 3), dtype=torch.uint8, generator=g).cuda()
        elif a_kind == 2:
            a = torch.rand((n, hw, hw, 3), generator=g).cuda()
        else:
            a = torch.randn((n, hw, hw, 8), generator=g).to(torch.bfloat16).cuda()
        dp = (torch.randn((n, hp, hp, 8), generator=g) * 0.1).to(torch.bfloat16).cuda()
        add = (torch.randn((n, hp, hp, 8), generator=g) * 0.1).to(torch.bfloat16).cuda() if with_addend else None
        codes = torch.randint(0, 5, (n, hp, hp, 8), dtype=torch.uint8, generator=g).cuda()
        w = (torch.randn((9, ca, 8), generator=g) * 0.3).cuda()
        dyf = torch.empty((n, hw, hw, 8), device="cuda", dtype=torch.bfloat16)
        _lib.call("cgs_bf16_pool_expand", n, hp, 8, P(dp), P(add), P(codes), P(dyf), S())
        nsl, cnt = lib.cgs_bf16_hwgrad_slabs(n, hw, ca, 0, 8), 9 * ca * 8 + 8
        s1 = torch.full((nsl, cnt), float("nan"), device="cuda")
        s2 = torch.full((nsl, cnt), float("nan"), device="cuda")
        _lib.call("cgs_bf16_hwgrad", n, hw, ca, 0, 8, a_kind, P(a), None, P(dyf), P(s1), S())
        _lib.call("cgs_bf16_hwgrad_pooled", n, hw, ca, a_kind, P(a), P(dp), P(add), P(codes), P(s2), S())
        torch.cuda.synchronize()
        assert torch.isfinite(s1).all() and torch.equal(s1, s2)
        if hw == 128:
            d1 = torch.full((n, 128, 128, 3), 7.0, device="cuda")
            d2 = torch.full((n, 128, 128, 3), 8.0, device="cuda")
            _lib.call("cgs_bf16_enc0_bwd_data", n, P(dyf), P(w), P(d1), S())
            _lib.call("cgs_bf16_enc0_bwd_data_pooled", n, P(dp), P(add), P(codes), P(w), P(d2), S())
        elif hw == 64:
            d1 = torch.full((n, 64, 64, 8), 7.0, device="cuda", dtype=torch.bfloat16)
            d2 = torch.full((n, 64, 64, 8), 8.0, device="cuda", dtype=torch.bfloat16)
            _lib.call("cgs_bf16_h5conv", _lib.H5_ENC1_BWD_DATA, n, P(dyf), None, P(w), None, P(d1), None, S())
            _lib.call("cgs_bf16_h5conv", _lib.H5_ENC1_BWD_DATA_POOLED, n, P(dp), P(add), P(w), None, P(d2), P(codes), S())
        else:       # 32x32: only the pooled form exists; its reference = float64 transposed convolution of the expanded gradient
            d2 = torch.full((n, 32, 32, 8), 8.0, device="cuda", dtype=torch.bfloat16)
            _lib.call("cgs_bf16_h5conv", _lib.H5_ENC2_BWD_DATA_POOLED, n, P(dp), P(add), P(w), None, P(d2), P(codes), S())
            torch.cuda.synchronize()
            wt = bf(w.cpu().reshape(3, 3, 8, 8).permute(3, 2, 0, 1).double())
            ref = F.conv_transpose2d(dyf.cpu().double().permute(0, 3, 1, 2), wt, padding=1).permute(0, 2, 3, 1)
            assert (d2.double().cpu() - ref).abs().max().item() <= 2.0 ** -8 * ref.abs().max().item()
            d1 = d2
        torch.cuda.synchronize()
        assert torch.equal(d1, d2)


def test_h5conv_32x32_layers_vs_float64():
    """The 32x32 forms of cgs_bf16_h5conv (features.6 forward -> fp32 + argmax bytes, dec_model.1 forward and its two data gradients, the
    cell-summed one in fp32) vs float64 torch ops on the same bf16-rounded operands; n = 1100: images > persistent workgroups."""
    from cgs_amd import _lib
    _lib.load()
    n = 1100
    g = torch.Generator().manual_seed(32)
    hwio = lambda w: w.permute(2, 3, 1, 0).contiguous().cuda()
    relmax = lambda got, ref: (got.double().cpu() - ref).abs().max().item() / ref.abs().max().item()
    e1 = torch.randn((n, 32, 32, 8), generator=g).to(torch.bfloat16)
    w = torch.randn((8, 8, 3, 3), generator=g) * 0.2
    b = torch.randn(8, generator=g) * 0.1
    e1_g, w_g, b_g = e1.cuda(), hwio(w), b.cuda()
    e2_g = torch.full((n, 16, 16, 8), 7.0, device="cuda")
    codes = torch.full((n, 16, 16, 8), 99, device="cuda", dtype=torch.uint8)
    _lib.call("cgs_bf16_h5conv", _lib.H5_ENC2_FWD, n, P(e1_g), None, P(w_g), P(b_g), P(e2_g), P(codes), S())
    pre = F.relu(F.conv2d(e1.double().permute(0, 3, 1, 2), bf(w), b.double(), padding=1))
    pooled, idx = F.max_pool2d(pre, 2, return_indices=True)
    pooled = pooled.permute(0, 2, 3, 1)
    ref_code = (((idx // 32) % 2) * 2 + (idx % 32) % 2).permute(0, 2, 3, 1)
    ref_code = torch.where(pooled > 0, ref_code, torch.full_like(ref_code, 4))
    assert (codes.cpu().long() == ref_code).double().mean().item() > 0.999
    assert relmax(e2_g, pooled) <= 1e-5
    o2 = torch.randn((n, 16, 16, 8), generator=g).to(torch.bfloat16)
    wd = torch.randn((8, 16, 3, 3), generator=g) * 0.2
    o2_g, wd_g = o2.cuda(), hwio(wd)
    o1_g = torch.full((n, 32, 32, 8), 7.0, device="cuda", dtype=torch.bfloat16)
    _lib.call("cgs_bf16_h5conv", _lib.H5_DEC1_FWD, n, P(e1_g), P(o2_g), P(wd_g), P(b_g), P(o1_g), None, S())
    cat = cat_up(e1.double().permute(0, 3, 1, 2), o2.double().permute(0, 3, 1, 2), 2)
    assert relmax(o1_g, F.conv2d(cat, bf(wd), b.double(), padding=1).permute(0, 2, 3, 1)) <= 2.0 ** -8
    dy = (torch.randn((n, 32, 32, 8), generator=g) * 0.1).to(torch.bfloat16)
    dy_g = dy.cuda()
    ds_g = torch.full((n, 32, 32, 8), 7.0, device="cuda", dtype=torch.bfloat16)
    dl_g = torch.full((n, 16, 16, 8), 7.0, device="cuda")
    _lib.call("cgs_bf16_h5conv", _lib.H5_DEC1_BWD_SKIP, n, P(dy_g), None, P(wd_g), None, P(ds_g), None, S())
    _lib.call("cgs_bf16_h5conv", _lib.H5_DEC1_BWD_LOW, n, P(dy_g), None, P(wd_g), None, P(dl_g), None, S())
    torch.cuda.synchronize()
    dcat = F.conv_transpose2d(dy.double().permute(0, 3, 1, 2), bf(wd), padding=1)
    assert relmax(ds_g, dcat[:, :8].permute(0, 2, 3, 1)) <= 2.0 ** -8
    assert relmax(dl_g, dcat[:, 8:].reshape(n, 8, 16, 2, 16, 2).sum((3, 5)).permute(0, 2, 3, 1)) <= 1e-5


def _grad_dicts(net):
    """The flat gradient buffer of Hourglass128 as (critic, masker) dicts in the oracle's shapes."""
    saved = net.flat.clone()
    net.flat.copy_(net.grad)
    out = net.state_dicts()
    net.flat.copy_(saved)
    return out


@pytest.mark.parametrize("n,chfak,generic", [(6, 1, False), (256, 1, False), (6, 1, True)])
def test_config5_training_step_vs_autograd_of_the_build_restatement(n, chfak, generic):
    """BASELINE config 5 as a TRAINING step (batch 256 = its stated size; 6 = a ragged small case): Hourglass128.phase2_step -- bf16
    activations / gradients, fp32 accumulate, fp32 master weights -- against the fp32 CPU autograd of oracle.hourglass128_phase2_loss.
    PARITY UNPINNED (no reference counterpart).  Stated bf16 tolerance: loss parts within 2e-2 relative (+1e-4), every gradient
    tensor with cosine similarity >= 0.999 to the fp32 gradient and relative L2 error <= 0.06 at n = 256 (>= 0.995 / <= 0.12 at n = 6: twice
    the measured worst, 0.99962 / 0.030 at n = 256 and 0.99798 / 0.064 at n = 6; bf16 carries 8 significant bits and the
    mask head sums 16 k pixels per weight; a pool-argmax flip between the bf16 and fp32 forward moves a whole window's gradient),
    parameters after the Adam step within 2.2e-3 absolute (lr = 1e-3: the update is +-lr).
    Kernel paths: chfak 1 on its dedicated whole-strip / tail kernels (the benchmarked form); the same model with those switched off
    (generic = True) on the shape-generic 16-bit family (chfak != 1 is an inference-only configuration: see the test below)."""
    from cgs_amd import hourglass128
    pc = orc.seeded_params(orc.critic128_shapes(chfak), 31)
    pm = orc.seeded_params(orc.masker128_shapes(chfak), 32)
    flags = ("TAIL", "H5CONV", "HWGRAD", "POOL_FUSED", "ENC0_DIRECT")
    saved_flags = {k: getattr(hourglass128, k) for k in flags}
    if generic:
        for k in flags:
            setattr(hourglass128, k, False)
    rs = np.random.RandomState(n)
    A = rs.randint(0, 256, (n, 128, 128, 3)).astype(np.uint8)
    B = rs.randint(0, 256, (n, 128, 128, 3)).astype(np.uint8)
    A[: n // 2] = (A[: n // 2] * 0.4).astype(np.uint8)
    Y = rs.rand(n).astype(np.float32)
    try:
        net = hourglass128.Hourglass128(pc, pm, chfak=chfak)
        assert net.tail == (chfak == 1) and net.h5 == (chfak == 1)
        losses = net.phase2_step(torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda(), torch.from_numpy(Y).cuda(), use_graph=False)
        torch.cuda.synchronize()
        got_l = losses.cpu().tolist()
        gc, gm = _grad_dicts(net)
        # a second and third step through the HIP graph run and stay finite (same kernel path)
        l2 = net.phase2_step(use_graph=True)
        l3 = net.phase2_step()
        torch.cuda.synchronize()
        assert torch.isfinite(l2).all() and torch.isfinite(l3).all() and int(net.step_t.item()) == 3
    finally:
        for k, v in saved_flags.items():
            setattr(hourglass128, k, v)
    torch.set_num_threads(16)
    Pc, Pm = orc.leafify(pc), orc.leafify(pm)
    total, parts, Z, pred = orc.hourglass128_phase2_loss(Pc, Pm, torch.from_numpy(A).permute(0, 3, 1, 2).float() / 255.0,
                                                         torch.from_numpy(B).permute(0, 3, 1, 2).float() / 255.0, torch.from_numpy(Y))
    total.backward()
    want_l = [float(parts["critic"].detach()), float(parts["replace"].detach()), float(parts["inject"].detach()), float(parts["norm"].detach()), 0.0, float(total.detach())]
    print("losses", got_l[:6], "oracle", want_l)
    for a, b in zip(got_l[:6], want_l):
        assert abs(a - b) <= 2e-2 * abs(b) + 1e-4, (got_l, want_l)
    worst = (1.0, "", 0.0)
    # bounds = 2x what is measured (round-4 review): n = 256 (the stated batch): 0.99962 / 0.030 -> 0.999 / 0.06; n = 6: 0.99798 / 0.064 -> 0.995 / 0.12
    cos_min, rel_max = (0.999, 0.06) if n >= 256 else (0.995, 0.12)
    for got, ref in ((gc, Pc), (gm, Pm)):
        for k, t in ref.items():
            g, r = got[k].double().cpu().reshape(-1), t.grad.double().reshape(-1)
            cos = float((g @ r) / (g.norm() * r.norm() + 1e-30))
            rel = float((g - r).norm() / (r.norm() + 1e-30))
            if cos < worst[0]:
                worst = (cos, k, rel)
            assert cos >= cos_min and rel <= rel_max, f"{k}: cosine {cos:.4f}, relative L2 error {rel:.3f} (bounds {cos_min} / {rel_max})"
    print(f"n={n}: worst gradient cosine {worst[0]:.5f} ({worst[1]}, relative L2 error {worst[2]:.3f})")
    # three Adam steps moved every parameter by at most ~3 lr
    sc, sm = net.state_dicts()
    for got, ref in ((sc, pc), (sm, pm)):
        for k, t in ref.items():
            assert (got[k].cpu() - t).abs().max().item() <= 3 * 2.2e-3, k


def test_config5_chfak2_is_inference_only():
    """chfak 2 of the 128x128 variant: infer() runs on the shape-generic 16-bit kernels (checked against the fp32 restatement with the bf16
    bound of the chfak-1 inference test); phase2_step refuses with a message instead of a bad-argument error from a kernel launcher."""
    from cgs_amd import hourglass128, _lib
    pc = orc.seeded_params(orc.critic128_shapes(2), 31)
    pm = orc.seeded_params(orc.masker128_shapes(2), 32)
    rs = np.random.RandomState(5)
    X = rs.randint(0, 256, (4, 128, 128, 3)).astype(np.uint8)
    net = hourglass128.Hourglass128(pc, pm, chfak=2)
    assert not net.tail and not net.h5
    pred, Z = net.infer(torch.from_numpy(X).cuda())
    torch.cuda.synchronize()
    with torch.no_grad():
        rp, rz = orc.hourglass128_apply(pc, pm, torch.from_numpy(X).permute(0, 3, 1, 2).float() / 255.0)
    assert (pred.cpu() - rp.reshape(-1)).abs().max().item() < 2e-2
    assert (Z.cpu() - rz.reshape(4, 128, 128)).abs().max().item() < 4e-2
    with pytest.raises(_lib.CgsError, match="chfak 1"):
        net.phase2_step(torch.from_numpy(X).cuda(), torch.from_numpy(X).cuda(), torch.rand(4).cuda())


DP128_WORKER = r"""
import os, sys
sys.path.insert(0, {repo!r})
import numpy as np, torch
import torch.distributed as dist
import cgs_amd
from cgs_amd import parallel, hourglass128
from oracle import hourglass_ref as orc
pg = parallel.init_from_env("gloo")          # two ranks share the one GPU of the box; the bucket is staged through the host
rank, _, world = parallel.env_world()
pc = orc.seeded_params(orc.critic128_shapes(), 31)
pm = orc.seeded_params(orc.masker128_shapes(), 32)
rs = np.random.RandomState(12)
n = 8
A = torch.from_numpy(rs.randint(0, 256, (n, 128, 128, 3)).astype(np.uint8)).cuda()
B = torch.from_numpy(rs.randint(0, 256, (n, 128, 128, 3)).astype(np.uint8)).cuda()
Y = torch.from_numpy(rs.rand(n).astype(np.float32)).cuda()
sl = parallel.shard_slice(n, rank, world)
net = hourglass128.Hourglass128(pc, pm, process_group=pg)
for _ in range(2):
    net.phase2_step(A[sl].contiguous(), B[sl].contiguous(), Y[sl].contiguous())
torch.cuda.synchronize()
flat = net.flat.cpu()
others = [torch.empty_like(flat) for _ in range(world)]
dist.all_gather(others, flat)
assert all(torch.equal(o, others[0]) for o in others), "replicas diverged"
if rank == 0:
    np.save({out!r}, flat.numpy())
dist.barrier(); dist.destroy_process_group()
"""


def test_config5_training_data_parallel_world2(tmp_path):
    """Hourglass128.phase2_step under data parallelism (process_group: one flat fp32 gradient bucket, sum all-reduce, Adam reads it
    scaled by 1 / world): two ranks (gloo rendezvous on this box's GPU), half the batch each, 2 steps -- replicas bit-identical, and
    equal to the single-process full-batch run up to the bf16 path's summation order (mean |d parameter| < 5e-5, 99.9 % of the
    parameters within 2.1e-3: Adam's first steps move a weight by ~lr = 1e-3 whatever the gradient's size, so a near-zero gradient
    whose sign differs between the two summation orders shows up as 2 lr)."""
    import os, subprocess, sys
    from cgs_amd import hourglass128
    REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "dp128_flat.npy")
    script = tmp_path / "dp128_worker.py"
    script.write_text(DP128_WORKER.format(repo=REPO, out=out))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29543", str(script)], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    pc = orc.seeded_params(orc.critic128_shapes(), 31)
    pm = orc.seeded_params(orc.masker128_shapes(), 32)
    rs = np.random.RandomState(12)
    n = 8
    A = torch.from_numpy(rs.randint(0, 256, (n, 128, 128, 3)).astype(np.uint8)).cuda()
    B = torch.from_numpy(rs.randint(0, 256, (n, 128, 128, 3)).astype(np.uint8)).cuda()
    Y = torch.from_numpy(rs.rand(n).astype(np.float32)).cuda()
    net = hourglass128.Hourglass128(pc, pm)
    for _ in range(2):
        net.phase2_step(A, B, Y, use_graph=False)
    torch.cuda.synchronize()
    got, want = np.load(out), net.flat.cpu().numpy()
    d = np.abs(got - want)
    print(f"DP(2) vs single process after 2 steps: max |d param| {d.max():.2e}, mean {d.mean():.2e}")
    assert d.mean() < 5e-5 and np.quantile(d, 0.999) < 2.1e-3


DP128_RCCL_WORKER = r"""
import os, sys
sys.path.insert(0, {repo!r})
import numpy as np, torch
import torch.distributed as dist
import cgs_amd
from cgs_amd import hourglass128
from oracle import hourglass_ref as orc
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0), rank=0, world_size=1)       # nccl == RCCL on ROCm
pc = orc.seeded_params(orc.critic128_shapes(), 31)
pm = orc.seeded_params(orc.masker128_shapes(), 32)
rs = np.random.RandomState(12)
n = {n}
A = torch.from_numpy(rs.randint(0, 256, (n, 128, 128, 3)).astype(np.uint8)).cuda()
B = torch.from_numpy(rs.randint(0, 256, (n, 128, 128, 3)).astype(np.uint8)).cuda()
Y = torch.from_numpy(rs.rand(n).astype(np.float32)).cuda()
res = {{}}
for name, kw in (("single", dict()), ("dp_graph", dict(process_group=dist.group.WORLD, force_allreduce=True)),
                 ("dp_eager", dict(process_group=dist.group.WORLD, force_allreduce=True, dp_graph=False))):
    net = hourglass128.Hourglass128(pc, pm, **kw)
    for _ in range(4):
        net.phase2_step(A, B, Y)
    torch.cuda.synchronize()
    res[name] = dict(flat=net.flat.cpu().numpy(), m=net.m.cpu().numpy(), t=int(net.step_t.item()), losses=net._train.losses.cpu().numpy(),
                     single_graph=bool(net.dp_single_graph), note=net.dp_capture_note, graphed=not isinstance(net._graph, str))
np.save({out!r}, np.array([res], dtype=object), allow_pickle=True)
dist.destroy_process_group()
"""


@pytest.mark.parametrize("n", [8, 256])
def test_config5_dp_launch_form_on_one_rank_rccl_group(tmp_path, n):
    """(n = 256: config 5's stated batch.)  Hourglass128's data-parallel launch form on a ONE-rank RCCL group: with the all-reduce recorded in the step's HIP graph (one graph
    launch per step) and with the eager form, against the plain single-GPU step -- the same kernels in the same order plus an identity
    all-reduce and the 1 / world = 1 scale in Adam: parameters, first moments, step counter and losses after 4 steps are bit-identical."""
    import os, subprocess, sys
    REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "dp128_rccl.npy")
    script = tmp_path / "dp128_rccl_worker.py"
    script.write_text(DP128_RCCL_WORKER.format(repo=REPO, out=out, n=n))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    res = np.load(out, allow_pickle=True)[0]
    base = res["single"]
    assert base["t"] == 4 and base["graphed"]
    print(f"all-reduce in the step graph: {res['dp_graph']['single_graph']} ({res['dp_graph']['note']})")
    assert res["dp_graph"]["single_graph"] == res["dp_graph"]["graphed"] and not res["dp_eager"]["graphed"]
    for name in ("dp_graph", "dp_eager"):
        got = res[name]
        assert got["t"] == 4
        np.testing.assert_array_equal(got["losses"], base["losses"])
        np.testing.assert_array_equal(got["flat"], base["flat"])
        np.testing.assert_array_equal(got["m"], base["m"])



def test_enc0_data_gradient_with_fused_mix_backward_vs_the_two_launches():
    """cgs_bf16_enc0_bwd_mix (features.0's image gradient of the replaced and injected mixes + the mix backward in one pass over the difference
    of the two pooled gradients) against cgs_bf16_enc0_bwd_data_pooled on both mixes followed by cgs_mix_bwd: same d(pre-sigmoid mask) up to the
    bf16 rounding of (d rep - d inj) in the staged tile (the two-launch form rounds each to bf16 first) -- 2^-7 of the largest value."""
    from cgs_amd import _lib
    _lib.load()
    n = 5
    g = torch.Generator().manual_seed(41)
    dp = (torch.randn((2 * n, 64, 64, 8), generator=g) * 0.1).to(torch.bfloat16).cuda()
    codes = torch.randint(0, 5, (2 * n, 64, 64, 8), dtype=torch.uint8, generator=g).cuda()
    w = (torch.randn((9, 3, 8), generator=g) * 0.3).cuda()
    A = torch.randint(0, 256, (n, 128, 128, 3), dtype=torch.uint8, generator=g).cuda()
    B = torch.randint(0, 256, (n, 128, 128, 3), dtype=torch.uint8, generator=g).cuda()
    Z = torch.rand((n, 128, 128), generator=g).cuda()
    l1s, l2s = 0.5 / (n * 16384), 0.25 / (n * 16384)
    dmixed = torch.empty((2 * n, 128, 128, 3), device="cuda")
    _lib.call("cgs_bf16_enc0_bwd_data_pooled", 2 * n, P(dp), None, P(codes), P(w), P(dmixed), S())
    want = torch.full((n, 128, 128), 7.0, device="cuda")
    _lib.call("cgs_mix_bwd", n, 16384, P(A), P(B), P(Z), P(dmixed), 1, l1s, l2s, P(want), S())
    got = torch.full((n, 128, 128), 8.0, device="cuda")
    _lib.call("cgs_bf16_enc0_bwd_mix", n, P(dp), P(codes), P(w), P(A), P(B), P(Z), l1s, l2s, P(got), S())
    torch.cuda.synchronize()
    err = (got - want).abs().max().item() / want.abs().max().item()
    print(f"fused mix backward vs two launches: max |d| / max |ref| = {err:.2e}")
    assert err <= 2.0 ** -7


def test_virtual_mixes_match_the_materialised_ones():
    """cgs_bf16_enc0_fwd_mix / cgs_bf16_hwgrad_pooled_mix form the replaced / injected mixes from (A, B, Z) while they stage a tile, with
    cgs_mix_fwd's formula A (1 - Z) + Z B: the same results as cgs_mix_fwd followed by the kernels on the fp32 mixes up to the compiler's
    choice of which product of that expression is fused into the add (1 fp32 ulp of a mix value before its bf16 rounding): outputs within one
    bf16 ulp on a handful of elements, argmax bytes equal on > 99.9 %, weight-gradient sums within 1e-5; cgs_mix_fwd(mixed = NULL) leaves
    the same partial sums bit for bit."""
    from cgs_amd import _lib
    lib = _lib.load()
    n = 5
    g = torch.Generator().manual_seed(17)
    A = torch.randint(0, 256, (n, 128, 128, 3), dtype=torch.uint8, generator=g).cuda()
    B = torch.randint(0, 256, (n, 128, 128, 3), dtype=torch.uint8, generator=g).cuda()
    Z = torch.rand((n, 128, 128), generator=g).cuda()
    w = (torch.randn((9, 3, 8), generator=g) * 0.3).cuda()
    b = (torch.randn(8, generator=g) * 0.1).cuda()
    mixed = torch.empty((2 * n, 128, 128, 3), device="cuda")
    nz = lib.cgs_mix_fwd_partials(n, 16384)
    z1, z2 = torch.zeros(2 * nz, device="cuda"), torch.ones(2 * nz, device="cuda")
    _lib.call("cgs_mix_fwd", n, 16384, P(A), P(B), P(Z), 1, P(mixed), P(z1), S())
    _lib.call("cgs_mix_fwd", n, 16384, None, None, P(Z), 1, None, P(z2), S())
    e1 = torch.empty((2 * n, 64, 64, 8), device="cuda", dtype=torch.bfloat16)
    e2 = torch.full((2 * n, 64, 64, 8), 3.0, device="cuda", dtype=torch.bfloat16)
    c1 = torch.empty((2 * n, 64, 64, 8), device="cuda", dtype=torch.uint8)
    c2 = torch.full((2 * n, 64, 64, 8), 9, device="cuda", dtype=torch.uint8)
    _lib.call("cgs_bf16_enc0_fwd", 2 * n, P(mixed), 1, P(w), P(b), P(e1), P(c1), S())
    _lib.call("cgs_bf16_enc0_fwd_mix", n, P(A), P(B), P(Z), P(w), P(b), P(e2), P(c2), S())
    dp = (torch.randn((2 * n, 64, 64, 8), generator=g) * 0.1).to(torch.bfloat16).cuda()
    nsl = lib.cgs_bf16_hwgrad_slabs(2 * n, 128, 3, 0, 8)
    s1 = torch.full((nsl, 224), float("nan"), device="cuda")
    s2 = torch.full((nsl, 224), float("nan"), device="cuda")
    _lib.call("cgs_bf16_hwgrad_pooled", 2 * n, 128, 3, 2, P(mixed), P(dp), None, P(c1), P(s1), S())
    _lib.call("cgs_bf16_hwgrad_pooled_mix", n, P(A), P(B), P(Z), P(dp), P(c1), P(s2), S())
    torch.cuda.synchronize()
    assert torch.equal(z1, z2)
    d = (e1.float() - e2.float()).abs()
    assert (d > 0).double().mean().item() < 1e-2 and d.max().item() <= 2.0 ** -7 * e1.float().abs().max().item()
    assert (c1 == c2).double().mean().item() > 0.999
    assert torch.isfinite(s1).all() and torch.isfinite(s2).all()
    g1, g2 = s1.double().sum(0), s2.double().sum(0)
    assert (g1 - g2).abs().max().item() <= 1e-5 * g1.abs().max().item()
