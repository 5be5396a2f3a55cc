"""GPU parity of the lane = pixel shape-generic 3x3 convolution (csrc/gen4.hip) on its own: every tile geometry (hw 4 .. 64,
multi-image tiles with a ragged last tile), every source kind (fp32 / odd-width fp32 / uint8, with and without a nearest-upsampled
second source, pooled gradient + argmax bytes), channel counts that are not multiples of 4 or 16 on either side, several output
passes, the max-pool epilogue with its argmax / no-gradient byte, the addend, and the data-gradient operand -- against torch's
conv2d on the CPU in float64 (the reference's own numeric backend, nets.py:170-183 / 480-489 layer forms)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from test_gpu_kernels import rel_close


def _ref_conv(a, b, ups, w_hwio, bias, act, slope, pool):
    xa = (a.double() / 255.0 if a.dtype == torch.uint8 else a.double()).permute(0, 3, 1, 2)
    xin = xa if b is None else torch.cat([xa, F.interpolate(b.double().permute(0, 3, 1, 2), scale_factor=ups, mode="nearest")], 1)
    ci, co = w_hwio.shape[1], w_hwio.shape[2]
    wt = w_hwio.double().view(3, 3, ci, co).permute(3, 2, 0, 1).contiguous()
    y = F.conv2d(xin, wt, bias.double(), padding=1)
    if act == "relu":
        y = F.relu(y)
    elif act == "lrelu":
        y = F.leaky_relu(y, slope)
    elif act == "sigmoid":
        y = torch.sigmoid(y)
    idx = None
    if pool:
        y, idx = F.max_pool2d(y, 2, return_indices=True)
    return y.permute(0, 2, 3, 1).float(), idx


@pytest.mark.parametrize("n,hw,ca,cb,ups,co,act,pool,u8", [
    (3, 64, 3, 0, 1, 40, "relu", True, True),        # features.0 at chfak 5: uint8 frames, 27 k-steps, 10 output groups
    (3, 64, 3, 0, 1, 8, "relu", True, False),        # odd-width fp32 source (the mixes)
    (5, 32, 40, 0, 1, 40, "relu", True, False),      # three chunks (16, 16, 8)
    (5, 16, 24, 0, 1, 24, "relu", True, False),      # chfak 3: 6 groups
    (6, 8, 40, 0, 1, 80, "relu", True, False),       # four images per tile (ragged: 6), two output passes
    (19, 4, 16, 32, 4, 16, "none", False, False),    # sixteen images per tile (ragged: 19), x4 upsampled second source
    (6, 8, 8, 16, 2, 8, "none", False, False),
    (3, 16, 40, 40, 2, 40, "none", False, False),
    (2, 32, 16, 16, 2, 16, "lrelu", False, False),
    (2, 64, 3, 40, 2, 16, "lrelu", False, True),     # masker.0: uint8 image + upsampled decoder channels
    (2, 64, 3, 8, 2, 16, "lrelu", False, False),     # ... with an fp32 image
    (2, 64, 16, 0, 1, 1, "sigmoid", False, False),   # one output channel
    (2, 32, 8, 0, 1, 6, "relu", True, False),        # pooled, channel count not a multiple of 4 (scalar epilogue)
    (2, 32, 12, 0, 1, 43, "none", False, False),     # 11 groups: two passes, scalar copy-out
    (2, 16, 160, 0, 1, 160, "relu", True, False),    # ten chunks, four passes
    # the four-workgroups-per-CU instances (8 / 10 groups, fp32 source of whole quads) at the map sizes chfak 5 does not send them
    (2, 64, 8, 0, 1, 32, "relu", False, False),      # 64 x 64, eight groups, ONE half-empty chunk (the peeled loop runs its last chunk only)
    (2, 64, 40, 0, 1, 40, "relu", True, False),      # 64 x 64, ten groups, pooled (epilogue pitch 40)
    (3, 16, 16, 0, 1, 32, "none", False, False),     # one full chunk
    (21, 4, 32, 0, 1, 40, "relu", False, False),     # sixteen images per tile (ragged: 21), two full chunks
    (5, 8, 36, 16, 2, 72, "none", False, False),     # second source through the vector loader, two passes of nine groups -> 10-group instance
])
def test_gen4_forward_matches_float64_conv2d(n, hw, ca, cb, ups, co, act, pool, u8):
    from cgs_amd import generic as gen
    rs = np.random.RandomState(hw * 1000 + ca * 10 + co)
    dev = torch.device("cuda:0")
    a = torch.from_numpy(rs.randint(0, 256, (n, hw, hw, ca)).astype(np.uint8)) if u8 else torch.from_numpy(rs.randn(n, hw, hw, ca).astype(np.float32))
    b = torch.from_numpy(rs.randn(n, hw // ups, hw // ups, cb).astype(np.float32)) if cb else None
    w = torch.from_numpy((rs.randn(9, ca + cb, co) / (3.0 * np.sqrt(ca + cb))).astype(np.float32))
    bias = torch.from_numpy((0.1 * rs.randn(co)).astype(np.float32))
    ref, idx = _ref_conv(a, b, ups, w, bias, act, 0.2, pool)
    wd, bd = w.to(dev), bias.to(dev)
    out = gen.conv3x3(a.to(dev), None if b is None else b.to(dev), wd.data_ptr(), bd.data_ptr(), co, act=act, slope=0.2, pool=pool, ups=ups,
                      want_argmax=pool)
    if pool:
        out, am = out
        am = am.cpu().numpy()
        # argmax position inside the 2x2 window (row-major), bit 2 where the pooled ReLU value is not positive
        oy, ox = np.divmod(idx.permute(0, 2, 3, 1).numpy(), hw)
        pos = (oy % 2) * 2 + (ox % 2)
        refv = ref.numpy()
        live = refv > 0
        # exact agreement wherever the window's maximum is unique in float32 (random data: everywhere but the ReLU-dead windows)
        assert ((am & 3) == pos)[live].mean() > 0.999
        assert ((am & 4) != 0)[~live].all() and ((am & 4) == 0)[live].all()
    rel_close(out.cpu().numpy(), ref.numpy(), "forward")


@pytest.mark.parametrize("n,hw,ca,cb,co,act,u8", [
    (3, 32, 40, 40, 40, "lrelu", False),    # dec_model.0 at chfak 5: A = 16 + 16 + 8 channels, B = 16 + 16 + 8, ten groups
    (2, 64, 3, 40, 16, "lrelu", True),      # masker.0: uint8 frames + 40 decoder channels
    (2, 64, 3, 24, 16, "none", False),      # fp32 frames (odd width), B = 16 + 8
    (5, 16, 80, 80, 40, "none", False),     # one image per tile, low-resolution tile 8 x 8
    (2, 16, 8, 4, 43, "relu", False),       # one-plane B chunk; 11 groups: two passes, scalar copy-out
    (1, 32, 16, 16, 24, "sigmoid", False),  # six groups
])
def test_gen4_folded_upsample_forward(n, hw, ca, cb, co, act, u8):
    """A layer over cat(A, nearest-up_2(B)) with B staged at its own resolution and its nine taps folded to four per pixel parity
    (cgs_gen_conv3x3_fwd_folded; nets.py:480-489 decoder layers) == float64 conv2d over the explicit upsampled cat, and == the unfolded
    kernel up to the rounding of the folded weight sums."""
    from cgs_amd import generic as gen
    rs = np.random.RandomState(hw * 100 + ca + cb + co)
    dev = torch.device("cuda:0")
    a = torch.from_numpy(rs.randint(0, 256, (n, hw, hw, ca)).astype(np.uint8)) if u8 else torch.from_numpy(rs.randn(n, hw, hw, ca).astype(np.float32))
    b = torch.from_numpy(rs.randn(n, hw // 2, hw // 2, cb).astype(np.float32))
    w = torch.from_numpy((rs.randn(9, ca + cb, co) / (3.0 * np.sqrt(ca + cb))).astype(np.float32))
    bias = torch.from_numpy((0.1 * rs.randn(co)).astype(np.float32))
    ref, _ = _ref_conv(a, b, 2, w, bias, act, 0.2, False)
    wd, bd = w.to(dev), bias.to(dev)
    outs = {}
    saved = gen.UPS_FOLD
    try:
        for fold in (True, False):
            gen.UPS_FOLD = fold
            outs[fold] = gen.conv3x3(a.to(dev), b.to(dev), wd.data_ptr(), bd.data_ptr(), co, act=act, slope=0.2, ups=2).cpu().numpy()
    finally:
        gen.UPS_FOLD = saved
    rel_close(outs[True], ref.numpy(), "folded forward")
    rel_close(outs[False], ref.numpy(), "forward")
    assert np.abs(outs[True] - outs[False]).max() <= 2e-5 * max(1.0, np.abs(ref.numpy()).max())


@pytest.mark.parametrize("n,hw,ci,co,pooled", [
    (3, 64, 3, 40, True),       # the image layer's data gradient: 3 output channels (one group, scalar copy-out), pooled gradient source
    (5, 32, 40, 40, True),
    (6, 8, 40, 80, True),
    (19, 4, 48, 16, False),
    (2, 64, 43, 16, False),     # masker.0: d cat(image, decoder channels)
    (3, 16, 80, 40, False),
    (2, 64, 40, 8, False),      # ten output groups at 64 x 64 from an 8-channel dY (four-per-CU instance, one half-empty chunk)
    (2, 64, 32, 40, True),      # eight groups, pooled gradient source at 64 x 64
    (21, 4, 32, 48, False),     # sixteen images per tile, three chunks
])
def test_gen4_data_gradient_matches_autograd(n, hw, ci, co, pooled):
    """cgs_gen_conv_pack_weights(transposed = 1) + cgs_gen_conv3x3_bwd_data vs float64 autograd of conv2d (+ ReLU + max-pool), with the
    addend on the leading images."""
    from cgs_amd import generic as gen
    rs = np.random.RandomState(hw + ci + co)
    dev = torch.device("cuda:0")
    x = torch.from_numpy(rs.randn(n, ci, hw, hw)).double().requires_grad_(True)
    w_hwio = torch.from_numpy((rs.randn(9, ci, co) / (3.0 * np.sqrt(ci))).astype(np.float32))
    wt = w_hwio.double().view(3, 3, ci, co).permute(3, 2, 0, 1).contiguous()
    y = F.conv2d(x, wt, None, padding=1)
    if pooled:
        yp, idx = F.max_pool2d(F.relu(y), 2, return_indices=True)
        dout = torch.from_numpy(rs.randn(*yp.shape))
        yp.backward(dout)
        oy, ox = np.divmod(idx.permute(0, 2, 3, 1).numpy(), hw)
        am = ((oy % 2) * 2 + (ox % 2)).astype(np.uint8)
        am[(yp.detach().permute(0, 2, 3, 1).numpy() <= 0)] |= 4
        am_d = torch.from_numpy(np.ascontiguousarray(am)).to(dev)
    else:
        dout = torch.from_numpy(rs.randn(*y.shape))
        y.backward(dout)
        am_d = None
    dy = dout.float().permute(0, 2, 3, 1).contiguous().to(dev)
    add = torch.from_numpy(rs.randn(2, hw, hw, ci).astype(np.float32)).to(dev)
    out = torch.full((n, hw, hw, ci), float("nan"), device=dev)
    wd = w_hwio.to(dev)
    gen._bwd_data(n, hw, co, ci, dy, am_d, wd.data_ptr(), out, addend=add)
    ref = x.grad.permute(0, 2, 3, 1).float().numpy().copy()
    ref[:2] += add.cpu().numpy()
    rel_close(out.cpu().numpy(), ref, "data gradient + addend")


@pytest.mark.parametrize("n,hw,ca,cb,co", [
    (2, 64, 3, 40, 16),        # masker.0 at chfak 5: one 16-channel chunk per parity block, ten output groups
    (3, 32, 40, 40, 40),       # dec_model.0: parity blocks of 40 channels padded to 48 (three chunks, the last half empty)
    (5, 16, 40, 40, 40),       # dec_model.1: low-resolution map 8 x 8, four images per tile (ragged: 5)
    (2, 32, 16, 16, 16),       # chfak 2
    (1, 64, 3, 24, 8),         # eight dY channels: a half-empty chunk per block; six output groups
    (3, 32, 8, 44, 12),        # eleven output groups: two passes
    (2, 16, 16, 32, 16),       # eight output groups (four-per-CU instance of the s2d form), 8 x 8 cells, four images per tile (ragged: 2)
    (3, 64, 4, 32, 20),        # 32 x 32 cells, parity blocks of 20 channels padded to 32
])
def test_gen4_up2_data_gradient_vs_float64_autograd(n, hw, ca, cb, co):
    """cgs_gen_conv3x3_bwd_data_up2: the gradient of conv3x3(cat(A, nearest-up_2(B))) w.r.t. B (nets.py:501-517 under autograd), from the
    space-to-depth view of dY with 16 (parity block, cell offset) steps per cell, == float64 autograd through F.interpolate + conv2d, and ==
    the 2 x 2 cell sums of the full-resolution data gradient (cgs_gen_conv3x3_bwd_data_split) up to rounding."""
    from cgs_amd import _lib, generic as gen
    rs = np.random.RandomState(hw + 7 * ca + 13 * cb + co)
    dev = torch.device("cuda:0")
    w = torch.from_numpy((rs.randn(9, ca + cb, co) / (3.0 * np.sqrt(ca + cb))).astype(np.float32))
    dy = torch.from_numpy(rs.randn(n, hw, hw, co).astype(np.float32))
    b = torch.zeros(n, cb, hw // 2, hw // 2, dtype=torch.float64, requires_grad=True)
    a = torch.zeros(n, ca, hw, hw, dtype=torch.float64)
    wt = w.double().view(3, 3, ca + cb, co).permute(3, 2, 0, 1).contiguous()
    y = F.conv2d(torch.cat([a, F.interpolate(b, scale_factor=2, mode="nearest")], 1), wt, None, padding=1)
    y.backward(dy.double().permute(0, 3, 1, 2))
    ref = b.grad.permute(0, 2, 3, 1).float().numpy()
    wd, dyd = w.to(dev), dy.to(dev)
    out = torch.full((n, hw // 2, hw // 2, cb), float("nan"), device=dev)
    wp = gen.pack_weights_up2(wd.data_ptr(), co, ca + cb, ca, cb, dev)
    _lib.call("cgs_gen_conv3x3_bwd_data_up2", n, hw, co, cb, gen._p(dyd), gen._p(wp), gen._p(out), gen._s())
    got = out.cpu().numpy()
    assert np.isfinite(got).all()
    rel_close(got, ref, "d B at its own resolution")


def test_gen4_rejects_bad_arguments():
    from cgs_amd import _lib
    lib = _lib.load()
    assert lib.cgs_gen_conv_packed_floats(40, 0, 40) == 3 * 9 * 10 * 64
    assert lib.cgs_gen_conv_packed_floats(3, 40, 16) == 3 * 9 * 4 * 64
    assert lib.cgs_gen_conv_packed_floats(40, 6, 40) < 0          # second source: whole groups of 4 channels
    assert lib.cgs_gen_conv_pack_weights(8, 4, 8, 1, C.c_void_p(16), C.c_void_p(16), None) < 0      # transposed: one source only


@pytest.mark.parametrize("n,hw,co,ca,cb,ups", [
    (3, 32, 40, 40, 40, 2),      # dec_model.0 at chfak 5: two passes of 10 groups, one per side
    (5, 16, 24, 24, 24, 2),      # chfak 3
    (19, 4, 80, 80, 160, 4),     # dec_model.3: x4 cells = whole 4x4 images, ragged last tile
    (6, 8, 40, 40, 80, 2),
])
def test_gen4_split_data_gradient_equals_two_step_form(n, hw, co, ca, cb, ups):
    """cgs_gen_conv3x3_bwd_data_split (d_a and the cell-summed d_b written by the convolution's epilogue) against
    cgs_gen_conv3x3_bwd_data + cgs_gen_cat_split: d_a bit-identical, d_b up to the order of the cell sum."""
    from cgs_amd import _lib, generic as gen
    rs = np.random.RandomState(n + hw + co)
    dev = torch.device("cuda:0")
    dy = torch.from_numpy(rs.randn(n, hw, hw, co).astype(np.float32)).to(dev)
    w = torch.from_numpy((rs.randn(9, ca + cb, co) / (3.0 * np.sqrt(ca + cb))).astype(np.float32)).to(dev)
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    dcat = torch.empty(n, hw, hw, ca + cb, device=dev)
    gen._bwd_data(n, hw, co, ca + cb, dy, None, w.data_ptr(), dcat)
    da_ref = torch.empty(n, hw, hw, ca, device=dev)
    db_ref = torch.empty(n, hw // ups, hw // ups, cb, device=dev)
    _lib.call("cgs_gen_cat_split", n, hw, ca, cb, ups, P(dcat), P(da_ref), P(db_ref), gen._s())
    wp = gen.pack_weights(w.data_ptr(), co, 0, ca + cb, dev, transposed=True)
    d_a = torch.full_like(da_ref, float("nan"))
    d_b = torch.full_like(db_ref, float("nan"))
    _lib.call("cgs_gen_conv3x3_bwd_data_split", n, hw, co, ca, cb, ups, P(dy), P(wp), P(d_a), P(d_b), gen._s())
    assert torch.equal(d_a, da_ref)
    rel_close(d_b.cpu().numpy(), db_ref.cpu().numpy(), "cell-summed low-resolution gradient")
    # the skip gradient is optional
    d_b2 = torch.full_like(db_ref, float("nan"))
    _lib.call("cgs_gen_conv3x3_bwd_data_split", n, hw, co, ca, cb, ups, P(dy), P(wp), None, P(d_b2), gen._s())
    assert torch.equal(d_b2, d_b)


def test_gen4_split_refuses_passes_that_straddle_the_split():
    from cgs_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    t = torch.zeros(1 << 16, device=dev)
    P = lambda x: C.c_void_p(x.data_ptr())
    # 16 + 16 channels = 8 groups = ONE pass of 8: both sides in one workgroup
    assert lib.cgs_gen_conv3x3_bwd_data_split(1, 8, 16, 16, 16, 2, P(t), P(t), P(t), P(t), None) == _lib.ERR_UNSUPPORTED
    assert lib.cgs_gen_conv3x3_bwd_data_split(1, 8, 16, 6, 16, 2, P(t), P(t), P(t), P(t), None) == _lib.ERR_BADARG


def test_gen4_window_operand_gives_the_second_source_gradient_alone():
    """masker.0's backward: only the decoder channels of cat(image, up2(o0)) need a gradient -- the operand of input channels [3, 3 + c)
    (cgs_gen_conv_pack_weights_window) with the all-cell-sum epilogue against the full d_cat + cgs_gen_cat_split."""
    from cgs_amd import _lib, generic as gen
    rs = np.random.RandomState(5)
    dev = torch.device("cuda:0")
    n, hw, mc, c = 3, 64, 16, 40
    dy = torch.from_numpy(rs.randn(n, hw, hw, mc).astype(np.float32)).to(dev)
    w = torch.from_numpy((rs.randn(9, 3 + c, mc) / 20.0).astype(np.float32)).to(dev)
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    dcat = torch.empty(n, hw, hw, 3 + c, device=dev)
    gen._bwd_data(n, hw, mc, 3 + c, dy, None, w.data_ptr(), dcat)
    db_ref = torch.empty(n, hw // 2, hw // 2, c, device=dev)
    _lib.call("cgs_gen_cat_split", n, hw, 3, c, 2, P(dcat), None, P(db_ref), gen._s())
    wp = torch.empty(int(_lib.load().cgs_gen_conv_packed_floats(mc, 0, c)), device=dev)
    _lib.call("cgs_gen_conv_pack_weights_window", mc, 3 + c, 3, c, P(w), P(wp), gen._s())
    d_b = torch.full_like(db_ref, float("nan"))
    _lib.call("cgs_gen_conv3x3_bwd_data_split", n, hw, mc, 0, c, 2, P(dy), P(wp), None, P(d_b), gen._s())
    rel_close(d_b.cpu().numpy(), db_ref.cpu().numpy(), "d o0")


@pytest.mark.parametrize("n,co,u8", [(3, 40, True), (5, 40, False), (300, 16, True), (2, 24, False), (2, 32, True), (160, 40, True)])
def test_gen_enc0_dedicated_forward_vs_float64(n, co, u8):
    """cgs_gen_enc0_fwd (features.0 of chfak 2 .. 5 on its own kernel: lane = pool cell, all weights in registers) against float64 conv2d +
    ReLU + MaxPool2d: pooled output within the suite's forward bound, argmax bytes = max_pool2d's indices (first maximum wins; bit 2 where the
    pooled value is <= 0); n = 300 / 160: more strips than persistent workgroups."""
    from cgs_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(n * 100 + co)
    a = torch.randint(0, 256, (n, 64, 64, 3), dtype=torch.uint8, generator=g) if u8 else torch.rand((n, 64, 64, 3), generator=g)
    if u8:
        a[0, :8, :8] = 90                   # a flat patch: exact ties
    w = torch.randn((9, 3, co), generator=g) * 0.3
    b = torch.randn(co, generator=g) * 0.2
    b[0] = 0.7
    ref, idx = _ref_conv(a, None, 1, w, b, "relu", 0.0, True)
    a_g, w_g, b_g = a.cuda(), w.cuda(), b.cuda()
    out = torch.full((n, 32, 32, co), 7.0, device="cuda")
    am = torch.full((n, 32, 32, co), 99, device="cuda", dtype=torch.uint8)
    rc = lib.cgs_gen_enc0_fwd(n, co, int(u8), C.c_void_p(a_g.data_ptr()), C.c_void_p(w_g.data_ptr()), C.c_void_p(b_g.data_ptr()),
                              C.c_void_p(out.data_ptr()), C.c_void_p(am.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    rel_close(out.cpu(), ref, f"gen_enc0 forward n={n} co={co}", rtol=1e-4, atol_scale=2e-5)
    pos = (((idx // 64) % 2) * 2 + (idx % 64) % 2).permute(0, 2, 3, 1)
    want = torch.where(ref > 0, pos, torch.full_like(pos, 4))
    got = am.cpu().long()
    live = ref > 1e-4                       # (a pooled value within rounding of zero may carry either flag)
    assert ((got & 4) == (want & 4))[live | (ref == 0)].all()
    agree = ((got & 3) == (want & 3))[live].double().mean().item()
    assert agree > 0.999, agree
    if u8:      # the flat patch: the four positions of a cell run the same FMA chain on the same values -> exact ties -> position 0
        assert ((got & 3)[0, 1:3, 1:3][live[0, 1:3, 1:3]] == 0).all()
    assert lib.cgs_gen_enc0_fwd(1, 8, 1, C.c_void_p(a_g.data_ptr()), C.c_void_p(w_g.data_ptr()), C.c_void_p(b_g.data_ptr()),
                                C.c_void_p(out.data_ptr()), None, None) == _lib.ERR_UNSUPPORTED


@pytest.mark.parametrize("n,co", [(3, 40), (2, 16), (2, 24), (2, 32), (150, 40)])
def test_gen_enc0_dedicated_image_gradient_vs_float64(n, co):
    """cgs_gen_enc0_bwd_data (features.0's data gradient at chfak 2 .. 5 on its own kernel: lane = 2x2 cell of d x, the 4x4 patch of dY rebuilt in
    registers from the pooled gradient + argmax bytes) against float64 conv_transpose2d of the re-expanded gradient; n = 150: more strips than
    persistent workgroups."""
    from cgs_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(5 * co + n)
    dE = torch.randn(n, 32, 32, co, generator=g)
    am = torch.randint(0, 5, (n, 32, 32, co), dtype=torch.uint8, generator=g)
    am = torch.where(torch.rand(am.shape, generator=g) < 0.1, am | 4, am)
    w = torch.randn(9, 3, co, generator=g) * 0.3
    full = torch.zeros(n, 32, 2, 32, 2, co, dtype=torch.float64)
    for py in range(2):
        for px in range(2):
            full[:, :, py, :, px, :] = torch.where(am == 2 * py + px, dE.double(), torch.zeros((), dtype=torch.float64))
    wt = w.double().view(3, 3, 3, co).permute(3, 2, 0, 1)              # OIHW of the forward layer (O = co, I = 3)
    ref = F.conv_transpose2d(full.view(n, 64, 64, co).permute(0, 3, 1, 2), wt, padding=1).permute(0, 2, 3, 1).float()
    de_g, am_g, w_g = dE.cuda(), am.cuda(), w.cuda()
    dx = torch.full((n, 64, 64, 3), 7.0, device="cuda")
    rc = lib.cgs_gen_enc0_bwd_data(n, co, C.c_void_p(de_g.data_ptr()), C.c_void_p(am_g.data_ptr()), C.c_void_p(w_g.data_ptr()),
                                   C.c_void_p(dx.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    rel_close(dx.cpu(), ref, f"gen_enc0 image gradient n={n} co={co}", rtol=1e-4, atol_scale=2e-5)
