"""GPU parity of the shape-generic TRAINING pass (csrc/gen_train.hip, cgs_amd/generic_engine.py): model sizes outside the specialised
kernel set (chfak != 1; the paper's model is chfak = 5) -- the single kernels against torch's CPU autograd, the engine against the
reference capture at chfak = 2 (tests/golden/g3_train_chfak2.npz) and against the CPU oracle (dropout with the kernels' own masks,
frozen critic, -separate, chfak = 5)."""
import ctypes as C

import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import hourglass_ref as orc
from test_gpu_kernels import rel_close, nhwc
from test_gpu_engine import split


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _hwio(w):   # OIHW -> flat HWIO
    return w.permute(2, 3, 1, 0).contiguous().reshape(-1)


@pytest.mark.parametrize("hw,ca,cb,co,ups,pool,u8", [
    (8, 24, 0, 40, 2, True, False),       # pooled ReLU layer, channel counts that are not multiples of 16
    (16, 8, 16, 24, 2, False, False),     # decoder layer: cat(skip, nearest-up(low))
    (4, 16, 32, 16, 4, False, False),     # dec_model.3: the low source is the 1x1 bottleneck
    (64, 3, 8, 16, 2, False, True),       # masker.0: uint8 frames + upsampled decoder output
    (64, 16, 0, 1, 2, False, False),      # masker.2: one output channel
    (32, 3, 0, 8, 2, True, False),        # image layer (3 channels, fp32)
])
def test_generic_conv_backward_kernels_vs_autograd(hw, ca, cb, co, ups, pool, u8):
    """cgs_gen_conv3x3_bwd_data (with cgs_gen_flip_weights / cgs_gen_cat_split) and cgs_gen_conv3x3_bwd_weight (+ cgs_reduce_slabs)
    against float64 autograd of conv2d(cat(a, up(b))) [-> ReLU -> MaxPool2d(2)]."""
    from cgs_amd import _lib, generic as gen, hourglass as hg
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(hw * 131 + ca * 7 + co)
    n = 5
    if u8:
        a_np = rs.randint(0, 256, (n, hw, hw, ca)).astype(np.uint8)
        a_ref = torch.from_numpy(a_np).double().div(255.0).permute(0, 3, 1, 2).requires_grad_(True)
    else:
        a_np = rs.randn(n, hw, hw, ca).astype(np.float32)
        a_ref = torch.from_numpy(a_np).double().permute(0, 3, 1, 2).requires_grad_(True)
    b_np = rs.randn(n, hw // ups, hw // ups, cb).astype(np.float32) if cb else None
    w = torch.from_numpy((rs.randn(co, ca + cb, 3, 3) * 0.2).astype(np.float32))
    bias = torch.from_numpy(rs.randn(co).astype(np.float32) * 0.1)
    # reference
    wr, br = w.double().requires_grad_(True), bias.double().requires_grad_(True)
    src = a_ref
    b_ref = None
    if cb:
        b_ref = torch.from_numpy(b_np).double().permute(0, 3, 1, 2).requires_grad_(True)
        src = torch.cat((a_ref, F.interpolate(b_ref, scale_factor=ups, mode="nearest")), 1)
    y = F.conv2d(src, wr, br, padding=1)
    out_ref = F.max_pool2d(F.relu(y), 2) if pool else y
    dout = torch.from_numpy(rs.randn(*out_ref.shape).astype(np.float32))
    out_ref.backward(dout.double())
    # HIP: forward (for the argmax bytes), flipped weights, data gradient, split, weight gradient
    a_d = torch.from_numpy(a_np).to(dev)
    b_d = torch.from_numpy(b_np).to(dev) if cb else None
    wk, bk = _hwio(w).to(dev), bias.to(dev)
    am = None
    if pool:
        o, am = gen.conv3x3(a_d, b_d, wk.data_ptr(), bk.data_ptr(), co, act="relu", pool=True, ups=ups, want_argmax=True)
        rel_close(nhwc(o), out_ref.detach().numpy(), "forward")
    dy = dout.permute(0, 2, 3, 1).contiguous().to(dev)
    ci = ca + cb
    # the data gradient's operand two ways: the layer's weights packed transposed == the flipped weights packed as a forward kernel
    nfl = int(_lib.load().cgs_gen_conv_packed_floats(co, 0, ci))
    wf = torch.empty(9 * ci * co, device=dev)
    _lib.call("cgs_gen_flip_weights", ci, co, _P(wk), _P(wf), _st())
    wp, wp2 = torch.empty(nfl, device=dev), torch.empty(nfl, device=dev)
    _lib.call("cgs_gen_conv_pack_weights", co, 0, ci, 1, _P(wk), _P(wp), _st())
    _lib.call("cgs_gen_conv_pack_weights", co, 0, ci, 0, _P(wf), _P(wp2), _st())
    assert torch.equal(wp, wp2)
    dcat = torch.empty(n, hw, hw, ci, device=dev)
    _lib.call("cgs_gen_conv3x3_bwd_data", n, hw, co, ci, _P(dy), _P(am), _P(wp), None, 0, _P(dcat), _st())
    d_a = torch.empty(n, hw, hw, ca, device=dev)
    d_b = torch.empty(n, hw // ups, hw // ups, cb, device=dev) if cb else None
    _lib.call("cgs_gen_cat_split", n, hw, ca, cb, ups, _P(dcat), _P(d_a), _P(d_b), _st())
    rel_close(nhwc(d_a), a_ref.grad.numpy(), "d_a")      # (uint8 frames: gradient w.r.t. frame / 255)
    if cb:
        rel_close(nhwc(d_b), b_ref.grad.numpy(), "d_b")
    nsl = _lib.load().cgs_gen_conv3x3_bwd_weight_slabs(n, ca, cb, co)
    cnt = 9 * ci * co + co
    slab = torch.full((nsl, cnt), float("nan"), device=dev)
    _lib.call("cgs_gen_conv3x3_bwd_weight", n, hw, ca, cb, co, int(u8), ups, _P(a_d), _P(b_d), _P(dy), _P(am), _P(slab), _st())
    g = torch.zeros(cnt, device=dev)
    plan = hg.SlabPlan()
    plan.add(slab, nsl, cnt, 0)
    plan.build(g).run()
    dw = g[:9 * ci * co].reshape(3, 3, ci, co).permute(3, 2, 0, 1).cpu().numpy()
    rel_close(dw, wr.grad.numpy(), "dW")
    rel_close(g[9 * ci * co:].cpu().numpy(), br.grad.numpy(), "dbias")
    # the addend of the data gradient (skip gradients arriving at the leading images)
    add = torch.from_numpy(rs.randn(2, hw, hw, ci).astype(np.float32)).to(dev)
    dcat2 = torch.empty_like(dcat)
    _lib.call("cgs_gen_conv3x3_bwd_data", n, hw, co, ci, _P(dy), _P(am), _P(wp), _P(add), 2, _P(dcat2), _st())
    exp = dcat.clone()
    exp[:2] += add
    assert torch.equal(dcat2, exp)


def test_generic_gemm_ex_and_grad_fix():
    from cgs_amd import _lib, generic as gen
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(3)
    m, k, n = 37, 50, 21
    x, w = rs.randn(m, k).astype(np.float32), rs.randn(k, n).astype(np.float32)
    xd, wd = torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev)
    out = torch.empty(m, n, device=dev)
    gen.gemm_ex(m, k, n, xd, k, 1, wd, n, 1, out)
    rel_close(out.cpu().numpy(), x.astype(np.float64) @ w, "x w")
    out2 = torch.empty(m, k, device=dev)        # dY W^T
    gen.gemm_ex(m, n, k, out, n, 1, wd, 1, n, out2)
    rel_close(out2.cpu().numpy(), out.cpu().numpy().astype(np.float64) @ w.T, "dy w^T")
    out3 = torch.empty(k, n, device=dev)        # X^T dY, then accumulated once more
    gen.gemm_ex(k, m, n, xd, 1, k, out, n, 1, out3)
    ref3 = x.T.astype(np.float64) @ out.cpu().numpy()
    rel_close(out3.cpu().numpy(), ref3, "x^T dy")
    gen.gemm_ex(k, m, n, xd, 1, k, out, n, 1, out3, accumulate=True)
    rel_close(out3.cpu().numpy(), 2 * ref3, "accumulate")
    # (d * mask + addend) * act'(saved)
    cnt = 8 * 36
    d = torch.from_numpy(rs.randn(cnt).astype(np.float32)).to(dev)
    saved = torch.from_numpy(rs.randn(cnt).astype(np.float32)).to(dev)
    add = torch.from_numpy(rs.randn(cnt // 2).astype(np.float32)).to(dev)
    step = torch.full((1,), 5, dtype=torch.int64, device=dev)
    drop = _lib.Dropout(0.3, 1, 1234, step.data_ptr(), 16, 0)
    mask = torch.empty(cnt, device=dev)
    _lib.call("cgs_dropout_mask", drop, cnt, _P(mask), _st())
    exp = d * mask
    exp[:cnt // 2] += add
    exp = exp * torch.where(saved > 0, torch.ones_like(saved), torch.full_like(saved, 0.01))
    gen.grad_fix(d, saved=saved, act="lrelu", slope=0.01, addend=add, drop=drop)
    assert torch.allclose(d, exp, rtol=1e-6, atol=1e-7)
    xx = torch.from_numpy(rs.randn(cnt).astype(np.float32)).to(dev)
    yy = torch.empty_like(xx)
    _lib.call("cgs_gen_dropout_fwd", cnt, _P(xx), _P(yy), drop, _st())
    assert torch.equal(yy, xx * mask)


def make_generic_engine(chfak, n, neck=32, seeds=(21, 22), **kw):
    from cgs_amd import generic_engine
    pc, pm = orc.seeded_params(orc.critic_shapes(chfak, neck), seeds[0]), orc.seeded_params(orc.masker_shapes(chfak, neck), seeds[1])
    e = generic_engine.GenericEngine(n, chfak=chfak, neck=neck, **kw)
    e.load_state(pc, pm)
    return e, pc, pm


@pytest.mark.parametrize("use_graph,fixture,kw", [(False, "g3_train_chfak2.npz", {}), (True, "g3_train_chfak2.npz", {}),
                                                  # -staticnorm '' (main.py:415-418) on the generic kernels: regulariser weighted by 1 - pred
                                                  (True, "g3_train_chfak2_valuefak.npz", dict(staticnorm=False, L2=0.1))])
def test_generic_phase2_matches_reference_capture_chfak2(golden, use_graph, fixture, kw):
    g = golden(fixture)
    dev = torch.device("cuda:0")
    e, pc, pm = make_generic_engine(2, 8, dropout=0.0, use_graph=use_graph, **kw)
    A, B, Y = (torch.from_numpy(g[k]).to(dev) for k in ("A", "B", "Y"))
    for s in range(2):
        losses = e.phase2_step(A, B, Y).cpu().numpy().astype(np.float64)
        np.testing.assert_allclose(losses[:5], g[f"parts{s}"], rtol=1e-3, atol=1e-7, err_msg=f"losses step {s}")
        assert losses[5] == pytest.approx(float(g[f"total{s}"]), rel=1e-3)
        if s == 0:
            gc, gm = e.lc.unflatten(e.gc), e.lm.unflatten(e.gm)
            for k, v in split(g, "grad/masker").items():
                rel_close(gm[k].cpu().numpy(), v, f"masker grad {k}")
            for k, v in split(g, "grad/critic").items():
                rel_close(gc[k].cpu().numpy(), v, f"critic grad {k}")
            rel_close(e.mbuf["Z"].cpu().numpy(), g["Z0"][:, 0], "Z")
    for k, v in split(g, "step2/critic").items():
        rel_close(e.critic_state()[k].cpu().numpy(), v, f"critic {k} after step 2", rtol=1e-3, atol_scale=1e-4)
    for k, v in split(g, "step2/masker").items():
        rel_close(e.masker_state()[k].cpu().numpy(), v, f"masker {k} after step 2", rtol=1e-3, atol_scale=1e-4)


def _export_masks(e, n_slots, chfak, neck):
    from cgs_amd import _lib
    d3, d2, nb = 16 * chfak, 8 * chfak, neck * chfak
    out = []
    for site, per_img in ((0, (8, 8, d2)), (1, (4, 4, d3)), (2, (nb,))):
        cnt = n_slots * int(np.prod(per_img))
        buf = torch.empty(cnt, device=e.dev)
        _lib.call("cgs_dropout_mask", e.drop.desc(site), cnt, _P(buf), _st())
        mk = (buf.cpu().reshape((n_slots,) + per_img) != 0).float()
        out.append(mk.permute(0, 3, 1, 2).contiguous() if len(per_img) == 3 else mk)
    return out


@pytest.mark.parametrize("chfak,neck,n,kw", [
    (5, 32, 6, dict()),                         # the paper's model size, Dropout 0.3
    (3, 16, 6, dict(inject=False, L2=0.1)),
    (2, 32, 6, dict(live=False)),
    (2, 32, 6, dict(threshrew=0.5)),
])
def test_generic_phase2_with_dropout_vs_oracle(chfak, neck, n, kw):
    rs = np.random.RandomState(7)
    dev = torch.device("cuda:0")
    A = rs.randint(0, 256, (n, 64, 64, 3)).astype(np.uint8)
    B = rs.randint(0, 256, (n, 64, 64, 3)).astype(np.uint8)
    Y = rs.rand(n).astype(np.float32)
    if kw.get("threshrew"):
        Y = (Y > 0.5).astype(np.float32)
    e, pc, pm = make_generic_engine(chfak, n, neck=neck, dropout=0.3, use_graph=True, **kw)
    masks = _export_masks(e, 4 * n, chfak, neck)
    losses = e.phase2_step(torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev), torch.from_numpy(Y).to(dev)).cpu().numpy()
    sl = {"B": slice(0, n), "A": slice(n, 2 * n), "rep": slice(2 * n, 3 * n), "inj": slice(3 * n, 4 * n)}
    omasks = [[m[sl[k]] for m in masks] for k in ("A", "B", "rep", "inj")]
    okw = {k: v for k, v in kw.items()}
    rec = orc.train_phase2(pc, pm, [(orc.u8_to_nchw(A), orc.u8_to_nchw(B), torch.from_numpy(Y))], steps=1, p=0.3, training=True,
                           masks=omasks, **okw)[0]
    parts = rec["parts"]
    live = kw.get("live", True)
    exp = [parts.get("critic", 0.0) if live else losses[0], parts["replace"], parts.get("inject", 0.0), parts["norm"], parts.get("norm2", 0.0)]
    np.testing.assert_allclose(losses[:5], exp, rtol=1e-3, atol=1e-7)
    gc, gm = e.lc.unflatten(e.gc), e.lm.unflatten(e.gm)
    for k, v in rec["grads_m"].items():
        rel_close(gm[k].cpu().numpy(), v.numpy(), f"masker grad {k}")
    if live:
        for k, v in rec["grads_c"].items():
            rel_close(gc[k].cpu().numpy(), v.numpy(), f"critic grad {k}")
    for k, v in rec["params_c"].items():
        rel_close(e.critic_state()[k].cpu().numpy(), v.numpy(), f"critic {k} after the step", rtol=1e-3, atol_scale=1e-4)
    for k, v in rec["params_m"].items():
        rel_close(e.masker_state()[k].cpu().numpy(), v.numpy(), f"masker {k} after the step", rtol=1e-3, atol_scale=1e-4)
    # graph replay of a second step == an eager engine's second step (same weights, same counter-based masks)
    e2, _, _ = make_generic_engine(chfak, n, neck=neck, dropout=0.3, use_graph=False, **kw)
    for eng in (e2,):
        eng.phase2_step(torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev), torch.from_numpy(Y).to(dev))
    l2a = e.phase2_step().clone()
    l2b = e2.phase2_step().clone()
    assert torch.equal(l2a, l2b)
    assert torch.equal(e.flat, e2.flat)


def _pool_pick_flips(e, pc, n_lo, n_hi, x_in):
    """Pooled cells of the critic's four 3x3 layers (slots n_lo:n_hi of the engine's buffers) where the kernels' argmax byte differs from
    the argmax of a FLOAT64 convolution of the layer's own (fp32, device-computed) input: per layer (count, largest relative gap between the
    two candidates of such a cell).  A cell whose two largest candidates agree to ~1e-7 relative is decided by fp32 rounding; taking the other
    candidate moves dy of that cell to a neighbouring pixel -- a discrete change of the gradients below it that no summation order removes."""
    out = []
    srcs = [x_in, e.cbuf["e0"][n_lo:n_hi], e.cbuf["e1"][n_lo:n_hi], (e.cbuf["e2d"] if "e2d" in e.cbuf else e.cbuf["e2"])[n_lo:n_hi]]
    for li, key in enumerate(("features.0", "features.3", "features.6", "features.10")):
        x = srcs[li].cpu().double().permute(0, 3, 1, 2)
        y = F.conv2d(x, pc[key + ".weight"].double(), pc[key + ".bias"].double(), padding=1)     # (pre-ReLU: max(relu) = relu(max))
        N, Cc, H, W = y.shape
        cells = y.reshape(N, Cc, H // 2, 2, W // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(N, Cc, H // 2, W // 2, 4)
        srt = cells.sort(-1, descending=True).values
        top, idx = cells.max(-1)
        gap = (srt[..., 0] - srt[..., 1]) / srt[..., 0].abs().clamp_min(1e-30)
        am = e.cbuf[f"am{li}"][n_lo:n_hi].cpu().permute(0, 3, 1, 2).long()
        assert not ((top > 1e-4) & (am >= 4)).any() and not ((top < -1e-4) & (am < 4)).any(), \
            f"{key}: ReLU gate of the pooled cells differs from float64"
        mism = (top > 0) & (am < 4) & (am != idx)
        out.append((int(mism.sum()), float(gap[mism].max()) if mism.any() else 0.0))
    return out


def test_generic_phase2_chfak5_batch128_vs_float64_oracle():
    """The paper's model size (chfak 5, main.py:1510 / docs/index.html:151) at a batch its batch-tiled kernels really tile (n = 128: the wgrad-rows /
    split-K / fold kernels take several images per workgroup), Dropout 0.3, ONE phase-2 step on the shape-generic
    engine against the oracle in FLOAT64 fed the keep-masks the kernels drew (as test_gpu_engine.py::test_phase2_with_dropout_vs_oracle_masks
    does for chfak 1 at N = 512: at 0.5 M pixels per weight the CPU's own fp32 summation is no longer a checker): losses, all 28 gradients,
    the parameters after the step.

    Of the 20 M max-pooled cells of features.0 / features.3 in this step a handful (measured: 1 + 3, gpurun_out/r06_diag_c5_flips.txt) have
    their two largest candidates within 1e-7 .. 1e-6 relative: fp32 and float64 pick different pixels there (_pool_pick_flips shows every
    such cell and asserts the gap is rounding-sized).  One moved dy changes 9 x ci weight-gradient elements of that layer (and, through
    the data gradient, the layers below) by |dy| |dx| -- up to 1e-3 of the tensor's maximum here, against 1e-6 everywhere else.  The
    weight gradients at and below a layer with such a cell are therefore checked with an absolute term of 2e-3 of the maximum, all others
    (and every tensor when no cell flips) at the usual 2e-5."""
    chfak, neck, n = 5, 32, 128
    rs = np.random.RandomState(11)
    dev = torch.device("cuda:0")
    A = rs.randint(0, 256, (n, 64, 64, 3)).astype(np.uint8)
    B = rs.randint(0, 256, (n, 64, 64, 3)).astype(np.uint8)
    Y = rs.rand(n).astype(np.float32)
    e, pc, pm = make_generic_engine(chfak, n, neck=neck, dropout=0.3, use_graph=True)
    masks = _export_masks(e, 4 * n, chfak, neck)
    losses = e.phase2_step(torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev), torch.from_numpy(Y).to(dev)).cpu().numpy()
    sl = {"B": slice(0, n), "A": slice(n, 2 * n), "rep": slice(2 * n, 3 * n), "inj": slice(3 * n, 4 * n)}
    omasks = [[m[sl[k]].double() for m in masks] for k in ("A", "B", "rep", "inj")]
    dd = lambda P: {k: v.double() for k, v in P.items()}
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    rec = orc.train_phase2(dd(pc), dd(pm), [(orc.u8_to_nchw(A).double(), orc.u8_to_nchw(B).double(), torch.from_numpy(Y).double())], steps=1,
                           p=0.3, training=True, masks=omasks)[0]
    flips = _pool_pick_flips(e, pc, n, 4 * n, e.x3)       # the images that carry the critic's weight gradients: [A | rep | inj]
    assert sum(c for c, _ in flips) <= 16 and all(g < 2e-6 for _, g in flips), f"pool picks differ on cells that are not near ties: {flips}"
    deepest = max([i for i, (c, _) in enumerate(flips) if c], default=-1)
    loose = {f"{k}.weight" for k in ("features.0", "features.3", "features.6", "features.10")[:deepest + 1]}
    loose |= {f"{k}.bias" for k in ("features.0", "features.3", "features.6", "features.10")[:max(deepest, 0)]}
    parts = rec["parts"]
    np.testing.assert_allclose(losses[:4], [parts["critic"], parts["replace"], parts["inject"], parts["norm"]], rtol=1e-3)
    gc, gm = e.lc.unflatten(e.gc), e.lm.unflatten(e.gm)
    for k, v in rec["grads_c"].items():
        rel_close(gc[k].cpu().numpy(), v.numpy(), f"chfak 5 n=128 critic grad {k}", atol_scale=2e-3 if k in loose else 2e-5, report=k not in loose)
    for k, v in rec["grads_m"].items():
        rel_close(gm[k].cpu().numpy(), v.numpy(), f"chfak 5 n=128 masker grad {k}")
    # Adam's first step is lr * g / (|g| + eps), eps = 1e-8: where |g| is within fp32 noise of zero (< 1e-4 of the tensor's maximum) the step is
    # anything in [-lr, lr] -- those elements are checked for that bound, every other element against the oracle's parameter
    for rec_p, rec_g, state, p0, who in ((rec["params_c"], rec["grads_c"], e.critic_state(), pc, "critic"),
                                         (rec["params_m"], rec["grads_m"], e.masker_state(), pm, "masker")):
        for k, v in rec_p.items():
            if who == "critic" and k in loose:
                continue
            got, g64 = state[k].cpu().numpy(), np.abs(rec_g[k].numpy())
            firm = g64 >= 1e-4 * g64.max()
            assert firm.mean() > 0.5
            rel_close(got[firm], v.numpy()[firm], f"chfak 5 n=128 {who} {k} after the step", rtol=1e-3, atol_scale=1e-4)
            assert (np.abs(got - p0[k].numpy())[~firm] <= 1e-3 * (1 + 1e-3)).all()


def test_generic_phase2_separate_critic_vs_oracle():
    chfak, n = 2, 6
    rs = np.random.RandomState(9)
    dev = torch.device("cuda:0")
    A = rs.randint(0, 256, (n, 64, 64, 3)).astype(np.uint8)
    B = rs.randint(0, 256, (n, 64, 64, 3)).astype(np.uint8)
    Y = rs.rand(n).astype(np.float32)
    from cgs_amd import generic_engine
    pc, pm = orc.seeded_params(orc.critic_shapes(chfak), 21), orc.seeded_params(orc.masker_shapes(chfak), 22)
    ps = orc.seeded_params(orc.critic_shapes(chfak), 23)
    e = generic_engine.GenericEngine(n, chfak=chfak, dropout=0.0, separate=True)
    e.load_state(pc, pm, ps)
    losses = e.phase2_step(torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev), torch.from_numpy(Y).to(dev)).cpu().numpy()
    rec = orc.train_phase2(pc, pm, [(orc.u8_to_nchw(A), orc.u8_to_nchw(B), torch.from_numpy(Y))], steps=1, Ps=ps)[0]
    assert losses[5] == pytest.approx(rec["total"], rel=1e-3)
    gs = e.lc.unflatten(e.gs)
    for k, v in rec["grads_s"].items():
        if v is None:      # the second critic's head sees no loss
            assert float(gs[k].abs().max()) == 0.0
        else:
            rel_close(gs[k].cpu().numpy(), v.numpy(), f"sepcrit grad {k}")
    for k, v in rec["params_s"].items():
        rel_close(e.sepcrit_state()[k].cpu().numpy(), v.numpy(), f"sepcrit {k}", rtol=1e-3, atol_scale=1e-4)
    for k, v in rec["params_c"].items():
        rel_close(e.critic_state()[k].cpu().numpy(), v.numpy(), f"critic {k}", rtol=1e-3, atol_scale=1e-4)


def test_generic_phase1_and_saliency_vs_oracle():
    chfak, n = 5, 6
    rs = np.random.RandomState(11)
    dev = torch.device("cuda:0")
    X = rs.randint(0, 256, (n, 64, 64, 3)).astype(np.uint8)
    Y = rs.rand(n).astype(np.float32)
    e, pc, pm = make_generic_engine(chfak, n, dropout=0.0)
    losses = e.phase1_step(torch.from_numpy(X).to(dev), torch.from_numpy(Y).to(dev)).cpu().numpy()
    rec = orc.train_phase1(pc, [(orc.u8_to_nchw(X), torch.from_numpy(Y))], steps=1, p=0.0)[0]
    assert losses[0] == pytest.approx(rec["loss"], rel=1e-3)
    gc = e.lc.unflatten(e.gc)
    for k, v in rec["grads"].items():
        rel_close(gc[k].cpu().numpy(), v.numpy(), f"grad {k}")
    for k, v in rec["params"].items():
        rel_close(e.critic_state()[k].cpu().numpy(), v.numpy(), f"{k} after the step", rtol=1e-3, atol_scale=1e-4)
    # saliency: d mean(pred) / dX (main.py:941-953) with the UPDATED weights
    Xf = orc.u8_to_nchw(X).requires_grad_(True)
    P = {k: v.cpu() for k, v in e.critic_state().items()}
    orc.critic_apply(P, Xf).mean().backward()
    pred, dx = e.saliency(torch.from_numpy(X).to(dev).float().div(255.0))
    rel_close(nhwc(dx), Xf.grad.numpy(), "saliency gradient")


def test_reference_style_training_loop_through_modules_chfak2(golden):
    """main.py:360-463 against the module API at chfak = 2 (NewCritic / UnetDecoder on the shape-generic kernels, torch ops for the
    mix and the losses, torch.optim.Adam): the reference capture's losses, gradients and parameters."""
    from itertools import chain
    from cgs_amd import nets
    g = golden("g3_train_chfak2.npz")
    critic, masker = nets.NewCritic(chfak=2, dropout=0.0).to("cuda").train(), nets.UnetDecoder(chfak=2).to("cuda").train()
    critic.load_state_dict(orc.seeded_params(orc.critic_shapes(2), 21))
    masker.load_state_dict(orc.seeded_params(orc.masker_shapes(2), 22))
    opti = torch.optim.Adam(chain(critic.parameters(), masker.parameters()))
    A, B = orc.u8_to_nchw(g["A"]).to("cuda"), orc.u8_to_nchw(g["B"]).to("cuda")
    Y = torch.from_numpy(g["Y"]).to("cuda")
    for s in range(2):
        pred, embeds = critic(A, collect=True)
        negpred = critic(B).squeeze().detach()
        pred = pred.squeeze()
        cl = F.mse_loss(pred, Y)
        Z = masker(A, embeds)
        rl = F.mse_loss(critic(A * (1 - Z) + Z * B).squeeze(), negpred)
        il = F.mse_loss(critic(B * (1 - Z) + Z * A).squeeze(), pred.detach())
        nl = 0.5 * F.l1_loss(Z, torch.zeros_like(Z))
        loss = 5 * cl + rl + il + nl
        opti.zero_grad()
        loss.backward()
        if s == 0:
            gm, gc = {k: q.grad for k, q in masker.named_parameters()}, {k: q.grad for k, q in critic.named_parameters()}
            for k, v in split(g, "grad/masker").items():
                rel_close(gm[k].cpu().numpy(), v, f"masker grad {k}")
            for k, v in split(g, "grad/critic").items():
                rel_close(gc[k].cpu().numpy(), v, f"critic grad {k}")
        opti.step()
        np.testing.assert_allclose([cl.item(), rl.item(), il.item(), nl.item()], g[f"parts{s}"][:4], rtol=1e-3, atol=1e-7)
    for k, v in split(g, "step2/masker").items():
        rel_close(masker.state_dict()[k].cpu().numpy(), v, f"masker {k} after 2 steps", atol_scale=1e-4)
    for k, v in split(g, "step2/critic").items():
        rel_close(critic.state_dict()[k].cpu().numpy(), v, f"critic {k} after 2 steps", atol_scale=1e-4)


def test_cli_train_at_chfak2(tmp_path):
    """`main.py -train --chfak 2`: both phases run on the shape-generic engine, the checkpoints carry the reference's names / shapes
    at that size and the critic learns the synthetic bright/dark split."""
    import os, subprocess, sys
    from test_gpu_modules import _write_dataset, REPO
    root = str(tmp_path)
    X, Y = _write_dataset(root)
    common = ["--model", "m", "--datasize", "1536", "--testsize", "512", "--cepochs", "6", "--mepochs", "1", "--chfak", "2"]
    r = subprocess.run([sys.executable, os.path.join(REPO, "main.py"), "-train"] + common, cwd=root, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    saves = os.listdir(os.path.join(root, "m", "saves"))
    cpath = [f for f in saves if f.startswith("critic-") and "chfak=2" in f]
    mpath = [f for f in saves if f.startswith("masker-")]
    assert cpath and mpath, saves
    pc = torch.load(os.path.join(root, "m", "saves", cpath[0]), map_location="cpu")
    pm = torch.load(os.path.join(root, "m", "saves", mpath[0]), map_location="cpu")
    assert [(k, tuple(v.shape)) for k, v in pc.items()] == [(k, s) for k, s in orc.critic_shapes(2)]
    assert [(k, tuple(v.shape)) for k, v in pm.items()] == [(k, s) for k, s in orc.masker_shapes(2)]
    with torch.no_grad():
        p = orc.critic_apply(pc, orc.u8_to_nchw(X[:256])).squeeze(1).numpy()
    assert np.corrcoef(p, Y[1, :256])[0, 1] > 0.9


GEN_DP_WORKER = r"""
import os, sys
sys.path.insert(0, {repo!r})
import numpy as np, torch
import torch.distributed as dist
import cgs_amd
from cgs_amd import parallel, generic_engine
from oracle import hourglass_ref as orc          # (test infrastructure: the seeded stand-in weights)
pg = parallel.init_from_env("gloo")              # 2 ranks share the one GPU of the box
rank, _, world = parallel.env_world()
pc, pm = orc.seeded_params(orc.critic_shapes(2), 21), orc.seeded_params(orc.masker_shapes(2), 22)
rs = np.random.RandomState(3)
n = 16
A = torch.from_numpy(rs.randint(0, 256, (n, 64, 64, 3)).astype(np.uint8)).cuda()
B = torch.from_numpy(rs.randint(0, 256, (n, 64, 64, 3)).astype(np.uint8)).cuda()
Y = torch.from_numpy(rs.rand(n).astype(np.float32)).cuda()
sl = parallel.shard_slice(n, rank, world)
e = generic_engine.GenericEngine(n // world, chfak=2, dropout=0.0, process_group=pg)
e.load_state(pc, pm)
for _ in range(2):
    e.phase2_step(A[sl], B[sl], Y[sl])
torch.cuda.synchronize()
flat = e.flat.cpu()
others = [torch.empty_like(flat) for _ in range(world)]
dist.all_gather(others, flat)
assert all(torch.equal(o, others[0]) for o in others), "replicas diverged"
if rank == 0:
    np.save({out!r}, flat.numpy())
dist.barrier(); dist.destroy_process_group()
"""


def test_generic_engine_data_parallel_world2(tmp_path):
    """GenericEngine (chfak 2) under data parallelism: two ranks (gloo rendezvous, both on this box's GPU), half the batch each,
    2 steps: replicas stay bit-identical and match the single-process full-batch run."""
    import os, subprocess, sys
    from test_gpu_modules import REPO
    out = str(tmp_path / "gen_dp_flat.npy")
    script = tmp_path / "gen_dp_worker.py"
    script.write_text(GEN_DP_WORKER.format(repo=REPO, out=out))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29541", str(script)], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    rs = np.random.RandomState(3)
    n = 16
    dev = torch.device("cuda:0")
    A = torch.from_numpy(rs.randint(0, 256, (n, 64, 64, 3)).astype(np.uint8)).to(dev)
    B = torch.from_numpy(rs.randint(0, 256, (n, 64, 64, 3)).astype(np.uint8)).to(dev)
    Y = torch.from_numpy(rs.rand(n).astype(np.float32)).to(dev)
    e, _, _ = make_generic_engine(2, n, dropout=0.0)
    for _ in range(2):
        e.phase2_step(A, B, Y)
    rel_close(np.load(out), e.flat.cpu().numpy(), "generic DP(2) parameters vs single process", rtol=1e-3, atol_scale=1e-4)
