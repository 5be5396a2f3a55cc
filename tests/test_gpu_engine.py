"""GPU parity of the fused training engine (HIP graph of kernels) against the committed reference captures
(tests/golden/g3_*, g4_*) and against the CPU oracle with the kernels' own dropout masks."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import hourglass_ref as orc
from test_gpu_kernels import rel_close


def split(raw, prefix):
    return {k[len(prefix) + 1:]: v for k, v in raw.items() if k.startswith(prefix + "/")}


def make_engine(g1, n, **kw):
    from cgs_amd import engine
    pc, pm = g1
    e = engine.HourglassEngine(n, **kw)
    e.load_state(pc, pm)
    return e


@pytest.mark.parametrize("tag,kw", [
    ("g3_train_default", dict()),
    ("g3_train_noinject", dict(inject=False)),
    ("g3_train_frozen", dict(live=False)),
    ("g3_train_l2", dict(L2=0.1)),
    ("g3_train_bce", dict(threshrew=0.5)),
    ("g3_train_valuefak", dict(staticnorm=False, L2=0.1)),       # -staticnorm '': regulariser weighted by 1 - pred (main.py:415-418)      # --threshrew: BCE live-critic loss (main.py:380-381)
])
@pytest.mark.parametrize("use_graph", [False, True])
def test_phase2_matches_reference_capture(golden, g1, tag, kw, use_graph):
    g = golden(tag + ".npz")
    dev = torch.device("cuda:0")
    e = make_engine(g1, 8, dropout=0.0, use_graph=use_graph, **kw)
    A, B, Y = (torch.from_numpy(g[k]).to(dev) for k in ("A", "B", "Y"))
    for s in range(3):
        losses = e.phase2_step(A, B, Y).cpu().numpy().astype(np.float64)
        ref = g[f"parts{s}"]  # critic, replace, inject, L1, L2
        got = losses[:5].copy()
        if not kw.get("live", True):
            got[0] = 0.0   # the reference does not evaluate the critic loss when frozen
        np.testing.assert_allclose(got, ref, rtol=1e-3, atol=1e-7, err_msg=f"losses step {s}")
        assert losses[5] == pytest.approx(float(g[f"total{s}"]), rel=1e-3)
        if s == 0:
            gc, gm = e.lc.unflatten(e.gc), e.lm.unflatten(e.gm)
            for k, v in split(g, "grad/masker").items():
                rel_close(gm[k].cpu().numpy(), v, f"masker grad {k}")
            if kw.get("live", True):
                for k, v in split(g, "grad/critic").items():
                    rel_close(gc[k].cpu().numpy(), v, f"critic grad {k}")
            rel_close(e.mbuf["Z"].cpu().numpy(), g["Z0"][:, 0], "Z")
            rel_close(e.cbuf["pred"][8:16].cpu().numpy(), g["pred0"], "pred")
        if s in (0, 2):
            for k, v in split(g, f"step{s + 1}/critic").items():
                rel_close(e.critic_state()[k].cpu().numpy(), v, f"critic {k} after step {s + 1}", rtol=1e-3, atol_scale=1e-4)
            for k, v in split(g, f"step{s + 1}/masker").items():
                rel_close(e.masker_state()[k].cpu().numpy(), v, f"masker {k} after step {s + 1}", rtol=1e-3, atol_scale=1e-4)


@pytest.mark.parametrize("tag,live", [("g3_train_separate", True), ("g3_train_separate_frozen", False)])
def test_phase2_separate_critic_matches_reference_capture(golden, g1, tag, live):
    """-separate (main.py:110-111, 328-334, 389-390): the masker is fed by a second critic that trains along."""
    from cgs_amd import engine
    g = golden(tag + ".npz")
    dev = torch.device("cuda:0")
    pc, pm = g1
    e = engine.HourglassEngine(8, dropout=0.0, live=live, separate=True)
    e.load_state(pc, pm, {k: torch.from_numpy(v) for k, v in split(g, "sepcrit0").items()})
    A, B, Y = (torch.from_numpy(g[k]).to(dev) for k in ("A", "B", "Y"))
    for s in range(3):
        losses = e.phase2_step(A, B, Y).cpu().numpy().astype(np.float64)
        got = losses[:5].copy()
        if not live:
            got[0] = 0.0
        np.testing.assert_allclose(got, g[f"parts{s}"], rtol=1e-3, atol=1e-7, err_msg=f"losses step {s}")
        if s == 0:
            gm, gs = e.lm.unflatten(e.gm), e.lc.unflatten(e.gs)
            for k, v in split(g, "grad/masker").items():
                rel_close(gm[k].cpu().numpy(), v, f"masker grad {k}")
            for k, v in split(g, "grad/sepcrit").items():
                rel_close(gs[k].cpu().numpy(), v, f"second-critic grad {k}")
            if live:
                gc = e.lc.unflatten(e.gc)
                for k, v in split(g, "grad/critic").items():
                    rel_close(gc[k].cpu().numpy(), v, f"critic grad {k}")
    for k, v in split(g, "step3/sepcrit").items():
        rel_close(e.sepcrit_state()[k].cpu().numpy(), v, f"second critic {k} after step 3", rtol=1e-3, atol_scale=1e-4)
    for k, v in split(g, "step3/masker").items():
        rel_close(e.masker_state()[k].cpu().numpy(), v, f"masker {k} after step 3", rtol=1e-3, atol_scale=1e-4)
    for k, v in split(g, "step3/critic").items():
        rel_close(e.critic_state()[k].cpu().numpy(), v, f"critic {k} after step 3", rtol=1e-3, atol_scale=1e-4)


@pytest.mark.parametrize("one_launch", [True, False])
def test_device_batch_assembly_matches_host_gather_and_roll(monkeypatch, one_launch):
    """cgs_gather_contrastive (one launch, round 5) and cgs_gather_roll_u8 / cgs_gather_f32 (five) -- main.py:344-356 + shift_batch 584-591 on the
    device -- vs numpy fancy indexing + torch.roll."""
    from cgs_amd import engine
    monkeypatch.setattr(engine, "GATHER_ONE_LAUNCH", one_launch)
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(3)
    Xpos = rs.randint(0, 256, (50, 64, 64, 3)).astype(np.uint8)
    Xneg = rs.randint(0, 256, (70, 64, 64, 3)).astype(np.uint8)
    ypos, yneg = rs.rand(50).astype(np.float32), rs.rand(70).astype(np.float32)
    n = 16
    e = engine.HourglassEngine(n, dropout=0.0)
    for roll in (0, 5, -7, 11, -12):
        H, L, Cc = rs.choice(50, n // 2), rs.choice(70, n // 2), rs.choice(70, n)
        idx = torch.from_numpy(np.concatenate((H, L, Cc))).to(dev)
        e.gather_contrastive(torch.from_numpy(Xpos).to(dev), torch.from_numpy(Xneg).to(dev), torch.from_numpy(ypos).to(dev),
                             torch.from_numpy(yneg).to(dev), idx, shift_px=(-roll) % 64)
        torch.cuda.synchronize()
        want_a = torch.roll(torch.from_numpy(np.concatenate((Xpos[H], Xneg[L]))), shifts=roll, dims=2).numpy()
        np.testing.assert_array_equal(e.ab[n:].cpu().numpy(), want_a)
        np.testing.assert_array_equal(e.ab[:n].cpu().numpy(), Xneg[Cc])
        np.testing.assert_array_equal(e.y.cpu().numpy(), np.concatenate((ypos[H], yneg[L])))


def test_infer_with_dropout_noevalmode(g1):
    """-noevalmode (main.py:1109-1118): Dropout stays active at inference -- a fresh mask per call, eval-mode results otherwise."""
    e = make_engine(g1, 8, dropout=0.3)
    dev = torch.device("cuda:0")
    X = torch.from_numpy(np.random.RandomState(8).randint(0, 256, (40, 64, 64, 3)).astype(np.uint8)).to(dev)
    p0, z0 = e.infer(X)
    p1, z1 = e.infer(X, train_mode=True)
    p2, z2 = e.infer(X, train_mode=True)
    p3, z3 = e.infer(X)
    assert torch.equal(p0, p3) and torch.equal(z0, z3)                 # eval mode is deterministic
    assert not torch.equal(p1, p2) and not torch.equal(p0, p1)         # train mode draws new masks every call
    assert float((z1 - z0).abs().max()) > 0 and torch.isfinite(z1).all() and float(z1.min()) > 0 and float(z1.max()) < 1


@pytest.mark.parametrize("tag,thr", [("g4_phase1_mse", 0.0), ("g4_phase1_bce", 0.5)])
def test_phase1_matches_reference_capture(golden, g1, tag, thr):
    g = golden(tag + ".npz")
    dev = torch.device("cuda:0")
    e = make_engine(g1, 8, dropout=0.0, threshrew=thr)
    loss = e.phase1_step(torch.from_numpy(g["X"]).to(dev), torch.from_numpy(g["Y"]).to(dev)).cpu().numpy()
    assert float(loss[0]) == pytest.approx(float(g["loss"]), rel=1e-3)
    rel_close(e.cbuf["pred"][:8].cpu().numpy(), g["pred"], "pred")
    gc = e.lc.unflatten(e.gc)
    for k, v in split(g, "grad").items():
        rel_close(gc[k].cpu().numpy(), v, f"grad {k}")
    for k, v in split(g, "step1").items():
        rel_close(e.critic_state()[k].cpu().numpy(), v, f"{k} after step 1", atol_scale=1e-4)


def _export_masks(e, n_imgs_total, step_value):
    """Keep-masks of the three dropout sites for all image slots of the engine's batch buffer."""
    from cgs_amd import _lib
    out = []
    for site, per_img in ((0, (8, 8, 8)), (1, (4, 4, 16)), (2, (32,))):
        cnt = n_imgs_total * int(np.prod(per_img))
        buf = torch.empty(cnt, device=e.dev)
        _lib.call("cgs_dropout_mask", e.drop.desc(site), cnt, C.c_void_p(buf.data_ptr()),
                  C.c_void_p(torch.cuda.current_stream().cuda_stream))
        mk = (buf.cpu().reshape((n_imgs_total,) + per_img) != 0).float()
        out.append(mk.permute(0, 3, 1, 2).contiguous() if len(per_img) == 3 else mk)
    return out


@pytest.mark.parametrize("n", [12, 512])
def test_phase2_with_dropout_vs_oracle_masks(g1, n):
    """dropout 0.3, train mode: the oracle is fed the keep-masks the kernels drew (slot order B, A, rep, inj).
    n = 512 is the benchmark configuration itself (BASELINE.json config 2): losses, all 28 gradients, updated parameters."""
    rs = np.random.RandomState(42)
    dev = torch.device("cuda:0")
    A = rs.randint(0, 256, (n, 64, 64, 3)).astype(np.uint8)
    B = rs.randint(0, 256, (n, 64, 64, 3)).astype(np.uint8)
    Y = rs.rand(n).astype(np.float32)
    e = make_engine(g1, n, dropout=0.3, use_graph=True)
    start = 7 if n == 12 else 0   # masks depend on the step counter; export them for that step before the step ticks it
    e.step_t.fill_(start)
    masks = _export_masks(e, 4 * n, start)
    losses = e.phase2_step(torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev), torch.from_numpy(Y).to(dev)).cpu().numpy()
    assert int(e.step_t.item()) == start + 1
    sl = {"B": slice(0, n), "A": slice(n, 2 * n), "rep": slice(2 * n, 3 * n), "inj": slice(3 * n, 4 * n)}
    omasks = [[m[sl[k]] for m in masks] for k in ("A", "B", "rep", "inj")]  # oracle order: A, B, replaced, injected
    pc, pm = g1
    if n == 512:
        # At 2 M pixels per weight the CPU's own fp32 summation is off by 1-3e-3 of a tensor's maximum on the mask-head
        # gradients (measured against float64), more than the tolerance: the checker runs in float64 here.
        dd = lambda P: {k: v.double() for k, v in P.items()}
        rec = orc.train_phase2(dd(pc), dd(pm), [(orc.u8_to_nchw(A).double(), orc.u8_to_nchw(B).double(), torch.from_numpy(Y).double())],
                               steps=1, p=0.3, training=True, masks=[[m.double() for m in mm] for mm in omasks])[0]
    else:
        rec = orc.train_phase2(pc, pm, [(orc.u8_to_nchw(A), orc.u8_to_nchw(B), torch.from_numpy(Y))], steps=1,
                               p=0.3, training=True, masks=omasks)[0]
    parts = rec["parts"]
    np.testing.assert_allclose(losses[:4], [parts["critic"], parts["replace"], parts["inject"], parts["norm"]], rtol=1e-3)
    gc, gm = e.lc.unflatten(e.gc), e.lm.unflatten(e.gm)
    for k, v in rec["grads_c"].items():
        rel_close(gc[k].cpu().numpy(), v.numpy(), f"critic grad {k}")
    for k, v in rec["grads_m"].items():
        rel_close(gm[k].cpu().numpy(), v.numpy(), f"masker grad {k}")
    if start == 0:     # Adam's bias correction follows the same counter: only a run from step 0 is the oracle's first step
        for k, v in rec["params_c"].items():
            rel_close(e.critic_state()[k].cpu().numpy(), v.numpy(), f"critic {k} after the step", rtol=1e-3, atol_scale=1e-4)
        for k, v in rec["params_m"].items():
            rel_close(e.masker_state()[k].cpu().numpy(), v.numpy(), f"masker {k} after the step", rtol=1e-3, atol_scale=1e-4)


def test_graph_replay_equals_eager_and_is_reproducible(g1):
    n = 16
    rs = np.random.RandomState(1)
    dev = torch.device("cuda:0")
    A = torch.from_numpy(rs.randint(0, 256, (n, 64, 64, 3)).astype(np.uint8)).to(dev)
    B = torch.from_numpy(rs.randint(0, 256, (n, 64, 64, 3)).astype(np.uint8)).to(dev)
    Y = torch.from_numpy(rs.rand(n).astype(np.float32)).to(dev)
    finals = []
    for use_graph in (False, True, True):
        e = make_engine(g1, n, dropout=0.3, use_graph=use_graph)
        for _ in range(4):
            e.phase2_step(A, B, Y)
        finals.append(e.flat.cpu().numpy().copy())
    np.testing.assert_array_equal(finals[1], finals[2])   # slab reduction: no float atomics on the parameter path
    np.testing.assert_array_equal(finals[0], finals[1])


def test_large_batch_properties(g1):
    """Full bench size (N=512): size-independent properties instead of an oracle run."""
    n = 512
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(0)
    A = torch.randint(0, 256, (n, 64, 64, 3), dtype=torch.uint8, generator=gen).to(dev)
    B = torch.randint(0, 256, (n, 64, 64, 3), dtype=torch.uint8, generator=gen).to(dev)
    Y = torch.rand(n, generator=gen).to(dev)
    e = make_engine(g1, n, dropout=0.0)
    p0 = e.flat.clone()
    l0 = e.phase2_step(A, B, Y).cpu().numpy().copy()
    g_full = e.grad.cpu().numpy().copy()
    assert np.isfinite(l0).all() and np.isfinite(g_full).all()
    Z = e.mbuf["Z"]
    assert float(Z.min()) > 0 and float(Z.max()) < 1
    # the batch-mean gradient equals the mean of the two half-batch gradients (linearity; what DP relies on)
    halves = []
    for sl in (slice(0, n // 2), slice(n // 2, n)):
        h = make_engine(g1, n // 2, dropout=0.0, use_graph=False)
        h.phase2_step(A[sl], B[sl], Y[sl])
        halves.append(h.grad.cpu().numpy().copy())
    rel_close(g_full, 0.5 * (halves[0] + halves[1]), "full-batch grad vs mean of half-batch grads", rtol=2e-3, atol_scale=1e-4)
    # a step moved every parameter by at most ~lr (Adam's first step is +-lr)
    delta = (e.flat - p0).abs().max().item()
    assert 0 < delta <= 1.01e-3


def test_saliency_gradient_matches_oracle_autograd(g1):
    """engine.saliency (eval-mode critic forward + backward to the input, main.py:941-953) vs torch autograd on the oracle."""
    from cgs_amd import engine
    pc, pm = g1
    dev = torch.device("cuda:0")
    n = 19
    x_u8 = np.random.RandomState(91).randint(0, 256, (n, 64, 64, 3)).astype(np.uint8)
    Xc = orc.u8_to_nchw(x_u8).clone().requires_grad_(True)
    pred = orc.critic_apply(pc, Xc)
    pred.mean().backward()
    want = Xc.grad.permute(0, 2, 3, 1).numpy()
    eng = engine.HourglassEngine(8, device=dev, dropout=0.3)
    eng.load_state(pc, pm)
    xd = (torch.from_numpy(x_u8).to(dev).double() / 255.0).float()
    p, dx = eng.saliency(xd)
    torch.cuda.synchronize()
    np.testing.assert_allclose(p.cpu().numpy(), pred.detach()[:, 0].numpy(), rtol=1e-3, atol=1e-6)
    got = dx.cpu().numpy()
    scale = np.abs(want).max()
    assert scale > 0
    assert np.abs(got - want).max() <= 1e-3 * scale + 1e-9, (np.abs(got - want).max(), scale)
    # the map the reference thresholds: |grad| summed over the colour channels
    np.testing.assert_allclose(np.abs(got).sum(-1), np.abs(want).sum(-1), rtol=2e-3, atol=2e-3 * scale)


@pytest.mark.parametrize("n", [1, 3, 7, 33])
def test_ragged_batch_sizes_vs_oracle(g1, n):
    """The last batch of an epoch has any size (main.py's DataLoader keeps it): phase 2 and phase 1 at odd / tiny n against the
    oracle -- losses, all gradients (weight-gradient tiles, slab counts and the image loops of the tail kernels all depend on n)."""
    rs = np.random.RandomState(100 + n)
    dev = torch.device("cuda:0")
    A = rs.randint(0, 256, (n, 64, 64, 3)).astype(np.uint8)
    B = rs.randint(0, 256, (n, 64, 64, 3)).astype(np.uint8)
    Y = rs.rand(n).astype(np.float32)
    pc, pm = g1
    e = make_engine(g1, n, dropout=0.0)
    losses = e.phase2_step(torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev), torch.from_numpy(Y).to(dev)).cpu().numpy()
    rec = orc.train_phase2(pc, pm, [(orc.u8_to_nchw(A), orc.u8_to_nchw(B), torch.from_numpy(Y))], steps=1)[0]
    assert losses[5] == pytest.approx(rec["total"], rel=1e-3)
    gc, gm = e.lc.unflatten(e.gc), e.lm.unflatten(e.gm)
    for k, v in rec["grads_c"].items():
        rel_close(gc[k].cpu().numpy(), v.numpy(), f"critic grad {k} (n={n})")
    for k, v in rec["grads_m"].items():
        rel_close(gm[k].cpu().numpy(), v.numpy(), f"masker grad {k} (n={n})")
    e1 = make_engine(g1, n, dropout=0.0)
    l1 = e1.phase1_step(torch.from_numpy(A).to(dev), torch.from_numpy(Y).to(dev)).cpu().numpy()
    r1 = orc.train_phase1(pc, [(orc.u8_to_nchw(A), torch.from_numpy(Y))], steps=1, p=0.0)[0]
    assert l1[0] == pytest.approx(r1["loss"], rel=1e-3)
    g1c = e1.lc.unflatten(e1.gc)
    for k, v in r1["grads"].items():
        rel_close(g1c[k].cpu().numpy(), v.numpy(), f"phase-1 grad {k} (n={n})")


@pytest.mark.parametrize("n", [37, 600, 1100])
def test_per_image_fusions_are_bit_identical_to_the_launches_they_replace(g1, monkeypatch, n):
    """Round 5: features.3's data gradient behind the encoder tail's backward (cgs_tail_enc_bwd_enc1; its weight gradient as extra workgroups
    of the features.0 backward launch: cgs_enc0_bwd_mix_enc1 / cgs_enc0_wgrad_u8_with_head_enc1) against cgs_tail_enc_bwd_rider +
    cgs_conv3x3_bwd_both.  Round 4: the whole critic forward of an image in one workgroup (cgs_critic_fwd_fused), decoder tail + dec_model.0
    (cgs_tail_dec_fwd_dec0) and dec_model.0's data gradient + decoder tail backward (cgs_dec0_tail_dec_bwd) run the same kernel bodies in
    the same order as the separate launches (cgs_conv3x3_fwd x2 + cgs_tail_enc_fwd, cgs_tail_dec_fwd_pack + cgs_conv3x3_fwd,
    cgs_conv3x3_bwd_data + cgs_tail_dec_bwd): two dropout-0.3 training steps and an inference give bitwise equal losses, parameters,
    Adam moments, critic values and masks with the fusions switched off one by one and all together (this also keeps the un-fused entry
    points exercised).  n = 600: fused decoder forward with the un-fused decoder backward (its one-workgroup-per-image form stops at 512);
    n = 1100: beyond the forward cap too -- the entry points answer CGS_ERR_UNSUPPORTED and the host falls back (all fusions toggled together
    at those sizes)."""
    from cgs_amd import engine, hourglass as hg
    pc, pm = g1
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(21)
    A = torch.from_numpy(rs.randint(0, 256, (n, 64, 64, 3)).astype(np.uint8)).to(dev)
    B = torch.from_numpy(rs.randint(0, 256, (n, 64, 64, 3)).astype(np.uint8)).to(dev)
    Y = torch.from_numpy(rs.rand(n).astype(np.float32)).to(dev)

    def run():
        e = engine.HourglassEngine(n, device=dev, dropout=0.3, use_graph=False)
        e.load_state(pc, pm)
        for _ in range(2):
            e.phase2_step(A, B, Y)
        pred, Z = e.infer(A)
        torch.cuda.synchronize()
        return [t.clone() for t in (e.losses, e.flat, e.m, e.v, pred, Z)]

    # features.3's weight gradient: formed inside the fused tail backward kernel by default (ENC1_WGRAD_IN_TAIL: one slab row per workgroup over
    # ITS images -- a different partition of the sum over the images than the stand-alone launch's, so equal only up to fp32 summation
    # order); with that switched off it rides as extra workgroups that reproduce the stand-alone launch's slabs bit for bit.  The bitwise
    # comparisons therefore run in the rider form; the default form is compared with it at the end, within a summation-order tolerance.
    # (the per-image forms are used from a batch size on -- hourglass.MASK_TRAIN_FUSED_MIN_N / ENC1_TAIL_BWD_FUSED_MIN_N --: here at every n)
    small_batch = run() if n < 100 else None                 # the small-batch forms a batch of 37 runs by default (compared at the end)
    monkeypatch.setattr(hg, "MASK_TRAIN_FUSED_MIN_N", 0)
    monkeypatch.setattr(hg, "ENC1_TAIL_BWD_FUSED_MIN_N", 0)
    default = run()
    monkeypatch.setattr(hg, "ENC1_WGRAD_IN_TAIL", False)
    monkeypatch.setattr(hg, "ENC0_WGRAD_IN_TAIL", "")        # (features.0's weight gradient in the same kernel: likewise a per-workgroup partition)
    monkeypatch.setattr(hg, "DEC3_WGRAD_RIDER", False)       # (dec_model.3's weight gradient as a GEMM over the images in 64 rider workgroups: likewise)
    base = run()
    flags = ("CRITIC_FWD_FUSED", "ENC1_TAIL_FUSED", "DEC_TAIL_DEC0_FUSED", "DEC0_TAIL_BWD_FUSED", "ENC1_TAIL_BWD_FUSED")
    for off in ([(f,) for f in flags] if n < 100 else []) + [flags]:
        for f in off:
            monkeypatch.setattr(hg, f, False)
        got = run()
        for f in off:
            monkeypatch.setattr(hg, f, True)
        for a, b, what in zip(got, base, ("losses", "parameters", "m", "v", "pred", "Z")):
            assert torch.equal(a, b), f"{what} differ with {off} switched off"
    monkeypatch.setattr(hg, "ENC1_WGRAD_IN_TAIL", True)
    monkeypatch.setattr(hg, "DEC3_WGRAD_RIDER", True)
    monkeypatch.setattr(hg, "ENC0_WGRAD_IN_TAIL", "both")    # the measured-slower opt-in: features.0's weight gradient in the tail kernel too
    opt_in = run()
    monkeypatch.setattr(hg, "ENC0_WGRAD_IN_TAIL", "")
    if small_batch is not None:
        # below the thresholds: the mask head forward as two launches (Z equal to ~2e-5 relative: another summation order) and features.3's backward
        # as launches of its own -- the same step within that tolerance
        for a, b, what in zip(small_batch, base, ("losses", "parameters", "m", "v", "pred", "Z")):
            d = float((a - b).abs().max())
            print(f"small-batch forms vs per-image forms, {what}: max |diff| {d:.3e}")
            assert torch.allclose(a, b, rtol=1e-4, atol=1e-6), f"{what}: small-batch launch forms vs the per-image forms: {d}"
    for got in (default, opt_in):
        for a, b, what in zip(got, base, ("losses", "parameters", "m", "v", "pred", "Z")):
            # two Adam steps: a gradient element that differs by an ulp moves its parameter by up to ~lr * 1e-3 (Adam's normalised step)
            assert torch.allclose(a, b, rtol=2e-4, atol=2e-6), \
                f"{what}: the in-kernel weight gradients of features.3 / features.0 and dec_model.3's rider GEMM vs the stand-alone forms: {float((a - b).abs().max())}"
