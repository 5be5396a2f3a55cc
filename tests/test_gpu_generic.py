"""GPU parity of the shape-generic kernels (csrc/gen.hip): NewCritic / UnetDecoder at chfak != 1 (the paper's chfak = 5 against
the committed reference capture), the legacy Unet with its ConvTranspose2d decoder (reference capture G8), and the transposed
convolution's forward / data gradient / weight gradient against torch's CPU autograd."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import hourglass_ref as orc
from test_gpu_kernels import rel_close, nhwc


def test_chfak5_modules_match_reference_capture(golden):
    """The paper's model size (channels 40/40/40/80/160): NewCritic(chfak=5) / UnetDecoder(chfak=5) in eval mode against the
    capture of the reference classes (tests/golden/g2_eval_chfak5.npz) and against the oracle for every embed."""
    from cgs_amd import nets
    g = golden("g2_eval_chfak5.npz")
    pc = orc.seeded_params(orc.critic_shapes(5), 11)
    pm = orc.seeded_params(orc.masker_shapes(5), 12)
    critic, masker = nets.NewCritic(chfak=5).to("cuda").eval(), nets.UnetDecoder(chfak=5).to("cuda").eval()
    critic.load_state_dict(pc); masker.load_state_dict(pm)
    X = orc.u8_to_nchw(g["X"])
    with torch.no_grad():
        pred, embeds = critic(X.cuda(), collect=True)
        Z = masker(X.cuda(), embeds)
        rp, re = orc.critic_apply(pc, X, collect=True)
        rz, inter = orc.masker_apply(pm, X, re, return_all=True)
    rel_close(pred.cpu().numpy(), g["pred"], "pred vs reference capture")
    rel_close(Z.cpu().numpy(), g["Z"], "Z vs reference capture")
    rel_close(embeds[4].cpu().numpy(), g["e4"], "e4 vs reference capture")
    for i in range(4):
        rel_close(embeds[i].cpu().numpy(), re[i].numpy(), f"e{i} vs oracle")
    # round trip of the checkpoint contract at this size
    for k, v in critic.state_dict().items():
        np.testing.assert_array_equal(v.cpu().numpy(), pc[k].numpy())
    # train mode works at this size too (shape-generic backward kernels; tests/test_gpu_generic_train.py)
    critic.train()(X.cuda().requires_grad_(True)).sum().backward()
    assert all(q.grad is not None and q.grad.shape == q.shape and bool(torch.isfinite(q.grad).all()) for q in critic.parameters())


@pytest.mark.parametrize("chfak,neck", [(1, 32), (2, 32), (3, 16)])
def test_generic_forward_other_sizes_match_oracle(chfak, neck):
    from cgs_amd import generic as gen, spec
    dev = torch.device("cuda:0")
    pc = orc.seeded_params(orc.critic_shapes(chfak, neck), 21)
    pm = orc.seeded_params(orc.masker_shapes(chfak, neck), 22)
    lc, lm = spec.critic_layout(chfak, neck), spec.masker_layout(chfak, neck)
    fc, fm = torch.empty(lc.total, device=dev), torch.empty(lm.total, device=dev)
    lc.flatten({k: v.to(dev) for k, v in pc.items()}, fc)
    lm.flatten({k: v.to(dev) for k, v in pm.items()}, fm)
    x_u8 = np.random.RandomState(4).randint(0, 256, (5, 64, 64, 3)).astype(np.uint8)
    X = orc.u8_to_nchw(x_u8)
    with torch.no_grad():
        rp, re = orc.critic_apply(pc, X, collect=True)
        rz, inter = orc.masker_apply(pm, X, re, return_all=True)
    c = gen.critic_forward(fc, lc, torch.from_numpy(x_u8).to(dev), chfak, neck)          # uint8 loader
    m = gen.masker_forward(fm, lm, torch.from_numpy(x_u8).to(dev), [c[f"e{i}"] for i in range(5)], chfak, neck)
    for i in range(4):
        rel_close(nhwc(c[f"e{i}"]), re[i].numpy(), f"e{i}")
    rel_close(c["pred"].cpu().numpy(), rp[:, 0].numpy(), "pred")
    for k in ("o3", "o2", "o1", "o0", "hm"):
        rel_close(nhwc(m[k]), inter[k].numpy(), k)
    rel_close(m["Z"].cpu().numpy(), rz[:, 0].numpy(), "Z")
    if chfak == 1 and neck == 32:      # the generic and the specialised kernels agree on the default size
        from cgs_amd import hourglass as hg
        c1 = hg.critic_forward(fc, lc, torch.from_numpy(x_u8).to(dev), 5)
        rel_close(c["pred"].cpu().numpy(), c1["pred"].cpu().numpy(), "generic vs specialised pred", rtol=1e-5)


def test_generic_pool_argmax_first_index():
    from cgs_amd import generic as gen
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(0)
    x = torch.from_numpy(np.repeat(np.repeat(rs.rand(2, 4, 4, 8).astype(np.float32), 2, axis=1), 2, axis=2))   # 2x2 constant blocks
    w = torch.zeros(3, 3, 8, 8); w[1, 1] = torch.eye(8)                                                      # identity conv
    b = torch.zeros(8)
    wd, bd = w.reshape(-1).to(dev), b.to(dev)
    out, am = gen.conv3x3(x.to(dev), None, wd.data_ptr(), bd.data_ptr(), 8, act="lrelu", slope=0.2, pool=True, want_argmax=True)
    ref, idx = F.max_pool2d(x.permute(0, 3, 1, 2), 2, return_indices=True)
    np.testing.assert_allclose(nhwc(out), ref.numpy(), rtol=1e-6)
    assert (am.cpu().numpy() == 0).all()          # every window is a 4-way tie: the first position wins, as max_pool2d


def test_legacy_unet_convtranspose_matches_reference_capture(golden):
    """nets.Unet(upsample=False): ConvTranspose2d(4,2,1) decoder + LeakyReLU(0.2) (nets.py:356-449) vs the reference class."""
    from cgs_amd import nets
    g = golden("g8_unet_convt.npz")
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd/")}
    net = nets.Unet(upsample=False).to("cuda").eval()
    assert list(net.state_dict().keys()) == list(sd.keys())
    net.load_state_dict(sd)
    X = orc.u8_to_nchw(g["X"]).cuda()
    with torch.no_grad():
        y, u0 = net(X, embeds=True)
        c = net(X, critic=True)
    rel_close(y.cpu().numpy(), g["y"], "mask")
    rel_close(u0.cpu().numpy(), g["u0"], "u0")
    rel_close(c.cpu().numpy(), g["critic"], "critic value")
    with pytest.raises(NotImplementedError):
        nets.Unet(upsample=True)


@pytest.mark.parametrize("n,h,ca,cb,co", [(3, 4, 16, 16, 8), (2, 8, 8, 8, 8), (2, 16, 8, 8, 8), (1, 32, 8, 8, 1), (2, 8, 5, 0, 3)])
def test_convtranspose_4_2_1_forward_and_gradients_vs_torch(n, h, ca, cb, co):
    """cgs_gen_convt4s2_{fwd,bwd_data,bwd_weight} against F.conv_transpose2d(stride 2, padding 1) and its CPU autograd."""
    from cgs_amd import generic as gen
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(n * 100 + h)
    a = torch.from_numpy(rs.randn(n, ca, h, h).astype(np.float32)).requires_grad_(True)
    b = torch.from_numpy(rs.randn(n, cb, h, h).astype(np.float32)).requires_grad_(True) if cb else None
    W = torch.from_numpy((rs.randn(ca + cb, co, 4, 4) * 0.2).astype(np.float32)).requires_grad_(True)
    bias = torch.from_numpy(rs.randn(co).astype(np.float32)).requires_grad_(True)
    cot = torch.from_numpy(rs.randn(n, co, 2 * h, 2 * h).astype(np.float32))
    x = torch.cat((a, b), dim=1) if cb else a
    pre = F.conv_transpose2d(x, W, bias, stride=2, padding=1)
    out = F.leaky_relu(pre, 0.2)
    (out * cot).sum().backward()
    dpre = (cot * torch.where(pre > 0, torch.ones_like(pre), torch.full_like(pre, 0.2))).detach()
    to_d = lambda t: t.detach().permute(0, 2, 3, 1).contiguous().to(dev)
    wk = gen.convt_weight_to_kernel(W.detach()).to(dev)
    got = gen.convt_fwd(to_d(a), to_d(b) if cb else None, wk, bias.detach().to(dev), act="lrelu", slope=0.2)
    rel_close(nhwc(got), out.detach().numpy(), "forward")
    da, db, dw, dbias = gen.convt_bwd(to_d(a), to_d(b) if cb else None, wk, to_d(dpre))
    rel_close(nhwc(da), a.grad.numpy(), "d A")
    if cb:
        rel_close(nhwc(db), b.grad.numpy(), "d B")
    rel_close(gen.convt_weight_from_kernel(dw.cpu(), ca + cb, co).numpy(), W.grad.numpy(), "d W")
    rel_close(dbias.cpu().numpy(), bias.grad.numpy(), "d bias")


def test_handler_generic_inference_chfak5(tmp_path, monkeypatch):
    """`main.py -process --chfak 5`'s inner loop: Handler routes other model sizes to the shape-generic engine."""
    from cgs_amd import cli, handler
    monkeypatch.chdir(tmp_path)
    H = handler.Handler(cli.parse_args(["--model", "m", "--chfak", "5"]))
    pc = orc.seeded_params(orc.critic_shapes(5), 11)
    pm = orc.seeded_params(orc.masker_shapes(5), 12)
    H.critic.load_state_dict(pc); H.masker.load_state_dict(pm)
    x_u8 = np.random.RandomState(2).randint(0, 256, (6, 64, 64, 3)).astype(np.uint8)
    pred, Z = H._engine(64).infer(torch.from_numpy(x_u8).cuda())
    with torch.no_grad():
        rp, re = orc.critic_apply(pc, orc.u8_to_nchw(x_u8), collect=True)
        rz = orc.masker_apply(pm, orc.u8_to_nchw(x_u8), re)
    rel_close(pred.cpu().numpy(), rp[:, 0].numpy(), "pred")
    rel_close(Z.cpu().numpy(), rz[:, 0].numpy(), "Z")
    from cgs_amd import generic_engine
    assert isinstance(H._engine(64, training=True), generic_engine.GenericEngine)


@pytest.mark.parametrize("chfak,neck,n", [(1, 32, 64), (5, 32, 12), (2, 16, 9)])
def test_fp16_inference_error_distribution_vs_fp32_oracle(g1, chfak, neck, n):
    """BASELINE config 4 (fp16 conv kernels): fp16 activations / weights, fp32 accumulation in EVERY layer (csrc/gen_f16.hip) against
    the fp32 oracle: the error distribution of the masks and the critic values (opt-in precision mode, so an absolute bound, not
    the 1e-3 relative one of the fp32 path).  chfak 1 runs on the fused engine's parameters, the others on GenericEngine."""
    from cgs_amd import engine, generic_engine
    dev = torch.device("cuda:0")
    if chfak == 1:
        pc, pm = g1
        e = engine.HourglassEngine(8, dropout=0.0)
    else:
        pc, pm = orc.seeded_params(orc.critic_shapes(chfak, neck), 11), orc.seeded_params(orc.masker_shapes(chfak, neck), 12)
        e = generic_engine.GenericEngine(8, chfak=chfak, neck=neck, dropout=0.0)
    e.load_state(pc, pm)
    rs = np.random.RandomState(5)
    x = rs.randint(0, 256, (n, 64, 64, 3)).astype(np.uint8)
    x[0, 16:40, 8:56] = 200        # a flat patch: ties in the pooling windows
    pred, Z = e.infer(torch.from_numpy(x).to(dev), fp16=True)
    with torch.no_grad():
        rp, re = orc.critic_apply(pc, orc.u8_to_nchw(x), collect=True)
        rz = orc.masker_apply(pm, orc.u8_to_nchw(x), re)[:, 0].numpy()
    dz = np.abs(Z.cpu().numpy().astype(np.float64) - rz)
    dp = np.abs(pred.cpu().numpy().astype(np.float64) - rp[:, 0].numpy())
    print(f"fp16 inference chfak {chfak}: |dZ| max {dz.max():.2e} p99.9 {np.quantile(dz, 0.999):.2e} mean {dz.mean():.2e}; |dpred| max {dp.max():.2e}")
    # measured (G1 / seeded weights): max 1.7-2.7e-5, p99.9 1.3-2.0e-5, mean 3.0-4.9e-6, |dpred| 2-9e-6: bounds ~3x above
    assert dz.max() < 1e-4 and np.quantile(dz, 0.999) < 6e-5 and dz.mean() < 1.5e-5
    assert dp.max() < 4e-5
    # thresholded masks (what -process writes) differ on a vanishing fraction of the pixels
    assert ((Z.cpu().numpy() > 0.5) != (rz > 0.5)).mean() < 2e-3
    # the weight copies follow the parameters: a training step changes the fp16 result
    A = torch.from_numpy(x[:8]).to(dev)
    e.phase2_step(A, torch.from_numpy(x[:8][::-1].copy()).to(dev), torch.rand(8, device=dev))
    pred2, _ = e.infer(torch.from_numpy(x).to(dev), fp16=True)
    p32, _ = e.infer(torch.from_numpy(x).to(dev))
    assert float((pred2 - p32).abs().max()) < 5e-3 and not torch.equal(pred2, pred)


def test_config4_batch2048_fp16_inference_vs_fp32_oracle(g1):
    """BASELINE config 4 AT ITS STATED SIZE: the -process inference path on 2048 frames with fp16 conv kernels, three forms -- the
    fused fp16 path (engine.infer(fp16=True), round 4: csrc/hconv.hip + the fp16 mask head), every layer fp16 layer by layer
    (fp16_layerwise=True, csrc/gen_f16.hip) and fp16 operands in the mask head only (fp16_mask_head=True) --
    against the fp32 CPU oracle on the same frames.  The bounds are ~3x the measured distribution on the G1 weights (printed), so
    a regression of the kernels' precision by a factor of a few fails here."""
    from cgs_amd import engine
    dev = torch.device("cuda:0")
    pc, pm = g1
    e = engine.HourglassEngine(8, dropout=0.0)
    e.load_state(pc, pm)
    rs = np.random.RandomState(9)
    x = rs.randint(0, 256, (2048, 64, 64, 3)).astype(np.uint8)
    x[:256] = (x[:256] * 0.25).astype(np.uint8)          # a darker population: a spread of critic values
    x[300, 16:40, 8:56] = 200                            # flat patches: pooling ties
    with torch.no_grad():
        rp, rz = [], []
        for b in range(0, 2048, 256):
            p, emb = orc.critic_apply(pc, orc.u8_to_nchw(x[b:b + 256]), collect=True)
            rp.append(p[:, 0]); rz.append(orc.masker_apply(pm, orc.u8_to_nchw(x[b:b + 256]), emb)[:, 0])
        rp, rz = torch.cat(rp).numpy(), torch.cat(rz).numpy()
    xd = torch.from_numpy(x).to(dev)
    p32, z32 = e.infer(xd)
    rel_close(z32.cpu().numpy(), rz, "fp32 Z at batch 2048")
    rel_close(p32.cpu().numpy(), rp, "fp32 pred at batch 2048")
    for name, kw, zmax, zmean, pmax in (("fused fp16 path (hconv + fp16 mask head)", dict(fp16=True), 2e-4, 3e-5, 6e-5),
                                        ("all layers fp16, layer by layer", dict(fp16=True, fp16_layerwise=True), 1e-4, 1.5e-5, 4e-5),
                                        ("fp16 mask-head operands", dict(fp16_mask_head=True), 1.2e-4, 1.5e-5, 1e-6)):
        pred, Z = e.infer(xd, **kw)
        dz = np.abs(Z.cpu().numpy().astype(np.float64) - rz)
        dp = np.abs(pred.cpu().numpy().astype(np.float64) - rp)
        print(f"config 4, batch 2048, {name}: |dZ| max {dz.max():.2e} p99.9 {np.quantile(dz, 0.999):.2e} mean {dz.mean():.2e}; "
              f"|dpred| max {dp.max():.2e}; Z range [{rz.min():.3f}, {rz.max():.3f}]")
        assert dz.max() < zmax and dz.mean() < zmean, name
        assert dp.max() < pmax, name
        assert ((Z.cpu().numpy() > 0.5) != (rz > 0.5)).mean() < 1e-3


def test_config4_tail_launch_forms_agree(g1, monkeypatch):
    """The four forms of the fused fp16 inference path between features.0 and dec_model.0 (engine.F16_TAILS: fp32 tail kernels, the two tail launches
    with fp16 operands -- csrc/tail_h16.h --, one launch per image on fp16 tiles -- csrc/tail_infer.hip, the default --, and that launch with
    features.3 in front) against each other: same frames, same weights; they differ by fp16 roundings of the 16x16-and-smaller maps only."""
    from cgs_amd import engine
    dev = torch.device("cuda:0")
    pc, pm = g1
    e = engine.HourglassEngine(8, dropout=0.0)
    e.load_state(pc, pm)
    rs = np.random.RandomState(4)
    xall = torch.from_numpy(rs.randint(0, 256, (1100, 64, 64, 3)).astype(np.uint8)).to(dev)
    for b in (1, 3, 300, 1100):      # (ragged: not multiples of anything the kernels tile by; 1100 = more images than persistent workgroups)
        x = xall[:b].contiguous()
        out = {}
        for form in ("0", "1", "fused", "fused1"):
            monkeypatch.setattr(engine, "F16_TAILS", form)
            p, z = e.infer(x, fp16=True)
            p2, _ = e.infer(x, fp16=True, want_mask=False)
            assert torch.equal(p, p2), form            # the critic-only instance computes the same values
            out[form] = (p.cpu().numpy().astype(np.float64), z.cpu().numpy().astype(np.float64))
        for form in ("1", "fused", "fused1"):
            dp = np.abs(out[form][0] - out["0"][0]).max()
            dz = np.abs(out[form][1] - out["0"][1]).max()
            print(f"config 4 tails, batch {b}, form {form!r} vs the fp32 tail kernels: |dpred| max {dp:.2e}, |dZ| max {dz:.2e}")
            assert dp < 2e-5 and dz < 5e-5, (b, form)
        # a batch is the concatenation of its images: image 0 of every batch equals the one-image batch's result bit for bit
        if b == 1:
            first = out["fused"]
        else:
            assert out["fused"][0][0] == first[0][0] and np.array_equal(out["fused"][1][0], first[1][0]), b


def test_legacy_unet_trains_through_the_module_vs_reference_capture(golden):
    """nets.Unet(upsample=False) under autograd (nets.py:356-449, built at TrainHandler.py:159-161): one
    training step of the REFERENCE class (loss = MSE(mask, target) + MSE(critic value, target), captured by make_golden.py) --
    the two losses and all 24 parameter gradients, through ConvTranspose2d(4,2,1) / (4,1,0), LeakyReLU(0.2), MaxPool2d and the
    critic's Linear layers, every one on the HIP kernels (_UnetFn).  Then an optimiser step through torch.optim.Adam moves the
    parameters the module API holds."""
    from cgs_amd import nets
    g = golden("g8_unet_train.npz")
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd/")}
    net = nets.Unet(upsample=False).to("cuda").train()
    net.load_state_dict(sd)
    X = orc.u8_to_nchw(g["X"]).cuda()
    tgt, ytc = torch.from_numpy(g["target_mask"]).cuda(), torch.from_numpy(g["target_critic"]).cuda()
    opt = torch.optim.Adam(net.parameters())
    y = net(X)
    c = net(X, critic=True).squeeze()
    lmask, lcrit = F.mse_loss(y, tgt), F.mse_loss(c, ytc)
    np.testing.assert_allclose(float(lmask), float(g["loss_mask"]), rtol=1e-5)
    np.testing.assert_allclose(float(lcrit), float(g["loss_critic"]), rtol=1e-5)
    (lmask + lcrit).backward()
    names = [k for k, _ in net.named_parameters()]
    assert len(names) == 24
    for k, q in net.named_parameters():
        assert q.grad is not None, k
        rel_close(q.grad.cpu().numpy(), g["grad/" + k], "grad " + k, rtol=1e-3, atol_scale=5e-5)
    before = {k: q.detach().clone() for k, q in net.named_parameters()}
    opt.step()
    assert all(not torch.equal(before[k], q.detach()) for k, q in net.named_parameters())
    # eval-mode forward still matches after the step's parameters are in place (no stale weight copies)
    with torch.no_grad():
        y2 = net.eval()(X)
    assert torch.isfinite(y2).all() and not torch.equal(y2, y.detach())


@pytest.mark.parametrize("n", [5, 16])
def test_config5_128x128_bf16_forward_vs_build_restatement(n):
    """BASELINE config 5: the build-defined six-stage 128x128 Hourglass on the bf16 kernels (hourglass128.py, csrc/gen_f16.hip with
    bf16 elements) against the build's own fp32 CPU restatement (oracle.hourglass128_apply).  There is no reference counterpart
    (nets.py:184,189-190 cannot take 128x128): PARITY UNPINNED, a self-consistency check with the bf16 tolerance stated here:
    |dZ| < 6e-4 max, < 1e-4 mean; |dpred| < 1e-4 -- three times the measured 2.0e-4 / 3.5e-5 / 2.7e-5 (bf16 has 8 significant bits;
    the masks of these weights lie in [0.43, 0.51])."""
    from cgs_amd import hourglass128
    pc = orc.seeded_params(orc.critic128_shapes(), 31)
    pm = orc.seeded_params(orc.masker128_shapes(), 32)
    rs = np.random.RandomState(n)
    x = rs.randint(0, 256, (n, 128, 128, 3)).astype(np.uint8)
    x[0, 40:90, 20:100] = 180           # a flat patch
    net = hourglass128.Hourglass128(pc, pm)
    pred, Z = net.infer(torch.from_numpy(x).cuda())
    with torch.no_grad():
        rp, rz = orc.hourglass128_apply(pc, pm, torch.from_numpy(x).permute(0, 3, 1, 2).float() / 255.0)
    dz = np.abs(Z.cpu().numpy().astype(np.float64) - rz[:, 0].numpy())
    dp = np.abs(pred.cpu().numpy().astype(np.float64) - rp[:, 0].numpy())
    print(f"config 5 (128x128, bf16), n={n}: |dZ| max {dz.max():.2e} p99.9 {np.quantile(dz, 0.999):.2e} mean {dz.mean():.2e}; |dpred| max {dp.max():.2e}; "
          f"Z range [{float(rz.min()):.3f}, {float(rz.max()):.3f}]")
    assert dz.max() < 6e-4 and dz.mean() < 1e-4
    assert dp.max() < 1e-4
    assert Z.shape == (n, 128, 128) and pred.shape == (n,)


def test_fused_fp16_convolutions_vs_float64_conv_on_fp16_operands(g1):
    """The three fp16 convolutions of the fused inference path (csrc/hconv.hip, v_mfma_f32_16x16x32_f16) one by one against float64
    conv2d on the SAME fp16-rounded inputs and weights: only the fp32 accumulation order and the output rounding differ
    (fp16 outputs: 2^-11 relative; the fp32 output of features.3: 1e-5 of the tensor's maximum)."""
    import ctypes as C
    import torch.nn.functional as F
    from cgs_amd import _lib
    pc, pm = g1
    dev = torch.device("cuda:0")
    P = lambda t: C.c_void_p(t.data_ptr())
    S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    h = lambda t: t.to(torch.float16).double()
    n = 5
    rs = np.random.RandomState(3)
    x = torch.from_numpy(rs.randint(0, 256, (n, 64, 64, 3)).astype(np.uint8))
    x[1, 10:30, 5:50] = 77                      # a flat patch
    hwio = lambda w: w.permute(2, 3, 1, 0).contiguous().to(dev)
    # features.0
    w0, b0 = pc["features.0.weight"], pc["features.0.bias"]
    w0g, b0g, xg = hwio(w0), b0.to(dev), x.to(dev)
    e0 = torch.empty((n, 32, 32, 8), device=dev, dtype=torch.float16)
    _lib.call("cgs_f16_enc0_fwd", n, P(xg), P(w0g), P(b0g), P(e0), S())
    ref0 = F.max_pool2d(F.relu(F.conv2d(h(x.double().permute(0, 3, 1, 2) / 255.0), h(w0), b0.double(), padding=1)), 2).permute(0, 2, 3, 1)
    torch.cuda.synchronize()
    err = (e0.double().cpu() - ref0).abs().max().item() / ref0.abs().max().item()
    print(f"features.0 fp16: max err / max = {err:.2e}")
    assert err < 1e-3
    # features.3 on that e0 (fp32 output)
    w1, b1 = pc["features.3.weight"], pc["features.3.bias"]
    w1g, b1g = hwio(w1), b1.to(dev)
    e1 = torch.empty((n, 16, 16, 8), device=dev, dtype=torch.float32)
    _lib.call("cgs_f16_enc1_fwd", n, P(e0), P(w1g), P(b1g), P(e1), S())
    e0c = e0.double().cpu().permute(0, 3, 1, 2)
    ref1 = F.max_pool2d(F.relu(F.conv2d(e0c, h(w1), b1.double(), padding=1)), 2).permute(0, 2, 3, 1)
    torch.cuda.synchronize()
    err = (e1.double().cpu() - ref1).abs().max().item() / ref1.abs().max().item()
    print(f"features.3 fp16 operands, fp32 out: max err / max = {err:.2e}")
    assert err < 1e-5
    # dec_model.0 on cat(e0, up(o1)): linear, fp16 output
    wd, bd = pm["dec_model.0.weight"], pm["dec_model.0.bias"]
    wdg, bdg = hwio(wd), bd.to(dev)
    o1 = torch.from_numpy(rs.randn(n, 16, 16, 8).astype(np.float32)).to(dev)
    o0 = torch.empty((n, 32, 32, 8), device=dev, dtype=torch.float16)
    _lib.call("cgs_f16_dec0_fwd", n, P(e0), P(o1), P(wdg), P(bdg), P(o0), S())
    cat = torch.cat((e0c, F.interpolate(h(o1.cpu()).permute(0, 3, 1, 2), scale_factor=2, mode="nearest")), 1)
    refd = F.conv2d(cat, h(wd), bd.double(), padding=1).permute(0, 2, 3, 1)
    torch.cuda.synchronize()
    err = (o0.double().cpu() - refd).abs().max().item() / refd.abs().max().item()
    print(f"dec_model.0 fp16: max err / max = {err:.2e}")
    assert err < 1e-3
