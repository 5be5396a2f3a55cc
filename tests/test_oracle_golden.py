"""Pins the CPU oracle (oracle/hourglass_ref.py) to fixtures captured from the reference's own
nets.py classes (tests/golden/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import hourglass_ref as orc

torch.set_num_threads(4)
TOL = dict(rtol=1e-5, atol=1e-6)


def t(a):
    return torch.from_numpy(np.asarray(a))


def split(raw, prefix):
    return {k[len(prefix) + 1:]: t(v) for k, v in raw.items() if k.startswith(prefix + "/")}


def test_key_inventory_matches_reference():
    with open(os.path.join(os.path.dirname(__file__), "golden", "g1_keys.json")) as fp:
        keys = json.load(fp)
    for cf in (1, 5):
        assert {k: list(s) for k, s in orc.critic_shapes(cf)} == keys[f"chfak{cf}"]["critic"]
        assert {k: list(s) for k, s in orc.masker_shapes(cf)} == keys[f"chfak{cf}"]["masker"]
    # parameter counts quoted in SURVEY.md section 8a
    assert sum(int(np.prod(s)) for _, s in orc.critic_shapes(1)) == 11873
    assert sum(int(np.prod(s)) for _, s in orc.masker_shapes(1)) == 13785
    assert sum(int(np.prod(s)) for _, s in orc.critic_shapes(5)) == 289761
    assert sum(int(np.prod(s)) for _, s in orc.masker_shapes(5)) == 305913


def test_eval_forward_matches_reference(golden, g1):
    pc, pm = g1
    g = golden("g2_eval.npz")
    X = orc.u8_to_nchw(g["X"])
    with torch.no_grad():
        pred, embeds = orc.critic_apply(pc, X, collect=True)
        Z, inter = orc.masker_apply(pm, X, embeds, return_all=True)
    np.testing.assert_allclose(pred.numpy(), g["pred"], **TOL)
    for i in range(5):
        np.testing.assert_allclose(embeds[i].numpy(), g[f"e{i}"], **TOL)
    for i in range(5):
        np.testing.assert_allclose(inter[f"o{i}"].numpy(), g[f"o{i}"], **TOL)
    np.testing.assert_allclose(inter["hm"][:2].numpy(), g["hm"], **TOL)
    np.testing.assert_allclose(Z.numpy(), g["Z"], **TOL)


def test_eval_forward_chfak5(golden):
    g = golden("g2_eval_chfak5.npz")
    pc = orc.seeded_params(orc.critic_shapes(5), 11)
    pm = orc.seeded_params(orc.masker_shapes(5), 12)
    X = orc.u8_to_nchw(g["X"])
    with torch.no_grad():
        pred, embeds = orc.critic_apply(pc, X, collect=True)
        Z = orc.masker_apply(pm, X, embeds)
    np.testing.assert_allclose(pred.numpy(), g["pred"], **TOL)
    np.testing.assert_allclose(embeds[4].numpy(), g["e4"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(Z.numpy(), g["Z"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("tag,kw", [
    ("g3_train_default", dict()),
    ("g3_train_noinject", dict(inject=False)),
    ("g3_train_frozen", dict(live=False)),
    ("g3_train_l2", dict(L2=0.1)),
    ("g3_train_bce", dict(threshrew=0.5)),
    ("g3_train_valuefak", dict(staticnorm=False, L2=0.1)),       # -staticnorm '': regulariser weighted by 1 - pred (main.py:415-418)
])
def test_phase2_steps_match_reference(golden, g1, tag, kw):
    pc, pm = g1
    g = golden(tag + ".npz")
    A, B, Y = orc.u8_to_nchw(g["A"]), orc.u8_to_nchw(g["B"]), t(g["Y"])
    recs = orc.train_phase2(pc, pm, [(A, B, Y)], steps=3, **kw)
    order = ["critic", "replace", "inject", "norm", "norm2"]
    for s in range(3):
        assert recs[s]["total"] == pytest.approx(float(g[f"total{s}"]), rel=2e-5)
        for i, name in enumerate(order):
            assert recs[s]["parts"].get(name, 0.0) == pytest.approx(float(g[f"parts{s}"][i]), rel=2e-5, abs=1e-8)
    np.testing.assert_allclose(recs[0]["Z"].numpy(), g["Z0"], **TOL)
    np.testing.assert_allclose(recs[0]["pred"].numpy(), g["pred0"], **TOL)
    for which, gk in (("grads_c", "grad/critic"), ("grads_m", "grad/masker")):
        ref = split(g, gk)
        assert len(ref) == 14
        for k, v in ref.items():
            got = recs[0][which][k]
            np.testing.assert_allclose(got.numpy(), v.numpy(), rtol=2e-4, atol=2e-6, err_msg=k)
    for s in (1, 3):
        for which, gk in (("params_c", f"step{s}/critic"), ("params_m", f"step{s}/masker")):
            for k, v in split(g, gk).items():
                np.testing.assert_allclose(recs[s - 1][which][k].numpy(), v.numpy(), rtol=1e-4, atol=2e-6, err_msg=f"{s}:{k}")


def test_phase2_steps_match_reference_at_chfak2(golden):
    """A model size the specialised kernels do not cover (chfak = 2: 16/16/16/32 channels, neck 64) -- the pin of the oracle for
    the shape-generic kernels' training pass."""
    g = golden("g3_train_chfak2.npz")
    pc, pm = orc.seeded_params(orc.critic_shapes(2), 21), orc.seeded_params(orc.masker_shapes(2), 22)
    A, B, Y = orc.u8_to_nchw(g["A"]), orc.u8_to_nchw(g["B"]), t(g["Y"])
    recs = orc.train_phase2(pc, pm, [(A, B, Y)], steps=2)
    for s in range(2):
        assert recs[s]["total"] == pytest.approx(float(g[f"total{s}"]), rel=2e-5)
    np.testing.assert_allclose(recs[0]["Z"].numpy(), g["Z0"], **TOL)
    for which, gk in (("grads_c", "grad/critic"), ("grads_m", "grad/masker")):
        ref = split(g, gk)
        assert len(ref) == 14
        for k, v in ref.items():
            np.testing.assert_allclose(recs[0][which][k].numpy(), v.numpy(), rtol=2e-4, atol=2e-6, err_msg=k)
    for which, gk in (("params_c", "step2/critic"), ("params_m", "step2/masker")):
        for k, v in split(g, gk).items():
            np.testing.assert_allclose(recs[1][which][k].numpy(), v.numpy(), rtol=1e-4, atol=2e-6, err_msg=k)


@pytest.mark.parametrize("tag,live", [("g3_train_separate", True), ("g3_train_separate_frozen", False)])
def test_phase2_separate_critic_matches_reference(golden, g1, tag, live):
    """-separate (main.py:110-111, 389-390): a second critic's embeds feed the masker; it joins the optimiser group."""
    pc, pm = g1
    g = golden(tag + ".npz")
    ps = {k: v for k, v in split(g, "sepcrit0").items()}
    A, B, Y = orc.u8_to_nchw(g["A"]), orc.u8_to_nchw(g["B"]), t(g["Y"])
    recs = orc.train_phase2(pc, pm, [(A, B, Y)], steps=3, live=live, Ps=ps)
    for s in range(3):
        assert recs[s]["total"] == pytest.approx(float(g[f"total{s}"]), rel=2e-5)
    for which, gk in (("grads_m", "grad/masker"), ("grads_s", "grad/sepcrit")) + ((("grads_c", "grad/critic"),) if live else ()):
        for k, v in split(g, gk).items():
            np.testing.assert_allclose(recs[0][which][k].numpy(), v.numpy(), rtol=2e-4, atol=2e-6, err_msg=f"{gk}/{k}")
    for k, v in split(g, "step3/sepcrit").items():
        np.testing.assert_allclose(recs[2]["params_s"][k].numpy(), v.numpy(), rtol=1e-4, atol=2e-6, err_msg=k)
    for k, v in split(g, "step3/masker").items():
        np.testing.assert_allclose(recs[2]["params_m"][k].numpy(), v.numpy(), rtol=1e-4, atol=2e-6, err_msg=k)


def test_frozen_leaves_critic_untouched(golden, g1):
    pc, pm = g1
    g = golden("g3_train_frozen.npz")
    for k, v in split(g, "step3/critic").items():
        np.testing.assert_array_equal(v.numpy(), pc[k].numpy())


@pytest.mark.parametrize("tag,thr", [("g4_phase1_mse", 0.0), ("g4_phase1_bce", 0.5)])
def test_phase1_step_matches_reference(golden, g1, tag, thr):
    pc, _ = g1
    g = golden(tag + ".npz")
    recs = orc.train_phase1(pc, [(orc.u8_to_nchw(g["X"]), t(g["Y"]))], steps=1, threshrew=thr)
    assert recs[0]["loss"] == pytest.approx(float(g["loss"]), rel=1e-5)
    np.testing.assert_allclose(recs[0]["pred"].numpy(), g["pred"], **TOL)
    for k, v in split(g, "grad").items():
        np.testing.assert_allclose(recs[0]["grads"][k].numpy(), v.numpy(), rtol=2e-4, atol=2e-6, err_msg=k)
    for k, v in split(g, "step1").items():
        np.testing.assert_allclose(recs[0]["params"][k].numpy(), v.numpy(), rtol=1e-4, atol=2e-6, err_msg=k)


def test_shift_batch_matches_reference(golden):
    g = golden("g5_shift.npz")
    X = t(g["X"])
    for k in range(4):
        torch.manual_seed(k)
        rolled, amount, left = orc.shift_batch(X, 12)
        assert amount == int(g[f"amount{k}"]) and left == bool(g[f"left{k}"])
        np.testing.assert_array_equal(rolled.numpy(), g[f"rolled{k}"])
        # property: a roll is a permutation of columns
        np.testing.assert_array_equal(np.sort(rolled.numpy(), axis=2), np.sort(X.numpy(), axis=2))


def test_dropout_step_with_recorded_masks(golden, g1):
    """Train-mode p=0.3 step: feeding the reference's recorded keep-masks reproduces its losses/grads."""
    pc, pm = g1
    g = golden("g7_dropout.npz")
    masks = [[t(g[f"mask{p * 3 + s:02d}"]).float() for s in range(3)] for p in range(4)]
    A, B, Y = orc.u8_to_nchw(g["A"]), orc.u8_to_nchw(g["B"]), t(g["Y"])
    recs = orc.train_phase2(pc, pm, [(A, B, Y)], steps=1, p=0.3, training=True, masks=masks)
    assert recs[0]["total"] == pytest.approx(float(g["total0"]), rel=2e-5)
    for k, v in split(g, "grad/critic").items():
        np.testing.assert_allclose(recs[0]["grads_c"][k].numpy(), v.numpy(), rtol=2e-4, atol=2e-6, err_msg=k)
    for k, v in split(g, "grad/masker").items():
        np.testing.assert_allclose(recs[0]["grads_m"][k].numpy(), v.numpy(), rtol=2e-4, atol=2e-6, err_msg=k)


def test_adamref_equals_torch_adam():
    torch.manual_seed(0)
    p0 = [torch.randn(7, 5), torch.randn(11)]
    a = [q.clone() for q in p0]
    b = [q.clone().requires_grad_(True) for q in p0]
    mine = orc.AdamRef(a)
    theirs = torch.optim.Adam(b)
    for s in range(5):
        grads = [torch.randn_like(q) * (10.0 ** (s - 2)) for q in p0]
        mine.step(grads)
        for q, gq in zip(b, grads):
            q.grad = gq.clone()
        theirs.step()
    for x, y in zip(a, b):
        np.testing.assert_allclose(x.numpy(), y.detach().numpy(), rtol=1e-6, atol=1e-7)


def test_postprocess_shapes(golden, g1):
    pc, pm = g1
    g = golden("g2_eval.npz")
    X01 = g["X"][:3] / 255.0
    preds, M = orc.infer_masks(pc, pm, X01, batchsize=2)
    np.testing.assert_allclose(M, g["Z"][:3], **TOL)
    np.testing.assert_allclose(preds, g["pred"][:3, 0], **TOL)
    hard, stack = orc.postprocess_masks(X01, M, 0.5)
    assert hard.dtype == np.bool_ and hard.shape == (3, 1, 64, 64)
    assert stack.dtype == np.uint8 and stack.shape == (3, 3, 64, 64, 3)
    np.testing.assert_array_equal(stack[:, 0], g["X"][:3])
    assert set(np.unique(stack[:, 2])) <= {0, 255}


def test_eval_iou_restatement(g1):
    """eval_iou = main.py:891-1020 (plain branch) + get_iou main.py:1265-1270: subset 100:5000:2, strict threshold,
    |A&B| / |A|B| over the whole set, 3 digits."""
    pc, pm = g1
    rs = np.random.RandomState(5)
    X = rs.randint(0, 256, (140, 64, 64, 3)).astype(np.uint8)
    Y = rs.rand(140, 64, 64, 3) > 0.2
    _, M = orc.infer_masks(pc, pm, X[100:5000:2] / 255.0)
    thr = float(np.median(M))
    A = (M > thr).squeeze()
    B = np.all(Y, axis=-1)[100:5000:2]
    assert orc.eval_iou(pc, pm, X, Y, eval_thresh=thr) == round(float((A & B).sum() / (A | B).sum()), 3)
    assert orc.eval_iou(pc, pm, X, Y, eval_thresh=2.0) == 0.0      # nothing above 2: empty prediction


def test_saliency_restatement_shapes_and_clipping(g1):
    """eval_saliency_iou: maps are clipped at 1 and weighted by pred; with the reference's default threshold 1.5 nothing
    can exceed it (the reference's saliency IoU is then 0): restated literally."""
    pc, _ = g1
    rs = np.random.RandomState(6)
    X = rs.randint(0, 256, (124, 64, 64, 3)).astype(np.uint8)
    Y = rs.rand(124, 64, 64, 3) > 0.1
    iou, salM, raw = orc.eval_saliency_iou(pc, X, Y)
    assert salM.shape == (12, 1, 64, 64) and raw.shape == salM.shape
    assert salM.max() <= 1.0 and raw.min() >= 0.0 and iou == 0.0
    iou2, _, _ = orc.eval_saliency_iou(pc, X, Y, salience_thresh=0.25)
    assert 0.0 <= iou2 <= 1.0


def test_oracle_unet_restatement_matches_reference_capture(golden):
    g = golden("g8_unet_convt.npz")
    P = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd/")}
    with torch.no_grad():
        y, u0 = orc.unet_convt_apply(P, orc.u8_to_nchw(g["X"]))
        c = orc.unet_convt_apply(P, orc.u8_to_nchw(g["X"]), critic=True)
    np.testing.assert_allclose(y.numpy(), g["y"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(u0.numpy(), g["u0"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(c.numpy(), g["critic"], rtol=1e-5, atol=1e-6)


def test_g6_process_writer_matches_the_reference_cli(golden, g1):
    """G6: the reference's OWN `main.py -process` was run end to end in the build container
    (tests/golden/make_golden_g6.py) and its PNG outputs captured.  The oracle's hand restatement of the post-processing
    (infer_masks + postprocess_masks, main.py:1130-1223) reproduces every file bit for bit: raw masks, thresholded masks at 0.5 and
    0.52, the -concatenated strips; the file names keep everything before the LAST dot of the source name; the checkpoint names are
    the ones the reference's Handler computed for the default flags (main.py:86-104)."""
    import json
    pc, pm = g1
    g = golden("g6_process.npz")
    X, names = g["frames"], [str(s) for s in g["names"]]
    listing = json.loads(str(g["listing_json"]))
    _, M = orc.infer_masks(pc, pm, X / 255.0)
    for tag, thr in (("default", 0.5), ("thr052", 0.52)):
        hard, stack = orc.postprocess_masks(X / 255.0, M, thr)
        assert listing[tag] == sorted(f"{nm}-{c}.png" for nm in names for c in ("raw-mask", "thresholded-mask"))
        for i, nm in enumerate(names):
            np.testing.assert_array_equal(g[f"{tag}/{nm}-raw-mask.png"], stack[i, 1])
            np.testing.assert_array_equal(g[f"{tag}/{nm}-thresholded-mask.png"], stack[i, 2])
    hard, stack = orc.postprocess_masks(X / 255.0, M, 0.5)
    assert listing["concat"] == sorted(f"{nm}_with_mask.png" for nm in names)
    for i, nm in enumerate(names):
        np.testing.assert_array_equal(g[f"concat/{nm}_with_mask.png"], np.concatenate(list(stack[i]), axis=-2))
    # checkpoint naming contract, pinned by the reference's Handler itself
    from cgs_amd import cli, handler
    cn, mn = handler.checkpoint_names(cli.parse_args(["--model", "m"]))
    assert [f"m/saves/critic-{cn}.pt", f"m/saves/masker-{mn}.pt"] == [str(s) for s in g["checkpoint_names"]]
