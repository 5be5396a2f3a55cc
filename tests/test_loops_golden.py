"""CPU tests of the loop-level captures G9 / G10 / G11 (tests/golden/make_golden_loops.py: the reference's own main.main() run on synthetic
data with recorders around its library calls): the oracle's hand restatements of the LOOPS -- the two training loops replayed step by step,
Handler.eval's IoU / saliency post-processing, collect_data's labelling -- against what the reference itself did."""
import os
import sys

import numpy as np
import pytest
import torch

import cgs_amd  # noqa: F401
from cgs_amd import dataformat as df
from oracle import hourglass_ref as orc

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import loop_inputs as li  # noqa: E402


def load(name):
    return dict(np.load(os.path.join(HERE, "golden", name), allow_pickle=False))


def roll_from_draws(r1, r2, shift=12):
    """main.py:584-591 as a signed torch.roll along the width: the first draw is the amount, the second the direction."""
    amount = int(shift * r1)
    return -amount if r2 > 0.5 else amount


# ------------------------------------------------------------------------------------------------ G11
@pytest.mark.parametrize("tag,size,gammas", [("trunk", 800, (0.98, 0.97, 0.96, 0.95)), ("trunk_small", 200, (0.9, 0.5))])
def test_g11_build_dataset_equals_the_reference_collect_data(tag, size, gammas):
    """dataformat.build_dataset on the same synthetic episodes == the gz-pickle the reference's Handler.collect_data wrote under a stub
    minerl (trunk filter main.py:1325, 0/1 rewards, clipped discounted rows main.py:1336-1346, per-episode frame counter I): bit-exact."""
    g = load("g11_collect.npz")
    eps = [(pov, rew) for _name, pov, rew in li.synthetic_episodes()]
    X, Y, I = df.build_dataset(eps, size, mode="trunk", gammas=gammas)
    assert [str(X.dtype), str(Y.dtype), str(I.dtype)] == g[f"{tag}/pickle_dtypes"].tolist()
    np.testing.assert_array_equal(Y, g[f"{tag}/pickle_Y"])
    np.testing.assert_array_equal(I, g[f"{tag}/pickle_I"])
    np.testing.assert_array_equal(X.reshape(len(X), -1).sum(1).astype(np.int64), g[f"{tag}/pickle_X_rowsum"])
    # the file name the reference derived from its flags
    datasize = {"trunk": 600, "trunk_small": 150}[tag]
    assert os.path.basename(df.dataset_path(datasize=datasize, gammas="-".join(str(x) for x in gammas))) == str(g[f"{tag}/file"])
    # what collect_data RETURNS after a fresh collection: the full-size buffers (zero rows when the episodes ran out, main.py:1293-1295,1359)
    assert g[f"{tag}/ret_shapes"].tolist() == [size, size, size]
    np.testing.assert_array_equal(g[f"{tag}/ret_Y"][:, :len(X)], Y)
    assert not g[f"{tag}/ret_Y"][:, len(X):].any() and not g[f"{tag}/ret_I"][len(X):].any()


# ------------------------------------------------------------------------------------------------ G10
def _g1():
    raw = load("g1_weights_chfak1.npz")
    pc = {k.split("/", 1)[1]: torch.from_numpy(v) for k, v in raw.items() if k.startswith("critic/")}
    pm = {k.split("/", 1)[1]: torch.from_numpy(v) for k, v in raw.items() if k.startswith("masker/")}
    return pc, pm


def test_g10_oracle_eval_iou_and_saliency_equal_the_reference_eval():
    """oracle.eval_iou / eval_saliency_iou (hand restatements of main.py:891-1016) against Handler.eval of the reference on the same
    synthetic red-trees set: the IoUs it returned, the hard masks and ground truth it handed to get_iou, its raw saliency maps."""
    g = load("g10_eval.npz")
    pc, pm = _g1()
    X, Yrgb = li.synthetic_eval_set(int(g["n_frames"]), int(g["data_seed"]))
    truth = np.unpackbits(g["plain/gt"]).astype(bool)
    np.testing.assert_array_equal(Yrgb[100:5000:2].all(-1).reshape(-1), truth[:Yrgb[100:5000:2].all(-1).size])
    for tag, thr in (("plain", 0.05), ("thr02", 0.2)):
        assert orc.eval_iou(pc, pm, X, Yrgb, eval_thresh=thr) == float(g[f"{tag}/ious"][0])
    for tag, kw in (("sal_global", dict(salience_thresh=1.5, salglobal=True)), ("sal_k", dict(salience_thresh=0.9, salglobal=False))):
        iou, _maps, raw = orc.eval_saliency_iou(pc, X, Yrgb, **kw)
        want = float(g[f"{tag}/ious"][1])
        assert iou == want or (np.isnan(iou) and want == 0.0), (tag, iou, want)
    raw40 = g["sal_k/sal_raw_first40"]
    np.testing.assert_allclose(raw[:40, 0], raw40, rtol=1e-5, atol=1e-9)


# ------------------------------------------------------------------------------------------------ G9
def test_g9_rng_consumption_order_of_the_two_loops():
    """The draws the reference's loops took from the global RNGs, re-drawn here in the order the build's Handler takes them: phase 2 =
    per step choice(npos, 32), choice(nneg, 32), choice(nneg, 64) from numpy, then rand(1) x 2 from torch (main.py:310-312, 585-586);
    phase 1 = the DataLoader's permutation, then rand(1) x 2 per batch."""
    g = load("g9_train_loop.npz")
    npos, nneg = int(g["npos"]), int(g["nneg"])
    s1, s2 = (int(v) for v in g["seeds"])
    np.random.seed(s2)
    torch.manual_seed(s2)
    for step in range(len(g["p2_choice"])):
        got = np.concatenate((np.random.choice(np.arange(npos), 32), np.random.choice(np.arange(nneg), 32), np.random.choice(np.arange(nneg), 64)))
        np.testing.assert_array_equal(got, g["p2_choice"][step])
        assert (float(torch.rand(1)), float(torch.rand(1))) == tuple(g["p2_shift_draws"][step])
    torch.manual_seed(s1)
    order = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(torch.arange(li.DATASIZE)), batch_size=64, shuffle=True)
    pos = 0
    for b, (idx,) in enumerate(order):
        n = int(g["p1_batch_len"][b])
        np.testing.assert_array_equal(idx.numpy(), g["p1_batch_idx"][pos:pos + n])
        pos += n
        assert (float(torch.rand(1)), float(torch.rand(1))) == tuple(g["p1_shift_draws"][b])


def test_g9_oracle_replays_the_reference_training_loops():
    """The oracle's step functions driven by the RECORDED batches / index draws / shift draws reproduce the reference's run of
    `main.py -train` (critic_pipe, the contrastive sweep + split, segmentation_training): every step's loss values, the critic after
    phase 1, the sweep values, the split sizes, the final masker and critic."""
    g = load("g9_train_loop.npz")
    pc, pm = _g1()
    X, Y, _I = li.synthetic_frames(li.DATASIZE + li.TESTSIZE, int(g["data_seed"]))
    X, Y = X[:li.DATASIZE], Y[:, :li.DATASIZE]
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    # ---- phase 1
    Pc = orc.leafify(pc)
    opt = orc.AdamRef(list(Pc.values()))
    pos = 0
    for b in range(len(g["p1_batch_len"])):
        n = int(g["p1_batch_len"][b])
        idx = g["p1_batch_idx"][pos:pos + n]
        pos += n
        xb = torch.roll(torch.from_numpy(X[idx]), roll_from_draws(*g["p1_shift_draws"][b]), dims=2)
        for t in Pc.values():
            t.grad = None
        loss, _pred = orc.phase1_loss(Pc, orc.u8_to_nchw(xb.numpy()), torch.from_numpy(Y[1, idx]).float())
        loss.backward()
        opt.step([t.grad for t in Pc.values()])
        assert abs(float(loss.detach()) - g["p1_loss"][b]) <= 2e-5 * max(1.0, abs(g["p1_loss"][b])), (b, float(loss), g["p1_loss"][b])
    for k, v in Pc.items():
        np.testing.assert_allclose(v.detach().numpy(), g["critic_after_p1/" + k], rtol=0, atol=1e-5, err_msg=k)      # measured 7e-7
    # ---- sweep + split (main.py:245-281)
    pc1 = {k: v.detach() for k, v in Pc.items()}
    with torch.no_grad():
        preds = torch.cat([orc.critic_apply(pc1, orc.u8_to_nchw(X[b:b + 128])).squeeze(1) for b in range(0, len(X), 128)]).numpy()
    np.testing.assert_allclose(preds, g["sweep_preds"], rtol=0, atol=1e-5)       # measured 7e-7; the thresholds sit in gaps >= 1.3e-3
    hi, lo = float(g["high_thresh"]), float(g["low_thresh"])
    positives, negatives = preds > hi, preds < lo
    assert (int(positives.sum()), int(negatives.sum())) == (int(g["npos"]), int(g["nneg"]))
    import json
    assert json.loads(str(g["side_files_json"])) == [f"{int(positives.sum())}>{hi}__{int(negatives.sum())}<{lo}.txt"]
    Xpos, Ypos, Xneg, Yneg = X[positives], Y[:, positives], X[negatives], Y[:, negatives]
    # ---- phase 2
    Pm = orc.leafify(pm)
    tensors = list(Pc.values()) + list(Pm.values())
    opt = orc.AdamRef(tensors)
    for s in range(len(g["p2_choice"])):
        Hi, Li, Ci = g["p2_choice"][s][:32], g["p2_choice"][s][32:64], g["p2_choice"][s][64:]
        A8 = torch.roll(torch.from_numpy(np.concatenate((Xpos[Hi], Xneg[Li]))), roll_from_draws(*g["p2_shift_draws"][s]), dims=2)
        Yb = torch.from_numpy(np.concatenate((Ypos[1, Hi], Yneg[1, Li]))).float()
        for t in tensors:
            t.grad = None
        total, parts, _Z, _pred = orc.phase2_loss(Pc, Pm, orc.u8_to_nchw(A8.numpy()), orc.u8_to_nchw(Xneg[Ci]), Yb)
        total.backward()
        opt.step([t.grad for t in tensors])
        want = g["p2_loss_critic_replace_inject"][s]
        got = [float(parts[k]) for k in ("critic", "replace", "inject")]
        np.testing.assert_allclose(got, want, rtol=1e-2, atol=1e-6, err_msg=f"step {s}")       # measured 4e-4 / 3e-4 / 4.4e-3 over the 34 steps
        assert abs(float(parts["norm"]) / 0.5 - g["p2_loss_l1_mean"][s]) <= 5e-3 * g["p2_loss_l1_mean"][s], s
    for k, v in Pm.items():
        np.testing.assert_allclose(v.detach().numpy(), g["masker_final/" + k], rtol=0, atol=2e-4, err_msg=k)          # measured 3.2e-5
    for k, v in Pc.items():
        np.testing.assert_allclose(v.detach().numpy(), g["critic_final/" + k], rtol=0, atol=5e-5, err_msg=k)          # measured 6e-6


def g12_states():
    """G12 (round 6): per phase-1 batch of G9's run the reference's critic parameters, Adam moments and step count BEFORE the step and the
    gradient the step produced -- rows of concatenated tensors in named_parameters() order, reference (OIHW) layout."""
    import json
    g = load("g12_phase1_steps.npz")
    keys, shapes = json.loads(str(g["keys_json"])), json.loads(str(g["shapes_json"]))

    def unflat(row):
        out, pos = {}, 0
        for k, shp in zip(keys, shapes):
            cnt = int(np.prod(shp))
            out[k] = torch.from_numpy(row[pos:pos + cnt].reshape(shp).copy())
            pos += cnt
        assert pos == len(row)
        return out
    return g, keys, unflat


def test_g12_oracle_single_steps_from_the_reference_states():
    """Phase 1 pinned STEP BY STEP: for each of the 48 batches of G9's run the oracle starts from the reference's
    recorded critic + Adam state, takes ONE step on the recorded batch, and reproduces the reference's loss, its gradient and its next
    state -- tight bounds on all 48 steps, because no trajectory accumulates (compare test_g9_phase1_trajectory_is_sensitive...)."""
    g9 = load("g9_train_loop.npz")
    g, keys, unflat = g12_states()
    import json
    assert keys == list(json.load(open(os.path.join(HERE, "golden", "g1_keys.json")))["chfak1"]["critic"])
    X, Y, _I = li.synthetic_frames(li.DATASIZE + li.TESTSIZE, int(g9["data_seed"]))
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    nb = len(g9["p1_batch_len"])
    assert g["params_before"].shape == (nb, 11873) and g["adam_step_before"].tolist() == list(range(nb))
    np.testing.assert_array_equal(g["loss"], g9["p1_loss"])
    np.testing.assert_array_equal(g["params_after_last"], np.concatenate([g9["critic_after_p1/" + k].reshape(-1) for k in keys]))
    assert not g["adam_m_before"][0].any() and not g["adam_v_before"][0].any()
    pos, worst = 0, [0.0, 0.0, 0.0]
    for b in range(nb):
        n = int(g9["p1_batch_len"][b])
        idx = g9["p1_batch_idx"][pos:pos + n]
        pos += n
        xb = torch.roll(torch.from_numpy(X[idx]), roll_from_draws(*g9["p1_shift_draws"][b]), dims=2)
        Pc = orc.leafify(unflat(g["params_before"][b]))
        opt = orc.AdamRef(list(Pc.values()))
        opt.m, opt.v, opt.t = list(unflat(g["adam_m_before"][b]).values()), list(unflat(g["adam_v_before"][b]).values()), int(g["adam_step_before"][b])
        loss, _ = orc.phase1_loss(Pc, orc.u8_to_nchw(xb.numpy()), torch.from_numpy(Y[1, idx]).float())
        loss.backward()
        grads = [t.grad for t in Pc.values()]
        opt.step(grads)
        want_g = unflat(g["grads"][b])
        want_p = unflat(g["params_before"][b + 1] if b + 1 < nb else g["params_after_last"])
        worst[0] = max(worst[0], abs(float(loss.detach()) - g["loss"][b]) / abs(g["loss"][b]))
        for k, gr in zip(keys, grads):
            worst[1] = max(worst[1], float((gr - want_g[k]).abs().max() / want_g[k].abs().max()))
            worst[2] = max(worst[2], float((Pc[k].detach() - want_p[k]).abs().max()))
    print(f"G12 oracle single steps: loss rel {worst[0]:.1e}, gradient (of each tensor's max) {worst[1]:.1e}, next parameters abs {worst[2]:.1e}")
    assert worst[0] <= 5e-6 and worst[1] <= 1e-4 and worst[2] <= 5e-6      # measured 0 / 1.1e-5 (thread count != the reference's 1) / 1e-7


def test_g9_phase1_trajectory_is_sensitive_to_gradient_rounding():
    """Why tests/test_gpu_loops.py bounds the END of the GPU's phase-1 trajectory loosely (loss 5e-2, parameters 6e-2) while its first steps
    agree to 1e-7: the oracle itself, replaying the reference's 48 recorded batches twice -- once as is, once with every gradient tensor
    perturbed by Gaussian noise of 3e-7 of its largest element (the size of an fp32 summation-order difference) -- ends 1e-3..1e-2 apart in
    the loss and in the parameters, because Adam's normalised step m / sqrt(v) turns a relative error on a near-zero gradient element into a
    full-size update.  The same replay with the WEIGHTS perturbed by 1e-7 stays together (6e-7): the spread is the optimiser's, not the net's."""
    g = load("g9_train_loop.npz")
    pc, _pm = _g1()
    X, Y, _I = li.synthetic_frames(li.DATASIZE + li.TESTSIZE, int(g["data_seed"]))
    torch.set_num_threads(min(8, os.cpu_count() or 1))

    def replay(noise):
        gen = torch.Generator().manual_seed(0)
        Pc = orc.leafify(pc)
        opt = orc.AdamRef(list(Pc.values()))
        pos, losses = 0, []
        for b in range(len(g["p1_batch_len"])):
            n = int(g["p1_batch_len"][b])
            idx = g["p1_batch_idx"][pos:pos + n]
            pos += n
            xb = torch.roll(torch.from_numpy(X[idx]), roll_from_draws(*g["p1_shift_draws"][b]), dims=2)
            for t in Pc.values():
                t.grad = None
            loss, _ = orc.phase1_loss(Pc, orc.u8_to_nchw(xb.numpy()), torch.from_numpy(Y[1, idx]).float())
            loss.backward()
            opt.step([t.grad + noise * t.grad.abs().max() * torch.randn(t.grad.shape, generator=gen) for t in Pc.values()])
            losses.append(float(loss.detach()))
        return np.array(losses), {k: v.detach().numpy() for k, v in Pc.items()}
    l0, p0 = replay(0.0)
    l1, p1 = replay(3e-7)
    np.testing.assert_allclose(l0, g["p1_loss"], rtol=2e-5)
    dev_loss = np.abs(l1 - l0) / np.abs(l0)
    dev_par = max(float(np.abs(p0[k] - p1[k]).max()) for k in p0)
    print(f"gradient noise 3e-7: loss deviation first 8 steps {dev_loss[:8].max():.1e}, all {dev_loss.max():.1e}; parameters {dev_par:.1e}")
    assert dev_loss[:8].max() <= 1e-5 and dev_loss.max() >= 1e-3 and dev_par >= 1e-3
