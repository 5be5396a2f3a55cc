"""GPU tests of the loop-level pins G9 / G10 / G11: THIS build's `main.py` command lines (cgs_amd.cli.main, in-process so that the same
two seeding hooks the generator put around the reference's Handler can be put around this one) against what the reference's own
`main.main()` did on the same synthetic data (tests/golden/make_golden_loops.py):

  G9  `-train`: the frame indices of every phase-1 batch, every numpy / torch RNG draw of both loops in order (bit-exact), the split sizes and
      the side-file name (exact), every step's loss values, both checkpoints (within the stated tolerances);
  G10 `-eval [-salience]`: the IoUs the reference printed, the hard masks it compared;
  G11 `Handler.collect_data` under the same stub `minerl`: the gz-pickle and the returned arrays (bit-exact)."""
import gzip
import json
import os
import pickle
import sys
import types

import numpy as np
import pytest
import torch

import cgs_amd  # noqa: F401
from oracle import hourglass_ref as orc      # checker only

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import loop_inputs as li  # noqa: E402


def load(name):
    return dict(np.load(os.path.join(HERE, "golden", name), allow_pickle=False))


def g1():
    raw = load("g1_weights_chfak1.npz")
    pc = {k.split("/", 1)[1]: torch.from_numpy(v) for k, v in raw.items() if k.startswith("critic/")}
    pm = {k.split("/", 1)[1]: torch.from_numpy(v) for k, v in raw.items() if k.startswith("masker/")}
    return pc, pm


class DrawRecorder:
    """np.random.choice / torch.rand(1) wrapped exactly as the generator wrapped them around the reference."""

    def __init__(self, monkeypatch):
        self.choice, self.rand = [], []
        real_choice, real_rand = np.random.choice, torch.rand

        def choice(a, size=None, *args, **kw):
            r = real_choice(a, size, *args, **kw)
            self.choice.append(np.asarray(r).copy())
            return r

        def rand(*size, **kw):
            r = real_rand(*size, **kw)
            if tuple(size) == (1,):
                self.rand.append(float(r))
            return r
        monkeypatch.setattr(np.random, "choice", choice)
        monkeypatch.setattr(torch, "rand", rand)


def test_g9_main_train_reproduces_the_reference_run(tmp_path, monkeypatch):
    from cgs_amd import cli, handler
    g = load("g9_train_loop.npz")
    pc, pm = g1()
    s1, s2 = (int(v) for v in g["seeds"])
    X, Y, I = li.synthetic_frames(li.DATASIZE + li.TESTSIZE, int(g["data_seed"]))
    monkeypatch.chdir(tmp_path)
    os.makedirs("runs/data/straight")
    with gzip.GzipFile(f"runs/data/straight/Treechop-trunk-{li.DATASIZE}-[0.98-0.97-0.96-0.95].pickle", "wb") as fp:
        pickle.dump((X, Y, I), fp)
    real_cp, real_seg = handler.Handler.critic_pipe, handler.Handler.segmentation_training

    def critic_pipe(self, mode="train", test=0):       # the generator's hook: G1 weights, seeds, then the loop
        self.critic.load_state_dict(pc)
        self.masker.load_state_dict(pm)
        self.start_trace()
        np.random.seed(s1)
        torch.manual_seed(s1)
        return real_cp(self, mode, test)

    def segmentation_training(self):
        np.random.seed(s2)
        torch.manual_seed(s2)
        return real_seg(self)
    monkeypatch.setattr(handler.Handler, "critic_pipe", critic_pipe)
    monkeypatch.setattr(handler.Handler, "segmentation_training", segmentation_training)
    rec = DrawRecorder(monkeypatch)
    argv = json.loads(str(g["argv_json"]))
    nb1, ns2 = len(g["p1_batch_len"]), len(g["p2_choice"])
    # ================= run A: the whole command line (critic_pipe -> split -> segmentation_training), checked up to the end of phase 1
    H = cli.main(argv)
    torch.cuda.synchronize()
    tr = H._trace
    assert [len(b) for b in tr["p1_idx"]] == g["p1_batch_len"].tolist()
    np.testing.assert_array_equal(np.concatenate(tr["p1_idx"]), g["p1_batch_idx"])                      # the DataLoader's batches: bit-exact
    np.testing.assert_array_equal(np.array(rec.rand[:2 * nb1]).reshape(nb1, 2), g["p1_shift_draws"])    # the shift draws: bit-exact
    p1 = torch.cat(tr["p1_loss"]).cpu().numpy()
    e1 = np.abs(p1 - g["p1_loss"]) / np.abs(g["p1_loss"])
    ck_c = torch.load(H.save_paths["critic"], map_location="cpu")
    ec = max(float((ck_c[k] - torch.from_numpy(g["critic_after_p1/" + k])).abs().max()) for k in ck_c)
    print(f"G9 run A: phase-1 loss rel err: first 8 steps {e1[:8].max():.1e}, first 24 {e1[:24].max():.1e}, all 48 {e1.max():.1e}; critic after "
          f"phase 1 max abs {ec:.1e}; split {len(H.Xpos)} / {len(H.Xneg)} (reference {int(g['npos'])} / {int(g['nneg'])})")
    # SMOKE bounds only -- phase 1's parity statement is test_g12_phase1_single_steps_from_the_reference_states below (each of the 48
    # steps from the reference's own pre-step state, tight bounds).  A fresh critic's TRAJECTORY is chaotic: Adam's normalised step turns an
    # fp32 summation-order difference on a near-zero gradient element into a full-size update and the runs drift apart by ~1.3x per step
    # (reproduced on the CPU alone: test_g9_phase1_trajectory_is_sensitive_to_gradient_rounding).  Measured on MI355X: 1.2e-7 / 7.6e-5 /
    # 8.4e-3, parameters 1.7e-2; the loose end bounds only catch a run that went somewhere else entirely.
    assert e1[:8].max() <= 1e-5 and e1[:24].max() <= 1e-3 and e1.max() <= 5e-2
    assert ec <= 6e-2
    assert sorted(os.listdir("m/saves")) == json.loads(str(g["listing_json"]))["saves"]                  # both checkpoint names
    # ================= run B: the same command line with the REFERENCE's phase-1 critic in place: critic_pipe returns early
    # (main.py:164-166), so the sweep, the split and the mask-training loop start from the state the reference's did
    import shutil
    shutil.rmtree("m")
    os.makedirs("m/saves")
    torch.save({k[len("critic_after_p1/"):]: torch.from_numpy(v) for k, v in g.items() if k.startswith("critic_after_p1/")}, H.save_paths["critic"])
    rec.choice.clear(); rec.rand.clear()
    H = cli.main(argv)
    torch.cuda.synchronize()
    tr = H._trace
    assert not tr["p1_idx"], "the critic checkpoint exists: no phase-1 step may run"
    # ---- the split: sizes and side file exact, directory layout as the reference left it
    assert (len(H.Xpos), len(H.Xneg)) == (int(g["npos"]), int(g["nneg"]))
    assert sorted(f for f in os.listdir("m") if f.endswith(".txt")) == json.loads(str(g["side_files_json"]))
    ref_listing = json.loads(str(g["listing_json"]))
    assert sorted(os.listdir("m/saves")) == ref_listing["saves"]
    assert {"log.txt", "_loss.png"} <= set(os.listdir("m/segment")) and {"log.txt", "_loss.png"} <= set(ref_listing["segment"])
    # ---- the loop's draws: bit-exact, in order (three numpy index draws, then two torch shift draws per step)
    assert len(rec.rand) == 2 * ns2 and len(rec.choice) == 3 * ns2
    np.testing.assert_array_equal(np.array(rec.rand).reshape(ns2, 2), g["p2_shift_draws"])
    np.testing.assert_array_equal(np.stack([np.concatenate(rec.choice[3 * s:3 * s + 3]) for s in range(ns2)]), g["p2_choice"])
    np.testing.assert_array_equal(np.stack(tr["p2_idx"]), g["p2_choice"])
    want_roll = [(-int(12 * a) if b > 0.5 else int(12 * a)) for a, b in g["p2_shift_draws"]]
    assert [int(r) for r in tr["p2_roll"]] == want_roll
    # ---- every step's four loss terms, the final masker and critic
    p2 = torch.stack(tr["p2_loss"]).cpu().numpy()
    want = np.concatenate((g["p2_loss_critic_replace_inject"], 0.5 * g["p2_loss_l1_mean"][:, None]), axis=1)
    e2 = np.abs(p2[:, :4] - want) / np.abs(want)
    ck_m = torch.load(H.save_paths["masker"], map_location="cpu")
    em = max(float((ck_m[k] - torch.from_numpy(g["masker_final/" + k])).abs().max()) for k in ck_m)
    sd = {k: v.detach().cpu() for k, v in H.critic.state_dict().items()}
    ecf = max(float((sd[k] - torch.from_numpy(g["critic_final/" + k])).abs().max()) for k in sd)
    print(f"G9 run B: phase-2 loss rel err (critic, replace, inject, L1): first 8 steps {e2[:8].max(0)}, all 34 {e2.max(0)}; final masker max abs "
          f"{em:.1e}, final critic {ecf:.1e}")
    # measured on MI355X: first 8 steps 1.9e-5; all 34 steps 4.3e-4 / 3.7e-4 / 4.6e-3 / 3.1e-4 (the CPU oracle's own distance from the
    # reference run is the same: 4.0e-4 / 3.5e-4 / 4.4e-3 / 2.9e-4 -- the inject term is ~1e-5 in absolute size); masker 3.0e-5, critic 6.1e-6
    assert e2[:8].max() <= 1e-3                       # the G3 step tolerance over the first K = 8 steps
    assert e2[:, [0, 1, 3]].max() <= 2e-3 and e2[:, 2].max() <= 1.5e-2
    assert em <= 2e-4 and ecf <= 5e-5


def test_g10_main_eval_matches_the_reference_eval(tmp_path, monkeypatch):
    from cgs_amd import cli, handler
    g = load("g10_eval.npz")
    pc, pm = g1()
    X, Yrgb = li.synthetic_eval_set(int(g["n_frames"]), int(g["data_seed"]))
    monkeypatch.chdir(tmp_path)
    os.makedirs("red-trees")
    np.save("red-trees/X.npy", X)
    np.save("red-trees/Y.npy", Yrgb)
    os.makedirs("m/saves")
    probe = handler.checkpoint_names(cli.parse_args(["--model", "m"]))
    torch.save(pc, f"m/saves/critic-{probe[0]}.pt")
    torch.save(pm, f"m/saves/masker-{probe[1]}.pt")
    seen = []
    real_iou = handler.Handler.get_iou
    monkeypatch.setattr(handler.Handler, "get_iou", staticmethod(lambda A, B: (seen.append((np.asarray(A).copy(), np.asarray(B).copy())), real_iou(A, B))[1]))
    returned = []
    real_eval = handler.Handler.eval
    monkeypatch.setattr(handler.Handler, "eval", lambda self, *a, **k: (returned.append(real_eval(self, *a, **k)), returned[-1])[1])
    npx = 120 * 64 * 64
    for tag in ("plain", "thr02", "sal_global", "sal_k"):
        seen.clear(); returned.clear()
        cli.main(json.loads(str(g[f"{tag}/argv_json"])))
        ious = returned[0]
        want = g[f"{tag}/ious"]
        assert len(ious) == len(want)
        ref_hard = np.unpackbits(g[f"{tag}/mask_hard"])[:npx].astype(bool)
        got_hard = seen[0][0].reshape(-1).astype(bool)
        np.testing.assert_array_equal(seen[0][1].reshape(-1), np.unpackbits(g[f"{tag}/gt"])[:npx].astype(bool))
        flips = float((ref_hard != got_hard).mean())
        assert flips <= 1e-4, (tag, flips)             # mask values within 1e-3 of the threshold may land on the other side
        assert abs(ious[0] - want[0]) <= 2e-3, (tag, ious, want)
        if len(want) > 1:
            ref_sal = np.unpackbits(g[f"{tag}/sal_hard"])[:npx].astype(bool)
            got_sal = seen[-1][0].reshape(-1).astype(bool)
            sflips = float((ref_sal != got_sal).mean())
            assert sflips <= 2e-4, (tag, sflips)
            assert abs(ious[1] - want[1]) <= 2e-3 or (np.isnan(ious[1]) and want[1] == 0.0), (tag, ious, want)
        print(f"G10 {tag}: IoUs {ious} vs reference {want.tolist()}, mask flips {flips:.1e}")


def test_g11_collect_data_under_a_stub_minerl_writes_the_reference_pickle(tmp_path, monkeypatch):
    from cgs_amd import cli, handler
    g = load("g11_collect.npz")
    eps = li.synthetic_episodes(int(g["episodes_seed"]), int(g["n_episodes"]))
    by_name = {n: (p, r) for n, p, r in eps}

    class Data:
        def get_trajectory_names(self):
            return [n for n, _, _ in eps]

        def load_data(self, name):
            pov, reward = by_name[name]
            for t in range(len(reward)):
                yield ({"pov": pov[t]}, {"vector": np.zeros(4)}, reward[t], {"pov": pov[t]}, t == len(reward) - 1)
    stub = types.ModuleType("minerl")
    stub.data = types.SimpleNamespace(make=lambda *a, **k: Data(), download=lambda *a, **k: None)
    monkeypatch.setitem(sys.modules, "minerl", stub)
    monkeypatch.chdir(tmp_path)
    for tag in ("trunk", "trunk_small"):
        argv = json.loads(str(g[f"{tag}/argv_json"]))
        H = handler.Handler(cli.parse_args(argv))
        Xr, Yr, Ir = H.collect_data()
        files = sorted(os.listdir(H.data_path))
        assert files == [str(g[f"{tag}/file"])]
        with gzip.open(H.data_path + files[0], "rb") as fp:
            Xp, Yp, Ip = pickle.load(fp)
        assert [str(Xp.dtype), str(Yp.dtype), str(Ip.dtype)] == g[f"{tag}/pickle_dtypes"].tolist()
        np.testing.assert_array_equal(Yp, g[f"{tag}/pickle_Y"])
        np.testing.assert_array_equal(Ip, g[f"{tag}/pickle_I"])
        np.testing.assert_array_equal(Xp.reshape(len(Xp), -1).sum(1).astype(np.int64), g[f"{tag}/pickle_X_rowsum"])
        assert [len(Xr), Yr.shape[1], len(Ir)] == g[f"{tag}/ret_shapes"].tolist()
        np.testing.assert_array_equal(Yr, g[f"{tag}/ret_Y"])
        np.testing.assert_array_equal(Ir, g[f"{tag}/ret_I"])
        # a second call finds the pickle and returns ITS rows (main.py:1279-1284)
        X2, Y2, I2 = H.collect_data()
        np.testing.assert_array_equal(Y2, Yp)
        os.remove(H.data_path + files[0])


def test_g12_phase1_single_steps_from_the_reference_states():
    """Phase 1 pinned STEP BY STEP (replaces the loose end-of-trajectory bound of run A as the parity statement for
    row a9 at the loop level): for EACH of the 48 batches of the reference's `main.py -train` run (G9) the engine is put into the
    reference's recorded state before that batch -- critic parameters, Adam moments, step count (G12) -- runs ONE phase1_step on the recorded
    frames (DataLoader indices + shift draws of G9) and is compared with what the reference computed: the loss, all 14 gradients, and the
    parameters after the step (= the recorded state before the next batch), at the G4 tolerances."""
    from cgs_amd import engine
    from test_gpu_kernels import rel_close
    from test_loops_golden import g12_states, roll_from_draws
    g9 = load("g9_train_loop.npz")
    g, keys, unflat = g12_states()
    X, Y, _I = li.synthetic_frames(li.DATASIZE + li.TESTSIZE, int(g9["data_seed"]))
    dev = torch.device("cuda:0")
    nb = len(g9["p1_batch_len"])
    assert len(set(g9["p1_batch_len"].tolist())) == 1
    n = int(g9["p1_batch_len"][0])
    e = engine.HourglassEngine(n, device=dev, dropout=0.0)
    nc = e.lc.total
    pos, worst_loss, noise, viol = 0, 0.0, {}, []
    NOISE_X = 8.0
    for b in range(nb):
        idx = g9["p1_batch_idx"][pos:pos + n]
        pos += n
        xb = torch.roll(torch.from_numpy(X[idx]), roll_from_draws(*g9["p1_shift_draws"][b]), dims=2).contiguous()
        e.load_state(critic_sd=unflat(g["params_before"][b]))
        e.lc.flatten({k: v.to(dev) for k, v in unflat(g["adam_m_before"][b]).items()}, e.m[:nc])
        e.lc.flatten({k: v.to(dev) for k, v in unflat(g["adam_v_before"][b]).items()}, e.v[:nc])
        e.step_t.fill_(int(g["adam_step_before"][b]))
        loss = e.phase1_step(xb.to(dev), torch.from_numpy(Y[1, idx]).float().to(dev))
        torch.cuda.synchronize()
        assert int(e.step_t.item()) == b + 1
        rl = abs(float(loss[0]) - float(g["loss"][b])) / abs(float(g["loss"][b]))
        worst_loss = max(worst_loss, rl)
        assert rl <= 1e-4, (b, float(loss[0]), float(g["loss"][b]))
        got_g, want_g = e.lc.unflatten(e.gc), unflat(g["grads"][b])
        got_p, want_p = e.critic_state(), unflat(g["params_before"][b + 1] if b + 1 < nb else g["params_after_last"])
        # Late in the run some gradients are small sums of large cancelling terms (features.0.weight at batch 1: |g| <= 9e-5 from 262 144
        # products per element): there fp32 summation ORDER moves the result by more than 1e-3 of it -- for the reference's own CPU kernels
        # too.  The float64 oracle on the same state says how far the REFERENCE's fp32 gradient is from the exact one (ref_noise, per tensor);
        # the tolerance is the G4 one plus a small multiple of that, so it collapses to the G4 tolerance wherever the reference is accurate.
        P64 = {k: v.double().requires_grad_(True) for k, v in unflat(g["params_before"][b]).items()}
        l64, _ = orc.phase1_loss(P64, orc.u8_to_nchw(xb.numpy()).double(), torch.from_numpy(Y[1, idx]).double())
        l64.backward()
        for k in keys:
            g64 = P64[k].grad.numpy()
            ref_noise = float(np.abs(want_g[k].numpy().astype(np.float64) - g64).max())
            gpu_noise = float(np.abs(got_g[k].cpu().numpy().astype(np.float64) - g64).max())
            noise[k] = max(noise.get(k, (0, 0, 0)), (gpu_noise / max(ref_noise, 1e-30), gpu_noise, ref_noise))
            err = np.abs(got_g[k].cpu().numpy().astype(np.float64) - want_g[k].numpy())
            tol = 1e-3 * np.abs(want_g[k].numpy()) + 2e-5 * float(want_g[k].abs().max()) + NOISE_X * ref_noise
            if not (err <= tol).all():
                viol.append((b, k, float(err.max() / want_g[k].abs().max()), int((err > tol).sum()), err.size, ref_noise, gpu_noise))
                continue
            rel_close(got_p[k].cpu().numpy(), want_p[k].numpy(), f"G12 batch {b} {k} after the step", atol_scale=1e-4)
    # Measured on MI355X: 669 of the 672 (batch, tensor) pairs inside that tolerance; batch 25 has ONE routing difference in one image --
    # features.6.bias off in 1 of 8 channels (8e-4 of the tensor's maximum), features.6.weight in 10 of that channel's 72 elements (2.7e-4),
    # features.10.weight in one element (7e-5): a pre-activation / pooling candidate within fp32 rounding of its decision point falls on the
    # other side than on the CPU (the float64 oracle sides with the reference).  Such events are admitted explicitly: at most 4 pairs, each
    # within 1e-3 of its tensor's maximum, every one printed.
    for v in viol:
        print("G12 outside the elementwise tolerance: batch %d %s err/max %.2e in %d/%d elements (reference's distance from float64 %.2e, GPU's %.2e)" % v)
    assert len(viol) <= 4 and all(v[2] <= 1e-3 for v in viol), viol
    print("G12: worst (GPU distance from the float64 gradient) / (the reference's own), per tensor over the 48 steps:")
    for k, (ratio, gn, rn) in noise.items():
        print(f"   {k:22s} ratio {ratio:6.2f}  (GPU {gn:.2e}, reference {rn:.2e})")
    print(f"G12: 48 single phase-1 steps from the reference's states: worst loss rel err {worst_loss:.1e}")
