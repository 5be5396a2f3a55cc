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
    # Phase 1 trains a fresh critic: Adam's normalised step turns an fp32 summation-order difference on a near-zero gradient element into a
    # full-size update, and the trajectories drift apart by ~1.3x per step -- reproduced on the CPU alone by adding noise of 3e-7 of each
    # gradient tensor's maximum to the oracle's replay (test_g9_phase1_trajectory_is_sensitive_to_gradient_rounding: loss 4e-2, parameters
    # 2e-2 at step 48).  Hence tight bounds where the trajectories are still together and loose ones at the end.  Measured on MI355X:
    # 1.2e-7 / 7.6e-5 / 8.4e-3, parameters 1.7e-2.  (Phase 2 below starts from a trained critic and stays together: 3e-5.)
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
