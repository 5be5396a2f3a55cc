#!/usr/bin/env python3
"""G9 / G10 / G11 ("pin the loop level"): the reference's OWN `main.main()` run end to end here on synthetic data, with
recorders around the library calls its loops make, so that the LOOPS (not only the step arithmetic of G2-G4) are pinned by the reference:

  G9  `main.py -train`: Handler.load_data -> critic_pipe (main.py:158-236: DataLoader(shuffle) batches, shift_batch draws, one Adam step
      per batch) -> extract_contrastive_data (main.py:238-312: sweep, threshold split, the `{npos}>{thr}__{nneg}<{thr}.txt` side file)
      -> segmentation_training (main.py:314-575: three np.random.choice draws + two torch shift draws per step, the phase-2 step)
      -> both checkpoints.  Captured: every batch's frame indices, every RNG draw in order, every step's loss values, the sweep's critic
      values, the side-file name, the checkpoints.
  G10 `main.py -eval -salience`: Handler.eval (main.py:891-1016) on a synthetic `red-trees/` -- the masks, the saliency maps before and after
      the normalisation of main.py:976-1003, what get_iou (main.py:1265-1270) was called with and what it returned.
  G11 Handler.collect_data (main.py:1272-1359) under a stub `minerl.data.make` that yields synthetic episodes: the gz-pickle it writes
      (trunk filter main.py:1325, clipped discounted rewards main.py:1336-1346) and the arrays it returns.

Run in the build container only (the reference does not exist on the GPU box):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python /root/repo/tests/golden/make_golden_loops.py

Supplied around the reference, as for G6 (make_golden_g6.py): `torchvision` stubs + `np.int = int` for nets.py, `np.float = float`
(main.py:120-121,1296,1331 under numpy 2), empty stub modules for cv2 / ffmpeg, a stub `minerl` (G11: a `data` namespace with `make` /
`download`), a TrueType file at the relative path Handler.__init__ opens.  The networks start from the G1 weights and the RNGs are seeded at
the top of critic_pipe / segmentation_training (wrappers below): the build's tests install the same two hooks around ITS Handler.
Only DATA is written (g9_train_loop.npz, g10_eval.npz, g11_collect.npz); no reference source is copied."""
import gzip
import json
import os
import pickle
import shutil
import sys
import tempfile
import types

sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")
HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

import numpy as np
import torch
from PIL import Image

np.int = int
np.float = float
for name in ("torchvision", "torchvision.models", "cv2", "ffmpeg"):
    sys.modules.setdefault(name, types.ModuleType(name))
minerl_stub = types.ModuleType("minerl")
minerl_stub.data = types.SimpleNamespace()
sys.modules["minerl"] = minerl_stub
sys.path.insert(0, REF)

torch.set_num_threads(1)
torch.use_deterministic_algorithms(True)

sys.path.insert(0, HERE)
from loop_inputs import SEED_P1, SEED_P2, DATASIZE, TESTSIZE, synthetic_frames, synthetic_episodes, synthetic_eval_set  # noqa: E402


def g1():
    raw = dict(np.load(os.path.join(HERE, "g1_weights_chfak1.npz")))
    pc = {k.split("/", 1)[1]: torch.from_numpy(v) for k, v in raw.items() if k.startswith("critic/")}
    pm = {k.split("/", 1)[1]: torch.from_numpy(v) for k, v in raw.items() if k.startswith("masker/")}
    return pc, pm


def scratch():
    import matplotlib
    tmp = tempfile.mkdtemp(prefix="g9_")
    os.makedirs(os.path.join(tmp, "isy_minerl/segm/etc"))
    shutil.copy(os.path.join(matplotlib.get_data_path(), "fonts", "ttf", "DejaVuSans.ttf"), os.path.join(tmp, "isy_minerl/segm/etc/Ubuntu-R.ttf"))
    return tmp


def run_main(argv):
    import main as refmain
    old = sys.argv
    sys.argv = ["main.py"] + argv
    try:
        refmain.main()
    finally:
        sys.argv = old


class Recorders:
    """Wrappers around the library entry points the reference's loops call; every record is plain data."""

    def __init__(self, refmain):
        self.m = refmain
        self.choice, self.rand, self.mse, self.l1, self.hist, self.iou, self.batches = [], [], [], [], [], [], []
        self._saved = []
        self.before_mse = None       # optional observer called when F.mse_loss is entered (G12: the state before a phase-1 step)

    def install(self):
        m = self.m
        real_choice, real_rand = np.random.choice, torch.rand
        real_mse, real_l1, real_hist = m.F.mse_loss, m.F.l1_loss, m.plt.hist

        def choice(a, size=None, *args, **kw):
            r = real_choice(a, size, *args, **kw)
            self.choice.append(np.asarray(r).copy())
            return r

        def rand(*size, **kw):
            r = real_rand(*size, **kw)
            if tuple(size) == (1,):               # shift_batch's two draws (main.py:585-586)
                self.rand.append(float(r))
            return r

        def mse(a, b, *args, **kw):
            if self.before_mse is not None:
                self.before_mse()
            r = real_mse(a, b, *args, **kw)
            self.mse.append(float(r))
            return r

        def l1(a, b, *args, **kw):
            r = real_l1(a, b, *args, **kw)
            self.l1.append(float(r))
            return r

        def hist(x, *args, **kw):
            self.hist.append(np.asarray(x, dtype=np.float64).copy())
            return real_hist(x, *args, **kw)

        self._saved = [(np.random, "choice", real_choice), (torch, "rand", real_rand), (m.T, "rand", real_rand),
                       (m.F, "mse_loss", real_mse), (m.F, "l1_loss", real_l1), (m.plt, "hist", real_hist)]
        np.random.choice, torch.rand, m.F.mse_loss, m.F.l1_loss, m.plt.hist = choice, rand, mse, l1, hist

    def remove(self):
        for obj, name, fn in self._saved:
            setattr(obj, name, fn)


class LoaderProxy:
    """Stands in for Handler.train_loader inside critic_pipe: yields the real DataLoader's batches and keeps their index column."""

    def __init__(self, loader, sink):
        self.loader, self.sink = loader, sink

    def __iter__(self):
        for X, Y, I in self.loader:
            self.sink.append(I.numpy().copy())
            yield X, Y, I


def g9(out_path):
    import main as refmain
    pc, pm = g1()
    X, Y, I = synthetic_frames(DATASIZE + TESTSIZE, 9)
    tmp = scratch()
    cwd = os.getcwd()
    os.chdir(tmp)
    rec = Recorders(refmain)
    handlers = []
    real_init, real_cp, real_seg = refmain.Handler.__init__, refmain.Handler.critic_pipe, refmain.Handler.segmentation_training

    def init(self, args):
        real_init(self, args)
        handlers.append(self)

    # G12 (round 6): the reference's critic + Adam state BEFORE each phase-1 batch and the gradient each batch produced -- observers only
    # (clones of tensors the run holds anyway), the run itself is untouched and g9_train_loop.npz regenerates byte-identically
    p1 = {"on": False, "params": [], "m": [], "v": [], "step": [], "grads": []}
    optis = []
    real_adam = torch.optim.Adam

    def adam_ctor(params, *a, **kw):
        o = real_adam(params, *a, **kw)
        optis.append(o)
        return o

    def flat_of(tensors):
        return np.concatenate([t.detach().numpy().reshape(-1) for t in tensors]).astype(np.float32)

    def before_p1_step():
        if not p1["on"]:
            return
        H, opt = handlers[-1], optis[-1]
        params = list(H.critic.parameters())
        if p1["params"]:                 # .grad still holds the previous batch's gradient (opti.zero_grad() comes after this call)
            p1["grads"].append(flat_of([p.grad for p in params]))
        st = [opt.state.get(p, {}) for p in params]
        p1["params"].append(flat_of(params))
        p1["m"].append(flat_of([s["exp_avg"] if "exp_avg" in s else torch.zeros_like(p) for s, p in zip(st, params)]))
        p1["v"].append(flat_of([s["exp_avg_sq"] if "exp_avg_sq" in s else torch.zeros_like(p) for s, p in zip(st, params)]))
        steps = {int(s["step"]) for s in st if "step" in s} or {0}
        assert len(steps) == 1
        p1["step"].append(steps.pop())

    def critic_pipe(self, mode="train", test=0):
        self.critic.load_state_dict(pc)
        self.masker.load_state_dict(pm)
        self.train_loader = LoaderProxy(self.train_loader, rec.batches)
        np.random.seed(SEED_P1)
        torch.manual_seed(SEED_P1)
        p1["on"] = True
        try:
            return real_cp(self, mode, test)
        finally:
            p1["on"] = False
            if p1["params"] and len(p1["grads"]) < len(p1["params"]):
                p1["grads"].append(flat_of([p.grad for p in self.critic.parameters()]))

    def segmentation_training(self):
        np.random.seed(SEED_P2)
        torch.manual_seed(SEED_P2)
        return real_seg(self)

    refmain.Handler.__init__, refmain.Handler.critic_pipe, refmain.Handler.segmentation_training = init, critic_pipe, segmentation_training
    try:
        os.makedirs("runs/data/straight")
        with gzip.GzipFile(f"runs/data/straight/Treechop-trunk-{DATASIZE}-[0.98-0.97-0.96-0.95].pickle", "wb") as fp:
            pickle.dump((X, Y, I), fp)
        base = ["-train", "--model", "m", "--datasize", str(DATASIZE), "--testsize", str(TESTSIZE), "--dropout", "0", "--cepochs", "1",
                "--mepochs", "1"]
        # pass 1: phase 1 + the sweep only, to see the critic's values and put the thresholds into gaps of them
        rec.install()

        class Stop(Exception):
            pass
        real_extract = refmain.Handler.extract_contrastive_data

        def extract_then_stop(self):
            try:
                real_extract(self)
            except AssertionError:
                pass
            raise Stop()
        refmain.Handler.extract_contrastive_data = extract_then_stop
        try:
            run_main(base)
        except Stop:
            pass
        refmain.Handler.extract_contrastive_data = real_extract
        rec.remove()
        preds = np.sort(rec.hist[0])

        def gap_threshold(q):
            """the middle of the widest gap between neighbouring critic values within +-4 % of the quantile q"""
            lo, hi = int((q - 0.04) * len(preds)), int((q + 0.04) * len(preds))
            k = lo + int(np.argmax(np.diff(preds[lo:hi + 1])))
            return float(np.float32(round((preds[k] + preds[k + 1]) / 2, 5))), float(preds[k + 1] - preds[k])
        hi_thr, hi_gap = gap_threshold(0.62)
        lo_thr, lo_gap = gap_threshold(0.38)
        hi_thr, lo_thr = round(hi_thr, 5), round(lo_thr, 5)
        print(f"pass 1: critic values {preds.min():.4f} .. {preds.max():.4f}; thresholds {lo_thr} (gap {lo_gap:.2e}) / {hi_thr} (gap {hi_gap:.2e})")
        shutil.rmtree("m")
        # pass 2: the full command line
        rec = Recorders(refmain)
        handlers.clear()
        for k in ("params", "m", "v", "step", "grads"):
            p1[k].clear()
        rec.before_mse = before_p1_step
        rec.install()
        torch.optim.Adam = adam_ctor
        argv = base + ["--high-rew-thresh", repr(hi_thr), "--low-rew-thresh", repr(lo_thr)]
        try:
            run_main(argv)
        finally:
            torch.optim.Adam = real_adam
        rec.remove()
        H = handlers[-1]
        nb1 = len(rec.batches)
        nsteps2 = len(rec.choice) // 3
        assert len(rec.rand) == 2 * (nb1 + nsteps2) and len(rec.mse) == nb1 + 3 * nsteps2 and len(rec.l1) == nsteps2
        side = sorted(f for f in os.listdir("m") if f.endswith(".txt"))
        ck_c = torch.load(H.save_paths["critic"])
        ck_m = torch.load(H.save_paths["masker"])
        out = {"argv_json": np.array(json.dumps(argv)), "seeds": np.array([SEED_P1, SEED_P2]),
               "data_seed": np.array(9), "high_thresh": np.array(hi_thr), "low_thresh": np.array(lo_thr),
               "threshold_gaps": np.array([lo_gap, hi_gap]),
               "side_files_json": np.array(json.dumps(side)),
               "listing_json": np.array(json.dumps({d: sorted(os.listdir(os.path.join("m", d))) for d in ("saves", "critic", "segment")})),
               "npos": np.array(len(H.Xpos)), "nneg": np.array(len(H.Xneg)),
               "sweep_preds": rec.hist[0].astype(np.float32),
               "p1_batch_idx": np.concatenate([b.astype(np.int32) for b in rec.batches]),
               "p1_batch_len": np.array([len(b) for b in rec.batches]),
               "p1_shift_draws": np.array(rec.rand[:2 * nb1]).reshape(nb1, 2),
               "p1_loss": np.array(rec.mse[:nb1]),
               "p2_choice": np.stack([np.concatenate(rec.choice[3 * s:3 * s + 3]) for s in range(nsteps2)]).astype(np.int32),
               "p2_shift_draws": np.array(rec.rand[2 * nb1:]).reshape(nsteps2, 2),
               "p2_loss_critic_replace_inject": np.array(rec.mse[nb1:]).reshape(nsteps2, 3),
               "p2_loss_l1_mean": np.array(rec.l1)}
        for k, v in ck_c.items():
            out["critic_after_p1/" + k] = v.numpy()
        for k, v in ck_m.items():
            out["masker_final/" + k] = v.numpy()
        for k, v in H.critic.state_dict().items():
            out["critic_final/" + k] = v.detach().numpy()
        np.savez_compressed(out_path, **out)
        # ---- G12: one row per phase-1 batch, tensors concatenated in named_parameters() order (= g1_keys.json's), reference (OIHW) layout
        assert len(p1["params"]) == len(p1["grads"]) == nb1 and p1["step"] == list(range(nb1)), (len(p1["params"]), len(p1["grads"]), p1["step"])
        keys = [k for k, _ in H.critic.named_parameters()]
        last = flat_of([ck_c[k] for k in keys])          # the state after the last batch = the phase-1 checkpoint
        np.savez_compressed(os.path.join(os.path.dirname(out_path), "g12_phase1_steps.npz"),
                            keys_json=np.array(json.dumps(keys)), shapes_json=np.array(json.dumps([list(ck_c[k].shape) for k in keys])),
                            params_before=np.stack(p1["params"]), adam_m_before=np.stack(p1["m"]), adam_v_before=np.stack(p1["v"]),
                            adam_step_before=np.array(p1["step"]), grads=np.stack(p1["grads"]), params_after_last=last,
                            loss=np.array(rec.mse[:nb1]))
        print(f"wrote g12_phase1_steps.npz: {nb1} pre-step states of {len(last)} floats")
        print(f"wrote {os.path.basename(out_path)}: {nb1} phase-1 batches, {nsteps2} phase-2 steps, split {len(H.Xpos)} / {len(H.Xneg)}, side file {side}")
    finally:
        rec.remove()
        refmain.Handler.__init__, refmain.Handler.critic_pipe, refmain.Handler.segmentation_training = real_init, real_cp, real_seg
        os.chdir(cwd)
        shutil.rmtree(tmp, ignore_errors=True)


def g10(out_path):
    """Handler.eval on a synthetic red-trees/ (X.npy uint8 [N,64,64,3], Y.npy [N,64,64,3] whose all-channels-non-zero pixels are the
    ground truth, main.py:920-928); the reference evaluates the fixed slice X[100:5000:2].  The test re-generates the frames from the
    seed (tests/golden/loop_inputs.py); stored: hard masks, ground truth, IoUs, the first 40 raw saliency maps."""
    import main as refmain
    pc, pm = g1()
    n = 340                                      # X[100:5000:2] of 340 frames = 120 evaluated frames
    X, Yrgb = synthetic_eval_set(n, 10)
    tmp = scratch()
    cwd = os.getcwd()
    os.chdir(tmp)
    rec = {"iou_in": [], "iou_out": []}
    real_iou = refmain.Handler.get_iou

    def get_iou(self, A, B):
        r = real_iou(self, A, B)
        rec["iou_in"].append((np.asarray(A).copy(), np.asarray(B).copy()))
        rec["iou_out"].append(float(r))
        return r
    refmain.Handler.get_iou = get_iou
    # the saliency maps before / after main.py:976-1003 are locals of eval(): np.sort is called on the flattened raw maps (the input of the
    # normalisation); the normalised maps come back through get_iou's first argument as the thresholded uint8 mask
    sal = {}
    real_sort = np.sort

    def sort(a, *args, **kw):
        if getattr(a, "ndim", 0) == 3 and a.shape[1] == 1:
            sal["raw_flat"] = np.asarray(a).copy()
        return real_sort(a, *args, **kw)
    try:
        os.makedirs("red-trees")
        np.save("red-trees/X.npy", X)
        np.save("red-trees/Y.npy", Yrgb)
        # checkpoint names from the reference's own Handler, as in G6
        handlers = []
        real_init = refmain.Handler.__init__

        def init(self, args):
            real_init(self, args)
            handlers.append(self)
            if not os.path.exists(self.save_paths["critic"]):
                os.makedirs(self.save_path, exist_ok=True)
                torch.save(pc, self.save_paths["critic"])
                torch.save(pm, self.save_paths["masker"])
        refmain.Handler.__init__ = init
        out = {"n_frames": np.array(n), "data_seed": np.array(10)}
        runs = {"plain": ["-eval", "--model", "m", "-visbesteval", ""],
                "sal_k": ["-eval", "-salience", "-salglobal", "", "--salience-thresh", "0.9", "--model", "m", "-visbesteval", ""],
                "sal_global": ["-eval", "-salience", "--model", "m", "-visbesteval", ""],
                "thr02": ["-eval", "--eval-thresh", "0.2", "--model", "m", "-visbesteval", ""]}
        for tag, argv in runs.items():
            rec["iou_in"].clear(); rec["iou_out"].clear(); sal.clear()
            np.sort = sort
            try:
                run_main(argv)
            finally:
                np.sort = real_sort
            out[f"{tag}/argv_json"] = np.array(json.dumps(argv))
            out[f"{tag}/ious"] = np.array(rec["iou_out"])
            out[f"{tag}/mask_hard"] = np.packbits(rec["iou_in"][0][0].astype(bool))
            out[f"{tag}/gt"] = np.packbits(rec["iou_in"][0][1].astype(bool))
            if len(rec["iou_in"]) > 1:
                out[f"{tag}/sal_hard"] = np.packbits(rec["iou_in"][1][0].astype(bool))
            if "raw_flat" in sal:
                out[f"{tag}/sal_raw_first40"] = sal["raw_flat"].reshape(-1, 64, 64)[:40].astype(np.float32)
            print(tag, "ious", rec["iou_out"])
        np.savez_compressed(out_path, **out)
        print("wrote", os.path.basename(out_path))
    finally:
        refmain.Handler.get_iou = real_iou
        refmain.Handler.__init__ = real_init
        os.chdir(cwd)
        shutil.rmtree(tmp, ignore_errors=True)


def g11(out_path):
    import main as refmain
    eps = synthetic_episodes()
    by_name = {n: (p, r) for n, p, r in eps}

    class Data:
        def get_trajectory_names(self):
            return [n for n, _, _ in eps]

        def load_data(self, name):
            pov, reward = by_name[name]
            for t in range(len(reward)):
                yield ({"pov": pov[t]}, {"vector": np.zeros(4)}, reward[t], {"pov": pov[t]}, t == len(reward) - 1)
    minerl_stub.data.make = lambda *a, **k: Data()
    minerl_stub.data.download = lambda *a, **k: None
    refmain.minerl = minerl_stub
    tmp = scratch()
    cwd = os.getcwd()
    os.chdir(tmp)
    captured = {}
    real_collect = refmain.Handler.collect_data

    def collect(self):
        r = real_collect(self)
        captured["ret"] = r
        captured["path"] = self.data_path
        raise SystemExit(0)
    refmain.Handler.collect_data = collect
    try:
        out = {"episodes_seed": np.array(11), "n_episodes": np.array(len(eps))}
        for tag, argv in {"trunk": ["-train", "--model", "m", "--datasize", "600", "--testsize", "200"],
                          "trunk_small": ["-train", "--model", "m", "--datasize", "150", "--testsize", "50", "--gammas", "0.9-0.5"]}.items():
            try:
                run_main(argv)
            except SystemExit:
                pass
            files = sorted(os.listdir(captured["path"]))
            assert len(files) == 1, files
            with gzip.open(captured["path"] + files[0], "rb") as fp:
                Xp, Yp, Ip = pickle.load(fp)
            Xr, Yr, Ir = captured["ret"]
            out[f"{tag}/argv_json"] = np.array(json.dumps(argv))
            out[f"{tag}/file"] = np.array(files[0])
            # frames: the (episode, frame) each row came from is enough to rebuild X -- store I and a checksum of X instead of 7 MB of noise
            out[f"{tag}/pickle_Y"], out[f"{tag}/pickle_I"] = Yp, Ip
            out[f"{tag}/pickle_X_rowsum"] = Xp.reshape(len(Xp), -1).sum(1).astype(np.int64)
            out[f"{tag}/pickle_dtypes"] = np.array([str(Xp.dtype), str(Yp.dtype), str(Ip.dtype)])
            out[f"{tag}/ret_shapes"] = np.array([Xr.shape[0], Yr.shape[1], Ir.shape[0]])
            out[f"{tag}/ret_Y"], out[f"{tag}/ret_I"] = Yr, Ir
            print(tag, files[0], "pickle rows", len(Xp), "returned rows", len(Xr))
            shutil.rmtree(captured["path"])
        np.savez_compressed(out_path, **out)
        print("wrote", os.path.basename(out_path))
    finally:
        refmain.Handler.collect_data = real_collect
        os.chdir(cwd)
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g9", "g10", "g11"]
    if "g11" in which:
        g11(os.path.join(HERE, "g11_collect.npz"))
    if "g10" in which:
        g10(os.path.join(HERE, "g10_eval.npz"))
    if "g9" in which:
        g9(os.path.join(HERE, "g9_train_loop.npz"))
