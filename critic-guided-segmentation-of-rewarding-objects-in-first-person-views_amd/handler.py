"""Host-side orchestration of the ``-train`` and ``-process`` paths: the reference's ``Handler`` surface
(main.py:66-156, 158-236, 238-312, 314-575, 584-591, 1103-1223) with its inner loops replaced by the fused
HIP engine.  Same method names, same checkpoint / dataset file naming, same output file names.

Out of scope here: CRF, videos / PNG debug grids.  ``collect_data`` reads an existing gz-pickle or, when the
``minerl`` package is importable, builds it from MineRL episodes exactly as the reference labels them (the MineRL download / decoder
itself is the package's; it is absent from this image).  ``-eval`` (section 8 f2) is carried over without CRF / videos.
"""
import gzip
import math
import os
import pickle

import numpy as np
import torch

from . import _lib, dataformat, parallel
from .engine import HourglassEngine
from .generic_engine import GenericEngine
from .nets import NewCritic, UnetDecoder


def hg_mix_fused():
    from . import hourglass
    return hourglass.ENC0_MIX_FUSED


def checkpoint_names(args):
    """Checkpoint file stems of the reference: ``k=v`` joined by '-' for TRUTHY values only (main.py:86-91), e.g.
    critic-rewidx=1-cepochs=15-datamode=trunk-datasize=100000-shift=12-chfak=1-dropout=0.3 / masker-mepochs=1-L1=0.5-inject=True."""
    d = args.__dict__
    critic_args = "-".join(f"{a}={d[a]}" for a in
                           ["rewidx", "cepochs", "datamode", "datasize", "threshrew", "shift", "chfak", "dropout"] if d[a])
    masker_args = "-".join(f"{a}={d[a]}" for a in ["mepochs", "L1", "L2", "inject"] if d[a])
    return critic_args, masker_args


class Handler:
    def __init__(self, args):
        self.args = args
        if not torch.cuda.is_available():
            raise _lib.CgsError("this build runs the Hourglass on an MI355X through HIP kernels; no GPU is visible "
                                "and there is no CPU fallback")
        # one process per GPU under torchrun (RANK / LOCAL_RANK / WORLD_SIZE): the training steps then all-reduce their gradients
        # and the contrastive sweep is sharded by frame; a plain `python main.py` is world size 1
        self.pg = parallel.init_from_env()
        self.rank, local, self.world = parallel.env_world()
        self.device = f"cuda:{local}" if self.world > 1 else "cuda"
        print("device:", self.device)
        # attribute names, directory layout and checkpoint file names are the reference's (main.py:66-107): they are the contract
        self.criticname, self.maskername = "critic", "masker"
        self.ious, self.bestepoch = (0, 0), 0
        self.reset_models()
        self.models = {self.criticname: self.critic, self.maskername: self.masker}
        self.critic_args, self.masker_args = checkpoint_names(args)
        root = f"{args.name}/"
        self.path, self.data_path = root, "runs/data/straight/"
        self.train_path, self.result_path, self.save_path = (root + sub for sub in ("train/", "results/", "saves/"))
        self.save_paths = {name: f"{self.save_path}{name}-{tag}.pt"
                           for name, tag in ((self.criticname, self.critic_args), (self.maskername, self.masker_args))}
        self._engines = {}
        self._trace = None          # tests set a dict of lists (Handler.start_trace): per-step indices / losses of the two training loops

    def start_trace(self):
        """Keep every step's frame indices, shift and loss values of critic_pipe / segmentation_training (device clones, no host sync)."""
        self._trace = {"p1_idx": [], "p1_loss": [], "p2_idx": [], "p2_roll": [], "p2_loss": []}
        return self._trace

    # ------------------------------------------------------------------ models / checkpoints
    def reset_models(self):
        args = self.args
        self.critic = NewCritic(bottleneck=args.neck, chfak=args.chfak, dropout=args.dropout).to(self.device)
        self.masker = UnetDecoder(bottleneck=args.neck, chfak=args.chfak).to(self.device)
        if args.separate:       # main.py:110-111: a second critic feeds the masker; like the reference it is never checkpointed
            self.sepcrit = NewCritic(bottleneck=args.neck, chfak=args.chfak, dropout=args.dropout).to(self.device)

    def load_models(self, modelnames=[]):
        """Loads the named checkpoints (all when empty); False at the first missing file (main.py:130-141)."""
        for name in (modelnames or list(self.models)):
            ckpt = self.save_paths[name]
            if not os.path.isfile(ckpt):
                if not self.args.train:
                    print(f"{ckpt} not found")
                return False
            print("loading:", ckpt)
            self.models[name].load_state_dict(torch.load(ckpt, map_location=torch.device(self.device)))
        return True

    def save_models(self, modelnames=[]):
        if self.rank != 0:          # replicas are identical: one writer
            return
        os.makedirs(self.save_path, exist_ok=True)
        for name in (modelnames or list(self.models)):
            print("saving:", self.save_paths[name])
            torch.save(self.models[name].state_dict(), self.save_paths[name])

    # ------------------------------------------------------------------ data
    def collect_data(self):
        args = self.args
        filepath = dataformat.dataset_path(args.envname, args.datamode, args.datasize, args.gammas, self.data_path)
        print("collecting dataset at", filepath)
        if not os.path.exists(filepath):
            return self._collect_fresh(filepath)
        print("loading existing dataset...")
        X, Y, I = dataformat.read_dataset(filepath)
        print("finished loading exisiting dataset")
        return X, Y, I

    def _collect_fresh(self, filepath):
        """main.py:1286-1359: no pickle yet -> read MineRL episodes through the `minerl` package (imported lazily; absent in this image,
        where only a test stub stands in for it), label them (dataformat.build_dataset: trunk filter main.py:1325, clipped discounted
        rewards main.py:1336-1346) and write the gz-pickle.  Like the reference, the pickle holds the rows that were filled and the
        RETURNED arrays are the full --datasize + --testsize buffers (zero rows at the end when the episodes ran out)."""
        args = self.args
        try:
            import minerl
        except ImportError as e:
            raise FileNotFoundError(
                f"{filepath} not found and the `minerl` package is not installed. This build reads the reference's gz-pickle (X uint8 "
                "[N,64,64,3], Y float [7,N], I uint16 [N]); collecting it from MineRL needs `minerl` + its data (SURVEY.md 2.3).") from e
        root = os.getenv("MINERL_DATA_ROOT", "data/")
        env = f"MineRL{args.envname}VectorObf-v0"
        os.makedirs(self.data_path, exist_ok=True)
        if not os.path.exists(f"{root}/{env}"):
            minerl.data.download(root, experiment=env)
        data = minerl.data.make(env, data_dir=root, num_workers=args.workers[0], worker_batch_size=args.workers[1])
        size = args.datasize + args.testsize
        print("collecting straight data set with", args.datasize, "+", args.testsize, "frames")

        def episodes():
            for name in data.get_trajectory_names():
                state, _action, reward, _next, _done = zip(*data.load_data(name))
                yield np.stack([s["pov"] for s in state]), np.array(reward)
        X, Y, I = dataformat.build_dataset(episodes(), size, mode=args.datamode, gammas=[float(g) for g in args.gammas.split("-")])
        rows = len(X)
        with gzip.GzipFile(filepath, "wb") as fp:
            pickle.dump((X, Y, I), fp)
        Xf = np.zeros((size, 64, 64, 3), dtype=np.uint8)
        Yf = np.zeros((dataformat.Y_ROWS, size), dtype=np.float64)
        If = np.zeros(size, dtype=np.uint16)
        Xf[:rows], Yf[:, :rows], If[:rows] = X, Y, I
        return Xf, Yf, If

    def load_data(self, batch_size=64):
        """Train / test split (the last --testsize frames are the test set) and the --threshrew binarisation (main.py:113-128)."""
        X, Y, I = self.collect_data()
        cut, thr = -self.args.testsize, self.args.threshrew
        label = (lambda y: (y > thr).astype(np.float64)) if thr else (lambda y: y)
        self.X, self.Y, self.I = X[:cut], label(Y[:, :cut]), I[:cut]
        self.XX, self.YY, self.II = X[cut:], label(Y[:, cut:]), I[cut:]
        print("dataset shapes", X.shape, Y.shape, self.X.shape, self.Y.shape)
        self.batch_size = batch_size

    def _batches(self):
        """Shuffled mini-batches of (X uint8, Y[rewidx], frame indices) in the order the reference's
        DataLoader(TensorDataset(X, Y.t(), arange), batch_size, shuffle=True) yields them (main.py:125-129): the index stream comes from
        the same torch.utils.data.DataLoader over the frame indices, so a seeded run draws from the global torch RNG exactly as the
        reference does (one base-seed draw per epoch, one sampler-seed draw, the permutation from the sampler's private generator) and
        the shift draws that follow line up with the reference's (pinned by the G9 capture, tests/test_gpu_loops.py)."""
        order = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(torch.arange(len(self.X))), batch_size=self.batch_size, shuffle=True)
        for (idx_t,) in order:
            idx = idx_t.numpy()
            yield torch.from_numpy(self.X[idx]), torch.from_numpy(self.Y[self.args.rewidx, idx]).float(), idx

    def _shift_draw(self):
        """The two draws main.py:585-586 takes from the global torch RNG, as a signed roll along the width (positive = the
        reference's "right" branch, negative = its "left" branch)."""
        amount = int(self.args.shift * torch.rand(1))
        return -amount if bool(torch.rand(1) > 0.5) else amount

    def shift_batch(self, X):
        """Whole-batch circular roll along the width (main.py:584-591): one torch.roll instead of the reference's cat of slices."""
        return torch.roll(X, shifts=Handler._shift_draw(self), dims=2)

    # ------------------------------------------------------------------ engines
    def _generic_size(self):
        return self.args.chfak != 1 or self.args.neck != 32

    def _engine(self, n, live=True, training=False, dropout=None):
        """The engine for batches of n images: the fused fixed-shape kernels at chfak = 1, neck = 32 (the code default), the
        shape-generic ones for every other model size (the paper's chfak = 5).  dropout: override of --dropout (0 after a
        -directeval evaluation, which leaves the reference's modules in eval mode for the rest of the training)."""
        a = self.args
        p_drop = a.dropout if dropout is None else dropout
        key = (n, live, p_drop)
        if key not in self._engines:
            first = next(iter(self._engines.values()), None)
            kw = dict(device=self.device, dropout=p_drop, lfak=a.lfak, L1=a.L1, L2=a.L2, inject=a.inject, live=live,
                      threshrew=a.threshrew, share_with=first, process_group=self.pg, separate=bool(a.separate),
                      staticnorm=bool(a.staticnorm))
            if self._generic_size():
                e = GenericEngine(n, chfak=a.chfak, neck=a.neck, **kw)
            else:
                e = HourglassEngine(n, **kw)
            if first is None:
                # modules and engine share one parameter buffer from now on
                e.adopt(self.critic, self.masker, self.sepcrit if a.separate else None)
                parallel.broadcast_params_(e.flat, self.pg)
            self._engines[key] = e
        return self._engines[key]

    def _reset_adam(self):
        for e in self._engines.values():
            e.reset_optimizer()
            break

    # ------------------------------------------------------------------ phase 1: critic regression
    def _refuse_unbuilt_flags(self):
        """Flags the reference reads on this path that this build does not implement: refuse instead of training something else."""
        a = self.args
        if not a.staticnorm and not hg_mix_fused():
            raise NotImplementedError("-staticnorm '' (mask regulariser weighted by 1 - pred, main.py:415-418) needs the fused "
                                      "features.0 + mix backward (hourglass.ENC0_MIX_FUSED, this build's fixed configuration)")

    def critic_pipe(self, mode="train", test=0):
        args = self.args
        self._refuse_unbuilt_flags()
        if args.cload and self.load_models([self.criticname]):
            print("loaded critic, no new training")
            return
        result_path = self.path + "critic/"
        os.makedirs(result_path, exist_ok=True)
        if self.rank == 0:                        # (data parallel: one writer)
            with open(result_path + "log.txt", "w") as log_file:
                log_file.write(f"{self.args}\n\n")
        llog = []
        self.critic.train()
        # -directeval (main.py:179-180): Handler.eval() puts critic and masker into eval mode and never back (main.py:900-902; the
        # re-.train() at main.py:1019-1020 is commented out), so -- unless -noevalmode -- the reference then trains with Dropout OFF
        self._p1_dropout = 0.0 if (args.directeval and not args.noevalmode) else None
        self._engine(self.batch_size, training=True, dropout=self._p1_dropout)
        self._reset_adam()                       # a fresh torch.optim.Adam(critic.parameters()) (main.py:178)
        if args.directeval:
            self.eval()
            if args.noevalmode:
                self.critic.train()
        for epoch in range(int(mode == "test") or args.cepochs):
            for b_idx, (X, Y, idx) in enumerate(self._batches()):
                if args.shift:
                    X = self.shift_batch(X)
                eng = self._engine(len(X), dropout=self._p1_dropout)
                losses = eng.phase1_step(X.contiguous().to(self.device, non_blocking=True), Y.to(self.device, non_blocking=True))
                if self._trace is not None:      # (tests: the loop-level pin G9) no host sync: device clones
                    self._trace["p1_idx"].append(idx.copy())
                    self._trace["p1_loss"].append(losses[:1].clone())
                if not b_idx % 10:
                    val = float(losses[0])       # the only host sync, every 10th batch
                    llog.append(val)
                    print(f"critic e{epoch + 1} b{b_idx}", val, end="\r")
            if not (epoch + 1) % args.saveevery:
                self.save_models(modelnames=[self.criticname])
            if self.rank == 0:                    # (data parallel: one writer)
                self._plot(result_path + "_loss.png", {"Train Loss": llog})
        print()

    # ------------------------------------------------------------------ contrastive split
    def _sweep_preds(self, eng, X, batchsize=4096):
        """Eval-mode critic value of every frame (main.py:245-253), on the device in large batches; under data parallelism each
        rank sweeps a contiguous shard of the frames and the shards are all-gathered."""
        n = len(X)
        lo, hi = 0, n
        per = n
        if self.world > 1:
            per = -(-n // self.world)
            lo, hi = min(n, self.rank * per), min(n, (self.rank + 1) * per)
        out = torch.zeros(per, device=self.device)
        for b in range(lo, hi, batchsize):
            e = min(hi, b + batchsize)
            pred, _ = eng.infer(torch.from_numpy(X[b:e]).to(self.device), want_mask=False)
            out[b - lo:e - lo] = pred
        if self.world > 1:
            parts = [torch.zeros_like(out) for _ in range(self.world)]
            torch.distributed.all_gather(parts, out, group=self.pg)
            out = torch.cat(parts)[:n]
        return out.cpu()

    def extract_contrastive_data(self):
        """Splits the training frames by the critic's value into the high set (> --high-rew-thresh) and the low set
        (< --low-rew-thresh) (main.py:238-312), keeps both resident on the device as uint8 and sets up the reference's index
        sampler (32 high + 32 low frames for A, 64 low frames for B, drawn with replacement from the global numpy RNG)."""
        args = self.args
        self.critic.eval()
        eng = self._engine(2 * 32)
        if args.critic or args.cload:
            preds = self._sweep_preds(eng, self.X)
            positives, negatives = preds > args.high_rew_thresh, preds < args.low_rew_thresh
        else:
            print("no critic provided -> using random pos and neg frames")
            positives = torch.rand(len(self.X)) > 0.5
            if self.world > 1:                    # every rank must hold the same split (same set sizes, same number of steps)
                flag = positives.to(torch.uint8).to(self.device)
                torch.distributed.broadcast(flag, src=0, group=self.pg)
                positives = flag.cpu().bool()
            negatives = ~positives
            preds = torch.cat((positives, negatives), dim=0)
        npos, nneg = int(positives.sum()), int(negatives.sum())
        os.makedirs(self.path, exist_ok=True)
        if self.rank == 0:      # the reference leaves the two counts behind as an (empty) file name
            open(self.path + f"{npos}>{args.high_rew_thresh}__{nneg}<{args.low_rew_thresh}.txt", "w").close()
        assert npos >= 500 and nneg >= 500
        assert preds[positives].float().mean() > args.high_rew_thresh
        pos, neg = positives.numpy(), negatives.numpy()
        self.Xpos, self.Ypos = self.X[pos], self.Y[:, pos]
        self.Xneg, self.Yneg = self.X[neg], self.Y[:, neg]
        # device-resident copies: a training step then gathers its frames with one index upload, no host frames involved
        dev = self.device
        self._Xpos_d, self._Xneg_d = torch.from_numpy(self.Xpos).to(dev), torch.from_numpy(self.Xneg).to(dev)
        self._ypos_d = torch.from_numpy(np.ascontiguousarray(self.Ypos[args.rewidx])).float().to(dev)
        self._yneg_d = torch.from_numpy(np.ascontiguousarray(self.Yneg[args.rewidx])).float().to(dev)
        self.XposIdxs, self.XnegIdxs, self.ContrastIdxs = np.arange(npos), np.arange(nneg), np.arange(nneg)
        self.contrastive_batchsize = 32
        cb = self.contrastive_batchsize
        self.get_contrastive_idxs = lambda: (np.random.choice(self.XposIdxs, cb), np.random.choice(self.XnegIdxs, cb),
                                             np.random.choice(self.ContrastIdxs, 2 * cb))

    # ------------------------------------------------------------------ phase 2: mask training
    def segmentation_training(self):
        """main.py:314-575 without the debug image grids: per step three index draws (numpy RNG, as the reference), the two
        shift draws (torch RNG), ONE 128-entry index upload; the frames are gathered, rolled and trained on without leaving
        the device."""
        import time
        args = self.args
        self._refuse_unbuilt_flags()
        self.extract_contrastive_data()
        train_path = self.path + "segment/"
        os.makedirs(train_path, exist_ok=True)
        if self.rank == 0:
            with open(train_path + "log.txt", "w") as log_file:
                log_file.write(f"{self.args}\n\n")
        log = []
        self.critic.train()
        self.masker.train()
        n = 2 * self.contrastive_batchsize
        # -directeval (main.py:337-338) leaves the modules in eval mode (see critic_pipe): the training that follows runs without Dropout
        eng = self._engine(n, live=args.live, training=True, dropout=0.0 if (args.directeval and not args.noevalmode) else None)
        self._reset_adam()                       # a fresh Adam over critic+masker (live) or masker (frozen)
        if args.directeval:
            self.eval()
            if args.noevalmode:
                self.critic.train()
                self.masker.train()
        # The host runs ahead of the device (it only syncs every 10th step): ONE pinned buffer would be overwritten with later draws
        # while earlier asynchronous uploads are still queued, and several steps would train on the same (or a torn) index set.
        # A ring of pinned buffers, each reused only after ITS upload's event has completed, keeps one fresh draw per step.
        ring = 16
        idx_hosts = [torch.empty(2 * n, dtype=torch.int64).pin_memory() for _ in range(ring)]
        idx_events = [None] * ring
        idx_dev = torch.empty(2 * n, dtype=torch.int64, device=self.device)
        self._last_idx_draws = []                # (debug / tests) the host draws of the last few steps, in order
        names = ["replace", "inject", "norm", "live-critic"]
        steps, steps_t0, t0, dt = 0, 0, time.perf_counter(), 0.0
        for epoch in range(args.mepochs):
            for b_idx in range(math.ceil(self.Xpos.shape[0] / self.contrastive_batchsize)):
                Hidx, Lidx, Cidx = self.get_contrastive_idxs()
                slot = steps % ring
                if idx_events[slot] is not None:
                    idx_events[slot].synchronize()                   # the upload that last used this buffer has finished
                idx_host = idx_hosts[slot]
                idx_host.copy_(torch.from_numpy(np.concatenate((Hidx, Lidx, Cidx))))
                idx_dev.copy_(idx_host, non_blocking=True)
                idx_events[slot] = torch.cuda.Event()
                idx_events[slot].record()
                if getattr(self, "_record_idx_draws", False):
                    self._last_idx_draws.append(idx_host.clone())
                roll = self._shift_draw() if args.shift else 0      # torch.roll(X, roll, dims=2): dst[x] = src[x - roll]
                eng.gather_contrastive(self._Xpos_d, self._Xneg_d, self._ypos_d, self._yneg_d, idx_dev, shift_px=(-roll) % 64)
                losses = eng.phase2_step()
                steps += 1
                if self._trace is not None:      # (tests: the loop-level pin G9)
                    self._trace["p2_idx"].append(np.concatenate((Hidx, Lidx, Cidx)))
                    self._trace["p2_roll"].append(roll)
                    self._trace["p2_loss"].append(losses[:6].clone())
                if steps == 20:                                      # throughput is reported for the steady state (after the
                    torch.cuda.synchronize()                         # eager first step and the graph capture)
                    t0, steps_t0, dt = time.perf_counter(), steps, 0.0
                if not b_idx % 10:                                   # the only host sync
                    c, r, i, l1, l2, total = losses[:6].tolist()
                    log.append((r, i if args.inject else 0, l1 + l2, c if args.live else 0))
                    msg = f"e{epoch} b{b_idx}" + (f"    live-critic {c}" if args.live else "") + f"   replace: {r}"
                    msg += (f"   inject: {i}" if args.inject else "") + (f"   L1: {l1}" if args.L1 else "") + (f"   L2: {l2}" if args.L2 else "")
                    print(msg, end="\r")
            torch.cuda.synchronize()
            dt += time.perf_counter() - t0            # (plots and checkpoints are not part of the step throughput)
            if self.rank == 0:
                llog = np.array(log)
                self._plot(train_path + "_loss.png", {nm: llog[:, k] for k, nm in enumerate(names)})
            if not (epoch + 1) % args.saveevery:
                self.save_models(modelnames=[self.maskername])
            t0 = time.perf_counter()
        self.train_images_per_s = (steps - steps_t0) * n * self.world / dt if dt > 0 else 0.0
        print(f"\nmask training: {steps} steps of {n} A-images, {steps - steps_t0} of them in {dt:.3f} s = {self.train_images_per_s:.0f} images/s"
              + (f" over {self.world} ranks" if self.world > 1 else ""))
        self.save_models(modelnames=[self.maskername])

    # ------------------------------------------------------------------ -process: masks for a folder of images
    def segment(self, folder):
        from PIL import Image
        print("STARTING SEGMENTATION...")
        args = self.args
        os.makedirs(self.path, exist_ok=True)
        if args.crf:
            raise NotImplementedError("-crf is outside this build's scope")
        if args.noevalmode and args.salience:
            raise NotImplementedError("-noevalmode together with -salience (Dropout inside the saliency backward) is not implemented")
        if args.process_salience and not args.salience:
            raise ValueError("-process_salience needs -salience (the reference collects the maps only then, main.py:1136-1147)")
        files = os.listdir(folder)
        frames = np.stack([np.array(Image.open(os.path.join(folder, f)))[..., :3] for f in files]) / 255.0     # NHWC float64 in [0,1]
        stems = [f.rsplit(".", 1)[0] for f in files if "." in f]
        fp16 = bool(getattr(args, "fp16", False))      # (this build's switch) BASELINE config 4: fp16 layers on the uint8 frames

        def to_device(chunk):
            t = torch.from_numpy(chunk).float().to(self.device)
            return (t * 255.0).round().to(torch.uint8) if fp16 else t

        preds, M, sal = self._sweep_masks(frames, to_device, "segmentation in progress", want_saliency=bool(args.salience), fp16=fp16)
        print()
        print("postprocessing...")
        # one column per output kind, in the reference's order; the file name of a column is its POSITION in `kinds` (main.py:1212-1223)
        cols = [M]
        if args.binarymaskthreshold:
            cols.append(M >= args.binarymaskthreshold)
        if args.process_salience:       # main.py:1176-1197
            cols += list(self._saliency_post(sal, preds, args.salience_thresh, args.salglobal))
        kinds = ("raw-mask", "thresholded-mask", "crf-mask", "saliency-map", "thresholded-saliency", "crf-saliency")
        out_dir = args.mask_output_imgs
        os.makedirs(out_dir, exist_ok=True)
        to_u8 = lambda a: (a * 255).astype(np.uint8)
        grey_rgb = [np.repeat(c, 3, axis=1).transpose(0, 2, 3, 1) for c in cols]          # [n,1,64,64] -> [n,64,64,3]
        for i, stem in enumerate(stems[:len(frames)]):
            if args.concatenated:       # frame | column 1 | column 2 ... side by side
                strip = np.concatenate([to_u8(frames[i])] + [to_u8(g[i]) for g in grey_rgb], axis=1)
                Image.fromarray(strip).save(f"{out_dir}/{stem}_with_mask.png")
            else:
                for kind, g in zip(kinds, grey_rgb):
                    Image.fromarray(to_u8(g[i])).save(f"{out_dir}/{stem}-{kind}.png")
        return M

    def _sweep_masks(self, X, to_device, progress, want_saliency=False, fp16=False, batchsize=128):
        """The inference loop shared by -process and -eval (main.py:1130-1151, 900-953): eval-mode critic + masker over X in batches
        of 128, optionally the saliency baseline |d mean(pred) / d batch| summed over the colour channels.
        Returns (preds [n], masks [n,1,64,64], saliency [n,1,64,64] or None) as numpy."""
        args = self.args
        self.critic.eval()
        self.masker.eval()
        eng = self._engine(2 * 32)
        preds, masks, sal = [], [], []
        for lo in range(0, len(X), batchsize):
            print(progress, round(lo / len(X), 2), end="%\r")
            batch = to_device(X[lo:lo + batchsize])
            if want_saliency:
                _unused, dx = eng.saliency(batch)
                sal.append(dx.abs().sum(dim=-1)[:, None].cpu().numpy())
            if fp16:
                pred, Z = eng.infer(batch, fp16=True)
            else:
                pred, Z = eng.infer(batch, train_mode=bool(args.noevalmode))      # -noevalmode: Dropout stays on (main.py:1109-1118)
            preds.append(pred.cpu().numpy())
            masks.append(Z.cpu().numpy()[:, None])
        cat = lambda parts: np.concatenate(parts, axis=0)
        return cat(preds), cat(masks), (cat(sal) if want_saliency else None)

    @staticmethod
    def _saliency_post(salM, preds, thresh, salglobal):
        """main.py:976-1003 / 1176-1197: normalise the |gradient| maps (by the global mean of the non-negative part x thresh, or by
        each map's k-th smallest value), weight them by the critic's prediction, clip at 1, threshold.  Returns (maps, hard uint8)."""
        tiny = np.finfo(np.float64).tiny                      # (= sys.float_info.min: keeps 0 / 0 finite)
        if salglobal:
            scale = np.where(salM >= 0, salM, 0.0).mean() * thresh
        else:
            n, hw = salM.shape[0], salM.shape[-1] * salM.shape[-2]
            scale = np.sort(salM.reshape(n, 1, hw), axis=-1)[:, :, int(hw * thresh), None, None]
        out = np.minimum(salM / (scale + tiny) * preds.reshape(-1, 1, 1, 1), 1.0)
        return out, (out > thresh).astype(np.uint8)

    # ------------------------------------------------------------------ -eval: IoU on the labelled red-trees set
    @staticmethod
    def get_iou(A, B):
        """Intersection over union of two boolean stacks taken over the WHOLE set, 3 digits (main.py:1265-1270)."""
        A, B = np.asarray(A, dtype=bool), np.asarray(B, dtype=bool)
        both, either = np.count_nonzero(A & B), np.count_nonzero(A | B)
        return round(both / either, 3) if either else float("nan")

    def eval(self, folder="", vis=False):
        """main.py:891-1020 without CRF / videos: masks of `red-trees/X.npy[100:5000:2]` (batch 128, eval mode), thresholded
        at --eval-thresh, IoU against `all(Y.npy, axis=-1)`; with -salience also the saliency baseline of main.py:941-953,
        976-1003 (|d mean(pred)/dX| summed over channels, normalised, weighted by pred, thresholded) and its IoU.
        Returns [iou] or [iou, saliou] like the reference."""
        args = self.args
        if args.noevalmode and args.salience:
            raise NotImplementedError("-noevalmode together with -salience (Dropout inside the saliency backward) is not implemented")
        if args.crf or args.resimages or folder or vis:
            raise NotImplementedError("-crf / -resimages / folder / video evaluation are outside this build's scope")
        pick = slice(100, 5000, 2)                                        # the reference's evaluation subset
        frames = np.load("red-trees/X.npy")[pick]                         # uint8 [n,64,64,3] (the reference divides by 255 here)
        truth = np.load("red-trees/Y.npy")[pick].all(axis=-1)             # [n,64,64] bool: all three label channels set
        want_sal = bool(args.salience)

        def to_device(chunk):
            t = torch.from_numpy(np.ascontiguousarray(chunk)).to(self.device)
            # uint8 frames go to the kernels as they are (/255 fused); the saliency backward needs the float batch (main.py:921,939)
            return (t.double() / 255.0).float() if (t.dtype != torch.uint8 or want_sal) else t

        preds, M, sal = self._sweep_masks(frames, to_device, "eval at", want_saliency=want_sal)
        ious = [self.get_iou(M[:, 0] > args.eval_thresh, truth)]
        if want_sal:
            _maps, hard = self._saliency_post(sal, preds, args.salience_thresh, args.salglobal)
            ious.append(self.get_iou(hard[:, 0], truth))
        print("\nRESULTS", ious)
        return ious

    # ------------------------------------------------------------------ helpers
    @staticmethod
    def _plot(path, series):
        try:
            import matplotlib
            matplotlib.use("Agg")
            from matplotlib import pyplot as plt
        except Exception:
            return
        plt.clf()
        for name, vals in series.items():
            vals = np.asarray(vals, dtype=np.float64)
            if len(vals) == 0:
                continue
            k = min(30, len(vals))
            avg = np.convolve(vals, np.ones(k) / k, mode="valid")
            plt.plot(avg, label=name)
        plt.legend()
        plt.savefig(path)
