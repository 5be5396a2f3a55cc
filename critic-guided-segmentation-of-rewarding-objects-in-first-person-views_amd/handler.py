"""Host-side orchestration of the ``-train`` and ``-process`` paths: the reference's ``Handler`` surface
(main.py:66-156, 158-236, 238-312, 314-575, 584-591, 1103-1223) with its inner loops replaced by the fused
HIP engine.  Same method names, same checkpoint / dataset file naming, same output file names.

Out of scope here: MineRL collection (``collect_data`` only reads an existing
gz-pickle), CRF, videos / PNG debug grids, the ``-eval`` IoU path (section 8 f2).
"""
import gzip
import math
import os
import pickle
import sys
from itertools import chain

import numpy as np
import torch

from . import _lib, dataformat
from .engine import HourglassEngine
from .nets import NewCritic, UnetDecoder


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
        self.device = "cuda"
        print("device:", self.device)
        self.models = dict()
        self.criticname = "critic"
        self.maskername = "masker"
        self.ious = 0, 0
        self.bestepoch = 0
        self.reset_models()
        self.models[self.criticname] = self.critic
        self.models[self.maskername] = self.masker
        self.critic_args, self.masker_args = checkpoint_names(args)
        self.path = f"{args.name}/"
        self.train_path = self.path + "train/"
        self.result_path = self.path + "results/"
        self.save_path = self.path + "saves/"
        self.data_path = "runs/data/straight/"
        self.save_paths = {
            self.criticname: f"{self.save_path}critic-{self.critic_args}.pt",
            self.maskername: f"{self.save_path}masker-{self.masker_args}.pt",
        }
        self._engines = {}

    # ------------------------------------------------------------------ models / checkpoints
    def reset_models(self):
        args = self.args
        if args.separate:
            raise NotImplementedError("-separate (second critic) is not implemented on the HIP path")
        self.critic = NewCritic(bottleneck=args.neck, chfak=args.chfak, dropout=args.dropout).to(self.device)
        self.masker = UnetDecoder(bottleneck=args.neck, chfak=args.chfak).to(self.device)

    def load_models(self, modelnames=[]):
        if not modelnames:
            modelnames = self.models.keys()
        for model in modelnames:
            save_path = self.save_paths[model]
            if not os.path.exists(save_path):
                if not self.args.train:
                    print(f"{save_path} not found")
                return False
            print("loading:", save_path)
            self.models[model].load_state_dict(torch.load(save_path, map_location=torch.device(self.device)))
        return True

    def save_models(self, modelnames=[]):
        os.makedirs(self.save_path, exist_ok=True)
        if not modelnames:
            modelnames = self.models.keys()
        for model in modelnames:
            save_path = self.save_paths[model]
            print("saving:", save_path)
            torch.save(self.models[model].state_dict(), save_path)

    # ------------------------------------------------------------------ data
    def collect_data(self):
        args = self.args
        filepath = dataformat.dataset_path(args.envname, args.datamode, args.datasize, args.gammas, self.data_path)
        print("collecting dataset at", filepath)
        if not os.path.exists(filepath):
            raise FileNotFoundError(
                f"{filepath} not found. This build reads the reference's gz-pickle (X uint8 [N,64,64,3], Y float [7,N], "
                "I uint16 [N]) but does not collect it: MineRL download/decoding is out of scope (SURVEY.md 2.3).")
        print("loading existing dataset...")
        X, Y, I = dataformat.read_dataset(filepath)
        print("finished loading exisiting dataset")
        return X, Y, I

    def load_data(self, batch_size=64):
        args = self.args
        X, Y, I = self.collect_data()
        train = slice(0, -args.testsize)
        test = slice(-args.testsize, None)
        self.X, self.Y, self.I = X[train], Y[:, train], I[train]
        self.XX, self.YY, self.II = X[test], Y[:, test], I[test]
        if args.threshrew:
            self.Y = (self.Y > args.threshrew).astype(np.float64)
            self.YY = (self.YY > args.threshrew).astype(np.float64)
        print("dataset shapes", X.shape, Y.shape, self.X.shape, self.Y.shape)
        self.batch_size = batch_size

    def _batches(self):
        """Shuffled mini-batches of (X uint8, Y[rewidx]) like the reference's DataLoader(shuffle=True)."""
        n = len(self.X)
        perm = torch.randperm(n).numpy()
        for b in range(0, n, self.batch_size):
            idx = np.sort(perm[b:b + self.batch_size])
            yield torch.from_numpy(self.X[idx]), torch.from_numpy(self.Y[self.args.rewidx, idx]).float()

    def shift_batch(self, X):
        """Whole-batch circular roll along width, two draws from the torch RNG (main.py:584-591)."""
        xshift = int(self.args.shift * torch.rand(1))
        if torch.rand(1) > 0.5:
            X = torch.cat((X[:, :, xshift:], X[:, :, :xshift]), dim=2)
        else:
            X = torch.cat((X[:, :, -xshift:], X[:, :, :-xshift]), dim=2)
        return X

    # ------------------------------------------------------------------ engines
    def _engine(self, n, live=True):
        key = (n, live)
        if key not in self._engines:
            a = self.args
            first = next(iter(self._engines.values()), None)
            e = HourglassEngine(n, device=self.device, dropout=a.dropout, lfak=a.lfak, L1=a.L1, L2=a.L2, inject=a.inject,
                                live=live, threshrew=a.threshrew, share_with=first)
            if first is None:
                e.adopt(self.critic, self.masker)   # modules and engine share one parameter buffer from now on
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
        if not a.staticnorm:
            raise NotImplementedError("-staticnorm '' (mask regulariser weighted by 1 - pred, main.py:415-418) is not implemented "
                                      "by the HIP loss kernels; the engine uses valuefak = 1")

    def critic_pipe(self, mode="train", test=0):
        args = self.args
        self._refuse_unbuilt_flags()
        if args.cload and self.load_models([self.criticname]):
            print("loaded critic, no new training")
            return
        result_path = self.path + "critic/"
        os.makedirs(result_path, exist_ok=True)
        with open(result_path + "log.txt", "w") as log_file:
            log_file.write(f"{self.args}\n\n")
        llog = []
        self.critic.train()
        self._engine(self.batch_size)
        self._reset_adam()                       # a fresh torch.optim.Adam(critic.parameters()) (main.py:178)
        if args.directeval:                      # main.py:179-180
            self.eval()
            self.critic.train()
        for epoch in range(int(mode == "test") or args.cepochs):
            for b_idx, (X, Y) in enumerate(self._batches()):
                if args.shift:
                    X = self.shift_batch(X)
                eng = self._engine(len(X))
                losses = eng.phase1_step(X.contiguous().to(self.device, non_blocking=True), Y.to(self.device, non_blocking=True))
                if not b_idx % 10:
                    val = float(losses[0])       # the only host sync, every 10th batch
                    llog.append(val)
                    print(f"critic e{epoch + 1} b{b_idx}", val, end="\r")
            if not (epoch + 1) % args.saveevery:
                self.save_models(modelnames=[self.criticname])
            self._plot(result_path + "_loss.png", {"Train Loss": llog})
        print()

    # ------------------------------------------------------------------ contrastive split
    def extract_contrastive_data(self):
        args = self.args
        self.critic.eval()
        eng = self._engine(2 * 32)
        batchsize = 4096
        if args.critic or args.cload:
            preds = []
            for b in range(0, len(self.X), batchsize):
                xb = torch.from_numpy(self.X[b:b + batchsize]).to(self.device)
                pred, _ = eng.infer(xb, want_mask=False)
                preds.append(pred.cpu())
            preds = torch.cat(preds, dim=0)
            positives = preds > args.high_rew_thresh
            negatives = preds < args.low_rew_thresh
        else:
            print("no critic provided -> using random pos and neg frames")
            positives = torch.rand(len(self.X)) > 0.5
            negatives = positives == False  # noqa: E712
            preds = torch.cat((positives, negatives), dim=0)
        os.makedirs(self.path, exist_ok=True)
        with open(self.path + f"{positives.sum()}>{args.high_rew_thresh}__{negatives.sum()}<{args.low_rew_thresh}.txt", "w") as fp:
            fp.write("")
        assert (sum(positives) >= 500 and sum(negatives) >= 500)
        positives, negatives = positives.numpy(), negatives.numpy()
        self.Xpos, self.Ypos = self.X[positives], self.Y[:, positives]
        self.Xneg, self.Yneg = self.X[negatives], self.Y[:, negatives]
        assert (preds[torch.from_numpy(positives)].float().mean()) > args.high_rew_thresh
        self.XposIdxs = np.arange(len(self.Xpos))
        self.XnegIdxs = np.arange(len(self.Xneg))
        self.ContrastIdxs = np.arange(len(self.Xneg))
        self.contrastive_batchsize = 32
        self.get_contrastive_idxs = lambda: (np.random.choice(self.XposIdxs, self.contrastive_batchsize),
                                             np.random.choice(self.XnegIdxs, self.contrastive_batchsize),
                                             np.random.choice(self.ContrastIdxs, 2 * self.contrastive_batchsize))

    # ------------------------------------------------------------------ phase 2: mask training
    def segmentation_training(self):
        args = self.args
        self._refuse_unbuilt_flags()
        self.extract_contrastive_data()
        train_path = self.path + "segment/"
        os.makedirs(train_path, exist_ok=True)
        with open(train_path + "log.txt", "w") as log_file:
            log_file.write(f"{self.args}\n\n")
        log = []
        self.critic.train()
        self.masker.train()
        n = 2 * self.contrastive_batchsize
        eng = self._engine(n, live=args.live)
        self._reset_adam()                       # a fresh Adam over critic+masker (live) or masker (frozen)
        if args.directeval:                      # main.py:337-338
            self.eval()
            self.critic.train()
            self.masker.train()
        for epoch in range(args.mepochs):
            for b_idx in range(math.ceil(self.Xpos.shape[0] / self.contrastive_batchsize)):
                Hidx, Lidx, Cidx = self.get_contrastive_idxs()
                X = torch.cat((torch.from_numpy(self.Xpos[Hidx]), torch.from_numpy(self.Xneg[Lidx])), dim=0)
                Y = torch.cat((torch.from_numpy(self.Ypos[args.rewidx, Hidx]), torch.from_numpy(self.Yneg[args.rewidx, Lidx])), dim=0)
                CX = torch.from_numpy(self.Xneg[Cidx])
                if args.shift:
                    X = self.shift_batch(X)
                losses = eng.phase2_step(X.contiguous().to(self.device, non_blocking=True),
                                         CX.to(self.device, non_blocking=True), Y.float().to(self.device, non_blocking=True))
                if not b_idx % 10:
                    c, r, i, l1, l2, total = losses[:6].tolist()
                    log.append((r, i if args.inject else 0, l1 + l2, c if args.live else 0))
                    s = f"e{epoch} b{b_idx}"
                    if args.live:
                        s += f"    live-critic {c}"
                    s += f"   replace: {r}"
                    if args.inject:
                        s += f"   inject: {i}"
                    if args.L1:
                        s += f"   L1: {l1}"
                    if args.L2:
                        s += f"   L2: {l2}"
                    print(s, end="\r")
            llog = np.array(log)
            self._plot(train_path + "_loss.png", {nm: llog[:, k] for k, nm in enumerate(["replace", "inject", "norm", "live-critic"])})
            if not (epoch + 1) % args.saveevery:
                self.save_models(modelnames=[self.maskername])
        print()
        self.save_models(modelnames=[self.maskername])

    # ------------------------------------------------------------------ -process: masks for a folder of images
    def segment(self, folder):
        from PIL import Image
        print("STARTING SEGMENTATION...")
        args = self.args
        os.makedirs(self.path, exist_ok=True)
        if args.noevalmode:
            raise NotImplementedError("-noevalmode (dropout at inference) is not implemented on the HIP path")
        if args.crf:
            raise NotImplementedError("-crf is outside this build's scope")
        if args.process_salience and not args.salience:
            raise ValueError("-process_salience needs -salience (the reference collects the maps only then, main.py:1136-1147)")
        self.critic.eval()
        self.masker.eval()
        eng = self._engine(2 * 32)
        batchsize = 128
        img_names = os.listdir(folder)
        X = np.stack([np.array(Image.open(f"{folder}/{name}"))[..., :3] for name in img_names]) / 255.0
        img_names = [a[:-1 - a[::-1].index(".")] for a in img_names if "." in a]
        M, preds, salM = [], [], []
        for bidx in range(0, len(X), batchsize):
            print("segmentation in progress", round(bidx / len(X), 2), end="%\r")
            batch = torch.from_numpy(X[bidx:bidx + batchsize]).float().to(self.device)   # NHWC fp32 in [0,1]
            if args.salience:           # main.py:1136-1147: |d mean(pred) / d batch| summed over the colour channels
                _p, dx = eng.saliency(batch)
                salM.append(dx.abs().sum(dim=-1)[:, None].cpu().numpy())
            pred, Z = eng.infer(batch)
            preds.append(pred.cpu().numpy())
            M.append(Z.cpu().numpy()[:, None])
        print()
        print("postprocessing...")
        M = np.concatenate(M, axis=0)
        preds = np.concatenate(preds, axis=0)
        allM = [M]
        if args.binarymaskthreshold:
            allM.append(M >= args.binarymaskthreshold)
        if args.process_salience:       # main.py:1176-1197 (file names follow the reference's column list by POSITION)
            allM.extend(self._saliency_post(np.concatenate(salM, axis=0), preds, args.salience_thresh, args.salglobal))
        outpath = args.mask_output_imgs
        os.makedirs(outpath, exist_ok=True)
        masks = np.stack([X] + [np.concatenate((m, m, m), axis=1).transpose(0, 2, 3, 1) for m in allM], axis=1)
        columns = ["raw-mask", "thresholded-mask", "crf-mask", "saliency-map", "thresholded-saliency", "crf-saliency"]
        for fidx in range(masks.shape[0]):
            if args.concatenated:
                array = np.concatenate((masks[fidx] * 255).astype(np.uint8), axis=-2)
                Image.fromarray(array).save(f"{outpath}/{img_names[fidx]}_with_mask.png")
            else:
                for midx in range(1, masks.shape[1]):
                    Image.fromarray((masks[fidx, midx] * 255).astype(np.uint8)).save(
                        f"{outpath}/{img_names[fidx]}-{columns[midx - 1]}.png")
        return M

    @staticmethod
    def _saliency_post(salM, preds, thresh, salglobal):
        """main.py:976-1003 / 1176-1197: normalise the |gradient| maps (global mean x thresh, or each map's k-th sorted
        value), weight by the critic's prediction, clip at 1, threshold.  Returns (salM, salhardM uint8)."""
        import sys as _sys
        if salglobal:
            norm = (salM * (salM >= 0)).mean() * thresh
        else:
            k = int(salM.shape[-1] * salM.shape[-2] * thresh)
            norm = np.sort(salM.reshape(salM.shape[0], 1, -1), axis=-1)[:, :, k, None, None]
        salM = salM / (norm + _sys.float_info.min)
        salM = salM * preds[:, None, None, None]
        salM[(salM >= 1)] = 1
        return salM, (salM > thresh).astype(np.uint8)

    # ------------------------------------------------------------------ -eval: IoU on the labelled red-trees set
    @staticmethod
    def get_iou(A, B):
        """main.py:1265-1270: |A & B| / |A | B| over the whole set, rounded to 3 digits."""
        intersection = np.sum(A & B)
        union = np.sum(A | B)
        return round(float(intersection / union), 3) if union else float("nan")

    def eval(self, folder="", vis=False):
        """main.py:891-1020 without CRF / videos: masks of `red-trees/X.npy[100:5000:2]` (batch 128, eval mode), thresholded
        at --eval-thresh, IoU against `all(Y.npy, axis=-1)`; with -salience also the saliency baseline of main.py:941-953,
        976-1003 (|d mean(pred)/dX| summed over channels, normalised, weighted by pred, thresholded) and its IoU.
        Returns [iou] or [iou, saliou] like the reference."""
        args = self.args
        if args.noevalmode:
            raise NotImplementedError("-noevalmode (dropout at inference) is not implemented on the HIP path")
        if args.crf or args.resimages or folder or vis:
            raise NotImplementedError("-crf / -resimages / folder / video evaluation are outside this build's scope")
        evaldatapath = "red-trees/"
        X = np.load(evaldatapath + "X.npy")                       # uint8 [n,64,64,3] (the reference divides by 255 here)
        Y = np.expand_dims(np.all(np.load(evaldatapath + "Y.npy"), axis=-1), axis=-1)
        X = X[100:5000:2]
        Y = Y[100:5000:2]
        self.critic.eval()
        self.masker.eval()
        eng = self._engine(2 * 32)
        batchsize = 128
        M, salM, preds = [], [], []
        for bidx in range(0, len(X), batchsize):
            print("eval at", bidx / len(X), end="\r")
            xb = X[bidx:bidx + batchsize]
            batch = torch.from_numpy(np.ascontiguousarray(xb)).to(self.device)
            if batch.dtype != torch.uint8 or args.salience:
                batch = (batch.double() / 255.0).float()            # main.py:921,939 (float64 / 255 -> float32)
            if args.salience:
                _p, dx = eng.saliency(batch)
                salM.append(dx.abs().sum(dim=-1)[:, None].cpu().numpy())
            pred, Z = eng.infer(batch)
            preds.append(pred.cpu().numpy())
            M.append(Z.cpu().numpy()[:, None])
        M = np.concatenate(M, axis=0)
        preds = np.concatenate(preds, axis=0)
        hardM = M > args.eval_thresh
        Yc = Y.transpose(0, 3, 1, 2)
        ious = [self.get_iou(hardM.squeeze(), Yc.squeeze())]
        if args.salience:
            salM, salhardM = self._saliency_post(np.concatenate(salM, axis=0), preds, args.salience_thresh, args.salglobal)
            ious.append(self.get_iou(salhardM.squeeze(), Yc.squeeze()))
        print(f"\nRESULTS", ious)
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
