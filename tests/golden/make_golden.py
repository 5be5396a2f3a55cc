#!/usr/bin/env python3
"""Generates the golden fixtures in this directory from the REFERENCE implementation.

Run in the build container only (the reference does not exist on the GPU box):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden.py

It imports ``/root/reference/nets.py`` (the reference's NewCritic / UnetDecoder classes) with the
two in-process shims SURVEY.md section 8c names (empty ``torchvision`` stub modules, ``np.int = int``)
and drives them with ``torch.optim.Adam`` / ``torch.nn.functional`` losses arranged the way
``main.py`` arranges them (main.py itself cannot be imported: minerl/cv2/ffmpeg are absent).
Only DATA (inputs, weights, outputs) is written; no reference source is copied.

Fixtures (all float32 unless noted):
  g1_weights_chfak1.npz  default-init state_dicts under torch.manual_seed(0)
  g1_keys.json           state_dict key -> shape for chfak 1 and 5
  g2_eval.npz            eval-mode forward: pred, 5 embeds, decoder intermediates, mask
  g2_eval_chfak5.npz     same for the paper-size model (weights = oracle.seeded_params seeds)
  g3_train_*.npz         phase-2 step, dropout=0: losses, grads, params after steps 1..3
                         (default / -noinject / -frozen / --L2 0.1 / --threshrew 0.5 = BCE live-critic loss, main.py:380-381 /
                         -separate = a second critic feeds the masker, main.py:110-111,389-390, live and frozen)
  g4_phase1_*.npz        phase-1 step: loss, grads, params after step 1 (mse and bce variants)
  g5_shift.npz           shift_batch under torch.manual_seed(k)
  g7_dropout.npz         phase-2 step with dropout 0.3 and the recorded keep-masks
  g8_unet_convt.npz      legacy Unet(upsample=False) (ConvTranspose2d decoder, LeakyReLU(0.2)): weights, mask, u0, critic value
  g8_unet_train.npz      the same class, one training step: loss = MSE(mask, target) + MSE(critic, target), all 24 parameter gradients
  g6_process.npz         (make_golden_g6.py) the PNG outputs of the reference's own `main.py -process` CLI run end to end
"""
import json
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

import numpy as np
import torch
import torch.nn.functional as F

np.int = int  # shim 1: nets.py:456-457 uses np.int (removed in numpy >= 1.24)
for name in ("torchvision", "torchvision.models"):  # shim 2: nets.py:5 imports torchvision
    sys.modules.setdefault(name, types.ModuleType(name))
sys.path.insert(0, REF)
import nets as refnets  # noqa: E402  (the reference)

sys.path.insert(0, REPO)
from oracle import hourglass_ref as orc  # noqa: E402  (only for seeded_params)

torch.set_num_threads(1)
torch.use_deterministic_algorithms(True)


def sd_np(prefix, sd):
    return {f"{prefix}/{k}": v.detach().cpu().numpy().copy() for k, v in sd.items()}


def frames(seed, n):
    return np.random.RandomState(seed).randint(0, 256, (n, 64, 64, 3)).astype(np.uint8)


def to_nchw(x_u8):
    return torch.from_numpy(x_u8).permute(0, 3, 1, 2).float() / 255.0


def build(chfak=1, dropout=0.3, seed=0):
    torch.manual_seed(seed)
    critic = refnets.NewCritic(bottleneck=32, chfak=chfak, dropout=dropout)
    masker = refnets.UnetDecoder(bottleneck=32, chfak=chfak)
    return critic, masker


def load_np(module, prefix, arrays):
    sd = {k[len(prefix) + 1:]: torch.from_numpy(v) for k, v in arrays.items() if k.startswith(prefix + "/")}
    module.load_state_dict(sd)


# ---------------------------------------------------------------- G1
critic, masker = build(1)
g1 = {}
g1.update(sd_np("critic", critic.state_dict()))
g1.update(sd_np("masker", masker.state_dict()))
np.savez(os.path.join(HERE, "g1_weights_chfak1.npz"), **g1)
keys = {}
for cf in (1, 5):
    c, m = build(cf)
    keys[f"chfak{cf}"] = {
        "critic": {k: list(v.shape) for k, v in c.state_dict().items()},
        "masker": {k: list(v.shape) for k, v in m.state_dict().items()},
    }
with open(os.path.join(HERE, "g1_keys.json"), "w") as fp:
    json.dump(keys, fp, indent=1)


# ---------------------------------------------------------------- G2
def eval_forward(critic, masker, x_u8):
    critic.eval(); masker.eval()
    inter = {}
    hooks = []
    for i, mod in enumerate(masker.dec):
        hooks.append(mod.register_forward_hook(lambda m, a, o, i=i: inter.__setitem__(f"o{i}", o.detach().numpy().copy())))
    hooks.append(masker.masker[1].register_forward_hook(lambda m, a, o: inter.__setitem__("hm", o.detach().numpy().copy())))
    with torch.no_grad():
        X = to_nchw(x_u8)
        pred, embeds = critic(X, collect=True)
        Z = masker(X, embeds)
    for h in hooks:
        h.remove()
    out = {"X": x_u8, "pred": pred.numpy(), "Z": Z.numpy()}
    for i, e in enumerate(embeds):
        out[f"e{i}"] = e.numpy()
    out.update(inter)
    return out


critic, masker = build(1)
load_np(critic, "critic", g1); load_np(masker, "masker", g1)
g2 = eval_forward(critic, masker, frames(0, 8))
g2["hm"] = g2["hm"][:2]
np.savez(os.path.join(HERE, "g2_eval.npz"), **g2)

c5, m5 = build(5)
c5.load_state_dict(orc.seeded_params(orc.critic_shapes(5), 11))
m5.load_state_dict(orc.seeded_params(orc.masker_shapes(5), 12))
g25 = eval_forward(c5, m5, frames(0, 4))
np.savez(os.path.join(HERE, "g2_eval_chfak5.npz"), X=g25["X"], pred=g25["pred"], Z=g25["Z"], e4=g25["e4"])


# ---------------------------------------------------------------- G3 / G7: phase-2 steps
def phase2(tag, steps=3, dropout=0.0, lfak=5, L1=0.5, L2=0.0, inject=True, live=True, record_masks=False, threshrew=0.0,
           separate=False, staticnorm=True, chfak=1):
    from itertools import chain
    critic, masker = build(chfak, dropout=dropout)
    if chfak == 1:
        load_np(critic, "critic", g1); load_np(masker, "masker", g1)
    else:       # other model sizes: the oracle's seeded stand-in weights (as g2_eval_chfak5)
        critic.load_state_dict(orc.seeded_params(orc.critic_shapes(chfak), 21))
        masker.load_state_dict(orc.seeded_params(orc.masker_shapes(chfak), 22))
    critic.train(); masker.train()
    sepcrit = None
    if separate:        # main.py:110-111: a second NewCritic; its embeds of A feed the masker (main.py:389-390)
        torch.manual_seed(77)
        sepcrit = refnets.NewCritic(bottleneck=32, chfak=1, dropout=dropout)
        sepcrit.train()
    sp = sepcrit.parameters() if separate else []
    opti = torch.optim.Adam(chain(critic.parameters(), masker.parameters(), sp)) if live \
        else torch.optim.Adam(chain(masker.parameters(), sp))
    a_u8, b_u8 = frames(0, 8), frames(1, 8)
    Y = torch.from_numpy(np.random.RandomState(2).rand(8)).float()
    if threshrew:       # load_data binarises the targets (main.py:124-127), the step then uses BCE (main.py:380-381)
        Y = (Y > threshrew).float()
    out = {"A": a_u8, "B": b_u8, "Y": Y.numpy()}
    if separate:
        out.update(sd_np("sepcrit0", sepcrit.state_dict()))
    masks = []
    orig_dropout = F.dropout
    if record_masks:
        def rec_dropout(x, p=0.5, training=True, inplace=False):
            if not training or p == 0:
                return x
            keep = (torch.rand_like(x) >= p).float()
            masks.append(keep.numpy().copy())
            return x * keep / (1.0 - p)
        F.dropout = rec_dropout
        torch.manual_seed(1234)
    try:
        for s in range(steps):
            A, B = to_nchw(a_u8), to_nchw(b_u8)
            pred, embeds = critic(A, collect=True)
            negpred = critic(B)
            pred = pred.squeeze(); negpred = negpred.squeeze().detach()
            loss = 0
            parts = np.zeros(5, np.float64)  # critic, replace, inject, L1, L2
            if live:
                cl = F.binary_cross_entropy(pred, Y) if threshrew else F.mse_loss(pred, Y)
                loss = loss + lfak * cl; parts[0] = cl.item()
            if separate:
                _, embeds = sepcrit(A, collect=True)
            Z = masker(A, embeds)
            replaced = A * (1 - Z) + Z * B
            rl = F.mse_loss(critic(replaced).squeeze(), negpred.detach())
            loss = loss + rl; parts[1] = rl.item()
            if inject:
                injected = B * (1 - Z) + Z * A
                il = F.mse_loss(critic(injected).squeeze(), pred.detach())
                loss = loss + il; parts[2] = il.item()
            valuefak = 1 if staticnorm else 1 - pred.detach().view(-1, 1, 1, 1)     # main.py:415-418
            if L1:
                nl = L1 * F.l1_loss(valuefak * Z, torch.zeros_like(Z))
                loss = loss + nl; parts[3] = nl.item()
            if L2:
                nl2 = L2 * F.mse_loss(valuefak * Z, torch.zeros_like(Z))
                loss = loss + nl2; parts[4] = nl2.item()
            opti.zero_grad()
            loss.backward()
            if s == 0:
                out["Z0"] = Z.detach().numpy().copy()
                out["pred0"] = pred.detach().numpy().copy()
                for k, v in critic.named_parameters():
                    if v.grad is not None:
                        out[f"grad/critic/{k}"] = v.grad.numpy().copy()
                for k, v in masker.named_parameters():
                    if v.grad is not None:
                        out[f"grad/masker/{k}"] = v.grad.numpy().copy()
                if separate:
                    for k, v in sepcrit.named_parameters():
                        if v.grad is not None:
                            out[f"grad/sepcrit/{k}"] = v.grad.numpy().copy()
            opti.step()
            out[f"total{s}"] = np.float64(loss.item())
            out[f"parts{s}"] = parts
            if s in (0, steps - 1) and (chfak == 1 or s == steps - 1):
                out.update(sd_np(f"step{s + 1}/critic", critic.state_dict()))
                out.update(sd_np(f"step{s + 1}/masker", masker.state_dict()))
                if separate:
                    out.update(sd_np(f"step{s + 1}/sepcrit", sepcrit.state_dict()))
    finally:
        F.dropout = orig_dropout
    for i, mk in enumerate(masks):
        out[f"mask{i:02d}"] = mk.astype(np.uint8)
    np.savez(os.path.join(HERE, f"{tag}.npz"), **out)


phase2("g3_train_default")
phase2("g3_train_noinject", inject=False)
phase2("g3_train_frozen", live=False)
phase2("g3_train_l2", L2=0.1)
phase2("g3_train_bce", threshrew=0.5)
phase2("g3_train_valuefak", staticnorm=False, L2=0.1)
phase2("g3_train_separate", separate=True)
phase2("g3_train_separate_frozen", separate=True, live=False)
phase2("g7_dropout", steps=1, dropout=0.3, record_masks=True)
phase2("g3_train_chfak2", steps=2, chfak=2)
phase2("g3_train_chfak2_valuefak", steps=2, chfak=2, staticnorm=False, L2=0.1)   # -staticnorm '' on the generic kernels      # a model size the specialised kernels do not cover (generic kernels' training pass)


# ---------------------------------------------------------------- G4: phase-1 step
def phase1(tag, bce=False):
    critic, _ = build(1, dropout=0.0)
    load_np(critic, "critic", g1)
    critic.train()
    opti = torch.optim.Adam(critic.parameters())
    x_u8 = frames(3, 8)
    Y = torch.from_numpy(np.random.RandomState(4).rand(8)).float()
    if bce:
        Y = (Y > 0.5).float()
    out = {"X": x_u8, "Y": Y.numpy()}
    pred = critic(to_nchw(x_u8)).squeeze()
    loss = F.binary_cross_entropy(pred, Y) if bce else F.mse_loss(pred, Y)
    opti.zero_grad(); loss.backward()
    out["loss"] = np.float64(loss.item()); out["pred"] = pred.detach().numpy()
    for k, v in critic.named_parameters():
        out[f"grad/{k}"] = v.grad.numpy().copy()
    opti.step()
    out.update(sd_np("step1", critic.state_dict()))
    np.savez(os.path.join(HERE, f"{tag}.npz"), **out)


phase1("g4_phase1_mse")
phase1("g4_phase1_bce", bce=True)

# ---------------------------------------------------------------- G5: shift
x = torch.from_numpy(frames(5, 2))
g5 = {"X": x.numpy()}
for k in range(4):
    torch.manual_seed(k)
    xs = int(12 * torch.rand(1))
    flag = bool(torch.rand(1) > 0.5)
    if flag:
        r = torch.cat((x[:, :, xs:], x[:, :, :xs]), dim=2)
    else:
        r = torch.cat((x[:, :, -xs:], x[:, :, :-xs]), dim=2)
    g5[f"rolled{k}"] = r.numpy(); g5[f"amount{k}"] = np.int64(xs); g5[f"left{k}"] = np.bool_(flag)
np.savez(os.path.join(HERE, "g5_shift.npz"), **g5)

# ---------------------------------------------------------------- G8: legacy Unet with the ConvTranspose2d decoder (nets.py:356-449)
torch.manual_seed(5)
unet = refnets.Unet(upsample=False)
unet.eval()
xu = frames(8, 4)
with torch.no_grad():
    yu, u0 = unet(to_nchw(xu), embeds=True)
    cu = unet(to_nchw(xu), critic=True)
g8 = {"X": xu, "y": yu.numpy(), "u0": u0.numpy(), "critic": cu.numpy()}
g8.update(sd_np("sd", unet.state_dict()))
np.savez(os.path.join(HERE, "g8_unet_convt.npz"), **g8)

# one training step of the same class (train mode; it has no Dropout): loss = MSE(mask, target) + MSE(critic value, target),
# all 24 parameter gradients -- pins the ConvTranspose2d / LeakyReLU(0.2) backward of nets.Unet(upsample=False)
unet.train()
for q in unet.parameters():
    q.grad = None
tgt = torch.from_numpy(np.random.RandomState(6).rand(xu.shape[0], 1, 64, 64).astype(np.float32))
ytc = torch.from_numpy(np.random.RandomState(7).rand(xu.shape[0]).astype(np.float32))
yt = unet(to_nchw(xu))
ct = unet(to_nchw(xu), critic=True).squeeze()
lmask, lcrit = F.mse_loss(yt, tgt), F.mse_loss(ct, ytc)
(lmask + lcrit).backward()
g8t = {"X": xu, "target_mask": tgt.numpy(), "target_critic": ytc.numpy(), "loss_mask": np.float64(lmask.item()), "loss_critic": np.float64(lcrit.item())}
g8t.update(sd_np("sd", unet.state_dict()))
for k, v in unet.named_parameters():
    g8t["grad/" + k] = v.grad.numpy().copy()
np.savez(os.path.join(HERE, "g8_unet_train.npz"), **g8t)

tot = sum(os.path.getsize(os.path.join(HERE, f)) for f in os.listdir(HERE) if f.endswith((".npz", ".json")))
print("fixtures written, total bytes:", tot)
assert not os.path.exists(os.path.join(REF, "__pycache__")), "reference tree was written to"
