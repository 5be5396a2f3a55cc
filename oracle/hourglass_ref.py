"""CPU oracle for the Hourglass (encoder + critic head + decoder/mask head) hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product: only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker / the timed CPU baseline.  The shipped path
(``cgs_amd``) never imports this module and fails loudly when its HIP library is missing.

What this is: a restatement, in this repo's own words, of the arithmetic the reference
performs on its training / inference path, written against stock PyTorch **CPU** fp32 ops
(the reference's own numeric backend, ``requirements.txt:79``).  It is functional (no
``nn.Module``): parameters are plain ``dict[str, Tensor]`` using the reference's
``state_dict`` key names and OIHW shapes, so a reference checkpoint can be fed in as is.

Parity pin: every function here is checked against fixtures captured from the reference's
own ``nets.py`` classes imported in the build container (``tests/golden/make_golden.py`` ->
``tests/golden/*.npz``; checked by ``tests/test_oracle_golden.py``).  The reference ships no
tests or golden vectors of its own, so those captures are the pin.

Reference lines restated (relative to /root/reference):
  critic_apply      nets.py:160-212   (NewCritic.__init__/forward)
  masker_apply      nets.py:452-523   (UnetDecoder.__init__/forward)
  shift_batch       main.py:584-591
  phase1_loss       main.py:189-195
  phase2_loss       main.py:360-429
  adam_step         main.py:178,330-334,461-463 (torch.optim.Adam defaults)
  infer_masks       main.py:1130-1151
  postprocess_masks main.py:1163-1167,1212-1223
  eval_iou          main.py:891-1020 (no salience / CRF), get_iou main.py:1265-1270
  eval_saliency_iou main.py:941-953,976-1003,1010-1012 (the saliency baseline of -eval -salience)
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]

ENC_CONV_KEYS = ("features.0", "features.3", "features.6", "features.10")
BOTTLENECK_KEY = "features.14"
HEAD_KEYS = ("crit.1", "crit.4")
DEC_KEYS = ("dec_model.0", "dec_model.1", "dec_model.2", "dec_model.3", "dec_model.4")
MASK_KEYS = ("masker.0", "masker.2")


# --------------------------------------------------------------------------------------
# parameter inventory (nets.py:161-195, 453-492)
# --------------------------------------------------------------------------------------
def critic_shapes(chfak: int = 1, neck: int = 32, colorchs: int = 3) -> List[Tuple[str, Tuple[int, ...]]]:
    d = [8 * chfak, 8 * chfak, 8 * chfak, 16 * chfak]
    b = neck * chfak
    out = []
    cin = colorchs
    for key, cout in zip(ENC_CONV_KEYS, d):
        out += [(key + ".weight", (cout, cin, 3, 3)), (key + ".bias", (cout,))]
        cin = cout
    out += [(BOTTLENECK_KEY + ".weight", (b, d[3], 4, 4)), (BOTTLENECK_KEY + ".bias", (b,))]
    out += [("crit.1.weight", (b, b)), ("crit.1.bias", (b,))]
    out += [("crit.4.weight", (1, b)), ("crit.4.bias", (1,))]
    return out


def masker_shapes(chfak: int = 1, neck: int = 32, colorchs: int = 3, masker_channels: int = 16):
    e = [8 * chfak, 8 * chfak, 8 * chfak, 16 * chfak]
    d = list(e)
    b = neck * chfak
    out = [
        ("dec_model.0.weight", (d[0], e[0] + d[1], 3, 3)), ("dec_model.0.bias", (d[0],)),
        ("dec_model.1.weight", (d[1], e[1] + d[2], 3, 3)), ("dec_model.1.bias", (d[1],)),
        ("dec_model.2.weight", (d[2], e[2] + d[3], 3, 3)), ("dec_model.2.bias", (d[2],)),
        ("dec_model.3.weight", (d[3], e[3] + b, 3, 3)), ("dec_model.3.bias", (d[3],)),
        ("dec_model.4.weight", (b, b, 1, 1)), ("dec_model.4.bias", (b,)),
        ("masker.0.weight", (masker_channels, colorchs + d[0], 3, 3)), ("masker.0.bias", (masker_channels,)),
        ("masker.2.weight", (1, masker_channels, 3, 3)), ("masker.2.bias", (1,)),
    ]
    return out


def seeded_params(shapes, seed: int) -> Params:
    """Reproducible stand-in weights: uniform(-1/sqrt(fan_in), 1/sqrt(fan_in)) from a numpy
    RandomState.  (Not torch's default-init stream, which differs across torch versions --
    SURVEY.md section 8c -- fixtures therefore always carry explicit weights or this seed.)"""
    rs = np.random.RandomState(seed)
    out = {}
    for key, shp in shapes:
        if key.endswith(".weight"):
            fan_in = int(np.prod(shp[1:]))
            last_bound = 1.0 / math.sqrt(fan_in)
        bound = last_bound
        out[key] = torch.from_numpy(rs.uniform(-bound, bound, size=shp).astype(np.float32))
    return out


# --------------------------------------------------------------------------------------
# forward passes
# --------------------------------------------------------------------------------------
def _drop(h, p, training, mask):
    """nn.Dropout semantics (nets.py:179,183,192): keep-mask / (1-p) in training mode."""
    if not training or p <= 0.0:
        return h
    if mask is None:
        mask = (torch.rand_like(h) >= p).to(h.dtype)
    return h * mask.to(h.dtype) / (1.0 - p)


def critic_apply(P: Params, X: torch.Tensor, collect: bool = False, p: float = 0.0,
                 training: bool = False, masks: Optional[Sequence[torch.Tensor]] = None):
    """Encoder + critic head.  X is NCHW fp32 in [0,1].  ``masks`` (optional) are the three
    dropout keep-masks, shaped like embed2, embed3 and the first Linear's output.
    Returns pred [N,1] (and the 5 embeds when ``collect``): embeds are the four post-pool
    tensors taken BEFORE dropout plus the post-ReLU bottleneck (nets.py:200-205)."""
    m = list(masks) if masks is not None else [None, None, None]
    h = X
    embeds = []
    for i, key in enumerate(ENC_CONV_KEYS):
        h = F.conv2d(h, P[key + ".weight"], P[key + ".bias"], stride=1, padding=1)
        h = F.max_pool2d(F.relu(h), 2)
        embeds.append(h)
        if i >= 2:
            h = _drop(h, p, training, m[i - 2])
    h = F.relu(F.conv2d(h, P[BOTTLENECK_KEY + ".weight"], P[BOTTLENECK_KEY + ".bias"]))
    embeds.append(h)
    f = h.flatten(1)
    f = F.relu(F.linear(f, P["crit.1.weight"], P["crit.1.bias"]))
    f = _drop(f, p, training, m[2])
    pred = torch.sigmoid(F.linear(f, P["crit.4.weight"], P["crit.4.bias"]))
    return (pred, embeds) if collect else pred


def _up2(t):
    return F.interpolate(t, scale_factor=2, mode="nearest")


def masker_apply(P: Params, X: torch.Tensor, embeds: Sequence[torch.Tensor], return_all: bool = False):
    """Decoder + mask head.  Linear trunk (no activation between dec convs, nets.py:494-517),
    concat order (skip, upsampled) and (X, upsampled) for the mask head (nets.py:504-520)."""
    o4 = F.conv2d(embeds[4], P["dec_model.4.weight"], P["dec_model.4.bias"])
    o3 = F.conv2d(torch.cat((embeds[3], _up2(_up2(o4))), 1), P["dec_model.3.weight"], P["dec_model.3.bias"], padding=1)
    o2 = F.conv2d(torch.cat((embeds[2], _up2(o3)), 1), P["dec_model.2.weight"], P["dec_model.2.bias"], padding=1)
    o1 = F.conv2d(torch.cat((embeds[1], _up2(o2)), 1), P["dec_model.1.weight"], P["dec_model.1.bias"], padding=1)
    o0 = F.conv2d(torch.cat((embeds[0], _up2(o1)), 1), P["dec_model.0.weight"], P["dec_model.0.bias"], padding=1)
    hm = F.leaky_relu(F.conv2d(torch.cat((X, _up2(o0)), 1), P["masker.0.weight"], P["masker.0.bias"], padding=1), 0.01)
    Z = torch.sigmoid(F.conv2d(hm, P["masker.2.weight"], P["masker.2.bias"], padding=1))
    if return_all:
        return Z, dict(o4=o4, o3=o3, o2=o2, o1=o1, o0=o0, hm=hm)
    return Z


def u8_to_nchw(X_u8) -> torch.Tensor:
    """NHWC uint8 -> NCHW fp32 / 255 (main.py:189,360-361)."""
    if isinstance(X_u8, np.ndarray):
        X_u8 = torch.from_numpy(X_u8)
    return X_u8.permute(0, 3, 1, 2).float() / 255.0


# --------------------------------------------------------------------------------------
# augmentation (main.py:584-591)
# --------------------------------------------------------------------------------------
def shift_batch(X: torch.Tensor, shift: int, generator: Optional[torch.Generator] = None):
    """Whole-batch circular roll along width (dim 2 of NHWC).  Two draws from the torch RNG:
    the amount ``int(shift*u0)`` then the direction ``u1 > 0.5``.  Returns (rolled, amount, flag)."""
    u0 = torch.rand(1, generator=generator)
    amount = int(shift * u0)
    u1 = torch.rand(1, generator=generator)
    left = bool(u1 > 0.5)
    if left:
        out = torch.cat((X[:, :, amount:], X[:, :, :amount]), dim=2)
    else:
        # amount == 0: X[:, :, -0:] is all of X and X[:, :, :-0] is empty, i.e. the identity
        out = torch.cat((X[:, :, -amount:], X[:, :, :-amount]), dim=2)
    return out, amount, left


# --------------------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------------------
def phase1_loss(Pc: Params, XP: torch.Tensor, Y: torch.Tensor, threshrew: float = 0.0, p: float = 0.0,
                training: bool = True, masks=None):
    """Critic regression (main.py:189-195): mse(critic(X).squeeze(), Y) or BCE under --threshrew."""
    pred = critic_apply(Pc, XP, p=p, training=training, masks=masks).squeeze()
    if threshrew:
        return F.binary_cross_entropy(pred, Y), pred
    return F.mse_loss(pred, Y), pred


def phase2_loss(Pc: Params, Pm: Params, A: torch.Tensor, B: torch.Tensor, Y: torch.Tensor,
                lfak: float = 5, L1: float = 0.5, L2: float = 0.0, inject: bool = True, live: bool = True,
                threshrew: float = 0.0, p: float = 0.0, training: bool = True, masks=None, Ps: Optional[Params] = None,
                staticnorm: bool = True):
    """Joint mask/critic objective of one phase-2 step (main.py:364-429, staticnorm => valuefak=1).

    ``masks`` (optional) = 4 lists of 3 dropout keep-masks for the passes [A, B, replaced, injected]
    in the order the reference draws them (a 5th list: the second critic's pass over A).  ``Ps``: parameters of the second
    critic of -separate (main.py:389-390): the masker then takes ITS embeds of A.  Returns (total, parts dict, Z, pred)."""
    mk = masks if masks is not None else [None, None, None, None, None]
    pred, embeds = critic_apply(Pc, A, collect=True, p=p, training=training, masks=mk[0])
    negpred = critic_apply(Pc, B, p=p, training=training, masks=mk[1])
    pred = pred.squeeze()
    negpred = negpred.squeeze().detach()
    total = 0
    parts = {}
    if live:
        cl = F.binary_cross_entropy(pred, Y) if threshrew else F.mse_loss(pred, Y)
        total = total + lfak * cl
        parts["critic"] = cl
    if Ps is not None:
        _, embeds = critic_apply(Ps, A, collect=True, p=p, training=training, masks=mk[4] if len(mk) > 4 else None)
    Z = masker_apply(Pm, A, embeds)
    replaced = A * (1 - Z) + Z * B
    rv = critic_apply(Pc, replaced, p=p, training=training, masks=mk[2]).squeeze()
    rl = F.mse_loss(rv, negpred.detach())
    total = total + rl
    parts["replace"] = rl
    if inject:
        injected = B * (1 - Z) + Z * A
        iv = critic_apply(Pc, injected, p=p, training=training, masks=mk[3]).squeeze()
        il = F.mse_loss(iv, pred.detach())
        total = total + il
        parts["inject"] = il
    valuefak = 1 if staticnorm else 1 - pred.detach().view(-1, 1, 1, 1)      # main.py:415-418
    if L1:
        nl = L1 * F.l1_loss(valuefak * Z, torch.zeros_like(Z))
        total = total + nl
        parts["norm"] = nl
    if L2:
        nl2 = L2 * F.mse_loss(valuefak * Z, torch.zeros_like(Z))
        total = total + nl2
        parts["norm2"] = nl2
    return total, parts, Z, pred


# --------------------------------------------------------------------------------------
# optimiser (torch.optim.Adam defaults; main.py never reads --lr)
# --------------------------------------------------------------------------------------
class AdamRef:
    """Hand-written Adam (lr 1e-3, betas (0.9,0.999), eps 1e-8, no weight decay/amsgrad),
    checked against torch.optim.Adam in tests/test_oracle_golden.py."""

    def __init__(self, params: Sequence[torch.Tensor], lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
        self.params = list(params)
        self.lr, self.b1, self.b2, self.eps = lr, b1, b2, eps
        self.m = [torch.zeros_like(q) for q in self.params]
        self.v = [torch.zeros_like(q) for q in self.params]
        self.t = 0

    @torch.no_grad()
    def step(self, grads: Sequence[Optional[torch.Tensor]]):
        self.t += 1
        c1 = 1.0 - self.b1 ** self.t
        c2 = 1.0 - self.b2 ** self.t
        for q, g, m, v in zip(self.params, grads, self.m, self.v):
            if g is None:
                continue
            m.mul_(self.b1).add_(g, alpha=1.0 - self.b1)
            v.mul_(self.b2).addcmul_(g, g, value=1.0 - self.b2)
            denom = (v.sqrt() / math.sqrt(c2)).add_(self.eps)
            q.addcdiv_(m, denom, value=-self.lr / c1)


def leafify(P: Params) -> Params:
    return {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}


def train_phase2(Pc: Params, Pm: Params, batches, steps: int, live=True, Ps: Optional[Params] = None, **loss_kw):
    """Runs ``steps`` optimiser steps of phase 2 on (A,B,Y) batches; returns per-step records.
    Optimiser membership follows main.py:330-334 (critic+masker when live, masker only when frozen; + the second critic
    of -separate in both cases)."""
    Pc, Pm = leafify(Pc), leafify(Pm)
    Ps = leafify(Ps) if Ps is not None else None
    keys = ([("c", k) for k in Pc] if live else []) + [("m", k) for k in Pm] + ([("s", k) for k in Ps] if Ps is not None else [])
    tensors = [{"c": Pc, "m": Pm, "s": Ps}[w][k] for w, k in keys]
    opt = AdamRef(tensors)
    records = []
    for s in range(steps):
        A, B, Y = batches[s % len(batches)]
        for t in list(Pc.values()) + list(Pm.values()) + (list(Ps.values()) if Ps is not None else []):
            t.grad = None
        total, parts, Z, pred = phase2_loss(Pc, Pm, A, B, Y, live=live, Ps=Ps, **loss_kw)
        total.backward()
        rec = dict(total=float(total.detach()), parts={k: float(v.detach()) for k, v in parts.items()},
                   grads_c={k: (v.grad.clone() if v.grad is not None else None) for k, v in Pc.items()},
                   grads_m={k: (v.grad.clone() if v.grad is not None else None) for k, v in Pm.items()},
                   Z=Z.detach().clone(), pred=pred.detach().clone())
        if Ps is not None:
            rec["grads_s"] = {k: (v.grad.clone() if v.grad is not None else None) for k, v in Ps.items()}
        opt.step([t.grad for t in tensors])
        rec["params_c"] = {k: v.detach().clone() for k, v in Pc.items()}
        rec["params_m"] = {k: v.detach().clone() for k, v in Pm.items()}
        if Ps is not None:
            rec["params_s"] = {k: v.detach().clone() for k, v in Ps.items()}
        records.append(rec)
    return records


def train_phase1(Pc: Params, batches, steps: int, **loss_kw):
    Pc = leafify(Pc)
    tensors = list(Pc.values())
    opt = AdamRef(tensors)
    records = []
    for s in range(steps):
        XP, Y = batches[s % len(batches)]
        for t in tensors:
            t.grad = None
        loss, pred = phase1_loss(Pc, XP, Y, **loss_kw)
        loss.backward()
        rec = dict(loss=float(loss.detach()), pred=pred.detach().clone(),
                   grads={k: v.grad.clone() for k, v in Pc.items()})
        opt.step([t.grad for t in tensors])
        rec["params"] = {k: v.detach().clone() for k, v in Pc.items()}
        records.append(rec)
    return records


# --------------------------------------------------------------------------------------
# inference + mask post-processing
# --------------------------------------------------------------------------------------
@torch.no_grad()
def infer_masks(Pc: Params, Pm: Params, X01: np.ndarray, batchsize: int = 128):
    """-process loop (main.py:1130-1151): X01 is float64 NHWC in [0,1]; eval mode; returns
    (preds [N], masks [N,1,H,W]) as float32 numpy."""
    preds, masks = [], []
    for b in range(0, len(X01), batchsize):
        batch = torch.from_numpy(X01[b:b + batchsize]).permute(0, 3, 1, 2).float()
        pred, embeds = critic_apply(Pc, batch, collect=True)
        preds.append(pred.squeeze(1).numpy())
        masks.append(masker_apply(Pm, batch, embeds).numpy())
    return np.concatenate(preds, 0), np.concatenate(masks, 0)


def postprocess_masks(X01: np.ndarray, M: np.ndarray, threshold: float = 0.5):
    """main.py:1163-1167,1212: stack [image, raw mask x3, thresholded mask x3] -> uint8 arrays.
    Returns (hardM bool [N,1,H,W], stack uint8 [N,3,H,W,3])."""
    hard = M >= threshold
    cols = [X01] + [np.concatenate((m, m, m), axis=1).transpose(0, 2, 3, 1) for m in (M, hard)]
    stack = np.stack(cols, axis=1)
    return hard, (stack * 255).astype(np.uint8)


def eval_iou(PC, PM, X_u8: np.ndarray, Y: np.ndarray, eval_thresh: float = 0.05, batchsize: int = 128) -> float:
    """main.py:891-1020 (plain branch): X_u8 [n,64,64,3] and Y [n,64,64,k] are the FULL red-trees arrays; the reference
    evaluates rows 100:5000:2, thresholds the eval-mode masks at `eval_thresh` (strictly greater) and returns
    get_iou(hardM, all(Y, -1)) rounded to 3 digits (main.py:1265-1270)."""
    Yb = np.expand_dims(np.all(Y, axis=-1), axis=-1)
    Xs, Ys = X_u8[100:5000:2], Yb[100:5000:2]
    M = []
    with torch.no_grad():
        for b in range(0, len(Xs), batchsize):
            X = u8_to_nchw(Xs[b:b + batchsize])
            _pred, embeds = critic_apply(PC, X, collect=True)
            M.append(masker_apply(PM, X, embeds).numpy())
    M = np.concatenate(M, axis=0)
    hardM = M > eval_thresh
    A, B = hardM.squeeze(), Ys.transpose(0, 3, 1, 2).squeeze()
    return round(float(np.sum(A & B) / np.sum(A | B)), 3)


def eval_saliency_iou(PC, X_u8: np.ndarray, Y: np.ndarray, salience_thresh: float = 1.5, salglobal: bool = True,
                      batchsize: int = 128):
    """The saliency baseline of main.py:941-953, 976-1003, 1010-1012 (eval-mode critic, pred.mean().backward() per batch,
    m = |grad|.sum(channels); normalise, weight by pred, clip at 1, threshold) on rows 100:5000:2.
    Returns (saliou or nan, salM [n,1,64,64] after normalisation, raw |grad| maps)."""
    import sys
    Yb = np.expand_dims(np.all(Y, axis=-1), axis=-1)
    Xs, Ys = X_u8[100:5000:2], Yb[100:5000:2]
    sal, preds = [], []
    for b in range(0, len(Xs), batchsize):
        batch = u8_to_nchw(Xs[b:b + batchsize]).clone().requires_grad_(True)
        pred = critic_apply(PC, batch)
        pred.mean().backward()
        sal.append(batch.grad.abs().sum(dim=1)[:, None].numpy())
        preds.append(pred.detach().squeeze(1).numpy())
    raw = np.concatenate(sal, axis=0)
    preds = np.concatenate(preds, axis=0)
    salM = raw.copy()
    if salglobal:
        norm = (salM * (salM >= 0)).mean() * salience_thresh
    else:
        k = int(salM.shape[-1] * salM.shape[-2] * salience_thresh)
        norm = np.sort(salM.reshape(salM.shape[0], 1, -1), axis=-1)[:, :, k, None, None]
    salM = salM / (norm + sys.float_info.min)
    salM = salM * preds[:, None, None, None]
    salM[(salM >= 1)] = 1
    hard = (salM > salience_thresh).astype(np.uint8)
    A, B = hard.squeeze(), Ys.transpose(0, 3, 1, 2).squeeze()
    union = np.sum(A | B)
    iou = round(float(np.sum(A & B) / union), 3) if union else float("nan")
    return iou, salM, raw


# --------------------------------------------------------------------------------------
# legacy Unet(upsample=False) (nets.py:356-449): Conv + LeakyReLU(0.2) + MaxPool encoder, ConvTranspose2d decoder
# --------------------------------------------------------------------------------------
def unet_convt_apply(P: Params, X: torch.Tensor, critic: bool = False):
    """Restates Unet.forward with upsample=False, pure=False (nets.py:398-449).  Returns (y, u0) or the critic value."""
    act = lambda t: F.leaky_relu(t, 0.2)
    p, h = [], X
    for i in range(4):
        h = F.max_pool2d(act(F.conv2d(h, P[f"enc_model.{i}.weight"], P[f"enc_model.{i}.bias"], padding=1)), 2)
        p.append(h)
    x4 = act(F.conv2d(p[3], P["enc_model.4.weight"], P["enc_model.4.bias"]))
    if critic:
        return F.linear(F.relu(F.linear(x4.flatten(1), P["critic.1.weight"], P["critic.1.bias"])), P["critic.3.weight"], P["critic.3.bias"])
    u = act(F.conv_transpose2d(x4, P["dec_model.4.weight"], P["dec_model.4.bias"]))
    for i in (3, 2, 1):
        u = act(F.conv_transpose2d(torch.cat((u, p[i]), dim=1), P[f"dec_model.{i}.weight"], P[f"dec_model.{i}.bias"], stride=2, padding=1))
    y = torch.sigmoid(F.conv_transpose2d(torch.cat((u, p[0]), dim=1), P["dec_model.0.weight"], P["dec_model.0.bias"], stride=2, padding=1))
    return y, u


# ------------------------------------------------------------------------------------------------
# BUILD-DEFINED 128x128 variant (BASELINE config 5) -- NOT a restatement of the reference: the reference cannot run 128x128 frames
# (its 4x4 valid convolution nets.py:184 would see 8x8 and Flatten -> Linear nets.py:189-190 shape-errors).
# One extra Conv2d(8,8,3)+ReLU+MaxPool stage in front of the encoder and one extra Upsample+cat+Conv2d stage behind the decoder keep
# every other layer's shape.  PARITY UNPINNED: this fp32 CPU form is only the self-consistency check of the bf16 HIP path
# (critic-..._amd/hourglass128.py); nothing in the reference pins it.
# ------------------------------------------------------------------------------------------------
ENC128_KEYS = ("features.0", "features.3", "features.6", "features.9", "features.13")


def critic128_shapes(chfak: int = 1, neck: int = 32):
    d = [8 * chfak, 8 * chfak, 8 * chfak, 8 * chfak, 16 * chfak]
    b = neck * chfak
    out, cin = [], 3
    for key, cout in zip(ENC128_KEYS, d):
        out += [(key + ".weight", (cout, cin, 3, 3)), (key + ".bias", (cout,))]
        cin = cout
    out += [("features.17.weight", (b, d[4], 4, 4)), ("features.17.bias", (b,)),
            ("crit.1.weight", (b, b)), ("crit.1.bias", (b,)), ("crit.4.weight", (1, b)), ("crit.4.bias", (1,))]
    return out


def masker128_shapes(chfak: int = 1, neck: int = 32, masker_channels: int = 16):
    e = [8 * chfak, 8 * chfak, 8 * chfak, 8 * chfak, 16 * chfak]
    b = neck * chfak
    return [("dec_model.0.weight", (e[0], e[0] + e[1], 3, 3)), ("dec_model.0.bias", (e[0],)),       # 64x64
            ("dec_model.1.weight", (e[1], e[1] + e[2], 3, 3)), ("dec_model.1.bias", (e[1],)),       # 32x32
            ("dec_model.2.weight", (e[2], e[2] + e[3], 3, 3)), ("dec_model.2.bias", (e[2],)),       # 16x16
            ("dec_model.3.weight", (e[3], e[3] + e[4], 3, 3)), ("dec_model.3.bias", (e[3],)),       # 8x8
            ("dec_model.4.weight", (e[4], e[4] + b, 3, 3)), ("dec_model.4.bias", (e[4],)),          # 4x4
            ("dec_model.5.weight", (b, b, 1, 1)), ("dec_model.5.bias", (b,)),                       # 1x1 (the pointwise GEMM)
            ("masker.0.weight", (masker_channels, 3 + e[0], 3, 3)), ("masker.0.bias", (masker_channels,)),
            ("masker.2.weight", (1, masker_channels, 3, 3)), ("masker.2.bias", (1,))]


def critic128_apply(Pc: Params, X: torch.Tensor):
    """Critic of the 128x128 variant: (pred [n,1], [e0..e4, e5])."""
    h, emb = X, []
    for key in ENC128_KEYS:
        h = F.max_pool2d(F.relu(F.conv2d(h, Pc[key + ".weight"], Pc[key + ".bias"], padding=1)), 2)
        emb.append(h)
    e5 = F.relu(F.conv2d(h, Pc["features.17.weight"], Pc["features.17.bias"]))
    hid = F.relu(F.linear(e5.flatten(1), Pc["crit.1.weight"], Pc["crit.1.bias"]))
    return torch.sigmoid(F.linear(hid, Pc["crit.4.weight"], Pc["crit.4.bias"])), emb + [e5]


def masker128_apply(Pm: Params, X: torch.Tensor, emb):
    e5 = emb[5]
    o = F.conv2d(e5, Pm["dec_model.5.weight"], Pm["dec_model.5.bias"])
    o = F.conv2d(torch.cat((emb[4], _up2(_up2(o))), 1), Pm["dec_model.4.weight"], Pm["dec_model.4.bias"], padding=1)
    for i in (3, 2, 1, 0):
        o = F.conv2d(torch.cat((emb[i], _up2(o)), 1), Pm[f"dec_model.{i}.weight"], Pm[f"dec_model.{i}.bias"], padding=1)
    hm = F.leaky_relu(F.conv2d(torch.cat((X, _up2(o)), 1), Pm["masker.0.weight"], Pm["masker.0.bias"], padding=1), 0.01)
    return torch.sigmoid(F.conv2d(hm, Pm["masker.2.weight"], Pm["masker.2.bias"], padding=1))


def hourglass128_phase2_loss(Pc: Params, Pm: Params, A: torch.Tensor, B: torch.Tensor, Y: torch.Tensor, lfak: float = 5, L1: float = 0.5):
    """The phase-2 objective of main.py:364-429 (live critic, inject, staticnorm, Dropout off) on the 128x128 variant: the
    self-consistency check of Hourglass128.phase2_step (parity unpinned).  A, B: NCHW fp32 [n,3,128,128].  Returns (total, parts, Z, pred)."""
    pred, emb = critic128_apply(Pc, A)
    negpred = critic128_apply(Pc, B)[0].squeeze().detach()
    pred = pred.squeeze()
    parts = {"critic": F.mse_loss(pred, Y)}
    Z = masker128_apply(Pm, A, emb)
    rv = critic128_apply(Pc, A * (1 - Z) + Z * B)[0].squeeze()
    iv = critic128_apply(Pc, B * (1 - Z) + Z * A)[0].squeeze()
    parts["replace"], parts["inject"] = F.mse_loss(rv, negpred), F.mse_loss(iv, pred.detach())
    parts["norm"] = L1 * F.l1_loss(Z, torch.zeros_like(Z))
    return lfak * parts["critic"] + parts["replace"] + parts["inject"] + parts["norm"], parts, Z, pred


def hourglass128_apply(Pc: Params, Pm: Params, X: torch.Tensor):
    """Eval-mode critic value and mask of the 128x128 variant.  X: NCHW fp32 [n,3,128,128] in [0,1]."""
    pred, emb = critic128_apply(Pc, X)
    return pred, masker128_apply(Pm, X, emb)
