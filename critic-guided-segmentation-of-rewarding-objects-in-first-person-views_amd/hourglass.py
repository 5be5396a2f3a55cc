"""Kernel orchestration of the Hourglass on one MI355X: which libcgs_hip.so entry point runs for which
reference op, in NHWC, over a flat kernel-layout parameter buffer.  PyTorch is used for device memory
and the current stream only; every arithmetic step is a HIP kernel (no torch math on this path).

Reference map:
  critic_forward   NewCritic.forward            nets.py:197-212
  masker_forward   UnetDecoder.forward          nets.py:494-523
  *_backward       what loss.backward() does for those modules (main.py:198,462)
"""
import os
import ctypes as C
from typing import Dict, List, Optional

import torch

from . import _lib
from .spec import (DEC_LAYERS, DROP_SITE_E2, DROP_SITE_E3, DROP_SITE_H1, ENC_LAYERS, Layout)

_ACT = {"none": _lib.ACT_NONE, "relu": _lib.ACT_RELU, "lrelu": _lib.ACT_LRELU, "sigmoid": _lib.ACT_SIGMOID}
HEAD_SLAB = 8192 + 32 + 1024 + 32 + 32 + 1
# Which kernel forms the step is composed of.  These are the build's fixed configuration (round 3: environment switches removed);
# the per-layer forms they replaced stay reachable through the C ABI and are parity-tested there (tests/test_gpu_kernels.py).
# small layers: weight- and data-gradient halves share one launch (csrc/conv_bwd_both.hip)
BWD_BOTH = True
MASK_INFER_FUSED = True   # inference: masker.0 + masker.2 in one kernel
ENC0_MIX_FUSED = True   # features.0 backward + mix backward in one launch
MASK_HEAD_FUSED = True   # masker.2+masker.0 data gradients in one pass
# training forward: masker.0 + masker.2 in one kernel on v_mfma_f32_4x4x1 (csrc/mask_fwd.hip, round 3: 68 us vs 60 + 38 us)
MASK_TRAIN_FUSED = True
# ... from this many images on: the one-kernel form is one workgroup per image (two per CU), the two-launch form splits an image into strips -- on a
# 256-CU device the step is faster WITHOUT it at the reference's own batch (r05, tools/ab_flags_n.py: N = 64: -14 us of 257, 128: -5, 192: +9, 256: +14)
# (this threshold, ENC1_TAIL_BWD_FUSED_MIN_N below, the 768-workgroup caps of the tail kernels and cgs_stagger were all measured on ONE device
#  shape: MI355X in SPX mode, 8 XCDs x 32 CUs = 256 CUs; on another partitioning they are only a starting point.  The small-batch forms the
#  reference's N = 64 takes are pinned by the reference's own run: tests/test_gpu_loops.py G9 (34 phase-2 steps) and G12 (48 phase-1 steps).)
MASK_TRAIN_FUSED_MIN_N = int(os.environ.get("CGS_MASK_TRAIN_FUSED_MIN_N", "160"))
# the 16x16-and-smaller layers image by image in one workgroup (csrc/tail.hip) instead of one launch per layer
TAIL_FWD = True
TAIL_BWD = True
# features.3 and the encoder tail as ONE launch, one workgroup per image (csrc/tail.hip: tail_enc_fwd_kernel<true>, round 4)
ENC1_TAIL_FUSED = True
# ... and features.0 in front of it: the whole critic forward of an image in one workgroup (uint8 frames and virtual mixes)
CRITIC_FWD_FUSED = True
# dec_model.0's data gradient and the decoder tail's backward as ONE launch, one workgroup per image (tail_dec_bwd_kernel<true>, round 4)
DEC0_TAIL_BWD_FUSED = True
# the decoder tail's forward and dec_model.0 as ONE launch, one workgroup per image (tail_dec_fwd_kernel<true>, round 4)
DEC_TAIL_DEC0_FUSED = True
# features.3's data gradient behind the encoder tail's backward in ONE launch (tail_enc_bwd_kernel<true>, round 5); its weight gradient then
# rides in the features.0 backward launch of the same pass (cgs_enc0_bwd_mix_enc1 / cgs_enc0_wgrad_u8_with_head_enc1)
ENC1_TAIL_BWD_FUSED = os.environ.get("CGS_ENC1_TAIL_BWD_FUSED", "1") != "0"      # (A/B switch for tools/; the product default is on)
# ... for passes of this many images or more (same measurement: N = 64: -11 us, 128: -5, 192: -3, 256: +1 without it)
ENC1_TAIL_BWD_FUSED_MIN_N = int(os.environ.get("CGS_ENC1_TAIL_BWD_FUSED_MIN_N", "224"))


def mask_train_fused(n: int) -> bool:
    return MASK_TRAIN_FUSED and n >= MASK_TRAIN_FUSED_MIN_N


def enc1_tail_bwd_fused(n: int) -> bool:
    return ENC1_TAIL_BWD_FUSED and n >= ENC1_TAIL_BWD_FUSED_MIN_N
# ... and features.3's (sparse) weight gradient inside that kernel too, per workgroup over its own images (False: as rider workgroups of the
# features.0 backward launch, which reproduces cgs_conv3x3_bwd_both's slabs bit for bit but costs 24 us per step, r05k)
ENC1_WGRAD_IN_TAIL = os.environ.get("CGS_ENC1_WGRAD_IN_TAIL", "1") != "0"
# ... and features.0's (sparse) weight gradient of the pass in the same kernel as well (uint8 frames / virtual mixes), after the image's d e0:
# "A" = only the pass over A (features.0's stand-alone weight-gradient launch then holds only the head's GEMM), "mix" = only the mixes' pass
# (cgs_enc0_bwd_mix then runs without its weight-gradient role), "both", "" = neither.  MEASURED SLOWER, default off (r05n, three interleaved
# runs: 0.5642 ms without, 0.5649 "A", 0.5717 "mix", 0.5733 "both"): eight 8-row tiles per image lengthen the per-image chain by more than the
# stand-alone roles cost next to features.0's data gradient.  Kept as an opt-in that the bit-identity test still exercises.
ENC0_WGRAD_IN_TAIL = os.environ.get("CGS_ENC0_WGRAD_IN_TAIL", "")
# dec_model.0's weight gradient as spare workgroups of the last critic pass's tail backward launch (live critic; csrc/tail.hip)
DEC0_WGRAD_RIDER = True
DEC0_RIDERS = 256
# dec_model.3's weight gradient as a GEMM over the images in <= 64 rider workgroups of the A pass's features.0 weight-gradient launch instead of
# one 27.7 KB slab row per image inside the decoder tail's backward (round 5; equal up to fp32 summation order).  MEASURED SLOWER, default off
# (r05z: 0.5581 ms with, 0.5508 without): the decoder backward gets 3.7 us shorter and the final reduction reads 12 MB less, but the riders'
# 8-image chains lengthen the step's last weight-gradient launch by more.  Kept as an opt-in that the bit-identity test still exercises.
DEC3_WGRAD_RIDER = os.environ.get("CGS_DEC3_WGRAD_RIDER", "0") != "0"
_both = "c3,c2,c1,c0,d3,d2,d1"   # measured: d0 and m0 do not gain
BOTH_ENC = {int(t[1]) for t in _both.split(",") if t.startswith("c")} if BWD_BOTH else set()
BOTH_DEC = {t for t in _both.split(",") if t[0] in "dm"} if BWD_BOTH else set()
_DEC_TAG = {0: "d3", 1: "d2", 2: "d1", 3: "d0", 4: "m0", 5: "m2"}
PW_SLAB = 32 * 32 + 32


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _chk(t: torch.Tensor, dtype, what: str):
    if not t.is_cuda:
        raise _lib.CgsError(f"{what}: HIP kernels need a device tensor (got {t.device}); there is no CPU fallback")
    if t.dtype != dtype or not t.is_contiguous():
        raise _lib.CgsError(f"{what}: expected contiguous {dtype}, got {t.dtype} contiguous={t.is_contiguous()}")
    return t


def _chk_img(x, n: int, what: str):
    """The kernels hard-code 64x64x3 NHWC frames and read n*64*64*3 elements: anything else is refused before a launch
    (the reference fails with a shape error from its first Linear in the same situation, nets.py:189-190)."""
    shp = tuple(x.shape) if not isinstance(x, MixInput) else (x.n_a, 64, 64, 3)
    if len(shp) != 4 or shp[1:] != (64, 64, 3) or (not isinstance(x, MixInput) and shp[0] < n):
        raise _lib.CgsError(f"{what}: expected an NHWC image batch [>={n},64,64,3], got {shp}")


class DropState:
    """Dropout configuration of a pass: probability, Philox seed, the device step counter and the index of
    the pass's first image inside the batch buffer (so differently sliced launches draw the same masks)."""

    def __init__(self, p: float, seed: int, step: Optional[torch.Tensor], img_off: int = 0):
        self.p, self.seed, self.step, self.img_off = float(p), int(seed), step, int(img_off)

    def shifted(self, img_off: int) -> "DropState":
        return DropState(self.p, self.seed, self.step, img_off)

    def desc(self, site: int, active: bool = True, per_img4: int = 0) -> _lib.Dropout:
        if not active or self.p <= 0.0:
            return _lib.Dropout(0.0, 0, 0, None, 0, 0)
        return _lib.Dropout(self.p, site, self.seed, self.step.data_ptr(), self.img_off * per_img4, 0)


NO_DROP = DropState(0.0, 0, None)


def _wp(flat: torch.Tensor, lay: Layout, key: str):
    return flat.data_ptr() + 4 * lay.off(key)


def tail_enc_weights(fc: torch.Tensor, lc: Layout, pw=None) -> _lib.TailEncWeights:
    """Pointers into the flat critic buffer for the tail kernels (pw = (w_ptr, b_ptr, ...) of the decoder's 1x1 conv)."""
    g = lambda k: _wp(fc, lc, k)
    return _lib.TailEncWeights(g("features.6.weight"), g("features.6.bias"), g("features.10.weight"), g("features.10.bias"),
                               g("features.14.weight"), g("features.14.bias"), g("crit.1.weight"), g("crit.1.bias"),
                               g("crit.4.weight"), g("crit.4.bias"), pw[0] if pw else None, pw[1] if pw else None)


def tail_dec_weights(fm: torch.Tensor, lm: Layout) -> _lib.TailDecWeights:
    g = lambda k: _wp(fm, lm, k)
    return _lib.TailDecWeights(g("dec_model.3.weight"), g("dec_model.3.bias"), g("dec_model.2.weight"), g("dec_model.2.bias"),
                               g("dec_model.1.weight"), g("dec_model.1.bias"))


def conv_desc(n, hw, ca, cb, co, src_u8, ups, act, pool, drop: _lib.Dropout) -> _lib.ConvDesc:
    return _lib.ConvDesc(n, hw, hw, ca, cb, co, _lib.SRC_U8 if src_u8 else _lib.SRC_F32, ups, _ACT[act], pool, drop)


class SideStream:
    """Weight-gradient kernels only feed the slab reduction at the end of the step, so they are launched on a
    second HIP stream (captured into the same graph as a parallel branch): the MFMA wgrad kernels then overlap
    the VALU data-gradient chain instead of serialising with it."""

    def __init__(self, stream: Optional["torch.cuda.Stream"]):
        self.stream = stream

    def fork(self):
        """Context manager: work launched inside runs on the side stream after everything enqueued so far."""
        if self.stream is None:
            import contextlib
            return contextlib.nullcontext()
        self.stream.wait_stream(torch.cuda.current_stream())
        return torch.cuda.stream(self.stream)

    def join(self):
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)


NO_SIDE = SideStream(None)


class MixInput:
    """The replaced | injected mixes of main.py:395,406 as a VIRTUAL critic input: features.0's kernels compute
    A(1-Z)+ZB / B(1-Z)+ZA in their tile loaders from the uint8 frames and the mask, nothing is materialised."""

    def __init__(self, A_u8: torch.Tensor, B_u8: torch.Tensor, Z: torch.Tensor):
        _chk(A_u8, torch.uint8, "A"); _chk(B_u8, torch.uint8, "B"); _chk(Z, torch.float32, "Z")
        self.A, self.B, self.Z = A_u8, B_u8, Z
        self.n_a = A_u8.shape[0]
        self.device = A_u8.device
        self.dtype = torch.float32
        self.src = _lib.MixSrc(A_u8.data_ptr(), B_u8.data_ptr(), Z.data_ptr(), self.n_a, 0)

    def ptr(self):
        return C.cast(C.pointer(self.src), C.c_void_p)


class SlabPlan:
    """Collects (slab, destination) pairs of one backward pass; run() sums every slab into the flat
    gradient buffer with ONE cgs_reduce_slabs launch (fixed order => bitwise reproducible)."""

    def __init__(self):
        self.jobs = []   # (slab tensor, nslab, count, dst offset)
        self._table = None
        self._keep = None
        self._extra = []

    def add(self, slab: torch.Tensor, nslab: int, count: int, dst_off: int):
        # two passes that write adjacent slabs for the same destination become ONE job (no write race)
        for i, (s0, n0, c0, o0) in enumerate(self.jobs):
            if o0 == dst_off and c0 == count:
                if slab.data_ptr() != s0.data_ptr() + 4 * n0 * c0:
                    raise _lib.CgsError("slabs of passes that share a destination must be adjacent in memory")
                self.jobs[i] = (s0, n0 + nslab, c0, o0)
                self._extra.append(slab)
                return
        self.jobs.append((slab, nslab, count, dst_off))

    def build(self, grad_flat: torch.Tensor, accumulate: bool = False):
        # The launch is a (max columns / 32) x (jobs) grid: wide slabs are cut into column chunks (same row stride) so that
        # every job fills its share of the grid instead of most workgroups exiting at once.
        CH = 1024
        rows = []
        for slab, nslab, count, off in self.jobs:
            for c0 in range(0, count, CH):
                rows.append((slab.data_ptr() + 4 * c0, grad_flat.data_ptr() + 4 * (off + c0), nslab, count, min(CH, count - c0)))
        arr = (_lib.ReduceJob * len(rows))()
        for i, (sp, dp, nslab, stride, cnt) in enumerate(rows):
            arr[i] = _lib.ReduceJob(sp, dp, nslab, stride, cnt, int(accumulate))
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        self._table = host.to(grad_flat.device)
        self._keep = grad_flat
        self._njobs = len(rows)
        self._max = max(r[4] for r in rows)
        return self

    def run(self, step: Optional[torch.Tensor] = None):
        _lib.call("cgs_reduce_slabs", _p(self._table), self._njobs, self._max, _p(step), _stream())

    def run_adam(self, step, param, grad, m, v, lr, b1, b2, eps, ticket, loss=None):
        """Reduction + Adam (+ the loss values) in one launch (single GPU).  loss = (n, pred, y, zpart, nzpart, lfak, l1, l2, flags,
        nz, losses) or None."""
        ls = loss if loss is not None else (0, None, None, None, 0, 0.0, 0.0, 0.0, 0, 0, None)
        _lib.call("cgs_reduce_adam", _p(self._table), self._njobs, self._max, _p(step), _p(param), _p(grad), _p(m), _p(v),
                  float(lr), float(b1), float(b2), float(eps), _p(ticket), int(ls[0]), _p(ls[1]), _p(ls[2]), _p(ls[3]), int(ls[4]),
                  float(ls[5]), float(ls[6]), float(ls[7]), int(ls[8]), int(ls[9]), _p(ls[10]), _stream())


# ------------------------------------------------------------------------------------------------
# critic
# ------------------------------------------------------------------------------------------------
def critic_forward(flat: torch.Tensor, lay: Layout, x: torch.Tensor, n: int, drop: DropState = NO_DROP,
                   out: Optional[Dict[str, torch.Tensor]] = None, pw=None) -> Dict[str, torch.Tensor]:
    """x: NHWC uint8 or NHWC fp32 [n,64,64,3].  Returns pooled embeds e0..e3 (pre-dropout), their argmax
    masks am0..am3, e4 [n,32], h1 [n,32], pred [n].  ``out`` may hold preallocated views to fill.
    pw = (w_ptr, b_ptr, o4[n,32]): also emit the decoder's bottleneck 1x1 conv of e4 (dec_model.4) from the head kernel."""
    mixin = isinstance(x, MixInput)
    u8 = (not mixin) and x.dtype == torch.uint8
    if not mixin:
        _chk(x, torch.uint8 if u8 else torch.float32, "critic input")
    _chk_img(x, n, "critic input")
    dev = x.device
    o = out if out is not None else {}
    src = x
    whole = TAIL_FWD and ENC1_TAIL_FUSED and CRITIC_FWD_FUSED and (mixin or u8)      # the whole forward in one launch (not for fp32 frames)
    for i, (key, hw, ca, cb, co, ups, act, pool, site) in enumerate(ENC_LAYERS):
        e = o.get(f"e{i}")
        if e is None:
            e = o[f"e{i}"] = torch.empty((n, hw // 2, hw // 2, co), device=dev, dtype=torch.float32)
        am = o.get(f"am{i}")
        if am is None:
            am = o[f"am{i}"] = torch.empty((n, hw // 2, hw // 2, co // 8), device=dev, dtype=torch.int32)
        if TAIL_FWD and (i >= 2 or (i == 1 and ENC1_TAIL_FUSED) or (i == 0 and whole)):
            src = e
            continue        # features.6 / features.10 / head (and, fused, features.3 / features.0): one tail kernel below
        d = conv_desc(n, hw, ca, cb, co, u8 and i == 0, ups, act, pool, drop.desc(DROP_SITE_E2, site is not None, 128))
        if mixin and i == 0:
            d.src_a = _lib.SRC_MIX
        _lib.call("cgs_conv3x3_fwd", C.byref(d), src.ptr() if (mixin and i == 0) else _p(src), None,
                  C.c_void_p(flat.data_ptr() + 4 * lay.off(key + ".weight")),
                  C.c_void_p(flat.data_ptr() + 4 * lay.off(key + ".bias")), _p(e), _p(am), _stream())
        src = e
    for k, shape in (("e4", (n, 32)), ("h1", (n, 32)), ("pred", (n,))):
        if o.get(k) is None:
            o[k] = torch.empty(shape, device=dev, dtype=torch.float32)
    if TAIL_FWD:
        pwp = (pw[0], pw[1]) if pw else None
        tw = tail_enc_weights(flat, lay, (pwp[0].value, pwp[1].value) if pwp else None)
        if whole:
            k0, k3 = ENC_LAYERS[0][0], ENC_LAYERS[1][0]
            wp_ = lambda k: C.c_void_p(flat.data_ptr() + 4 * lay.off(k))
            _lib.call("cgs_critic_fwd_fused", n, C.byref(tw), x.ptr() if mixin else _p(x), int(mixin), wp_(k0 + ".weight"), wp_(k0 + ".bias"),
                      _p(o["e0"]), _p(o["am0"]), wp_(k3 + ".weight"), wp_(k3 + ".bias"), _p(o["e1"]), _p(o["am1"]), _p(o["e2"]), _p(o["am2"]),
                      _p(o["e3"]), _p(o["am3"]), _p(o["e4"]), _p(o["h1"]), _p(o["pred"]), _p(pw[2]) if pw else None,
                      drop.desc(DROP_SITE_E2, True, 128), drop.desc(DROP_SITE_E3, True, 64), drop.desc(DROP_SITE_H1, True, 8), _stream())
            return o
        if ENC1_TAIL_FUSED:
            k3 = ENC_LAYERS[1][0]
            _lib.call("cgs_enc1_tail_fwd", n, C.byref(tw), _p(o["e0"]), C.c_void_p(flat.data_ptr() + 4 * lay.off(k3 + ".weight")),
                      C.c_void_p(flat.data_ptr() + 4 * lay.off(k3 + ".bias")), _p(o["e1"]), _p(o["am1"]), _p(o["e2"]), _p(o["am2"]), _p(o["e3"]),
                      _p(o["am3"]), _p(o["e4"]), _p(o["h1"]), _p(o["pred"]), _p(pw[2]) if pw else None, drop.desc(DROP_SITE_E2, True, 128),
                      drop.desc(DROP_SITE_E3, True, 64), drop.desc(DROP_SITE_H1, True, 8), _stream())
            return o
        _lib.call("cgs_tail_enc_fwd", n, C.byref(tw), _p(o["e1"]), _p(o["e2"]), _p(o["am2"]), _p(o["e3"]), _p(o["am3"]),
                  _p(o["e4"]), _p(o["h1"]), _p(o["pred"]), _p(pw[2]) if pw else None, drop.desc(DROP_SITE_E2, True, 128),
                  drop.desc(DROP_SITE_E3, True, 64), drop.desc(DROP_SITE_H1, True, 8), _stream())
        return o
    fp = flat.data_ptr()
    _lib.call("cgs_head_fwd", n, _p(o["e3"]), C.c_void_p(fp + 4 * lay.off("features.14.weight")),
              C.c_void_p(fp + 4 * lay.off("features.14.bias")), C.c_void_p(fp + 4 * lay.off("crit.1.weight")),
              C.c_void_p(fp + 4 * lay.off("crit.1.bias")), C.c_void_p(fp + 4 * lay.off("crit.4.weight")),
              C.c_void_p(fp + 4 * lay.off("crit.4.bias")), drop.desc(DROP_SITE_E3, True, 64), drop.desc(DROP_SITE_H1, True, 8),
              _p(o["e4"]), _p(o["h1"]), _p(o["pred"]), pw[0] if pw else None, pw[1] if pw else None,
              _p(pw[2]) if pw else None, _stream())
    return o


def enc0_in_tail(mixin: bool, u8: bool, dx) -> bool:
    """Does this critic pass form features.0's weight gradient inside its tail backward kernel (ENC0_WGRAD_IN_TAIL)?  Only the two forms the
    training step uses: the virtual mixes, and uint8 frames without an image gradient."""
    if mixin:
        return ENC0_WGRAD_IN_TAIL in ("both", "mix")
    return u8 and dx is None and ENC0_WGRAD_IN_TAIL in ("both", "A")


def head_wgrad(ranges, plan: "SlabPlan", lay: Layout, ws: Dict[str, torch.Tensor]):
    """Weight gradients of the critic head (+ the decoder's 1x1 conv) for the image ranges the tail backward kernel(s) left
    behind: ranges = [(hvec, e4, d_o4 or None, n, n_o4, pw_bwd or None)], at most two; one small GEMM over the images."""
    enc0 = next((r["enc0"] for r in ranges if isinstance(r, dict)), None)     # a deferred features.0 weight gradient (uint8 frames)
    ranges = [r for r in ranges if not isinstance(r, dict)]
    if not ranges:
        if enc0 is not None:
            raise _lib.CgsError("head_wgrad: a deferred features.0 weight gradient needs a head range to ride with")
        return
    assert len(ranges) <= 2
    lib = _lib.load()
    total = sum(r[3] for r in ranges)
    nsl = lib.cgs_tail_head_wgrad_slabs(total)
    dev = ranges[0][0].device
    key = f"slab_head_{total}"
    if ws.get(key) is None:
        ws[key] = torch.empty((nsl, HEAD_SLAB), device=dev, dtype=torch.float32)
    pwb = next((r[5] for r in ranges if r[5] is not None), None)
    if pwb is not None and ws.get(key + "_pw") is None:
        ws[key + "_pw"] = torch.empty((nsl, PW_SLAB), device=dev, dtype=torch.float32)
    sl, slpw = ws[key], (ws[key + "_pw"] if pwb is not None else None)
    r0 = ranges[0]
    r1 = ranges[1] if len(ranges) > 1 else (None, None, None, 0, 0, None)
    if enc0 is not None:
        n_e, x_e, dy_e, am_e, slab_e = enc0[:5]
        e1w = enc0[5] if len(enc0) > 5 else None          # features.3's deferred weight gradient: (n, e0, d e1, am1, slab, rows)
        d3 = enc0[6] if len(enc0) > 6 else None           # dec_model.3's deferred weight gradient: (n, e3, o4, d o3, slab)
        if d3 is not None:
            w1 = e1w if e1w is not None else (0, None, None, None, None, 0)
            _lib.call("cgs_enc0_wgrad_u8_with_head_riders", n_e, _p(x_e), _p(dy_e), _p(am_e), _p(slab_e), r0[3], _p(r0[0]), _p(r0[1]), _p(r0[2]),
                      r0[4], r1[3], _p(r1[0]), _p(r1[1]), _p(r1[2]), r1[4], _p(sl), _p(slpw),
                      int(w1[0]), _p(w1[1]), _p(w1[2]), _p(w1[3]), _p(w1[4]), int(w1[5]),
                      int(d3[0]), _p(d3[1]), _p(d3[2]), _p(d3[3]), _p(d3[4]), _stream())
        elif e1w is not None:
            _lib.call("cgs_enc0_wgrad_u8_with_head_enc1", n_e, _p(x_e), _p(dy_e), _p(am_e), _p(slab_e), r0[3], _p(r0[0]), _p(r0[1]), _p(r0[2]),
                      r0[4], r1[3], _p(r1[0]), _p(r1[1]), _p(r1[2]), r1[4], _p(sl), _p(slpw),
                      int(e1w[0]), _p(e1w[1]), _p(e1w[2]), _p(e1w[3]), _p(e1w[4]), int(e1w[5]), _stream())
        else:
            _lib.call("cgs_enc0_wgrad_u8_with_head", n_e, _p(x_e), _p(dy_e), _p(am_e), _p(slab_e), r0[3], _p(r0[0]), _p(r0[1]), _p(r0[2]), r0[4],
                      r1[3], _p(r1[0]), _p(r1[1]), _p(r1[2]), r1[4], _p(sl), _p(slpw), _stream())
    else:
        _lib.call("cgs_tail_head_wgrad", r0[3], _p(r0[0]), _p(r0[1]), _p(r0[2]), r0[4], r1[3], _p(r1[0]), _p(r1[1]), _p(r1[2]), r1[4],
                  _p(sl), _p(slpw), _stream())
    plan.add(sl, nsl, HEAD_SLAB, lay.off("features.14.weight"))
    if pwb is not None:
        pwb[2].add(slpw, nsl, PW_SLAB, pwb[3])


def critic_backward(flat: torch.Tensor, lay: Layout, x: torch.Tensor, n: int, saved: Dict[str, torch.Tensor],
                    dpred: torch.Tensor, plan: SlabPlan, drop: DropState = NO_DROP,
                    d_embeds: Optional[List[torch.Tensor]] = None, n_add: int = 0,
                    dx: Optional[torch.Tensor] = None, dx_from: int = 0,
                    ws: Optional[Dict[str, torch.Tensor]] = None, side: "SideStream" = None,
                    need_wgrad: bool = True, pw_bwd=None, mix_bwd=None, loss=None, head_sink=None, rider=None,
                    dec3_rider=None) -> Optional[torch.Tensor]:
    """Backward of critic_forward for images [0,n).  pw_bwd = (d_o4 [n_add,32], w_pw_ptr, plan_pw, dst_off): the decoder
    bottleneck's backward (dec_model.4) runs inside the head kernel; its slab is registered in plan_pw at dst_off.
    mix_bwd = (A_u8, B_u8, Z, inject, l1_scale, l2_scale, dzpre): x are the replaced|injected mixes of n_a = len(A) images;
    features.0's backward then also performs the mix backward (cgs_enc0_bwd_mix) and writes dzpre; no dx is produced.  d_embeds = [dE0..dE4] gradients arriving at the embeds
    from the decoder (valid for images < n_add; their buffers are reused as the running totals).
    dx: optional [n-dx_from,64,64,3] output for the image gradient of images >= dx_from.
    Weight-gradient slabs are registered in ``plan`` (dst offsets = this module's flat layout)."""
    # dpred may be None with loss = (target [n], scale, bce): the tail kernel derives d loss / d pred itself
    # (scale * 2 (pred - target), or the BCE form); both None = no loss on this pass's head.  Tail path only.
    mixin = isinstance(x, MixInput)
    if dpred is None and not TAIL_BWD:
        raise _lib.CgsError("critic_backward without dpred needs the tail kernels (CGS_TAIL_BWD=1)")
    if mixin and mix_bwd is None:
        raise _lib.CgsError("a MixInput critic input needs mix_bwd (features.0's backward consumes the virtual mixes)")
    u8 = (not mixin) and x.dtype == torch.uint8
    dev = x.device
    fp = flat.data_ptr()
    ws = ws if ws is not None else {}

    def buf(name, shape, dtype=torch.float32):
        t = ws.get(name)
        if t is None:
            t = ws[name] = torch.empty(shape, device=dev, dtype=dtype)
        # a cached workspace (e.g. the slab rows engine._slab_views sized for the launch form it expected) must hold what THIS call
        # writes: a different launch form asking for more rows than were allocated would write out of bounds
        want = 1
        for d in shape:
            want *= int(d)
        if t.numel() < want or t.dtype != dtype:
            raise _lib.CgsError(f"workspace '{name}': cached {tuple(t.shape)} {t.dtype} cannot hold the requested {tuple(shape)} {dtype} "
                                "(the launch form changed after the workspace was sized)")
        return t

    has_add = d_embeds is not None and n_add > 0
    side = side if side is not None else NO_SIDE
    lib = _lib.load()
    first_layer = 3
    enc0_done = False        # features.0's weight gradient was formed inside the tail backward kernel (ENC0_WGRAD_IN_TAIL)
    enc1_wgrad = None        # features.3's weight gradient, deferred into this pass's features.0 backward launch (ENC1_TAIL_BWD_FUSED)
    if TAIL_BWD and not (has_add and d_embeds[4] is not None and pw_bwd is None):
        # head + features.10 + features.6 backward in one tail kernel: d e1 (skip gradients included)
        use_pw = pw_bwd is not None and has_add
        nsl = lib.cgs_tail_enc_bwd_slabs(n)
        sl10 = buf("slab_enc3", (nsl, 9 * 8 * 16 + 16)) if need_wgrad else None
        sl6 = buf("slab_enc2", (nsl, 9 * 8 * 8 + 8)) if need_wgrad else None
        hvec = buf("hvec", (n, 384)) if need_wgrad else None
        d_cur = buf("de1", (n, 16, 16, 8))
        tw = tail_enc_weights(flat, lay, (pw_bwd[1].value, None) if use_pw else None)
        rd = rider if rider is not None else (0, None, None, None, None, 0)      # (n, e0, o1, dy, slab, rows): dec_model.0's weight gradient
        targs = (n, C.byref(tw), _p(saved["e1"]), _p(saved["e2"]), _p(saved["am2"]), _p(saved["e3"]),
                 _p(saved["am3"]), _p(saved["e4"]), _p(saved["h1"]), _p(saved["pred"]), _p(dpred),
                 _p(loss[0]) if (dpred is None and loss is not None) else None, float(loss[1]) if loss is not None else 0.0,
                 int(bool(loss[2])) if loss is not None else 0, _p(d_embeds[1]) if has_add else None,
                 _p(d_embeds[2]) if has_add else None, _p(d_embeds[3]) if has_add else None, _p(pw_bwd[0]) if use_pw else None,
                 n_add if has_add else 0, _p(d_cur), _p(hvec), _p(sl10), _p(sl6), drop.desc(DROP_SITE_E2, True, 128),
                 drop.desc(DROP_SITE_E3, True, 64), drop.desc(DROP_SITE_H1, True, 8),
                 int(rd[0]), _p(rd[1]), _p(rd[2]), _p(rd[3]), _p(rd[4]), int(rd[5]))
        # features.3's data gradient in the same launch when the pass's features.0 backward launch can take its weight gradient along (the
        # mixes' cgs_enc0_bwd_mix, or the uint8 frames' weight gradient deferred into head_wgrad) -- or when no weight gradient is needed
        enc1_host = (mix_bwd is not None) or (u8 and head_sink is not None and dx is None)
        fused1 = False
        if enc1_tail_bwd_fused(n) and 1 in BOTH_ENC and (ENC1_WGRAD_IN_TAIL or enc1_host or not need_wgrad):
            de0 = buf("de0", (n, 32, 32, 8))
            # features.3's weight gradient inside the same kernel (ENC1_WGRAD_IN_TAIL) or as riders of the features.0 backward launch
            # (one row per tail workgroup; engine._slab_views allocates the two passes' rows adjacent under the same name)
            slab1_in = buf("slab_enc1", (nsl, 9 * 8 * 8 + 8)) if (need_wgrad and ENC1_WGRAD_IN_TAIL) else None
            slab0_in, xk, xp = None, 0, None
            if need_wgrad and slab1_in is not None and enc0_in_tail(mixin, u8, dx):
                slab0_in = buf("slab_enc0", (nsl, 9 * 3 * 8 + 8))
                xk, xp = (_lib.SRC_MIX, x.ptr()) if mixin else (_lib.SRC_U8, _p(x))
            rc = lib.cgs_tail_enc_bwd_enc1(*targs, _p(saved["am1"]), C.c_void_p(fp + 4 * lay.off("features.3.weight")),
                                           _p(d_embeds[0]) if has_add else None, n_add if has_add else 0, _p(de0),
                                           _p(saved["e0"]) if slab1_in is not None else None, _p(slab1_in),
                                           xk, xp, _p(saved["am0"]) if slab0_in is not None else None, _p(slab0_in), _stream())
            if rc == 0:
                fused1 = True
            elif rc != _lib.ERR_UNSUPPORTED:
                _lib.check(rc, "cgs_tail_enc_bwd_enc1")
        if not fused1:
            _lib.call("cgs_tail_enc_bwd_rider", *targs, _stream())
        rider = None
        if need_wgrad:
            plan.add(sl10, nsl, 9 * 8 * 16 + 16, lay.off("features.10.weight"))
            plan.add(sl6, nsl, 9 * 8 * 8 + 8, lay.off("features.6.weight"))
            rng = (hvec, saved["e4"], pw_bwd[0] if use_pw else None, n, n_add if use_pw else 0, pw_bwd if use_pw else None)
            if head_sink is not None:
                head_sink.append(rng)            # the caller forms the head's weight gradients for all its passes at once
            else:
                head_wgrad([rng], plan, lay, ws)
        first_layer = 1
        if fused1:
            if slab0_in is not None:
                plan.add(slab0_in, nsl, 9 * 3 * 8 + 8, lay.off("features.0.weight"))
                enc0_done = True
            if need_wgrad and slab1_in is not None:
                plan.add(slab1_in, nsl, 9 * 8 * 8 + 8, lay.off("features.3.weight"))
            elif need_wgrad:
                nsl1 = lib.cgs_enc1_wgrad_rider_slabs(n)
                slab1 = buf("slab_enc1", (nsl1, 9 * 8 * 8 + 8))
                enc1_wgrad = (n, saved["e0"], d_cur, saved["am1"], slab1, nsl1)
                plan.add(slab1, nsl1, 9 * 8 * 8 + 8, lay.off("features.3.weight"))
            d_cur, first_layer = de0, 0
    if rider is not None:
        raise _lib.CgsError("critic_backward: a deferred dec_model.0 weight gradient needs the tail backward launch to ride with")
    # ---- head: d e3 = head gradient (through dropout) + decoder skip gradient ----
    if first_layer == 3:
      if True:
        nsl = lib.cgs_head_bwd_slabs(n)
        slab = buf("slab_head", (nsl, HEAD_SLAB))
        d_cur = buf("de3", (n, 4, 4, 16))
        use_pw = pw_bwd is not None and has_add
        slab_pw = buf("slab_head_pw", (nsl, PW_SLAB)) if use_pw else None
        de4 = d_embeds[4] if has_add else None
        _lib.call("cgs_head_bwd", n, _p(saved["e3"]), _p(saved["e4"]), _p(saved["h1"]), _p(saved["pred"]), _p(dpred),
                  _p(de4) if (has_add and de4 is not None) else None, _p(d_embeds[3]) if has_add else None, n_add if has_add else 0,
                  C.c_void_p(fp + 4 * lay.off("features.14.weight")), C.c_void_p(fp + 4 * lay.off("crit.1.weight")),
                  C.c_void_p(fp + 4 * lay.off("crit.4.weight")), drop.desc(DROP_SITE_E3, True, 64), drop.desc(DROP_SITE_H1, True, 8),
                  _p(d_cur), _p(slab), _p(pw_bwd[0]) if use_pw else None, pw_bwd[1] if use_pw else None,
                  _p(slab_pw) if use_pw else None, _stream())
        plan.add(slab, nsl, HEAD_SLAB, lay.off("features.14.weight"))
        if use_pw:
            pw_bwd[2].add(slab_pw, nsl, PW_SLAB, pw_bwd[3])
    for i in range(first_layer, -1, -1):
        key, hw, ca, cb, co, ups, act, pool, site = ENC_LAYERS[i]
        src = x if i == 0 else saved[f"e{i - 1}"]
        d = conv_desc(n, hw, ca, cb, co, u8 and i == 0, ups, act, pool, drop.desc(DROP_SITE_E2, site is not None, 128))
        cnt = 9 * ca * co + co
        wptr = C.c_void_p(fp + 4 * lay.off(key + ".weight"))
        if i == 0 and mix_bwd is not None:
            A8, B8, Zm, inj, l1s, l2s, dzp = mix_bwd[:7]
            vfp = mix_bwd[7] if len(mix_bwd) > 7 else None      # -staticnorm '': pred of A, the regulariser's per-image weight
            slab = None
            if need_wgrad and not enc0_done:
                nsl = lib.cgs_enc0_bwd_mix_slabs(n)
                slab = buf("slab_enc0", (nsl, cnt))
                plan.add(slab, nsl, cnt, lay.off(key + ".weight"))
            e1w = enc1_wgrad if enc1_wgrad is not None else (0, None, None, None, None, 0)
            _lib.call("cgs_enc0_bwd_mix_enc1", A8.shape[0], int(bool(inj)), _p(src) if (need_wgrad and not mixin) else None, _p(d_cur),
                      _p(saved["am0"]), wptr, _p(A8), _p(B8), _p(Zm), float(l1s), float(l2s), _p(vfp), _p(dzp), _p(slab),
                      _p(e1w[1]), _p(e1w[2]), _p(e1w[3]), _p(e1w[4]), _stream())
            return None
        if need_wgrad and i in BOTH_ENC and (i > 0 or (dx is not None and dx_from == 0 and not u8)):
            # both halves in one launch: slab + d e{i-1} (dropout mask and decoder skip gradient fused)
            nsl = lib.cgs_conv3x3_bwd_both_slabs(C.byref(d))
            if nsl < 0:
                _lib.check(nsl, "cgs_conv3x3_bwd_both_slabs")
            slab = buf(f"slab_enc{i}", (nsl, cnt))
            nxt = buf(f"de{i - 1}", (n, hw, hw, ca)) if i > 0 else dx
            _lib.call("cgs_conv3x3_bwd_both", C.byref(d), _p(src), None, _p(d_cur), _p(saved[f"am{i}"]), wptr,
                      _p(d_embeds[i - 1]) if (has_add and i > 0) else None, n_add if (has_add and i > 0) else 0, _p(nxt), None,
                      _p(slab), _stream())
            plan.add(slab, nsl, cnt, lay.off(key + ".weight"))
            d_cur = nxt
            continue
        nsl = lib.cgs_conv3x3_bwd_weight_slabs(C.byref(d))
        if nsl < 0:
            _lib.check(nsl, "cgs_conv3x3_bwd_weight_slabs")
        if need_wgrad and not (i == 0 and enc0_done):
            slab = buf(f"slab_enc{i}", (nsl, cnt))
            if i == 0 and head_sink is not None and u8:
                # features.0 on the uint8 frames: launched together with the head's weight gradients (head_wgrad below)
                head_sink.append({"enc0": (n, src, d_cur, saved["am0"], slab, enc1_wgrad, dec3_rider)})
                enc1_wgrad, dec3_rider = None, None
            else:
                with side.fork():
                    _lib.call("cgs_conv3x3_bwd_weight", C.byref(d), _p(src), None, _p(d_cur), _p(saved[f"am{i}"]), _p(slab), _stream())
            plan.add(slab, nsl, cnt, lay.off(key + ".weight"))
        if i > 0:   # d e{i-1} = conv_bwd * dropout mask + decoder skip gradient (fused epilogue)
            nxt = buf(f"de{i - 1}", (n, hw, hw, ca))
            _lib.call("cgs_conv3x3_bwd_data", C.byref(d), _p(d_cur), _p(saved[f"am{i}"]), wptr, None, _lib.ACT_NONE,
                      _p(d_embeds[i - 1]) if has_add else None, n_add if has_add else 0, _p(nxt), None, _stream())
            d_cur = nxt
        elif dx is not None and n - dx_from > 0:
            m = n - dx_from
            dd = conv_desc(m, hw, ca, cb, co, False, ups, act, pool, _lib.Dropout(0.0, 0, 0, None))
            _lib.call("cgs_conv3x3_bwd_data", C.byref(dd), _p(d_cur[dx_from:]), _p(saved["am0"][dx_from:]), wptr, None,
                      _lib.ACT_NONE, None, 0, _p(dx), None, _stream())
    if enc1_wgrad is not None:
        raise _lib.CgsError("critic_backward: features.3's deferred weight gradient found no features.0 launch to ride with")
    if dec3_rider is not None:
        raise _lib.CgsError("critic_backward: dec_model.3's deferred weight gradient found no features.0 launch to ride with")
    return dx


# ------------------------------------------------------------------------------------------------
# masker (decoder + mask head)
# ------------------------------------------------------------------------------------------------
def masker_forward(flat: torch.Tensor, lay: Layout, x: torch.Tensor, embeds: List[torch.Tensor], n: int,
                   out: Optional[Dict[str, torch.Tensor]] = None, o4_done: bool = False,
                   keep_hm: bool = True, fp16_mask_head: bool = False,
                   zpart: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """x: NHWC uint8/fp32 image [n,64,64,3]; embeds = [e0,e1,e2,e3 (NHWC), e4 [n,32]].
    Returns o4 [n,32], o3..o0, hm [n,64,64,16], Z [n,64,64].  o4_done: out['o4'] was already produced by the critic's
    head kernel (critic_forward(..., pw=...)), skip the stand-alone 1x1 conv.  keep_hm=False (inference): the 16-channel
    masker.0 output is not needed afterwards -- masker.0 and masker.2 run as one kernel and 'hm' is never stored;
    fp16_mask_head (with keep_hm=False only, opt-in): that kernel's masker.0 GEMM takes fp16 operands (~1e-3 abs in Z).
    zpart [zpart_count(n), 2] (training): the mask layer also leaves its per-tile (sum |z|, sum z^2) there for the L1/L2 losses."""
    u8 = x.dtype == torch.uint8
    _chk(x, torch.uint8 if u8 else torch.float32, "masker image input")
    _chk_img(x, n, "masker image input")
    dev = x.device
    fp = flat.data_ptr()
    o = out if out is not None else {}
    if o4_done:
        assert o.get("o4") is not None, "o4_done: the caller's head kernel already produced out['o4']"
    else:
        if o.get("o4") is None:
            o["o4"] = torch.empty((n, 32), device=dev, dtype=torch.float32)
        _lib.call("cgs_pointwise_fwd", n, 32, 32, _p(embeds[4]), C.c_void_p(fp + 4 * lay.off("dec_model.4.weight")),
                  C.c_void_p(fp + 4 * lay.off("dec_model.4.bias")), _p(o["o4"]), _stream())
    names = ("o3", "o2", "o1", "o0", "hm", "Z")
    srcs_a = (embeds[3], embeds[2], embeds[1], embeds[0], x, None)
    prev = o["o4"]
    if fp16_mask_head and (keep_hm or not MASK_INFER_FUSED):
        raise _lib.CgsError("fp16_mask_head is an inference-only option of the one-kernel mask head (keep_hm=False)")
    if TAIL_FWD:      # dec_model.3 / .2 / .1 in one tail kernel
        for nm, shp in (("o3", (n, 4, 4, 16)), ("o2", (n, 8, 8, 8)), ("o1", (n, 16, 16, 8))):
            if o.get(nm) is None:
                o[nm] = torch.empty(shp, device=dev, dtype=torch.float32)
        tw = tail_dec_weights(flat, lay)
        m0pack = None
        if (keep_hm and mask_train_fused(n) and zpart is not None) or (not keep_hm and MASK_INFER_FUSED and not fp16_mask_head):
            # the mask head forward's weight registers, built once by a spare workgroup of this launch (the mask head follows two launches later;
            # inference runs the same kernel, storing nothing but Z)
            m0pack = o.get("m0pack")
            if m0pack is None:
                m0pack = o["m0pack"] = torch.empty(40 * 64, device=dev, dtype=torch.float32)
        dec0_done = False
        if DEC_TAIL_DEC0_FUSED:
            # one workgroup per image up to the library's own cap (tail_fwd_cap in csrc/tail.hip): beyond it the entry point answers
            # CGS_ERR_UNSUPPORTED and the two-launch form below runs -- the cap is not restated here
            if o.get("o0") is None:
                o["o0"] = torch.empty((n, 32, 32, 8), device=dev, dtype=torch.float32)
            rc = _lib.load().cgs_tail_dec_fwd_dec0(n, C.byref(tw), _p(embeds[0]), _p(embeds[1]), _p(embeds[2]), _p(embeds[3]), _p(o["o4"]),
                                                   _p(o["o3"]), _p(o["o2"]), _p(o["o1"]), C.c_void_p(fp + 4 * lay.off("dec_model.0.weight")),
                                                   C.c_void_p(fp + 4 * lay.off("dec_model.0.bias")), _p(o["o0"]),
                                                   C.c_void_p(fp + 4 * lay.off("masker.0.weight")), _p(m0pack), _stream())
            if rc == 0:
                dec0_done = True
            elif rc != _lib.ERR_UNSUPPORTED:
                _lib.check(rc, "cgs_tail_dec_fwd_dec0")
        if not dec0_done:
            _lib.call("cgs_tail_dec_fwd_pack", n, C.byref(tw), _p(embeds[1]), _p(embeds[2]), _p(embeds[3]), _p(o["o4"]),
                      _p(o["o3"]), _p(o["o2"]), _p(o["o1"]), C.c_void_p(fp + 4 * lay.off("masker.0.weight")), _p(m0pack), _stream())
        prev = o["o0"] if dec0_done else o["o1"]
    for name, sa, (key, hw, ca, cb, co, ups, act, pool, _s) in zip(names, srcs_a, DEC_LAYERS):
        if TAIL_FWD and (name in ("o3", "o2", "o1") or (name == "o0" and dec0_done)):
            continue
        if name == "hm" and not keep_hm and MASK_INFER_FUSED:
            if o.get("Z") is None:
                o["Z"] = torch.empty((n, 64, 64), device=dev, dtype=torch.float32)
            margs = (n, _lib.SRC_U8 if u8 else _lib.SRC_F32, _p(x), _p(prev),
                     C.c_void_p(fp + 4 * lay.off("masker.0.weight")), C.c_void_p(fp + 4 * lay.off("masker.0.bias")),
                     C.c_void_p(fp + 4 * lay.off("masker.2.weight")), C.c_void_p(fp + 4 * lay.off("masker.2.bias")), _p(o["Z"]))
            if fp16_mask_head:
                rc = _lib.load().cgs_mask_infer_fwd_f16(*margs, _stream())
            else:
                rc = _lib.load().cgs_mask_infer_fwd_packed(*margs, _p(o.get("m0pack")) if TAIL_FWD else None, _stream())
            if rc == 0:
                return o
            if rc != _lib.ERR_UNSUPPORTED:
                _lib.check(rc, "cgs_mask_infer_fwd")
            # unsupported in this build: fall through to the two-kernel form
        if name == "hm" and keep_hm and mask_train_fused(n) and zpart is not None:
            # training: masker.0 and masker.2 in one kernel -- h is stored once and never re-read for Z
            for k, shp in (("hm", (n, 64, 64, 16)), ("Z", (n, 64, 64))):
                if o.get(k) is None:
                    o[k] = torch.empty(shp, device=dev, dtype=torch.float32)
            _lib.call("cgs_mask_train_fwd_packed", n, _lib.SRC_U8 if u8 else _lib.SRC_F32, _p(x), _p(prev),
                      C.c_void_p(fp + 4 * lay.off("masker.0.weight")), C.c_void_p(fp + 4 * lay.off("masker.0.bias")),
                      C.c_void_p(fp + 4 * lay.off("masker.2.weight")), C.c_void_p(fp + 4 * lay.off("masker.2.bias")),
                      _p(o["hm"]), _p(o["Z"]), _p(zpart), _p(o.get("m0pack")) if TAIL_FWD else None, _stream())
            return o
        shape = (n, hw, hw) if co == 1 else (n, hw, hw, co)
        if o.get(name) is None:
            o[name] = torch.empty(shape, device=dev, dtype=torch.float32)
        a = sa if sa is not None else prev
        b = prev if cb > 0 else None
        d = conv_desc(n, hw, ca, cb, co, u8 and name == "hm", ups, act, pool, _lib.Dropout(0.0, 0, 0, None))
        _lib.call("cgs_conv3x3_fwd", C.byref(d), _p(a), _p(b), C.c_void_p(fp + 4 * lay.off(key + ".weight")),
                  C.c_void_p(fp + 4 * lay.off(key + ".bias")), _p(o[name]), _p(zpart) if name == "Z" else None, _stream())
        prev = o[name]
    return o


def zpart_count(n: int) -> int:
    """Number of (sum |z|, sum z^2) partial pairs the training mask layer writes for n images."""
    return _lib.load().cgs_mask_train_fwd_partials(n) if mask_train_fused(n) else 4 * n


def masker_backward(flat: torch.Tensor, lay: Layout, x: torch.Tensor, embeds: List[torch.Tensor], n: int,
                    saved: Dict[str, torch.Tensor], dzpre: torch.Tensor, plan: SlabPlan,
                    ws: Optional[Dict[str, torch.Tensor]] = None, side: "SideStream" = None,
                    pw_in_head: bool = False, defer_dec0: Optional[list] = None) -> List[torch.Tensor]:
    """dzpre: gradient w.r.t. the mask head's PRE-sigmoid output [n,64,64].
    Returns [dE0, dE1, dE2, dE3, dE4]: gradients w.r.t. the encoder embeds (skip connections).
    pw_in_head: stop at the bottleneck -- the 5th entry is then d o4 [n,32] (gradient w.r.t. the 1x1 conv's OUTPUT) and
    the caller hands it to critic_backward(..., pw_bwd=...), whose head kernel runs the 1x1 conv's backward."""
    u8 = x.dtype == torch.uint8
    dev = x.device
    fp = flat.data_ptr()
    ws = ws if ws is not None else {}
    side = side if side is not None else NO_SIDE
    lib = _lib.load()

    def buf(name, shape):
        t = ws.get(name)
        if t is None:
            t = ws[name] = torch.empty(shape, device=dev, dtype=torch.float32)
        return t

    nd = _lib.Dropout(0.0, 0, 0, None)
    names = ("o3", "o2", "o1", "o0", "hm", "Z")
    srcs_a = [embeds[3], embeds[2], embeds[1], embeds[0], x, saved["hm"]]
    srcs_b = [saved["o4"], saved["o3"], saved["o2"], saved["o1"], saved["o0"], None]
    d_embeds: List[Optional[torch.Tensor]] = [None] * 5
    dy = dzpre
    fused_head_do0 = None
    head_done = False    # masker.2 AND masker.0 fully handled by the mask-head kernel
    fuse_dec0 = None
    u8_or_f32_needs_da = False   # the image never needs a gradient on this path
    for li in (5, 4, 3, 2, 1, 0):
        if li == 4 and head_done:
            continue
        if TAIL_BWD and li == 2:
            # dec_model.1 / .2 / .3 backward in one tail kernel: dy = d o1 -> skip gradients dE1..dE3 and d o4
            nsl = lib.cgs_tail_dec_bwd_slabs(n)
            cnts = {1: 9 * 16 * 8 + 8, 2: 9 * 24 * 8 + 8, 3: 9 * 48 * 16 + 16}
            sl = {k: buf(f"slab_dec{3 - k}", (nsl, c)) for k, c in cnts.items()}     # names as the per-layer path: li = 3 - k
            for ei, shp in ((1, (n, 16, 16, 8)), (2, (n, 8, 8, 8)), (3, (n, 4, 4, 16))):
                d_embeds[ei] = buf(f"dE{ei}", shp)
            do4 = buf("do4", (n, 32))
            tw = tail_dec_weights(flat, lay)
            dec3_rider = (fuse_dec0 is not None and defer_dec0 is not None and DEC3_WGRAD_RIDER and u8 and not enc0_in_tail(False, True, None))
            nsl_k = {1: nsl, 2: nsl, 3: nsl}
            if dec3_rider:      # the layer's slab rows come from the riders: allocate their (few) rows instead of one per image
                nsl_k[3] = lib.cgs_dec3_wgrad_rider_slabs(n)
                sl[3] = buf("slab_dec0_rider", (nsl_k[3], cnts[3]))
            if fuse_dec0 is not None:      # dec_model.0's data gradient rides in front, one workgroup per image
                dy_o0, w0ptr, de0 = fuse_dec0
                do3 = buf("do3", (n, 4, 4, 16)) if dec3_rider else None
                _lib.call("cgs_dec0_tail_dec_bwd_do3", n, C.byref(tw), _p(dy_o0), w0ptr, _p(de0), _p(embeds[1]), _p(embeds[2]), _p(embeds[3]),
                          _p(saved["o4"]), _p(saved["o3"]), _p(saved["o2"]), _p(dy), _p(d_embeds[1]), _p(d_embeds[2]), _p(d_embeds[3]), _p(do4),
                          None if dec3_rider else _p(sl[3]), _p(sl[2]), _p(sl[1]), _p(do3), _stream())
                if dec3_rider:
                    defer_dec0.append({"dec3": (n, embeds[3], saved["o4"], do3, sl[3])})
            else:
                _lib.call("cgs_tail_dec_bwd", n, C.byref(tw), _p(embeds[1]), _p(embeds[2]), _p(embeds[3]), _p(saved["o4"]),
                          _p(saved["o3"]), _p(saved["o2"]), _p(dy), _p(d_embeds[1]), _p(d_embeds[2]), _p(d_embeds[3]), _p(do4),
                          _p(sl[3]), _p(sl[2]), _p(sl[1]), _stream())
            for k, c in cnts.items():
                plan.add(sl[k], nsl_k[k], c, lay.off(f"dec_model.{k}.weight"))
            dy = do4
            break
        key, hw, ca, cb, co, ups, act, pool, _s = DEC_LAYERS[li]
        is_img = li == 4
        d = conv_desc(n, hw, ca, cb, co, u8 and is_img, ups, act, pool, nd)
        cnt = 9 * (ca + cb) * co + co
        wptr = C.c_void_p(fp + 4 * lay.off(key + ".weight"))
        if _DEC_TAG[li] in BOTH_DEC and li != 5:
            nsl = lib.cgs_conv3x3_bwd_both_slabs(C.byref(d))
            if nsl < 0:
                _lib.check(nsl, "cgs_conv3x3_bwd_both_slabs")
            slab = buf(f"slab_dec{li}", (nsl, cnt))
            if li == 4:   # masker.0: only the upsampled decoder channels need a gradient
                de, db = None, buf("do0", (n, 32, 32, 8))
            else:
                ei = 3 - li
                de = buf(f"dE{ei}", (n, hw, hw, ca))
                db = buf(f"do{ei + 1}", (n, 32) if ups == 4 else (n, hw // 2, hw // 2, cb))
                d_embeds[ei] = de
            _lib.call("cgs_conv3x3_bwd_both", C.byref(d), _p(srcs_a[li]), _p(srcs_b[li]), _p(dy), None, wptr, None, 0,
                      _p(de), _p(db), _p(slab), _stream())
            plan.add(slab, nsl, cnt, lay.off(key + ".weight"))
            dy = db
            continue
        head_fused = li == 5 and MASK_HEAD_FUSED and not u8_or_f32_needs_da
        nsl = lib.cgs_mask_head_bwd_slabs(n) if head_fused else 0
        if nsl > 0:
            # the whole mask head in one launch: d o0, masker.2 and masker.0 weight gradients; d hm is never stored
            cnt0 = 9 * 11 * 16 + 16
            slab2, slab0 = buf("slab_dec5", (nsl, cnt)), buf("slab_dec4", (nsl, cnt0))
            do0 = buf("do0", (n, 32, 32, 8))
            _lib.call("cgs_mask_head_bwd", n, _lib.SRC_U8 if u8 else _lib.SRC_F32, _p(x), _p(saved["o0"]), _p(dzpre),
                      _p(saved["hm"]), wptr, C.c_void_p(fp + 4 * lay.off("masker.0.weight")), None, _p(do0),
                      _p(slab2), _p(slab0), _stream())
            plan.add(slab2, nsl, cnt, lay.off(key + ".weight"))
            plan.add(slab0, nsl, cnt0, lay.off("masker.0.weight"))
            dy = do0
            head_done = True
            continue
        nsl = lib.cgs_conv3x3_bwd_weight_slabs(C.byref(d))
        if nsl < 0:
            _lib.check(nsl, "cgs_conv3x3_bwd_weight_slabs")
        slab = buf(f"slab_dec{li}", (nsl, cnt))
        if li == 3 and defer_dec0 is not None and DEC0_WGRAD_RIDER:
            # dec_model.0's weight gradient: only the final reduction waits for it -- the caller hands it to the last critic pass's tail
            # launch, whose spare workgroup slots compute it (critic_backward(rider=...)): one persistent rider per CU
            nsl = min(nsl, DEC0_RIDERS)
            defer_dec0.append((n, srcs_a[li], srcs_b[li], dy, slab, nsl))
        else:
            with side.fork():
                _lib.call("cgs_conv3x3_bwd_weight", C.byref(d), _p(srcs_a[li]), _p(srcs_b[li]), _p(dy), None, _p(slab), _stream())
        plan.add(slab, nsl, cnt, lay.off(key + ".weight"))
        if head_fused:
            # (VALU build) masker.2 data gradient rebuilt inside the masker.0 data-gradient kernel
            dhm, do0 = buf("dhm", (n, 64, 64, 16)), buf("do0", (n, 32, 32, 8))
            _lib.call("cgs_mask_head_bwd", n, 0, None, None, _p(dzpre), _p(saved["hm"]), wptr,
                      C.c_void_p(fp + 4 * lay.off("masker.0.weight")), _p(dhm), _p(do0), None, None, _stream())
            dy = dhm
            fused_head_do0 = do0
        elif li == 5:    # masker.2: d hm = conv_bwd(dzpre) * LeakyReLU'(hm)
            dhm = buf("dhm", (n, 64, 64, 16))
            _lib.call("cgs_conv3x3_bwd_data", C.byref(d), _p(dy), None, wptr, _p(saved["hm"]), _lib.ACT_LRELU, None, 0,
                      _p(dhm), None, _stream())
            dy = dhm
        elif li == 4 and fused_head_do0 is not None:
            dy = fused_head_do0      # already produced together with d hm
        elif li == 4:  # masker.0: only the upsampled decoder channels need a gradient
            do0 = buf("do0", (n, 32, 32, 8))
            _lib.call("cgs_conv3x3_bwd_data", C.byref(d), _p(dy), None, wptr, None, _lib.ACT_NONE, None, 0, None, _p(do0), _stream())
            dy = do0
        else:          # dec_model.{0,1,2,3}: skip gradient at full res + upsample-backward sum
            ei = 3 - li
            de = buf(f"dE{ei}", (n, hw, hw, ca))
            shape_b = (n, 32) if ups == 4 else (n, hw // 2, hw // 2, cb)
            db = buf(f"do{ei + 1}", shape_b)
            if li == 3 and TAIL_BWD and DEC0_TAIL_BWD_FUSED and n <= lib.cgs_tail_dec_bwd_slabs(n) and lib.cgs_tail_dec_bwd_slabs(n) == n:
                fuse_dec0 = (dy, wptr, de)      # launched together with the decoder tail's backward below
            else:
                _lib.call("cgs_conv3x3_bwd_data", C.byref(d), _p(dy), None, wptr, None, _lib.ACT_NONE, None, 0, _p(de), _p(db), _stream())
            d_embeds[ei] = de
            dy = db
    # bottleneck 1x1 conv: dy = d o4 [n,32]
    if pw_in_head:
        d_embeds[4] = dy
        return d_embeds
    nsl = lib.cgs_pointwise_bwd_slabs(n)
    slab = buf("slab_pw", (nsl, PW_SLAB))
    de4 = buf("dE4", (n, 32))
    _lib.call("cgs_pointwise_bwd", n, 32, 32, _p(embeds[4]), _p(dy), C.c_void_p(fp + 4 * lay.off("dec_model.4.weight")),
              _p(de4), _p(slab), _stream())
    plan.add(slab, nsl, PW_SLAB, lay.off("dec_model.4.weight"))
    d_embeds[4] = de4
    return d_embeds
