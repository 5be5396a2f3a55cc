"""Orchestration of the shape-generic HIP kernels (csrc/gen.hip) for the model sizes the specialised kernels do not cover:

  * NewCritic / UnetDecoder with chfak != 1 (nets.py:160-212, 452-523; the paper's model is chfak = 5, docs/index.html:151):
    forward passes (eval mode) -- `-process`, `-eval`, extract_contrastive_data's sweep, the module API under no_grad;
  * the legacy single-module hourglass `Unet(upsample=False)` (nets.py:356-449): Conv + LeakyReLU(0.2) + MaxPool encoder,
    ConvTranspose2d decoder, forward pass; its ConvTranspose2d(4,2,1) kernels also have data / weight gradients.

Activations are NHWC fp32 on the device, weights come from the flat kernel-layout buffer (spec.Layout).  PyTorch is used for
device memory and the current stream only."""
import ctypes as C
from typing import Dict, List, Optional

import torch

from . import _lib
from .spec import Layout

UPS_FOLD = True             # forward of a layer over cat(A, nearest-up_2(B)) at hw >= 16: B staged at its own resolution, its 9 taps folded to 4 per pixel parity (gen4.hip, FOLD)
WGRAD_FOLD = True           # weight gradient of a layer over cat(A, nearest-up_2(B)) at hw >= 16: A's rows by the row-block kernel, B's by gen_wgrad_fold_kernel on B at its own resolution (4 folds per parity class instead of 9 taps)
DGRAD_UP2 = True            # data gradient towards a x2-upsampled source at the source's resolution (space-to-depth view of dY, 16 instead of 36 steps per cell; gen4.hip FOLD = 2)
ENC0_DEDICATED = True       # features.0 (3 -> 8 chfak channels at 64x64) of chfak 2 .. 5 on csrc/gen_enc0.hip (False: the shape-generic gen4 kernel)
_ACT = {"none": _lib.ACT_NONE, "relu": _lib.ACT_RELU, "lrelu": _lib.ACT_LRELU, "sigmoid": _lib.ACT_SIGMOID}


def _s():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


_PACKED: Dict[tuple, torch.Tensor] = {}


class PackJob(C.Structure):          # = cgs_gen_pack_job (include/cgs_hip.h)
    _fields_ = [("w", C.c_void_p), ("wp", C.c_void_p), ("ca", C.c_int32), ("cb", C.c_int32), ("co", C.c_int32), ("transposed", C.c_int32),
                ("ci_layer", C.c_int32), ("ci_off", C.c_int32)]


class PackPlan:
    """The 3x3 layers' weight operands of ONE training step, packed by one launch at the start of the step (the parameters only change
    in the optimiser step at its end).  First step: recording -- every pack_weights call packs as usual, into a buffer of its own, and
    registers its job.  After freeze(): launch() packs all of them at once and pack_weights returns the registered buffers."""

    def __init__(self):
        self.jobs: Dict[tuple, torch.Tensor] = {}
        self.order: List[tuple] = []
        self.frozen = False
        self._table = None

    def freeze(self):
        self.frozen = True
        arr = (PackJob * max(len(self.order), 1))()
        for i, key in enumerate(self.order):
            w_ptr, ca, cb, co, transposed, ci_layer, ci_off = key
            arr[i] = PackJob(w_ptr, self.jobs[key].data_ptr(), ca, cb, co, transposed, ci_layer, ci_off)
        self._table = arr

    def launch(self):
        if self.order:
            _lib.call("cgs_gen_conv_pack_batch", C.cast(self._table, C.c_void_p), len(self.order), _s())


_PLAN: Optional[PackPlan] = None


class pack_plan:
    """with pack_plan(plan): ... -- the body's pack_weights calls go through `plan` (see PackPlan)."""

    def __init__(self, plan: Optional[PackPlan]):
        self.plan = plan

    def __enter__(self):
        global _PLAN
        self.prev, _PLAN = _PLAN, self.plan
        if self.plan is not None and self.plan.frozen:
            self.plan.launch()
        return self.plan

    def __exit__(self, *exc):
        global _PLAN
        if self.plan is not None and not self.plan.frozen and exc[0] is None:
            self.plan.freeze()
        _PLAN = self.prev
        return False


def _pack(w_ptr: int, ca: int, cb: int, co: int, dev, transposed: bool, ci_layer: int, ci_off: int, scratch_key) -> torch.Tensor:
    """One packed operand: ci_layer == 0 -> cgs_gen_conv_pack_weights(ca, cb, co, transposed); ci_layer > 0 -> the window form
    cgs_gen_conv_pack_weights_window(co_layer = ca, ci_layer, ci_off, ci_n = co)."""
    plan = _PLAN
    key = (int(w_ptr), ca, cb, co, int(transposed), ci_layer, ci_off)
    if plan is not None and plan.frozen:
        wp = plan.jobs.get(key)
        if wp is not None:
            return wp                                   # packed by the plan's launch at the start of this step
    lib = _lib.load()
    nfl = int(lib.cgs_gen_conv_packed_floats_up2(ca, co) if int(transposed) == 3 else
              (lib.cgs_gen_conv_packed_floats_folded(ca, cb, co) if int(transposed) == 2 else lib.cgs_gen_conv_packed_floats(ca, cb, co)))
    if nfl <= 0:
        raise _lib.CgsError(f"generic conv: no packed form for ca={ca} cb={cb} co={co}")
    if plan is not None and not plan.frozen:
        wp = plan.jobs.get(key)
        if wp is None:
            wp = plan.jobs[key] = torch.empty(nfl, device=dev, dtype=torch.float32)
            plan.order.append(key)
    else:
        wp = _PACKED.get(scratch_key)
        if wp is None:
            wp = _PACKED[scratch_key] = torch.empty(nfl, device=dev, dtype=torch.float32)
    if int(transposed) == 3:
        _lib.call("cgs_gen_conv_pack_weights_up2", ca, ci_layer, ci_off, co, C.c_void_p(w_ptr), _p(wp), _s())
    elif ci_layer > 0:
        _lib.call("cgs_gen_conv_pack_weights_window", ca, ci_layer, ci_off, co, C.c_void_p(w_ptr), _p(wp), _s())
    else:
        _lib.call("cgs_gen_conv_pack_weights", ca, cb, co, int(transposed), C.c_void_p(w_ptr), _p(wp), _s())
    return wp


def pack_weights(w_ptr: int, ca: int, cb: int, co: int, dev, transposed=False) -> torch.Tensor:
    """HWIO 3x3 weights at w_ptr -> the convolution kernel's operand form (cgs_gen_conv_pack_weights).  transposed: w_ptr is the
    [9][co][ca] weight of the layer whose DATA GRADIENT is wanted (ca = its output channels, co = its input channels).  One scratch
    tensor per (device, shape): the pack and the convolution that reads it are consecutive launches of one stream -- or, inside a
    pack_plan, the plan's buffer for this layer."""
    di = dev.index if dev.index is not None else torch.cuda.current_device()
    return _pack(w_ptr, ca, cb, co, dev, transposed, 0, 0, (di, ca, cb, co, int(transposed) == 2))


def pack_weights_window(w_ptr: int, co_layer: int, ci_layer: int, ci_off: int, ci_n: int, dev) -> torch.Tensor:
    """The data gradient's operand for input channels [ci_off, ci_off + ci_n) of a layer (cgs_gen_conv_pack_weights_window)."""
    di = dev.index if dev.index is not None else torch.cuda.current_device()
    return _pack(w_ptr, co_layer, 0, ci_n, dev, True, ci_layer, ci_off, (di, "window", co_layer, ci_layer, ci_off, ci_n))


def pack_weights_up2(w_ptr: int, co_layer: int, ci_layer: int, ci_off: int, ci_n: int, dev) -> torch.Tensor:
    """The operand of cgs_gen_conv3x3_bwd_data_up2: dY [co_layer channels] -> the cell-summed gradient of the x2-upsampled input channels
    [ci_off, ci_off + ci_n) of a layer with HWIO weights [9][ci_layer][co_layer] at w_ptr."""
    di = dev.index if dev.index is not None else torch.cuda.current_device()
    return _pack(w_ptr, co_layer, 0, ci_n, dev, 3, ci_layer, ci_off, (di, "up2", co_layer, ci_layer, ci_off, ci_n))


def conv3x3(a: torch.Tensor, b: Optional[torch.Tensor], w_ptr: int, bias_ptr: int, co: int, act: str = "none",
            slope: float = 0.01, pool: bool = False, ups: int = 2, want_argmax: bool = False, out: Optional[torch.Tensor] = None,
            am: Optional[torch.Tensor] = None):
    """act(conv3x3(cat(a, up_ups(b))) + bias) (+ MaxPool2d(2)).  a: NHWC uint8 or fp32 [n,hw,hw,ca]; b: NHWC fp32 or None.
    out / am: optional preallocated outputs (am = the pooling argmax bytes, bit 2 set where a ReLU'd pooled value is <= 0)."""
    if not a.is_cuda or not a.is_contiguous() or a.dtype not in (torch.uint8, torch.float32):
        raise _lib.CgsError("generic conv: source A must be a contiguous uint8 / fp32 device tensor (no CPU fallback)")
    n, hw, ca = a.shape[0], a.shape[1], a.shape[3]
    cb = 0 if b is None else b.shape[-1]
    oh = hw // 2 if pool else hw
    if out is None:
        out = torch.empty((n, oh, oh, co), device=a.device, dtype=torch.float32)
    if am is None and pool and want_argmax:
        am = torch.empty((n, oh, oh, co), device=a.device, dtype=torch.uint8)
    if ENC0_DEDICATED and hw == 64 and ca == 3 and cb == 0 and pool and act == "relu" and co in (16, 24, 32, 40):
        # features.0 at chfak 2 .. 5: the kernel of its own (csrc/gen_enc0.hip: lane = pool cell, all weights in registers)
        _lib.call("cgs_gen_enc0_fwd", n, co, int(a.dtype == torch.uint8), _p(a), C.c_void_p(w_ptr), C.c_void_p(bias_ptr), _p(out), _p(am), _s())
        return (out, am) if want_argmax else out
    if UPS_FOLD and cb > 0 and ups == 2 and hw >= 16 and not pool:
        wp = pack_weights(w_ptr, ca, cb, co, a.device, transposed=2)      # (2 = the folded forward operand)
        _lib.call("cgs_gen_conv3x3_fwd_folded", n, hw, ca, cb, co, int(a.dtype == torch.uint8), _ACT[act], float(slope), _p(a), _p(b), _p(wp),
                  C.c_void_p(bias_ptr), _p(out), _s())
        return (out, am) if want_argmax else out
    wp = pack_weights(w_ptr, ca, cb, co, a.device)
    _lib.call("cgs_gen_conv3x3_fwd", n, hw, ca, cb, co, int(a.dtype == torch.uint8), ups, _ACT[act], float(slope), int(pool), _p(a),
              _p(b), _p(wp), C.c_void_p(bias_ptr), _p(out), _p(am), _s())
    return (out, am) if want_argmax else out


def gemm(x: torch.Tensor, w_ptr: int, bias_ptr: int, k: int, n_out: int, act: str = "none", slope: float = 0.01,
         out: Optional[torch.Tensor] = None) -> torch.Tensor:
    m = x.shape[0]
    if out is None:
        out = torch.empty((m, n_out), device=x.device, dtype=torch.float32)
    if k >= 256:       # long reductions (features.14: k = 16 x 16 chfak): four waves per tile split K (one wave walks it in 40 rounds of loads)
        _lib.call("cgs_gen_gemm_ex", m, k, n_out, _p(x), k, 1, C.c_void_p(w_ptr), n_out, 1, C.c_void_p(bias_ptr), _ACT[act], float(slope), 0,
                  _p(out), _s())
    else:
        _lib.call("cgs_gen_gemm", m, k, n_out, _ACT[act], float(slope), _p(x), C.c_void_p(w_ptr), C.c_void_p(bias_ptr), _p(out), _s())
    return out


def gemm_ex(m: int, k: int, n: int, x, sxm: int, sxk: int, w, swk: int, swn: int, out, bias=None, act: str = "none",
            accumulate: bool = False):
    """out [m,n] (+)= act(sum_k x[m sxm + k sxk] w[k swk + n swn] + bias); x / w / out / bias: tensors or raw device addresses."""
    q = lambda t: t if isinstance(t, C.c_void_p) else (C.c_void_p(t) if isinstance(t, int) else _p(t))
    _lib.call("cgs_gen_gemm_ex", m, k, n, q(x), sxm, sxk, q(w), swk, swn, q(bias), _ACT[act], 0.01, int(accumulate), q(out), _s())


def gemm_ex_batch(plan, ws, tag: str, dst_off: int, m: int, k: int, n: int, x, sxm: int, sxk: int, w, swk: int, swn: int, dev):
    """grad[dst_off : dst_off + m n] = sum_k x[m sxm + k sxk] w[k swk + n swn] for a reduction over the batch (k = images): the K
    range is split over up to 32 workgroup rows, one slab row each, summed in order by the plan's cgs_reduce_slabs launch -- a
    single 16 x 16 tile would walk all k images alone (50 us for crit.4.weight at k = 1536)."""
    nsplit = max(1, min(32, k // 64))
    slab = ws.buf("gslab_" + tag, (nsplit, m * n), dev)
    q = lambda t: t if isinstance(t, C.c_void_p) else (C.c_void_p(t) if isinstance(t, int) else _p(t))
    _lib.call("cgs_gen_gemm_ex_splitk", m, k, n, q(x), sxm, sxk, q(w), swk, swn, nsplit, _p(slab), _s())
    plan.add(slab, nsplit, m * n, dst_off)


def grad_fix(d: torch.Tensor, saved: Optional[torch.Tensor] = None, act: str = "none", slope: float = 0.01,
             addend: Optional[torch.Tensor] = None, drop: Optional[_lib.Dropout] = None):
    """In place: d = (d * dropout multiplier + addend (on its leading elements)) * act'(saved output)."""
    nd = drop if drop is not None else _lib.Dropout(0.0, 0, 0, None, 0, 0)
    _lib.call("cgs_gen_grad_fix", d.numel(), _p(d), _p(saved), _ACT[act], float(slope), _p(addend),
              0 if addend is None else addend.numel(), nd, _s())


ENC_KEYS = ("features.0", "features.3", "features.6", "features.10")
ENC_HW = (64, 32, 16, 8)
DROP_SITE_E2, DROP_SITE_E3, DROP_SITE_H1 = 0, 1, 2       # the specialised engine's Philox sites (spec.py)


def dims(chfak: int, neck: int = 32):
    return [8 * chfak, 8 * chfak, 8 * chfak, 16 * chfak], neck * chfak


class NoDrop:
    """Stand-in for hourglass.DropState with p = 0."""
    p = 0.0

    def desc(self, site, active=True, per_img4=0):
        return _lib.Dropout(0.0, 0, 0, None, 0, 0)


def critic_buffers(n: int, chfak: int, neck: int, dev, training: bool = False) -> Dict[str, torch.Tensor]:
    """Activation buffers of n images: e0..e3 (pooled, pre-dropout), am0..am3, e4, h1, pred (+ e2d / e3d / h1d: the dropped tensors
    the next layer consumed, kept for the weight gradients)."""
    d, nb = dims(chfak, neck)
    z = lambda *shape, dt=torch.float32: torch.zeros(shape, device=dev, dtype=dt)
    o = {}
    for i, hw in enumerate(ENC_HW):
        o[f"e{i}"] = z(n, hw // 2, hw // 2, d[i])
        o[f"am{i}"] = z(n, hw // 2, hw // 2, d[i], dt=torch.uint8)
    o["e4"], o["h1"], o["pred"] = z(n, nb), z(n, nb), z(n)
    if training:
        o["e2d"], o["e3d"], o["h1d"] = z(n, 8, 8, d[2]), z(n, 4, 4, d[3]), z(n, nb)
    return o


def critic_forward(flat: torch.Tensor, lay: Layout, x: torch.Tensor, chfak: int, neck: int = 32,
                   out: Optional[Dict[str, torch.Tensor]] = None, drop=None) -> Dict[str, torch.Tensor]:
    """NewCritic.forward (nets.py:197-212) for any chfak.  x: NHWC uint8 / fp32 [n,64,64,3].  Eval mode unless `drop` (a
    hourglass.DropState with p > 0) is given: Dropout then sits in front of features.10, features.14 and crit.4 as in the
    reference (nets.py:179,183,192).  out: buffers from critic_buffers (allocated when omitted).
    Returns e0..e3 (NHWC, before dropout), e4 [n,neck*chfak], h1, pred [n] (+ am*, e2d, e3d, h1d)."""
    fp = flat.data_ptr()
    off = lambda k: fp + 4 * lay.off(k)
    d, nb = dims(chfak, neck)
    n = x.shape[0]
    training = drop is not None and drop.p > 0.0
    o = out if out is not None else critic_buffers(n, chfak, neck, x.device, training)
    src = x
    for i, (key, co) in enumerate(zip(ENC_KEYS, d)):
        conv3x3(src, None, off(key + ".weight"), off(key + ".bias"), co, act="relu", pool=True, out=o[f"e{i}"], am=o[f"am{i}"])
        src = o[f"e{i}"]
        if training and i >= 2:
            dst = o["e2d" if i == 2 else "e3d"]
            _lib.call("cgs_gen_dropout_fwd", src[:n].numel(), _p(src), _p(dst),
                      drop.desc(DROP_SITE_E2 if i == 2 else DROP_SITE_E3, True, src[0].numel() // 4), _s())
            src = dst
    gemm(src.reshape(n, 16 * d[3]), off("features.14.weight"), off("features.14.bias"), 16 * d[3], nb, act="relu", out=o["e4"])
    gemm(o["e4"], off("crit.1.weight"), off("crit.1.bias"), nb, nb, act="relu", out=o["h1"])
    h = o["h1"]
    if training:
        _lib.call("cgs_gen_dropout_fwd", n * nb, _p(h), _p(o["h1d"]), drop.desc(DROP_SITE_H1, True, nb // 4), _s())
        h = o["h1d"]
    gemm(h, off("crit.4.weight"), off("crit.4.bias"), nb, 1, act="sigmoid", out=o["pred"].view(n, 1))
    return o


def masker_buffers(n: int, chfak: int, neck: int, dev, masker_channels: int = 16) -> Dict[str, torch.Tensor]:
    d, nb = dims(chfak, neck)
    z = lambda *shape: torch.zeros(shape, device=dev, dtype=torch.float32)
    return {"o4": z(n, nb), "o3": z(n, 4, 4, d[3]), "o2": z(n, 8, 8, d[2]), "o1": z(n, 16, 16, d[1]), "o0": z(n, 32, 32, d[0]),
            "hm": z(n, 64, 64, masker_channels), "Z": z(n, 64, 64)}


def masker_forward(flat: torch.Tensor, lay: Layout, x: torch.Tensor, embeds: List[torch.Tensor], chfak: int, neck: int = 32,
                   masker_channels: int = 16, out: Optional[Dict[str, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
    """UnetDecoder.forward (nets.py:494-523) for any chfak.  embeds = [e0..e3 NHWC, e4 [n,neck*chfak]]."""
    fp = flat.data_ptr()
    off = lambda k: fp + 4 * lay.off(k)
    d, nb = dims(chfak, neck)
    n = x.shape[0]
    o = out if out is not None else masker_buffers(n, chfak, neck, x.device, masker_channels)
    gemm(embeds[4], off("dec_model.4.weight"), off("dec_model.4.bias"), nb, nb, out=o["o4"])
    conv3x3(embeds[3], o["o4"].view(n, 1, 1, nb), off("dec_model.3.weight"), off("dec_model.3.bias"), d[3], ups=4, out=o["o3"])
    conv3x3(embeds[2], o["o3"], off("dec_model.2.weight"), off("dec_model.2.bias"), d[2], out=o["o2"])
    conv3x3(embeds[1], o["o2"], off("dec_model.1.weight"), off("dec_model.1.bias"), d[1], out=o["o1"])
    conv3x3(embeds[0], o["o1"], off("dec_model.0.weight"), off("dec_model.0.bias"), d[0], out=o["o0"])
    conv3x3(x, o["o0"], off("masker.0.weight"), off("masker.0.bias"), masker_channels, act="lrelu", slope=0.01, out=o["hm"])
    if masker_channels == 16:      # the mask layer (16 -> 1) has the same shape at every chfak: the fixed-shape kernel
        _lib.call("cgs_conv3x3_fwd", C.byref(_mask2_desc(n)), _p(o["hm"]), None, C.c_void_p(off("masker.2.weight")),
                  C.c_void_p(off("masker.2.bias")), _p(o["Z"]), None, _s())
    else:
        conv3x3(o["hm"], None, off("masker.2.weight"), off("masker.2.bias"), 1, act="sigmoid", out=o["Z"].view(n, 64, 64, 1))
    return o


def _mask2_desc(n: int) -> _lib.ConvDesc:
    return _lib.ConvDesc(n, 64, 64, 16, 0, 1, _lib.SRC_F32, 2, _lib.ACT_SIGMOID, 0, _lib.Dropout(0.0, 0, 0, None, 0, 0))


# ------------------------------------------------------------------------------------------------
# fp16 inference (BASELINE config 4): fp16 activations and weights, fp32 accumulation (csrc/gen_f16.hip); any chfak
# ------------------------------------------------------------------------------------------------
class F16Weights:
    """The fp16 operand copies of the 3x3 layers' weights for one parameter state (repacked when `version` moves on)."""

    def __init__(self):
        self.version = None
        self.w: Dict[str, torch.Tensor] = {}

    def get(self, flat_c: torch.Tensor, lc: Layout, flat_m: Optional[torch.Tensor], lm: Optional[Layout], chfak: int, neck: int,
            masker_channels: int, version) -> Dict[str, torch.Tensor]:
        if self.version == version and self.w:
            return self.w
        d, nb = dims(chfak, neck)
        shapes = {"features.0": (3, 0, d[0]), "features.3": (d[0], 0, d[1]), "features.6": (d[1], 0, d[2]), "features.10": (d[2], 0, d[3])}
        if flat_m is not None:
            shapes.update({"dec_model.3": (d[3], nb, d[3]), "dec_model.2": (d[2], d[3], d[2]), "dec_model.1": (d[1], d[2], d[1]),
                           "dec_model.0": (d[0], d[1], d[0]), "masker.0": (3, d[0], masker_channels), "masker.2": (masker_channels, 0, 1)})
        self.w = {}
        for key, (ca, cb, co) in shapes.items():
            flat, lay = (flat_c, lc) if key.startswith("features") else (flat_m, lm)
            nh = _lib.load().cgs_gen16_packed_weight_halves(ca, cb, co)
            w16 = torch.empty(nh, device=flat.device, dtype=torch.float16)
            _lib.call("cgs_gen16_pack_weights", ca, cb, co, C.c_void_p(flat.data_ptr() + 4 * lay.off(key + ".weight")), _p(w16), _s())
            self.w[key] = w16
        self.version = version
        return self.w


def _conv16(a, b, w16, bias_ptr, co, act="none", slope=0.01, pool=False, ups=2, out_f32=False):
    n, hw, ca = a.shape[0], a.shape[1], a.shape[3]
    cb = 0 if b is None else b.shape[-1]
    oh = hw // 2 if pool else hw
    out = torch.empty((n, oh, oh, co), device=a.device, dtype=torch.float32 if out_f32 else torch.float16)
    _lib.call("cgs_gen16_conv3x3_fwd", n, hw, ca, cb, co, int(a.dtype == torch.uint8), ups, _ACT[act], float(slope), int(pool), int(out_f32),
              _p(a), _p(b), _p(w16), C.c_void_p(bias_ptr), _p(out), _s())
    return out


def _gemm16(x, w_ptr, bias_ptr, k, n_out, act="none", out_f16=True):
    m = x.shape[0]
    out = torch.empty((m, n_out), device=x.device, dtype=torch.float16 if out_f16 else torch.float32)
    _lib.call("cgs_gen16_gemm", m, k, n_out, _ACT[act], 0.01, int(x.dtype == torch.float16), int(out_f16), _p(x), C.c_void_p(w_ptr),
              C.c_void_p(bias_ptr), _p(out), _s())
    return out


def infer_f16(flat_c: torch.Tensor, lc: Layout, flat_m: Optional[torch.Tensor], lm: Optional[Layout], x_u8: torch.Tensor, chfak: int,
              neck: int, w16: Dict[str, torch.Tensor], masker_channels: int = 16, embeds_from=None):
    """Eval-mode critic (+ masker) with fp16 activations / weights and fp32 accumulation.  x_u8: NHWC uint8 [n,64,64,3].
    Returns (pred [n] fp32, Z [n,64,64] fp32 or None, embeds) -- embeds (fp16 e0..e3, e4) feed a masker call (-separate)."""
    if x_u8.dtype != torch.uint8 or not x_u8.is_cuda or not x_u8.is_contiguous():
        raise _lib.CgsError("fp16 inference reads the uint8 frames (NHWC, contiguous, on the device)")
    cp = flat_c.data_ptr()
    offc = lambda k: cp + 4 * lc.off(k)
    d, nb = dims(chfak, neck)
    n = x_u8.shape[0]
    e, src = [], x_u8
    for key, co in zip(ENC_KEYS, d):
        src = _conv16(src, None, w16[key], offc(key + ".bias"), co, act="relu", pool=True)
        e.append(src)
    e4 = _gemm16(e[3].reshape(n, 16 * d[3]), offc("features.14.weight"), offc("features.14.bias"), 16 * d[3], nb, act="relu")
    h1 = _gemm16(e4, offc("crit.1.weight"), offc("crit.1.bias"), nb, nb, act="relu")
    pred = _gemm16(h1, offc("crit.4.weight"), offc("crit.4.bias"), nb, 1, act="sigmoid", out_f16=False).reshape(n)
    emb = e + [e4]
    if flat_m is None:
        return pred, None, emb
    if embeds_from is not None:
        emb = embeds_from
    mp = flat_m.data_ptr()
    offm = lambda k: mp + 4 * lm.off(k)
    o4 = _gemm16(emb[4], offm("dec_model.4.weight"), offm("dec_model.4.bias"), nb, nb)
    o3 = _conv16(emb[3], o4.view(n, 1, 1, nb), w16["dec_model.3"], offm("dec_model.3.bias"), d[3], ups=4)
    o2 = _conv16(emb[2], o3, w16["dec_model.2"], offm("dec_model.2.bias"), d[2])
    o1 = _conv16(emb[1], o2, w16["dec_model.1"], offm("dec_model.1.bias"), d[1])
    o0 = _conv16(emb[0], o1, w16["dec_model.0"], offm("dec_model.0.bias"), d[0])
    hm = _conv16(x_u8, o0, w16["masker.0"], offm("masker.0.bias"), masker_channels, act="lrelu", slope=0.01)
    Z = _conv16(hm, None, w16["masker.2"], offm("masker.2.bias"), 1, act="sigmoid", out_f32=True).reshape(n, 64, 64)
    return pred, Z, emb


# ------------------------------------------------------------------------------------------------
# backward passes (training at chfak != 1)
# ------------------------------------------------------------------------------------------------
class Workspace(dict):
    """Named scratch tensors that stay put between the eager warm-up step and the captured graph."""

    def buf(self, name: str, shape, dev, dtype=torch.float32) -> torch.Tensor:
        t = self.get(name)
        if t is None or tuple(t.shape) != tuple(shape):
            t = self[name] = torch.zeros(tuple(shape), device=dev, dtype=dtype)
        return t

    def ones(self, n: int, dev) -> torch.Tensor:
        """A vector of n ones, filled once (it used to be re-filled by a torch kernel in every step)."""
        name = f"ones_{n}"
        t = self.get(name)
        if t is None or t.numel() != max(n, 1):
            t = self[name] = torch.ones(max(n, 1), device=dev, dtype=torch.float32)
        return t


def _flip(ws: Workspace, flat: torch.Tensor, lay: Layout, key: str, ci: int, co: int) -> int:
    """The layer's HWIO weight pointer: _bwd_data packs it (taps reversed, channels transposed) for the data-gradient pass."""
    return flat.data_ptr() + 4 * lay.off(key + ".weight")


def _bwd_data(n, hw, co, ci, dy, am, w_ptr: int, out, addend=None):
    """d_cat [n,hw,hw,ci] of a 3x3 layer with HWIO weights [9][ci][co] at w_ptr from the gradient dy at its output."""
    for t in (dy, am, out, addend):
        if t is not None and not t.is_contiguous():
            raise _lib.CgsError("generic conv backward: dy / argmax / out / addend must be contiguous NHWC device tensors")
    if ENC0_DEDICATED and hw == 64 and ci == 3 and am is not None and addend is None and co in (16, 24, 32, 40):
        # the image gradient of features.0 at chfak 2 .. 5: csrc/gen_enc0.hip (lane = 2x2 cell of d x, all weight steps in registers)
        _lib.call("cgs_gen_enc0_bwd_data", n, co, _p(dy), _p(am), C.c_void_p(w_ptr), _p(out), _s())
        return
    wp = pack_weights(w_ptr, co, 0, ci, dy.device, transposed=True)
    _lib.call("cgs_gen_conv3x3_bwd_data", n, hw, co, ci, _p(dy), _p(am), _p(wp), _p(addend),
              0 if addend is None else addend.shape[0], _p(out), _s())


def _wgrad(plan, ws: Workspace, tag: str, dst_off: int, n, hw, a, b, ups, dy, am, co):
    """Weight + bias gradient of one 3x3 layer over n images: slab rows registered in `plan` for cgs_reduce_slabs."""
    ca, cb = a.shape[-1], (0 if b is None else b.shape[-1])
    cnt = 9 * (ca + cb) * co + co
    if ENC0_DEDICATED and hw == 64 and ca == 3 and cb == 0 and co in (16, 24, 32, 40):      # features.0 at chfak 2 .. 5: csrc/gen_enc0.hip
        nsl = _lib.load().cgs_gen_enc0_bwd_weight_slabs(n, co)
        slab = ws.buf("slab_" + tag, (nsl, cnt), a.device)
        _lib.call("cgs_gen_enc0_bwd_weight", n, co, int(a.dtype == torch.uint8), _p(a), _p(dy), _p(am), _p(slab), _s())
        plan.add(slab, nsl, cnt, dst_off)
        return
    if WGRAD_FOLD and cb > 0 and ups == 2 and am is None and hw >= 16:
        nsl = _lib.load().cgs_gen_conv3x3_bwd_weight_folded_slabs(n, hw, ca, cb, co)
        if nsl > 0:
            slab = ws.buf("slab_" + tag, (nsl, cnt), a.device)
            rc = _lib.load().cgs_gen_conv3x3_bwd_weight_folded(n, hw, ca, cb, co, int(a.dtype == torch.uint8), _p(a), _p(b), _p(dy), _p(slab), _s())
            if rc == 0:
                plan.add(slab, nsl, cnt, dst_off)
                return
            if rc != -1:        # (-1 = CGS_ERR_UNSUPPORTED, decided before anything is launched: the unfolded kernel below)
                raise _lib.CgsError(f"cgs_gen_conv3x3_bwd_weight_folded failed: {rc}")
    nsl = _lib.load().cgs_gen_conv3x3_bwd_weight_slabs(n, ca, cb, co)
    slab = ws.buf("slab_" + tag, (nsl, cnt), a.device)
    _lib.call("cgs_gen_conv3x3_bwd_weight", n, hw, ca, cb, co, int(a.dtype == torch.uint8), ups, _p(a), _p(b), _p(dy), _p(am),
              _p(slab), _s())
    plan.add(slab, nsl, cnt, dst_off)


def critic_grad_buffers(n: int, chfak: int, neck: int, dev) -> Dict[str, torch.Tensor]:
    d, nb = dims(chfak, neck)
    z = lambda *shape: torch.zeros(shape, device=dev, dtype=torch.float32)
    g = {f"dE{i}": z(n, hw // 2, hw // 2, d[i]) for i, hw in enumerate(ENC_HW)}
    g["dz2"], g["dz1"], g["dz14"] = z(n), z(n, nb), z(n, nb)
    return g


def critic_backward_data(flat: torch.Tensor, lay: Layout, chfak: int, neck: int, s: Dict[str, torch.Tensor],
                         g: Dict[str, torch.Tensor], dpred: torch.Tensor, ws: Workspace, drop=None,
                         d_embeds: Optional[List[torch.Tensor]] = None, dx: Optional[torch.Tensor] = None):
    """Data-gradient pass of critic_forward for the images of `s` (saved activations) from dpred [n] down to the gradients at
    the pooled layer outputs g[dE3..dE0] and the head's pre-activation gradients g[dz2, dz1, dz14] -- everything the weight
    gradients need.  d_embeds = [dE0..dE3, de4]: gradients arriving at the embeds from the decoder (for the leading images).
    dx: optional [n,64,64,3] gradient w.r.t. the (fp32) input images."""
    fp = flat.data_ptr()
    off = lambda k: fp + 4 * lay.off(k)
    d, nb = dims(chfak, neck)
    n = dpred.shape[0]
    drop = drop if drop is not None else NoDrop()
    de = d_embeds if d_embeds is not None else [None] * 5
    g["dz2"][:n].copy_(dpred)
    grad_fix(g["dz2"][:n], saved=s["pred"], act="sigmoid")
    gemm_ex(n, 1, nb, g["dz2"], 1, 0, off("crit.4.weight"), 0, 1, g["dz1"])                         # dh1 = dz2 (x) w2
    grad_fix(g["dz1"][:n], saved=s["h1"], act="relu", drop=drop.desc(DROP_SITE_H1, True, nb // 4))
    gemm_ex(n, nb, nb, g["dz1"], nb, 1, off("crit.1.weight"), 1, nb, g["dz14"])                     # de4 = dz1 W1^T
    grad_fix(g["dz14"][:n], saved=s["e4"], act="relu", addend=de[4])
    k14 = 16 * d[3]
    gemm_ex(n, nb, k14, g["dz14"], nb, 1, off("features.14.weight"), 1, nb, g["dE3"])               # d(dropped e3)
    p3 = drop.desc(DROP_SITE_E3, True, k14 // 4)
    if p3.p > 0.0 or de[3] is not None:
        grad_fix(g["dE3"][:n], addend=de[3], drop=p3)
    for i in (3, 2, 1):
        wf = _flip(ws, flat, lay, ENC_KEYS[i], d[i - 1], d[i])
        dropped = i == 3 and drop.p > 0.0
        _bwd_data(n, ENC_HW[i], d[i], d[i - 1], g[f"dE{i}"], s[f"am{i}"], wf, g[f"dE{i - 1}"],
                  addend=None if dropped else de[i - 1])
        if dropped:
            grad_fix(g[f"dE{i - 1}"][:n], addend=de[i - 1], drop=drop.desc(DROP_SITE_E2, True, 64 * d[2] // 4))
    if dx is not None:
        wf = _flip(ws, flat, lay, ENC_KEYS[0], 3, d[0])
        _bwd_data(n, 64, d[0], 3, g["dE0"], s["am0"], wf, dx)


def critic_backward_weights(grad: torch.Tensor, goff: int, lay: Layout, chfak: int, neck: int, s: Dict[str, torch.Tensor],
                            g: Dict[str, torch.Tensor], x: torch.Tensor, n: int, plan, ws: Workspace, tag: str,
                            training: bool = False):
    """Weight gradients of the critic over the n images of s / g / x (any number of passes' images, contiguous): the Linear
    layers as GEMMs straight into grad[goff + ...], the 3x3 layers as slabs registered in `plan` (offsets goff + ...)."""
    gp = grad.data_ptr() + 4 * goff
    dst = lambda k: gp + 4 * lay.off(k)
    d, nb = dims(chfak, neck)
    ones = ws.ones(n, x.device)
    h1 = s["h1d"] if training else s["h1"]
    e3 = s["e3d"] if training else s["e3"]
    e2 = s["e2d"] if training else s["e2"]
    k14 = 16 * d[3]
    dev = x.device
    gb = lambda key, *a: gemm_ex_batch(plan, ws, f"{tag}_{key}", goff + lay.off(key), *a, dev)
    gb("crit.4.weight", nb, n, 1, h1, 1, nb, g["dz2"], 1, 0)
    gb("crit.4.bias", 1, n, 1, ones, 0, 1, g["dz2"], 1, 0)
    gb("crit.1.weight", nb, n, nb, s["e4"], 1, nb, g["dz1"], nb, 1)
    gb("crit.1.bias", 1, n, nb, ones, 0, 1, g["dz1"], nb, 1)
    gb("features.14.weight", k14, n, nb, e3, 1, k14, g["dz14"], nb, 1)
    gb("features.14.bias", 1, n, nb, ones, 0, 1, g["dz14"], nb, 1)
    srcs = [x, s["e0"], s["e1"], e2]
    for i in range(4):
        _wgrad(plan, ws, f"{tag}_enc{i}", goff + lay.off(ENC_KEYS[i] + ".weight"), n, ENC_HW[i], srcs[i], None, 2,
               g[f"dE{i}"], s[f"am{i}"], d[i])


def masker_backward(flat: torch.Tensor, lay: Layout, grad: torch.Tensor, goff: int, x: torch.Tensor, embeds: List[torch.Tensor],
                    m: Dict[str, torch.Tensor], dzpre: torch.Tensor, chfak: int, neck: int, plan, ws: Workspace,
                    masker_channels: int = 16, need_embed_grads: bool = True) -> List[Optional[torch.Tensor]]:
    """Backward of masker_forward from dzpre [n,64,64] (gradient at the pre-sigmoid mask): weight gradients (slabs in `plan`,
    the 1x1 bottleneck conv straight into grad) and the gradients [dE0, dE1, dE2, dE3, de4] at the embeds."""
    fp = flat.data_ptr()
    off = lambda k: fp + 4 * lay.off(k)
    d, nb = dims(chfak, neck)
    n, dev, mc = x.shape[0], x.device, masker_channels
    dzp = dzpre.view(n, 64, 64, 1)
    # masker.2 (16 -> 1, sigmoid handled by the caller) and the LeakyReLU(0.01) in front of it
    d_hm = ws.buf("d_hm", (n, 64, 64, mc), dev)
    if mc == 16:       # fixed-shape kernels of the mask layer (LeakyReLU' fused into the data gradient)
        d2 = _mask2_desc(n)
        nsl = _lib.load().cgs_conv3x3_bwd_weight_slabs(C.byref(d2))
        slab = ws.buf("slab_mask2", (nsl, 9 * mc + 1), dev)
        _lib.call("cgs_conv3x3_bwd_weight", C.byref(d2), _p(m["hm"]), None, _p(dzpre), None, _p(slab), _s())
        plan.add(slab, nsl, 9 * mc + 1, goff + lay.off("masker.2.weight"))
        _lib.call("cgs_conv3x3_bwd_data", C.byref(d2), _p(dzpre), None, C.c_void_p(off("masker.2.weight")), _p(m["hm"]), _lib.ACT_LRELU,
                  None, 0, _p(d_hm), None, _s())
    else:
        _wgrad(plan, ws, "mask2", goff + lay.off("masker.2.weight"), n, 64, m["hm"], None, 2, dzp, None, 1)
        _bwd_data(n, 64, 1, mc, dzp, None, _flip(ws, flat, lay, "masker.2", mc, 1), d_hm)
        grad_fix(d_hm, saved=m["hm"], act="lrelu", slope=0.01)
    # masker.0 over cat(X, up2(o0))
    _wgrad(plan, ws, "mask0", goff + lay.off("masker.0.weight"), n, 64, x, m["o0"], 2, d_hm, None, mc)
    # only the decoder channels of cat(X, up2(o0)) need a gradient: the operand covers input channels [3, 3 + d0) and the epilogue sums
    # the 2x2 cells (no d_cat tensor, no gradient for the image channels)
    d_o = ws.buf("d_o0", (n, 32, 32, d[0]), dev)
    rc = _lib.ERR_UNSUPPORTED
    if DGRAD_UP2 and mc % 4 == 0:
        wpu = pack_weights_up2(off("masker.0.weight"), mc, 3 + d[0], 3, d[0], dev)
        rc = _lib.load().cgs_gen_conv3x3_bwd_data_up2(n, 64, mc, d[0], _p(d_hm), _p(wpu), _p(d_o), _s())
    if rc == _lib.ERR_UNSUPPORTED:
        wpw = pack_weights_window(off("masker.0.weight"), mc, 3 + d[0], 3, d[0], dev)
        rc = _lib.load().cgs_gen_conv3x3_bwd_data_split(n, 64, mc, 0, d[0], 2, _p(d_hm), _p(wpw), None, _p(d_o), _s())
    if rc == _lib.ERR_UNSUPPORTED:
        dcat = ws.buf("dcat_m0", (n, 64, 64, 3 + d[0]), dev)
        _bwd_data(n, 64, mc, 3 + d[0], d_hm, None, _flip(ws, flat, lay, "masker.0", 3 + d[0], mc), dcat)
        _lib.call("cgs_gen_cat_split", n, 64, 3, d[0], 2, _p(dcat), None, _p(d_o), _s())
    else:
        _lib.check(rc, "cgs_gen_conv3x3_bwd_data_split")
    # linear trunk dec_model.0 .. dec_model.3 over cat(e_i, up(o_{i+1}))
    d_emb: List[Optional[torch.Tensor]] = [None] * 5
    lows = [m["o1"], m["o2"], m["o3"], m["o4"].view(n, 1, 1, nb)]
    for i, hw in enumerate((32, 16, 8, 4)):
        key, ups, low = f"dec_model.{i}", (4 if i == 3 else 2), lows[i]
        ca, cb = d[i], low.shape[-1]
        _wgrad(plan, ws, f"dec{i}", goff + lay.off(key + ".weight"), n, hw, embeds[i], low, ups, d_o, None, d[i])
        d_emb[i] = ws.buf(f"dEmb{i}", (n, hw, hw, ca), dev) if need_embed_grads else None
        d_low = ws.buf(f"d_o{i + 1}", (n, hw // ups, hw // ups, cb), dev)
        rc = _lib.ERR_UNSUPPORTED
        if DGRAD_UP2 and ups == 2 and hw >= 16 and d[i] % 4 == 0:
            # the low-resolution gradient at its own resolution (16 instead of 36 steps per cell); the skip gradient (if anybody wants it) from
            # the window of the skip channels alone
            wpu = pack_weights_up2(off(key + ".weight"), d[i], ca + cb, ca, cb, dev)
            rc = _lib.load().cgs_gen_conv3x3_bwd_data_up2(n, hw, d[i], cb, _p(d_o), _p(wpu), _p(d_low), _s())
            if rc == 0 and d_emb[i] is not None:
                wpa = pack_weights_window(off(key + ".weight"), d[i], ca + cb, 0, ca, dev)
                _lib.call("cgs_gen_conv3x3_bwd_data", n, hw, d[i], ca, _p(d_o), None, _p(wpa), None, 0, _p(d_emb[i]), _s())
        if rc == _lib.ERR_UNSUPPORTED:
            # d_cat straight as (skip gradient, cell-summed low-resolution gradient) where the kernel's output passes allow it
            wp = pack_weights(_flip(ws, flat, lay, key, ca + cb, d[i]), d[i], 0, ca + cb, dev, transposed=True)
            rc = _lib.load().cgs_gen_conv3x3_bwd_data_split(n, hw, d[i], ca, cb, ups, _p(d_o), _p(wp), _p(d_emb[i]), _p(d_low), _s())
        if rc == _lib.ERR_UNSUPPORTED:
            dcat = ws.buf(f"dcat_d{i}", (n, hw, hw, ca + cb), dev)
            _bwd_data(n, hw, d[i], ca + cb, d_o, None, _flip(ws, flat, lay, key, ca + cb, d[i]), dcat)
            _lib.call("cgs_gen_cat_split", n, hw, ca, cb, ups, _p(dcat), _p(d_emb[i]), _p(d_low), _s())
        else:
            _lib.check(rc, "cgs_gen_conv3x3_bwd_data_split")
        d_o = d_low
    # dec_model.4: o4 = e4 W + b (1x1 convolution on the 1x1 map)
    d_o4 = d_o.view(n, nb)
    gp = grad.data_ptr() + 4 * goff
    ones = ws.ones(n, dev)
    gemm_ex_batch(plan, ws, "dec4w", goff + lay.off("dec_model.4.weight"), nb, n, nb, embeds[4], 1, nb, d_o4, nb, 1, dev)
    gemm_ex_batch(plan, ws, "dec4b", goff + lay.off("dec_model.4.bias"), 1, n, nb, ones, 0, 1, d_o4, nb, 1, dev)
    if need_embed_grads:
        d_emb[4] = ws.buf("dEmb4", (n, nb), dev)
        gemm_ex(n, nb, nb, d_o4, nb, 1, off("dec_model.4.weight"), 1, nb, d_emb[4])
    return d_emb


# ------------------------------------------------------------------------------------------------
# ConvTranspose2d(4, 2, 1) over cat(a, b)
# ------------------------------------------------------------------------------------------------
def convt_weight_to_kernel(w: torch.Tensor) -> torch.Tensor:
    """PyTorch ConvTranspose2d weight [ci][co][ky][kx] -> kernel layout [ky][kx][ci][co] (flat)."""
    return w.permute(2, 3, 0, 1).contiguous().reshape(-1)


def convt_weight_from_kernel(v: torch.Tensor, ci: int, co: int) -> torch.Tensor:
    return v.reshape(4, 4, ci, co).permute(2, 3, 0, 1).contiguous()


def convt_fwd(a: torch.Tensor, b: Optional[torch.Tensor], wk: torch.Tensor, bias: torch.Tensor, act: str = "none", slope: float = 0.2):
    n, h, ca = a.shape[0], a.shape[1], a.shape[3]
    cb = 0 if b is None else b.shape[3]
    co = bias.numel()
    out = torch.empty((n, 2 * h, 2 * h, co), device=a.device, dtype=torch.float32)
    _lib.call("cgs_gen_convt4s2_fwd", n, h, ca, cb, co, _ACT[act], float(slope), _p(a), _p(b), _p(wk), _p(bias), _p(out), _s())
    return out


def convt_bwd(a: torch.Tensor, b: Optional[torch.Tensor], wk: torch.Tensor, dy: torch.Tensor):
    """dy: gradient at the PRE-activation output [n,2h,2h,co].  Returns (da, db or None, dw (kernel layout), dbias)."""
    n, h, ca = a.shape[0], a.shape[1], a.shape[3]
    cb = 0 if b is None else b.shape[3]
    co = dy.shape[3]
    da = torch.empty_like(a)
    db = torch.empty_like(b) if b is not None else None
    dw = torch.empty(16 * (ca + cb) * co, device=a.device, dtype=torch.float32)
    dbias = torch.empty(co, device=a.device, dtype=torch.float32)
    _lib.call("cgs_gen_convt4s2_bwd_data", n, h, ca, cb, co, _p(dy), _p(wk), _p(da), _p(db), _s())
    _lib.call("cgs_gen_convt4s2_bwd_weight", n, h, ca, cb, co, _p(a), _p(b), _p(dy), _p(dw), _p(dbias), _s())
    return da, db, dw, dbias
