"""BASELINE config 5: "128x128x3 frames batch=256 (upscaled Hourglass) ... bf16 with MFMA 1x1 pointwise" -- a BUILD-DEFINED variant.

The reference cannot take 128x128 frames: its 4x4 valid convolution (/root/reference/nets.py:184) would see an 8x8 map and
Flatten -> Linear (nets.py:189-190) shape-errors.  This module defines the smallest change that keeps every
other layer's shape -- one extra Conv2d(8,8,3)+ReLU+MaxPool stage in front of the encoder, one extra Upsample+cat+Conv2d stage
behind the decoder -- and runs

  * infer()        the eval-mode forward pass (critic value + mask, the -process path), and
  * phase2_step()  (round 4) the mask-training step with the loss arrangement of main.py:364-429 (critic on [B|A], mask on A,
                   replaced / injected mixes, critic on the mixes, lfak MSE(A) + MSE(replaced, B) + MSE(injected, A) + L1 |Z|,
                   Adam on critic + masker), Dropout off

on bf16 kernels: bf16 activations and activation gradients in HBM, fp32 accumulation, fp32 master weights / Adam state in ONE flat
buffer (kernel layout: HWIO weights followed by their bias, k-major GEMM matrices).  At chfak 1 (the benchmarked configuration):
  * the 128x128 / 64x64 / 32x32 levels run on compile-time-shaped whole-strip kernels -- forward and data gradients with the step's
    element-wise neighbours fused (csrc/hconv.hip: h5conv_kernel, v_mfma_f32_16x16x32_bf16), weight gradients over K = 32 pixels through
    ds_read_b64_tr_b16 (csrc/hwgrad.hip);
  * the 16x16-and-smaller levels have exactly the shapes of the 64x64 model's lower levels and run on ITS fp32 per-image tail kernels
    (csrc/tail.hip: cgs_tail_enc_fwd / _bwd, cgs_tail_dec_fwd / _bwd, cgs_tail_head_wgrad); their activations are fp32.
Other channel counts (chfak != 1, neck != 32): the shape-generic 16-bit family (csrc/gen_f16.hip forward / data gradient = the same
kernel on the flipped / transposed operand; csrc/gen_bf16_train.hip weight gradient and element-wise steps; the 4x4 valid convolution,
the Linear layers and the decoder's 1x1 pointwise convolution as MFMA GEMMs).
PARITY UNPINNED: there is no reference counterpart; tests compare against the build's own fp32 CPU restatement and its autograd
(oracle/hourglass_ref.py, hourglass128_apply / hourglass128_phase2_loss) with a stated bf16 tolerance.  Not wired into main.py (the
reference's CLI has no such size); `bench.py --config 5 [--mode train]` measures it."""
import ctypes as C
import os
from typing import Dict, Optional

import torch

from . import _lib
from . import generic as gen
from . import hourglass as hg
from . import parallel
from .generic import _ACT, _p, _s

TAIL = True                 # the 16x16-and-smaller layers of chfak 1 (same shapes as the 64x64 model's) on the fp32 per-image tail kernels (csrc/tail.hip)
TAIL_INFER_H16 = os.environ.get("CGS_C5_TAIL_H16", "1") != "0"      # inference: the two tail launches as ONE, on fp16 tiles (csrc/tail_infer.hip; A/B switch)
H5CONV = True               # the 128x128 layers of chfak 1 (masker.0 / masker.2 forward, the three data gradients) on h5conv_kernel (csrc/hconv.hip)
MIX_VIRTUAL = True          # the two mixes are formed inside features.0's forward / weight-gradient staging (no fp32 mix tensor, no mix kernel)
MIX_BWD_FUSED = True        # features.0's image gradient of the two mixes + cgs_mix_bwd as one kernel (cgs_bf16_enc0_bwd_mix)
POOL_FUSED = True           # ... and the pooled gradients of features.0 / features.3 re-expanded inside their consumers (no cgs_bf16_pool_expand)
HWGRAD = True               # weight gradients of the 128x128 / 64x64 layers of chfak 1 on csrc/hwgrad.hip (False: the shape-generic kernel)
ENC0_DIRECT = True          # features.0 of chfak 1 on cgs_bf16_enc0_fwd (False: the generic bf16 convolution; r4 A/B)
ENC_KEYS = ("features.0", "features.3", "features.6", "features.9", "features.13")
ENC_HW = (128, 64, 32, 16, 8)            # pre-pool map size of the five encoder stages
GEMM_KEYS = ("features.17", "crit.1", "crit.4")
class MixSrc:
    """The virtual mix batch [replaced | injected] of n frame pairs: features.0's kernels form it from (A, B, Z) while they stage."""

    def __init__(self, A, B, Z):
        self.A, self.B, self.Z = A, B, Z


_NODROP = _lib.Dropout(0.0, 0, 0, None, 0, 0)        # Dropout off (the tail kernels take the 64x64 model's three sites)


def _hwio(w: torch.Tensor) -> torch.Tensor:
    return w.permute(2, 3, 1, 0).contiguous().reshape(-1)          # OIHW -> HWIO (the kernels' weight order)


class Hourglass128:
    """One parameter set (dicts of fp32 tensors with the key names of oracle.critic128_shapes / masker128_shapes) as a flat fp32
    master buffer in the kernels' layouts + the bf16 operand copies of the 3x3 layers."""

    def __init__(self, critic_params: Dict[str, torch.Tensor], masker_params: Dict[str, torch.Tensor], device="cuda:0", chfak: int = 1,
                 neck: int = 32, masker_channels: int = 16, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, lfak: float = 5.0,
                 L1: float = 0.5, L2: float = 0.0, process_group=None, force_allreduce: bool = False, dp_graph: Optional[bool] = None):
        if not torch.cuda.is_available():
            raise _lib.CgsError("Hourglass128 needs an MI355X (HIP device); there is no CPU fallback")
        self.lib = _lib.load()
        self.dev = dev = torch.device(device)
        d = [8 * chfak, 8 * chfak, 8 * chfak, 8 * chfak, 16 * chfak]
        self.d, self.nb, self.mc = d, neck * chfak, masker_channels
        self.h5 = chfak == 1 and masker_channels == 16       # the dedicated 128x128 kernels are compiled for these channel counts
        self.tail = self.h5 and neck == 32                   # ... and the tail kernels for the 64x64 model's 16x16-and-smaller shapes
        nb = self.nb
        self.lr, self.b1, self.b2, self.eps, self.lfak, self.L1, self.L2 = lr, betas[0], betas[1], eps, lfak, L1, L2
        self.pg = process_group
        self.world = torch.distributed.get_world_size(process_group) if process_group is not None else 1
        # data parallel (one replica per rank, the flat gradient bucket all-reduced before Adam; force_allreduce: also on a 1-rank group, the
        # rehearsal of the N > 1 launch form).  dp_graph: record the collective in the step's HIP graph when RCCL allows (parallel.collective_capturable)
        self.dp = process_group is not None and (self.world > 1 or force_allreduce)
        self.dp_graph, self.dp_single_graph, self.dp_capture_note = parallel.resolve_dp_graph(dp_graph, self.world), False, None
        f = lambda t: t.detach().to(dev, torch.float32).contiguous()
        # ---- layout: (key, kind, shape info) in flat order; conv = [9 ci co | co], gemm = [k n | n] ----
        self.convs = {}      # key -> (ca, cb, co)
        cin = 3
        for key, co in zip(ENC_KEYS, d):
            self.convs[key] = (cin, 0, co)
            cin = co
        self.convs["dec_model.4"] = (d[4], nb, d[4])
        for i in (3, 2, 1, 0):
            self.convs[f"dec_model.{i}"] = (d[i], d[i + 1], d[i])
        self.convs["masker.0"] = (3, d[0], masker_channels)
        self.convs["masker.2"] = (masker_channels, 0, 1)
        self.gemms = {"features.17": (16 * d[4], nb), "crit.1": (nb, nb), "crit.4": (nb, 1), "dec_model.5": (nb, nb)}
        self.critic_keys = list(ENC_KEYS) + list(GEMM_KEYS)
        self.masker_keys = ["dec_model.5", "dec_model.4", "dec_model.3", "dec_model.2", "dec_model.1", "dec_model.0", "masker.0", "masker.2"]
        self.off, total = {}, 0
        for key in self.critic_keys + self.masker_keys:
            cnt = (9 * (self.convs[key][0] + self.convs[key][1]) * self.convs[key][2] + self.convs[key][2]) if key in self.convs else \
                  (self.gemms[key][0] * self.gemms[key][1] + self.gemms[key][1])
            self.off[key] = (total, cnt)
            total += (cnt + 3) // 4 * 4
        self.total = total
        self.flat = torch.zeros(total, device=dev)
        self.grad, self.m, self.v = torch.zeros_like(self.flat), torch.zeros_like(self.flat), torch.zeros_like(self.flat)
        self.step_t = torch.zeros(1, device=dev, dtype=torch.int64)
        self.load_state(critic_params, masker_params)
        self.w16, self.w16T = {}, {}
        for key, (ca, cb, co) in self.convs.items():
            self.w16[key] = torch.empty(self.lib.cgs_gen16_packed_weight_halves(ca, cb, co), device=dev, dtype=torch.bfloat16)
            self.w16T[key] = torch.empty(self.lib.cgs_gen16_packed_weight_halves(co, 0, ca + cb), device=dev, dtype=torch.bfloat16)
        self._pack_weights(transposed=False)
        self._train = None
        torch.cuda.current_stream().synchronize()

    # ---- parameters -----------------------------------------------------------------------------------------------------------
    def _wview(self, key):
        o, cnt = self.off[key]
        nbias = self.convs[key][2] if key in self.convs else self.gemms[key][1]
        return self.flat[o:o + cnt - nbias], self.flat[o + cnt - nbias:o + cnt]

    def load_state(self, critic_params=None, masker_params=None):
        f = lambda t: t.detach().to(self.dev, torch.float32).contiguous()
        with torch.no_grad():
            for P in (critic_params, masker_params):
                if P is None:
                    continue
                for key in self.critic_keys + self.masker_keys:
                    if key + ".weight" not in P:
                        continue
                    w, b = self._wview(key)
                    W = f(P[key + ".weight"])
                    if key in self.convs:
                        w.copy_(_hwio(W))
                    elif key == "features.17":      # 4x4 valid convolution: k = (y*4+x)*c
                        w.copy_(W.permute(2, 3, 1, 0).contiguous().reshape(-1))
                    elif key == "dec_model.5":      # 1x1 convolution
                        w.copy_(W.reshape(self.nb, self.nb).t().contiguous().reshape(-1))
                    else:
                        w.copy_(W.t().contiguous().reshape(-1))
                    b.copy_(f(P[key + ".bias"]))
        if self.pg is not None:
            parallel.broadcast_params_(self.flat, self.pg)
        if getattr(self, "w16", None):
            self._pack_weights(transposed=self._train is not None)

    def state_dicts(self):
        """(critic, masker) parameter dicts in the shapes of oracle.critic128_shapes / masker128_shapes (fp32 master weights)."""
        out = ({}, {})
        for which, keys in enumerate((self.critic_keys, self.masker_keys)):
            for key in keys:
                w, b = self._wview(key)
                if key in self.convs:
                    ca, cb, co = self.convs[key]
                    W = w.reshape(3, 3, ca + cb, co).permute(3, 2, 0, 1)
                elif key == "features.17":
                    W = w.reshape(4, 4, self.d[4], self.nb).permute(3, 2, 0, 1)
                elif key == "dec_model.5":
                    W = w.reshape(self.nb, self.nb).t().reshape(self.nb, self.nb, 1, 1)
                else:
                    k, n = self.gemms[key]
                    W = w.reshape(k, n).t()
                out[which][key + ".weight"], out[which][key + ".bias"] = W.contiguous().clone(), b.clone()
        return out

    def _packs_unused(self) -> bool:
        """chfak 1 with every dedicated path on: no layer runs on the shape-generic 16-bit convolution, so its bf16 operand copies are never read
        (the dedicated kernels convert the fp32 master weights themselves)."""
        return bool(self.tail and TAIL and H5CONV and HWGRAD and POOL_FUSED and ENC0_DIRECT)

    def _pack_weights(self, transposed: bool):
        """bf16 operand copies of every 3x3 layer from the fp32 master weights: ONE launch (a device job table, built once)."""
        key = bool(transposed)
        tabs = self.__dict__.setdefault("_pack_tables", {})
        if key not in tabs:
            jobs = []
            for k, (ca, cb, co) in self.convs.items():
                w, _ = self._wview(k)
                jobs.append(_lib.Pack16Job(w.data_ptr(), self.w16[k].data_ptr(), ca, cb, co, 0))
                if transposed:
                    jobs.append(_lib.Pack16Job(w.data_ptr(), self.w16T[k].data_ptr(), ca, cb, co, 1))
            arr = (_lib.Pack16Job * len(jobs))(*jobs)
            tabs[key] = (torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.dev), len(jobs))
        tab, n = tabs[key]
        _lib.call("cgs_genbf16_pack_batch", _p(tab), n, _s())

    # ---- layer wrappers ---------------------------------------------------------------------------------------------------------
    def _conv(self, key, a, b, co, act="none", pool=False, ups=2, out_f32=False, out=None, codes=None, w16=None, a_kind=None):
        n, hw, ca = a.shape[0], a.shape[1], a.shape[3]
        cb = 0 if b is None else b.shape[-1]
        oh = hw // 2 if pool else hw
        if out is None:
            out = torch.empty((n, oh, oh, co), device=a.device, dtype=torch.float32 if out_f32 else torch.bfloat16)
        if a_kind is None:
            a_kind = 1 if a.dtype == torch.uint8 else (2 if a.dtype == torch.float32 else 0)
        if ENC0_DIRECT and key == "features.0" and w16 is None and (ca, cb, co, hw) == (3, 0, 8, 128) and pool and act == "relu" \
                and not out_f32 and a_kind in (1, 2):
            # chfak 1: the whole-strip kernel of the fp16 inference path in bfloat16 (csrc/hconv.hip), weights from the fp32 master copy
            w, bias = self._wview(key)
            _lib.call("cgs_bf16_enc0_fwd", n, _p(a), int(a_kind == 2), _p(w), _p(bias), _p(out), _p(codes), _s())
            return out
        if H5CONV and self.h5 and w16 is None and out_f32 and a_kind == 0 and key == "features.6" and (hw, ca, cb, co) == (32, 8, 0, 8) and pool \
                and act == "relu":       # (fp32 output: the tail kernels' input)
            w, bias = self._wview(key)
            _lib.call("cgs_bf16_h5conv", _lib.H5_ENC2_FWD, n, _p(a), None, _p(w), _p(bias), _p(out), _p(codes), _s())
            return out
        if H5CONV and self.h5 and w16 is None and not out_f32 and a_kind == 0:
            w, bias = self._wview(key)
            if key == "dec_model.1" and (hw, ca, cb, co, ups) == (32, 8, 8, 8, 2) and not pool and act == "none":
                which = _lib.H5_DEC1_FWD_F32B if b.dtype == torch.float32 else _lib.H5_DEC1_FWD      # fp32 o2: straight from the tail kernel
                _lib.call("cgs_bf16_h5conv", which, n, _p(a), _p(b), _p(w), _p(bias), _p(out), None, _s())
                return out
            if key == "features.3" and (hw, ca, cb, co) == (64, 8, 0, 8) and pool and act == "relu":
                _lib.call("cgs_bf16_h5conv", _lib.H5_ENC1_FWD, n, _p(a), None, _p(w), _p(bias), _p(out), _p(codes), _s())
                return out
            if key == "dec_model.0" and (hw, ca, cb, co, ups) == (64, 8, 8, 8, 2) and not pool and act == "none":
                _lib.call("cgs_bf16_h5conv", _lib.H5_DEC0_FWD, n, _p(a), _p(b), _p(w), _p(bias), _p(out), None, _s())
                return out
        bias = self._wview(key)[1] if w16 is None else self._zero_bias(co)
        _lib.call("cgs_genbf16_conv3x3_fwd_train", n, hw, ca, cb, co, a_kind, ups, _ACT[act], 0.01, int(pool), int(out_f32), _p(a), _p(b),
                  _p(self.w16[key] if w16 is None else w16), _p(bias), _p(out), _p(codes), _s())
        return out

    def _mask_head_fwd(self, n, x_u8, o0, hm, Z):
        """masker.0 + masker.2 of chfak 1 on the whole-strip kernels (weights straight from the fp32 master copy)."""
        w0, b0 = self._wview("masker.0")
        w2, b2 = self._wview("masker.2")
        _lib.call("cgs_bf16_mask0_fwd", n, _p(x_u8), _p(o0), _p(w0), _p(b0), _p(hm), _s())
        _lib.call("cgs_bf16_mask2_fwd", n, _p(hm), _p(w2), _p(b2), _p(Z), _s())

    def _zero_bias(self, co):
        z = getattr(self, "_zb", None)
        if z is None or z.numel() < co:
            z = self._zb = torch.zeros(max(64, co), device=self.dev)
        return z

    def _gemm(self, key, x, k, n_out, act="none", out_bf16=True, out=None):
        m = x.shape[0]
        if out is None:
            out = torch.empty((m, n_out), device=x.device, dtype=torch.bfloat16 if out_bf16 else torch.float32)
        w, b = self._wview(key)
        _lib.call("cgs_genbf16_gemm", m, k, n_out, _ACT[act], 0.01, int(x.dtype == torch.bfloat16), int(out_bf16), _p(x), _p(w), _p(b),
                  _p(out), _s())
        return out

    # ---- inference (eval-mode forward) ------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def infer(self, x_u8: torch.Tensor):
        """x_u8: NHWC uint8 [n,128,128,3] on the device.  Returns (pred [n] fp32, Z [n,128,128] fp32)."""
        if x_u8.dtype != torch.uint8 or not x_u8.is_cuda or not x_u8.is_contiguous() or tuple(x_u8.shape[1:]) != (128, 128, 3):
            raise _lib.CgsError("Hourglass128.infer reads uint8 frames [n,128,128,3] (NHWC, contiguous, on the device)")
        n, d, nb = x_u8.shape[0], self.d, self.nb
        e, src = [], x_u8
        if TAIL and self.tail:      # the 16x16-and-smaller levels: two fp32 per-image launches (csrc/tail.hip) instead of eleven
            f32 = lambda *sh: torch.empty(sh, device=self.dev, dtype=torch.float32)
            i32 = lambda *sh: torch.empty(sh, device=self.dev, dtype=torch.int32)
            for i in (0, 1):
                src = self._conv(ENC_KEYS[i], src, None, d[i], act="relu", pool=True)
                e.append(src)
            e2 = self._conv("features.6", src, None, d[2], act="relu", pool=True, out_f32=True)
            tw, tdw = self._tail_enc_w(True), self._tail_dec_w()
            if TAIL_INFER_H16:      # (round 6) both of them in one launch per image on fp16 LDS tiles (csrc/tail_infer.hip, written for config 4)
                pred, o2 = f32(n), f32(n, 16, 16, 8)
                _lib.call("cgs_tail_infer_h16", n, C.byref(tw), C.byref(tdw), _p(e2), _p(pred), _p(o2), _s())
            else:
                e3, am3, e4, am4, e5, h1, pred, o5 = f32(n, 8, 8, 8), i32(n, 8, 8, 1), f32(n, 4, 4, 16), i32(n, 4, 4, 2), f32(n, 32), f32(n, 32), f32(n), f32(n, 32)
                _lib.call("cgs_tail_enc_fwd", n, C.byref(tw), _p(e2), _p(e3), _p(am3), _p(e4), _p(am4), _p(e5), _p(h1), _p(pred), _p(o5),
                          _NODROP, _NODROP, _NODROP, _s())
                o4, o3, o2 = f32(n, 4, 4, 16), f32(n, 8, 8, 8), f32(n, 16, 16, 8)
                _lib.call("cgs_tail_dec_fwd", n, C.byref(tdw), _p(e2), _p(e3), _p(e4), _p(o5), _p(o4), _p(o3), _p(o2), _s())
            if H5CONV:      # dec_model.1 reads the tail kernel's fp32 o2 directly
                o = o2
            else:
                o = torch.empty((n, 16, 16, 8), device=self.dev, dtype=torch.bfloat16)
                _lib.call("cgs_bf16_convert", n * 256, 8, 8, 0, _p(o2), _p(o), _s())
            lower = (1, 0)
        else:
            for key, co in zip(ENC_KEYS, d):
                src = self._conv(key, src, None, co, act="relu", pool=True)
                e.append(src)
            e5 = self._gemm("features.17", e[4].reshape(n, 16 * d[4]), 16 * d[4], nb, act="relu")
            h1 = self._gemm("crit.1", e5, nb, nb, act="relu")
            pred = self._gemm("crit.4", h1, nb, 1, act="sigmoid", out_bf16=False).reshape(n)
            o = self._gemm("dec_model.5", e5, nb, nb)                              # the 1x1 pointwise convolution: an MFMA GEMM
            o = self._conv("dec_model.4", e[4], o.view(n, 1, 1, nb), d[4], ups=4)
            lower = (3, 2, 1, 0)
        for i in lower:
            o = self._conv(f"dec_model.{i}", e[i], o, d[i])
        if H5CONV and self.h5:
            hm = torch.empty((n, 128, 128, 16), device=self.dev, dtype=torch.bfloat16)
            Z = torch.empty((n, 128, 128), device=self.dev)
            self._mask_head_fwd(n, x_u8, o, hm, Z)
        else:
            hm = self._conv("masker.0", x_u8, o, self.mc, act="lrelu")
            Z = self._conv("masker.2", hm, None, 1, act="sigmoid", out_f32=True).reshape(n, 128, 128)
        return pred, Z

    # ---- training (round 4) ---------------------------------------------------------------------------------------------------------
    class _Train:
        """Buffers of a phase-2 step for n A-images (+ n B-images): everything preallocated, the step is a static launch sequence."""

        def __init__(self, net: "Hourglass128", n: int):
            dev, d, nb = net.dev, net.d, net.nb
            bf = lambda *s: torch.empty(s, device=dev, dtype=torch.bfloat16)
            f32 = lambda *s: torch.empty(s, device=dev, dtype=torch.float32)
            u8 = lambda *s: torch.empty(s, device=dev, dtype=torch.uint8)
            self.n = n
            n4 = 4 * n
            self.ab = u8(2 * n, 128, 128, 3)              # [B | A]
            self.y = f32(n)
            self.mixed = self.dmixed = None          # fp32 mixes / their gradient: only the paths without MIX_VIRTUAL / MIX_BWD_FUSED materialise them
            # critic activations for the 4n slots [B | A | replaced | injected]
            self.e = [bf(n4, hw // 2, hw // 2, co) for hw, co in zip(ENC_HW, d)]
            self.codes = [u8(n4, hw // 2, hw // 2, co) for hw, co in zip(ENC_HW, d)]
            self.e4f = f32(n4, 16 * d[4])
            if TAIL and net.tail:       # fp32 activations of the levels the tail kernels own (their names: e1 e2 am2 e3 am3 | o4 o3 o2 o1)
                i32 = lambda *s: torch.empty(s, device=dev, dtype=torch.int32)
                self.t_e2, self.t_e3, self.t_am3 = f32(n4, 16, 16, 8), f32(n4, 8, 8, 8), i32(n4, 8, 8, 1)
                self.t_e4, self.t_am4 = f32(n4, 4, 4, 16), i32(n4, 4, 4, 2)
                self.t_o5, self.t_o4, self.t_o3, self.t_o2 = f32(2 * n, 32), f32(n, 4, 4, 16), f32(n, 8, 8, 8), f32(n, 16, 16, 8)
            self.e5, self.h1, self.pred = f32(n4, nb), f32(n4, nb), f32(n4)
            # masker
            self.o5 = bf(n, nb)
            self.o = [bf(n, 64, 64, d[0]), bf(n, 32, 32, d[1]), bf(n, 16, 16, d[2]), bf(n, 8, 8, d[3]), bf(n, 4, 4, d[4])]   # o0 .. o4
            self.hm, self.Z = bf(n, 128, 128, net.mc), f32(n, 128, 128)
            self.nzpart = net.lib.cgs_mix_fwd_partials(n, 16384)
            self.zsum, self.losses, self.dpred = f32(2 * self.nzpart), torch.zeros(8, device=dev), f32(n4)
            self.dzpre = f32(n, 128, 128)
            self.ws: Dict[str, torch.Tensor] = {}
            self.gws = gen.Workspace()
            self.plan_a = self.plan_b = None

        def buf(self, name, shape, dtype=torch.bfloat16, dev=None):
            t = self.ws.get(name)
            if t is None:
                t = self.ws[name] = torch.empty(shape, device=dev, dtype=dtype)
            return t

    # ---- the fp32 tail kernels of the 64x64 model (csrc/tail.hip) on this model's identical lower levels:
    #      theirs features.6 / .10 / .14, dec_model.4 (1x1) / .3 / .2 / .1  =  ours features.9 / .13 / .17, dec_model.5 / .4 / .3 / .2
    def _tail_enc_w(self, with_pw: bool):
        w = lambda k: self._wview(k)[0].data_ptr()
        b = lambda k: self._wview(k)[1].data_ptr()
        return _lib.TailEncWeights(w("features.9"), b("features.9"), w("features.13"), b("features.13"), w("features.17"), b("features.17"),
                                   w("crit.1"), b("crit.1"), w("crit.4"), b("crit.4"),
                                   w("dec_model.5") if with_pw else None, b("dec_model.5") if with_pw else None)

    def _tail_dec_w(self):
        w = lambda k: self._wview(k)[0].data_ptr()
        b = lambda k: self._wview(k)[1].data_ptr()
        return _lib.TailDecWeights(w("dec_model.4"), b("dec_model.4"), w("dec_model.3"), b("dec_model.3"), w("dec_model.2"), b("dec_model.2"))

    def _critic_forward_tail(self, T, src, lo: int, hi: int, want_o5: bool):
        n, d = hi - lo, self.d
        x = src
        for i in (0, 1):
            if i == 0 and isinstance(src, MixSrc):
                w, bias = self._wview(ENC_KEYS[0])
                _lib.call("cgs_bf16_enc0_fwd_mix", n // 2, _p(src.A), _p(src.B), _p(src.Z), _p(w), _p(bias), _p(T.e[0][lo:hi]), _p(T.codes[0][lo:hi]), _s())
                x = T.e[0][lo:hi]
                continue
            x = self._conv(ENC_KEYS[i], x, None, d[i], act="relu", pool=True, out=T.e[i][lo:hi], codes=T.codes[i][lo:hi])
        self._conv("features.6", x, None, d[2], act="relu", pool=True, out_f32=True, out=T.t_e2[lo:hi], codes=T.codes[2][lo:hi])
        tw = self._tail_enc_w(want_o5)
        _lib.call("cgs_tail_enc_fwd", n, C.byref(tw), _p(T.t_e2[lo:hi]), _p(T.t_e3[lo:hi]), _p(T.t_am3[lo:hi]), _p(T.t_e4[lo:hi]),
                  _p(T.t_am4[lo:hi]), _p(T.e5[lo:hi]), _p(T.h1[lo:hi]), _p(T.pred[lo:hi]), _p(T.t_o5[lo:hi]) if want_o5 else None,
                  _NODROP, _NODROP, _NODROP, _s())

    def _critic_forward(self, T, src, lo: int, hi: int, want_o5: bool = False):
        """Critic on slots [lo, hi): src = the frames / mixes of those slots (uint8 or fp32 NHWC)."""
        if TAIL and self.tail:
            return self._critic_forward_tail(T, src, lo, hi, want_o5)
        n, d, nb = hi - lo, self.d, self.nb
        x = src
        for i, (key, co) in enumerate(zip(ENC_KEYS, d)):
            x = self._conv(key, x, None, co, act="relu", pool=True, out=T.e[i][lo:hi], codes=T.codes[i][lo:hi])
        e4 = T.e[4][lo:hi].reshape(n, 16 * d[4])
        _lib.call("cgs_bf16_convert", n, 16 * d[4], 16 * d[4], 1, _p(e4), _p(T.e4f[lo:hi]), _s())      # fp32 copy for the head's weight gradient
        self._gemm("features.17", e4, 16 * d[4], nb, act="relu", out_bf16=False, out=T.e5[lo:hi])
        self._gemm("crit.1", T.e5[lo:hi], nb, nb, act="relu", out_bf16=False, out=T.h1[lo:hi])
        self._gemm("crit.4", T.h1[lo:hi], nb, 1, act="sigmoid", out_bf16=False, out=T.pred[lo:hi])

    def _masker_forward(self, T, A):
        n, d, nb = T.n, self.d, self.nb
        ea = [t[n:2 * n] for t in T.e]
        if TAIL and self.tail:      # dec_model.5 came out of the critic's tail kernel; .4 / .3 / .2 in one launch, fp32
            tdw = self._tail_dec_w()
            _lib.call("cgs_tail_dec_fwd", n, C.byref(tdw), _p(T.t_e2[n:2 * n]), _p(T.t_e3[n:2 * n]), _p(T.t_e4[n:2 * n]), _p(T.t_o5[n:2 * n]),
                      _p(T.t_o4), _p(T.t_o3), _p(T.t_o2), _s())
            _lib.call("cgs_bf16_convert", n * 256, 8, 8, 0, _p(T.t_o2), _p(T.o[2]), _s())      # (bf16 o2: dec_model.1's weight gradient reads it too)
            lower = (1, 0)
        else:
            self._gemm("dec_model.5", T.e5[n:2 * n], nb, nb, out=T.o5)
            self._conv("dec_model.4", ea[4], T.o5.view(n, 1, 1, nb), d[4], ups=4, out=T.o[4])
            lower = (3, 2, 1, 0)
        for i in lower:
            self._conv(f"dec_model.{i}", ea[i], T.o[i + 1], d[i], out=T.o[i])
        if H5CONV and self.h5:
            self._mask_head_fwd(n, A, T.o[0], T.hm, T.Z)
            return
        self._conv("masker.0", A, T.o[0], self.mc, act="lrelu", out=T.hm)
        self._conv("masker.2", T.hm, None, 1, act="sigmoid", out_f32=True, out=T.Z.view(n, 128, 128, 1))

    def _wgrad(self, T, plan, key, tag, n, hw, a, b, ups, dy, dyc=None, dy_f32=None):
        """dW / db slabs of layer `key`; dy_f32: the fp32 single-channel gradient the dedicated kernel reads instead of the padded bf16 dy."""
        ca, cb, co = self.convs[key]
        cnt = 9 * (ca + cb) * co + co
        a_kind = 1 if a.dtype == torch.uint8 else (2 if a.dtype == torch.float32 else 0)
        nsl = self.lib.cgs_bf16_hwgrad_slabs(n, hw, ca, cb, co) if HWGRAD and (cb == 0 or ups == 2) and (co > 1 or dy_f32 is not None) else 0
        if nsl > 0:          # the large-map layers of chfak 1: csrc/hwgrad.hip
            slab = T.buf(f"slab_{key}_{tag}", (nsl, cnt), torch.float32, self.dev)
            _lib.call("cgs_bf16_hwgrad", n, hw, ca, cb, co, a_kind, _p(a), _p(b), _p(dy_f32 if co == 1 else dy), _p(slab), _s())
            plan.add(slab, nsl, cnt, self.off[key][0])
            return
        nsl = self.lib.cgs_bf16_conv3x3_bwd_weight_slabs(n, hw, ca, cb)
        slab = T.buf(f"slab_{key}_{tag}", (nsl, cnt), torch.float32, self.dev)
        _lib.call("cgs_bf16_conv3x3_bwd_weight", n, hw, ca, cb, co, dyc if dyc is not None else co, a_kind, ups, _p(a), _p(b), _p(dy), _p(slab), _s())
        plan.add(slab, nsl, cnt, self.off[key][0])

    def _dgrad(self, T, key, dy, name, out_f32=False, dy_channels=None):
        """d cat(A, up(B)) [n,hw,hw,ca+cb] of layer `key` from dy [n,hw,hw,co (padded to dy_channels)]."""
        ca, cb, co = self.convs[key]
        n, hw = dy.shape[0], dy.shape[1]
        out = T.buf(name, (n, hw, hw, ca + cb), torch.float32 if out_f32 else torch.bfloat16, self.dev)
        _lib.call("cgs_genbf16_conv3x3_fwd_train", n, hw, dy_channels if dy_channels is not None else co, 0, ca + cb, 0, 1, _lib.ACT_NONE, 0.01, 0,
                  int(out_f32), _p(dy), None, _p(self.w16T[key]), _p(self._zero_bias(ca + cb)), _p(out), None, _s())
        return out

    def _gemm_bwd(self, T, plan, key, tag, x, g, need_dx=True):
        """y = x [n,k] . W [k,m] + b: registers dW / db slabs (split over the batch), returns d x = g . W^T [n,k] (fp32)."""
        k, m = self.gemms[key]
        n = x.shape[0]
        w, _ = self._wview(key)
        o = self.off[key][0]
        gen.gemm_ex_batch(plan, T.gws, f"{key}_{tag}_w", o, k, n, m, x, 1, k, g, m, 1, self.dev)                  # x^T . g
        ones = T.ws.get("ones")
        if ones is None:
            ones = T.ws["ones"] = torch.ones(max(4 * T.n, 64), device=self.dev)
        gen.gemm_ex_batch(plan, T.gws, f"{key}_{tag}_b", o + k * m, 1, n, m, ones, 0, 1, g, m, 1, self.dev)       # column sums
        if not need_dx:
            return None
        dx = T.buf(f"dx_{key}_{tag}", (n, k), torch.float32, self.dev)
        gen.gemm_ex(n, m, k, g, m, 1, w, 1, m, dx)                                                                # g . W^T
        return dx

    def _head_backward(self, T, plan, tag, lo, hi, d_e5_add):
        """The critic head's backward as GEMMs (the path without the tail kernels): returns d e4 (pooled, bf16)."""
        n, d, nb = hi - lo, self.d, self.nb
        dz = T.buf(f"dz_{tag}", (n, 1), torch.float32, self.dev)
        dz.copy_(T.dpred[lo:hi].reshape(n, 1))
        gen.grad_fix(dz, T.pred[lo:hi].reshape(n, 1), act="sigmoid")
        dh1 = self._gemm_bwd(T, plan, "crit.4", tag, T.h1[lo:hi], dz)
        gen.grad_fix(dh1, T.h1[lo:hi], act="relu")
        de5 = self._gemm_bwd(T, plan, "crit.1", tag, T.e5[lo:hi], dh1)
        gen.grad_fix(de5, T.e5[lo:hi], act="relu", addend=d_e5_add)
        de4f = self._gemm_bwd(T, plan, "features.17", tag, T.e4f[lo:hi], de5)
        dp = T.buf(f"dp4_{tag}", (n, 4, 4, d[4]), torch.bfloat16, self.dev)
        _lib.call("cgs_bf16_convert", n, 16 * d[4], 16 * d[4], 0, _p(de4f), _p(dp), _s())
        return dp

    def _head_wgrad(self, T, ranges, plan, plan_pw, tag):
        """Weight gradients of features.17 / crit.1 / crit.4 (+ dec_model.5 where a range carries d o5) from the per-image vectors the tail
        backward kernel(s) left: ONE launch for up to two image ranges (cgs_tail_head_wgrad sums them into the same slab rows)."""
        assert 1 <= len(ranges) <= 2
        f32 = torch.float32
        total = sum(r[0] for r in ranges)
        has_pw = any(r[3] is not None for r in ranges)
        nh = self.lib.cgs_tail_head_wgrad_slabs(total)
        slh = T.buf(f"slab_thead_{tag}", (nh, hg.HEAD_SLAB), f32, self.dev)
        slpw = T.buf(f"slab_tpw_{tag}", (nh, hg.PW_SLAB), f32, self.dev) if has_pw else None
        r0 = ranges[0]
        r1 = ranges[1] if len(ranges) > 1 else (0, None, None, None, 0)
        _lib.call("cgs_tail_head_wgrad", r0[0], _p(r0[1]), _p(r0[2]), _p(r0[3]), r0[4], r1[0], _p(r1[1]), _p(r1[2]), _p(r1[3]), r1[4],
                  _p(slh), _p(slpw), _s())
        plan.add(slh, nh, hg.HEAD_SLAB, self.off["features.17"][0])
        if has_pw:
            plan_pw.add(slpw, nh, hg.PW_SLAB, self.off["dec_model.5"][0])        # (a masker parameter: the overwriting plan)

    def _critic_backward(self, T, plan, tag, src, lo, hi, d_e5_add=None, skips=None, want_dx=None, plan_pw=None, mix_bwd=None, head_sink=None):
        """Backward of the critic on slots [lo, hi) from T.dpred; skips = [dskip_0..4] gradients arriving at the embeds from the
        decoder (bf16), d_e5_add fp32 [n,nb]; want_dx: fp32 [n,128,128,3] output for the image gradient."""
        n, d, nb = hi - lo, self.d, self.nb
        if TAIL and self.tail:
            # head + features.13 + features.9 in one launch from dpred: d e2 (fp32, the decoder's skip gradient included); skips[2..4] / d_e5_add
            # are the fp32 tensors cgs_tail_dec_bwd left (dE1 dE2 dE3 / d_o4 in its names)
            f32 = torch.float32
            nsl = self.lib.cgs_tail_enc_bwd_slabs(n)
            sl13, sl9 = T.buf(f"slab_t13_{tag}", (nsl, 1168), f32, self.dev), T.buf(f"slab_t9_{tag}", (nsl, 584), f32, self.dev)
            hvec, de2 = T.buf(f"hvec_{tag}", (n, 384), f32, self.dev), T.buf(f"de2f_{tag}", (n, 16, 16, 8), f32, self.dev)
            has = skips is not None
            tw = self._tail_enc_w(has)
            _lib.call("cgs_tail_enc_bwd", n, C.byref(tw), _p(T.t_e2[lo:hi]), _p(T.t_e3[lo:hi]), _p(T.t_am3[lo:hi]), _p(T.t_e4[lo:hi]),
                      _p(T.t_am4[lo:hi]), _p(T.e5[lo:hi]), _p(T.h1[lo:hi]), _p(T.pred[lo:hi]), _p(T.dpred[lo:hi]), None, 0.0, 0,
                      _p(skips[2]) if has else None, _p(skips[3]) if has else None, _p(skips[4]) if has else None,
                      _p(d_e5_add) if has else None, n if has else 0, _p(de2), _p(hvec), _p(sl13), _p(sl9), _NODROP, _NODROP, _NODROP, _s())
            plan.add(sl13, nsl, 1168, self.off["features.13"][0])
            plan.add(sl9, nsl, 584, self.off["features.9"][0])
            rng = (n, hvec, T.e5[lo:hi], d_e5_add if has else None, n if has else 0)
            if head_sink is not None:
                head_sink.append(rng)        # the caller forms the head's weight gradients of all its passes in ONE launch (_head_wgrad)
            else:
                self._head_wgrad(T, [rng], plan, plan_pw or plan, tag)
            dp = T.buf(f"dp2_{tag}", (n, 16, 16, d[2]), torch.bfloat16, self.dev)
            _lib.call("cgs_bf16_convert", n * 256, 8, 8, 0, _p(de2), _p(dp), _s())
            levels = (2, 1, 0)
            if has:
                skips = [skips[0], skips[1], None, None, None]
        else:
            levels = (4, 3, 2, 1, 0)
            dp = self._head_backward(T, plan, tag, lo, hi, d_e5_add)
        for i in levels:
            key, hw, co = ENC_KEYS[i], ENC_HW[i], d[i]
            if (i <= 1 or (i == 2 and TAIL and self.tail)) and H5CONV and HWGRAD and POOL_FUSED and self.h5:
                # the 128x128 / 64x64 (/ 32x32) levels: weight and data gradient read the pooled gradient + argmax bytes (no re-expanded copy)
                a = src if i == 0 else T.e[i - 1][lo:hi]
                add, cod = (_p(skips[i]) if skips is not None and skips[i] is not None else None), _p(T.codes[i][lo:hi])
                nsl, cnt = self.lib.cgs_bf16_hwgrad_slabs(n, hw, self.convs[key][0], 0, 8), 9 * self.convs[key][0] * 8 + 8
                slab = T.buf(f"slab_{key}_{tag}", (nsl, cnt), torch.float32, self.dev)
                if isinstance(a, MixSrc):
                    _lib.call("cgs_bf16_hwgrad_pooled_mix", n // 2, _p(a.A), _p(a.B), _p(a.Z), _p(dp), cod, _p(slab), _s())
                else:
                    a_kind = 1 if a.dtype == torch.uint8 else (2 if a.dtype == torch.float32 else 0)
                    _lib.call("cgs_bf16_hwgrad_pooled", n, hw, self.convs[key][0], a_kind, _p(a), _p(dp), add, cod, _p(slab), _s())
                plan.add(slab, nsl, cnt, self.off[key][0])
                w = self._wview(key)[0]
                if i >= 1:
                    de = T.buf(f"de{i - 1}_{tag}", (n, hw, hw, d[i - 1]), torch.bfloat16, self.dev)
                    _lib.call("cgs_bf16_h5conv", _lib.H5_ENC1_BWD_DATA_POOLED if i == 1 else _lib.H5_ENC2_BWD_DATA_POOLED, n, _p(dp), add, _p(w), None,
                              _p(de), cod, _s())
                    dp = de
                elif mix_bwd is not None:       # the image gradient of both mixes + the mix backward in one pass: d(pre-sigmoid mask), no d mix tensor
                    A8, B8, l1s, l2s = mix_bwd
                    _lib.call("cgs_bf16_enc0_bwd_mix", n // 2, _p(dp), cod, _p(w), _p(A8), _p(B8), _p(T.Z), float(l1s), float(l2s), _p(T.dzpre), _s())
                elif want_dx is not None:
                    _lib.call("cgs_bf16_enc0_bwd_data_pooled", n, _p(dp), add, cod, _p(w), _p(want_dx), _s())
                continue
            dyf = T.buf(f"dyf{i}_{tag}", (n, hw, hw, co), torch.bfloat16, self.dev)
            _lib.call("cgs_bf16_pool_expand", n, hw // 2, co, _p(dp), _p(skips[i]) if skips is not None else None, _p(T.codes[i][lo:hi]), _p(dyf), _s())
            a = src if i == 0 else T.e[i - 1][lo:hi]
            self._wgrad(T, plan, key, tag, n, hw, a, None, 2, dyf)
            if i == 1 and H5CONV and self.h5:
                dp = T.buf(f"de0_{tag}", (n, 64, 64, d[0]), torch.bfloat16, self.dev)
                _lib.call("cgs_bf16_h5conv", _lib.H5_ENC1_BWD_DATA, n, _p(dyf), None, _p(self._wview(key)[0]), None, _p(dp), None, _s())
            elif i > 0:
                dp = self._dgrad(T, key, dyf, f"de{i - 1}_{tag}")
            elif want_dx is not None and H5CONV and self.h5:
                _lib.call("cgs_bf16_enc0_bwd_data", n, _p(dyf), _p(self._wview(key)[0]), _p(want_dx), _s())
            elif want_dx is not None:
                ca = 3
                _lib.call("cgs_genbf16_conv3x3_fwd_train", n, hw, co, 0, ca, 0, 1, _lib.ACT_NONE, 0.01, 0, 1, _p(dyf), None, _p(self.w16T[key]),
                          _p(self._zero_bias(ca)), _p(want_dx), None, _s())

    def _masker_backward(self, T, plan, A):
        n, d, nb = T.n, self.d, self.nb
        ea = [t[n:2 * n] for t in T.e]
        do = T.buf("do0", (n, 64, 64, d[0]), torch.bfloat16, self.dev)
        if H5CONV and HWGRAD and self.h5:        # fp32 dz read directly; LeakyReLU' and the 2x2 cell sums in the data gradients' epilogues
            self._wgrad(T, plan, "masker.2", "m", n, 128, T.hm, None, 2, None, dy_f32=T.dzpre)
            dh = T.buf("dhm", (n, 128, 128, 16), torch.bfloat16, self.dev)
            _lib.call("cgs_bf16_mask2_bwd_data", n, _p(T.dzpre), _p(T.hm), _p(self._wview("masker.2")[0]), _p(dh), _s())
            self._wgrad(T, plan, "masker.0", "m", n, 128, A, T.o[0], 2, dh)
            _lib.call("cgs_bf16_mask0_bwd_data", n, _p(dh), _p(self._wview("masker.0")[0]), _p(do), _s())
        else:
            dz4 = T.buf("dz4", (n, 128, 128, 4), torch.bfloat16, self.dev)
            _lib.call("cgs_bf16_convert", n * 16384, 1, 4, 0, _p(T.dzpre), _p(dz4), _s())
            self._wgrad(T, plan, "masker.2", "m", n, 128, T.hm, None, 2, dz4, dyc=4, dy_f32=T.dzpre)
            dh = self._dgrad(T, "masker.2", dz4, "dhm", dy_channels=4)
            _lib.call("cgs_bf16_lrelu_bwd", dh.numel(), _p(dh), _p(T.hm), 0.01, _s())
            self._wgrad(T, plan, "masker.0", "m", n, 128, A, T.o[0], 2, dh)
            dcat = self._dgrad(T, "masker.0", dh, "dcat_m0")
            _lib.call("cgs_bf16_cat_split", n, 128, 3, d[0], 2, _p(dcat), None, _p(do), 0, _s())
        skips = [None] * 5
        tail = TAIL and self.tail
        for i in ((0, 1) if tail else (0, 1, 2, 3)):
            key, hw = f"dec_model.{i}", 64 >> i
            self._wgrad(T, plan, key, "m", n, hw, ea[i], T.o[i + 1], 2, do)
            skips[i] = T.buf(f"dskip{i}", (n, hw, hw, d[i]), torch.bfloat16, self.dev)
            low_f32 = tail and i == 1        # the tail kernel below reads its d o (16x16) in fp32
            dlow = T.buf(f"do{i + 1}" + ("f" if low_f32 else ""), (n, hw // 2, hw // 2, d[i + 1]), torch.float32 if low_f32 else torch.bfloat16, self.dev)
            if (i == 0 or (i == 1 and low_f32)) and H5CONV and self.h5:      # the two halves of d cat(e, up(o)) straight from the whole-strip kernel
                w = self._wview(key)[0]
                _lib.call("cgs_bf16_h5conv", _lib.H5_DEC0_BWD_SKIP if i == 0 else _lib.H5_DEC1_BWD_SKIP, n, _p(do), None, _p(w), None, _p(skips[i]), None, _s())
                _lib.call("cgs_bf16_h5conv", _lib.H5_DEC0_BWD_LOW if i == 0 else _lib.H5_DEC1_BWD_LOW, n, _p(do), None, _p(w), None, _p(dlow), None, _s())
            else:
                dcat = self._dgrad(T, key, do, f"dcat_d{i}")
                _lib.call("cgs_bf16_cat_split", n, hw, d[i], d[i + 1], 2, _p(dcat), _p(skips[i]), _p(dlow), int(low_f32), _s())
            do = dlow
        if tail:        # dec_model.2 / .3 / .4 backward in one launch (fp32): skip gradients at e2 / e3 / e4 and d o5
            f32 = torch.float32
            nsl = self.lib.cgs_tail_dec_bwd_slabs(n)
            cnts = {"dec_model.4": 6912 + 16, "dec_model.3": 1728 + 8, "dec_model.2": 1152 + 8}
            sl = {k: T.buf(f"slab_t_{k}", (nsl, c), f32, self.dev) for k, c in cnts.items()}
            skips[2], skips[3], skips[4] = (T.buf("dE2f", (n, 16, 16, 8), f32, self.dev), T.buf("dE3f", (n, 8, 8, 8), f32, self.dev),
                                            T.buf("dE4f", (n, 4, 4, 16), f32, self.dev))
            do5 = T.buf("do5", (n, nb), f32, self.dev)
            tdw = self._tail_dec_w()
            _lib.call("cgs_tail_dec_bwd", n, C.byref(tdw), _p(T.t_e2[n:2 * n]), _p(T.t_e3[n:2 * n]), _p(T.t_e4[n:2 * n]), _p(T.t_o5[n:2 * n]),
                      _p(T.t_o4), _p(T.t_o3), _p(do), _p(skips[2]), _p(skips[3]), _p(skips[4]), _p(do5),
                      _p(sl["dec_model.4"]), _p(sl["dec_model.3"]), _p(sl["dec_model.2"]), _s())
            for k, c in cnts.items():
                plan.add(sl[k], nsl, c, self.off[k][0])
            return skips, do5          # dec_model.5's backward runs in the A pass's tail kernel (cgs_tail_enc_bwd, d_o4)
        self._wgrad(T, plan, "dec_model.4", "m", n, 4, ea[4], T.o5, 4, do)
        dcat = self._dgrad(T, "dec_model.4", do, "dcat_d4")
        skips[4] = T.buf("dskip4", (n, 4, 4, d[4]), torch.bfloat16, self.dev)
        do5 = T.buf("do5", (n, nb), torch.float32, self.dev)
        _lib.call("cgs_bf16_cat_split", n, 4, d[4], nb, 4, _p(dcat), _p(skips[4]), _p(do5), 1, _s())
        de5 = self._gemm_bwd(T, plan, "dec_model.5", "m", T.e5[n:2 * n], do5)
        return skips, de5

    def _phase2_body(self, T):
        n = T.n
        A, B = T.ab[n:], T.ab[:n]
        self._critic_forward(T, T.ab, 0, 2 * n, want_o5=True)
        self._masker_forward(T, A)
        virt = MIX_VIRTUAL and MIX_BWD_FUSED and TAIL and H5CONV and HWGRAD and POOL_FUSED and self.tail
        if not virt and T.mixed is None:
            T.mixed = torch.empty((2 * n, 128, 128, 3), device=self.dev)
        mixsrc = MixSrc(A, B, T.Z) if virt else T.mixed
        _lib.call("cgs_mix_fwd", n, 16384, _p(A), _p(B), _p(T.Z), 1, None if virt else _p(T.mixed), _p(T.zsum), _s())      # virt: only the sums of |Z|, Z^2
        self._critic_forward(T, mixsrc, 2 * n, 4 * n)
        nz = n * 16384
        _lib.call("cgs_phase2_losses", n, _p(T.pred), _p(T.y), _p(T.zsum), T.nzpart, self.lfak, self.L1, self.L2, 1 | 2, nz, _p(T.losses),
                  _p(T.dpred), _s())
        first = T.plan_a is None
        pa, pb = hg.SlabPlan(), hg.SlabPlan()
        sink = [] if (TAIL and self.tail) else None      # the head's weight gradients of the mix pass and the A pass: one launch at the end
        if MIX_BWD_FUSED and H5CONV and HWGRAD and POOL_FUSED and self.h5:
            self._critic_backward(T, pa, "mix", mixsrc, 2 * n, 4 * n, mix_bwd=(A, B, self.L1 / nz, self.L2 / nz), head_sink=sink)
        else:
            if T.dmixed is None:
                T.dmixed = torch.empty((2 * n, 128, 128, 3), device=self.dev)
            self._critic_backward(T, pa, "mix", mixsrc, 2 * n, 4 * n, want_dx=T.dmixed, head_sink=sink)
            _lib.call("cgs_mix_bwd", n, 16384, _p(A), _p(B), _p(T.Z), _p(T.dmixed), 1, self.L1 / nz, self.L2 / nz, _p(T.dzpre), _s())
        skips, de5 = self._masker_backward(T, pa, A)
        self._critic_backward(T, pb, "a", A, n, 2 * n, d_e5_add=de5, skips=skips, plan_pw=pa, head_sink=sink)
        if sink:
            self._head_wgrad(T, sink, pa, pa, "both")      # both passes summed in the slab rows: the overwriting plan takes them whole
        if first:
            T.plan_a, T.plan_b = pa.build(self.grad), pb.build(self.grad, accumulate=True)
        T.plan_a.run(None)
        T.plan_b.run(self.step_t)        # ticks the step counter
        if self.dp:
            parallel.allreduce_sum_(self.grad, self.pg)
        _lib.call("cgs_adam_flat", self.total, _p(self.flat), _p(self.grad), _p(self.m), _p(self.v), _p(self.step_t), self.lr, self.b1, self.b2,
                  self.eps, 1.0 / self.world, _s())
        if not self._packs_unused():
            self._pack_weights(transposed=True)

    def phase2_step(self, A_u8: Optional[torch.Tensor] = None, B_u8: Optional[torch.Tensor] = None, Y: Optional[torch.Tensor] = None,
                    use_graph: bool = True):
        """One optimiser step (loss arrangement of main.py:364-429, Dropout off) on n A-frames / n B-frames (uint8 NHWC [n,128,128,3])
        and targets Y [n]; omitted inputs reuse the resident batch.  Returns the device tensor losses[8] = (critic, replace, inject,
        l1, l2, total, 0, 0)."""
        if self._train is None:
            if self.d[0] != 8 or self.nb > 48 or self.mc > 16:
                raise _lib.CgsError("Hourglass128.phase2_step: the bf16 weight-gradient kernels take the chfak 1 channel counts (<= 16 output, <= 48 input "
                                    "channels per layer); chfak != 1 runs infer() only")
            if A_u8 is None:
                raise _lib.CgsError("the first phase2_step needs the batch")
            self._train = Hourglass128._Train(self, A_u8.shape[0])
            self._pack_weights(transposed=True)
            self._graph = None
        T = self._train
        n = T.n
        for t, dst in ((A_u8, T.ab[n:]), (B_u8, T.ab[:n])):
            if t is not None:
                if t.dtype != torch.uint8 or tuple(t.shape) != (n, 128, 128, 3):
                    raise _lib.CgsError(f"phase2_step reads uint8 frames [{n},128,128,3]")
                dst.copy_(t, non_blocking=True)
        if Y is not None:
            T.y.copy_(Y.to(torch.float32), non_blocking=True)
        if self._graph is None:
            self._phase2_body(T)                 # eager: allocations + reduction tables
            self._graph = "eager"
            capturable = not self.dp
            if use_graph and self.dp:
                # the request is agreed on collectively before the collective trial (a rank that skipped it while its peer entered it would hang)
                if not parallel.agree_all(bool(self.dp_graph), self.pg, self.dev):
                    self.dp_capture_note = ("not requested on every rank (dp_graph False here or on a peer; the default at world > 1, see "
                                            "parallel.resolve_dp_graph)")
                else:
                    capturable, self.dp_capture_note = parallel.collective_capturable(self.pg, self.dev)
            if use_graph and capturable:        # data parallel: kernels -> all-reduce -> Adam as ONE graph launch per step
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, **({"capture_error_mode": "thread_local"} if self.dp else {})):
                    self._phase2_body(T)
                self._graph = g
                self.dp_single_graph = self.dp
        elif self._graph == "eager":
            self._phase2_body(T)
        else:
            self._graph.replay()
        return T.losses

    @staticmethod
    def param_shapes(chfak: int = 1, neck: int = 32, masker_channels: int = 16):
        """(critic, masker) parameter tables [(key, OIHW / [out, in] shape)] of the six-stage 128x128 variant, reference-style key names:
        five 3x3 encoder stages (128 -> 4) + the 4x4 valid bottleneck convolution + the two Linear layers; six decoder layers (the 1x1
        pointwise layer is dec_model.5) + the two mask-head layers."""
        d, b = [8 * chfak, 8 * chfak, 8 * chfak, 8 * chfak, 16 * chfak], neck * chfak
        crit, cin = [], 3
        for key, co in zip(ENC_KEYS, d):
            crit += [(key + ".weight", (co, cin, 3, 3)), (key + ".bias", (co,))]
            cin = co
        crit += [("features.17.weight", (b, d[4], 4, 4)), ("features.17.bias", (b,)), ("crit.1.weight", (b, b)), ("crit.1.bias", (b,)),
                 ("crit.4.weight", (1, b)), ("crit.4.bias", (1,))]
        mask = []
        for i in range(5):                       # dec_model.i: cat(skip e_i, upsampled lower level) -> e_i channels, 64x64 ... 4x4
            mask += [(f"dec_model.{i}.weight", (d[i], d[i] + (d[i + 1] if i < 4 else b), 3, 3)), (f"dec_model.{i}.bias", (d[i],))]
        mask += [("dec_model.5.weight", (b, b, 1, 1)), ("dec_model.5.bias", (b,)),
                 ("masker.0.weight", (masker_channels, 3 + d[0], 3, 3)), ("masker.0.bias", (masker_channels,)),
                 ("masker.2.weight", (1, masker_channels, 3, 3)), ("masker.2.bias", (1,))]
        return crit, mask

    @staticmethod
    def seeded_state(seed: int, chfak: int = 1, neck: int = 32, masker_channels: int = 16):
        """(critic, masker) stand-in weights for benchmarks and smoke runs: U(-1/sqrt(fan_in), +1/sqrt(fan_in)) per layer from
        numpy RandomState(seed) / RandomState(seed + 1) (there is no checkpoint of this build-defined model)."""
        import math
        import numpy as np
        out = []
        for k, shapes in enumerate(Hourglass128.param_shapes(chfak, neck, masker_channels)):
            rs, sd, bound = np.random.RandomState(seed + k), {}, 1.0
            for key, shp in shapes:
                if key.endswith(".weight"):
                    bound = 1.0 / math.sqrt(float(np.prod(shp[1:])))
                sd[key] = torch.from_numpy(rs.uniform(-bound, bound, size=shp).astype(np.float32))
            out.append(sd)
        return out[0], out[1]

    @staticmethod
    def critic_cost(chfak: int = 1, neck: int = 32):
        """(elements, FLOPs) per image of the critic's forward pass alone (same model as model_cost)."""
        d, nb = [8 * chfak, 8 * chfak, 8 * chfak, 8 * chfak, 16 * chfak], neck * chfak
        el, mac, cin, hw = 0, 0, 3, 128
        for co in d:
            el += hw * hw * (cin + co); mac += hw * hw * 9 * cin * co
            cin, hw = co, hw // 2
        el += 16 * d[4] + nb + 2 * nb + nb + 1; mac += 16 * d[4] * nb + nb * nb + nb
        return el, 2 * mac

    # fp32 algorithmic traffic / FLOPs per image (layer-granular model of SURVEY.md section 8d, applied to this variant)
    @staticmethod
    def model_cost(chfak: int = 1, neck: int = 32, mc: int = 16):
        d, nb = [8 * chfak, 8 * chfak, 8 * chfak, 8 * chfak, 16 * chfak], neck * chfak
        el, mac, cin, hw = 0, 0, 3, 128
        for co in d:                                            # encoder: reads its input, writes the pre-pool output
            el += hw * hw * (cin + co); mac += hw * hw * 9 * cin * co
            cin, hw = co, hw // 2
        el += 16 * d[4] + nb + 2 * nb + nb + 1; mac += 16 * d[4] * nb + nb * nb + nb
        el += 2 * nb; mac += nb * nb                            # 1x1
        el += 16 * (d[4] + nb + d[4]); mac += 16 * 9 * (d[4] + nb) * d[4]
        hw = 8
        for i in (3, 2, 1, 0):
            el += hw * hw * (d[i] + d[i + 1] + d[i]); mac += hw * hw * 9 * (d[i] + d[i + 1]) * d[i]
            hw *= 2
        el += 128 * 128 * (3 + d[0] + mc) + 128 * 128 * (mc + 1); mac += 128 * 128 * 9 * ((3 + d[0]) * mc + mc)
        return el, 2 * mac
