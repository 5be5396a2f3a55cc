"""BASELINE config 5: "128x128x3 frames batch=256 (upscaled Hourglass) ... bf16 with MFMA 1x1 pointwise" -- a BUILD-DEFINED variant.

The reference cannot take 128x128 frames: its 4x4 valid convolution (/root/reference/nets.py:184) would see an 8x8 map and
Flatten -> Linear (nets.py:189-190) shape-errors.  This module defines the smallest change that keeps every
other layer's shape -- one extra Conv2d(8,8,3)+ReLU+MaxPool stage in front of the encoder, one extra Upsample+cat+Conv2d stage
behind the decoder -- and runs its eval-mode forward pass (critic value + mask, the -process path) on the bf16 form of the
shape-generic kernels (csrc/gen_f16.hip: bf16 activations and weights in HBM / LDS, fp32 accumulation on v_mfma_f32_16x16x16_bf16;
the 4x4 valid convolution, the Linear layers and the decoder's 1x1 pointwise convolution as MFMA GEMMs).
PARITY UNPINNED: there is no reference counterpart; tests compare against the build's own fp32 CPU restatement
(oracle/hourglass_ref.py, hourglass128_apply) with a stated bf16 tolerance.  Not wired into main.py (the reference's CLI has no
such size); `bench.py --config 5` measures it."""
import ctypes as C
from typing import Dict

import torch

from . import _lib
from .generic import _ACT, _p, _s

ENC_KEYS = ("features.0", "features.3", "features.6", "features.9", "features.13")


def _hwio(w: torch.Tensor) -> torch.Tensor:
    return w.permute(2, 3, 1, 0).contiguous().reshape(-1)          # OIHW -> HWIO (the kernels' weight order)


class Hourglass128:
    """Holds one parameter set (dicts of fp32 tensors with the key names of oracle.critic128_shapes / masker128_shapes) in the
    kernels' layouts: bf16 operand copies of the 3x3 layers, fp32 k-major matrices for the GEMM-shaped layers."""

    def __init__(self, critic_params: Dict[str, torch.Tensor], masker_params: Dict[str, torch.Tensor], device="cuda:0", chfak: int = 1,
                 neck: int = 32, masker_channels: int = 16):
        if not torch.cuda.is_available():
            raise _lib.CgsError("Hourglass128 needs an MI355X (HIP device); there is no CPU fallback")
        self.lib = _lib.load()
        self.dev = torch.device(device)
        d = [8 * chfak, 8 * chfak, 8 * chfak, 8 * chfak, 16 * chfak]
        self.d, self.nb, self.mc = d, neck * chfak, masker_channels
        dev = self.dev
        f = lambda t: t.detach().to(dev, torch.float32).contiguous()
        self.w16, self.bias, self.mat = {}, {}, {}

        def conv(key, P, ca, cb, co):
            w = _hwio(f(P[key + ".weight"]))
            nh = self.lib.cgs_gen16_packed_weight_halves(ca, cb, co)
            w16 = torch.empty(nh, device=dev, dtype=torch.bfloat16)
            _lib.call("cgs_genbf16_pack_weights", ca, cb, co, _p(w), _p(w16), _s())
            self.w16[key], self.bias[key] = w16, f(P[key + ".bias"])

        cin = 3
        for key, co in zip(ENC_KEYS, d):
            conv(key, critic_params, cin, 0, co)
            cin = co
        nb = self.nb
        # GEMM-shaped layers, k-major [k][n]: the 4x4 valid convolution (k = (y*4+x)*c), the Linear layers, the 1x1 convolution
        self.mat["features.17"] = f(critic_params["features.17.weight"]).permute(2, 3, 1, 0).contiguous().reshape(16 * d[4], nb)
        self.mat["crit.1"] = f(critic_params["crit.1.weight"]).t().contiguous()
        self.mat["crit.4"] = f(critic_params["crit.4.weight"]).t().contiguous()
        self.mat["dec_model.5"] = f(masker_params["dec_model.5.weight"]).reshape(nb, nb).t().contiguous()
        for k, P in (("features.17", critic_params), ("crit.1", critic_params), ("crit.4", critic_params), ("dec_model.5", masker_params)):
            self.bias[k] = f(P[k + ".bias"])
        conv("dec_model.4", masker_params, d[4], nb, d[4])
        for i in (3, 2, 1, 0):
            conv(f"dec_model.{i}", masker_params, d[i], d[i + 1], d[i])
        conv("masker.0", masker_params, 3, d[0], masker_channels)
        conv("masker.2", masker_params, masker_channels, 0, 1)
        torch.cuda.current_stream().synchronize()

    def _conv(self, key, a, b, co, act="none", pool=False, ups=2, out_f32=False):
        n, hw, ca = a.shape[0], a.shape[1], a.shape[3]
        cb = 0 if b is None else b.shape[-1]
        oh = hw // 2 if pool else hw
        out = torch.empty((n, oh, oh, co), device=a.device, dtype=torch.float32 if out_f32 else torch.bfloat16)
        _lib.call("cgs_genbf16_conv3x3_fwd", n, hw, ca, cb, co, int(a.dtype == torch.uint8), ups, _ACT[act], 0.01, int(pool), int(out_f32),
                  _p(a), _p(b), _p(self.w16[key]), _p(self.bias[key]), _p(out), _s())
        return out

    def _gemm(self, key, x, k, n_out, act="none", out_bf16=True):
        m = x.shape[0]
        out = torch.empty((m, n_out), device=x.device, dtype=torch.bfloat16 if out_bf16 else torch.float32)
        _lib.call("cgs_genbf16_gemm", m, k, n_out, _ACT[act], 0.01, int(x.dtype == torch.bfloat16), int(out_bf16), _p(x), _p(self.mat[key]),
                  _p(self.bias[key]), _p(out), _s())
        return out

    @torch.no_grad()
    def infer(self, x_u8: torch.Tensor):
        """x_u8: NHWC uint8 [n,128,128,3] on the device.  Returns (pred [n] fp32, Z [n,128,128] fp32)."""
        if x_u8.dtype != torch.uint8 or not x_u8.is_cuda or not x_u8.is_contiguous() or tuple(x_u8.shape[1:]) != (128, 128, 3):
            raise _lib.CgsError("Hourglass128.infer reads uint8 frames [n,128,128,3] (NHWC, contiguous, on the device)")
        n, d, nb = x_u8.shape[0], self.d, self.nb
        e, src = [], x_u8
        for key, co in zip(ENC_KEYS, d):
            src = self._conv(key, src, None, co, act="relu", pool=True)
            e.append(src)
        e5 = self._gemm("features.17", e[4].reshape(n, 16 * d[4]), 16 * d[4], nb, act="relu")
        h1 = self._gemm("crit.1", e5, nb, nb, act="relu")
        pred = self._gemm("crit.4", h1, nb, 1, act="sigmoid", out_bf16=False).reshape(n)
        o = self._gemm("dec_model.5", e5, nb, nb)                                  # the 1x1 pointwise convolution: an MFMA GEMM
        o = self._conv("dec_model.4", e[4], o.view(n, 1, 1, nb), d[4], ups=4)
        for i in (3, 2, 1, 0):
            o = self._conv(f"dec_model.{i}", e[i], o, d[i])
        hm = self._conv("masker.0", x_u8, o, self.mc, act="lrelu")
        Z = self._conv("masker.2", hm, None, 1, act="sigmoid", out_f32=True).reshape(n, 128, 128)
        return pred, Z

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
