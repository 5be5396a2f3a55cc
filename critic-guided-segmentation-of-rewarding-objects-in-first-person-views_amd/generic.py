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

_ACT = {"none": _lib.ACT_NONE, "relu": _lib.ACT_RELU, "lrelu": _lib.ACT_LRELU, "sigmoid": _lib.ACT_SIGMOID}


def _s():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def conv3x3(a: torch.Tensor, b: Optional[torch.Tensor], w_ptr: int, bias_ptr: int, co: int, act: str = "none",
            slope: float = 0.01, pool: bool = False, ups: int = 2, want_argmax: bool = False):
    """act(conv3x3(cat(a, up_ups(b))) + bias) (+ MaxPool2d(2)).  a: NHWC uint8 or fp32 [n,hw,hw,ca]; b: NHWC fp32 or None."""
    if not a.is_cuda or not a.is_contiguous() or a.dtype not in (torch.uint8, torch.float32):
        raise _lib.CgsError("generic conv: source A must be a contiguous uint8 / fp32 device tensor (no CPU fallback)")
    n, hw, ca = a.shape[0], a.shape[1], a.shape[3]
    cb = 0 if b is None else b.shape[-1]
    oh = hw // 2 if pool else hw
    out = torch.empty((n, oh, oh, co), device=a.device, dtype=torch.float32)
    am = torch.empty((n, oh, oh, co), device=a.device, dtype=torch.uint8) if (pool and want_argmax) else None
    _lib.call("cgs_gen_conv3x3_fwd", n, hw, ca, cb, co, int(a.dtype == torch.uint8), ups, _ACT[act], float(slope), int(pool), _p(a),
              _p(b), C.c_void_p(w_ptr), C.c_void_p(bias_ptr), _p(out), _p(am), _s())
    return (out, am) if want_argmax else out


def gemm(x: torch.Tensor, w_ptr: int, bias_ptr: int, k: int, n_out: int, act: str = "none", slope: float = 0.01) -> torch.Tensor:
    m = x.shape[0]
    out = torch.empty((m, n_out), device=x.device, dtype=torch.float32)
    _lib.call("cgs_gen_gemm", m, k, n_out, _ACT[act], float(slope), _p(x), C.c_void_p(w_ptr), C.c_void_p(bias_ptr), _p(out), _s())
    return out


def critic_forward(flat: torch.Tensor, lay: Layout, x: torch.Tensor, chfak: int, neck: int = 32) -> Dict[str, torch.Tensor]:
    """NewCritic.forward in eval mode (nets.py:197-212) for any chfak.  x: NHWC uint8 / fp32 [n,64,64,3].
    Returns e0..e3 (NHWC), e4 [n,neck*chfak], h1, pred [n]."""
    fp = flat.data_ptr()
    off = lambda k: fp + 4 * lay.off(k)
    dims = [8 * chfak, 8 * chfak, 8 * chfak, 16 * chfak]
    nb = neck * chfak
    o, src = {}, x
    for i, (key, co) in enumerate(zip(("features.0", "features.3", "features.6", "features.10"), dims)):
        src = o[f"e{i}"] = conv3x3(src, None, off(key + ".weight"), off(key + ".bias"), co, act="relu", pool=True)
    n = x.shape[0]
    e3 = o["e3"].reshape(n, 16 * dims[3])
    o["e4"] = gemm(e3, off("features.14.weight"), off("features.14.bias"), 16 * dims[3], nb, act="relu")
    o["h1"] = gemm(o["e4"], off("crit.1.weight"), off("crit.1.bias"), nb, nb, act="relu")
    o["pred"] = gemm(o["h1"], off("crit.4.weight"), off("crit.4.bias"), nb, 1, act="sigmoid").reshape(n)
    return o


def masker_forward(flat: torch.Tensor, lay: Layout, x: torch.Tensor, embeds: List[torch.Tensor], chfak: int, neck: int = 32,
                   masker_channels: int = 16) -> Dict[str, torch.Tensor]:
    """UnetDecoder.forward (nets.py:494-523) for any chfak.  embeds = [e0..e3 NHWC, e4 [n,neck*chfak]]."""
    fp = flat.data_ptr()
    off = lambda k: fp + 4 * lay.off(k)
    d = [8 * chfak, 8 * chfak, 8 * chfak, 16 * chfak]
    nb = neck * chfak
    n = x.shape[0]
    o = {}
    o["o4"] = gemm(embeds[4], off("dec_model.4.weight"), off("dec_model.4.bias"), nb, nb)
    o["o3"] = conv3x3(embeds[3], o["o4"].reshape(n, 1, 1, nb), off("dec_model.3.weight"), off("dec_model.3.bias"), d[3], ups=4)
    o["o2"] = conv3x3(embeds[2], o["o3"], off("dec_model.2.weight"), off("dec_model.2.bias"), d[2])
    o["o1"] = conv3x3(embeds[1], o["o2"], off("dec_model.1.weight"), off("dec_model.1.bias"), d[1])
    o["o0"] = conv3x3(embeds[0], o["o1"], off("dec_model.0.weight"), off("dec_model.0.bias"), d[0])
    o["hm"] = conv3x3(x, o["o0"], off("masker.0.weight"), off("masker.0.bias"), masker_channels, act="lrelu", slope=0.01)
    o["Z"] = conv3x3(o["hm"], None, off("masker.2.weight"), off("masker.2.bias"), 1, act="sigmoid").reshape(n, 64, 64)
    return o


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
