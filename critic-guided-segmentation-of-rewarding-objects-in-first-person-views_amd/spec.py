"""Static description of the Hourglass (NewCritic + UnetDecoder, nets.py:160-212, 452-523) as the HIP
kernels see it: per-layer shapes, the flat kernel-layout parameter buffer, and the conversion between
that buffer and the reference's ``state_dict`` (OIHW, reference key names -- the checkpoint contract)."""
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, List, Tuple

import torch


@dataclass(frozen=True)
class Seg:
    key: str          # reference state_dict key
    ref_shape: Tuple[int, ...]
    offset: int       # float offset in the module's flat buffer
    count: int
    kind: str         # conv_w | head_w4 | linear_w | vec_w | pw_w | bias


def _to_kernel(t: torch.Tensor, kind: str) -> torch.Tensor:
    """reference layout -> kernel layout (flattened)."""
    if kind in ("conv_w", "head_w4"):
        return t.permute(2, 3, 1, 0).reshape(-1)          # OIHW -> HWIO
    if kind == "linear_w":
        return t.t().reshape(-1)                           # [o][k] -> [k][o]
    if kind == "pw_w":
        return t[:, :, 0, 0].t().reshape(-1)               # [o][i][1][1] -> [i][o]
    return t.reshape(-1)                                   # vec_w, bias


def _from_kernel(v: torch.Tensor, seg: Seg) -> torch.Tensor:
    s = seg.ref_shape
    if seg.kind in ("conv_w", "head_w4"):
        o, i, kh, kw = s
        return v.reshape(kh, kw, i, o).permute(3, 2, 0, 1).contiguous()
    if seg.kind == "linear_w":
        o, k = s
        return v.reshape(k, o).t().contiguous()
    if seg.kind == "pw_w":
        o, i = s[0], s[1]
        return v.reshape(i, o).t().reshape(o, i, 1, 1).contiguous()
    return v.reshape(s).clone()


def view_as_reference(v: torch.Tensor, seg: Seg) -> torch.Tensor:
    """A reference-shaped (OIHW / [o][k]) strided VIEW of a segment of the flat kernel-layout buffer -- no copy: writes through
    either side are seen by the other.  The per-layer ``nn.Parameter``s of nets.NewCritic / nets.UnetDecoder (and their ``.grad``s
    over the flat gradient) are such views, so ``named_parameters()`` has the reference's keys and shapes (nets.py:170-194, 479-492)
    while the kernels keep their one contiguous buffer."""
    s = seg.ref_shape
    if seg.kind in ("conv_w", "head_w4"):
        o, i, kh, kw = s
        return v.view(kh, kw, i, o).permute(3, 2, 0, 1)
    if seg.kind == "linear_w":
        o, k = s
        return v.view(k, o).t()
    if seg.kind == "pw_w":
        o, i = s[0], s[1]
        return v.view(i, o).t().unsqueeze(-1).unsqueeze(-1)
    return v.view(s)


class Layout:
    """Flat parameter layout of one module."""

    def __init__(self, entries: List[Tuple[str, Tuple[int, ...], str]]):
        self.segs: "OrderedDict[str, Seg]" = OrderedDict()
        off = 0
        for key, shape, kind in entries:
            cnt = 1
            for d in shape:
                cnt *= d
            self.segs[key] = Seg(key, tuple(shape), off, cnt, kind)
            off += cnt
        self.total = off

    def off(self, key: str) -> int:
        return self.segs[key].offset

    def flatten(self, sd: Dict[str, torch.Tensor], out: torch.Tensor) -> None:
        """Writes a reference-format state_dict into the flat kernel-layout buffer ``out``."""
        missing = [k for k in self.segs if k not in sd]
        unexpected = [k for k in sd if k not in self.segs]
        if missing or unexpected:
            raise RuntimeError(f"state_dict mismatch: missing {missing}, unexpected {unexpected}")
        with torch.no_grad():
            for key, seg in self.segs.items():
                t = sd[key]
                if tuple(t.shape) != seg.ref_shape:
                    raise RuntimeError(f"size mismatch for {key}: checkpoint {tuple(t.shape)} vs model {seg.ref_shape}")
                out[seg.offset:seg.offset + seg.count].copy_(_to_kernel(t.detach().to(torch.float32), seg.kind))

    def views(self, flat: torch.Tensor) -> "OrderedDict[str, torch.Tensor]":
        """Reference-shaped views (no copies) of every segment of ``flat``: see view_as_reference."""
        return OrderedDict((key, view_as_reference(flat[seg.offset:seg.offset + seg.count], seg)) for key, seg in self.segs.items())

    def unflatten(self, flat: torch.Tensor) -> "OrderedDict[str, torch.Tensor]":
        """Flat kernel-layout buffer -> reference-format tensors (OIHW etc.), same device."""
        out = OrderedDict()
        for key, seg in self.segs.items():
            out[key] = _from_kernel(flat.detach()[seg.offset:seg.offset + seg.count], seg)
        return out


def critic_layout(chfak=1, neck=32, colorchs=3) -> Layout:
    d = [8 * chfak, 8 * chfak, 8 * chfak, 16 * chfak]
    b = neck * chfak
    ent, cin = [], colorchs
    for key, cout in zip(("features.0", "features.3", "features.6", "features.10"), d):
        ent += [(key + ".weight", (cout, cin, 3, 3), "conv_w"), (key + ".bias", (cout,), "bias")]
        cin = cout
    ent += [("features.14.weight", (b, d[3], 4, 4), "head_w4"), ("features.14.bias", (b,), "bias"),
            ("crit.1.weight", (b, b), "linear_w"), ("crit.1.bias", (b,), "bias"),
            ("crit.4.weight", (1, b), "vec_w"), ("crit.4.bias", (1,), "bias")]
    return Layout(ent)


def masker_layout(chfak=1, neck=32, colorchs=3, masker_channels=16) -> Layout:
    e = [8 * chfak, 8 * chfak, 8 * chfak, 16 * chfak]
    d = list(e)
    b = neck * chfak
    ent = [
        ("dec_model.0.weight", (d[0], e[0] + d[1], 3, 3), "conv_w"), ("dec_model.0.bias", (d[0],), "bias"),
        ("dec_model.1.weight", (d[1], e[1] + d[2], 3, 3), "conv_w"), ("dec_model.1.bias", (d[1],), "bias"),
        ("dec_model.2.weight", (d[2], e[2] + d[3], 3, 3), "conv_w"), ("dec_model.2.bias", (d[2],), "bias"),
        ("dec_model.3.weight", (d[3], e[3] + b, 3, 3), "conv_w"), ("dec_model.3.bias", (d[3],), "bias"),
        ("dec_model.4.weight", (b, b, 1, 1), "pw_w"), ("dec_model.4.bias", (b,), "bias"),
        ("masker.0.weight", (masker_channels, colorchs + d[0], 3, 3), "conv_w"), ("masker.0.bias", (masker_channels,), "bias"),
        ("masker.2.weight", (1, masker_channels, 3, 3), "conv_w"), ("masker.2.bias", (1,), "bias"),
    ]
    return Layout(ent)


# (key prefix, hw, ca, cb, co, ups, act, pool, dropout site on source A or None)
ENC_LAYERS = (
    ("features.0", 64, 3, 0, 8, 2, "relu", 1, None),
    ("features.3", 32, 8, 0, 8, 2, "relu", 1, None),
    ("features.6", 16, 8, 0, 8, 2, "relu", 1, None),
    ("features.10", 8, 8, 0, 16, 2, "relu", 1, 0),
)
DEC_LAYERS = (  # in execution order after the 1x1 bottleneck conv
    ("dec_model.3", 4, 16, 32, 16, 4, "none", 0, None),
    ("dec_model.2", 8, 8, 16, 8, 2, "none", 0, None),
    ("dec_model.1", 16, 8, 8, 8, 2, "none", 0, None),
    ("dec_model.0", 32, 8, 8, 8, 2, "none", 0, None),
    ("masker.0", 64, 3, 8, 16, 2, "lrelu", 0, None),
    ("masker.2", 64, 16, 0, 1, 2, "sigmoid", 0, None),
)
DROP_SITE_E2, DROP_SITE_E3, DROP_SITE_H1 = 0, 1, 2
