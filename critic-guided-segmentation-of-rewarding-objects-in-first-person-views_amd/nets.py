"""Drop-in module API of the reference's ``nets.py`` for the two networks ``main.py`` instantiates:

    NewCritic(width, dims, bottleneck, colorchs, chfak, activation, pool, dropout).forward(X, collect=False)
    UnetDecoder(width, edims, ddims, bottleneck, masker_channels, colorchs, chfak, ...).forward(X, embeds)

(reference: nets.py:160-212 and nets.py:452-523).  Same constructor signatures, same call signatures and
return shapes, same ``state_dict`` keys / OIHW tensors (checkpoints interchange with the reference), and
they behave as ``nn.Module`` for ``.to() / .train() / .eval() / .parameters()`` and ``torch.optim``.

``named_parameters()`` yields the reference's 14 keyed Parameters per module (nets.py:170-194, 479-492): OIHW-shaped
strided views of ONE flat fp32 buffer (``.flat``, a plain tensor) in the kernels' layout (HWIO conv weights, k-major
linears); forward / backward are HIP kernels through ``libcgs_hip.so`` (``torch.autograd.Function`` shims) that read
the flat buffer and hand the per-layer gradients back as the same views of one flat gradient.  Activations are NHWC on the device; tensors handed back to the caller
are NCHW-shaped views with channels-last strides, so any torch op on them sees the reference's values.
There is no CPU fallback: calling a module whose parameters are not on a HIP device raises.
"""
import math
import os
from collections import OrderedDict

import torch
import torch.nn as nn

from . import _lib
from . import generic as gen
from . import hourglass as hg
from .spec import critic_layout, masker_layout, view_as_reference


def _init_like_reference(layout, flat):
    """PyTorch default init of Conv2d/Linear (kaiming_uniform(a=sqrt 5) weights, U(+-1/sqrt(fan_in)) biases)
    drawn in the reference's parameter creation order, so ``torch.manual_seed(s)`` before construction gives
    the same weights as constructing the reference module (same torch version)."""
    sd = OrderedDict()
    bound = 0.0
    for key, seg in layout.segs.items():
        t = torch.empty(seg.ref_shape)
        if key.endswith(".weight"):
            nn.init.kaiming_uniform_(t, a=math.sqrt(5))
            fan_in = t[0].numel()
            bound = 1.0 / math.sqrt(fan_in) if fan_in > 0 else 0.0
        else:
            nn.init.uniform_(t, -bound, bound)
        sd[key] = t
    layout.flatten(sd, flat)


def _to_nhwc(X: torch.Tensor) -> torch.Tensor:
    """NCHW (any strides / float dtype) -> NHWC fp32 contiguous.  Layout plumbing only."""
    if X.dim() != 4 or X.shape[1] != 3 or X.shape[2] != 64 or X.shape[3] != 64:
        raise _lib.CgsError(f"expected an [N,3,64,64] image batch, got {tuple(X.shape)}")
    return X.detach().to(torch.float32).permute(0, 2, 3, 1).contiguous()


class _Holder(nn.Module):
    """Name-space node that carries the reference's parameter names (``features.0.weight`` ...): a group (``features``, ``crit``,
    ``dec_model``, ``masker`` -- the reference's nn.Sequential attributes, nets.py:170-194, 479-492) or one layer of a group holding
    its ``weight`` / ``bias`` Parameters.  No arithmetic: the owning _HipModule runs the kernels on the flat buffer those Parameters
    alias, and it alone moves / converts them (``_apply`` here is a no-op so that .to() cannot split the aliases)."""

    def __init__(self, desc=""):
        super().__init__()
        self._desc = desc

    def extra_repr(self):
        return self._desc

    def __getitem__(self, idx):          # critic.features[3].weight, as on the reference's nn.Sequential
        return self._modules[str(int(idx))]

    def _apply(self, fn, recurse=True):
        return self


class _HipModule(nn.Module):
    _layout_fn = None

    def _setup(self, layout, dropout=0.0):
        self.layout = layout
        flat = torch.empty(layout.total)
        _init_like_reference(layout, flat)
        # The reference's per-layer parameters (14 keyed tensors per module; callers: main.py:178, 330-334) as nn.Parameters whose
        # storage IS the flat kernel-layout buffer: OIHW-shaped strided views of the HWIO segments (spec.view_as_reference).  The
        # flat buffer itself is a plain attribute -- not a parameter, not in state_dict().
        self._flat = flat
        pmap = OrderedDict()
        for key, seg in layout.segs.items():
            grp, idx, leaf = key.split(".")
            if grp not in self._modules:
                self.add_module(grp, _Holder())
            g = self._modules[grp]
            if idx not in g._modules:
                g.add_module(idx, _Holder())
            layer = g._modules[idx]
            layer.register_parameter(leaf, nn.Parameter(torch.empty(0)))
            if leaf == "weight":
                layer._desc = f"{seg.kind}, weight {tuple(seg.ref_shape)} aliasing flat[{seg.offset}:{seg.offset + seg.count}]"
            pmap[key] = layer._parameters[leaf]
        self.__dict__["_pmap"] = pmap               # key -> Parameter (plain dict: the holders own the registration)
        self.__dict__["_pstride"] = {}
        self._rebind(flat)
        self.dropout_p = float(dropout)
        self.register_buffer("_step", torch.zeros(1, dtype=torch.int64), persistent=False)
        self._seed = int.from_bytes(os.urandom(4), "little")   # Philox key; does not touch torch's RNG stream

    # ---- the flat buffer and its per-layer aliases ----
    def _rebind(self, flat):
        """Makes ``flat`` the module's storage: every Parameter's .data becomes a view of it (Parameter objects keep their identity, so
        optimisers built earlier stay valid)."""
        self._flat = flat
        for key, v in self.layout.views(flat).items():
            self._pmap[key].data = v
            self._pstride[key] = v.stride()

    def _sync(self):
        """The Parameters are the truth, the flat buffer is what the kernels read.  They are the same memory unless someone replaced a
        Parameter's storage (``p.data = t``, copy.deepcopy, pickling): such a Parameter is copied into its segment and re-aliased.
        Returns the flat buffer."""
        flat = self._flat
        base = flat.data_ptr()
        for key, seg in self.layout.segs.items():
            p = self._pmap[key]
            if p.device != flat.device or p.dtype != flat.dtype or p.data_ptr() != base + 4 * seg.offset or p.stride() != self._pstride[key]:
                with torch.no_grad():
                    v = view_as_reference(flat[seg.offset:seg.offset + seg.count], seg)
                    v.copy_(p.detach().to(device=flat.device, dtype=flat.dtype))
                    p.data = v
        return flat

    @property
    def flat(self):
        """The flat fp32 kernel-layout buffer all 14 Parameters alias (a plain tensor: not a Parameter, carries no .grad)."""
        return self._sync()

    def _rehome(self, buf):
        """Moves the parameters into ``buf`` (a segment of an engine's buffer): values copied once, Parameters re-aliased, so module and
        engine see the same weights from then on (engine.adopt)."""
        with torch.no_grad():
            buf.copy_(self._sync().to(buf.device))
        self._rebind(buf)

    def _apply(self, fn, recurse=True):
        # .to() / .cuda() / .cpu() / .float(): the flat buffer moves as ONE tensor and the Parameters are re-aliased (the default
        # per-parameter conversion would give each its own storage)
        old = self._sync()
        super()._apply(fn)                              # own buffers (_step); the holders' _apply is a no-op
        new = fn(old)
        if new.dtype != torch.float32:
            raise _lib.CgsError(f"{type(self).__name__}: the HIP kernels of the module API hold fp32 parameters (got {new.dtype}); the "
                                "fp16 / bf16 paths convert their own weight copies (engine.infer(fp16=True))")
        grads = [(p, p.grad) for p in self._pmap.values() if p.grad is not None]
        self._rebind(new)
        for p, g in grads:
            with torch.no_grad():
                p.grad = fn(g)
        return self

    def _param_list(self):
        return list(self._pmap.values())

    def _grad_views(self, g, needs):
        """Per-layer gradients as reference-shaped views of the flat gradient ``g`` (same strides as the Parameters: autograd takes them
        as .grad without a copy), None where a Parameter is frozen (requires_grad_(False) on one key freezes exactly that layer)."""
        return tuple(v if need else None for v, need in zip(self.layout.views(g).values(), needs))

    # ---- checkpoint contract: reference keys, reference (OIHW) tensors ----
    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        out = destination if destination is not None else OrderedDict()
        for k, v in self.layout.unflatten(self._sync()).items():
            out[prefix + k] = v
        return out

    def load_state_dict(self, state_dict, strict=True, assign=False):
        sd = {k: v for k, v in state_dict.items()}
        flat = self._sync()
        with torch.no_grad():
            self.layout.flatten({k: v.to(flat.device) for k, v in sd.items()}, flat)
        return torch.nn.modules.module._IncompatibleKeys([], [])

    def _need_device(self):
        if not self._flat.is_cuda:
            raise _lib.CgsError(f"{type(self).__name__}: parameters are on {self._flat.device}; the HIP kernels need "
                                "a GPU (call .to('cuda')).  There is no CPU fallback on this path.")

    def _drop_state(self):
        if self.training and self.dropout_p > 0.0:
            step = self._step.clone()
            self._step += 1
            return hg.DropState(self.dropout_p, self._seed, step)
        return hg.NO_DROP


def _unsupported(what):
    raise NotImplementedError(f"{what} is not implemented by the HIP kernels of this build (width=64, dims=[8,8,8,16] x chfak, "
                              "ReLU, max-pool, nearest upsampling are)")


class _GenCriticFn(torch.autograd.Function):
    """NewCritic at chfak != 1 / neck != 32 on the shape-generic kernels (generic.py): same contract as _CriticFn."""

    @staticmethod
    def forward(ctx, X, mod, collect, *params):
        x = _to_nhwc(X)
        n = x.shape[0]
        flat = mod._flat                   # the storage ``params`` alias (NewCritic.forward has just _sync()ed them)
        drop = mod._drop_state()
        out = gen.critic_forward(flat, mod.layout, x, mod.chfak, mod.neck, drop=drop if drop.p > 0.0 else None)
        ctx.mod, ctx.drop, ctx.n, ctx.collect, ctx.x, ctx.out, ctx.flat = mod, drop, n, collect, x, out, flat
        ctx.save_for_backward(*params)     # version counters: an in-place update between forward and backward raises
        pred = out["pred"].view(n, 1)
        if not collect:
            return pred
        return (pred,) + tuple(out[f"e{i}"].permute(0, 3, 1, 2) for i in range(4)) + (out["e4"].view(n, -1, 1, 1),)

    @staticmethod
    def backward(ctx, dpred, *de):
        _ = ctx.saved_tensors
        flat = ctx.flat
        mod, n, dev = ctx.mod, ctx.n, flat.device
        lay = mod.layout
        dp = torch.zeros(n, device=dev) if dpred is None else dpred.reshape(n).to(torch.float32).contiguous()
        d_embeds = None
        if ctx.collect and any(d is not None for d in de):
            d_embeds = [None if d is None else d.to(torch.float32).permute(0, 2, 3, 1).contiguous() for d in de[:4]]
            d_embeds.append(None if de[4] is None else de[4].to(torch.float32).reshape(n, -1).contiguous())
        training = ctx.drop.p > 0.0
        g = gen.critic_grad_buffers(n, mod.chfak, mod.neck, dev)
        dx = torch.empty((n, 64, 64, 3), device=dev) if ctx.needs_input_grad[0] else None
        ws = gen.Workspace()
        gen.critic_backward_data(flat.detach(), lay, mod.chfak, mod.neck, ctx.out, g, dp, ws, ctx.drop if training else None,
                                 d_embeds=d_embeds, dx=dx)
        grad = torch.zeros(lay.total, device=dev)
        plan = hg.SlabPlan()
        gen.critic_backward_weights(grad, 0, lay, mod.chfak, mod.neck, ctx.out, g, ctx.x, n, plan, ws, "mod", training=training)
        plan.build(grad).run()
        return ((dx.permute(0, 3, 1, 2) if dx is not None else None), None, None) + mod._grad_views(grad, ctx.needs_input_grad[3:])


class _GenMaskerFn(torch.autograd.Function):
    """UnetDecoder at chfak != 1 / neck != 32 on the shape-generic kernels: same contract as _MaskerFn."""

    @staticmethod
    def forward(ctx, X, e0, e1, e2, e3, e4, mod, *params):
        x = _to_nhwc(X)
        n = x.shape[0]
        flat = mod._flat
        embeds = [e.detach().to(torch.float32).permute(0, 2, 3, 1).contiguous() for e in (e0, e1, e2, e3)]
        embeds.append(e4.detach().to(torch.float32).reshape(n, -1).contiguous())
        m = gen.masker_forward(flat, mod.layout, x, embeds, mod.chfak, mod.neck, mod.masker_channels)
        ctx.mod, ctx.n, ctx.x, ctx.embeds, ctx.m, ctx.flat = mod, n, x, embeds, m, flat
        ctx.save_for_backward(*params)
        return m["Z"].view(n, 1, 64, 64)

    @staticmethod
    def backward(ctx, dZ):
        _ = ctx.saved_tensors
        flat = ctx.flat
        mod, n = ctx.mod, ctx.n
        lay = mod.layout
        Z = ctx.m["Z"]
        dzpre = (dZ.reshape(n, 64, 64).to(torch.float32) * Z * (1.0 - Z)).contiguous()   # sigmoid'
        grad = torch.zeros(lay.total, device=flat.device)
        plan = hg.SlabPlan()
        d_emb = gen.masker_backward(flat.detach(), lay, grad, 0, ctx.x, ctx.embeds, ctx.m, dzpre, mod.chfak, mod.neck, plan,
                                    gen.Workspace(), mod.masker_channels)
        plan.build(grad).run()
        de = [d.permute(0, 3, 1, 2) for d in d_emb[:4]] + [d_emb[4].view(n, -1, 1, 1)]
        return (None,) + tuple(de) + (None,) + mod._grad_views(grad, ctx.needs_input_grad[7:])


class _CriticFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, mod, collect, *params):
        x = _to_nhwc(X)
        n = x.shape[0]
        flat = mod._flat
        drop = mod._drop_state()
        out = hg.critic_forward(flat, mod.layout, x, n, drop)
        ctx.mod, ctx.drop, ctx.n, ctx.collect, ctx.x, ctx.out, ctx.flat = mod, drop, n, collect, x, out, flat
        ctx.save_for_backward(*params)
        pred = out["pred"].view(n, 1)
        if not collect:
            return pred
        return (pred,) + tuple(out[f"e{i}"].permute(0, 3, 1, 2) for i in range(4)) + (out["e4"].view(n, 32, 1, 1),)

    @staticmethod
    def backward(ctx, dpred, *de):
        _ = ctx.saved_tensors
        flat = ctx.flat
        n, dev = ctx.n, flat.device
        lay = ctx.mod.layout
        dp = torch.zeros(n, device=dev) if dpred is None else dpred.reshape(n).to(torch.float32).contiguous()
        d_embeds = None
        if ctx.collect and any(d is not None for d in de):
            shapes = [(n, 32, 32, 8), (n, 16, 16, 8), (n, 8, 8, 8), (n, 4, 4, 16)]
            d_embeds = []
            for d, shp in zip(de[:4], shapes):
                d_embeds.append(torch.zeros(shp, device=dev) if d is None else d.to(torch.float32).permute(0, 2, 3, 1).contiguous())
            d_embeds.append(torch.zeros((n, 32), device=dev) if de[4] is None else de[4].to(torch.float32).reshape(n, 32).contiguous())
        dx = torch.empty((n, 64, 64, 3), device=dev) if ctx.needs_input_grad[0] else None
        plan = hg.SlabPlan()
        hg.critic_backward(flat.detach(), lay, ctx.x, n, ctx.out, dp, plan, ctx.drop, d_embeds=d_embeds,
                           n_add=n if d_embeds is not None else 0, dx=dx, dx_from=0)
        g = torch.empty(lay.total, device=dev)
        plan.build(g).run()
        return ((dx.permute(0, 3, 1, 2) if dx is not None else None), None, None) + ctx.mod._grad_views(g, ctx.needs_input_grad[3:])


class NewCritic(_HipModule):
    """Encoder + critic head (nets.py:160-212)."""

    def __init__(self, width=64, dims=[8, 8, 8, 16], bottleneck=32, colorchs=3, chfak=1, activation=nn.ReLU, pool="max",
                 dropout=0.5):
        super().__init__()
        if width != 64 or list(dims) != [8, 8, 8, 16] or colorchs != 3 or int(chfak) < 1 or int(bottleneck) % 4:
            _unsupported(f"NewCritic(width={width}, dims={dims}, bottleneck={bottleneck}, colorchs={colorchs}, chfak={chfak})")
        if activation is not nn.ReLU or pool != "max":
            _unsupported(f"NewCritic(activation={activation}, pool={pool})")
        self.width = width
        self.chfak, self.neck = int(chfak), int(bottleneck)
        self._generic = self.chfak != 1 or self.neck != 32
        self._setup(critic_layout(chfak, bottleneck, colorchs), dropout)

    def forward(self, X, collect=False):
        self._need_device()
        fn = _GenCriticFn if self._generic else _CriticFn      # chfak != 1 / neck != 32: the shape-generic kernels
        dev = self._sync().device
        out = fn.apply(X.to(dev), self, bool(collect), *self._param_list())
        if collect:
            return out[0], list(out[1:])
        return out


class _MaskerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, e0, e1, e2, e3, e4, mod, *params):
        x = _to_nhwc(X)
        n = x.shape[0]
        flat = mod._flat
        embeds = [e.detach().to(torch.float32).permute(0, 2, 3, 1).contiguous() for e in (e0, e1, e2, e3)]
        embeds.append(e4.detach().to(torch.float32).reshape(n, 32).contiguous())
        m = hg.masker_forward(flat, mod.layout, x, embeds, n)
        ctx.mod, ctx.n, ctx.x, ctx.embeds, ctx.m, ctx.flat = mod, n, x, embeds, m, flat
        ctx.save_for_backward(*params)
        return m["Z"].view(n, 1, 64, 64)

    @staticmethod
    def backward(ctx, dZ):
        _ = ctx.saved_tensors
        flat = ctx.flat
        n, lay = ctx.n, ctx.mod.layout
        Z = ctx.m["Z"]
        dzpre = (dZ.reshape(n, 64, 64).to(torch.float32) * Z * (1.0 - Z)).contiguous()   # sigmoid'
        plan = hg.SlabPlan()
        d_emb = hg.masker_backward(flat.detach(), lay, ctx.x, ctx.embeds, n, ctx.m, dzpre, plan)
        g = torch.empty(lay.total, device=flat.device)
        plan.build(g).run()
        de = [d.permute(0, 3, 1, 2) for d in d_emb[:4]] + [d_emb[4].view(n, 32, 1, 1)]
        return (None,) + tuple(de) + (None,) + ctx.mod._grad_views(g, ctx.needs_input_grad[7:])


class UnetDecoder(_HipModule):
    """Decoder + mask head (nets.py:452-523): nearest upsampling, linear trunk, LeakyReLU(0.01) + sigmoid head.
    The image ``X`` is treated as a constant (the reference never differentiates the mask w.r.t. it)."""

    def __init__(self, width=64, edims=[8, 8, 8, 16], ddims=[8, 8, 8, 16], bottleneck=32, masker_channels=16,
                 colorchs=3, chfak=1, activation=nn.ReLU, pool="max", upsample=True, pure=False):
        super().__init__()
        if (width != 64 or list(edims) != [8, 8, 8, 16] or list(ddims) != [8, 8, 8, 16] or int(bottleneck) % 4 or
                masker_channels != 16 or colorchs != 3 or int(chfak) < 1 or pool != "max" or not upsample or pure):
            _unsupported(f"UnetDecoder(width={width}, edims={edims}, ddims={ddims}, bottleneck={bottleneck}, "
                         f"masker_channels={masker_channels}, chfak={chfak}, pool={pool}, upsample={upsample}, pure={pure})")
        self.width = width
        self.masker_channels = masker_channels
        self.chfak, self.neck = int(chfak), int(bottleneck)
        self._generic = self.chfak != 1 or self.neck != 32
        self._setup(masker_layout(chfak, bottleneck, colorchs, masker_channels))

    def forward(self, X, embeds):
        self._need_device()
        dev = self._sync().device
        e = [t.to(dev) for t in embeds]
        fn = _GenMaskerFn if self._generic else _MaskerFn
        return fn.apply(X.to(dev), e[0], e[1], e[2], e[3], e[4], self, *self._param_list())


class _UnetFn(torch.autograd.Function):
    """nets.Unet(upsample=False) forward + backward on the shape-generic HIP kernels (csrc/gen.hip, gen_train.hip): Conv2d +
    LeakyReLU(0.2) + MaxPool2d encoder (cgs_gen_conv3x3_*), the 4x4 bottleneck convolution, the critic's Linear layers and the
    ConvTranspose2d(.,.,4,1,0) on the 1x1 map as GEMMs (cgs_gen_gemm_ex), ConvTranspose2d(4,2,1) over cat(decoder, pooled encoder)
    (cgs_gen_convt4s2_*).  No arithmetic of the path runs in torch ops; torch only re-lays-out the weights / gradients between the
    reference's parameter shapes and the kernels' (HWIO conv weights, [ky][kx][ci][co] transposed-conv weights)."""

    @staticmethod
    def forward(ctx, net, X, mode, *params):
        dev = params[0].device
        ew, eb = [params[2 * i] for i in range(5)], [params[2 * i + 1] for i in range(5)]
        dw, db = [params[10 + 2 * i] for i in range(5)], [params[11 + 2 * i] for i in range(5)]
        cw1, cb1, cw2, cb2 = params[20:24]
        x = _to_nhwc(X.to(dev))
        n = x.shape[0]
        k = lambda t: t.detach().permute(2, 3, 1, 0).contiguous().reshape(-1)        # OIHW -> HWIO
        ewk = [k(w) for w in ew]
        p, am, src = [], [], x
        for i in range(4):       # x_i = LeakyReLU(0.2)(conv); p_i = MaxPool2d(2)(x_i)   (nets.py:405-419)
            o, a = gen.conv3x3(src, None, ewk[i].data_ptr(), eb[i].detach().contiguous().data_ptr(), ew[i].shape[0], act="lrelu", slope=0.2,
                               pool=True, want_argmax=True)
            p.append(o); am.append(a); src = o
        nb, e3 = ew[4].shape[0], ew[3].shape[0]
        x4 = gen.gemm(p[3].reshape(n, -1), ewk[4].data_ptr(), eb[4].detach().contiguous().data_ptr(), 16 * e3, nb, act="lrelu", slope=0.2)
        ctx.net, ctx.mode, ctx.n = net, mode, n
        ctx.set_materialize_grads(False)
        if mode == "critic":
            w1, w2 = cw1.detach().t().contiguous(), cw2.detach().t().contiguous()
            h = gen.gemm(x4, w1.data_ptr(), cb1.detach().contiguous().data_ptr(), nb, 32, act="relu")
            c = gen.gemm(h, w2.data_ptr(), cb2.detach().contiguous().data_ptr(), 32, 1)
            torch.cuda.current_stream().synchronize()      # temporaries of this call (re-laid-out weights) die with it
            ctx.save_for_backward(x, *p, *am, x4, h, *[t.detach() for t in params])
            return c
        d3 = dw[4].shape[1]
        wk4 = dw[4].detach().permute(0, 2, 3, 1).contiguous().reshape(nb, 16 * d3)      # [ci][ky][kx][co]
        u = [None] * 4
        u[3] = gen.gemm(x4, wk4.data_ptr(), db[4].detach().repeat(16).contiguous().data_ptr(), nb, 16 * d3, act="lrelu",
                        slope=0.2).reshape(n, 4, 4, d3)
        for i in (3, 2, 1):      # u_{i-1} = LeakyReLU(0.2)(dec[i](cat(u_i, p_i)))   (nets.py:437-443)
            u[i - 1] = gen.convt_fwd(u[i], p[i], gen.convt_weight_to_kernel(dw[i].detach()), db[i].detach().contiguous(), act="lrelu", slope=0.2)
        y = gen.convt_fwd(u[0], p[0], gen.convt_weight_to_kernel(dw[0].detach()), db[0].detach().contiguous(), act="sigmoid")
        torch.cuda.current_stream().synchronize()
        ctx.save_for_backward(x, *p, *am, x4, *u, y, *[t.detach() for t in params])
        return y.permute(0, 3, 1, 2), u[0].permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, *gouts):
        sv = ctx.saved_tensors
        n, mode = ctx.n, ctx.mode
        x, p, am, x4 = sv[0], list(sv[1:5]), list(sv[5:9]), sv[9]
        params = sv[-24:]
        ew, eb = [params[2 * i] for i in range(5)], [params[2 * i + 1] for i in range(5)]
        dw, db = [params[10 + 2 * i] for i in range(5)], [params[11 + 2 * i] for i in range(5)]
        cw1, cb1, cw2, cb2 = params[20:24]
        dev = x.device
        grads = [None] * 24
        nb, e3 = ew[4].shape[0], ew[3].shape[0]
        z = lambda *shape: torch.zeros(shape, device=dev, dtype=torch.float32)
        ones = torch.ones(max(16 * n, 64), device=dev, dtype=torch.float32)

        def colsum(t2d):          # [rows, cols] -> [cols] (column sums on the GEMM kernel: ones^T . t)
            rows, cols = t2d.shape
            out = z(cols)
            gen.gemm_ex(1, rows, cols, ones, 0, 1, t2d, cols, 1, out)
            return out

        def linear_bwd(xin, w_t, g):      # y = xin [n,k] . w_t [k,m]: returns (d xin, dW^T [k,m], db [m]); g = gradient at y [n,m]
            kk, m = w_t.shape
            dwt = z(kk, m)
            gen.gemm_ex(kk, n, m, xin, 1, kk, g, m, 1, dwt)                  # xin^T . g
            dxin = z(n, kk)
            gen.gemm_ex(n, m, kk, g, m, 1, w_t, 1, m, dxin)                  # g . w_t^T
            return dxin, dwt, colsum(g)

        dskip = [None] * 4        # gradients arriving at p_i from the decoder (NHWC)
        if mode == "critic":
            h = sv[10]
            gc = gouts[0].contiguous().to(torch.float32).reshape(n, 1).clone()
            w1t, w2t = cw1.t().contiguous(), cw2.t().contiguous()
            dh, dw2t, db2 = linear_bwd(h, w2t, gc)
            gen.grad_fix(dh, h, act="relu")
            dx4, dw1t, db1 = linear_bwd(x4, w1t, dh)
            grads[20], grads[21], grads[22], grads[23] = dw1t.t().contiguous(), db1, dw2t.t().contiguous(), db2
        else:
            u, y = list(sv[10:14]), sv[14]
            if gouts[0] is None:      # only the returned embed u0 feeds the loss (set_materialize_grads(False)): zero gradient at the mask
                gy = torch.zeros_like(y)
            else:
                gy = gouts[0].permute(0, 2, 3, 1).contiguous().to(torch.float32).clone()      # NHWC [n,64,64,1]
                gen.grad_fix(gy, y, act="sigmoid")
            g = gy
            for i in (0, 1, 2, 3):   # dec[i]: ConvTranspose2d(4,2,1) over cat(u_i, p_i)
                wk = gen.convt_weight_to_kernel(dw[i])
                du, dp_i, dwk, dbi = gen.convt_bwd(u[i], p[i], wk, g)
                ci, co = dw[i].shape[0], dw[i].shape[1]
                grads[10 + 2 * i], grads[11 + 2 * i] = gen.convt_weight_from_kernel(dwk, ci, co), dbi
                dskip[i] = dp_i
                if i == 0 and len(gouts) > 1 and gouts[1] is not None:      # gradient at the returned embed u0
                    gen.grad_fix(du, None, addend=gouts[1].permute(0, 2, 3, 1).contiguous().to(torch.float32))
                gen.grad_fix(du, u[i], act="lrelu", slope=0.2)
                g = du
            # dec[4]: ConvTranspose2d(bottleneck, d3, 4, 1, 0) on the 1x1 map = x4 [n,nb] . wk4 [nb, 16 d3]
            d3 = dw[4].shape[1]
            wk4 = dw[4].permute(0, 2, 3, 1).contiguous().reshape(nb, 16 * d3)
            dx4, dwk4, _ = linear_bwd(x4, wk4, g.reshape(n, 16 * d3))
            grads[18] = dwk4.reshape(nb, 4, 4, d3).permute(0, 3, 1, 2).contiguous()
            grads[19] = colsum(g.reshape(n * 16, d3))
        # ---- encoder: the 4x4 bottleneck convolution, then conv + LeakyReLU(0.2) + pool x 4 ----
        gen.grad_fix(dx4, x4, act="lrelu", slope=0.2)
        wk = ew[4].permute(2, 3, 1, 0).contiguous().reshape(16 * e3, nb)
        dp3, dwk, dbe = linear_bwd(p[3].reshape(n, 16 * e3), wk, dx4)
        grads[8], grads[9] = dwk.reshape(4, 4, e3, nb).permute(3, 2, 0, 1).contiguous(), dbe
        g = dp3.reshape(p[3].shape)
        lib = _lib.load()
        for i in (3, 2, 1, 0):
            if dskip[i] is not None:
                gen.grad_fix(g, None, addend=dskip[i])
            gen.grad_fix(g, p[i], act="lrelu", slope=0.2)          # pool(lrelu(.)) = lrelu(pool(.)): the pooled output carries the sign
            src = x if i == 0 else p[i - 1]
            hw, ca, co = src.shape[1], src.shape[3], ew[i].shape[0]
            nsl = lib.cgs_gen_conv3x3_bwd_weight_slabs(n, ca, 0, co)
            cnt = 9 * ca * co + co
            slab = z(nsl, cnt)
            _lib.call("cgs_gen_conv3x3_bwd_weight", n, hw, ca, 0, co, 0, 2, gen._p(src), None, gen._p(g), gen._p(am[i]), gen._p(slab), gen._s())
            flat = z(cnt)
            plan = hg.SlabPlan()
            plan.add(slab, nsl, cnt, 0)
            plan.build(flat).run(None)
            grads[2 * i] = flat[:9 * ca * co].reshape(3, 3, ca, co).permute(3, 2, 0, 1).contiguous()
            grads[2 * i + 1] = flat[9 * ca * co:].clone()
            if i > 0:
                wkf = ew[i].permute(2, 3, 1, 0).contiguous().reshape(-1)
                dsrc = z(*src.shape)
                gen._bwd_data(n, hw, co, ca, g, am[i], wkf.data_ptr(), dsrc)
                g = dsrc
        torch.cuda.current_stream().synchronize()          # temporaries (re-laid-out weights, slabs) die with this call
        return (None, None, None) + tuple(grads)


class Unet(nn.Module):
    """The legacy single-module hourglass (nets.py:356-449; `main.py` never builds it, TrainHandler.__init__old did): Conv2d +
    LeakyReLU(0.2) + MaxPool2d encoder, bottleneck 4x4 convolution, and -- with ``upsample=False``, the form implemented
    here -- a ConvTranspose2d decoder: ConvTranspose2d(bottleneck, 16, 4, 1, 0) on the 1x1 map, then three
    ConvTranspose2d(., ., 4, 2, 1) over cat(decoder, pooled encoder) with LeakyReLU(0.2), a last one into the sigmoid mask.
    Same constructor, parameter names (``enc_model.N`` / ``dec_model.N`` / ``critic.N``) and default initialisation order as the
    reference, so checkpoints and seeds interchange; the modules below only HOLD the parameters, the arithmetic -- forward AND
    backward (round 3: _UnetFn) -- runs on the shape-generic HIP kernels."""

    def __init__(self, width=64, edims=[8, 8, 8, 16], ddims=[8, 8, 8, 16], bottleneck=32, colorchs=3, chfak=1,
                 activation=nn.ReLU, pool="max", upsample=True, pure=False):
        super().__init__()
        if upsample or pure or pool != "max" or width != 64:
            _unsupported(f"Unet(upsample={upsample}, pure={pure}, pool={pool}, width={width}) -- the ConvTranspose2d form "
                         "(upsample=False, pure=False, max-pool) is the one built here")
        e = [int(v) * chfak for v in edims]
        d = [int(v) * chfak for v in ddims]
        self.width, self.upsample, self.pure = width, upsample, pure
        self.e, self.d, self.bottleneck = e, d, int(bottleneck)
        enc = [nn.Conv2d(colorchs, e[0], 3, 1, 1), nn.Conv2d(e[0], e[1], 3, 1, 1), nn.Conv2d(e[1], e[2], 3, 1, 1),
               nn.Conv2d(e[2], e[3], 3, 1, 1), nn.Conv2d(e[3], bottleneck, 4)]
        dec = [nn.ConvTranspose2d(e[0] + d[0], 1, 4, 2, 1), nn.ConvTranspose2d(e[1] + d[1], d[0], 4, 2, 1),
               nn.ConvTranspose2d(e[2] + d[2], d[1], 4, 2, 1), nn.ConvTranspose2d(e[3] + d[3], d[2], 4, 2, 1),
               nn.ConvTranspose2d(bottleneck, d[3], 4, 1, 0)]
        self.dec_model = nn.Sequential(*dec)
        self.enc_model = nn.Sequential(*enc)
        self.critic = nn.Sequential(nn.Flatten(), nn.Linear(bottleneck, 32), nn.ReLU(), nn.Linear(32, 1))

    @staticmethod
    def _k(t):       # conv weight OIHW -> HWIO flat / bias as is
        return t.detach().permute(2, 3, 1, 0).contiguous().reshape(-1)

    def forward(self, X, critic=False, embeds=False):
        w0 = self.enc_model[0].weight
        if not w0.is_cuda:
            raise _lib.CgsError("Unet: parameters are on the CPU; the HIP kernels need a GPU (call .to('cuda')), there is no CPU fallback")
        if torch.is_grad_enabled() and any(q.requires_grad for q in self.parameters()):
            if X.requires_grad:
                _unsupported("Unet: a gradient w.r.t. the input image (parameters' gradients only)")
            params = []
            for m in self.enc_model:
                params += [m.weight, m.bias]
            for m in self.dec_model:
                params += [m.weight, m.bias]
            params += [self.critic[1].weight, self.critic[1].bias, self.critic[3].weight, self.critic[3].bias]
            if critic:
                return _UnetFn.apply(self, X, "critic", *params)
            y, u0 = _UnetFn.apply(self, X, "mask", *params)
            return (y, u0) if embeds else y
        with torch.no_grad():
            x = _to_nhwc(X.to(w0.device))
            n = x.shape[0]
            p, src = [], x
            for i in range(4):       # x_i = LeakyReLU(0.2)(conv); p_i = MaxPool2d(2)(x_i)   (nets.py:405-419)
                m = self.enc_model[i]
                wk, b = self._k(m.weight), m.bias.detach().contiguous()
                src = gen.conv3x3(src, None, wk.data_ptr(), b.data_ptr(), m.out_channels, act="lrelu", slope=0.2, pool=True)
                p.append(src)
            m = self.enc_model[4]
            wk, b = self._k(m.weight), m.bias.detach().contiguous()
            x4 = gen.gemm(p[3].reshape(n, -1), wk.data_ptr(), b.data_ptr(), 16 * self.e[3], self.bottleneck, act="lrelu", slope=0.2)
            if critic:
                l1, l2 = self.critic[1], self.critic[3]
                w1, w2 = l1.weight.detach().t().contiguous(), l2.weight.detach().t().contiguous()
                h = gen.gemm(x4, w1.data_ptr(), l1.bias.detach().data_ptr(), self.bottleneck, 32, act="relu")
                return gen.gemm(h, w2.data_ptr(), l2.bias.detach().data_ptr(), 32, 1)
            # dec[4]: ConvTranspose2d(bottleneck, d3, 4, 1, 0) on a 1x1 map = a GEMM into the 4x4 x d3 map (NHWC)
            m = self.dec_model[4]
            wk = m.weight.detach().permute(0, 2, 3, 1).contiguous().reshape(self.bottleneck, 16 * self.d[3])    # [ci][ky][kx][co]
            b = m.bias.detach().repeat(16).contiguous()
            u = gen.gemm(x4, wk.data_ptr(), b.data_ptr(), self.bottleneck, 16 * self.d[3], act="lrelu", slope=0.2).reshape(n, 4, 4, self.d[3])
            for i in (3, 2, 1):      # u_{i-1} = LeakyReLU(0.2)(dec[i](cat(u_i, p_i)))   (nets.py:437-443)
                m = self.dec_model[i]
                u = gen.convt_fwd(u, p[i], gen.convt_weight_to_kernel(m.weight.detach()), m.bias.detach().contiguous(), act="lrelu", slope=0.2)
            m = self.dec_model[0]
            y = gen.convt_fwd(u, p[0], gen.convt_weight_to_kernel(m.weight.detach()), m.bias.detach().contiguous(), act="sigmoid")
            torch.cuda.current_stream().synchronize()      # temporaries of this call (re-laid-out weights) die with it
            y = y.permute(0, 3, 1, 2)
            return (y, u.permute(0, 3, 1, 2)) if embeds else y


class _BatchNormActFn(torch.autograd.Function):
    """cgs_bn_act_fwd / cgs_bn_act_bwd (csrc/bn.hip) on an NHWC fp32 view; torch only permutes layouts."""

    @staticmethod
    def forward(ctx, mod, X, weight, bias):
        lib = _lib.load()
        x = X.detach().to(torch.float32).permute(0, 2, 3, 1).contiguous()
        n, h, w, c = x.shape
        pixels = n * h * w
        dev = x.device
        train = bool(mod.training or mod.running_mean is None)
        y, stats = torch.empty_like(x), torch.empty((c, 4), device=dev)
        ws = torch.empty(((3 * lib.cgs_bn_rows(pixels, c) + 2) * c,), device=dev)
        rm, rv = (mod.running_mean, mod.running_var) if (mod.running_mean is not None and (mod.training or not train)) else (None, None)
        _lib.call("cgs_bn_act_fwd", pixels, c, gen._p(x), gen._p(weight.detach()) if weight is not None else None,
                  gen._p(bias.detach()) if bias is not None else None, float(mod.eps), gen._ACT[mod.act], float(mod.slope), int(train), gen._p(y), gen._p(stats),
                  gen._p(ws), gen._p(rm), gen._p(rv), float(mod.momentum), gen._s())
        if mod.training and mod.num_batches_tracked is not None:
            mod.num_batches_tracked += 1
        ctx.save_for_backward(x, y, stats, ws)
        ctx.meta = (pixels, c, gen._ACT[mod.act], float(mod.slope), int(train), weight is not None, bias is not None)
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gY):
        x, y, stats, ws = ctx.saved_tensors
        pixels, c, act, slope, train, has_w, has_b = ctx.meta
        dy = gY.to(torch.float32).permute(0, 2, 3, 1).contiguous()
        dx, dg, db = torch.empty_like(x), torch.empty(c, device=x.device), torch.empty(c, device=x.device)
        _lib.call("cgs_bn_act_bwd", pixels, c, gen._p(x), gen._p(y), gen._p(dy), gen._p(stats), act, slope, train, gen._p(dx), gen._p(dg), gen._p(db),
                  gen._p(ws), gen._s())
        return None, dx.permute(0, 3, 1, 2), (dg if has_w else None), (db if has_b else None)


class BatchNormAct2d(nn.Module):
    """OPTIONAL BatchNorm2d + activation epilogue on the HIP kernels of csrc/bn.hip (SURVEY section 8 row f4 "optional BN epilogue"; the
    north_star's BatchNorm wording).  The reference has no BatchNorm anywhere, so no reference module maps to this one and nothing pins it:
    same constructor / buffers / train-eval semantics as torch.nn.BatchNorm2d (+ ``act`` in {"none", "relu", "lrelu"}, ``slope``), tested
    against it.  NCHW in / out like every module here; channels a multiple of 4, at most 64.  Under data parallelism the statistics stay
    per GPU (no collective), as the north_star prescribes."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True, act="none", slope=0.01):
        super().__init__()
        if num_features % 4 or not 4 <= num_features <= 64:
            _unsupported(f"BatchNormAct2d({num_features}): the kernels take 4 .. 64 channels in multiples of 4")
        if act not in ("none", "relu", "lrelu"):
            _unsupported(f"BatchNormAct2d(act={act!r})")
        if momentum is None:
            _unsupported("BatchNormAct2d(momentum=None) (cumulative moving average)")
        self.num_features, self.eps, self.momentum, self.act, self.slope = num_features, eps, momentum, act, slope
        self.weight = nn.Parameter(torch.ones(num_features)) if affine else None
        self.bias = nn.Parameter(torch.zeros(num_features)) if affine else None
        if track_running_stats:
            self.register_buffer("running_mean", torch.zeros(num_features))
            self.register_buffer("running_var", torch.ones(num_features))
            self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        else:
            self.running_mean = self.running_var = self.num_batches_tracked = None

    def forward(self, X):
        if not X.is_cuda:
            raise _lib.CgsError("BatchNormAct2d: the HIP kernels need a GPU tensor; there is no CPU fallback")
        if X.dim() != 4 or X.shape[1] != self.num_features:
            raise _lib.CgsError(f"BatchNormAct2d({self.num_features}): expected [N,{self.num_features},H,W], got {tuple(X.shape)}")
        if X.dtype != torch.float32:      # the kernels compute and return fp32: a silent dtype change downstream is worse than an error
            raise _lib.CgsError(f"BatchNormAct2d: fp32 input only (got {X.dtype}); cast explicitly")
        return _BatchNormActFn.apply(self, X, self.weight, self.bias)

