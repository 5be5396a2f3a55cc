"""Training / inference engine for the model sizes outside the specialised kernel set (chfak != 1 or neck != 32; the paper's model is
chfak = 5, docs/index.html:151): the same step as engine.HourglassEngine -- Handler.critic_pipe (main.py:183-200) and
Handler.segmentation_training (main.py:344-463) -- composed from the shape-generic HIP kernels (csrc/gen.hip, csrc/gen_train.hip)
instead of the fused fixed-shape ones.  Same buffers-in, losses-out interface, same flat parameter / Adam state layout, same
HIP-graph replay and the same single gradient all-reduce for data parallelism.

Per phase-2 step:  critic fwd [B|A] -> masker fwd A -> mixes (materialised, fp32) -> critic fwd [rep|inj] -> losses
  -> critic data-gradient pass on [rep|inj] (down to the images) -> mix backward -> masker backward
  -> critic data-gradient pass on A (skip gradients added) -> ONE weight-gradient pass of the critic over [A|rep|inj]
  -> slab reduction [-> all-reduce] -> Adam."""
from typing import Dict, Optional

import torch

from . import _lib
from . import generic as gen
from . import hourglass as hg
from . import parallel
from .engine import HourglassEngine, _align4
from .spec import critic_layout, masker_layout

_P = hg._p
_S = hg._stream


class GenericEngine(HourglassEngine):
    def __init__(self, n: int, chfak: int = 5, neck: int = 32, device="cuda:0", dropout: float = 0.3, lfak: float = 5,
                 L1: float = 0.5, L2: float = 0.0, inject: bool = True, live: bool = True, threshrew: float = 0.0,
                 seed: int = 0x5EED, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, use_graph: bool = True,
                 process_group=None, share_with: "GenericEngine" = None, separate: bool = False, staticnorm: bool = True,
                 force_allreduce: bool = False, dp_graph: Optional[bool] = None):
        if not torch.cuda.is_available():
            raise _lib.CgsError("GenericEngine needs an MI355X (HIP device); there is no CPU fallback")
        _lib.load()
        self.n, self.dev = int(n), torch.device(device)
        self.chfak, self.neck = int(chfak), int(neck)
        self.p, self.lfak, self.L1, self.L2 = float(dropout), float(lfak), float(L1), float(L2)
        self.inject, self.live, self.bce = bool(inject), bool(live), bool(threshrew)
        self.lr, self.b1, self.b2, self.eps = lr, betas[0], betas[1], eps
        self.use_graph = use_graph
        self.pg = process_group
        self.world = torch.distributed.get_world_size(process_group) if process_group is not None else 1
        self.rank = torch.distributed.get_rank(process_group) if process_group is not None else 0
        self.dp = process_group is not None and (self.world > 1 or force_allreduce)
        self.dp_graph, self.dp_single_graph, self.dp_capture_note = parallel.resolve_dp_graph(dp_graph, self.world), False, None     # see HourglassEngine
        self.lc, self.lm = critic_layout(self.chfak, self.neck), masker_layout(self.chfak, self.neck)
        self.off_c, self.off_m = 0, _align4(self.lc.total)
        self.separate, self.staticnorm = bool(separate), bool(staticnorm)
        self.off_s = _align4(self.off_m + self.lm.total)
        self.total = self.off_s + self.lc.total if self.separate else self.off_m + self.lm.total
        z = lambda *s, dt=torch.float32: torch.zeros(s, device=self.dev, dtype=dt)
        if share_with is not None:
            self.flat, self.grad, self.m, self.v, self.step_t = (share_with.flat, share_with.grad, share_with.m,
                                                                 share_with.v, share_with.step_t)
        else:
            self.flat, self.grad, self.m, self.v = z(self.total), z(self.total), z(self.total), z(self.total)
            self.step_t = z(1, dt=torch.int64)
        self.fc, self.fm = self.flat[:self.lc.total], self.flat[self.off_m:self.off_m + self.lm.total]
        self.gc, self.gm = self.grad[:self.lc.total], self.grad[self.off_m:self.off_m + self.lm.total]
        if self.separate:
            self.fs, self.gs = self.flat[self.off_s:], self.grad[self.off_s:]
        self.drop = hg.DropState(self.p, (seed + 0x9E3779B97F4A7C15 * self.rank) & 0xFFFFFFFFFFFFFFFF, self.step_t)
        self.losses = z(8)
        self._allocated = False
        self.ws = gen.Workspace()
        self._infer_step = z(1, dt=torch.int64)
        self.fused_tail = False
        self._graphs: Dict[str, object] = {}
        self._forms: Dict[str, dict] = {}
        self._plans: Dict[str, hg.SlabPlan] = {}
        self._packplans: Dict[str, gen.PackPlan] = {}
        self._pver = share_with._pver if share_with is not None else [0]
        self._w16 = None

    # ---- helpers -----------------------------------------------------------------------------
    def _alloc(self):
        """Batch, activation and gradient buffers of the training step (first training call: an engine that only serves
        inference never holds them)."""
        if self._allocated:
            return
        n, n4 = self.n, 4 * self.n
        z = lambda *s, dt=torch.float32: torch.zeros(s, device=self.dev, dtype=dt)
        self.ab = z(2 * n, 64, 64, 3, dt=torch.uint8)         # [B | A]
        self.y = z(n)
        self.x3 = z(3 * n, 64, 64, 3)                         # [A / 255 | replaced | injected]: the critic's fp32 inputs with gradients
        self.mixed = self.x3[n:]
        self.dmixed = z(2 * n, 64, 64, 3)
        training = self.p > 0.0
        self.cbuf = gen.critic_buffers(n4, self.chfak, self.neck, self.dev, training)       # slots [B | A | rep | inj]
        self.gbuf = gen.critic_grad_buffers(n4, self.chfak, self.neck, self.dev)
        self.mbuf = gen.masker_buffers(n, self.chfak, self.neck, self.dev)
        self.sbuf: Dict[str, torch.Tensor] = {}
        if self.separate:
            self.sbuf = gen.critic_buffers(n, self.chfak, self.neck, self.dev, training)
            self.sgbuf = gen.critic_grad_buffers(n, self.chfak, self.neck, self.dev)
            self._zero_dpred = z(n)
        self.nzpart = _lib.load().cgs_mix_fwd_partials(n, 4096)
        self.zsum, self.dpred = z(2 * self.nzpart), z(n4)
        self.dzpre = z(n, 64, 64)
        self._allocated = True

    def phase2_step(self, A_u8=None, B_u8=None, Y=None):
        self._alloc()
        return super().phase2_step(A_u8, B_u8, Y)

    def phase1_step(self, X_u8=None, Y=None):
        self._alloc()
        return super().phase1_step(X_u8, Y)

    def gather_contrastive(self, *a, **kw):
        self._alloc()
        return super().gather_contrastive(*a, **kw)

    def _view(self, buf: Dict[str, torch.Tensor], a: int, b: int) -> Dict[str, torch.Tensor]:
        return {k: t[a:b] for k, t in buf.items()}

    def _fwd(self, flat, x, buf, drop):
        return gen.critic_forward(flat, self.lc, x, self.chfak, self.neck, out=buf, drop=drop if self.p > 0.0 else None)

    def _bwd_data(self, flat, s, g, dpred, drop, d_embeds=None, dx=None):
        gen.critic_backward_data(flat, self.lc, self.chfak, self.neck, s, g, dpred, self.ws, drop if self.p > 0.0 else None,
                                 d_embeds=d_embeds, dx=dx)

    def _finish(self, tag: str, plan: hg.SlabPlan):
        if tag not in self._plans:
            self._plans[tag] = plan.build(self.grad)
        self._plans[tag].run(self.step_t)

    # ---- phase 2 -----------------------------------------------------------------------------
    def _phase2_fwd_bwd(self):
        # (every 3x3 layer's weight operand of the step in one launch at its start: gen.PackPlan)
        with gen.pack_plan(self._packplans.setdefault("p2", gen.PackPlan())):
            self._phase2_body()

    def _phase2_body(self):
        n = self.n
        nmix = 2 * n if self.inject else n
        drop = self.drop
        A, B = self.ab[n:], self.ab[:n]
        self._fwd(self.fc, self.ab, self._view(self.cbuf, 0, 2 * n), drop.shifted(0))
        sa = self._view(self.cbuf, n, 2 * n)
        if self.separate:     # main.py:389-390: the masker's inputs come from the second critic's pass over A
            self._fwd(self.fs, A, self.sbuf, drop.shifted(4 * n))
            embeds = [self.sbuf[f"e{i}"] for i in range(5)]
        else:
            embeds = [sa[f"e{i}"] for i in range(5)]
        gen.masker_forward(self.fm, self.lm, A, embeds, self.chfak, self.neck, out=self.mbuf)
        Z = self.mbuf["Z"]
        _lib.call("cgs_mix_fwd", n, 4096, _P(A), _P(B), _P(Z), int(self.inject), _P(self.mixed), _P(self.zsum), _S())
        smix = self._view(self.cbuf, 2 * n, 2 * n + nmix)
        self._fwd(self.fc, self.mixed[:nmix], smix, drop.shifted(2 * n))
        flags = (1 if self.live else 0) | (2 if self.inject else 0) | (4 if self.bce else 0) | (0 if self.staticnorm else 8)
        _lib.call("cgs_phase2_losses", n, _P(self.cbuf["pred"]), _P(self.y), _P(self.zsum), self.nzpart, self.lfak, self.L1, self.L2,
                  flags, n * 4096, _P(self.losses), _P(self.dpred), _S())
        plan = hg.SlabPlan()
        # critic, data gradients of the mixes down to the images; then the mix backward
        self._bwd_data(self.fc, smix, self._view(self.gbuf, 2 * n, 2 * n + nmix), self.dpred[2 * n:2 * n + nmix], drop.shifted(2 * n),
                       dx=self.dmixed[:nmix])
        nz = float(n * 4096)
        # -staticnorm '' (main.py:415-418): the regulariser of A-image i weighted by 1 - pred[i].detach()
        _lib.call("cgs_mix_bwd_weighted", n, 4096, _P(A), _P(B), _P(Z), _P(self.dmixed), int(self.inject), self.L1 / nz, self.L2 / nz,
                  None if self.staticnorm else _P(self.cbuf["pred"][n:2 * n]), _P(self.dzpre), _S())
        d_emb = gen.masker_backward(self.fm, self.lm, self.grad, self.off_m, A, embeds, self.mbuf, self.dzpre, self.chfak, self.neck,
                                    plan, self.ws, need_embed_grads=self.live or self.separate)
        if self.separate:     # the skip gradients go into the SECOND critic; its own head sees no loss
            self._bwd_data(self.fs, self.sbuf, self.sgbuf, self._zero_dpred, drop.shifted(4 * n), d_embeds=d_emb)
            gen.critic_backward_weights(self.grad, self.off_s, self.lc, self.chfak, self.neck, self.sbuf, self.sgbuf, A, n, plan,
                                        self.ws, "sep", training=self.p > 0.0)
            d_emb = None
        if self.live:
            self._bwd_data(self.fc, sa, self._view(self.gbuf, n, 2 * n), self.dpred[n:2 * n], drop.shifted(n), d_embeds=d_emb)
            # ONE weight-gradient pass over the three passes with a loss: slots [A | rep | inj], inputs x3
            _lib.call("cgs_gen_u8_to_f32", A.numel(), _P(A), _P(self.x3), _S())
            gen.critic_backward_weights(self.grad, self.off_c, self.lc, self.chfak, self.neck, self._view(self.cbuf, n, 2 * n + nmix),
                                        self._view(self.gbuf, n, 2 * n + nmix), self.x3[:n + nmix], n + nmix, plan, self.ws, "crit",
                                        training=self.p > 0.0)
        self._finish("p2", plan)

    # ---- phase 1 -----------------------------------------------------------------------------
    def _phase1_fwd_bwd(self):
        with gen.pack_plan(self._packplans.setdefault("p1", gen.PackPlan())):
            self._phase1_body()

    def _phase1_body(self):
        n = self.n
        X = self.ab[:n]
        s, g = self._view(self.cbuf, 0, n), self._view(self.gbuf, 0, n)
        self._fwd(self.fc, X, s, self.drop.shifted(0))
        _lib.call("cgs_phase1_loss", n, _P(self.cbuf["pred"]), _P(self.y), int(self.bce), _P(self.losses), _P(self.dpred), _S())
        plan = hg.SlabPlan()
        self._bwd_data(self.fc, s, g, self.dpred[:n], self.drop.shifted(0))
        gen.critic_backward_weights(self.grad, self.off_c, self.lc, self.chfak, self.neck, s, g, X, n, plan, self.ws, "p1",
                                    training=self.p > 0.0)
        self._finish("p1", plan)

    # ---- saliency baseline (main.py:941-953) ------------------------------------------------------
    def saliency(self, X: torch.Tensor):
        hg._chk_img(X, 0, "saliency input")
        if X.dtype != torch.float32:
            raise _lib.CgsError("saliency needs the fp32 image batch (the gradient is taken w.r.t. it)")
        b = X.shape[0]
        X = X.contiguous()
        s = gen.critic_forward(self.fc, self.lc, X, self.chfak, self.neck)
        g = gen.critic_grad_buffers(b, self.chfak, self.neck, X.device)
        dx = torch.empty((b, 64, 64, 3), device=X.device, dtype=torch.float32)
        dpred = torch.full((b,), 1.0 / b, device=X.device, dtype=torch.float32)
        gen.critic_backward_data(self.fc, self.lc, self.chfak, self.neck, s, g, dpred, gen.Workspace(), dx=dx)
        torch.cuda.current_stream().synchronize()
        return s["pred"], dx

    # ---- inference (main.py:1130-1151) -----------------------------------------------------------
    @torch.no_grad()
    def infer(self, X: torch.Tensor, want_mask: bool = True, fp16_mask_head: bool = False, train_mode: bool = False, fp16: bool = False):
        hg._chk_img(X, 0, "infer input")
        if fp16_mask_head:
            raise NotImplementedError("the fp16-operand mask head is a chfak=1 kernel")
        if fp16:
            if train_mode:
                raise _lib.CgsError("fp16 inference is an eval-mode path (no Dropout)")
            return self._infer_f16(X, want_mask, self.chfak, self.neck)
        X = X.contiguous()
        drop = None
        if train_mode and self.p > 0.0:
            drop = hg.DropState(self.p, self.drop.seed ^ 0xD1CE, self._infer_step)
        c = gen.critic_forward(self.fc, self.lc, X, self.chfak, self.neck, drop=drop)
        Z = None
        if want_mask:
            src = gen.critic_forward(self.fs, self.lc, X, self.chfak, self.neck,
                                     drop=drop.shifted(X.shape[0]) if drop is not None else None) if self.separate else c
            Z = gen.masker_forward(self.fm, self.lm, X, [src[f"e{i}"] for i in range(5)], self.chfak, self.neck)["Z"]
        if drop is not None:
            self._infer_step += 1
        return c["pred"], Z
