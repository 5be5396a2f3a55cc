"""Fused training / inference engine: the loop bodies of Handler.critic_pipe (main.py:183-200) and
Handler.segmentation_training (main.py:344-463) as one static sequence of HIP kernels over preallocated
device buffers, captured into a HIP graph and replayed once per step.

Per phase-2 step (N = images in A = images in B):
  critic fwd on [B|A] (2N, uint8 loader)  ->  masker fwd on A  ->  mix  ->  critic fwd on [rep|inj] (2N)
  -> losses  ->  critic bwd on [rep|inj] (image gradient)  ->  mix bwd (+L1/L2)  ->  masker bwd
  -> critic bwd on A (skip gradients fused)  ->  slab reduction (+tick)  [-> RCCL all-reduce]  ->  flat Adam.
Data parallelism: one process per GPU, each with its own N images; the only exchange is one all-reduce
of the flat gradient buffer (25 661 floats) between the slab reduction and Adam.
"""
import os
import ctypes as C
from typing import Dict, Optional

import torch

from . import _lib
from . import hourglass as hg
from . import parallel
from .spec import ENC_LAYERS, critic_layout, masker_layout

_P = hg._p
_S = hg._stream


def _align4(x):
    return (x + 3) // 4 * 4


GATHER_ONE_LAUNCH = os.environ.get("CGS_GATHER_ONE_LAUNCH", "1") != "0"
PHASE1_FUSED_TAIL = os.environ.get("CGS_PHASE1_FUSED_TAIL", "1") != "0"      # (A/B switch: phase 1 with phase 2's fused step tail)
# config 4's layers between features.0 and dec_model.0: "fused1" = features.3 .. dec_model.1 in one launch per image on fp16 tiles (csrc/tail_infer.hip),
# "fused" = features.3 as a launch of its own (cgs_f16_enc1_fwd) + that kernel from features.6 on, "1" = ... + the two tail launches with fp16 operands
# (csrc/tail_h16.h), "0" = ... + the fp32 tail kernels (A/B switch)
# Measured at batch 2048 (profiles/r06_config4_fused_tail_ab.txt, r06_config4_enc1_in_tail_ab.txt): "0" 0.1819 ms, "1" 0.1685, "fused" 0.1562, "fused1" 0.1612 --
# features.3 is a throughput-bound convolution: inside the per-image latency chain (three workgroups per CU) it costs more than as a launch of its own.
F16_TAILS = os.environ.get("CGS_F16_TAILS", "fused")


class HourglassEngine:
    """Owns the parameters (critic | masker in one flat kernel-layout buffer), Adam state and every
    activation / gradient / slab buffer for a fixed batch size ``n``."""

    def __init__(self, n: int, device="cuda:0", dropout: float = 0.3, lfak: float = 5, L1: float = 0.5, L2: float = 0.0,
                 inject: bool = True, live: bool = True, threshrew: float = 0.0, seed: int = 0x5EED,
                 lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, use_graph: bool = True,
                 process_group=None, share_with: "HourglassEngine" = None, separate: bool = False, staticnorm: bool = True,
                 force_allreduce: bool = False, dp_graph: Optional[bool] = None):
        if not torch.cuda.is_available():
            raise _lib.CgsError("HourglassEngine needs an MI355X (HIP device); there is no CPU fallback")
        _lib.load()
        self.n, self.dev = int(n), torch.device(device)
        self.p, self.lfak, self.L1, self.L2 = float(dropout), float(lfak), float(L1), float(L2)
        self.inject, self.live, self.bce = bool(inject), bool(live), bool(threshrew)
        self.lr, self.b1, self.b2, self.eps = lr, betas[0], betas[1], eps
        self.use_graph = use_graph
        self.pg = process_group
        self.world = torch.distributed.get_world_size(process_group) if process_group is not None else 1
        self.rank = torch.distributed.get_rank(process_group) if process_group is not None else 0
        # data parallel: the all-reduce sits between the step graph and the Adam graph.  force_allreduce keeps that
        # form for a 1-rank group too (rehearsal of the RCCL path on a 1-GPU box).
        self.dp = process_group is not None and (self.world > 1 or force_allreduce)
        # dp_graph: record the all-reduce into the step's HIP graph when the backend allows it (RCCL, decided collectively); False keeps the
        # form step graph -> eager all-reduce -> Adam graph; None = parallel.resolve_dp_graph (graph on a 1-rank group, eager at world > 1)
        self.dp_graph, self.dp_single_graph, self.dp_capture_note = parallel.resolve_dp_graph(dp_graph, self.world), False, None
        self.lc, self.lm = critic_layout(), masker_layout()
        self.off_c, self.off_m = 0, _align4(self.lc.total)
        # -separate (main.py:110-111, 390): a second critic supplies the masker's skip inputs; its parameters sit behind the
        # masker's, so "masker + sepcrit" (the frozen optimiser group, main.py:334) is one contiguous range too
        self.separate = bool(separate)
        self.staticnorm = bool(staticnorm)     # False: main.py:415-418, the mask regulariser weighted per image by 1 - pred.detach()
        self.off_s = _align4(self.off_m + self.lm.total)
        self.total = self.off_s + self.lc.total if self.separate else self.off_m + self.lm.total
        z = lambda *s, dt=torch.float32: torch.zeros(s, device=self.dev, dtype=dt)
        if share_with is not None:   # another batch size over the SAME parameters / optimiser state
            self.flat, self.grad, self.m, self.v, self.step_t = (share_with.flat, share_with.grad, share_with.m,
                                                                 share_with.v, share_with.step_t)
        else:
            self.flat, self.grad, self.m, self.v = z(self.total), z(self.total), z(self.total), z(self.total)
            self.step_t = z(1, dt=torch.int64)
        self.fc, self.fm = self.flat[:self.lc.total], self.flat[self.off_m:self.off_m + self.lm.total]
        self.gc, self.gm = self.grad[:self.lc.total], self.grad[self.off_m:self.off_m + self.lm.total]
        if self.separate:
            self.fs, self.gs = self.flat[self.off_s:], self.grad[self.off_s:]
        # every rank draws its own dropout stream (the global batch then holds independent masks, as one process would)
        self.drop = hg.DropState(self.p, (seed + 0x9E3779B97F4A7C15 * self.rank) & 0xFFFFFFFFFFFFFFFF, self.step_t)
        n4 = 4 * n
        self.ab = z(2 * n, 64, 64, 3, dt=torch.uint8)       # [B | A]
        self.y = z(n)
        self.mixed = None if hg.ENC0_MIX_FUSED else z(2 * n, 64, 64, 3)      # [replaced | injected] (virtual on the fused path)
        self.cbuf: Dict[str, torch.Tensor] = {}              # critic activations for the 4N slots
        for i, (key, hw, ca, cb, co, *_r) in enumerate(ENC_LAYERS):
            self.cbuf[f"e{i}"] = z(n4, hw // 2, hw // 2, co)
            self.cbuf[f"am{i}"] = z(n4, hw // 2, hw // 2, co // 8, dt=torch.int32)
        self.cbuf["e4"], self.cbuf["h1"], self.cbuf["pred"] = z(n4, 32), z(n4, 32), z(n4)
        self.sbuf: Dict[str, torch.Tensor] = {}              # -separate: the second critic's activations on A
        if self.separate:
            for i, (key, hw, ca, cb, co, *_r) in enumerate(ENC_LAYERS):
                self.sbuf[f"e{i}"] = z(n, hw // 2, hw // 2, co)
                self.sbuf[f"am{i}"] = z(n, hw // 2, hw // 2, co // 8, dt=torch.int32)
            self.sbuf["e4"], self.sbuf["h1"], self.sbuf["pred"] = z(n, 32), z(n, 32), z(n)
            self._zero_dpred = z(n)
        self.mbuf: Dict[str, torch.Tensor] = {}
        # partial sums of |Z| and Z^2: from the mask layer's workgroups (4 per image) or from cgs_mix_fwd's
        self.nzpart = hg.zpart_count(n) if hg.ENC0_MIX_FUSED else _lib.load().cgs_mix_fwd_partials(n, 4096)
        self.zsum, self.losses, self.dpred = z(2 * self.nzpart), z(8), z(n4)
        self.dmixed = None if hg.ENC0_MIX_FUSED else z(2 * n, 64, 64, 3)
        self.dzpre = z(n, 64, 64)
        self._ws = {"mb": {}, "cb_mix": {}, "cb_a": {}, "cb_sep": {}, "p1": {}}
        self._infer_step = z(1, dt=torch.int64)              # -noevalmode: Dropout stream of inference batches
        self._ticket = z(4096, dt=torch.int32)               # cgs_reduce_adam's last-workgroup counters (1 + one per job row)
        # the loss gradients at pred are derived inside the tail backward kernels and the step ends in ONE launch: slab reduction
        # + loss values + Adam on one GPU; under data parallelism slab reduction + loss values, then the all-reduce, then ONE Adam
        # launch (round 3: the data-parallel step is the single-GPU step + one launch + the collective)
        self.fused_tail = hg.TAIL_BWD
        self._graphs: Dict[str, object] = {}
        self._forms: Dict[str, dict] = {}                    # data parallel: the captured launch forms of a step kind (set_dp_launch_form)
        # parameter version (shared by engines that share the parameters): the fp16 inference path repacks its weight copies when it moves
        self._pver = share_with._pver if share_with is not None else [0]
        self._w16 = None
        # (second streams inside the step graph -- weight gradients, critic(B) -- were measured neutral to slower on ROCm 7.2: the
        #  cross-queue joins cost what the overlap gains; the hooks stay as no-ops)
        self.side = hg.SideStream(None)
        self._plans: Dict[str, hg.SlabPlan] = {}

    # ---- parameters --------------------------------------------------------------------------
    def load_state(self, critic_sd=None, masker_sd=None, sepcrit_sd=None):
        if critic_sd is not None:
            self.lc.flatten({k: v.to(self.dev) for k, v in critic_sd.items()}, self.fc)
        if masker_sd is not None:
            self.lm.flatten({k: v.to(self.dev) for k, v in masker_sd.items()}, self.fm)
        if sepcrit_sd is not None:
            self.lc.flatten({k: v.to(self.dev) for k, v in sepcrit_sd.items()}, self.fs)
        parallel.broadcast_params_(self.flat, self.pg)    # replicas start identical
        self._pver[0] += 1

    def critic_state(self):
        return self.lc.unflatten(self.fc)

    def masker_state(self):
        return self.lm.unflatten(self.fm)

    def sepcrit_state(self):
        return self.lc.unflatten(self.fs)

    def adopt(self, critic_module, masker_module, sepcrit_module=None):
        """Re-homes the flat parameters of nets.NewCritic / nets.UnetDecoder into this engine's buffer, so the
        modules and the engine always see the same weights (no copies at save time)."""
        critic_module._rehome(self.fc)        # values copied once; the modules' 14 per-layer Parameters then alias the engine's buffer
        masker_module._rehome(self.fm)
        if sepcrit_module is not None:
            sepcrit_module._rehome(self.fs)
        self._pver[0] += 1

    def reset_optimizer(self):
        self.m.zero_(); self.v.zero_(); self.step_t.zero_()

    def snapshot_state(self):
        """Copies of (parameters, Adam m, v, step counter): restore_state() puts them back in place (the captured HIP graphs keep
        pointing at the same buffers)."""
        return tuple(t.clone() for t in (self.flat, self.m, self.v, self.step_t))

    def restore_state(self, snap):
        with torch.no_grad():
            for dst, src in zip((self.flat, self.m, self.v, self.step_t), snap):
                dst.copy_(src)
        self._pver[0] += 1

    # ---- helpers -----------------------------------------------------------------------------
    def _cview(self, a: int, b: int) -> Dict[str, torch.Tensor]:
        return {k: t[a:b] for k, t in self.cbuf.items()}

    def _slab_views(self, tag: str, n_first: int, n_second: int):
        """Per-layer slab buffers shared by two backward passes (adjacent => one reduction job each)."""
        lib = _lib.load()
        out1, out2 = {}, {}
        nd = _lib.Dropout(0.0, 0, 0, None, 0, 0)
        tail = hg.TAIL_BWD       # head / features.10 / features.6 slabs then come from the tail kernel's workgroups
        # (tail path: the head's slabs come from ONE cgs_tail_head_wgrad launch over both passes, allocated there)
        specs = [] if tail else [("slab_head", lambda n, first: lib.cgs_head_bwd_slabs(n), hg.HEAD_SLAB)]
        for i, (key, hw, ca, cb, co, ups, act, pool, site) in enumerate(ENC_LAYERS):
            def f(n, first, i=i, hw=hw, ca=ca, cb=cb, co=co, ups=ups, act=act, pool=pool):
                d = hg.conv_desc(n, hw, ca, cb, co, False, ups, act, pool, nd)
                # mirrors critic_backward: features.0 shares a launch with its data gradient only in the first pass
                # (fp32 mixes, image gradient wanted); the second pass reads uint8 frames and needs no image gradient
                in_tail1 = hg.enc1_tail_bwd_fused(n) and hg.ENC1_WGRAD_IN_TAIL and 1 in hg.BOTH_ENC
                # features.3 / features.0: weight gradients formed inside the tail backward kernel (first pass = the mixes, second = A's frames)
                if tail and (i >= 2 or (i == 1 and in_tail1) or (i == 0 and in_tail1 and tag == "p2" and hg.enc0_in_tail(first, not first, None))):
                    return lib.cgs_tail_enc_bwd_slabs(n)
                if i in hg.BOTH_ENC and (i > 0 or first):
                    return lib.cgs_conv3x3_bwd_both_slabs(C.byref(d))
                return lib.cgs_conv3x3_bwd_weight_slabs(C.byref(d))
            specs.append((f"slab_enc{i}", f, 9 * ca * co + co))
        for name, fn, cnt in specs:
            a, b = (fn(n_first, True) if n_first else 0), (fn(n_second, False) if n_second else 0)
            big = torch.zeros((a + b, cnt), device=self.dev)
            self._ws.setdefault("slabs_" + tag, []).append(big)
            out1[name], out2[name] = big[:a], big[a:]
        return out1, out2

    def _opt_range(self):
        """(first float, count) of the optimiser group: everything when live, else masker (+ sepcrit) (main.py:330-334)."""
        return (0, self.total) if self.live else (self.off_m, self.total - self.off_m)

    def _adam(self, tag="p2"):
        if self.fused_tail and not self.dp and tag == "p2":
            return          # phase 2 on one GPU: Adam ran inside cgs_reduce_adam
        lo, cnt = self._opt_range()
        _lib.call("cgs_adam_flat", cnt, C.c_void_p(self.flat.data_ptr() + 4 * lo), C.c_void_p(self.grad.data_ptr() + 4 * lo),
                  C.c_void_p(self.m.data_ptr() + 4 * lo), C.c_void_p(self.v.data_ptr() + 4 * lo), _P(self.step_t),
                  self.lr, self.b1, self.b2, self.eps, 1.0 / self.world, _S())

    def _allreduce(self):
        if self.dp:
            lo, cnt = self._opt_range()
            parallel.allreduce_sum_(self.grad[lo:lo + cnt], self.pg)

    # ---- phase 2 -----------------------------------------------------------------------------
    def _phase2_fwd_bwd(self):
        n = self.n
        nmix = 2 * n if self.inject else n
        drop = self.drop
        A = self.ab[n:]
        B = self.ab[:n]
        # critic on [B | A]
        # (the decoder's bottleneck 1x1 conv rides along with the critic head kernel: o4 for all 2n images, A's half used)
        if getattr(self, "_o4_full", None) is None:
            self._o4_full = torch.empty((2 * n, 32), device=self.dev, dtype=torch.float32)
            self.mbuf["o4"] = self._o4_full[n:]
        fm_ptr = self.fm.data_ptr()
        pw = (C.c_void_p(fm_ptr + 4 * self.lm.off("dec_model.4.weight")), C.c_void_p(fm_ptr + 4 * self.lm.off("dec_model.4.bias")),
              self._o4_full)
        hg.critic_forward(self.fc, self.lc, self.ab, 2 * n, drop.shifted(0), out=self._cview(0, 2 * n),
                          pw=None if self.separate else pw)
        sa = self._cview(n, 2 * n)
        if self.separate:     # main.py:389-390: the masker's inputs come from the second critic's pass over A
            pws = (pw[0], pw[1], self.mbuf["o4"])
            hg.critic_forward(self.fs, self.lc, A, n, drop.shifted(4 * n), out=self.sbuf, pw=pws)
            embeds = [self.sbuf[f"e{i}"] for i in range(5)]
        else:
            embeds = [sa[f"e{i}"] for i in range(5)]
        hg.masker_forward(self.fm, self.lm, A, embeds, n, out=self.mbuf, o4_done=True,
                          zpart=self.zsum if hg.ENC0_MIX_FUSED else None)
        if hg.ENC0_MIX_FUSED:
            # the mixes are virtual: features.0 computes them in its tile loaders (forward here, weight gradient below); the
            # mask layer left the (sum |z|, sum z^2) partials in self.zsum
            mixsrc = hg.MixInput(A, B, self.mbuf["Z"])
            hg.critic_forward(self.fc, self.lc, mixsrc, nmix, drop.shifted(2 * n), out=self._cview(2 * n, 2 * n + nmix))
        else:
            mixsrc = self.mixed[:nmix]
            _lib.call("cgs_mix_fwd", n, 4096, _P(A), _P(B), _P(self.mbuf["Z"]), int(self.inject), _P(self.mixed), _P(self.zsum), _S())
            hg.critic_forward(self.fc, self.lc, mixsrc, nmix, drop.shifted(2 * n), out=self._cview(2 * n, 2 * n + nmix))
        flags = (1 if self.live else 0) | (2 if self.inject else 0) | (4 if self.bce else 0) | (0 if self.staticnorm else 8)
        ft = self.fused_tail
        if not ft:
            _lib.call("cgs_phase2_losses", n, _P(self.cbuf["pred"]), _P(self.y), _P(self.zsum), self.nzpart, self.lfak, self.L1, self.L2,
                      flags, n * 4096, _P(self.losses), _P(self.dpred), _S())
        # (fused tail) targets of the three loss terms: replaced -> pred of B, injected -> pred of A (slots [B | A] = the first 2n
        # entries of pred, aligned with the mix slots [rep | inj]); A -> y with weight lfak (or BCE)
        loss_mix = (self.cbuf["pred"][:nmix], 1.0 / n, False) if ft else None
        loss_a = (self.y, self.lfak / n, self.bce) if ft else None
        plan = self._plans.get("p2")
        first = plan is None
        if first:
            plan = hg.SlabPlan()
            self._sl_mix, self._sl_a = self._slab_views("p2", nmix, n if self.live else 0)
            self._ws["cb_mix"].update(self._sl_mix)
            self._ws["cb_a"].update(self._sl_a)
        pc = plan if first else hg.SlabPlan()   # job registration only matters the first time
        sink_c = [] if hg.TAIL_BWD else None     # the critic's passes whose head weight gradients are formed together below
        # critic backward on the mixes: image gradient for the mask path (+ weight gradients when live)
        nz = float(n * 4096)
        if hg.ENC0_MIX_FUSED:
            # features.0's backward carries the mix backward: the image gradients of the mixes never leave the chip
            hg.critic_backward(self.fc, self.lc, mixsrc, nmix, self._cview(2 * n, 2 * n + nmix),
                               None if ft else self.dpred[2 * n:2 * n + nmix], pc, drop.shifted(2 * n), dx=None, dx_from=0,
                               ws=self._ws["cb_mix"], side=self.side, need_wgrad=self.live, loss=loss_mix, head_sink=sink_c,
                               mix_bwd=(A, B, self.mbuf["Z"], self.inject, self.L1 / nz, self.L2 / nz, self.dzpre,
                                        None if self.staticnorm else sa["pred"]))
        else:
            hg.critic_backward(self.fc, self.lc, self.mixed[:nmix], nmix, self._cview(2 * n, 2 * n + nmix),
                               self.dpred[2 * n:2 * n + nmix], pc, drop.shifted(2 * n), dx=self.dmixed[:nmix], dx_from=0,
                               ws=self._ws["cb_mix"], side=self.side, need_wgrad=self.live)
            _lib.call("cgs_mix_bwd", n, 4096, _P(A), _P(B), _P(self.mbuf["Z"]), _P(self.dmixed), int(self.inject),
                      self.L1 / nz, self.L2 / nz, _P(self.dzpre), _S())
        pm = hg.SlabPlan()
        # live: the 1x1 bottleneck conv's backward runs inside the critic head kernel (frozen: no critic backward on A,
        # the masker does it itself)
        # (live critic, one module: dec_model.0's weight gradient rides in the A pass's tail backward launch below)
        defer = [] if (self.live and not self.separate and hg.TAIL_BWD) else None
        d_emb = hg.masker_backward(self.fm, self.lm, A, embeds, n, self.mbuf, self.dzpre, pm, ws=self._ws["mb"], side=self.side,
                                   pw_in_head=self.live or self.separate, defer_dec0=defer)
        rider = next((d for d in (defer or []) if not isinstance(d, dict)), None)            # dec_model.0's weight gradient (tail launch riders)
        dec3 = next((d["dec3"] for d in (defer or []) if isinstance(d, dict)), None)          # dec_model.3's (features.0 launch riders)
        ps = hg.SlabPlan()
        if self.separate:
            # the skip gradients (and the bottleneck's) go into the SECOND critic; its own head sees no loss (dpred = 0)
            d_o4, d_emb[4] = d_emb[4], None
            hg.critic_backward(self.fs, self.lc, A, n, self.sbuf, None if ft else self._zero_dpred, ps, drop.shifted(4 * n), d_embeds=d_emb,
                               n_add=n, ws=self._ws["cb_sep"], side=self.side,
                               pw_bwd=(d_o4, pw[0], pm, self.lm.off("dec_model.4.weight")))
            if self.live:
                hg.critic_backward(self.fc, self.lc, A, n, sa, None if ft else self.dpred[n:2 * n], pc, drop.shifted(n),
                                   ws=self._ws["cb_a"], side=self.side, loss=loss_a, head_sink=sink_c)
        elif self.live:
            d_o4, d_emb[4] = d_emb[4], None
            hg.critic_backward(self.fc, self.lc, A, n, sa, None if ft else self.dpred[n:2 * n], pc, drop.shifted(n), d_embeds=d_emb,
                               n_add=n, ws=self._ws["cb_a"], side=self.side, loss=loss_a, head_sink=sink_c,
                               pw_bwd=(d_o4, pw[0], pm, self.lm.off("dec_model.4.weight")), rider=rider, dec3_rider=dec3)
        if sink_c:
            hg.head_wgrad(sink_c, pc, self.lc, self._ws["cb_a"])
        self.side.join()
        if first:
            full = hg.SlabPlan()
            if self.live:
                for slab, nsl, cnt, off in pc.jobs:
                    full.jobs.append((slab, nsl, cnt, self.off_c + off))
            for slab, nsl, cnt, off in pm.jobs:
                full.jobs.append((slab, nsl, cnt, self.off_m + off))
            for slab, nsl, cnt, off in ps.jobs:
                full.jobs.append((slab, nsl, cnt, self.off_s + off))
            self._plans["p2"] = full.build(self.grad)
        if ft:      # reduction + Adam + loss values in one launch; the optimiser group is exactly the set of reduced elements
            # (data parallel: param = None -> reduction + loss values + step tick; the all-reduce and one Adam launch follow)
            self._plans["p2"].run_adam(self.step_t, None if self.dp else self.flat, self.grad, self.m, self.v, self.lr, self.b1, self.b2, self.eps,
                                       self._ticket, loss=(n, self.cbuf["pred"], self.y, self.zsum, self.nzpart, self.lfak, self.L1,
                                                           self.L2, flags, n * 4096, self.losses))
        else:
            self._plans["p2"].run(self.step_t)

    def phase2_step(self, A_u8: Optional[torch.Tensor] = None, B_u8: Optional[torch.Tensor] = None,
                    Y: Optional[torch.Tensor] = None):
        """One optimiser step of main.py:344-463.  Inputs (optional: omitted => reuse the resident batch) are
        NHWC uint8 [n,64,64,3] and fp32 [n].  Returns the device tensor losses[8] =
        (critic, replace, inject, l1, l2, total, 0, 0) -- no host sync here."""
        n = self.n
        if A_u8 is not None:
            self.ab[n:].copy_(A_u8, non_blocking=True)
        if B_u8 is not None:
            self.ab[:n].copy_(B_u8, non_blocking=True)
        if Y is not None:
            self.y.copy_(Y.to(torch.float32), non_blocking=True)
        self._run("p2", self._phase2_fwd_bwd)
        self._pver[0] += 1
        return self.losses

    def gather_contrastive(self, Xpos: torch.Tensor, Xneg: torch.Tensor, ypos: torch.Tensor, yneg: torch.Tensor,
                           idx: torch.Tensor, shift_px: int = 0):
        """Assembles the resident batch of a phase-2 step on the device (main.py:344-356): A = [Xpos[idx[:h]] | Xneg[idx[h:n]]]
        rolled along the width by shift_px pixels, B = Xneg[idx[n:2n]] (not rolled), y = the targets of A.  Xpos / Xneg: uint8
        [*,64,64,3] frame sets resident on the device, ypos / yneg fp32 [*], idx int64 [2n] on the device (h = n / 2)."""
        n, h = self.n, self.n // 2
        A, B = self.ab[n:], self.ab[:n]
        ip = idx.data_ptr()
        if GATHER_ONE_LAUNCH:       # (round 5) one launch instead of five
            _lib.call("cgs_gather_contrastive", _P(Xpos), _P(Xneg), _P(ypos), _P(yneg), _P(idx), n, h, int(shift_px) % 64, _P(A), _P(B),
                      _P(self.y), _S())
            return
        g = lambda src, off, cnt, sh, dst: _lib.call("cgs_gather_roll_u8", _P(src), C.c_void_p(ip + 8 * off), cnt, int(sh) % 64,
                                                     C.c_void_p(dst), _S())
        g(Xpos, 0, h, shift_px, A.data_ptr())
        g(Xneg, h, n - h, shift_px, A.data_ptr() + h * 12288)
        g(Xneg, n, n, 0, B.data_ptr())
        _lib.call("cgs_gather_f32", _P(ypos), C.c_void_p(ip), h, _P(self.y), _S())
        _lib.call("cgs_gather_f32", _P(yneg), C.c_void_p(ip + 8 * h), n - h, C.c_void_p(self.y.data_ptr() + 4 * h), _S())

    # ---- phase 1 -----------------------------------------------------------------------------
    def _phase1_fwd_bwd(self):
        n = self.n
        X = self.ab[:n]
        hg.critic_forward(self.fc, self.lc, X, n, self.drop.shifted(0), out=self._cview(0, n))
        _lib.call("cgs_phase1_loss", n, _P(self.cbuf["pred"]), _P(self.y), int(self.bce), _P(self.losses), _P(self.dpred), _S())
        first = "p1" not in self._plans
        pc = hg.SlabPlan()
        # (round 5) as phase 2 does: the head's weight gradients in the features.0 weight-gradient launch, reduction + Adam in one launch on one GPU
        sink = [] if (hg.TAIL_BWD and PHASE1_FUSED_TAIL) else None
        hg.critic_backward(self.fc, self.lc, X, n, self._cview(0, n), self.dpred[:n], pc, self.drop.shifted(0), ws=self._ws["p1"],
                           side=self.side, head_sink=sink)
        if sink:
            hg.head_wgrad(sink, pc, self.lc, self._ws["p1"])
        self.side.join()
        if first:
            full = hg.SlabPlan()
            for slab, nsl, cnt, off in pc.jobs:
                full.jobs.append((slab, nsl, cnt, self.off_c + off))
            self._plans["p1"] = full.build(self.grad)
            if self._p1_fused_adam():
                # the fused reduction + Adam launch updates exactly the elements its jobs cover: that must be the whole critic
                cover = torch.zeros(self.lc.total, dtype=torch.bool)
                for _slab, _nsl, cnt, off in full.jobs:
                    cover[off - self.off_c:off - self.off_c + cnt] = True
                if not bool(cover.all()):
                    raise _lib.CgsError(f"phase 1: the slab jobs cover {int(cover.sum())} of the critic's {self.lc.total} parameters; the "
                                        "fused reduce + Adam launch would leave the rest without an update")
        if self._p1_fused_adam():
            self._plans["p1"].run_adam(self.step_t, self.flat, self.grad, self.m, self.v, self.lr, self.b1, self.b2, self.eps, self._ticket)
        else:
            self._plans["p1"].run(self.step_t)

    def _p1_fused_adam(self):
        return self.fused_tail and not self.dp and PHASE1_FUSED_TAIL

    def _adam_p1(self):
        if self._p1_fused_adam():
            return          # phase 1 on one GPU: Adam ran inside cgs_reduce_adam (the critic's parameters are exactly the reduced elements)
        _lib.call("cgs_adam_flat", self.lc.total, _P(self.flat), _P(self.grad), _P(self.m), _P(self.v), _P(self.step_t),
                  self.lr, self.b1, self.b2, self.eps, 1.0 / self.world, _S())

    def phase1_step(self, X_u8: Optional[torch.Tensor] = None, Y: Optional[torch.Tensor] = None):
        """One optimiser step of main.py:183-200 (critic regression) on n images; returns losses (device)."""
        n = self.n
        if X_u8 is not None:
            self.ab[:n].copy_(X_u8, non_blocking=True)
        if Y is not None:
            self.y.copy_(Y.to(torch.float32), non_blocking=True)
        self._run("p1", self._phase1_fwd_bwd)
        self._pver[0] += 1
        return self.losses

    # ---- execution: eager first call (allocates workspaces, builds job tables), then HIP-graph replay ----
    def _step_parts(self, tag: str):
        """(all-reduce, Adam) callables of a step kind: the two launches that follow the kernels of `body` under data parallelism."""
        adam = self._adam if tag == "p2" else self._adam_p1
        if tag == "p1" and self.dp:
            def allred():
                parallel.allreduce_sum_(self.grad[:self.lc.total], self.pg)
        else:
            allred = self._allreduce
        return allred, adam

    def _capture(self, tag: str, body, single: bool):
        """Records the step of kind `tag` as HIP graph(s).  single (data parallel, RCCL): kernels -> all-reduce -> Adam in ONE graph;
        otherwise one graph (kernels [+ Adam on one GPU]) and, under data parallelism, a second one for Adam with the EAGER all-reduce
        between them.  Nothing is communicated while capturing."""
        allred, adam = self._step_parts(tag)
        if single:
            g1 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1, capture_error_mode="thread_local"):
                body(); allred(); adam()
            return (g1, None, False)
        # (data parallel: thread-local capture mode -- the process group's watchdog thread polls its events while this thread captures)
        kw = {"capture_error_mode": "thread_local"} if self.dp else {}
        g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1, **kw):
            body()
            if not self.dp:
                adam()
        g2 = None
        if self.dp:
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, **kw):
                adam()
        return (g1, g2, self.dp)

    def _run(self, tag: str, body):
        allred, adam = self._step_parts(tag)
        g = self._graphs.get(tag)
        if g is None:
            body(); allred(); adam()                       # eager warm-up: allocations + tables
            if not self.use_graph:
                self._graphs[tag] = "eager"
                return
            # the warm-up was a real step; capture the static sequence for all later steps
            torch.cuda.synchronize()
            # data parallel, RCCL: the all-reduce is captured too when every rank can -- the whole step (kernels -> collective -> Adam)
            # is then ONE graph launch
            single = bool(self.dp and self._collective_capturable())
            self._graphs[tag] = self._capture(tag, body, single)
            self._forms[tag] = {single: self._graphs[tag]}
            if self.dp:
                self.dp_single_graph = single
            return
        if g == "eager":
            body(); allred(); adam()
            return
        g1, g2, eager_allred = g
        g1.replay()
        if eager_allred:
            allred()
            g2.replay()

    def set_dp_launch_form(self, single: bool, tag: str = "p2"):
        """Data parallel only, after the first step of kind `tag`: switches between the two launch forms of the step -- single = the
        all-reduce recorded in the step's HIP graph, else step graph -> eager all-reduce -> Adam graph -- capturing the missing one
        (no communication while capturing).  Both forms run the same kernels on the same buffers: a step is bit-identical in either
        (test_dp_launch_form_matches_single_gpu_step).  EVERY rank of the group must make the same call at the same point of the
        program (bench.py decides from all-reduced quantities); `single` needs parallel.collective_capturable() == True on the group."""
        if not self.dp or self._graphs.get(tag) in (None, "eager"):
            raise _lib.CgsError("set_dp_launch_form: a data-parallel engine with captured graphs (run one step first; use_graph=True)")
        forms = self._forms[tag]
        if single not in forms:
            torch.cuda.synchronize()
            # (the body is looked up here, not remembered: a stored bound method would tie the engine into a reference cycle, and an engine
            #  freed by the cycle collector -- at an arbitrary later moment, e.g. inside another engine's capture -- destroys its graphs there)
            forms[single] = self._capture(tag, self._phase2_fwd_bwd if tag == "p2" else self._phase1_fwd_bwd, single)
        self._graphs[tag] = forms[single]
        self.dp_single_graph = bool(single)
        self.dp_capture_note = ("selected by set_dp_launch_form: the all-reduce is recorded in the step graph" if single else
                                "selected by set_dp_launch_form: eager all-reduce between the step graph and the Adam graph")

    def _collective_capturable(self) -> bool:
        """True when the gradient all-reduce can be recorded into the step's HIP graph: the RCCL backend on device memory (gloo
        rehearsals stage through the host) and a trial capture + replay of a small all-reduce on this group succeeds.  The reason for
        a refusal is kept in self.dp_capture_note (bench.py prints it)."""
        if getattr(self, "_capturable", None) is not None:
            return self._capturable
        # the REQUEST itself (dp_graph argument / CGS_DP_GRAPH, resolved per rank) is agreed on first: a rank that skipped the collective
        # trial below while its peer entered it would deadlock the job
        if not parallel.agree_all(bool(self.dp_graph), self.pg, self.dev):
            self.dp_capture_note = ("not requested on every rank (dp_graph False here or on a peer; the default at world > 1, see "
                                    "parallel.resolve_dp_graph)")
            self._capturable = False
            return False
        ok, self.dp_capture_note = parallel.collective_capturable(self.pg, self.dev)
        self._capturable = ok
        return ok

    # ---- saliency baseline (main.py:941-953): |d mean(pred) / d X| summed over the colour channels ----------------
    def saliency(self, X: torch.Tensor):
        """Eval-mode critic forward + backward to the input.  X: NHWC fp32 [b,64,64,3] in [0,1] on the device.
        Returns (pred [b], d mean(pred)/dX [b,64,64,3]); the caller takes abs().sum(channels) like the reference."""
        hg._chk_img(X, 0, "saliency input")
        b = X.shape[0]
        X = X.contiguous()
        if X.dtype != torch.float32:
            raise _lib.CgsError("saliency needs the fp32 image batch (the gradient is taken w.r.t. it)")
        c = hg.critic_forward(self.fc, self.lc, X, b)
        dpred = torch.full((b,), 1.0 / b, device=X.device, dtype=torch.float32)      # pred.mean().backward()
        dx = torch.empty((b, 64, 64, 3), device=X.device, dtype=torch.float32)
        hg.critic_backward(self.fc, self.lc, X, b, c, dpred, hg.SlabPlan(), dx=dx, dx_from=0, need_wgrad=False)
        return c["pred"], dx

    # ---- inference (main.py:1130-1151) -----------------------------------------------------------
    def _infer_f16(self, X: torch.Tensor, want_mask: bool, chfak: int, neck: int):
        """BASELINE config 4: every layer with fp16 activations / weights, fp32 accumulation (csrc/gen_f16.hip)."""
        from . import generic as gen
        if self._w16 is None:
            self._w16 = gen.F16Weights()
        # the cache key also carries the flat buffer's torch version counter: module.load_state_dict / an external optimiser change the
        # weights through torch ops without going through this engine (its own kernels write through raw pointers and bump _pver)
        w16 = self._w16.get(self.fc, self.lc, self.fm, self.lm, chfak, neck, 16, (self._pver[0], self.flat._version, self.separate))
        X = X.contiguous()
        emb = None
        if self.separate and want_mask:
            if getattr(self, "_w16s", None) is None:
                self._w16s = gen.F16Weights()
            ws = self._w16s.get(self.fs, self.lc, None, None, chfak, neck, 16, (self._pver[0], self.flat._version))
            _, _, emb = gen.infer_f16(self.fs, self.lc, None, None, X, chfak, neck, ws)
        pred, Z, _ = gen.infer_f16(self.fc, self.lc, self.fm if want_mask else None, self.lm, X, chfak, neck, w16, embeds_from=emb)
        return pred, Z

    def _infer_f16_fused(self, X: torch.Tensor, want_mask: bool):
        """BASELINE config 4 on the FUSED path (round 4): features.0 / features.3 / dec_model.0 on the fp16 convolutions of
        csrc/hconv.hip (fp16 activations in HBM, v_mfma_f32_16x16x32_f16), the mask head on the fp16 one-kernel form reading the fp16
        o0, the 16x16-and-smaller layers on the fp32 tail kernels.  uint8 frames, eval mode, one critic."""
        b, dev = X.shape[0], X.device
        ws = getattr(self, "_f16ws", None)
        if ws is None or ws["b"] != b:
            f16 = lambda *s: torch.empty(s, device=dev, dtype=torch.float16)
            f32 = lambda *s: torch.empty(s, device=dev, dtype=torch.float32)
            ws = self._f16ws = dict(b=b, e0=f16(b, 32, 32, 8), e1=f32(b, 16, 16, 8), e2=f32(b, 8, 8, 8), am2=torch.empty((b, 8, 8, 1), device=dev, dtype=torch.int32),
                                    e3=f32(b, 4, 4, 16), am3=torch.empty((b, 4, 4, 2), device=dev, dtype=torch.int32), e4=f32(b, 32), h1=f32(b, 32),
                                    o4=f32(b, 32), o3=f32(b, 4, 4, 16), o2=f32(b, 8, 8, 8), o1=f32(b, 16, 16, 8), o0=f16(b, 32, 32, 8))
        pred = torch.empty(b, device=dev, dtype=torch.float32)
        cp, mp = self.fc.data_ptr(), self.fm.data_ptr()
        wc = lambda k: C.c_void_p(cp + 4 * self.lc.off(k))
        wm = lambda k: C.c_void_p(mp + 4 * self.lm.off(k))
        _lib.call("cgs_f16_enc0_fwd", b, _P(X), wc("features.0.weight"), wc("features.0.bias"), _P(ws["e0"]), _S())
        if F16_TAILS != "fused1":
            _lib.call("cgs_f16_enc1_fwd", b, _P(ws["e0"]), wc("features.3.weight"), wc("features.3.bias"), _P(ws["e1"]), _S())
        tw = hg.tail_enc_weights(self.fc, self.lc, (mp + 4 * self.lm.off("dec_model.4.weight"), mp + 4 * self.lm.off("dec_model.4.bias")))
        # (round 6) the tails' 3x3 layers with fp16 operands too (csrc/tail_h16.h): at this batch the fp32 decoder tail was bound by the fp32 matrix pipe
        if F16_TAILS == "fused1":
            td = hg.tail_dec_weights(self.fm, self.lm) if want_mask else None
            _lib.call("cgs_f16_enc1_tail_infer", b, _P(ws["e0"]), wc("features.3.weight"), wc("features.3.bias"), C.byref(tw),
                      C.byref(td) if want_mask else None, _P(pred), _P(ws["o1"]) if want_mask else None, _S())
            if not want_mask:
                return pred, None
        elif F16_TAILS == "fused":
            td = hg.tail_dec_weights(self.fm, self.lm) if want_mask else None
            _lib.call("cgs_tail_infer_h16", b, C.byref(tw), C.byref(td) if want_mask else None, _P(ws["e1"]), _P(pred),
                      _P(ws["o1"]) if want_mask else None, _S())
            if not want_mask:
                return pred, None
        elif F16_TAILS != "0":
            _lib.call("cgs_tail_enc_fwd_h16", b, C.byref(tw), _P(ws["e1"]), _P(ws["e2"]), _P(ws["am2"]), _P(ws["e3"]), _P(ws["am3"]), _P(ws["e4"]),
                      _P(ws["h1"]), _P(pred), _P(ws["o4"]) if want_mask else None, _S())
        else:
            nd = _lib.Dropout(0.0, 0, 0, None, 0, 0)
            _lib.call("cgs_tail_enc_fwd", b, C.byref(tw), _P(ws["e1"]), _P(ws["e2"]), _P(ws["am2"]), _P(ws["e3"]), _P(ws["am3"]), _P(ws["e4"]),
                      _P(ws["h1"]), _P(pred), _P(ws["o4"]) if want_mask else None, nd, nd, nd, _S())
        if F16_TAILS not in ("fused", "fused1"):
            if not want_mask:
                return pred, None
            td = hg.tail_dec_weights(self.fm, self.lm)
            _lib.call("cgs_tail_dec_fwd_h16" if F16_TAILS != "0" else "cgs_tail_dec_fwd", b, C.byref(td), _P(ws["e1"]), _P(ws["e2"]), _P(ws["e3"]), _P(ws["o4"]),
                      _P(ws["o3"]), _P(ws["o2"]), _P(ws["o1"]), _S())
        _lib.call("cgs_f16_dec0_fwd", b, _P(ws["e0"]), _P(ws["o1"]), wm("dec_model.0.weight"), wm("dec_model.0.bias"), _P(ws["o0"]), _S())
        Z = torch.empty((b, 64, 64), device=dev, dtype=torch.float32)
        _lib.call("cgs_mask_infer_fwd_f16o", b, _lib.SRC_U8, _P(X), _P(ws["o0"]), wm("masker.0.weight"), wm("masker.0.bias"), wm("masker.2.weight"),
                  wm("masker.2.bias"), _P(Z), _S())
        return pred, Z

    @torch.no_grad()
    def infer(self, X: torch.Tensor, want_mask: bool = True, fp16_mask_head: bool = False, train_mode: bool = False, fp16: bool = False,
              fp16_layerwise: bool = False):
        """Eval-mode critic (+ masker).  X: NHWC uint8 or fp32 [b,64,64,3] on the device.
        Returns (pred [b], Z [b,64,64] or None).  fp16_mask_head (opt-in): the mask head on fp16 MFMA operands -- the masker.0 GEMM
        (uint8 frames: bytes scaled by 1/1024 exactly, 1024/255 in the weights), its output h rounded to fp16 after a packed-fp16 LeakyReLU, and masker.2's
        tap products; fp32 accumulation, fp32 sums of the tap planes, sigmoid via v_exp / v_rcp.  Z differs from the fp32 path by < 2e-3 absolute
        (tests/test_gpu_kernels.py: max 2e-3, mean 3e-4 bounds; measured on the G1 weights ~3e-5 max).
        fp16 (opt-in, uint8 frames, eval mode): BASELINE config 4 -- the fused fp16 path (fp16 activations / weights in the 64x64 and 32x32
        convolutions and the mask head, fp32 accumulation, the 16x16-and-smaller tail in fp32); fp16_layerwise=True: the shape-generic
        chain with fp16 activations in EVERY layer (what chfak != 1 runs).
        train_mode (-noevalmode, main.py:1109-1118): Dropout stays active, a fresh mask per call.
        With a second critic (-separate, main.py:1140-1142) the masker's inputs come from it."""
        hg._chk_img(X, 0, "infer input")
        if fp16:
            if train_mode or fp16_mask_head:
                raise _lib.CgsError("fp16 inference is an eval-mode path of its own (no Dropout, no fp16_mask_head)")
            # chfak 1, uint8 frames, one critic: the fused fp16 path (round 4); fp16_layerwise keeps the shape-generic layer-by-layer chain
            if type(self) is HourglassEngine and not self.separate and X.dtype == torch.uint8 and not fp16_layerwise:
                return self._infer_f16_fused(X.contiguous(), want_mask)
            return self._infer_f16(X, want_mask, 1, 32)
        b = X.shape[0]
        drop = hg.NO_DROP
        if train_mode and self.p > 0.0:
            drop = hg.DropState(self.p, self.drop.seed ^ 0xD1CE, self._infer_step)
        if not want_mask:
            out = hg.critic_forward(self.fc, self.lc, X.contiguous(), b, drop)["pred"], None
        else:
            o4 = torch.empty((b, 32), device=X.device, dtype=torch.float32)
            fm_ptr = self.fm.data_ptr()
            pw = (C.c_void_p(fm_ptr + 4 * self.lm.off("dec_model.4.weight")), C.c_void_p(fm_ptr + 4 * self.lm.off("dec_model.4.bias")), o4)
            c = hg.critic_forward(self.fc, self.lc, X.contiguous(), b, drop, pw=None if self.separate else pw)
            src = hg.critic_forward(self.fs, self.lc, X.contiguous(), b, drop.shifted(b), pw=pw) if self.separate else c
            m = hg.masker_forward(self.fm, self.lm, X.contiguous(), [src[f"e{i}"] for i in range(5)], b, out={"o4": o4},
                                  o4_done=True, keep_hm=False, fp16_mask_head=fp16_mask_head)
            out = c["pred"], m["Z"]
        if drop is not hg.NO_DROP:
            self._infer_step += 1
        return out
