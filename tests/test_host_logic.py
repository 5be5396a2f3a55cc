"""CPU tests (no GPU): flat-layout <-> reference state_dict, module construction / checkpoint contract,
CLI surface, the C-ABI library exports, and the data-parallel scheme over gloo (world_size 2)."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

import cgs_amd
from cgs_amd import _lib, cli, handler, nets, parallel, spec
from oracle import hourglass_ref as orc

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_layout_totals_and_roundtrip(g1):
    pc, pm = g1
    for lay, sd, total in ((spec.critic_layout(), pc, 11873), (spec.masker_layout(), pm, 13785)):
        assert lay.total == total
        flat = torch.empty(lay.total)
        lay.flatten(sd, flat)
        back = lay.unflatten(flat)
        assert list(back.keys()) == list(sd.keys())
        for k in sd:
            np.testing.assert_array_equal(back[k].numpy(), sd[k].numpy())


def test_layout_is_hwio_and_k_major(g1):
    pc, pm = g1
    lay = spec.critic_layout()
    flat = torch.empty(lay.total)
    lay.flatten(pc, flat)
    w = pc["features.3.weight"]                      # OIHW [8,8,3,3]
    seg = lay.segs["features.3.weight"]
    hwio = flat[seg.offset:seg.offset + seg.count].reshape(3, 3, 8, 8)
    assert float(hwio[2, 1, 5, 3]) == float(w[3, 5, 2, 1])
    w4 = pc["features.14.weight"]                    # [32,16,4,4] -> [k=(y*4+x)*16+c][o]
    seg = lay.segs["features.14.weight"]
    k = flat[seg.offset:seg.offset + seg.count].reshape(256, 32)
    assert float(k[(2 * 4 + 3) * 16 + 7, 11]) == float(w4[11, 7, 2, 3])
    l1 = pc["crit.1.weight"]                         # [o][k] -> [k][o]
    seg = lay.segs["crit.1.weight"]
    assert float(flat[seg.offset + 5 * 32 + 9]) == float(l1[9, 5])
    lm = spec.masker_layout()
    fm = torch.empty(lm.total)
    lm.flatten(pm, fm)
    seg = lm.segs["dec_model.4.weight"]              # 1x1 conv [o][i][1][1] -> [i][o]
    assert float(fm[seg.offset + 3 * 32 + 20]) == float(pm["dec_model.4.weight"][20, 3, 0, 0])


def test_layout_rejects_wrong_checkpoints(g1):
    pc, _ = g1
    lay = spec.critic_layout()
    flat = torch.empty(lay.total)
    bad = dict(pc)
    bad.pop("crit.4.bias")
    with pytest.raises(RuntimeError, match="missing"):
        lay.flatten(bad, flat)
    bad = dict(pc)
    bad["features.0.weight"] = torch.zeros(8, 3, 5, 5)
    with pytest.raises(RuntimeError, match="size mismatch"):
        lay.flatten(bad, flat)


def test_modules_init_like_reference_and_state_dict_contract(golden, tmp_path):
    """torch.manual_seed(0) + construction gives the reference's default-init weights (fixture G1), the
    state_dict has the reference's keys/shapes and survives torch.save / torch.load."""
    g = golden("g1_weights_chfak1.npz")
    with open(os.path.join(REPO, "tests", "golden", "g1_keys.json")) as fp:
        keys = json.load(fp)["chfak1"]
    torch.manual_seed(0)
    c = nets.NewCritic(bottleneck=32, chfak=1, dropout=0.3)
    m = nets.UnetDecoder(bottleneck=32, chfak=1)
    for mod, name in ((c, "critic"), (m, "masker")):
        sd = mod.state_dict()
        assert {k: list(v.shape) for k, v in sd.items()} == keys[name]
        for k, v in sd.items():
            np.testing.assert_array_equal(v.numpy(), g[f"{name}/{k}"])
        path = tmp_path / f"{name}.pt"
        torch.save(sd, path)
        other = type(mod)()
        other.load_state_dict(torch.load(path, map_location="cpu"))
        np.testing.assert_array_equal(other.flat.detach().numpy(), mod.flat.detach().numpy())
    # row b: the reference's 14 keyed Parameters per module (nets.py:170-194, 479-492), views of ONE flat kernel-layout buffer
    import copy
    for mod, name in ((c, "critic"), (m, "masker")):
        assert [(k, list(q.shape)) for k, q in mod.named_parameters()] == list(keys[name].items())
        assert all(isinstance(q, torch.nn.Parameter) and q.is_leaf and q.requires_grad for q in mod.parameters())
        assert "flat" not in dict(mod.named_parameters()) and "_flat" not in dict(mod.named_buffers())
        assert sum(q.numel() for q in mod.parameters()) == mod.flat.numel()
        base = mod.flat.data_ptr()
        for q, seg in zip(mod.parameters(), mod.layout.segs.values()):
            assert q.data_ptr() == base + 4 * seg.offset
        for k, q in mod.named_parameters():
            np.testing.assert_array_equal(q.detach().numpy(), g[f"{name}/{k}"])
        # writes through a Parameter land in the flat buffer (kernel layout: HWIO), and vice versa
        w = dict(mod.named_parameters())[next(iter(keys[name]))]
        with torch.no_grad():
            w[1, 2, 0, 1] = 7.5
        seg = next(iter(mod.layout.segs.values()))
        o, i, kh, kw = seg.ref_shape
        assert float(mod.flat[seg.offset + ((0 * kw + 1) * i + 2) * o + 1]) == 7.5
        # .to() moves the buffer as one tensor and keeps Parameter identity + aliasing; deepcopy / p.data = t are re-aliased on use
        ids = [id(q) for q in mod.parameters()]
        mod.to(torch.device("cpu")).float()
        assert ids == [id(q) for q in mod.parameters()]
        dup = copy.deepcopy(mod)
        assert dup.flat.data_ptr() != mod.flat.data_ptr()
        assert all(q.data_ptr() == dup.flat.data_ptr() + 4 * s.offset for q, s in zip(dup.parameters(), dup.layout.segs.values()))
        np.testing.assert_array_equal(dup.flat.numpy(), mod.flat.numpy())
        with pytest.raises(_lib.CgsError, match="fp32"):
            copy.deepcopy(mod).half()
    assert c.flat.numel() == 11873 and len(list(c.parameters())) == 14 and len(list(m.parameters())) == 14
    assert [n for n, _ in c.named_children()] == ["features", "crit"] and [n for n, _ in m.named_children()] == ["dec_model", "masker"]
    assert c.training and not c.eval().training


def test_modules_fail_loudly_without_gpu_and_on_unsupported_shapes():
    c = nets.NewCritic()
    with pytest.raises(_lib.CgsError, match="no CPU fallback"):
        c(torch.zeros(2, 3, 64, 64))
    c5 = nets.NewCritic(chfak=5)        # the paper's size: constructs, checkpoint-compatible; runs on the generic forward kernels
    assert [(k, tuple(v.shape)) for k, v in c5.state_dict().items()] == [(k, s) for k, s in orc.critic_shapes(5)]
    with pytest.raises(NotImplementedError):
        nets.NewCritic(dims=[4, 4, 4, 8])
    with pytest.raises(NotImplementedError):
        nets.UnetDecoder(upsample=False)
    # the optional BatchNorm epilogue and the 128x128 variant: same rule -- no CPU path
    bn = nets.BatchNormAct2d(8, act="relu")
    assert sorted(k for k, _ in bn.state_dict().items()) == ["bias", "num_batches_tracked", "running_mean", "running_var", "weight"]
    with pytest.raises(_lib.CgsError, match="no CPU fallback"):
        bn(torch.zeros(2, 8, 4, 4))
    with pytest.raises(NotImplementedError):
        nets.BatchNormAct2d(10)
    if not torch.cuda.is_available():
        from cgs_amd import hourglass128
        with pytest.raises(_lib.CgsError, match="no CPU fallback"):
            hourglass128.Hourglass128(orc.seeded_params(orc.critic128_shapes(), 1), orc.seeded_params(orc.masker128_shapes(), 2))


def test_cli_surface_matches_reference():
    a = cli.parse_args([])
    assert (a.dropout, a.chfak, a.shift, a.lfak, a.neck, a.cepochs, a.mepochs, a.L1, a.L2) == (0.3, 1, 12, 5, 32, 15, 1, 0.5, 0.0)
    assert (a.high_rew_thresh, a.low_rew_thresh, a.saveevery, a.rewidx, a.testsize, a.datasize) == (0.7, 0.3, 5, 1, 5000, 100000)
    assert a.binarymaskthreshold == 0.5 and a.mask_output_imgs == "results" and a.gammas == "0.98-0.97-0.96-0.95"
    assert a.live and a.inject and a.cload and a.mload and a.name == "default-model"
    b = cli.parse_args(["-train", "-frozen", "-noinject", "--model", "m", "-cload", "False"])
    assert b.train and not b.live and not b.inject and b.name == "m"
    assert b.cload is True        # type=bool quirk of the reference: bool("False") is True
    t = cli.parse_args(["-test"])
    assert t.eval and t.salience and not t.train
    p = cli.parse_args(["-process", "-concatenated", "--source-imgs", "in", "--mask-output-imgs", "out"])
    assert p.process and p.concatenated and p.source_imgs == "in" and p.mask_output_imgs == "out"


def test_handler_shift_batch_matches_reference_capture(golden):
    """The PRODUCT shift_batch (handler.Handler.shift_batch, main.py:584-591) against the reference capture G5: same two
    draws from the global torch RNG, bit-exact rolled batch."""
    import types
    g = golden("g5_shift.npz")
    X = torch.from_numpy(g["X"])
    me = types.SimpleNamespace(args=types.SimpleNamespace(shift=12))
    for k in range(4):
        torch.manual_seed(k)
        rolled = handler.Handler.shift_batch(me, X)
        np.testing.assert_array_equal(rolled.numpy(), g[f"rolled{k}"])
        assert rolled.dtype == torch.uint8 and rolled.is_contiguous()


def test_flags_outside_the_build_are_refused_not_ignored():
    import types
    me = types.SimpleNamespace(args=cli.parse_args(["-staticnorm", ""]))
    assert me.args.staticnorm is False
    handler.Handler._refuse_unbuilt_flags(me)      # implemented on the default (fused mix backward) path
    handler.Handler._refuse_unbuilt_flags(types.SimpleNamespace(args=cli.parse_args([])))   # defaults pass


def test_checkpoint_names_follow_reference_mangling():
    c, m = handler.checkpoint_names(cli.parse_args([]))
    assert c == "rewidx=1-cepochs=15-datamode=trunk-datasize=100000-shift=12-chfak=1-dropout=0.3"
    assert m == "mepochs=1-L1=0.5-inject=True"
    c, m = handler.checkpoint_names(cli.parse_args(["--dropout", "0", "-noinject", "--L2", "0.1", "--threshrew", "0.5"]))
    assert c == "rewidx=1-cepochs=15-datamode=trunk-datasize=100000-threshrew=0.5-shift=12-chfak=1"
    assert m == "mepochs=1-L1=0.5-L2=0.1"


def test_c_abi_library_exports_every_declared_symbol():
    """libcgs_hip.so loads (no GPU needed for dlopen) and exports exactly the entry points include/cgs_hip.h
    declares; the ctypes signature table covers all of them.  No compute call is made here."""
    with open(os.path.join(REPO, "include", "cgs_hip.h")) as fp:
        text = fp.read()
    declared = set(re.findall(r"\b(cgs_[a-z0-9_]+)\s*\(", text))
    declared -= {"cgs_stream_t"}
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name)
    assert lib.cgs_build_arch() == b"gfx950" and lib.cgs_abi_version() == 1
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (cgs_[a-z0-9_]+)", nm))
    assert exported == declared
    # pure host-side queries are safe without a GPU
    assert lib.cgs_head_bwd_slabs(512) == 64 and lib.cgs_pointwise_bwd_slabs(512) == 16
    assert lib.cgs_mix_fwd_partials(512, 4096) == 1024
    d = _lib.ConvDesc(512, 64, 64, 3, 8, 16, _lib.SRC_U8, 2, _lib.ACT_LRELU, 0, _lib.Dropout())
    assert lib.cgs_conv3x3_bwd_weight_slabs(d) > 0
    bad = _lib.ConvDesc(512, 48, 48, 3, 8, 16, _lib.SRC_U8, 2, _lib.ACT_LRELU, 0, _lib.Dropout())
    assert lib.cgs_conv3x3_bwd_weight_slabs(bad) == _lib.ERR_UNSUPPORTED


def test_shard_slice():
    assert parallel.shard_slice(512, 3, 8) == slice(192, 256)
    with pytest.raises(ValueError):
        parallel.shard_slice(10, 0, 4)


GLOO_WORKER = r"""
import os, sys
sys.path.insert(0, {repo!r})
import numpy as np, torch
import torch.distributed as dist
import cgs_amd
from cgs_amd import parallel, spec
from oracle import hourglass_ref as orc
pg = parallel.init_from_env("gloo")
rank, _, world = parallel.env_world()
raw = dict(np.load(os.path.join({repo!r}, "tests", "golden", "g1_weights_chfak1.npz")))
pc = {{k.split("/", 1)[1]: torch.from_numpy(v) for k, v in raw.items() if k.startswith("critic/")}}
pm = {{k.split("/", 1)[1]: torch.from_numpy(v) for k, v in raw.items() if k.startswith("masker/")}}
rs = np.random.RandomState(0)
n = 8
A = rs.randint(0, 256, (n, 64, 64, 3)).astype(np.uint8); B = rs.randint(0, 256, (n, 64, 64, 3)).astype(np.uint8)
Y = torch.from_numpy(rs.rand(n).astype(np.float32))
lc, lm = spec.critic_layout(), spec.masker_layout()
def flat_grads(sl):
    rec = orc.train_phase2(pc, pm, [(orc.u8_to_nchw(A[sl]), orc.u8_to_nchw(B[sl]), Y[sl])], steps=1)[0]
    g = torch.zeros(lc.total + lm.total)
    lc.flatten(rec["grads_c"], g[:lc.total]); lm.flatten(rec["grads_m"], g[lc.total:])
    return g
ok, note = parallel.collective_capturable(pg, torch.device("cpu"))      # gloo stages through the host: never recorded in a HIP graph
assert ok is False and "gloo" in note, (ok, note)
# the launch-form decision is collective: one rank's failure makes EVERY rank fall back (ADVICE round 4: ranks must never end in different forms)
assert parallel._agree_all(True, pg, torch.device("cpu")) is True
assert parallel._agree_all(rank != 1, pg, torch.device("cpu")) is False
assert parallel.resolve_dp_graph(None, world) is False and parallel.resolve_dp_graph(None, 1) is True     # eager at N > 1 until exercised
assert parallel.resolve_dp_graph(True, world) is True and parallel.resolve_dp_graph(False, 1) is False
# round 6: replicas are compared bit for bit through all-gathered 64-bit checksums (bench.py --gpus N: config.replicas_bit_identical)
t = torch.arange(1000, dtype=torch.float32) * 0.37
same, sums = parallel.replica_checksums([t, t.to(torch.int64)], pg, torch.device("cpu"))
assert same is True and len(sums) == world and len(set(sums)) == 1 and len(sums[0]) == 4 * 16, (same, sums)
u = t.clone()
if rank == 1:
    u[123] = torch.nextafter(u[123], torch.tensor(1e9))          # one ulp on one rank
same, sums = parallel.replica_checksums([u], pg, torch.device("cpu"))
assert same is False and sums[0] != sums[1]
w = t.clone(); w[[3, 4]] = w[[4, 3]]                              # a reordering keeps the plain sum and moves the weighted one
a, b = parallel.checksum64(t), parallel.checksum64(w)
assert int(a[0]) == int(b[0]) and int(a[1]) != int(b[1])
# every rank starts from rank 0's parameters
p = torch.full((5,), float(rank)); parallel.broadcast_params_(p, pg); assert float(p.sum()) == 0.0
g = flat_grads(parallel.shard_slice(n, rank, world))
parallel.allreduce_sum_(g, pg)
g /= world                                   # what Adam's grad_scale = 1/world does on the device
if rank == 0:
    full = flat_grads(slice(0, n))
    err = float((g - full).abs().max() / full.abs().max())
    print("DP_MAX_REL_ERR", err)
    assert err < 2e-5, err
dist.barrier(); dist.destroy_process_group()
"""


def test_data_parallel_scheme_world2_gloo(tmp_path):
    """World-size-2 rehearsal of the DP step on CPU (gloo): shard the batch, all-reduce(sum) ONE flat gradient
    bucket, scale by 1/world == the single-process full-batch gradient (no BatchNorm => exact up to fp order)."""
    script = tmp_path / "dp_worker.py"
    script.write_text(GLOO_WORKER.format(repo=REPO))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29517", str(script)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "DP_MAX_REL_ERR" in r.stdout


def test_hourglass128_initialiser_matches_the_oracle_tables():
    """bench.py --config 5 takes its stand-in weights from the product module (Hourglass128.seeded_state), not from oracle/: same key
    names / shapes / values as the oracle's tables the GPU tests compare against."""
    import torch
    from cgs_amd import hourglass128 as h
    from oracle import hourglass_ref as orc
    crit, mask = h.Hourglass128.param_shapes()
    assert crit == orc.critic128_shapes() and mask == orc.masker128_shapes()
    pc, pm = h.Hourglass128.seeded_state(31)
    oc, om = orc.seeded_params(orc.critic128_shapes(), 31), orc.seeded_params(orc.masker128_shapes(), 32)
    assert all(torch.equal(pc[k], oc[k]) for k in oc) and all(torch.equal(pm[k], om[k]) for k in om)
    src = open(os.path.join(REPO, "bench.py")).read()
    assert src.count("from oracle import") == 1, "bench.py may import oracle/ only in its cpu_baseline leg"


def test_bench_side_child_exits_when_the_parent_goes_away():
    """The side-block child idles on its stdin pipe; EOF (the parent died before the go line) makes it exit without touching a GPU."""
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--side-child"], input="", capture_output=True, text=True, timeout=300)
    assert r.returncode == 3 and r.stdout.strip() == ""
