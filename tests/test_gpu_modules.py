"""GPU tests of the drop-in boundary: the nn.Module API driven exactly the way the reference's main.py drives
it (torch.optim.Adam + F.mse_loss around NewCritic / UnetDecoder), the -train / -process command line on a
synthetic data set, and the world_size-2 data-parallel engine."""
import gzip
import json
import os
import pickle
import subprocess
import sys
from itertools import chain

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import hourglass_ref as orc
from test_gpu_kernels import rel_close

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def split(raw, prefix):
    return {k[len(prefix) + 1:]: v for k, v in raw.items() if k.startswith(prefix + "/")}


def modules(g1, dropout=0.0):
    from cgs_amd import nets
    pc, pm = g1
    c = nets.NewCritic(bottleneck=32, chfak=1, dropout=dropout).to("cuda")
    m = nets.UnetDecoder(bottleneck=32, chfak=1).to("cuda")
    c.load_state_dict(pc)
    m.load_state_dict(pm)
    return c, m


def test_module_forward_shapes_and_values(golden, g1):
    g = golden("g2_eval.npz")
    c, m = modules(g1)
    c.eval(); m.eval()
    X = orc.u8_to_nchw(g["X"]).to("cuda")
    pred, embeds = c(X, collect=True)
    Z = m(X, embeds)
    assert pred.shape == (8, 1) and Z.shape == (8, 1, 64, 64)
    assert [tuple(e.shape) for e in embeds] == [(8, 8, 32, 32), (8, 8, 16, 16), (8, 8, 8, 8), (8, 16, 4, 4), (8, 32, 1, 1)]
    rel_close(pred.detach().cpu().numpy(), g["pred"], "pred")
    for i in range(5):
        rel_close(embeds[i].detach().cpu().numpy(), g[f"e{i}"], f"e{i}")
    rel_close(Z.detach().cpu().numpy(), g["Z"], "Z")
    assert c(X).shape == (8, 1)     # collect=False returns the prediction only


@pytest.mark.parametrize("tag,kw", [("g3_train_default", dict()), ("g3_train_noinject", dict(inject=False)),
                                    ("g3_train_frozen", dict(live=False))])
def test_reference_style_training_loop_through_modules(golden, g1, tag, kw):
    """main.py:360-463 written against the module API (torch ops for the mix and the losses, torch.optim.Adam
    for the update): 3 steps reproduce the reference's losses, gradients and parameters."""
    g = golden(tag + ".npz")
    inject, live = kw.get("inject", True), kw.get("live", True)
    critic, masker = modules(g1)
    critic.train(); masker.train()
    opti = torch.optim.Adam(chain(critic.parameters(), masker.parameters())) if live else torch.optim.Adam(masker.parameters())
    A = orc.u8_to_nchw(g["A"]).to("cuda")
    B = orc.u8_to_nchw(g["B"]).to("cuda")
    Y = torch.from_numpy(g["Y"]).to("cuda")
    for s in range(3):
        pred, embeds = critic(A, collect=True)
        negpred = critic(B)
        pred = pred.squeeze(); negpred = negpred.squeeze().detach()
        loss = 0
        parts = np.zeros(5)
        if live:
            cl = F.mse_loss(pred, Y); loss = loss + 5 * cl; parts[0] = cl.item()
        Z = masker(A, embeds)
        replaced = A * (1 - Z) + Z * B
        rl = F.mse_loss(critic(replaced).squeeze(), negpred.detach()); loss = loss + rl; parts[1] = rl.item()
        if inject:
            injected = B * (1 - Z) + Z * A
            il = F.mse_loss(critic(injected).squeeze(), pred.detach()); loss = loss + il; parts[2] = il.item()
        nl = 0.5 * F.l1_loss(Z, torch.zeros_like(Z)); loss = loss + nl; parts[3] = nl.item()
        opti.zero_grad()
        loss.backward()
        if s == 0:          # per-key .grad of the 14 + 14 reference-named Parameters (nets.py:170-194, 479-492)
            gm = {k: q.grad for k, q in masker.named_parameters()}
            for k, v in split(g, "grad/masker").items():
                assert gm[k].shape == v.shape
                rel_close(gm[k].cpu().numpy(), v, f"masker grad {k}")
            if live:
                gc = {k: q.grad for k, q in critic.named_parameters()}
                for k, v in split(g, "grad/critic").items():
                    assert gc[k].shape == v.shape
                    rel_close(gc[k].cpu().numpy(), v, f"critic grad {k}")
        opti.step()
        np.testing.assert_allclose(parts, g[f"parts{s}"], rtol=1e-3, atol=1e-7)
        assert loss.item() == pytest.approx(float(g[f"total{s}"]), rel=1e-3)
    for k, v in split(g, "step3/masker").items():
        rel_close(masker.state_dict()[k].cpu().numpy(), v, f"masker {k} after 3 steps", atol_scale=1e-4)
    for k, v in split(g, "step3/critic").items():
        rel_close(critic.state_dict()[k].cpu().numpy(), v, f"critic {k} after 3 steps", atol_scale=1e-4)


def test_named_parameters_are_the_references_and_alias_the_flat_buffer(g1):
    """Row b: 14 keyed Parameters per module (g1_keys.json = the reference's named_parameters), OIHW-shaped views of ONE flat buffer
    on the device; .to() keeps the aliasing; writing through a Parameter is what the kernels read."""
    import json
    with open(os.path.join(REPO, "tests", "golden", "g1_keys.json")) as fp:
        keys = json.load(fp)["chfak1"]
    critic, masker = modules(g1)
    for mod, name in ((critic, "critic"), (masker, "masker")):
        assert [(k, list(q.shape)) for k, q in mod.named_parameters()] == [(k, v) for k, v in keys[name].items()]
        flat = mod.flat
        assert flat.is_cuda and not isinstance(flat, torch.nn.Parameter)
        for (k, q), seg in zip(mod.named_parameters(), mod.layout.segs.values()):
            assert q.is_cuda and q.data_ptr() == flat.data_ptr() + 4 * seg.offset and q.is_leaf and q.requires_grad
        for k, v in g1[0 if name == "critic" else 1].items():
            assert torch.equal(dict(mod.named_parameters())[k].detach().cpu(), v)
    X = torch.rand(4, 3, 64, 64, device="cuda")
    critic.eval()
    p0 = critic(X)
    with torch.no_grad():
        critic.crit[4].bias += 1.0                    # the reference's attribute path (nets.py:190-194)
    assert torch.allclose(critic(X), torch.sigmoid(torch.logit(p0) + 1.0), atol=1e-5)      # (crit.5 is a Sigmoid)
    with torch.no_grad():
        critic.features[0].weight.data = critic.features[0].weight.detach().clone() * 0.0      # storage replaced: re-aliased on use
    pz = critic(X)
    assert critic.features[0].weight.data_ptr() == critic.flat.data_ptr() and not torch.allclose(pz, torch.sigmoid(torch.logit(p0) + 1.0), atol=1e-5)
    assert float(critic.flat[:216].abs().max()) == 0.0


def test_freezing_one_key_freezes_exactly_that_layer(golden, g1):
    """requires_grad_(False) on one Parameter: no .grad for it, torch.optim.Adam leaves it alone, every other layer's gradient is what
    it was (the reference: any nn.Module parameter can be frozen by name, main.py:330-334 builds the optimiser from .parameters())."""
    g = golden("g3_train_default.npz")
    A, B = orc.u8_to_nchw(g["A"]).to("cuda"), orc.u8_to_nchw(g["B"]).to("cuda")
    Y = torch.from_numpy(g["Y"]).to("cuda")

    def grads(freeze):
        critic, masker = modules(g1)
        critic.train(); masker.train()
        if freeze:
            critic.features[6].weight.requires_grad_(False)
            masker.dec_model[2].bias.requires_grad_(False)
        opti = torch.optim.Adam([q for q in chain(critic.parameters(), masker.parameters()) if q.requires_grad])
        pred, embeds = critic(A, collect=True)
        Z = masker(A, embeds)
        loss = 5 * F.mse_loss(pred.squeeze(), Y) + F.mse_loss(critic(A * (1 - Z) + Z * B).squeeze(), critic(B).squeeze().detach()) + \
            0.5 * F.l1_loss(Z, torch.zeros_like(Z))
        loss.backward()
        out = {"c/" + k: (None if q.grad is None else q.grad.clone()) for k, q in critic.named_parameters()}
        out.update({"m/" + k: (None if q.grad is None else q.grad.clone()) for k, q in masker.named_parameters()})
        before = {"c/" + k: q.detach().clone() for k, q in critic.named_parameters()}
        before.update({"m/" + k: q.detach().clone() for k, q in masker.named_parameters()})
        opti.step()
        after = {"c/" + k: q.detach().clone() for k, q in critic.named_parameters()}
        after.update({"m/" + k: q.detach().clone() for k, q in masker.named_parameters()})
        return out, before, after

    full, _, _ = grads(False)
    part, before, after = grads(True)
    frozen = {"c/features.6.weight", "m/dec_model.2.bias"}
    for k in full:
        if k in frozen:
            assert part[k] is None and torch.equal(before[k], after[k])
        else:
            assert torch.equal(part[k], full[k]), k          # the same kernels ran: bit-identical
            assert not torch.equal(before[k], after[k]), k


def test_layout_conversion_kernels():
    from cgs_amd import _lib
    import ctypes as C
    x = torch.randn(5, 3, 64, 64, device="cuda")
    y = torch.empty(5, 64, 64, 3, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.call("cgs_nchw_to_nhwc", 5, 3, 4096, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), st)
    assert torch.equal(y, x.permute(0, 2, 3, 1).contiguous())
    z = torch.empty_like(x)
    _lib.call("cgs_nhwc_to_nchw", 5, 3, 4096, C.c_void_p(y.data_ptr()), C.c_void_p(z.data_ptr()), st)
    assert torch.equal(z, x)


def _write_dataset(root, n_train=1536, n_test=512):
    """Synthetic stand-in for the MineRL gz-pickle: bright frames carry high targets so
    that the critic can separate them and the >=500 hi/lo asserts of extract_contrastive_data hold."""
    rs = np.random.RandomState(0)
    n = n_train + n_test
    hi = rs.rand(n) > 0.5
    X = rs.randint(0, 96, (n, 64, 64, 3)).astype(np.uint8)
    X[hi] += 150
    Y = np.zeros((7, n))
    Y[:] = np.where(hi, 0.95, 0.05)[None] + rs.randn(7, n) * 0.01
    I = (np.arange(n) % 2 ** 16).astype(np.uint16)
    d = os.path.join(root, "runs", "data", "straight")
    os.makedirs(d, exist_ok=True)
    with gzip.GzipFile(os.path.join(d, f"Treechop-trunk-{n_train}-[0.98-0.97-0.96-0.95].pickle"), "wb") as fp:
        pickle.dump((X, Y, I), fp)
    return X, Y


def test_cli_train_then_process(tmp_path):
    """`main.py -train` on a synthetic gz-pickle, then `main.py -process` on a folder of PNGs: checkpoint names,
    reference-format checkpoints (loadable by the oracle), output file names / dtypes, and mask values equal
    to the oracle's run on the trained weights."""
    from PIL import Image
    root = str(tmp_path)
    X, Y = _write_dataset(root)
    common = ["--model", "m", "--datasize", "1536", "--testsize", "512", "--cepochs", "10", "--mepochs", "1"]
    r = subprocess.run([sys.executable, os.path.join(REPO, "main.py"), "-train"] + common, cwd=root,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    cpath = os.path.join(root, "m", "saves", "critic-rewidx=1-cepochs=10-datamode=trunk-datasize=1536-shift=12-chfak=1-dropout=0.3.pt")
    mpath = os.path.join(root, "m", "saves", "masker-mepochs=1-L1=0.5-inject=True.pt")
    assert os.path.exists(cpath) and os.path.exists(mpath), os.listdir(os.path.join(root, "m", "saves"))
    pc = torch.load(cpath, map_location="cpu")
    pm = torch.load(mpath, map_location="cpu")
    assert [(k, tuple(v.shape)) for k, v in pc.items()] == [(k, s) for k, s in orc.critic_shapes(1)]
    assert [(k, tuple(v.shape)) for k, v in pm.items()] == [(k, s) for k, s in orc.masker_shapes(1)]
    # the critic learnt the bright/dark split
    with torch.no_grad():
        p = orc.critic_apply(pc, orc.u8_to_nchw(X[:256])).squeeze(1).numpy()
    assert np.corrcoef(p, Y[1, :256])[0, 1] > 0.9
    # -process
    src = os.path.join(root, "in")
    os.makedirs(src)
    for i in range(5):
        Image.fromarray(X[i]).save(os.path.join(src, f"frame.{i}.png"))
    r = subprocess.run([sys.executable, os.path.join(REPO, "main.py"), "-process", "--source-imgs", "in",
                        "--mask-output-imgs", "out"] + common, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    names = sorted(os.listdir(os.path.join(root, "out")))
    assert names == sorted(f"frame.{i}-{c}.png" for i in range(5) for c in ("raw-mask", "thresholded-mask"))
    _, M = orc.infer_masks(pc, pm, X[:5] / 255.0)
    hard, stack = orc.postprocess_masks(X[:5] / 255.0, M, 0.5)
    for i in range(5):
        raw = np.array(Image.open(os.path.join(root, "out", f"frame.{i}-raw-mask.png")))
        thr = np.array(Image.open(os.path.join(root, "out", f"frame.{i}-thresholded-mask.png")))
        assert raw.shape == (64, 64, 3) and raw.dtype == np.uint8
        assert np.abs(raw.astype(int) - stack[i, 1].astype(int)).max() <= 1       # uint8 quantisation of a 1e-3 match
        assert (thr != stack[i, 2]).mean() < 2e-3 and set(np.unique(thr)) <= {0, 255}
    # -concatenated writes one side-by-side image per input
    r = subprocess.run([sys.executable, os.path.join(REPO, "main.py"), "-process", "-concatenated", "--source-imgs", "in",
                        "--mask-output-imgs", "out2"] + common, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    img = np.array(Image.open(os.path.join(root, "out2", "frame.0_with_mask.png")))
    assert img.shape == (64, 192, 3)
    np.testing.assert_array_equal(img[:, :64], X[0])
    # -salience -process_salience: two more images per frame, named by position in the reference's column list
    r = subprocess.run([sys.executable, os.path.join(REPO, "main.py"), "-process", "-salience", "-process_salience",
                        "--salience-thresh", "0.5", "--source-imgs", "in", "--mask-output-imgs", "out3"] + common, cwd=root,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    names = sorted(os.listdir(os.path.join(root, "out3")))
    assert names == sorted(f"frame.{i}-{c}.png" for i in range(5)
                           for c in ("raw-mask", "thresholded-mask", "crf-mask", "saliency-map"))
    Xc = orc.u8_to_nchw(X[:5]).clone().requires_grad_(True)
    pr = orc.critic_apply(pc, Xc)
    pr.mean().backward()
    raw_sal = Xc.grad.abs().sum(dim=1)[:, None].numpy()
    sal = raw_sal / ((raw_sal * (raw_sal >= 0)).mean() * 0.5 + sys.float_info.min) * pr.detach().numpy()[:, :, None, None]
    sal[sal >= 1] = 1
    got_sal = np.array(Image.open(os.path.join(root, "out3", "frame.2-crf-mask.png")))[..., 0].astype(int)
    assert np.abs(got_sal - (sal[2, 0] * 255).astype(np.uint8).astype(int)).max() <= 2
    # -eval: IoU on a (synthetic) red-trees set in the working directory, against the oracle on the trained weights
    os.makedirs(os.path.join(root, "red-trees"))
    rs = np.random.RandomState(11)
    Xe = rs.randint(0, 256, (420, 64, 64, 3)).astype(np.uint8)
    Ye = np.zeros((420, 64, 64, 3), dtype=bool)
    Ye[:, 16:48, 8:40] = True                       # "all channels set" = labelled object
    Ye[:, 20:30, 10:20, 1] = False                  # ... except where one channel is not
    np.save(os.path.join(root, "red-trees", "X.npy"), Xe)
    np.save(os.path.join(root, "red-trees", "Y.npy"), Ye)
    with torch.no_grad():                           # a threshold inside the range of the mask values
        _, M0 = orc.infer_masks(pc, pm, Xe[100:164:2] / 255.0)
    thr = float(0.5 * (np.median(M0) + M0.min()))
    r = subprocess.run([sys.executable, os.path.join(REPO, "main.py"), "-eval", "--eval-thresh", repr(thr)] + common, cwd=root,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    want = orc.eval_iou(pc, pm, Xe, Ye, eval_thresh=thr)
    got = float(r.stdout.split("RESULTS [")[-1].split("]")[0])
    assert 0.0 <= want <= 1.0 and abs(got - want) <= 2e-3, (got, want)
    # -eval -salience: the saliency baseline's IoU next to the mask's
    r = subprocess.run([sys.executable, os.path.join(REPO, "main.py"), "-eval", "-salience", "--eval-thresh", repr(thr),
                        "--salience-thresh", "0.5"] + common, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    got2 = [float(v) for v in r.stdout.split("RESULTS [")[-1].split("]")[0].split(",")]
    want_sal, _, _ = orc.eval_saliency_iou(pc, Xe, Ye, salience_thresh=0.5)
    assert len(got2) == 2 and abs(got2[0] - want) <= 2e-3
    assert (np.isnan(got2[1]) and np.isnan(want_sal)) or abs(got2[1] - want_sal) <= 5e-3, (got2, want_sal)


def test_cli_process_matches_the_reference_cli_capture(tmp_path, golden, g1):
    """`main.py -process` of THIS build against G6 = the PNG files the reference's own `main.py -process` wrote for the same frames and
    G1 weights (tests/golden/make_golden_g6.py): identical file names for the three command lines, raw masks within one uint8 step of
    the reference's (the HIP fp32 mask differs from the CPU one by ~1e-6, (Z * 255).astype(uint8) truncates), thresholded masks equal
    except where Z is within 1e-5 of the threshold, the frame column of the concatenated strips exact."""
    import json
    from PIL import Image
    pc, pm = g1
    g = golden("g6_process.npz")
    X, names = g["frames"], [str(s) for s in g["names"]]
    listing = json.loads(str(g["listing_json"]))
    root = str(tmp_path)
    os.makedirs(os.path.join(root, "src"))
    for nm, x in zip(names, X):
        Image.fromarray(x).save(os.path.join(root, "src", nm + ".png"))
    cnames = [str(s) for s in g["checkpoint_names"]]
    os.makedirs(os.path.join(root, "m", "saves"))
    torch.save(pc, os.path.join(root, cnames[0]))
    torch.save(pm, os.path.join(root, cnames[1]))
    runs = {"default": [], "concat": ["-concatenated"], "thr052": ["--binarymaskthreshold", "0.52"]}
    _, Zref = orc.infer_masks(pc, pm, X / 255.0)
    for tag, extra in runs.items():
        r = subprocess.run([sys.executable, os.path.join(REPO, "main.py"), "-process", "--model", "m", "--source-imgs", "src",
                            "--mask-output-imgs", "out_" + tag] + extra, cwd=root, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        assert sorted(os.listdir(os.path.join(root, "out_" + tag))) == listing[tag], tag
        thr = 0.52 if tag == "thr052" else 0.5
        for i, f in enumerate(listing[tag]):
            got = np.array(Image.open(os.path.join(root, "out_" + tag, f))).astype(int)
            ref = g[f"{tag}/{f}"].astype(int)
            assert got.shape == ref.shape
            if f.endswith("raw-mask.png"):
                assert np.abs(got - ref).max() <= 1 and (got != ref).mean() < 2e-3, f
            elif f.endswith("thresholded-mask.png"):
                nm = f[:-len("-thresholded-mask.png")]
                near = np.abs(Zref[names.index(nm), 0] - thr) < 1e-5
                assert not ((got[..., 0] != ref[..., 0]) & ~near).any(), f
            else:       # frame | raw mask | thresholded mask
                np.testing.assert_array_equal(got[:, :64], ref[:, :64])
                assert np.abs(got[:, 64:128] - ref[:, 64:128]).max() <= 1


def test_cli_process_fp16_path_against_the_reference_capture(tmp_path, golden, g1):
    """`main.py -process -fp16` (this build's own switch: the fused fp16 inference path of BASELINE config 4) on the G6 frames: the same
    files as the reference's CLI, raw masks within 2 uint8 steps (|dZ| of the fp16 path is ~4e-5), thresholded masks equal except where
    the fp32 mask is within 2e-4 of the threshold."""
    import json
    from PIL import Image
    pc, pm = g1
    g = golden("g6_process.npz")
    X, names = g["frames"], [str(s) for s in g["names"]]
    listing = json.loads(str(g["listing_json"]))
    root = str(tmp_path)
    os.makedirs(os.path.join(root, "src"))
    for nm, x in zip(names, X):
        Image.fromarray(x).save(os.path.join(root, "src", nm + ".png"))
    cnames = [str(s) for s in g["checkpoint_names"]]
    os.makedirs(os.path.join(root, "m", "saves"))
    torch.save(pc, os.path.join(root, cnames[0]))
    torch.save(pm, os.path.join(root, cnames[1]))
    r = subprocess.run([sys.executable, os.path.join(REPO, "main.py"), "-process", "-fp16", "--model", "m", "--source-imgs", "src",
                        "--mask-output-imgs", "out"], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert sorted(os.listdir(os.path.join(root, "out"))) == listing["default"]
    _, Zref = orc.infer_masks(pc, pm, X / 255.0)
    for f in listing["default"]:
        got = np.array(Image.open(os.path.join(root, "out", f))).astype(int)
        ref = g[f"default/{f}"].astype(int)
        if f.endswith("raw-mask.png"):
            assert np.abs(got - ref).max() <= 2, f
        else:
            nm = f[:-len("-thresholded-mask.png")]
            near = np.abs(Zref[names.index(nm), 0] - 0.5) < 2e-4
            assert not ((got[..., 0] != ref[..., 0]) & ~near).any(), f


DP_WORKER = r"""
import os, sys
sys.path.insert(0, {repo!r})
import numpy as np, torch
import torch.distributed as dist
import cgs_amd
from cgs_amd import parallel, engine
pg = parallel.init_from_env({backend!r})     # gloo: 2 ranks share the one GPU of the box; nccl (= RCCL): one GPU per rank
rank, _, world = parallel.env_world()
raw = dict(np.load(os.path.join({repo!r}, "tests", "golden", "g1_weights_chfak1.npz")))
pc = {{k.split("/", 1)[1]: torch.from_numpy(v) for k, v in raw.items() if k.startswith("critic/")}}
pm = {{k.split("/", 1)[1]: torch.from_numpy(v) for k, v in raw.items() if k.startswith("masker/")}}
rs = np.random.RandomState(3)
n = 32
A = torch.from_numpy(rs.randint(0, 256, (n, 64, 64, 3)).astype(np.uint8)).cuda()
B = torch.from_numpy(rs.randint(0, 256, (n, 64, 64, 3)).astype(np.uint8)).cuda()
Y = torch.from_numpy(rs.rand(n).astype(np.float32)).cuda()
sl = parallel.shard_slice(n, rank, world)
e = engine.HourglassEngine(n // world, dropout=0.0, process_group=pg)
e.load_state(pc, pm)
for _ in range(3):
    e.phase2_step(A[sl], B[sl], Y[sl])
torch.cuda.synchronize()
flat = e.flat.cpu()
others = [torch.empty_like(flat) for _ in range(world)]
dist.all_gather(others, flat)
assert all(torch.equal(o, others[0]) for o in others), "replicas diverged"
# every rank draws its own dropout stream (the global batch then holds independent masks)
seeds = [None] * world
dist.all_gather_object(seeds, engine.HourglassEngine(2, dropout=0.3, process_group=pg).drop.seed)
assert len(set(seeds)) == world, seeds
if rank == 0:
    np.save({out!r}, flat.numpy())
dist.barrier(); dist.destroy_process_group()
"""


def test_data_parallel_engine_world2(tmp_path, g1):
    """Two ranks (gloo rendezvous, both on this box's GPU), half the batch each, 3 steps: replicas stay
    bit-identical and match the single-process full-batch run."""
    out = str(tmp_path / "dp_flat.npy")
    _run_dp_workers(tmp_path, out, "gloo")
    _check_dp_result(out, g1)


def _run_dp_workers(tmp_path, out, backend):
    script = tmp_path / f"dp_worker_{backend}.py"
    script.write_text(DP_WORKER.format(repo=REPO, out=out, backend=backend))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29531" if backend == "gloo" else "29532", str(script)],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]


def _check_dp_result(out, g1):
    from cgs_amd import engine
    pc, pm = g1
    rs = np.random.RandomState(3)
    n = 32
    A = torch.from_numpy(rs.randint(0, 256, (n, 64, 64, 3)).astype(np.uint8)).cuda()
    B = torch.from_numpy(rs.randint(0, 256, (n, 64, 64, 3)).astype(np.uint8)).cuda()
    Y = torch.from_numpy(rs.rand(n).astype(np.float32)).cuda()
    e = engine.HourglassEngine(n, dropout=0.0)
    e.load_state(pc, pm)
    for _ in range(3):
        e.phase2_step(A, B, Y)
    rel_close(np.load(out), e.flat.cpu().numpy(), "DP(2) parameters vs single process", rtol=1e-3, atol_scale=1e-4)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")
def test_data_parallel_engine_world2_rccl(tmp_path, g1):
    """The same two-rank run over backend "nccl" (= RCCL over xGMI), one GPU per rank: the production collective path."""
    out = str(tmp_path / "dp_flat_rccl.npy")
    _run_dp_workers(tmp_path, out, "nccl")
    _check_dp_result(out, g1)


DP_FORM_WORKER = r"""
import os, sys
sys.path.insert(0, {repo!r})
import numpy as np, torch
import torch.distributed as dist
import cgs_amd
from cgs_amd import engine
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
pg = dist.group.WORLD
raw = dict(np.load(os.path.join({repo!r}, "tests", "golden", "g1_weights_chfak1.npz")))
pc = {{k.split("/", 1)[1]: torch.from_numpy(v) for k, v in raw.items() if k.startswith("critic/")}}
pm = {{k.split("/", 1)[1]: torch.from_numpy(v) for k, v in raw.items() if k.startswith("masker/")}}
rs = np.random.RandomState(5)
n, steps = 24, 4
A = torch.from_numpy(rs.randint(0, 256, (n, 64, 64, 3)).astype(np.uint8)).cuda()
B = torch.from_numpy(rs.randint(0, 256, (n, 64, 64, 3)).astype(np.uint8)).cuda()
Y = torch.from_numpy(rs.rand(n).astype(np.float32)).cuda()
res = {{}}
for live in (True, False):
    runs = {{}}
    for name, kw in (("single", dict()), ("dp_graph", dict(process_group=pg, force_allreduce=True, dp_graph=True)),
                     ("dp_eager", dict(process_group=pg, force_allreduce=True, dp_graph=False))):
        e = engine.HourglassEngine(n, dropout=0.3, live=live, **kw)
        e.load_state(pc, pm)
        for _ in range(steps):
            e.phase2_step(A, B, Y)
        torch.cuda.synchronize()
        runs[name] = dict(flat=e.flat.cpu().numpy(), m=e.m.cpu().numpy(), v=e.v.cpu().numpy(), t=int(e.step_t.item()),
                          losses=e.losses.cpu().numpy(), single_graph=bool(e.dp_single_graph), note=str(e.dp_capture_note))
    res[live] = runs
np.save({out!r}, np.array([res], dtype=object), allow_pickle=True)
dist.destroy_process_group()
"""


def test_dp_launch_form_matches_single_gpu_step(tmp_path):
    """The data-parallel launch form (fused tail losses -> cgs_reduce_adam(param = NULL): reduction + loss values + step tick ->
    all-reduce -> cgs_adam_flat) on a ONE-rank RCCL group, with the collective inside the step graph and outside it, against the
    single-GPU fused step (Adam inside cgs_reduce_adam): parameters, Adam moments and the step counter after 4 steps, live and
    frozen critic (the frozen case exercises the optimiser range that excludes the critic).  The two Adam kernels compute the
    bias correction differently (expm1f vs pow), hence a tolerance, not bit equality."""
    out = str(tmp_path / "dp_form.npy")
    script = tmp_path / "dp_form_worker.py"
    script.write_text(DP_FORM_WORKER.format(repo=REPO, out=out))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29536", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[:6000] + " ... " + r.stderr[-1500:]
    res = np.load(out, allow_pickle=True)[0]
    for live, runs in res.items():
        base = runs["single"]
        assert base["t"] == 4
        print(f"live={live}: all-reduce in the step graph: {runs['dp_graph']['single_graph']} ({runs['dp_graph']['note']})")
        assert not runs["dp_eager"]["single_graph"]
        for name in ("dp_graph", "dp_eager"):
            got = runs[name]
            assert got["t"] == base["t"], (name, got["t"])
            np.testing.assert_allclose(got["losses"][:6], base["losses"][:6], rtol=1e-5, atol=1e-7)
            for k, sc in (("flat", 1e-5), ("m", 1e-5), ("v", 1e-5)):
                rel_close(got[k], base[k], f"{name} live={live} {k} vs the single-GPU fused step", rtol=1e-4, atol_scale=sc)
        if not live:      # frozen critic: its parameters and moments must not have moved at all
            from cgs_amd import spec
            nc = spec.critic_layout().total
            for name in ("single", "dp_graph", "dp_eager"):
                assert not runs[name]["m"][:nc].any() and not runs[name]["v"][:nc].any(), name
            np.testing.assert_array_equal(runs["dp_graph"]["flat"][:nc], base["flat"][:nc])


def _check_dp_report(line, forms):
    """Schema of the data-parallel self-validation block of a bench.py line (config.dp)."""
    cfg = line["config"]
    dp = cfg["dp"]
    assert set(dp["forms_timed"]) == forms, dp["forms_timed"].keys()
    for f, v in dp["forms_timed"].items():
        assert v["ms_per_step"] > 0 and len(v["per_rank_ms_per_step"]) == line["n_gpus"] and v["replicas_bit_identical"] is True, (f, v)
        assert np.isfinite(v["final_total_loss_rank0"])
    assert dp["value_uses_form"] in forms and cfg["allreduce_in_step_graph"] is (dp["value_uses_form"] == "graph")
    assert abs(line["ms_per_step"] - dp["forms_timed"][dp["value_uses_form"]]["ms_per_step"]) < 1e-9
    assert line["ms_per_step"] == min(v["ms_per_step"] for v in dp["forms_timed"].values())
    assert cfg["replicas_bit_identical"] is True and len(dp["replica_checksums_params_m_v_step"]) == line["n_gpus"]
    assert len(set(dp["replica_checksums_params_m_v_step"])) == 1 and len(dp["replica_checksums_params_m_v_step"][0]) == 8 * 16
    lat = dp["allreduce_latency_us"]
    assert lat["bytes"] == 4 * 25661 and lat["reps"] == 200 and lat["eager_median"] > 0      # (11 873 + 3 pad + 13 785 floats: the flat bucket)
    assert abs(line["value"] - 512 * line["n_gpus"] / (line["ms_per_step"] * 1e-3)) < 1e-3 * line["value"]


def test_bench_world2_gloo_rehearsal_on_one_gpu():
    """`bench.py --gpus 2 --backend gloo`: TWO ranks (started by bench.py's own child torchrun) share this box's GPU and reduce the gradient
    bucket through the host -- every world > 1 branch of the script runs (per-rank timing gathered by a collective, max over ranks, the
    replica checksums over two REAL replicas with different Dropout streams and different data, the all-reduce latency probe); gloo cannot
    be recorded in a HIP graph, so only the eager form is timed and the line says why.  Numbers are meaningless (shared GPU); the schema
    and the bit-identity of the two replicas are the point."""
    plain = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    plain["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "6", "--warmup", "2",
                        "--prime-s", "0.05", "--no-cpu-baseline"], capture_output=True, text=True, env=plain, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 alone prints the line"
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["ranks_seen_by_collective_backend"] == 2 and line["config"]["collective_backend"] == "gloo"
    assert line["config"]["global_batch"] == 1024 and line["config"]["parallelism"] == "dp2" and len(line["config"]["per_rank_ms_per_step"]) == 2
    _check_dp_report(line, {"eager"})
    dp = line["config"]["dp"]
    assert dp["collective_capturable"] is False and "gloo" in dp["collective_capturable_note"] and dp["allreduce_latency_us"]["graph_median"] is None


def test_rccl_path_one_rank_rehearsal():
    """bench.py with a ONE-rank RCCL group and the data-parallel launch form forced (step graph -> in-place device all-reduce
    -> Adam graph): init_process_group("nccl", device_id=...), the collective on the device buffer between the two HIP graphs,
    the barrier and the max-over-ranks timing all execute on this 1-GPU box; its cost shows as the step-time difference."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0",
               MASTER_ADDR="127.0.0.1", MASTER_PORT="29534", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "200", "--warmup", "20", "--no-cpu-baseline"]
    plain = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}

    def run(args, e):
        r = subprocess.run(cmd + args, capture_output=True, text=True, env=e, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])

    # two runs of each form, interleaved, the faster one counts: the difference of two 0.6 ms steps is compared with 40 us while
    # the clocks of a fresh box drift by a few per cent between processes
    pg, single = [], []
    for _ in range(2):
        pg.append(run(["--force-pg"], env))
        single.append(run([], plain))
    line = min(pg, key=lambda d: d["ms_per_step"])
    base = min(single, key=lambda d: d["ms_per_step"])
    assert line["config"]["collective_backend"] == "nccl" and line["config"]["ranks_seen_by_collective_backend"] == 1
    assert np.isfinite(line["final_losses"]["total"]) and line["n_gpus"] == 1
    # the contract's schema on the data-parallel line, and the launch form: on a 1-rank group the default records the collective in the
    # step graph (parallel.resolve_dp_graph), after a trial capture + replay agreed on by every rank
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline"):
        assert key in line, key
    assert line["scaling"] == "weak" and line["dtype"] == "f32" and line["roofline"]["bound"] == "hbm"
    assert "rccl" in line["config"] and line["config"]["rccl"]["version"], "bench.py --gpus N must report RCCL's view"
    # round 6: the data-parallel run is self-validating -- BOTH launch forms timed (the collective trial succeeded on the RCCL group), the
    # replicas' parameters / Adam moments / step counter compared through all-gathered checksums after each, `value` on the faster good one,
    # and the bare latency of the gradient bucket's all-reduce reported in both forms
    _check_dp_report(line, {"eager", "graph"})
    dp = line["config"]["dp"]
    assert dp["collective_capturable"] is True and dp["allreduce_latency_us"]["graph_median"] > 0
    print("dp forms:", {f: round(v["ms_per_step"], 4) for f, v in dp["forms_timed"].items()}, "value uses", dp["value_uses_form"],
          "| all-reduce of", dp["allreduce_latency_us"]["bytes"], "B: eager", round(dp["allreduce_latency_us"]["eager_median"], 1), "us, graph",
          round(dp["allreduce_latency_us"]["graph_median"], 1), "us")
    eager = run(["--force-pg", "--dp-eager-allreduce", "--steps", "5", "--warmup", "2"], env)
    assert eager["config"]["allreduce_in_step_graph"] is False and np.isfinite(eager["final_losses"]["total"])
    _check_dp_report(eager, {"eager"})
    assert eager["config"]["dp"]["choice"] == "forced on the command line"
    graph = run(["--force-pg", "--dp-graph-allreduce", "--steps", "5", "--warmup", "2"], env)
    assert graph["config"]["allreduce_in_step_graph"] is True
    _check_dp_report(graph, {"graph"})
    extra_ms = line["ms_per_step"] - base["ms_per_step"]
    print(f"three-launch form with a 1-rank RCCL all-reduce: {line['ms_per_step']:.3f} ms vs {base['ms_per_step']:.3f} ms single graph "
          f"(+{extra_ms * 1e3:.0f} us per step)")
    # round 3: the data-parallel step = the single-GPU step + ONE Adam launch + the collective (the loss / reduction tail stays fused)
    # a REPORTED number (two separate processes on a box whose clocks drift by a few per cent): only a gross regression fails
    assert extra_ms < 0.10, "the data-parallel launch form must cost at most tens of microseconds over the single-GPU step"


def test_bench_default_line_carries_the_side_block():
    """The driver's command (`python bench.py --gpus 1 --steps K --warmup W`) prints ONE JSON line whose extra key "side" holds the other
    BASELINE configurations, each measured by the child process after the headline's timed region: config 4 (fp16 inference, batch 2048),
    config 5 (128x128 bf16: training step and inference, batch 256), the paper's model size (chfak 5 training step)."""
    plain = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--cpu-steps", "1"],
                       capture_output=True, text=True, env=plain, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line on stdout"
    d = json.loads(lines[0])
    assert d["metric"].startswith("Hourglass+critic train images/sec") and d["dtype"] == "f32" and "cpu_baseline" in d and "roofline" in d
    side = d["side"]
    assert "error" not in side, side
    assert set(side) == {"config4_fp16_infer_batch2048", "config5_train_batch256", "config5_infer_batch256", "chfak5_train_batch512"}
    for key, line in side.items():
        assert "error" not in line, (key, line)
        assert line["value"] > 0 and line["unit"] == "images/s" and line["steps"] >= 20 and "roofline" in line and line["argv"]
        print(f"side {key}: {line['value'] / 1e3:.0f} k images/s, {line['ms_per_step']:.3f} ms, frac {line['roofline']['frac']:.3f} ({line['roofline']['bound']})")
    assert side["config4_fp16_infer_batch2048"]["config"]["batch"] == 2048 and side["config4_fp16_infer_batch2048"]["dtype"].startswith("f16")
    assert side["config5_train_batch256"]["config"]["batch"] == 256 and side["config5_train_batch256"]["dtype"].startswith("bf16")


def test_bench_gpus_flag_starts_ranks_or_refuses():
    """`python bench.py --gpus N` without a torchrun environment starts N ranks itself (child torchrun, the parent makes no GPU
    call) or exits non-zero when fewer than N GPUs are visible -- it must never silently run on one GPU."""
    plain = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    want = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(want), "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, env=plain, timeout=300)
    assert r.returncode != 0 and "GPU(s) are visible" in r.stderr


def test_extract_contrastive_split_matches_oracle(tmp_path, g1, monkeypatch):
    """Handler.extract_contrastive_data (main.py:238-312): the high / low split of the frames equals thresholding the CPU
    oracle's eval-mode critic values; the device-resident copies hold exactly those frames; the sampler draws in range."""
    from cgs_amd import cli, handler
    monkeypatch.chdir(tmp_path)
    pc, pm = g1
    rs = np.random.RandomState(5)
    n = 2000
    X = rs.randint(0, 256, (n, 64, 64, 3)).astype(np.uint8)
    X[: n // 2] = (X[: n // 2] * 0.3).astype(np.uint8)            # two brightness populations: a spread of critic values
    Y = rs.rand(7, n)
    with torch.no_grad():
        want = torch.cat([orc.critic_apply(pc, orc.u8_to_nchw(X[b:b + 250])).squeeze(1) for b in range(0, n, 250)]).numpy()
    hi, lo = float(np.quantile(want, 0.6)), float(np.quantile(want, 0.4))
    args = cli.parse_args(["--model", "m", "--high-rew-thresh", repr(hi), "--low-rew-thresh", repr(lo)])
    H = handler.Handler(args)
    H.critic.load_state_dict(pc)
    H.X, H.Y = X, Y
    H.extract_contrastive_data()
    border = (np.abs(want - hi) < 1e-5) | (np.abs(want - lo) < 1e-5)       # frames whose value sits on a threshold
    pos, neg = want > hi, want < lo
    got_pos = np.zeros(n, bool); got_neg = np.zeros(n, bool)
    # recover the split from the stored frames' targets (Y columns are unique)
    col = {tuple(np.round(Y[:, i], 12)): i for i in range(n)}
    for j in range(H.Ypos.shape[1]):
        got_pos[col[tuple(np.round(H.Ypos[:, j], 12))]] = True
    for j in range(H.Yneg.shape[1]):
        got_neg[col[tuple(np.round(H.Yneg[:, j], 12))]] = True
    assert (got_pos == pos)[~border].all() and (got_neg == neg)[~border].all()
    np.testing.assert_array_equal(H._Xpos_d.cpu().numpy(), H.Xpos)
    np.testing.assert_array_equal(H._Xneg_d.cpu().numpy(), H.Xneg)
    np.testing.assert_allclose(H._ypos_d.cpu().numpy(), H.Ypos[args.rewidx].astype(np.float32))
    Hi, Li, Ci = H.get_contrastive_idxs()
    assert len(Hi) == 32 and len(Li) == 32 and len(Ci) == 64 and Hi.max() < len(H.Xpos) and max(Li.max(), Ci.max()) < len(H.Xneg)


def test_mask_training_uses_one_fresh_index_draw_per_step(tmp_path, g1, monkeypatch):
    """Handler.segmentation_training (main.py:344-356): the host runs ahead of the device (one sync per 10 steps), so the
    per-step index upload must not be overwritten before it has been copied: the B batch the device assembled in EVERY step
    equals Xneg[the host's draw of that step] (B is not rolled), and consecutive steps differ."""
    from cgs_amd import cli, engine, handler
    monkeypatch.chdir(tmp_path)
    pc, pm = g1
    rs = np.random.RandomState(7)
    n = 3000
    X = rs.randint(0, 256, (n, 64, 64, 3)).astype(np.uint8)
    X[: n // 2] = (X[: n // 2] * 0.3).astype(np.uint8)
    Y = rs.rand(7, n)
    with torch.no_grad():
        want = torch.cat([orc.critic_apply(pc, orc.u8_to_nchw(X[b:b + 250])).squeeze(1) for b in range(0, n, 250)]).numpy()
    hi, lo = float(np.quantile(want, 0.6)), float(np.quantile(want, 0.4))
    args = cli.parse_args(["--model", "m", "--high-rew-thresh", repr(hi), "--low-rew-thresh", repr(lo), "--mepochs", "1"])
    H = handler.Handler(args)
    H.critic.load_state_dict(pc); H.masker.load_state_dict(pm)
    H.X, H.Y = X, Y
    H._record_idx_draws = True
    seen = []
    real = engine.HourglassEngine.phase2_step

    def spy(self, *a, **k):
        seen.append(self.ab[:self.n].clone())       # B of this step, as assembled on the device (stream-ordered copy)
        return real(self, *a, **k)
    monkeypatch.setattr(engine.HourglassEngine, "phase2_step", spy)
    H.segmentation_training()
    torch.cuda.synchronize()
    draws = H._last_idx_draws
    assert len(seen) == len(draws) >= 30
    nb = 2 * H.contrastive_batchsize
    Xneg = torch.from_numpy(H.Xneg)
    same_as_prev = 0
    for k, (b_dev, idx) in enumerate(zip(seen, draws)):
        want_b = Xneg[idx[nb:2 * nb]]
        assert torch.equal(b_dev.cpu(), want_b), f"step {k}: the device batch is not the host's draw of that step"
        if k and torch.equal(draws[k], draws[k - 1]):
            same_as_prev += 1
    assert same_as_prev == 0
