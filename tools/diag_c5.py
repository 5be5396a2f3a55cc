import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
import test_gpu_generic_train as T
from oracle import hourglass_ref as orc
chfak, neck = 5, 32
for n in (32, 128):
    rs = np.random.RandomState(11)
    dev = torch.device("cuda:0")
    A = rs.randint(0, 256, (n, 64, 64, 3)).astype(np.uint8)
    B = rs.randint(0, 256, (n, 64, 64, 3)).astype(np.uint8)
    Y = rs.rand(n).astype(np.float32)
    e, pc, pm = T.make_generic_engine(chfak, n, neck=neck, dropout=0.3, use_graph=True)
    masks = T._export_masks(e, 4 * n, chfak, neck)
    e.phase2_step(torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev), torch.from_numpy(Y).to(dev))
    sl = {"B": slice(0, n), "A": slice(n, 2 * n), "rep": slice(2 * n, 3 * n), "inj": slice(3 * n, 4 * n)}
    om = [[m[sl[k]] for m in masks] for k in ("A", "B", "rep", "inj")]
    torch.set_num_threads(16)
    dd = lambda P: {k: v.double() for k, v in P.items()}
    r64 = orc.train_phase2(dd(pc), dd(pm), [(orc.u8_to_nchw(A).double(), orc.u8_to_nchw(B).double(), torch.from_numpy(Y).double())], steps=1, p=0.3, training=True, masks=[[m.double() for m in mm] for mm in om])[0]
    r32 = orc.train_phase2(pc, pm, [(orc.u8_to_nchw(A), orc.u8_to_nchw(B), torch.from_numpy(Y))], steps=1, p=0.3, training=True, masks=om)[0]
    gc, gm = e.lc.unflatten(e.gc), e.lm.unflatten(e.gm)
    for name, G, key in (("c", gc, "grads_c"), ("m", gm, "grads_m")):
        for k, v in r64[key].items():
            ref = v.numpy(); mx = np.abs(ref).max()
            eg = np.abs(G[k].cpu().numpy().astype(np.float64) - ref)
            ec = np.abs(r32[key][k].numpy().astype(np.float64) - ref)
            print(f"n={n} {name} {k:28s} max {mx:.3e}  hip err max/mx {eg.max()/mx:.2e} rms/mx {np.sqrt((eg**2).mean())/mx:.2e} | cpu-fp32 err max/mx {ec.max()/mx:.2e} rms/mx {np.sqrt((ec**2).mean())/mx:.2e}", flush=True)
