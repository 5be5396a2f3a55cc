# needs the debug build of the library: python tools/build_variant.py stamps -DCGS_DEBUG_STAMPS ; CGS_LIB_PATH=<pkg>/libcgs_hip_stamps.so (the product library exports no dbg_* hooks)
"""Stage timing of the tail kernels (debug hook dbg_tail_stamps): mean s_memtime deltas between the stage boundaries of
every workgroup's first image, per kernel.  Usage (GPU box): python tools/tail_stamps.py [batch]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from cgs_amd import _lib, engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda:0")
eng = engine.HourglassEngine(n, device=dev, dropout=0.3, use_graph=False)
eng.load_state(*bench.g1_weights())
A, B, Y = bench.synthetic(n, 0, dev)
for _ in range(3):
    eng.phase2_step(A, B, Y)
torch.cuda.synchronize()
buf = torch.zeros(4 * 2048 * 16, dtype=torch.int64, device=dev)
lib = _lib.load()
if not hasattr(lib, "dbg_tail_stamps"):
    raise SystemExit("stamps need the debug build: python tools/build_variant.py stamps -DCGS_DEBUG_STAMPS, then CGS_LIB_PATH=<pkg>/libcgs_hip_stamps.so")
lib.dbg_tail_stamps.argtypes = [C.c_void_p]
lib.dbg_tail_stamps(C.c_void_p(buf.data_ptr()))
eng.phase2_step()
torch.cuda.synchronize()
lib.dbg_tail_stamps(C.c_void_p(0))
st = buf.cpu().numpy().reshape(4, 2048, 16)
for k, name in enumerate(("enc_fwd (last call: mixes)", "dec_fwd", "enc_bwd (last call: A pass)", "dec_bwd")):
    s = st[k]
    live = s[:, 0] != 0
    s = s[live].astype(np.float64)
    if not len(s):
        continue
    cols = [c for c in range(16) if (s[:, c] != 0).all()]
    cols.sort(key=lambda c: s[:, c].mean())
    d = np.diff(s[:, cols], axis=1)
    print(f"{name}: {live.sum()} workgroups, stamps {cols}")
    print("   mean ticks per stage:", np.round(d.mean(0), 0).tolist(), " total", round(float((s[:, cols[-1]] - s[:, cols[0]]).mean())))
    print("   span (first workgroup's first stamp .. last workgroup's last stamp):", float(s[:, cols[-1]].max() - s[:, cols[0]].min()))
    if (s[:, 14] != 0).all() and 14 not in cols[1:-1]:
        print("   kernel entry (stamp 14) -> first stage stamp: mean", round(float((s[:, cols[1] if cols[0] == 14 else cols[0]] - s[:, 14]).mean())),
              " entry -> end: mean", round(float((s[:, 15] - s[:, 14]).mean())), " max", float((s[:, 15] - s[:, 14]).max()),
              " launch span", float(s[:, 15].max() - s[:, 14].min()))
