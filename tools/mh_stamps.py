# needs the debug build: python tools/build_variant.py stamps -DCGS_DEBUG_STAMPS ; CGS_LIB_PATH=<pkg>/libcgs_hip_stamps.so
"""Where a mask_head workgroup's cycles go (debug hook dbg_mask_head_stamps): per role the cycles summed over its tiles between the two
workgroup barriers of a tile.  Usage (GPU box): python tools/mh_stamps.py [batch]"""
import ctypes as C, os, sys
import numpy as np, torch
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
buf = torch.zeros(256 * 16, dtype=torch.int64, device=dev)
lib = _lib.load()
lib.dbg_mask_head_stamps.argtypes = [C.c_void_p]
lib.dbg_mask_head_stamps(C.c_void_p(buf.data_ptr()))
eng.phase2_step(); torch.cuda.synchronize()
lib.dbg_mask_head_stamps(C.c_void_p(0))
s = buf.cpu().numpy().reshape(256, 16).astype(np.float64)
T = s[:, 4]
print(f"{len(s)} workgroups, tiles per workgroup {T.min():.0f}..{T.max():.0f}; per TILE (cycles of s_memtime), mean over workgroups:")
for k, name in enumerate(("builder phase 1 (dz tile -> LDS)", "builder wait at barrier 1", "builder phase 2 (rebuild dH)", "builder wait at barrier 2")):
    print(f"   {name:42s} {np.mean(s[:, k] / (T + 1)):9.0f}")
for k, name in enumerate(("matrix phase 1 (commit inputs)", "matrix wait at barrier 1", "matrix data-gradient GEMM (+fetch)", "matrix weight-gradient GEMM", "matrix wait at barrier 2")):
    print(f"   {name:42s} {np.mean(s[:, 8 + k] / (T + 1)):9.0f}")
print(f"   prologue (table + halos) {np.mean(s[:, 6] - s[:, 5]):.0f}, roles {np.mean(s[:, 13] - s[:, 6]):.0f}, epilogue {np.mean(s[:, 7] - s[:, 13]):.0f}, whole workgroup {np.mean(s[:, 7] - s[:, 5]):.0f}; "
      f"kernel span {s[:, 7].max() - s[:, 5].min():.0f}")
