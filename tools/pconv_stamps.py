import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench
from cgs_amd import _lib, engine
dev = torch.device("cuda:0"); n = 512
eng = engine.HourglassEngine(n, device=dev, dropout=0.3, use_graph=False); eng.load_state(*bench.g1_weights())
A, B, Y = bench.synthetic(n, 0, dev)
for _ in range(2): eng.phase2_step(A, B, Y)
torch.cuda.synchronize()
buf = torch.zeros(2048 * 8, dtype=torch.int64, device=dev)
lib = _lib.load(); lib.dbg_pconv_stamps.argtypes = [C.c_void_p]
lib.dbg_pconv_stamps(C.c_void_p(buf.data_ptr()))
eng.phase2_step(); torch.cuda.synchronize()
lib.dbg_pconv_stamps(C.c_void_p(0))
s = buf.cpu().numpy().reshape(2048, 8).astype(np.float64)
s = s[s[:, 0] != 0]
print("last pconv launch (enc1 of the mixes):", len(s), "workgroups; mean ticks per stage", np.round(np.diff(s[:, :5], axis=1).mean(0)).tolist())
