"""Times cgs_mask_train_fwd (the one-kernel mask head forward) at N = 512: tools/time_maskfwd.py"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cgs_amd import _lib
dev = torch.device("cuda:0")
P = lambda t: C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
x = torch.randint(0, 256, (n, 64, 64, 3), dtype=torch.uint8, device=dev)
o0 = torch.randn(n, 32, 32, 8, device=dev)
w0, b0, w2, b2 = torch.randn(9 * 11 * 16, device=dev) * 0.1, torch.randn(16, device=dev), torch.randn(144, device=dev) * 0.1, torch.randn(1, device=dev)
h, z, zp = torch.empty(n, 64, 64, 16, device=dev), torch.empty(n, 64, 64, device=dev), torch.empty(n, 2, device=dev)
call = lambda: _lib.call("cgs_mask_train_fwd", n, _lib.SRC_U8, P(x), P(o0), P(w0), P(b0), P(w2), P(b2), P(h), P(z), P(zp), st())
for _ in range(20): call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100): call()
e1.record(); torch.cuda.synchronize()
print(f"mask_train_fwd n={n}: {e0.elapsed_time(e1) / 100 * 1e3:7.1f} us")
