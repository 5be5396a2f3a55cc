# needs the debug build of the library: python tools/build_variant.py stamps -DCGS_DEBUG_STAMPS ; CGS_LIB_PATH=<pkg>/libcgs_hip_stamps.so (the product library exports no dbg_* hooks)
"""Stage timing of the fp16 conv kernel (debug hook dbg_gen16_stamps): python tools/gen16_stamps.py"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cgs_amd import _lib, generic as gen
dev = torch.device("cuda:0")
lib = _lib.load()
lib.dbg_gen16_stamps.argtypes = [C.c_void_p]
n = 2048
for name, hw, ca, cb, co, u8, pool, f32o in (("features.0", 64, 3, 0, 8, True, True, False), ("masker.2", 64, 16, 0, 1, False, False, True),
                                             ("masker.0", 64, 3, 8, 16, True, False, False), ("features.3", 32, 8, 0, 8, False, True, False)):
    a = torch.randint(0, 256, (n, hw, hw, ca), dtype=torch.uint8, device=dev) if u8 else torch.randn(n, hw, hw, ca, device=dev).half()
    b = torch.randn(n, hw // 2, hw // 2, cb, device=dev).half() if cb else None
    w = torch.randn(9 * (ca + cb) * co, device=dev) * 0.1
    bias = torch.zeros(co, device=dev)
    w16 = torch.empty(lib.cgs_gen16_packed_weight_halves(ca, cb, co), device=dev, dtype=torch.float16)
    _lib.call("cgs_gen16_pack_weights", ca, cb, co, C.c_void_p(w.data_ptr()), C.c_void_p(w16.data_ptr()), gen._s())
    for _ in range(3):
        gen._conv16(a, b, w16, bias.data_ptr(), co, act="relu", pool=pool, out_f32=f32o)
    torch.cuda.synchronize()
    buf = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
    lib.dbg_gen16_stamps(C.c_void_p(buf.data_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gen._conv16(a, b, w16, bias.data_ptr(), co, act="relu", pool=pool, out_f32=f32o); e1.record()
    torch.cuda.synchronize()
    lib.dbg_gen16_stamps(C.c_void_p(0))
    s = buf.cpu().numpy().reshape(4096, 8).astype(np.float64)
    d = np.diff(s[:, :6], axis=1)
    print(f"{name:12s} kernel {e0.elapsed_time(e1) * 1e3:7.1f} us; mean ticks (100 MHz) start->staged {d[:,0].mean():7.1f} barrier {d[:,1].mean():6.1f} "
          f"mfma {d[:,2].mean():7.1f} barrier {d[:,3].mean():6.1f} epilogue {d[:,4].mean():7.1f}; WG life {(s[:,5]-s[:,0]).mean():7.1f}")
