"""One layer shape of the generic weight gradient (gen_wgrad_rows_kernel), a few launches: the target of rocprofv3 --pmc runs.
Usage: python tools/genw_one.py [hw ca cb co n pooled]"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from cgs_amd import _lib, generic  # noqa: E402

hw, ca, cb, co, n, pooled = (int(v) for v in sys.argv[1:7]) if len(sys.argv) > 6 else (32, 40, 0, 40, 1536, 1)
dev = torch.device("cuda:0")
a = torch.randn(n, hw, hw, ca, device=dev)
b = torch.randn(n, hw // 2, hw // 2, cb, device=dev) if cb else None
if pooled:
    dy = torch.randn(n, hw // 2, hw // 2, co, device=dev)
    am = torch.randint(0, 5, (n, hw // 2, hw // 2, co), device=dev, dtype=torch.uint8)
else:
    dy, am = torch.randn(n, hw, hw, co, device=dev), None
lib = _lib.load()
nsl = lib.cgs_gen_conv3x3_bwd_weight_slabs(n, ca, cb, co)
slab = torch.zeros(nsl, 9 * (ca + cb) * co + co, device=dev)
for _ in range(5):
    _lib.call("cgs_gen_conv3x3_bwd_weight", n, hw, ca, cb, co, 0, 2, generic._p(a), generic._p(b), generic._p(dy), generic._p(am),
              generic._p(slab), generic._s())
torch.cuda.synchronize()
