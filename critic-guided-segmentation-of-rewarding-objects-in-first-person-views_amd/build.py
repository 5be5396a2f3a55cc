"""Builds libcgs_hip.so (gfx950) in-tree with hipcc.  No torch types cross this boundary: the library is
plain C ABI (include/cgs_hip.h) and is loaded with ctypes by ``_lib.py``."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcgs_hip.so")
SOURCES = ["conv_fwd.hip", "conv_wgrad.hip", "wgrad_dec0.hip", "conv_bwd_both.hip", "mconv.hip", "mask_head.hip", "mask_fwd.hip", "head.hip", "elementwise.hip", "tail.hip", "tail_infer.hip", "gen.hip", "gen4.hip", "gen_train.hip", "gen_f16.hip", "gen_bf16_train.hip", "hconv.hip", "hwgrad.hip", "bn.hip", "gen_enc0.hip"]
HEADERS = ["cgs_common.h", "head_wgrad.h", "conv_tile.h", "conv_body.h", "wgrad_body.h", "head_body.h", "tail_common.h", "tail4.h", "tail_h16.h", "gen_common.h", "gen4_common.h", "gen_wgrad_rows.h", "gen_wgrad_fold.h", "wgrad_dec0.h", "wgrad_sparse.h", os.path.join(REPO, "include", "cgs_hip.h")]
ARCH = "gfx950"
# conv_fwd / conv_bwd_both: no SLP vectorisation -- measured on the real step (profiles/r02_slp_ab_*.txt): with it the compiler pairs
# accumulators into v_pk_fma_f32 through extra v_mov and the step is slower (0.913 vs 0.837 ms at the time); the packed form itself is not
# slower than the scalar one
EXTRA_FLAGS = {"conv_fwd.hip": ["-fno-slp-vectorize"], "conv_bwd_both.hip": ["-fno-slp-vectorize"]}


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=True):
    """Compile every HIP source for gfx950 and link libcgs_hip.so next to this file."""
    if not force and not _stale():
        return LIB
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    cc = _hipcc()
    flags = ["-O3", f"--offload-arch={ARCH}", "-std=c++17", "-fPIC", "-I", os.path.join(REPO, "include"), "-I", CSRC]

    def one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        extra = EXTRA_FLAGS.get(src, [])
        cmd = [cc] + flags + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
        return obj

    with ThreadPoolExecutor(max_workers=6) as ex:
        objs = list(ex.map(one, SOURCES))
    r = subprocess.run([cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    if verbose:
        print(f"built {LIB} ({os.path.getsize(LIB) // 1024} KiB) for {ARCH}", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
