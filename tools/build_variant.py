"""A/B builds: python tools/build_variant.py NAME [-DFLAG ...]  ->  <pkg>/libcgs_hip_NAME.so (objects under <pkg>/build_NAME/), the product
sources compiled with extra hipcc flags.  Run a bench against it with CGS_LIB_PATH=<that file> (cgs_amd._lib reads it)."""
import importlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import cgs_amd
b = importlib.import_module(cgs_amd.__name__ + ".build") if hasattr(cgs_amd, "__name__") else None
from concurrent.futures import ThreadPoolExecutor
import subprocess
name, extra = sys.argv[1], sys.argv[2:]
objdir = os.path.join(b.HERE, "build_" + name)
os.makedirs(objdir, exist_ok=True)
lib = os.path.join(b.HERE, f"libcgs_hip_{name}.so")
flags = ["-O3", f"--offload-arch={b.ARCH}", "-std=c++17", "-fPIC", "-I", os.path.join(b.REPO, "include"), "-I", b.CSRC] + extra
def one(src):
    obj = os.path.join(objdir, src.replace(".hip", ".o"))
    r = subprocess.run([b._hipcc()] + flags + b.EXTRA_FLAGS.get(src, []) + ["-c", os.path.join(b.CSRC, src), "-o", obj], capture_output=True, text=True)
    if r.returncode:
        raise SystemExit(f"{src}:\n{r.stderr}")
    return obj
with ThreadPoolExecutor(max_workers=6) as ex:
    objs = list(ex.map(one, b.SOURCES))
r = subprocess.run([b._hipcc(), "-shared", "-fPIC", f"--offload-arch={b.ARCH}", "-o", lib] + objs, capture_output=True, text=True)
if r.returncode:
    raise SystemExit(r.stderr)
print("built", lib)
