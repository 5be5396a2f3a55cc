"""A/B against an earlier commit: python tools/build_rev.py REV NAME  ->  <pkg>/libcgs_hip_NAME.so built from REV's csrc/ + include/ (same flags as the
product build).  Run a bench against it with CGS_LIB_PATH (tools/ab_many.sh takes NAME as a variant).  The Python host code stays the working tree's:
only usable while the C ABI of the entry points the step calls is unchanged between REV and the tree."""
import importlib, os, subprocess, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import cgs_amd
b = importlib.import_module(cgs_amd.__name__ + ".build")
from concurrent.futures import ThreadPoolExecutor
rev, name = sys.argv[1], sys.argv[2]
tmp = tempfile.mkdtemp(prefix="cgs_rev_")
pkg = os.path.basename(b.HERE)
os.makedirs(tmp + "/csrc"); os.makedirs(tmp + "/include"); os.makedirs(tmp + "/obj")
files = subprocess.check_output(["git", "ls-tree", "-r", "--name-only", rev, f"{pkg}/csrc/", "include/"], cwd=b.REPO, text=True).split()
for f in files:
    dst = tmp + ("/include/" if f.startswith("include/") else "/csrc/") + os.path.basename(f)
    with open(dst, "wb") as fp:
        fp.write(subprocess.check_output(["git", "show", f"{rev}:{f}"], cwd=b.REPO))
srcs = [s for s in b.SOURCES if os.path.exists(f"{tmp}/csrc/{s}")]
flags = ["-O3", f"--offload-arch={b.ARCH}", "-std=c++17", "-fPIC", "-I", tmp + "/include", "-I", tmp + "/csrc"]
def one(src):
    obj = f"{tmp}/obj/{src.replace('.hip', '.o')}"
    r = subprocess.run([b._hipcc()] + flags + b.EXTRA_FLAGS.get(src, []) + ["-c", f"{tmp}/csrc/{src}", "-o", obj], capture_output=True, text=True)
    if r.returncode:
        raise SystemExit(f"{src}:\n{r.stderr}")
    return obj
with ThreadPoolExecutor(max_workers=6) as ex:
    objs = list(ex.map(one, srcs))
lib = os.path.join(b.HERE, f"libcgs_hip_{name}.so")
r = subprocess.run([b._hipcc(), "-shared", "-fPIC", f"--offload-arch={b.ARCH}", "-o", lib] + objs, capture_output=True, text=True)
if r.returncode:
    raise SystemExit(r.stderr)
print("built", lib, "from", rev)
