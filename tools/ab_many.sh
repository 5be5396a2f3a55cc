#!/bin/bash
# usage (GPU box): tools/ab_many.sh OUTDIR ROUNDS VARIANT...   interleaved bench runs: "base" (product library, fusion switched off through the
# environment), "prod" (product library) and libcgs_hip_VARIANT.so for every VARIANT (tools/build_variant.py)
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
pkg=$(ls -d *_amd)
out=$1; rounds=$2; shift 2
mkdir -p $out
one() {  # label, env assignments...
  label=$1; shift
  env "$@" python bench.py --steps 300 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', round(d['ms_per_step'],4))"
}
for i in $(seq 1 $rounds); do
  one base CGS_ENC1_TAIL_BWD_FUSED=0
  one prod CGS_X=0
  for v in "$@"; do one $v CGS_LIB_PATH=$root/$pkg/libcgs_hip_$v.so; done
done | tee $out/ab.txt
python - <<PY
import collections
d=collections.defaultdict(list)
for ln in open("$out/ab.txt"):
    k,v=ln.split(); d[k].append(float(v))
for k,v in d.items(): print(f"{k:6s} min {min(v):.4f}  median {sorted(v)[len(v)//2]:.4f}  all {v}")
PY
