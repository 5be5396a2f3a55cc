#!/bin/bash
# usage (GPU box): tools/ab_env.sh OUTDIR ROUNDS LIB "LABEL:ENV=VAL,ENV=VAL" ...   interleaved bench runs of one library under several environments
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
pkg=$(ls -d *_amd)
out=$1; rounds=$2; lib=$3; shift 3
mkdir -p $out
for i in $(seq 1 $rounds); do
  for spec in "$@"; do
    label=${spec%%:*}; envs=${spec#*:}
    env CGS_LIB_PATH=$root/$pkg/$lib $(echo $envs | tr ',' ' ') python bench.py --steps 300 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', round(d['ms_per_step'],4))"
  done
done | tee $out/ab_env.txt
python - <<PY
import collections
d=collections.defaultdict(list)
for ln in open("$out/ab_env.txt"):
    k,v=ln.split(); d[k].append(float(v))
for k,v in d.items(): print(f"{k:8s} min {min(v):.4f}  median {sorted(v)[len(v)//2]:.4f}  all {v}")
PY
