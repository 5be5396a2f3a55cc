#!/bin/bash
# usage (GPU box): tools/ab_infer.sh OUTDIR ROUNDS VARIANT...   interleaved config-4 runs (fp16 inference, batch 2048) of the product library and variants
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
pkg=$(ls -d *_amd)
out=$1; rounds=$2; shift 2
mkdir -p $out
one() { label=$1; shift; env "$@" python bench.py --mode infer --batch 2048 --fp16 --steps 200 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', round(d['ms_per_step'],4), round(d['value']/1e6,2))"; }
for i in $(seq 1 $rounds); do
  one prod CGS_X=0
  for v in "$@"; do one $v CGS_LIB_PATH=$root/$pkg/libcgs_hip_$v.so; done
done | tee $out/ab_infer.txt
