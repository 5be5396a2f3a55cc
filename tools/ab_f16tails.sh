#!/bin/bash
# usage (GPU box): tools/ab_f16tails.sh [VARIANT-LIB ...]   interleaved config-4 runs: the fused one-launch tail (product), variant libraries, the two
# fp16-operand tail launches (CGS_F16_TAILS=1) and the fp32 tail kernels (=0)
cd ${GRAFT_REPO_ROOT:-/root/repo}
pkg=$(ls -d *_amd)
one() { label=$1; shift; env "$@" python bench.py --mode infer --batch 2048 --fp16 --steps 200 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', round(d['ms_per_step'],4), round(d['value']/1e6,2))"; }
for i in 1 2 3; do
  one fused1 CGS_F16_TAILS=fused1; one fused CGS_F16_TAILS=fused
  for v in "$@"; do one fused1_$v CGS_F16_TAILS=fused1 CGS_LIB_PATH=$PWD/$pkg/libcgs_hip_$v.so; done
  one h16tails CGS_F16_TAILS=1; one f32tails CGS_F16_TAILS=0
done
