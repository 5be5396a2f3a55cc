#!/bin/bash
# usage (GPU box): tools/ab_thresholds.sh   the step at N = 128 .. 256 with the two batch-size dependent launch forms forced on / off (hourglass.*_MIN_N)
one() { label=$1; n=$2; shift 2; env "$@" python bench.py --batch $n --steps 300 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=$n $label', round(d['ms_per_step'],4))"; }
for n in 128 160 192 224 256; do
  one default $n CGS_X=0
  one mask_fused $n CGS_MASK_TRAIN_FUSED_MIN_N=0
  one mask_unfused $n CGS_MASK_TRAIN_FUSED_MIN_N=100000
  one enc1_fused $n CGS_ENC1_TAIL_BWD_FUSED_MIN_N=0
  one enc1_unfused $n CGS_ENC1_TAIL_BWD_FUSED_MIN_N=100000
done
