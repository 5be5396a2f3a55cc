#!/bin/bash
# usage on GPU box: sweeps the generic weight gradient's workgroup target (needs hipcc on the box).  The source file is saved first and
# put back (and the library rebuilt from it) when the sweep ends or is interrupted, so the tree and libcgs_hip.so leave as they came.
cd ${GRAFT_REPO_ROOT:-/root/repo}
f=$(ls -d *_amd)/csrc/gen_train.hip
cp $f /tmp/gen_train.hip.orig
trap 'cp /tmp/gen_train.hip.orig $f; python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1' EXIT
for t in 512 768 1024 1536 2048 3072; do
  sed -i "s/int g = ([0-9]* + ncib \* ncog - 1)/int g = ($t + ncib * ncog - 1)/" $f
  python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
  r=$(python bench.py --chfak 5 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
  echo "target $t: $r ms/step"
done
