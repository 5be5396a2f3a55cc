#!/bin/bash
mkdir -p gpurun_out/cvt
timeout -k 10 700 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py tests/test_gpu_modules.py -q -m gpu -x > gpurun_out/cvt/tests.log 2>&1 || { tail -30 gpurun_out/cvt/tests.log; exit 1; }
tail -2 gpurun_out/cvt/tests.log
tools/ab_many.sh gpurun_out/cvt 3 r5c
for v in prod r5c; do
  if [ $v = r5c ]; then export CGS_LIB_PATH=$PWD/$(ls -d *_amd)/libcgs_hip_r5c.so; fi
  python bench.py --mode infer --batch 2048 --steps 100 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('infer2048 fp32 $v', round(d['ms_per_step'],4), round(d['value']/1e6,3))"
done
