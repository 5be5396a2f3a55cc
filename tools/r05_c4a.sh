#!/bin/bash
mkdir -p gpurun_out/c4a
timeout -k 10 400 python -m pytest tests/test_gpu_generic.py tests/test_gpu_kernels.py -q -m gpu -x -s -k "fp16 or config4 or mask_head or infer" > gpurun_out/c4a/tests.log 2>&1 || { tail -30 gpurun_out/c4a/tests.log; exit 1; }
grep "config 4" gpurun_out/c4a/tests.log
tail -2 gpurun_out/c4a/tests.log
tools/ab_infer.sh gpurun_out/c4a 3 r5c pf
