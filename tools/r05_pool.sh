#!/bin/bash
mkdir -p gpurun_out/pool
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py -q -m gpu -x > gpurun_out/pool/tests.log 2>&1 || { tail -30 gpurun_out/pool/tests.log; exit 1; }
tail -2 gpurun_out/pool/tests.log
tools/ab_many.sh gpurun_out/pool 3 r5c
