#!/bin/bash
# usage (GPU box): tools/g4ab.sh VARIANT...  -- the generic family's parity tests on the product library, then interleaved chfak-5 benches (prod / libcgs_hip_VARIANT.so)
set -e
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root; mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_gen4.py tests/test_gpu_gen_wgrad.py tests/test_gpu_generic.py tests/test_gpu_generic_train.py -x -q -m gpu > gpurun_out/g4ab_tests.txt 2>&1 || { tail -30 gpurun_out/g4ab_tests.txt; exit 1; }
tail -3 gpurun_out/g4ab_tests.txt
bash tools/ab_c5.sh 3 "$@" | tee gpurun_out/g4ab.txt
