#!/bin/bash
# usage (GPU box): tools/g4ab.sh  -- gen4 parity tests on the `hold` variant, then interleaved chfak-5 benches (prod / peel / hold)
set -e
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root; pkg=$(ls -d *_amd); mkdir -p gpurun_out
CGS_LIB_PATH=$root/$pkg/libcgs_hip_tiny5.so timeout -k 10 500 python -m pytest tests/test_gpu_gen4.py tests/test_gpu_generic.py -x -q -m gpu > gpurun_out/g4ab_tests.txt 2>&1 || { tail -30 gpurun_out/g4ab_tests.txt; exit 1; }
tail -3 gpurun_out/g4ab_tests.txt
bash tools/ab_c5.sh 3 tiny4 tiny5 | tee gpurun_out/g4ab.txt
