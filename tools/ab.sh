#!/bin/bash
# usage (GPU box): tools/ab.sh NAME VARIANT  -> parity tests of the product build, then interleaved bench runs of the product library and
# libcgs_hip_VARIANT.so (tools/build_variant.py), kernel stats of both under gpurun_out/
name=$1; var=$2
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
pkg=$(ls -d *_amd)
python -m pytest tests -m gpu -x -q -k "${3:-engine or kernels or modules}" > gpurun_out/$name.tests.log 2>&1; rc=$?; tail -3 gpurun_out/$name.tests.log; [ $rc -ne 0 ] && exit $rc
for i in 1 2 3; do
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('product ', d['ms_per_step'])"
  CGS_LIB_PATH=$root/$pkg/libcgs_hip_$var.so python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var ', d['ms_per_step'])"
done
tools/prof.sh $name.prof && python tools/kernel_stats.py gpurun_out/$name.prof 0 22 > gpurun_out/$name.kernels.txt && cat gpurun_out/$name.kernels.txt
export CGS_LIB_PATH=$root/$pkg/libcgs_hip_$var.so
tools/prof.sh $name.prof_$var && python tools/kernel_stats.py gpurun_out/$name.prof_$var 0 22 > gpurun_out/$name.kernels_$var.txt && cat gpurun_out/$name.kernels_$var.txt
