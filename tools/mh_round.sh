#!/bin/bash
# usage (GPU box): tools/mh_round.sh TAG   -> mask-head tests, its phase stamps (debug library) and its time under the profiler
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root; pkg=$(ls -d *_amd)
python -m pytest tests -m gpu -x -q -k "mask or engine" > gpurun_out/$1.tests.log 2>&1; rc=$?; tail -1 gpurun_out/$1.tests.log; [ $rc -ne 0 ] && exit $rc
CGS_LIB_PATH=$root/$pkg/libcgs_hip_stamps.so python tools/mh_stamps.py 512 2>&1 | tail -11 | tee gpurun_out/$1.stamps.txt
tools/whatif.sh $1 "mask_head" prod
