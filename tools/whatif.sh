#!/bin/bash
# usage (GPU box): tools/whatif.sh TAG KERNEL-SUBSTRING VARIANT...   -> per variant library (tools/build_variant.py; "prod" = the product library) the
# per-step time of the matching kernels under rocprofv3 --kernel-trace --stats and the step's kernel sum (timing experiments: what-if builds compute
# wrong results on purpose)
tag=$1; pat=$2; shift 2
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
pkg=$(ls -d *_amd)
for v in "$@"; do
  if [ "$v" = "prod" ]; then unset CGS_LIB_PATH; else export CGS_LIB_PATH=$root/$pkg/libcgs_hip_$v.so; fi
  tools/prof.sh $tag.$v || { echo "$v: profile run failed"; tail -3 gpurun_out/$tag.$v.log; continue; }
  python tools/kernel_stats.py gpurun_out/$tag.$v 0 14 > gpurun_out/$tag.$v.kernels.txt
  echo "== $v: $(head -1 gpurun_out/$tag.$v.kernels.txt | sed 's/.*profile//')"
  grep -E "$pat" gpurun_out/$tag.$v.kernels.txt
done
