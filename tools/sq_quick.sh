#!/bin/bash
# usage (GPU box): tools/sq_quick.sh TAG [bench args]  -> the two SQ counter passes of tools/sq_counters.sh only (no FETCH/WRITE), summary CSV
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
run() {
  rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $root/gpurun_out/$tag/$1 -o pmc -- python3 $root/bench.py --steps 3 --warmup 2 --prime-s 0 --no-cpu-baseline --no-graph "${@:3}" > $root/gpurun_out/$tag.$1.log 2>&1
}
run sq1 "SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU" "$@" &&
run sq2 "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "$@"
cd $root && python3 tools/sq_counters.py gpurun_out/$tag gpurun_out/$tag.sq.csv
