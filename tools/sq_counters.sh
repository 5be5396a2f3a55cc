#!/bin/bash
# usage (GPU box): tools/sq_counters.sh TAG [bench args]
# Four rocprofv3 --pmc passes over the same bench.py command (eager launches, 3 steps): two SQ sets, FETCH_SIZE, WRITE_SIZE
# (FETCH_SIZE and WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md "rocprofv3 PMC slots").  Parse with tools/sq_counters.py.
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters
  rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $root/gpurun_out/$tag/$1 -o pmc -- python3 $root/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-graph "${@:3}" > $root/gpurun_out/$tag.$1.log 2>&1
}
run sq1 "SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU" "$@" &&
run sq2 "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "$@" &&
run fetch "FETCH_SIZE" "$@" &&
run write "WRITE_SIZE" "$@"
