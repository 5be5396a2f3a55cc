# SQ counters of the row-block weight gradient on one layer shape (default: features.3 at chfak 5, 1536 images) -> gpurun_out/gwpmc/
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/gpurun_out/gwpmc/a -o pmc -- python3 $R/tools/genw_one.py "$@" > $R/gpurun_out/gwpmc.a.log 2>&1 &&
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/gwpmc/b -o pmc -- python3 $R/tools/genw_one.py "$@" > $R/gpurun_out/gwpmc.b.log 2>&1
