#!/bin/bash
# rocprofv3 PMC passes over the fused gradient chains (tools/fchain_probe.py).  Output: gpurun_out/pmc_fchain/
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
run() { tag=$1; shift; timeout 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $ROOT/gpurun_out/pmc_fchain/$tag -o p -- python $ROOT/tools/fchain_probe.py > /dev/null 2>&1 < /dev/null; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM
run sq3 SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
run tcc WRITE_SIZE
python $ROOT/tools/pmc_summary.py $ROOT/gpurun_out/pmc_fchain "fchain_bwd_kernel<0>"
