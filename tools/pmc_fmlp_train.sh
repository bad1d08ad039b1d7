#!/bin/bash
# rocprofv3 PMC passes over the fused TRAINING forward (fmlp_kernel<0, false, true>).  Output: gpurun_out/pmc_fmlp_train/
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
run() { tag=$1; shift; timeout 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $ROOT/gpurun_out/pmc_fmlp_train/$tag -o p -- python $ROOT/tools/fmlp_single.py > /dev/null 2>&1 < /dev/null; }
run wr1 TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum
run wr2 WRITE_SIZE
run wr3 FETCH_SIZE
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
run sq2 SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCC_TAG_STALL_sum
run sq3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_MFMA
python $ROOT/tools/pmc_summary.py $ROOT/gpurun_out/pmc_fmlp_train "fmlp_kernel<0, false, true>"
