#!/bin/bash
# rocprofv3 PMC passes over the fused colour-head kernels (run on the GPU box).  Output: gpurun_out/pmc_fcolour/
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
run() { tag=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $ROOT/gpurun_out/pmc_fcolour/$tag -o p -- python $ROOT/tools/fcolour_probe.py > /dev/null 2>&1; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
run tcc1 TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
run tcc2 FETCH_SIZE
run tcc3 WRITE_SIZE
run tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum
for k in fcolour_fwd_kernelILb0ELi12 fcolour_fwd_kernelILb1ELi12 fcolour_bwd; do python $ROOT/tools/pmc_summary.py $ROOT/gpurun_out/pmc_fcolour $k; done
