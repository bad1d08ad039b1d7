#!/bin/bash
# HBM traffic of the dominant kernel (persistent 8-phase NT GEMM, M = 524288, N = K = 1024, bf16, bias + ReLU): FETCH_SIZE and WRITE_SIZE in
# separate rocprofv3 --pmc passes (MI355X_MICROARCH.md, HBM section), + one SQ pass (MFMA busy).  Output: gpurun_out/pmc_nt8_traffic/
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
run() { tag=$1; shift; timeout -k 5 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $ROOT/gpurun_out/pmc_nt8_traffic/$tag -o p -- python $ROOT/tools/gemm_single.py 8 nt > /dev/null 2>&1 < /dev/null; }
run tcc2 FETCH_SIZE
run tcc3 WRITE_SIZE
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
python $ROOT/tools/pmc_summary.py $ROOT/gpurun_out/pmc_nt8_traffic "gemm_nt8p"
