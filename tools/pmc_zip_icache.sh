#!/bin/bash
# instruction-cache, LDS and instruction-mix counters of the path-C featurisation / record kernels (separate rocprofv3 --pmc passes).  Output: gpurun_out/pmc_zip_icache/summary.txt (profiles/r6_zz_pathC_icache_lds_pmc.txt)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_zip_icache
cd /tmp && export TMPDIR=/tmp
run() { tag=$1; shift; timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$tag -o p -- python $ROOT/tools/bench_zip.py --rays 65536 --steps 2 --train-only > /dev/null 2>&1 < /dev/null; }
run ic1 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
run ic2 SQ_IFETCH SQ_IFETCH_LEVEL SQC_TC_INST_REQ SQC_ICACHE_BUSY_CYCLES
run lds SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
run sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES
{
for k in "zip_encode_fwd_all_kernelIfDF16_Li1ELb1" "zip_encode_fwd_all_kernelI6__half" "zip_bin_emit_all_kernelIDF16_Li1" "zip_bin_write_staged_kernelIDF16_Li4"; do
  python $ROOT/tools/pmc_summary.py $OUT "$k" | grep -v "^wait_\|^active_inst\|^L2 hit"
done
} > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
