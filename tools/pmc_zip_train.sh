#!/bin/bash
# HBM traffic of the path-C TRAIN-step kernels (binned table gradient), separate rocprofv3 --pmc passes.  Output: gpurun_out/pmc_zip_train/
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
run() { tag=$1; shift; timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $ROOT/gpurun_out/pmc_zip_train/$tag -o p -- python $ROOT/tools/bench_zip.py --rays 65536 --steps 2 --train-only > /dev/null 2>&1 < /dev/null; }
run tcc2 FETCH_SIZE
run tcc3 WRITE_SIZE
run tcc1 TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
for k in "zip_bin_write_staged_kernelIDF16bLi4" "E, 1>" "zip_bin_accumulate_kernel<4>" "zip_bin_accumulate_kernel<1>" "zip_encode_fwd_all_kernelI6__half" "E, true>" "zip_prop_mlp_bwd" "zip_prop_mlp_fwd"; do
  python $ROOT/tools/pmc_summary.py $ROOT/gpurun_out/pmc_zip_train "$k" | grep -v "^SQ\|wave_cyc" | head -12
done
