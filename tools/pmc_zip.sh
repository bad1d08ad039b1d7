#!/bin/bash
# HBM traffic of the path-C kernels (separate rocprofv3 --pmc passes; run on the GPU box).  Output: gpurun_out/pmc_zip/
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
run() { tag=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $ROOT/gpurun_out/pmc_zip/$tag -o p -- python $ROOT/tools/bench_zip.py --rays 16384 --steps 2 > /dev/null 2>&1; }
run tcc2 FETCH_SIZE
run tcc3 WRITE_SIZE
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum
