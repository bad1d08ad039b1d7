#!/bin/bash
# Path B training's store path against a pure store stream, in the write-request counters the round-5 review named (TCC_EA0_WRREQ_*):
#   (1) tools/probes/store_pattern_probe (row stores in the fused kernels' pattern and as a plain fill, nothing else going on)
#   (2) fmlp_kernel<0, false, true>  -- the fused TRAINING forward (tools/fmlp_single.py)
#   (3) fchain_bwd_kernel<0>        -- the fused gradient chain (tools/fchain_probe.py)
# Output: gpurun_out/pmc_store_path/{probe,fmlp,fchain}/..., summary on stdout.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/pmc_store_path; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $O/store_pattern_probe $ROOT/tools/probes/store_pattern_probe.hip 2>/dev/null
cd /tmp && export TMPDIR=/tmp
$O/store_pattern_probe | tee $O/probe_timing.txt
run() { dir=$1; tag=$2; shift 2; cmd=$1; shift; timeout -k 5 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$dir/$tag -o p -- $cmd > /dev/null 2>&1 < /dev/null; }
for spec in "probe|$O/store_pattern_probe" "fmlp|python $ROOT/tools/fmlp_single.py" "fchain|python $ROOT/tools/fchain_probe.py"; do
  d=${spec%%|*}; c=${spec#*|}
  run $d wr1 "$c" TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum
  run $d wr2 "$c" WRITE_SIZE
  run $d wr3 "$c" TCP_TCC_WRITE_REQ_sum TCC_TAG_STALL_sum TCP_PENDING_STALL_CYCLES_sum
  run $d wr4 "$c" TCC_EA0_WRREQ_IO_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_EA0_WR_UNCACHED_32B_sum TCC_WRITEBACK_sum
done
echo "== pure store stream, kernel pattern (k<0>)";  python $ROOT/tools/pmc_summary.py $O/probe "void k<0>"
echo "== pure store stream, plain fill (k<1>)";      python $ROOT/tools/pmc_summary.py $O/probe "void k<1>"
echo "== fused training forward";                    python $ROOT/tools/pmc_summary.py $O/fmlp "fmlp_kernel<0, false, true>"
echo "== fused gradient chain";                      python $ROOT/tools/pmc_summary.py $O/fchain "fchain_bwd_kernel<0>"
