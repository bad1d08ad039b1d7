#!/bin/bash
# Round-end recipe on the GPU box (gpurun -- 'bash tools/round_end_validation.sh'): the full GPU suite, smoke(), the default bench line,
# its rocprofv3 kernel-trace summary, the path-B / path-C benches and the HBM-traffic PMC passes of the dominant kernel.  Everything lands
# in gpurun_out/final/; the files that are judged are then copied into profiles/ by hand (profiles/README.md says which).
cd /root/repo
O=gpurun_out/final; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^Extension" | tail -2 | tee $O/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee $O/smoke.txt
timeout 900 python bench.py 2>&1 | tail -1 > $O/bench_default.json.log; cut -c1-200 $O/bench_default.json.log
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /root/repo/$O/prof -o b -- python /root/repo/bench.py --steps 8 --warmup 2 --no-frame --no-cpu --no-eager --no-f32 --no-dropin > /dev/null 2>&1 < /dev/null )
f=$(find $O/prof -name "b_kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/rocprof_summary.py $f > $O/bench_train_kernel_stats.txt; head -8 $O/bench_train_kernel_stats.txt
timeout 300 python tools/gemm_step_breakdown.py 2>&1 | grep -v amdgpu.ids > $O/gemm_step_breakdown.txt; head -2 $O/gemm_step_breakdown.txt
timeout 300 python tools/gemm_step_breakdown.py bf16x3 2>&1 | grep -v amdgpu.ids > $O/gemm_step_breakdown_bf16x3.txt; head -1 $O/gemm_step_breakdown_bf16x3.txt
timeout 300 python tools/bench_classic.py 2>&1 | tail -1 > $O/pathB_bench.json.log; cut -c1-300 $O/pathB_bench.json.log
timeout 400 python tools/bench_zip.py --rays 65536 2>&1 | tail -1 > $O/pathC_bench.json.log; cut -c1-300 $O/pathC_bench.json.log
bash tools/pmc_gemm_traffic.sh > $O/gemm_nt8p_traffic.txt 2>&1; tail -12 $O/gemm_nt8p_traffic.txt
rm -rf $O/prof
