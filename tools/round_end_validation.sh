#!/bin/bash
# Round-end recipe on the GPU box (gpurun -- 'bash tools/round_end_validation.sh'): the full GPU suite, smoke(), the default bench line
# (incl. its path-C / path-B / ERT legs), its rocprofv3 kernel-trace summary, the per-step launch lists, the GEMM step breakdown, the
# stand-alone path-B / path-C benches and the HBM-traffic PMC passes.  Everything lands in gpurun_out/final/; the files that are judged
# are then copied into profiles/ (profiles/README.md says which).
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
ROOT=$PWD
O=gpurun_out/final; mkdir -p $O
timeout -k 5 900 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^Extension" | tail -2 | tee $O/pytest.txt
timeout -k 5 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee $O/smoke.txt
timeout -k 5 400 python bench.py 2>$O/bench.err | tail -1 > $O/bench_default.json.log; cut -c1-200 $O/bench_default.json.log
tr() { tag=$1; shift; ( cd /tmp && export TMPDIR=/tmp && timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $ROOT/$O/prof_$tag -o b -- python $ROOT/bench.py "$@" --no-frame --no-cpu --no-eager --no-f32 --no-dropin --no-paths --no-ert-scene > /dev/null 2>&1 < /dev/null ); }
tr a4 --steps 4 --warmup 2; tr a14 --steps 14 --warmup 2
python tools/rocprof_summary.py $(find $O/prof_a14 -name "b_kernel_trace.csv") > $O/bench_train_kernel_stats.txt; head -8 $O/bench_train_kernel_stats.txt | cut -c1-160
python tools/per_step_launches.py $(find $O/prof_a4 -name "b_kernel_trace.csv") 4 $(find $O/prof_a14 -name "b_kernel_trace.csv") 14 > $O/pathA_per_step_launches.txt; head -3 $O/pathA_per_step_launches.txt | cut -c1-160
timeout -k 5 300 python tools/gemm_step_breakdown.py 2>&1 | grep -v amdgpu.ids > $O/gemm_step_breakdown.txt; head -2 $O/gemm_step_breakdown.txt
timeout -k 5 300 python tools/bench_classic.py 2>&1 | tail -1 > $O/pathB_bench.json.log; cut -c1-300 $O/pathB_bench.json.log
timeout -k 5 400 python tools/bench_zip.py --rays 65536 2>&1 | tail -1 > $O/pathC_bench.json.log; cut -c1-300 $O/pathC_bench.json.log
( cd /tmp && export TMPDIR=/tmp && timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $ROOT/$O/prof_z -o b -- python $ROOT/tools/bench_zip.py --rays 65536 --steps 8 --train-only > /dev/null 2>&1 < /dev/null )
python tools/rocprof_summary.py $(find $O/prof_z -name "b_kernel_trace.csv") > $O/pathC_train_kernel_stats.txt; head -12 $O/pathC_train_kernel_stats.txt | cut -c1-160
bash tools/pmc_gemm_traffic.sh > $O/gemm_nt8p_traffic.txt 2>&1; tail -12 $O/gemm_nt8p_traffic.txt
PMC_SOURCE=profiles/r6_x_pathC_pathB_pmc.txt bash tools/pmc_paths.sh > $O/pmc_paths.log 2>&1; cp gpurun_out/pmc_paths/summary.txt $O/pathC_pathB_pmc.txt; cp gpurun_out/pmc_paths/roofline_traffic_paths.json $O/; cat $O/roofline_traffic_paths.json
PMC_SOURCE=profiles/r6_x_grid_encoder_pmc.txt bash tools/pmc_grid.sh > $O/pmc_grid.log 2>&1; cp gpurun_out/pmc_grid/summary.txt $O/grid_encoder_pmc.txt; cat gpurun_out/pmc_grid/roofline_traffic_grid.json
python - <<'PY'
import json
a = json.load(open("gpurun_out/final/roofline_traffic_paths.json")); a.update(json.load(open("gpurun_out/pmc_grid/roofline_traffic_grid.json")))
json.dump(a, open("gpurun_out/final/roofline_traffic_paths.json", "w"), indent=1)
PY
timeout -k 5 200 python tools/bench_grid.py --sweep 2>&1 | grep -v amdgpu.ids > $O/grid_encoder_leg_and_sweep.txt; tail -4 $O/grid_encoder_leg_and_sweep.txt | cut -c1-200
( cd /tmp && export TMPDIR=/tmp && timeout -k 5 150 rocprofv3 --kernel-trace --output-format csv -d $ROOT/$O/prof_g -o b -- python $ROOT/tools/bench_grid.py --once > /dev/null 2>&1 < /dev/null )
python tools/rocprof_summary.py $(find $O/prof_g -name "b_kernel_trace.csv") > $O/grid_encoder_kernel_stats.txt; head -10 $O/grid_encoder_kernel_stats.txt | cut -c1-160
timeout -k 5 200 python tools/ert_classic_analysis.py --steps 600 --rows 900 --row0 0 2>&1 | grep -v amdgpu.ids | tail -1 > $O/pathB_ert_fitted_frame.json.log; cut -c1-300 $O/pathB_ert_fitted_frame.json.log
rm -rf $O/prof_*
# optional extras (ROUND_END_EXTRAS=1, +4 GPU-minutes): the suite under the LDS scribble, the stale-LDS repetition screen, the 512-ray step
if [ -n "$ROUND_END_EXTRAS" ]; then
  SNERF_TEST_SCRIBBLE_LDS=1 timeout -k 5 600 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^Extension" | tail -1 | tee $O/pytest_lds_scribble.txt
  timeout -k 5 300 python tools/stress_stale_lds.py 600 2>&1 | grep "path \|TOTAL" | tee $O/stress_stale_lds.txt
  bash tools/probes/small_step_profile.sh 2>&1 | head -12 | tee $O/small_step.txt
fi
