#!/bin/bash
# copy the judged files of a tools/round_end_validation.sh run (gpurun_out/final/) into profiles/ under a round tag:
#   bash tools/copy_validation_to_profiles.sh r6_x
set -e
TAG=${1:?tag, e.g. r6_x}
S=gpurun_out/final
for f in bench_default.json.log bench_train_kernel_stats.txt pathA_per_step_launches.txt gemm_step_breakdown.txt pathB_bench.json.log pathC_bench.json.log \
         pathC_train_kernel_stats.txt gemm_nt8p_traffic.txt pathC_pathB_pmc.txt grid_encoder_kernel_stats.txt grid_encoder_leg_and_sweep.txt grid_encoder_pmc.txt \
         pathB_ert_fitted_frame.json.log; do
  [ -s $S/$f ] && cp $S/$f profiles/${TAG}_$f
done
{ cat $S/pytest.txt; cat $S/smoke.txt; [ -s $S/pytest_lds_scribble.txt ] && { echo "# under SNERF_TEST_SCRIBBLE_LDS=1:"; cat $S/pytest_lds_scribble.txt; };
  [ -s $S/stress_stale_lds.txt ] && { echo "# tools/stress_stale_lds.py 600:"; cat $S/stress_stale_lds.txt; }; } > profiles/${TAG}_gpu_tests.txt
[ -s $S/small_step.txt ] && cp $S/small_step.txt profiles/${TAG}_small_step.txt
[ -s $S/roofline_traffic_paths.json ] && cp $S/roofline_traffic_paths.json profiles/roofline_traffic_paths.json
ls -la profiles/${TAG}_*
