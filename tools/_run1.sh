ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/r2_z_bench_default.json.log; cut -c1-600 gpurun_out/r2_z_bench_default.json.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_r2z -o b -- python $ROOT/bench.py --steps 8 --warmup 2 --no-frame --no-cpu --no-eager --no-f32 > $ROOT/gpurun_out/prof_r2z.log 2>&1 < /dev/null
f=$(find $ROOT/gpurun_out/prof_r2z -name "*kernel_trace.csv" | head -1)
if [ -n "$f" ]; then python $ROOT/tools/rocprof_summary.py $f 60 > $ROOT/gpurun_out/r2_z_bench_train_kernel_stats.txt; head -12 $ROOT/gpurun_out/r2_z_bench_train_kernel_stats.txt | cut -c1-170; fi
rm -f $ROOT/gpurun_out/prof_r2z/*kernel_trace.csv
cd $ROOT
timeout 300 python tools/bench_classic.py --steps 5 2>&1 | tail -1 > gpurun_out/r2_z_pathB_bench.json.log; cut -c1-400 gpurun_out/r2_z_pathB_bench.json.log
timeout 600 python tools/bench_zip.py 2>&1 | tail -1 > gpurun_out/r2_z_pathC_bench.json.log; cut -c1-500 gpurun_out/r2_z_pathC_bench.json.log
