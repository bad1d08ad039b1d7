cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$PWD; O=gpurun_out/small; mkdir -p $O
F="--no-frame --no-cpu --no-eager --no-f32 --no-dropin --no-paths --no-ert-scene"
python bench.py --rays 512 --steps 60 --warmup 10 $F 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('eager-launch 512:', d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['roofline']['avg_launch_ms'])"
python bench.py --rays 512 --steps 60 --warmup 10 --graph $F 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('graph 512:', d['ms_per_step'], d.get('ms_per_step_median'))"
tr() { tag=$1; shift; ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $ROOT/$O/prof_$tag -o b -- python $ROOT/bench.py "$@" $F > /dev/null 2>&1 < /dev/null ); }
tr a4 --rays 512 --steps 4 --warmup 2; tr a14 --rays 512 --steps 14 --warmup 2
python tools/per_step_launches.py $(find $O/prof_a4 -name "b_kernel_trace.csv") 4 $(find $O/prof_a14 -name "b_kernel_trace.csv") 14 > $O/small_per_step_launches.txt; head -60 $O/small_per_step_launches.txt | cut -c1-170
rm -rf $O/prof_*
