ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for mode in fused layered; do
  SNERF_CLASSIC_MODE=$mode timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/trainB_$mode -o p -- python $ROOT/tools/_train_b.py > /tmp/log_$mode.txt 2>&1 < /dev/null
  echo "== $mode rc=$?"; tail -3 /tmp/log_$mode.txt
  f=$(find $ROOT/gpurun_out/trainB_$mode -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then head -16 "$f" | cut -c1-220; fi
done
