#!/bin/bash
# per-kernel times (rocprofv3 --kernel-trace, 11 traced steps) of the path-C train step under two builds: snerf_amd/lib/libsnerf_hip_old.so (SNERF_HIP_LIB override) and the default library.   gpurun -- "bash tools/probes/zip_lib_kernels_ab.sh"
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
for v in old new; do
  if [ $v = old ]; then export SNERF_HIP_LIB=$ROOT/snerf_amd/lib/libsnerf_hip_old.so; else unset SNERF_HIP_LIB; fi
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/zip_lib_kernels_ab/prof_$v -o b -- python $ROOT/tools/bench_zip.py --rays 65536 --steps 8 --train-only > /dev/null 2>&1 < /dev/null )
  python tools/rocprof_summary.py $(find gpurun_out/zip_lib_kernels_ab/prof_$v -name "b_kernel_trace.csv") > gpurun_out/zip_lib_kernels_ab/stats_$v.txt
  echo "== $v"; head -14 gpurun_out/zip_lib_kernels_ab/stats_$v.txt | cut -c1-150
  rm -rf gpurun_out/zip_lib_kernels_ab/prof_$v
done
