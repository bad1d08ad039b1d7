#!/bin/bash
# same-box A/B of builds of libsnerf_hip.so on the path-C train step (tools/bench_zip.py --train-only): the default library ("new") against
# snerf_amd/lib/libsnerf_hip_<tag>.so for every tag in $VARIANTS (default "old"; SNERF_HIP_LIB override), three alternating repetitions.
#   gpurun -- 'bash tools/probes/zip_lib_ab.sh'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2 3; do
  for v in ${VARIANTS:-old} new; do
    if [ $v != new ]; then export SNERF_HIP_LIB=$PWD/snerf_amd/lib/libsnerf_hip_$v.so; else unset SNERF_HIP_LIB; fi
    python tools/bench_zip.py --rays 65536 --steps 10 --train-only 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read()); print('$v', 'train_ms', d.get('train_ms'))"
  done
done
