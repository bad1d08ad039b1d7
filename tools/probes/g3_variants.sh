#!/bin/bash
# Thread-count / points-per-thread variants of the staged writer (run HERE, cross-compiled; the .so files travel with gpurun):
#   snerf_amd/lib/probe_g3_<wt>x<ppt>.so; time them on the box with tools/probes/g3_time.py under SNERF_HIP_LIB.
cd "$(dirname "$0")/../.." || exit 1
for v in "1024 2 4" "512 4 2" "768 4 3" "1024 4 4" "512 8 2"; do
  set -- $v
  d=/tmp/g3var_$1x$2; rm -rf $d; mkdir -p $d; cp snerf_amd/csrc/*.hip snerf_amd/csrc/*.h $d/; cp snerf_amd/csrc/*.o $d/ 2>/dev/null; rm -f $d/zip.o
  ( cd $d && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value -DG3_WT_=$1 -DG3_PPT_=$2 -DG3_WPE_=$3 -Rpass-analysis=kernel-resource-usage -c zip.hip -o zip.o 2>&1 | grep -A8 "Function Name: _Z22g3_write_staged_kernelIDF16_Li4ELb1EEv3G3W" | grep -E "VGPRs Spill|ScratchSize" | tr '\n' ' '; echo " <- $1 threads x $2 points"; \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OLDPWD/snerf_amd/lib/probe_g3_$1x$2.so gemm.o fmlp.o encode.o sampler.o composite.o elementwise.o grid.o zip.o callers.o ert.o foreground.o ) || exit 1
done
