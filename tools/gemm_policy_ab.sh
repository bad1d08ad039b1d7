#!/bin/bash
# A/B of the persistent NT kernel's cache policies between NON-probe builds (snerf_amd/lib/ab_gemm_{pp,sp,pl,sl}.so: plain / streaming
# stores x plain / streaming activation loads, built with -DGEMM_NT_STORES / -DGEMM_NT_ALOADS), three interleaved rounds, one process each.
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
cp snerf_amd/lib/libsnerf_hip.so /tmp/orig.so
for round in 1 2 3; do
  for v in pp sp pl sl; do
    cp snerf_amd/lib/ab_gemm_$v.so snerf_amd/lib/libsnerf_hip.so
    python tools/gemm_flavour_time.py "stores=${v:0:1} loads=${v:1:1} (p=plain s/l=nt)" 2>&1 | grep -v amdgpu.ids
  done
done
cp /tmp/orig.so snerf_amd/lib/libsnerf_hip.so
