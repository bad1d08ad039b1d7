// Probe for rows_sum() of csrc/fmlp.hip (the reduce-scatter butterfly of the fused colour-head backward): every lane L of a wave
// loads registers a[r] = 1000 r + L and prints what it ends up with -- expected 32000 r' + (sum of the lane ids of its half) for
// r' = L & 15.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o rows_sum_probe rows_sum_probe.hip
#include "../../snerf_amd/csrc/fmlp.hip"
#include <cstdio>

__global__ void probe(float* out) {
  const int lane = threadIdx.x & 63;
  f32x16 a;
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = 1000.f * r + lane;
  out[lane] = rows_sum(a, lane);
}

int main() {
  float* d;
  float h[64];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    const int r = l & 15;
    const float want = 32000.f * r + (l < 32 ? 496.f : 1520.f);
    if (h[l] != want) ++bad;
    printf("lane %2d: got %9.1f want %9.1f%s\n", l, h[l], want, h[l] != want ? "  <-- MISMATCH" : "");
  }
  printf("rows_sum probe: %d mismatches\n", bad);
  return bad != 0;
}
