// Probe (round 6): v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands as a CORRECTION pass of a split GEMM.
//  (1) semantics: D[i][j] += 2^(sa-127) 2^(sb-127) sum_k A[i][k] B[k][j]; lane l holds 32 bytes of row (l & 31), k-half (l >> 5); the byte -> k map
//      inside a lane is irrelevant as long as A and B use the same one (checked against a host sum);
//  (2) rate: a register-resident loop of 32x32x64 fp8 MFMAs against 32x32x16 bf16 MFMAs (same bytes of operand per instruction pair).
//  hipcc --offload-arch=gfx950 -O3 tools/probes/f8_mfma_probe.hip -o /tmp/f8probe && /tmp/f8probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ void sem_kernel(const unsigned char* A, const unsigned char* B, float* D, int scale) {
  const int l = threadIdx.x;
  i32x8 a, b;
  // operand bytes: row (l & 31), 32 consecutive k of half (l >> 5)
  memcpy(&a, A + (l & 31) * 64 + (l >> 5) * 32, 32);
  memcpy(&b, B + (l & 31) * 64 + (l >> 5) * 32, 32);       // B stored as [j][k]
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, scale, 0, scale);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];   // row i' (of A), col j' = l & 31 (of B)
}

template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, int scale) {
  i32x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = 0x38383838 + threadIdx.x + e; b[e] = 0x38383838 - e; }
  f32x16 c[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (MODE == 0) {
        c[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c[i], 0, 0, 0, scale, 0, scale);
      } else {
        bf16x8 a0 = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, a, 0, 1, 2, 3)), a1 = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, a, 4, 5, 6, 7));
        bf16x8 b0 = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b, b, 0, 1, 2, 3)), b1 = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b, b, 4, 5, 6, 7));
        c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, c[i], 0, 0, 0);
        c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, c[i], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += c[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

static float e4m3(unsigned char v) {           // OCP e4m3fn
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float x = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.f + m / 8.f, e - 7);
  return s ? -x : x;
}

int main() {
  std::vector<unsigned char> A(32 * 64), B(32 * 64);
  srand(1);
  for (auto& v : A) { v = rand() & 0xff; if ((v & 0x7f) == 0x7f) v ^= 1; }    // (0x7f / 0xff = NaN in e4m3fn)
  for (auto& v : B) { v = rand() & 0xff; if ((v & 0x7f) == 0x7f) v ^= 1; }
  unsigned char *dA, *dB; float* dD;
  hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dD, 32 * 32 * 4);
  hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
  for (int sc : {127, 116, 0}) {
    const int sv = sc * 0x01010101;
    hipLaunchKernelGGL(sem_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD, sv);
    std::vector<float> D(32 * 32);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0, ref0 = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
      double r = 0;
      for (int k = 0; k < 64; ++k) r += (double)e4m3(A[i * 64 + k]) * e4m3(B[j * 64 + k]);
      r *= ldexp(1.0, 2 * (sc - 127));
      const double e = fabs(D[i * 32 + j] - r) / (fabs(r) + 1e-30);
      if (e > worst) worst = e;
      if (i == 3 && j == 5) ref0 = r;
    }
    printf("scale byte %3d: max rel err vs host sum %.3e   (D[3][5] = %.6e, host %.6e)\n", sc, worst, D[3 * 32 + 5], ref0);
  }
  float* dO; hipMalloc(&dO, 1024 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int mode = 0; mode < 2; ++mode) for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    if (mode == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(1024), dim3(256), 0, 0, dO, iters, 0x74747474);
    else hipLaunchKernelGGL(rate_kernel<1>, dim3(1024), dim3(256), 0, 0, dO, iters, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per iteration and wave: 4 accumulators x (one 32x32x64 | two 32x32x16) MFMAs
    const double flops = (double)1024 * 4 * iters * 4 * (mode == 0 ? 2.0 * 32 * 32 * 64 : 2 * 2.0 * 32 * 32 * 16);
    printf("%s: %.2f ms, %.0f TFLOP/s (operand bytes per instruction group identical)\n", mode == 0 ? "fp8 32x32x64 scaled" : "bf16 2 x 32x32x16  ", ms, flops / ms / 1e9);
  }
  return 0;
}
