// What does the MFMA + LDS-fragment-read loop sustain on this part, by wave tiling?  No global traffic, no epilogue: the bare inner
// loop of a 256x256 workgroup tile, (a) 8 waves of 128x64 (4 A + 2 B fragments per 8 MFMAs: the shipped gemm_nt8p tiling) and
// (b) 4 waves of 128x128 (4 A + 4 B fragments per 16 MFMAs).  Operands: random bf16 in LDS (power-realistic toggling).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_lds_probe mfma_lds_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// BAR: s_barrier after every BAR-th loop trip (0 = never): what workgroup-wide synchronisation at that granularity costs by itself
// PING: strict ping-pong -- in trip `it` only the wave row (wave >> 2) == (it & 1) issues its loads + MFMAs, the other row waits at
// the barrier (the shipped 8-phase kernel's alternation of MFMA and load sections, with empty load sections)
template <int WAVES, int NA, int NB, int BAR = 0, bool PING = false>
__global__ __launch_bounds__(WAVES * 64) void probe(const unsigned short* __restrict__ src, float* __restrict__ out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  for (int i = threadIdx.x; i < 128 * 1024 / 4; i += WAVES * 64) ((unsigned*)smem)[i] = ((const unsigned*)src)[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc[NA][NB];
  for (int i = 0; i < NA; ++i) for (int j = 0; j < NB; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const char* base = smem + lane * 16 + (wave & 3) * 4096;
  for (int it = 0; it < iters; ++it) {
    const char* p = base + (it & 7) * 12288;                 // a different fragment set every step (nothing to hoist)
    if (PING && (wave >> 2) != (it & 1)) { __builtin_amdgcn_s_barrier(); continue; }
    bf16x8 a[NA], b[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) a[i] = *(const bf16x8*)(p + i * 1024);
#pragma unroll
    for (int j = 0; j < NB; ++j) b[j] = *(const bf16x8*)(p + 16384 + j * 1024);
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    if (PING || (BAR > 0 && (it % BAR) == BAR - 1)) __builtin_amdgcn_s_barrier();
  }
  float s = 0.f;
  for (int i = 0; i < NA; ++i) for (int j = 0; j < NB; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * WAVES * 64 + threadIdx.x] = s;
}

template <int WAVES, int NA, int NB, int BAR = 0, bool PING = false>
static void run(const char* name, const unsigned short* src, float* out, int iters) {
  hipFuncSetAttribute((const void*)probe<WAVES, NA, NB, BAR, PING>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    for (int k = 0; k < 10; ++k) hipLaunchKernelGGL((probe<WAVES, NA, NB, BAR, PING>), dim3(256), dim3(WAVES * 64), 128 * 1024, 0, src, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = 10.0 * 256 * WAVES * (double)iters * NA * NB * 32768.0 * (PING ? 0.5 : 1.0);
    printf("%-44s %8.3f ms  %8.1f TFLOP/s  (LDS fragment bytes / MFMA: %d)\n", name, ms / 10, flop / (ms * 1e-3) / 1e12, (NA + NB) * 1024 / (NA * NB));
  }
}

int main() {
  std::vector<unsigned short> h(128 * 1024 / 2);
  srand(1);
  for (auto& v : h) { float f = (rand() / (float)RAND_MAX) * 2.f - 1.f; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
  unsigned short* src; float* out;
  hipMalloc(&src, 128 * 1024); hipMalloc(&out, 256 * 512 * 4);
  hipMemcpy(src, h.data(), 128 * 1024, hipMemcpyHostToDevice);
  run<8, 4, 2>("8 waves x (128x64):  4 A + 2 B per 8 MFMAs", src, out, 4096);
  run<4, 4, 4>("4 waves x (128x128): 4 A + 4 B per 16 MFMAs", src, out, 4096);
  run<8, 4, 2>("8 waves x (128x64) again", src, out, 4096);
  run<8, 2, 2>("8 waves x (64x64):   2 A + 2 B per 4 MFMAs", src, out, 8192);
  // barrier granularity (8 waves x 128x64, 8 MFMAs per trip): the shipped 8-phase kernel has two barriers per 8 MFMAs of a wave
  run<8, 4, 2, 1>("8 waves, barrier every 8 MFMAs", src, out, 4096);
  run<8, 4, 2, 2>("8 waves, barrier every 16 MFMAs", src, out, 4096);
  run<8, 4, 2, 4>("8 waves, barrier every 32 MFMAs (one k-tile of 64)", src, out, 4096);
  run<8, 4, 2, 16>("8 waves, barrier every 128 MFMAs", src, out, 4096);
  run<8, 4, 2, 1, true>("8 waves, strict ping-pong of the two wave rows (8 MFMAs each)", src, out, 8192);
  return 0;
}
