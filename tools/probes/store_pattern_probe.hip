// How fast does the chip take the row stores of the NT GEMM epilogue, as a pattern, with nothing else going on?
//   hipcc --offload-arch=gfx950 -O3 -o store_pattern_probe store_pattern_probe.hip && ./store_pattern_probe
// Output Y [M, 1024] bf16 (2 KiB rows).  One workgroup of 8 waves per CU walks 256 x 256 tiles like the GEMM does; per tile every wave
// (wave row wr, wave column wc) stores 4 units of 32 rows x 128 B.
//   pattern 0: as the kernel -- lane (prow = lane >> 3, pch = lane & 7) stores 16 B of row 8 it + prow: 8 rows x 128 B per instruction
//   pattern 1: the same bytes as 1 KiB-contiguous instructions (a plain fill of the buffer; not a GEMM output layout)
//   pattern 2: as 0, but the four instructions of a unit spaced by ~200 clocks of s_sleep (is it the burst?)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int PAT>
__global__ __launch_bounds__(512) void k(unsigned short* Y, int M, int tiles_per_wg) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wr = wave >> 2, wc = wave & 3;
  const u32x4 v = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, (unsigned)lane};
  for (int t = 0; t < tiles_per_wg; ++t) {
    const int tile = blockIdx.x + t * gridDim.x;
    const int m0 = (tile >> 2) << 8, n0 = (tile & 3) << 8;
    if (m0 >= M) break;
    for (int u = 0; u < 4; ++u) {
      if (PAT == 1) {
        // the tile's 128 KiB as a linear range: wave -> 16 KiB, unit -> 4 KiB, instruction -> 1 KiB contiguous
        char* base = (char*)Y + (long)tile * 131072 + wave * 16384 + u * 4096;
        for (int it = 0; it < 4; ++it) *(u32x4*)(base + it * 1024 + lane * 16) = v;
      } else {
        const int prow = lane >> 3, pch = lane & 7;
        unsigned short* row = Y + (long)(m0 + wr * 128 + u * 32 + prow) * 1024 + n0 + wc * 64 + pch * 8;
        for (int it = 0; it < 4; ++it) {
          *(u32x4*)(row + (long)it * 8 * 1024) = v;
          if (PAT == 2) __builtin_amdgcn_s_sleep(3);
        }
      }
    }
  }
}

template <int PAT>
static void run(unsigned short* Y, int M, const char* name) {
  const int tiles = (M / 256) * 4, grid = 256, per = (tiles + grid - 1) / grid;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k<PAT>, dim3(grid), dim3(512), 0, 0, Y, M, per);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-70s %8.1f us / GB-launch   %6.2f TB/s\n", name, ms * 100, (double)M * 2048 / (ms / 10 * 1e-3) / 1e12);
  }
}

int main() {
  const int M = 524288;
  unsigned short* Y; hipMalloc(&Y, (size_t)M * 2048 + (1 << 20));
  run<0>(Y, M, "kernel pattern: 8 rows x 128 B per instruction, 2 KiB row stride");
  run<1>(Y, M, "1 KiB contiguous per instruction (plain fill)");
  run<2>(Y, M, "kernel pattern, ~200 clocks between a unit's instructions");
  run<0>(Y, M, "kernel pattern again");
  return 0;
}
