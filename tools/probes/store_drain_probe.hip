// Is the ~12 B per clock and CU at which one 8-wave workgroup drains its row stores (store_pattern_probe.hip) the chip's HBM write
// rate shared by 256 CUs, or a limit of the CU's own store path?  The same store loop with 4 ... 256 workgroups (one per CU), with an
// HBM-sized footprint (every store a new line) and with an L2-resident footprint (each workgroup re-writes its own 64 KiB).
//   hipcc --offload-arch=gfx950 -O3 -o store_drain_probe store_drain_probe.hip && ./store_drain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// each iteration: every wave stores 1 KiB (the 8 waves: 8 KiB); `wrap` iterations later the workgroup starts over.
// STRIDED = false: the instruction's 1 KiB is contiguous; true: 8 segments of 128 B, `pitch` bytes apart (the row stores of the fused
// MLP kernels: 8 rows of a [M, 256] bf16 buffer per instruction, pitch 512) -- the same bytes of the same 8 KiB x 8 region per 8 iterations
template <bool STRIDED>
__global__ __launch_bounds__(512) void k(char* Y, long span, int iters, int wrap, long long* clk) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const u32x4 v = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, (unsigned)lane};
  constexpr int pitch = 512;
  // strided: iteration it writes column group (it & 3) (128 B) of rows [8 wave' .. ) of a 64-row x 512 B block: 4 iterations fill 8 rows x 512 B per wave
  char* base = STRIDED ? Y + blockIdx.x * span + (wave * 8 + (lane >> 3)) * pitch + (lane & 7) * 16 : Y + blockIdx.x * span + wave * 1024 + lane * 16;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (STRIDED) *(u32x4*)(base + (long)((it >> 2) % (wrap >> 2)) * 32768 + (it & 3) * 128) = v;
    else *(u32x4*)(base + (long)(it % wrap) * 8192) = v;
  }
  __builtin_amdgcn_s_waitcnt(0);
  const long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

int main() {
  const long span = 64l << 20;                      // 64 MiB per workgroup
  char* Y; hipMalloc(&Y, 256 * span);
  long long* clk; hipMallocManaged(&clk, 256 * sizeof(long long));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4096;                           // 32 MiB per workgroup and launch
  for (int mode = 0; mode < 4; ++mode) {
    const int wrap = (mode & 1) ? 8 : 4096;
    const bool strided = mode >= 2;
    printf("-- %s, %s\n", strided ? "8 x 128 B segments at pitch 512 per instruction" : "1 KiB contiguous per instruction",
           wrap == 8 ? "each workgroup re-writes its own 64 KiB (L2-resident)" : "every store a new line (32 MiB per workgroup)");
    for (int grid : {4, 16, 64, 128, 256}) {
      float best = 1e9; long long c = 0;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        if (strided) hipLaunchKernelGGL(k<true>, dim3(grid), dim3(512), 0, 0, Y, span, iters, wrap, clk);
        else hipLaunchKernelGGL(k<false>, dim3(grid), dim3(512), 0, 0, Y, span, iters, wrap, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) { best = ms; c = 0; for (int i = 0; i < grid; ++i) c += clk[i]; c /= grid; }
      }
      const double bytes = (double)iters * 8192;
      printf("  %3d workgroups: %8.1f us   %7.1f GB/s per workgroup   %6.2f TB/s total   %5.1f B per shader clock and CU (%lld clocks)\n", grid, best * 1e3,
             bytes / (best * 1e-3) / 1e9, bytes * grid / (best * 1e-3) / 1e12, bytes / (double)c, c);
    }
  }
  return 0;
}
