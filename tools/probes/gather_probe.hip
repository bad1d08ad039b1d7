// Calibration probe for the hash-grid gather: random-row gathers of W-byte rows (W = 4 / 8 / 16) at KNOWN indices from
// tables of 120 MB, 229 MB (the fp16 / fp32 NeRF grid) and 2 GB (nothing cache-resident), in two lane patterns:
//   scattered  every lane of a wave reads its own pseudo-random row (one distinct line per lane: the hashed levels)
//   clustered  the 64 lanes of a wave read 64 CONSECUTIVE rows at one pseudo-random base (one or a few lines per wave: the upper bound a
//              perfectly coherent gather could reach)
//   windowed   scattered, but every workgroup stays inside ONE 16 MB window of the table (a 2^21-row level of 8-byte entries: what a
//              workgroup of the training featurisation, one level per workgroup, touches) -- separates the address-translation reach
//              from the line-request rate
// Prints rows/s and useful GB/s per case from HIP events.  Run it under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` (and TCC_EA0_RDREQ_sum /
// TCC_HIT_sum TCC_MISS_sum in separate passes): FETCH_SIZE per launch / the known number of gathered rows = the bytes the counter books
// per row for THIS access width, which is the scale factor tools/pmc_summary.py applies to the gather kernels (the guide's x2 is stated
// for wide coalesced streams only), and the scattered rows/s is the ceiling zip_encode_kernel is reported against.
//   hipcc --offload-arch=gfx950 -O3 gather_probe.hip -o gather_probe && ./gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

template <typename T> struct Acc;
template <> struct Acc<uint32_t> { static __device__ unsigned f(uint32_t v) { return v; } };
template <> struct Acc<uint2> { static __device__ unsigned f(uint2 v) { return v.x ^ v.y; } };
template <> struct Acc<uint4> { static __device__ unsigned f(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; } };

// every thread gathers `per_thread` rows, 8 independent loads in flight (the featurisation has 8 corners in flight per multisample)
template <typename T, int MODE>       // 0 scattered, 1 clustered, 2 windowed
__global__ __launch_bounds__(256) void gather(const T* __restrict__ tab, unsigned rows, int per_thread, unsigned seed, unsigned* __restrict__ out) {
  constexpr bool CLUSTER = MODE == 1;
  const unsigned tid = blockIdx.x * 256 + threadIdx.x;
  const unsigned lane = threadIdx.x & 63, wave = tid >> 6;
  unsigned s = seed ^ ((CLUSTER ? wave : tid) * 2654435761u + 12345u);
  const unsigned wrows = (16u << 20) / sizeof(T), nwin = rows / wrows > 0 ? rows / wrows : 1;
  const unsigned wbase = MODE == 2 ? (blockIdx.x % nwin) * wrows : 0u, wspan = MODE == 2 ? (wrows < rows ? wrows : rows) : rows;
  unsigned acc = 0;
  for (int k = 0; k < per_thread; k += 8) {
    unsigned idx[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      s = s * 1664525u + 1013904223u;
      const unsigned r = (s >> 3) ^ (s << 11);
      idx[q] = CLUSTER ? ((r % (rows - 64u)) & ~63u) + lane : wbase + r % wspan;
    }
    T v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = tab[idx[q]];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc ^= Acc<T>::f(v[q]);
  }
  if (acc == 0x12345678u) out[0] = acc;                    // (keeps the loads alive)
}

__global__ void fill(uint32_t* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = (uint32_t)(i * 2654435761u);
}

template <typename T, int CLUSTER>
void run(const char* pat, const void* tab, size_t bytes, unsigned* out) {
  const unsigned rows = (unsigned)(bytes / sizeof(T));
  const int blocks = 256 * 32, per_thread = 128;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  gather<T, CLUSTER><<<blocks, 256>>>((const T*)tab, rows, 16, 3u, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  gather<T, CLUSTER><<<blocks, 256>>>((const T*)tab, rows, per_thread, 7u, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)blocks * 256 * per_thread;
  printf("%-9s row %2zu B  table %7.1f MB  rows gathered %.0f  %8.3f ms  %7.1f G rows/s  useful %7.1f GB/s\n", pat, sizeof(T), bytes / 1048576.0, n, ms,
         n / ms / 1e6, n * sizeof(T) / ms / 1e6);
}

int main() {
  const size_t sizes[3] = {(size_t)120 << 20, (size_t)229 << 20, (size_t)2048 << 20};
  void* tab = nullptr;
  unsigned* out = nullptr;
  if (hipMalloc(&tab, sizes[2]) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
  fill<<<4096, 256>>>((uint32_t*)tab, sizes[2] / 4);
  hipDeviceSynchronize();
  for (int t = 0; t < 3; ++t) {
    run<uint32_t, 0>("scattered", tab, sizes[t], out);
    run<uint2, 0>("scattered", tab, sizes[t], out);
    run<uint4, 0>("scattered", tab, sizes[t], out);
    run<uint32_t, 1>("clustered", tab, sizes[t], out);
    run<uint2, 1>("clustered", tab, sizes[t], out);
    run<uint4, 1>("clustered", tab, sizes[t], out);
    run<uint32_t, 2>("windowed", tab, sizes[t], out);
    run<uint2, 2>("windowed", tab, sizes[t], out);
    run<uint4, 2>("windowed", tab, sizes[t], out);
  }
  hipFree(tab); hipFree(out);
  return 0;
}
