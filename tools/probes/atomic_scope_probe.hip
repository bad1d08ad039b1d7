// Micro-probe: throughput of scattered fp32 atomic adds on MI355X as a function of memory scope and XCD locality.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics atomic_scope_probe.hip -o atomic_scope_probe && ./atomic_scope_probe
// Workgroup b runs on XCD b % 8.  "partitioned": every workgroup only touches the slice (1/8 of the table) of its own XCD,
// so all accessors of a cache line share one L2.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int SCOPE, bool PART>
__global__ __launch_bounds__(256) void probe(float* __restrict__ tab, long entries, int per_thread, unsigned seed) {
  const int xcd = blockIdx.x & 7;
  unsigned s = seed ^ (blockIdx.x * 9781u + threadIdx.x * 6271u + 12345u);
  const long slice = entries / 8;
  for (int k = 0; k < per_thread; ++k) {
    s = s * 1664525u + 1013904223u;
    unsigned r = s >> 4;
    long idx = PART ? (long)xcd * slice + (long)(r % (unsigned)slice) : (long)(r % (unsigned)entries);
    // 4 consecutive floats (one hash-grid corner with C = 4)
    float* p = tab + idx * 4;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (SCOPE == 0) __hip_atomic_fetch_add(p + c, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else if (SCOPE == 1) __hip_atomic_fetch_add(p + c, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else __hip_atomic_fetch_add(p + c, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
  }
}

__global__ void sum_kernel(const float* t, long n, double* out) {
  double a = 0;
  for (long i = blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) a += t[i];
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, a);
}

template <int SCOPE, bool PART>
void run(const char* name, float* tab, long entries, double* dsum) {
  const int blocks = 8 * 256, per_thread = 256;
  hipMemset(tab, 0, entries * 16);
  hipMemset(dsum, 0, 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  probe<SCOPE, PART><<<blocks, 256>>>(tab, entries, 8, 1u);       // warm-up
  hipMemset(tab, 0, entries * 16);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<SCOPE, PART><<<blocks, 256>>>(tab, entries, per_thread, 7u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  sum_kernel<<<1024, 256>>>(tab, entries * 4, dsum);
  double h = 0;
  hipMemcpy(&h, dsum, 8, hipMemcpyDeviceToHost);
  const double n = (double)blocks * 256 * per_thread * 4;
  printf("%-44s table %6.1f MB: %8.3f ms  %7.2f G atomics/s   sum %s (%.0f of %.0f)\n", name, entries * 16 / 1048576.0, ms, n / ms / 1e6,
         h == n ? "exact" : "MISMATCH", h, n);
}

int main() {
  for (long entries : {1L << 21, 1L << 24}) {       // 32 MB (one hashed level, C = 4 fp32) and 256 MB (eight levels)
    float* tab; double* dsum;
    hipMalloc(&tab, entries * 16); hipMalloc(&dsum, 8);
    run<0, false>("agent scope, whole table", tab, entries, dsum);
    run<0, true>("agent scope, XCD-partitioned", tab, entries, dsum);
    run<1, false>("workgroup scope, whole table (UNSAFE)", tab, entries, dsum);
    run<1, true>("workgroup scope, XCD-partitioned", tab, entries, dsum);
    run<2, true>("wavefront scope, XCD-partitioned", tab, entries, dsum);
    hipFree(tab); hipFree(dsum);
  }
  return 0;
}
