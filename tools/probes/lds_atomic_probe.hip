// Micro-probe: LDS atomic-add rates on gfx950 at the access pattern of zip_bin_accumulate_kernel (1024 threads per workgroup, one
// workgroup per CU, pseudo-random cells of a 128 KB image): 64-bit integer adds (what the kernel issues today), 32-bit integer adds,
// returning 32-bit adds, fp32 adds, and the "lo returns, hi carries" pair that would carry a 64-bit fixed-point sum on 32-bit atomics.
//   hipcc --offload-arch=gfx950 -O3 lds_atomic_probe.hip -o lds_atomic_probe && ./lds_atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CELLS 16384

template <int MODE>
__global__ __launch_bounds__(1024) void probe(int per_thread, unsigned seed, unsigned long long* __restrict__ out) {
  extern __shared__ unsigned long long img[];              // 128 KB
  unsigned* img32 = (unsigned*)img;
  float* imgf = (float*)img;
  for (int k = threadIdx.x; k < CELLS; k += 1024) img[k] = 0;
  __syncthreads();
  unsigned s = seed ^ (blockIdx.x * 9781u + threadIdx.x * 6271u + 12345u);
  for (int k = 0; k < per_thread; ++k) {
    s = s * 1664525u + 1013904223u;
    const unsigned cell = (s >> 9) & (CELLS - 1);
    const long long v = (long long)(int)(s ^ (s << 7)) * 5;     // ~35-bit signed values
    if (MODE == 0) atomicAdd(img + cell, (unsigned long long)v);
    else if (MODE == 1) atomicAdd(img32 + cell, (unsigned)v);                           // non-returning 32-bit
    else if (MODE == 2) { const unsigned old = atomicAdd(img32 + cell, (unsigned)v); s ^= old & 1u; }      // returning 32-bit
    else if (MODE == 3) atomicAdd(imgf + cell, (float)v);                                // fp32 (ds_add_f32)
    else if (MODE == 4) {                                                               // lo (returning) + hi with carry when non-zero
      const unsigned lo = (unsigned)v;
      const unsigned old = atomicAdd(img32 + 2 * (cell & (CELLS / 2 - 1)), lo);
      const int hi = (int)(v >> 32) + ((old + lo) < old ? 1 : 0);
      if (hi != 0) atomicAdd(img32 + 2 * (cell & (CELLS / 2 - 1)) + 1, (unsigned)hi);
    }
  }
  __syncthreads();
  unsigned long long a = 0;
  for (int k = threadIdx.x; k < CELLS; k += 1024) a ^= img[k];
  if (a == 0x123456789abcull) out[0] = a + s;
}

template <int MODE>
void run(const char* name, unsigned long long* out) {
  (void)hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int blocks = 256, per_thread = 4096;
  probe<MODE><<<blocks, 1024, CELLS * 8>>>(64, 1u, out);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  probe<MODE><<<blocks, 1024, CELLS * 8>>>(per_thread, 7u, out);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)blocks * 1024 * per_thread;
  printf("%-52s %8.3f ms  %8.1f G adds/s  = %6.2f per CU per ns\n", name, ms, n / ms / 1e6, n / ms / 1e6 / 256);
}

int main() {
  unsigned long long* out = nullptr;
  if (hipMalloc(&out, 64) != hipSuccess) return 1;
  run<0>("ds_add_u64 (zip_bin_accumulate today)", out);
  run<1>("ds_add_u32", out);
  run<2>("ds_add_rtn_u32", out);
  run<3>("ds_add_f32", out);
  run<4>("64-bit sum as lo (rtn_u32) + hi-with-carry (u32)", out);
  (void)hipFree(out);
  return 0;
}
