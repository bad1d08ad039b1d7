// Debug helper for the race / uninitialised-LDS screens: fills the whole 160 KiB LDS of every CU with a NaN bit pattern, so that any
// kernel that READS LDS it never wrote shows up as NaNs in its output instead of depending on what an earlier process left there.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o liblds_poison.so lds_poison.hip
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(1024) void lds_poison_kernel(unsigned pattern, unsigned* sink) {
  extern __shared__ unsigned lds[];
  for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 1024) lds[i] = pattern;
  __syncthreads();
  if (sink != nullptr && lds[(threadIdx.x * 37) % (160 * 1024 / 4)] != pattern) sink[0] = 1;   // keep the stores alive
}
extern "C" int lds_poison(unsigned pattern, void* stream) {
  static bool set = false;
  if (!set) { (void)hipFuncSetAttribute((const void*)lds_poison_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = true; }
  static unsigned* sink = nullptr;
  if (sink == nullptr) (void)hipMalloc(&sink, 4);
  hipLaunchKernelGGL(lds_poison_kernel, dim3(1024), dim3(1024), 160 * 1024, (hipStream_t)stream, pattern, sink);
  return (int)hipGetLastError();
}
