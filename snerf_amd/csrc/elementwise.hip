// Small streaming kernels around the MLP: fused Adam over a flat parameter
// arena, fp32 column sums for the head bias gradients, dtype casts.
#include "common.h"

// torch.optim.Adam semantics (reference optimiser: s-nerf/utils/model_utils.py:23-34 builds
// torch.optim.Adam over the model parameters): m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
// p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps).  grad_scale folds the 1/world_size
// of the data-parallel mean into the same pass; grads are zeroed for the next step.
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                                                   float bc1, float bc2_sqrt, float grad_scale, int zero_grad) {
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 pp = ((float4*)p)[i], gg = ((float4*)g)[i], mm = ((float4*)m)[i], vv = ((float4*)v)[i];
    float* pa = (float*)&pp; float* ga = (float*)&gg; float* ma = (float*)&mm; float* va = (float*)&vv;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gk = ga[k] * grad_scale;
      ma[k] = b1 * ma[k] + (1.f - b1) * gk;
      va[k] = b2 * va[k] + (1.f - b2) * gk * gk;
      pa[k] -= (lr / bc1) * ma[k] / (sqrtf(va[k]) / bc2_sqrt + eps);
    }
    ((float4*)p)[i] = pp; ((float4*)m)[i] = mm; ((float4*)v)[i] = vv;
    if (zero_grad) ((float4*)g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long i = (n4 << 2) + threadIdx.x;
    const float gk = g[i] * grad_scale;
    m[i] = b1 * m[i] + (1.f - b1) * gk;
    v[i] = b2 * v[i] + (1.f - b2) * gk * gk;
    p[i] -= (lr / bc1) * m[i] / (sqrtf(v[i]) / bc2_sqrt + eps);
    if (zero_grad) g[i] = 0.f;
  }
}

extern "C" int snerf_adam_step(float* p, float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, int step,
                               float grad_scale, int zero_grad, void* stream) {
  if (n <= 0) return SNERF_OK;
  if (step < 1 || (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15)) return SNERF_ERR_ARG;
  const float bc1 = 1.f - powf(b1, (float)step), bc2s = sqrtf(1.f - powf(b2, (float)step));
  const long n4 = n >> 2;
  int blocks = (int)((n4 + 255) / 256);
  blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, b1, b2, eps, bc1, bc2s,
                     grad_scale, zero_grad);
  return snerf_check_launch();
}

// The same update with the step count in device memory (incremented here): nothing in the launch depends on host state, so a whole
// training step can be captured in a hipGraph and replayed (trainer.MipTrainer.capture).
__global__ void adam_tick_kernel(int* step) { *step += 1; }

__global__ __launch_bounds__(256) void adam_dev_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
                                                       float lr, float b1, float b2, float eps, const int* __restrict__ step, float grad_scale,
                                                       int zero_grad) {
  const float t = (float)*step;
  const float bc1 = 1.f - powf(b1, t), bc2_sqrt = sqrtf(1.f - powf(b2, t));
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 pp = ((float4*)p)[i], gg = ((float4*)g)[i], mm = ((float4*)m)[i], vv = ((float4*)v)[i];
    float* pa = (float*)&pp; float* ga = (float*)&gg; float* ma = (float*)&mm; float* va = (float*)&vv;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gk = ga[k] * grad_scale;
      ma[k] = b1 * ma[k] + (1.f - b1) * gk;
      va[k] = b2 * va[k] + (1.f - b2) * gk * gk;
      pa[k] -= (lr / bc1) * ma[k] / (sqrtf(va[k]) / bc2_sqrt + eps);
    }
    ((float4*)p)[i] = pp; ((float4*)m)[i] = mm; ((float4*)v)[i] = vv;
    if (zero_grad) ((float4*)g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long i = (n4 << 2) + threadIdx.x;
    const float gk = g[i] * grad_scale;
    m[i] = b1 * m[i] + (1.f - b1) * gk;
    v[i] = b2 * v[i] + (1.f - b2) * gk * gk;
    p[i] -= (lr / bc1) * m[i] / (sqrtf(v[i]) / bc2_sqrt + eps);
    if (zero_grad) g[i] = 0.f;
  }
}

extern "C" int snerf_adam_step_dev(float* p, float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, int* step_dev,
                                   float grad_scale, int zero_grad, void* stream) {
  if (n <= 0) return SNERF_OK;
  if (step_dev == nullptr || (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15)) return SNERF_ERR_ARG;
  const long n4 = n >> 2;
  int blocks = (int)((n4 + 255) / 256);
  blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
  hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev);
  hipLaunchKernelGGL(adam_dev_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, b1, b2, eps, step_dev, grad_scale, zero_grad);
  return snerf_check_launch();
}

// out[c] += sum_m x[m, c] for a narrow fp32 matrix (head gradients: 3 + 1 columns)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, long ld, long M, int C, float* __restrict__ out) {
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (long m = (long)blockIdx.x * 256 + threadIdx.x; m < M; m += (long)gridDim.x * 256)
    for (int c = 0; c < C; ++c) acc[c] += x[m * ld + c];
  for (int c = 0; c < C; ++c) {
    const float s = wave_sum(acc[c]);
    if ((threadIdx.x & 63) == 0) atomicAdd(out + c, s);
  }
}

extern "C" int snerf_colsum_f32(const float* x, long ld, long M, int C, float* out, void* stream) {
  if (M <= 0) return SNERF_OK;
  if (C < 1 || C > 8) return SNERF_ERR_ARG;
  int blocks = (int)((M + 255) / 256);
  blocks = blocks > 1024 ? 1024 : blocks;
  hipLaunchKernelGGL(colsum_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, ld, M, C, out);
  return snerf_check_launch();
}

// dst[m, c] (T, ld_dst) = src[m, c] (fp32, ld_src) for c < C; zero for C <= c < Cpad.
template <typename T>
__global__ __launch_bounds__(256) void cast_pad_kernel(const float* __restrict__ src, long ld_src, long M, int C, int Cpad, T* __restrict__ dst,
                                                       long ld_dst) {
  const long total = M * Cpad;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long m = e / Cpad;
    const int c = (int)(e - m * Cpad);
    dst[m * ld_dst + c] = from_f32<T>(c < C ? src[m * ld_src + c] : 0.f);
  }
}

extern "C" int snerf_cast_pad(const float* src, long ld_src, long M, int C, int Cpad, void* dst, long ld_dst, int dtype, void* stream) {
  if (M <= 0) return SNERF_OK;
  if (C < 0 || Cpad < C) return SNERF_ERR_ARG;
  const long total = M * Cpad;
  const int blocks = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
  if (dtype == SNERF_DT_F32) hipLaunchKernelGGL(cast_pad_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, ld_src, M, C, Cpad, (float*)dst, ld_dst);
  else hipLaunchKernelGGL(cast_pad_kernel<__bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, ld_src, M, C, Cpad, (__bf16*)dst, ld_dst);
  return snerf_check_launch();
}

extern "C" int snerf_version() { return 1; }
