// Small streaming kernels around the MLP: fused Adam over a flat parameter
// arena, fp32 column sums for the head bias gradients, dtype casts.
#include "common.h"

// torch.optim.Adam semantics (reference optimiser: s-nerf/utils/model_utils.py:23-34 builds
// torch.optim.Adam over the model parameters): m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
// p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps).  grad_scale folds the 1/world_size
// of the data-parallel mean into the same pass; grads are zeroed for the next step.
//
// Gradient hygiene of the zipnerf training loop (s-nerfpp/zipnerf/internal/train_utils.py:234-243, train.py:336), folded into
// the same pass: optional global-norm clip (`clip_coef`: device scalar written by snerf_grad_clip_coef), optional value clip
// (`grad_max_val` > 0: clip_grad_value_), then the non-finite policy `nonfinite`:
//   0 = keep (plain torch.optim.Adam),  1 = NaN / +-Inf -> 0 (one poisoned gradient cannot reach m, v or the parameters),
//   2 = torch.Tensor.nan_to_num_() exactly as the reference calls it (NaN -> 0, +-Inf -> +-FLT_MAX).
struct AdamArgs {
  float lr, b1, b2, eps, grad_scale, grad_max_val;
  int zero_grad, nonfinite;
  const int* step_dev;        // bias-correction step count in device memory (graph capture), or nullptr: `step`
  const float* lr_dev;        // learning rate in device memory (graph capture with a schedule), or nullptr: `lr`
  const float* clip_coef;     // global-norm clip coefficient in device memory, or nullptr
  int step;
  unsigned long long* dropped; // += number of NaN / +-Inf gradient elements this launch saw (any `nonfinite` policy), or nullptr
};

__device__ __forceinline__ float adam_clean_grad(float g, const AdamArgs& a, float coef) {
  g *= a.grad_scale * coef;
  if (a.grad_max_val > 0.f && g == g) g = fminf(fmaxf(g, -a.grad_max_val), a.grad_max_val);   // torch.clamp keeps a NaN (fminf/fmaxf would not)
  if (a.nonfinite == 1) g = (fabsf(g) <= 3.402823466e+38f) ? g : 0.f;                          // false for NaN and Inf
  else if (a.nonfinite == 2) g = (g != g) ? 0.f : fminf(fmaxf(g, -3.402823466e+38f), 3.402823466e+38f);
  return g;
}

// one element's update: every product and sum rounded on its own, so that the vector and the scalar flavour of the kernel (and any
// future one) give the same bits -- a sharded table update must equal the dense one whatever alignment its slices have
__device__ __forceinline__ void adam_update(float& p, float& m, float& v, float gk, float b1, float b2, float step_size, float bc2_sqrt, float eps) {
#pragma clang fp contract(off)
  m = b1 * m + (1.f - b1) * gk;
  v = b2 * v + ((1.f - b2) * gk) * gk;
  p = p - (step_size * m) / (sqrtf(v) / bc2_sqrt + eps);
}

// VEC = 4: all four pointers 16-byte aligned (whole arenas, aligned spans); VEC = 1: any alignment -- a rank's 1 / world slice of a
// single-channel hash table starts at an odd float (6 606 952 / 8 = 825 869 rows per rank), trainer._TableShards.
template <int VEC>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, AdamArgs a) {
  const float t = (float)(a.step_dev != nullptr ? *a.step_dev : a.step);
  const float lr = a.lr_dev != nullptr ? *a.lr_dev : a.lr;
  const float bc1 = 1.f - powf(a.b1, t), bc2_sqrt = sqrtf(1.f - powf(a.b2, t));
  const float coef = a.clip_coef != nullptr ? *a.clip_coef : 1.f;
  const float b1 = a.b1, b2 = a.b2, eps = a.eps;
  const long n4 = VEC == 4 ? (n >> 2) : 0;
  unsigned bad = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 pp = ((float4*)p)[i], gg = ((float4*)g)[i], mm = ((float4*)m)[i], vv = ((float4*)v)[i];
    float* pa = (float*)&pp; float* ga = (float*)&gg; float* ma = (float*)&mm; float* va = (float*)&vv;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      bad += (fabsf(ga[k]) <= 3.402823466e+38f) ? 0u : 1u;
      const float gk = adam_clean_grad(ga[k], a, coef);
      adam_update(pa[k], ma[k], va[k], gk, b1, b2, lr / bc1, bc2_sqrt, eps);
    }
    ((float4*)p)[i] = pp; ((float4*)m)[i] = mm; ((float4*)v)[i] = vv;
    if (a.zero_grad) ((float4*)g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // VEC = 4: the n & 3 tail elements (block 0); VEC = 1: every element, one per thread and stride
  const long t0 = VEC == 4 ? (n4 << 2) + (blockIdx.x == 0 ? (long)threadIdx.x : n) : (long)blockIdx.x * 256 + threadIdx.x;
  for (long i = t0; i < n; i += (long)gridDim.x * 256) {
    bad += (fabsf(g[i]) <= 3.402823466e+38f) ? 0u : 1u;
    const float gk = adam_clean_grad(g[i], a, coef);
    float pi = p[i], mi = m[i], vi = v[i];
    adam_update(pi, mi, vi, gk, b1, b2, lr / bc1, bc2_sqrt, eps);
    p[i] = pi; m[i] = mi; v[i] = vi;
    if (a.zero_grad) g[i] = 0.f;
  }
  if (a.dropped != nullptr && __any(bad != 0)) {        // rare: one atomic per wave that saw a non-finite gradient
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) bad += __shfl_xor(bad, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(a.dropped, (unsigned long long)bad);
  }
}

static int adam_launch(float* p, float* g, float* m, float* v, long n, const AdamArgs& a, void* stream) {
  if (n <= 0) return SNERF_OK;
  if ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 3) || a.nonfinite < 0 || a.nonfinite > 2) return SNERF_ERR_ARG;
  const bool vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
  const long units = vec ? (n >> 2) : n;
  int blocks = (int)((units + 255) / 256);
  blocks = blocks < 1 ? 1 : (blocks > (vec ? 2048 : 8192) ? (vec ? 2048 : 8192) : blocks);
  if (vec) hipLaunchKernelGGL(adam_kernel<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, a);
  else hipLaunchKernelGGL(adam_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, a);
  return snerf_check_launch();
}

extern "C" int snerf_adam_step(float* p, float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, int step,
                               float grad_scale, int zero_grad, void* stream) {
  if (step < 1) return SNERF_ERR_ARG;
  return adam_launch(p, g, m, v, n, AdamArgs{lr, b1, b2, eps, grad_scale, 0.f, zero_grad, 0, nullptr, nullptr, nullptr, step, nullptr}, stream);
}

// The same update with the step count in device memory (incremented here): nothing in the launch depends on host state, so a whole
// training step can be captured in a hipGraph and replayed (trainer.MipTrainer.capture).
__global__ void adam_tick_kernel(int* step) { *step += 1; }

extern "C" int snerf_adam_step_dev(float* p, float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, int* step_dev,
                                   float grad_scale, int zero_grad, void* stream) {
  if (n <= 0) return SNERF_OK;
  if (step_dev == nullptr) return SNERF_ERR_ARG;
  hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev);
  return adam_launch(p, g, m, v, n, AdamArgs{lr, b1, b2, eps, grad_scale, 0.f, zero_grad, 0, step_dev, nullptr, nullptr, 0, nullptr}, stream);
}

extern "C" int snerf_adam_step_ex(float* p, float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, int step,
                                  int* step_dev, const float* lr_dev, float grad_scale, int zero_grad, int nonfinite,
                                  float grad_max_val, const float* clip_coef, void* stream) {
  if (n <= 0) return SNERF_OK;
  if (step_dev == nullptr && step < 1) return SNERF_ERR_ARG;
  if (step_dev != nullptr) hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev);
  return adam_launch(p, g, m, v, n, AdamArgs{lr, b1, b2, eps, grad_scale, grad_max_val, zero_grad, nonfinite, step_dev, lr_dev, clip_coef, step, nullptr}, stream);
}

// snerf_adam_step_ex + `dropped` (device uint64, never reset here): += the number of NaN / +-Inf gradient elements of this launch.  An
// fp16 run with a static loss scale and nonfinite = 1 otherwise drops overflowed gradients silently (GradScaler would skip the step).
extern "C" int snerf_adam_step_cnt(float* p, float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, int step,
                                   int* step_dev, const float* lr_dev, float grad_scale, int zero_grad, int nonfinite,
                                   float grad_max_val, const float* clip_coef, void* dropped, void* stream) {
  if (n <= 0) return SNERF_OK;
  if (step_dev == nullptr && step < 1) return SNERF_ERR_ARG;
  if (dropped != nullptr && ((uintptr_t)dropped & 7)) return SNERF_ERR_ARG;
  if (step_dev != nullptr) hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev);
  return adam_launch(p, g, m, v, n, AdamArgs{lr, b1, b2, eps, grad_scale, grad_max_val, zero_grad, nonfinite, step_dev, lr_dev, clip_coef, step,
                                             (unsigned long long*)dropped}, stream);
}

// Global-norm clip coefficient of torch.nn.utils.clip_grad_norm_ (accelerator.clip_grad_norm_, train_utils.py:236-237):
// coef = min(1, max_norm / (||grad_scale * g||_2 + 1e-6)) over the flat gradient arena, NaN-poisoned like torch's.  Two launches:
// per-block partial sums of squares in a fixed order (deterministic), then one block folds them.
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, long n, double* __restrict__ part) {
  __shared__ double red[4];
  double acc = 0.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) acc += (double)g[i] * (double)g[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void clip_coef_kernel(const double* __restrict__ part, int nb, float grad_scale, float max_norm, float* __restrict__ out) {
  double s = 0.0;
  for (int i = 0; i < nb; ++i) s += part[i];
  const float norm = fabsf(grad_scale) * (float)sqrt(s);
  out[0] = fminf(max_norm / (norm + 1e-6f), 1.f);
  out[1] = norm;
}

// ws: >= 1024 doubles of device scratch; out: float[2] = {coef, norm}
extern "C" int snerf_grad_clip_coef(const float* g, long n, float grad_scale, float max_norm, void* ws, float* out, void* stream) {
  if (n <= 0 || ws == nullptr || out == nullptr || max_norm <= 0.f) return SNERF_ERR_ARG;
  int blocks = (int)((n + 255) / 256);
  blocks = blocks > 1024 ? 1024 : blocks;
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, n, (double*)ws);
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (const double*)ws, blocks, grad_scale, max_norm, out);
  return snerf_check_launch();
}

// found-inf pass of a dynamic loss scaler (torch.cuda.amp.GradScaler.unscale_'s check; the reference trains its fp16 MLPs under
// accelerate's scaler: zipnerf/train.py:44,215,331): *flag |= 1 when any element of the fp32 arena is NaN or +-Inf.  One read of the
// arena (310 MB for waymo.gin: ~60 us); the caller zeroes the flag.
__global__ __launch_bounds__(256) void nonfinite_flag_kernel(const unsigned* __restrict__ g, long n, int* __restrict__ flag) {
  bool bad = false;
  const long head = ((16 - ((size_t)g & 15)) & 15) >> 2;                          // scalar words up to the first 16-byte boundary
  const long nv = n > head ? (n - head) >> 2 : 0;
  const uint4* gv = (const uint4*)(g + (n > head ? head : 0));
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
    const uint4 w = gv[i];
    bad |= ((w.x & 0x7f800000u) == 0x7f800000u) | ((w.y & 0x7f800000u) == 0x7f800000u) | ((w.z & 0x7f800000u) == 0x7f800000u) |
           ((w.w & 0x7f800000u) == 0x7f800000u);
  }
  if (blockIdx.x == 0) {
    const long tail0 = n > head ? head + (nv << 2) : 0;
    for (long i = threadIdx.x; i < (n > head ? head : n); i += 256) bad |= (g[i] & 0x7f800000u) == 0x7f800000u;
    for (long i = tail0 + threadIdx.x; i < n; i += 256) bad |= (g[i] & 0x7f800000u) == 0x7f800000u;
  }
  if (__ballot(bad) != 0 && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}
extern "C" int snerf_nonfinite_flag(const float* g, long n, int* flag, void* stream) {
  if (n < 0 || flag == nullptr || (n > 0 && g == nullptr) || ((size_t)g & 3) != 0) return SNERF_ERR_ARG;
  if (n == 0) return SNERF_OK;
  long blocks = (n / 4 + 255) / 256;
  blocks = blocks > 4096 ? 4096 : (blocks < 1 ? 1 : blocks);
  hipLaunchKernelGGL(nonfinite_flag_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, (const unsigned*)g, n, flag);
  return snerf_check_launch();
}

// out[c] += sum_m x[m, c] for a narrow fp32 matrix (head gradients: 3 + 1 columns)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, long ld, long M, int C, float* __restrict__ out) {
  __shared__ float red[4][8];
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (long m = (long)blockIdx.x * 256 + threadIdx.x; m < M; m += (long)gridDim.x * 256)
    for (int c = 0; c < C; ++c) acc[c] += x[m * ld + c];
  for (int c = 0; c < C; ++c) {
    const float s = wave_sum(acc[c]);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][c] = s;
  }
  __syncthreads();
  // one atomic per workgroup and column (thousands of same-address atomics were the cost of this kernel, not its 6 MB of reads)
  if (threadIdx.x < C) atomicAdd(out + threadIdx.x, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// the same sums by ONE workgroup in a fixed order (deterministic mode: no atomics between workgroups)
__global__ __launch_bounds__(1024) void colsum_det_kernel(const float* __restrict__ x, long ld, long M, int C, float* __restrict__ out) {
  __shared__ float red[16][8];
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (long m = threadIdx.x; m < M; m += 1024)
    for (int c = 0; c < C; ++c) acc[c] += x[m * ld + c];
  for (int c = 0; c < C; ++c) {
    const float s = wave_sum(acc[c]);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][c] = s;
  }
  __syncthreads();
  if (threadIdx.x < C) {
    float s = 0.f;
    for (int w = 0; w < 16; ++w) s += red[w][threadIdx.x];
    out[threadIdx.x] += s;
  }
}

extern "C" int snerf_colsum_f32_det(const float* x, long ld, long M, int C, float* out, void* stream) {
  if (M <= 0) return SNERF_OK;
  if (C < 1 || C > 8) return SNERF_ERR_ARG;
  hipLaunchKernelGGL(colsum_det_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, ld, M, C, out);
  return snerf_check_launch();
}

extern "C" int snerf_colsum_f32(const float* x, long ld, long M, int C, float* out, void* stream) {
  if (M <= 0) return SNERF_OK;
  if (C < 1 || C > 8) return SNERF_ERR_ARG;
  int blocks = (int)((M + 2047) / 2048);                  // >= 8 rows per thread
  blocks = blocks > 512 ? 512 : (blocks < 1 ? 1 : blocks);
  hipLaunchKernelGGL(colsum_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, ld, M, C, out);
  return snerf_check_launch();
}

// out[c] += sum_m x[m, c] for ANY number of columns (the per-ray gradient rows of the GLO modulation: 256 / 512 columns): a thread per
// column (coalesced rows), the rows split over blockIdx.y chunks that meet in fp32 atomics; one chunk = a fixed order (deterministic).
__global__ __launch_bounds__(256) void colsum_wide_kernel(const float* __restrict__ x, long ld, long M, int C, long rows_per_chunk, float* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const long r0 = (long)blockIdx.y * rows_per_chunk, r1 = r0 + rows_per_chunk < M ? r0 + rows_per_chunk : M;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  long r = r0;
  for (; r + 3 < r1; r += 4) { s0 += x[r * ld + c]; s1 += x[(r + 1) * ld + c]; s2 += x[(r + 2) * ld + c]; s3 += x[(r + 3) * ld + c]; }
  for (; r < r1; ++r) s0 += x[r * ld + c];
  const float s = (s0 + s1) + (s2 + s3);
  if (gridDim.y == 1) out[c] += s; else atomicAdd(out + c, s);
}

extern "C" int snerf_colsum_wide_f32(const float* x, long ld, long M, int C, float* out, int deterministic, void* stream) {
  if (M <= 0) return SNERF_OK;
  if (x == nullptr || out == nullptr || C < 1 || ld < C) return SNERF_ERR_ARG;
  long chunks = deterministic ? 1 : (M + 255) / 256;
  chunks = chunks > 256 ? 256 : chunks;
  const long rpc = (M + chunks - 1) / chunks;
  hipLaunchKernelGGL(colsum_wide_kernel, dim3((C + 255) / 256, (unsigned)chunks), dim3(256), 0, (hipStream_t)stream, x, ld, M, C, rpc, out);
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

// 16-bit outputs, Cpad % 8 == 0, 16-byte aligned rows: one thread per 8 consecutive outputs of a row (one 16-byte store)
template <typename T>
__global__ __launch_bounds__(256) void cast_pad8_kernel(const float* __restrict__ src, long ld_src, long M, int C, int G, T* __restrict__ dst,
                                                        long ld_dst) {
  typedef __attribute__((ext_vector_type(8))) T vec8;
  const long total = M * G;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long m = e / G;
    const int c0 = (int)(e - m * G) * 8;
    vec8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (T)(c0 + k < C ? src[m * ld_src + c0 + k] : 0.f);
    *(vec8*)(dst + m * ld_dst + c0) = o;
  }
}

extern "C" int snerf_cast_pad(const float* src, long ld_src, long M, int C, int Cpad, void* dst, long ld_dst, int dtype, void* stream) {
  if (M <= 0) return SNERF_OK;
  if (C < 0 || Cpad < C) return SNERF_ERR_ARG;
  if ((dtype == SNERF_DT_BF16 || dtype == SNERF_DT_F16) && (Cpad % 8) == 0 && (ld_dst % 8) == 0 && (((uintptr_t)dst) & 15) == 0) {
    const long total8 = M * (Cpad / 8);
    const int blocks8 = (int)((total8 + 255) / 256 < 16384 ? (total8 + 255) / 256 : 16384);
    if (dtype == SNERF_DT_BF16) hipLaunchKernelGGL(cast_pad8_kernel<__bf16>, dim3(blocks8), dim3(256), 0, (hipStream_t)stream, src, ld_src, M, C, Cpad / 8, (__bf16*)dst, ld_dst);
    else hipLaunchKernelGGL(cast_pad8_kernel<_Float16>, dim3(blocks8), dim3(256), 0, (hipStream_t)stream, src, ld_src, M, C, Cpad / 8, (_Float16*)dst, ld_dst);
    return snerf_check_launch();
  }
  const long total = M * Cpad;
  const int blocks = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
  if (dtype == SNERF_DT_F32) hipLaunchKernelGGL(cast_pad_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, ld_src, M, C, Cpad, (float*)dst, ld_dst);
  else if (dtype == SNERF_DT_F16) hipLaunchKernelGGL(cast_pad_kernel<_Float16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, ld_src, M, C, Cpad, (_Float16*)dst, ld_dst);
  else hipLaunchKernelGGL(cast_pad_kernel<__bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, ld_src, M, C, Cpad, (__bf16*)dst, ld_dst);
  return snerf_check_launch();
}

// Appearance embedding of the live mip path (s-nerf/model/models.py:63-64 `emb = Embedding(N_vocab, 48)`, :153-159: condition =
// cat([view encoding, emb(rays.app.long())]), tiled per sample like the view encoding, models.py:285-287): row m of the condition block
// gets the embedding row of its ray's image index.  One thread per output element; sample_id as in snerf_mip_viewenc.
template <typename T>
__global__ __launch_bounds__(256) void app_embed_kernel(const float* __restrict__ emb, const float* __restrict__ app, int V, long rows, int S, int dim,
                                                        T* __restrict__ dst, long ld, const int* __restrict__ sample_id) {
  const long total = rows * dim;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long m = e / dim;
    const int c = (int)(e - m * dim);
    const long ray = (sample_id != nullptr ? (long)sample_id[m] : m) / S;
    const float av = app[ray];
    int v = av == av ? (int)av : 0;                         // rays.app.long(): truncation (NaN: row 0; snerf_index_check reports it)
    v = v < 0 ? 0 : (v >= V ? V - 1 : v);
    dst[m * ld + c] = from_f32<T>(emb[(long)v * dim + c]);
  }
}
// d loss / d emb.weight: one wave per ray sums the condition-block gradient of the ray's S samples (lane = embedding column), then one
// atomic per (ray, column) into the row of the ray's image
__global__ __launch_bounds__(256) void app_embed_bwd_kernel(const float* __restrict__ dV, long ld, const float* __restrict__ app, int V, long n_rays, int S,
                                                            int dim, float* __restrict__ g_emb) {
  const int lane = threadIdx.x & 63;
  const long ray = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= n_rays) return;
  const float av = app[ray];
  int v = av == av ? (int)av : 0;
  v = v < 0 ? 0 : (v >= V ? V - 1 : v);
  for (int c = lane; c < dim; c += 64) {
    float s = 0.f;
    for (int i = 0; i < S; ++i) s += dV[(ray * S + i) * ld + c];
    atomicAdd(g_emb + (long)v * dim + c, s);
  }
}
// the same sums in a FIXED order (deterministic mode): one workgroup per table row walks the rays in order and adds the per-ray sums of
// the rays that select its row -- no atomics, bit-reproducible; V x n_rays index reads (a few hundred microseconds at 100 x 4096)
__global__ __launch_bounds__(64) void app_embed_bwd_det_kernel(const float* __restrict__ dV, long ld, const float* __restrict__ app, int V, long n_rays,
                                                               int S, int dim, float* __restrict__ g_emb) {
  const int row = blockIdx.x, lane = threadIdx.x;
  for (int c = lane; c < dim; c += 64) {
    float acc = 0.f;
    for (long ray = 0; ray < n_rays; ++ray) {
      const float av = app[ray];
      int v = av == av ? (int)av : 0;
      v = v < 0 ? 0 : (v >= V ? V - 1 : v);
      if (v != row) continue;
      float s = 0.f;
      for (int i = 0; i < S; ++i) s += dV[(ray * S + i) * ld + c];
      acc += s;
    }
    if (acc != 0.f) g_emb[(long)row * dim + c] += acc;
  }
}
// indices that nn.Embedding would refuse (models.py:153-159 rays.app.long(); zipnerf models.py:131-139 cam_idx): counts entries of a float
// index vector outside [0, n) or not finite into bad[0] -- the host reads the counter later (no sync in the step) and raises
__global__ __launch_bounds__(256) void index_check_kernel(const float* __restrict__ idx, long n, int hi, int* __restrict__ bad) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const float v = idx[e];
  if (!(v == v) || v <= -1.f || v >= (float)hi) atomicAdd(bad, 1);      // (truncation: (-1, 0) maps to row 0 like .long())
}

extern "C" int snerf_app_embed(const float* emb, const float* app, int n_vocab, long n_rays, int S, int dim, void* dst, long ld, int dtype,
                               const int* sample_id, long rows, void* stream) {
  const long M = sample_id != nullptr ? rows : n_rays * S;
  if (M <= 0) return SNERF_OK;
  if (emb == nullptr || app == nullptr || dst == nullptr || n_vocab <= 0 || S <= 0 || dim <= 0 || ld < dim) return SNERF_ERR_ARG;
  const long total = M * dim;
  const int blocks = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
  if (dtype == SNERF_DT_F32) hipLaunchKernelGGL(app_embed_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, emb, app, n_vocab, M, S, dim, (float*)dst, ld, sample_id);
  else if (dtype == SNERF_DT_BF16) hipLaunchKernelGGL(app_embed_kernel<__bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, emb, app, n_vocab, M, S, dim, (__bf16*)dst, ld, sample_id);
  else if (dtype == SNERF_DT_F16) hipLaunchKernelGGL(app_embed_kernel<_Float16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, emb, app, n_vocab, M, S, dim, (_Float16*)dst, ld, sample_id);
  else return SNERF_ERR_ARG;
  return snerf_check_launch();
}

extern "C" int snerf_app_embed_bwd(const float* dV, long ld, const float* app, int n_vocab, long n_rays, int S, int dim, float* g_emb, void* stream) {
  if (n_rays <= 0) return SNERF_OK;
  if (dV == nullptr || app == nullptr || g_emb == nullptr || n_vocab <= 0 || S <= 0 || dim <= 0 || ld < dim) return SNERF_ERR_ARG;
  hipLaunchKernelGGL(app_embed_bwd_kernel, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, (hipStream_t)stream, dV, ld, app, n_vocab, n_rays, S, dim, g_emb);
  return snerf_check_launch();
}

extern "C" int snerf_app_embed_bwd_det(const float* dV, long ld, const float* app, int n_vocab, long n_rays, int S, int dim, float* g_emb, void* stream) {
  if (n_rays <= 0) return SNERF_OK;
  if (dV == nullptr || app == nullptr || g_emb == nullptr || n_vocab <= 0 || S <= 0 || dim <= 0 || ld < dim) return SNERF_ERR_ARG;
  hipLaunchKernelGGL(app_embed_bwd_det_kernel, dim3((unsigned)n_vocab), dim3(64), 0, (hipStream_t)stream, dV, ld, app, n_vocab, n_rays, S, dim, g_emb);
  return snerf_check_launch();
}

extern "C" int snerf_index_check(const float* idx, long n, int n_rows, int* bad, void* stream) {
  if (n <= 0) return SNERF_OK;
  if (idx == nullptr || bad == nullptr || n_rows <= 0) return SNERF_ERR_ARG;
  hipLaunchKernelGGL(index_check_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, idx, n, n_rows, bad);
  return snerf_check_launch();
}

// fp32 rows -> split-bf16 rows in the GEMMs' interleaved layout (gemm.hip, GemmNT::split): logical columns [64 j, 64 j + 64) of
// hi = bf16(v) at physical columns [128 j, 128 j + 64), lo = bf16(v - hi) at [128 j + 64, 128 j + 128); columns C .. Cpad - 1 zero.
// One thread per 8 consecutive logical columns: two 16-byte stores.
__global__ __launch_bounds__(256) void split_cast_kernel(const float* __restrict__ src, long ld_src, long M, int C, int G, __bf16* __restrict__ dst,
                                                         long ld_dst) {
  const long total = M * G;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long m = e / G;
    const int c0 = (int)(e - m * G) * 8;
    bf16x8 h, l;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float v = c0 + k < C ? src[m * ld_src + c0 + k] : 0.f;
      h[k] = (__bf16)v;
      l[k] = (__bf16)(v - (float)h[k]);
    }
    __bf16* d = dst + m * ld_dst + ((c0 >> 6) << 7) + (c0 & 63);
    *(bf16x8*)d = h;
    *(bf16x8*)(d + 64) = l;
  }
}

extern "C" int snerf_split_cast(const float* src, long ld_src, long M, int C, int Cpad, void* dst, long ld_dst, void* stream) {
  if (M <= 0) return SNERF_OK;
  if (src == nullptr || dst == nullptr || C < 0 || Cpad < C || (Cpad % 64) != 0 || (ld_dst % 8) != 0 || ld_dst < 2 * Cpad || (((uintptr_t)dst) & 15))
    return SNERF_ERR_ARG;
  const long total8 = M * (Cpad / 8);
  const int blocks8 = (int)((total8 + 255) / 256 < 16384 ? (total8 + 255) / 256 : 16384);
  hipLaunchKernelGGL(split_cast_kernel, dim3(blocks8), dim3(256), 0, (hipStream_t)stream, src, ld_src, M, C, Cpad / 8, (__bf16*)dst, ld_dst);
  return snerf_check_launch();
}

// Weight packing as ONE gather: dst[i] = idx[i] >= 0 ? flat[idx[i]] : (idx[i] == -2 ? 1 : 0), rounded to dst's type.  An index with
// bit 30 set selects the LOW part of the split-bf16 pair of flat[idx & 0x3fffffff]: bf16(x - bf16(x)) (bf16 destinations).  The index image
// of every packed operand (padded / transposed / K-concatenated / MFMA-fragment-ordered copies of the parameters) depends only on the
// network's structure, so the host builds it once and refreshes all operands of a network with this launch after every optimiser step
// instead of ~60 slice copies.
template <typename T>
__device__ __forceinline__ T gather_one(const float* __restrict__ flat, int k) {
  if (k < 0) return from_f32<T>(k == -2 ? 1.f : 0.f);
  const float v = flat[k & 0x3fffffff];
  if (k & 0x40000000) return from_f32<T>(v - (float)(__bf16)v);
  return from_f32<T>(v);
}
template <typename T>
__global__ __launch_bounds__(256) void gather_pack_kernel(const float* __restrict__ flat, const int* __restrict__ idx, long n, T* __restrict__ dst) {
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
    if (i + 4 <= n) {
      const int4 k = *(const int4*)(idx + i);
      const int kk[4] = {k.x, k.y, k.z, k.w};
      T o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = gather_one<T>(flat, kk[e]);
#pragma unroll
      for (int e = 0; e < 4; ++e) dst[i + e] = o[e];
    } else {
      for (long j = i; j < n; ++j) dst[j] = gather_one<T>(flat, idx[j]);
    }
  }
}

extern "C" int snerf_gather_pack(const float* flat, const int* idx, long n, void* dst, int dtype, void* stream) {
  if (n <= 0) return SNERF_OK;
  if (flat == nullptr || idx == nullptr || dst == nullptr || (((uintptr_t)idx) & 15)) return SNERF_ERR_ARG;
  const long want = (n + 1023) / 1024;
  const int blocks = (int)(want < 4096 ? want : 4096);
  if (dtype == SNERF_DT_F32) hipLaunchKernelGGL(gather_pack_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, flat, idx, n, (float*)dst);
  else if (dtype == SNERF_DT_BF16) hipLaunchKernelGGL(gather_pack_kernel<__bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, flat, idx, n, (__bf16*)dst);
  else if (dtype == SNERF_DT_F16) hipLaunchKernelGGL(gather_pack_kernel<_Float16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, flat, idx, n, (_Float16*)dst);
  else return SNERF_ERR_ARG;
  return snerf_check_launch();
}

// The same refresh with the TRANSPOSED images taken out of the element-wise gather.  A packed W^T (the data-gradient GEMMs' operand) reads the
// arena at a stride of one weight row per destination element: 64 lanes x 4 scalar loads = 256 L2 requests per 256 elements, which is what the
// refresh of a 1024-wide network cost (75 us of the 512-ray step's 4.3 ms).  The host finds the 16 x 64 destination tiles with
// dst[r0 + i, c0 + j] = flat[base + j * stride + i] (base, stride multiples of 4: 16-byte loads) once per plan; here one wave takes a tile: lane j
// loads its 16 consecutive source floats (one 64-byte line) and the wave writes 16 rows of 64 consecutive destinations.  The tiles' elements carry
// idx = -3 in the gather map (skipped there).  Blocks [0, tile_blocks) take four tiles each, the rest run the element-wise gather.
struct PackTile { int dst_off, src_base, src_stride, dst_ld; };

// element-wise part: block b of nb walks the map four elements per thread; idx = -3 marks what the tile part writes
template <typename T>
__device__ __forceinline__ void gather_pack_span(const float* __restrict__ flat, const int* __restrict__ idx, long n, T* __restrict__ dst, long b, long nb) {
  for (long i = (b * 256 + threadIdx.x) * 4; i < n; i += nb * 1024) {
    if (i + 4 <= n) {
      const int4 k = *(const int4*)(idx + i);
      const int kk[4] = {k.x, k.y, k.z, k.w};
      if (kk[0] == -3 && kk[1] == -3 && kk[2] == -3 && kk[3] == -3) continue;   // (four tile elements: nothing to do)
      T o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = kk[e] == -3 ? from_f32<T>(0.f) : gather_one<T>(flat, kk[e]);
#pragma unroll
      for (int e = 0; e < 4; ++e) if (kk[e] != -3) dst[i + e] = o[e];
    } else {
      for (long j = i; j < n; ++j) if (idx[j] != -3) dst[j] = gather_one<T>(flat, idx[j]);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gather_pack_tiles_kernel(const float* __restrict__ flat, const int* __restrict__ idx, long n, T* __restrict__ dst,
                                                                const PackTile* __restrict__ tiles, int n_tiles, int tile_blocks, int gather_blocks,
                                                                const int* __restrict__ idx_b, long n_b, float* __restrict__ dst_b) {
  if ((int)blockIdx.x < tile_blocks) {
    const int t = (int)blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= n_tiles) return;
    const PackTile pt = tiles[t];
    const f32x4* src = (const f32x4*)(flat + pt.src_base + (long)lane * pt.src_stride);
    f32x4 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = src[q];
    T* d = dst + pt.dst_off + lane;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) d[(long)(q * 4 + e) * pt.dst_ld] = from_f32<T>(v[q][e]);
    return;
  }
  long b = (long)blockIdx.x - tile_blocks;
  if (b >= gather_blocks) {
    // the network's fp32 pool (biases, fused-kernel tables: small) rides in the same launch: blocks past the 16-bit pool's share
    gather_pack_span<float>(flat, idx_b, n_b, dst_b, b - gather_blocks, (long)gridDim.x - tile_blocks - gather_blocks);
    return;
  }
  gather_pack_span<T>(flat, idx, n, dst, b, gather_blocks);
}

static int gather_pack_launch(const float* flat, const int* idx, long n, void* dst, int dtype, const int* tiles, int n_tiles, const int* idx_b, long n_b,
                              float* dst_b, void* stream) {
  if (n <= 0 && n_b <= 0) return SNERF_OK;
  if (n <= 0 || flat == nullptr || idx == nullptr || dst == nullptr || (((uintptr_t)idx) & 15) || (((uintptr_t)flat) & 15) || n_tiles < 0 ||
      (n_tiles > 0 && (tiles == nullptr || (((uintptr_t)tiles) & 15)))) return SNERF_ERR_ARG;
  if (n_b < 0 || (n_b > 0 && (idx_b == nullptr || dst_b == nullptr || (((uintptr_t)idx_b) & 15)))) return SNERF_ERR_ARG;
  const long want = (n + 1023) / 1024, want_b = (n_b + 1023) / 1024;
  const int gblocks = (int)(want < 4096 ? want : 4096), tblocks = (n_tiles + 3) / 4, bblocks = (int)(want_b < 1024 ? want_b : 1024);
  const dim3 g((unsigned)(gblocks + tblocks + bblocks)), b(256);
  const PackTile* tl = (const PackTile*)tiles;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == SNERF_DT_F32) hipLaunchKernelGGL(gather_pack_tiles_kernel<float>, g, b, 0, s, flat, idx, n, (float*)dst, tl, n_tiles, tblocks, gblocks, idx_b, n_b, dst_b);
  else if (dtype == SNERF_DT_BF16) hipLaunchKernelGGL(gather_pack_tiles_kernel<__bf16>, g, b, 0, s, flat, idx, n, (__bf16*)dst, tl, n_tiles, tblocks, gblocks, idx_b, n_b, dst_b);
  else if (dtype == SNERF_DT_F16) hipLaunchKernelGGL(gather_pack_tiles_kernel<_Float16>, g, b, 0, s, flat, idx, n, (_Float16*)dst, tl, n_tiles, tblocks, gblocks, idx_b, n_b, dst_b);
  else return SNERF_ERR_ARG;
  return snerf_check_launch();
}

extern "C" int snerf_gather_pack_tiles(const float* flat, const int* idx, long n, void* dst, int dtype, const int* tiles, int n_tiles, void* stream) {
  return gather_pack_launch(flat, idx, n, dst, dtype, tiles, n_tiles, nullptr, 0, nullptr, stream);
}

extern "C" int snerf_gather_pack_pair(const float* flat, const int* idx, long n, void* dst, int dtype, const int* tiles, int n_tiles, const int* idx32,
                                      long n32, float* dst32, void* stream) {
  return gather_pack_launch(flat, idx, n, dst, dtype, tiles, n_tiles, idx32, n32, dst32, stream);
}

extern "C" int snerf_version() { return 1; }
int g_snerf_last_hip_error = 0;
extern "C" int snerf_last_hip_error() { return g_snerf_last_hip_error; }

// Test utility: fill the LDS of every CU with a pseudo-random pattern (seeded).  LDS content survives from one kernel to the next on a CU, so
// a kernel that reads a location before its own write / DMA into it has landed normally re-reads what its previous launch left there and
// every reproducibility loop passes.  tests/test_stale_lds.py runs whole steps with this kernel in front of every launch, twice with different
// seeds: any difference is such a read (a missing initialisation shows every time, a missing ORDERING -- the K = 128 race of gemm_nt8p_kernel,
// round 4 -- when it fires).
__global__ __launch_bounds__(1024) void lds_scribble_kernel(unsigned seed, unsigned* sink) {
  extern __shared__ unsigned scribble[];
  unsigned h = seed * 2654435761u + blockIdx.x * 40503u + 17u;
  for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 1024) {
    h = (h ^ (unsigned)i) * 1664525u + 1013904223u;
    scribble[i] = (h >> 7) ^ (h << 13);               // any bit pattern: NaNs and infinities included
  }
  __syncthreads();
  if (scribble[(seed + threadIdx.x) % (160 * 1024 / 4)] == 0x5eed5eedu && sink != nullptr) sink[0] = seed;   // (keeps the stores alive)
}
extern "C" int snerf_debug_lds_scribble(int seed, void* stream) {
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)lds_scribble_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
  hipLaunchKernelGGL(lds_scribble_kernel, dim3(4096), dim3(1024), 160 * 1024, (hipStream_t)stream, (unsigned)seed, (unsigned*)nullptr);
  return snerf_check_launch();
}

// fp32 rows -> the fp16 + fp8 split layout of the GEMMs (gemm.hip, GemmNT::split == 2; dtype SNERF_DT_F16F8): per 64 logical columns 256 bytes =
// [fp16(x) x 64 | 64 + 64 e4m3 bytes].  Activations (weight = 0): e4m3((x - fp16(x)) 2^13) then e4m3(x 2^2); weights (weight = 1): e4m3(w 2^9) then
// e4m3((w - fp16(w)) 2^20) -- the activation's residual meets the weight's value and the other way round in one e4m3 tile; values beyond the
// format's +-448 saturate.  Columns C .. Cpad - 1 zero.  One thread per 8 consecutive logical columns: one 16-byte and two 8-byte stores.
__global__ __launch_bounds__(256) void split8_cast_kernel(const float* __restrict__ src, long ld_src, long M, int C, int G, _Float16* __restrict__ dst,
                                                          long ld_dst, int weight) {
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const long total = M * G;
  const float mul_r = weight ? SNERF_F8_W_RES : SNERF_F8_ACT_RES, mul_x = weight ? SNERF_F8_W_VAL : SNERF_F8_ACT_VAL;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long m = e / G;
    const int c0 = (int)(e - m * G) * 8;
    f16x8 h;
    float x[8], r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      x[k] = c0 + k < C ? src[m * ld_src + c0 + k] : 0.f;
      h[k] = (_Float16)x[k];
      r[k] = x[k] - (float)h[k];
    }
    _Float16* d = dst + m * ld_dst + ((c0 >> 6) << 7) + (c0 & 63);
    *(f16x8*)d = h;
    char* b8 = (char*)(dst + m * ld_dst + ((c0 >> 6) << 7) + 64) + (c0 & 63);
    const u32x2 rr = {snerf_e4m3x4(r[0], r[1], r[2], r[3], mul_r), snerf_e4m3x4(r[4], r[5], r[6], r[7], mul_r)};
    const u32x2 xx = {snerf_e4m3x4(x[0], x[1], x[2], x[3], mul_x), snerf_e4m3x4(x[4], x[5], x[6], x[7], mul_x)};
    *(u32x2*)(b8 + (weight ? 64 : 0)) = rr;
    *(u32x2*)(b8 + (weight ? 0 : 64)) = xx;
  }
}

extern "C" int snerf_split8_cast(const float* src, long ld_src, long M, int C, int Cpad, void* dst, long ld_dst, int weight, void* stream) {
  if (M <= 0) return SNERF_OK;
  if (src == nullptr || dst == nullptr || C < 0 || Cpad < C || (Cpad % 64) != 0 || (ld_dst % 8) != 0 || ld_dst < 2 * Cpad || (((uintptr_t)dst) & 15))
    return SNERF_ERR_ARG;
  const long total8 = M * (Cpad / 8);
  const int blocks8 = (int)((total8 + 255) / 256 < 16384 ? (total8 + 255) / 256 : 16384);
  hipLaunchKernelGGL(split8_cast_kernel, dim3(blocks8), dim3(256), 0, (hipStream_t)stream, src, ld_src, M, C, Cpad / 8, (_Float16*)dst, ld_dst, weight);
  return snerf_check_launch();
}
