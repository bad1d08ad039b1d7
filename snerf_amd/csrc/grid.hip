// Multiresolution hash-grid encoder of S-NeRF++ / zipnerf for gfx950.
//
// Replaces the reference's only native code, s-nerfpp/zipnerf/gridencoder/src/gridencoder.cu:
//   snerf_grid_encode_fwd  <- kernel_grid              (:87-245)   forward (+ optional dy_dx)
//   snerf_grid_encode_bwd  <- kernel_grid_backward     (:248-340)  scatter-add into the table gradient
//                             kernel_input_backward    (:343-369)  gradient w.r.t. the coordinates
//   snerf_grid_tv_grad     <- kernel_grad_tv           (:506-610)  total-variation gradient
// with the same argument order as src/bindings.cpp:5-9 / gridencoder.h:12-15, plus explicit output strides
// (so the [B, L*C] layout the MLP wants can be written directly instead of [L,B,C] + permute, grid.py:47,57)
// and an explicit stream.
//
// Bound: HBM/MALL gather bandwidth (8 corners x L levels per point; SURVEY.md section 8d: 344 KB/ray).  One thread
// per (point, level); a block covers 256 consecutive points of ONE level and the grid is level-major, so the
// blocks in flight at any time gather from one level's table (dense levels stay L2-resident, hashed levels
// (<= 16 MB each) stay Infinity-Cache resident); each corner is fetched with ONE 2/4/8/16-byte load of all C channels.
#include "common.h"
#include <hip/hip_fp16.h>

#define SNERF_DT_F16 2

__device__ __forceinline__ uint32_t grid_fast_hash3(const uint32_t* p, int D) {
  const uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};
  uint32_t r = 0;
  for (int i = 0; i < D; ++i) r ^= p[i] * primes[i];
  return r;
}

template <int D>
__device__ __forceinline__ uint32_t grid_index(int gridtype, bool align, uint32_t hashmap_size, uint32_t resolution, const uint32_t* pg) {
  uint32_t stride = 1, index = 0;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    if (stride <= hashmap_size) {
      index += pg[d] * stride;
      stride *= align ? resolution : (resolution + 1);
    }
  }
  if (gridtype == 0 && stride > hashmap_size) index = grid_fast_hash3(pg, D);
  return index % hashmap_size;
}

// grid position of one coordinate on a level (gridencoder.cu:148): product and sum rounded separately -- like the CPU oracle's fp32 ops
// and like zip.hip's zip_cell, so that the corner-cached forward and the binned table gradient (zip.hip: g3_*) land in the same cell
// with the same fractions as the kernels of this file
__device__ __forceinline__ void grid_cell(float x, float scale, float shift, uint32_t* pg, float* fr) {
#pragma clang fp contract(off)
  const float ps = x * scale + shift;
  const float fl = floorf(ps);
  *pg = (uint32_t)fl;
  *fr = ps - fl;
}

template <typename T, int C> struct alignas(sizeof(T) * C) VecC { T v[C]; };

// arithmetic type of the channel values: the reference accumulates in the table's scalar type (results[ch] += w * grid[...] with a
// float weight, gridencoder.cu:137-147); half tables are accumulated in fp32 here (a superset of its precision), double stays double
template <typename T> struct Acc { typedef float type; };
template <> struct Acc<double> { typedef double type; };

template <typename T, int C>
__device__ __forceinline__ void load_c(const T* p, typename Acc<T>::type* out) {
  // one aligned vector load of the C channels of a table row (rows are C * sizeof(T) aligned)
  VecC<T, C> r = *reinterpret_cast<const VecC<T, C>*>(p);
#pragma unroll
  for (int c = 0; c < C; ++c) out[c] = (typename Acc<T>::type)r.v[c];
}

struct GridArgs {
  const float* inputs; const void* table; const int* offsets; void* out; void* dy_dx;
  long so_l, so_b;                 // output strides (elements) for level and point
  int B, L; float S; int H; int gridtype, align, interp;
};

template <typename T, int D, int C>
__global__ __launch_bounds__(256) void grid_fwd_kernel(GridArgs a) {
#pragma clang fp contract(off)        // products and sums rounded one by one, like the CPU oracle and like zip.hip's g3_fwd_kernel (bit-identical outputs)
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= a.B) return;
  const int level = blockIdx.y;
  const T* tab = (const T*)a.table + (long)a.offsets[level] * C;
  T* out = (T*)a.out + level * a.so_l + (long)b * a.so_b;
  float x[D];
  bool oob = false;
#pragma unroll
  for (int d = 0; d < D; ++d) { x[d] = a.inputs[(long)b * D + d]; oob |= (x[d] < 0.f || x[d] > 1.f); }
  T* dd = a.dy_dx ? (T*)a.dy_dx + ((long)b * a.L + level) * D * C : nullptr;
  if (oob) {
#pragma unroll
    for (int c = 0; c < C; ++c) out[c] = (T)0.f;
    if (dd) for (int i = 0; i < D * C; ++i) dd[i] = (T)0.f;
    return;
  }
  const uint32_t hs = a.offsets[level + 1] - a.offsets[level];
  const float scale = exp2f(level * a.S) * a.H - 1.0f;
  const uint32_t res = (uint32_t)ceilf(scale) + 1;
  float pos[D], pder[D];
  uint32_t pg[D];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    float p;
    grid_cell(x[d], scale, a.align ? 0.f : 0.5f, &pg[d], &p);
    if (a.interp == 1) { pder[d] = 6.f * p * (1.f - p); p = p * p * (3.f - 2.f * p); } else pder[d] = 1.f;
    pos[d] = p;
  }
  typedef typename Acc<T>::type A;
  A acc[C];
#pragma unroll
  for (int c = 0; c < C; ++c) acc[c] = 0.f;
#pragma unroll
  for (int idx = 0; idx < (1 << D); ++idx) {
    float w = 1.f;
    uint32_t pl[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (idx & (1 << d)) { w *= pos[d]; pl[d] = pg[d] + 1; } else { w *= 1.f - pos[d]; pl[d] = pg[d]; }
    }
    A v[C];
    load_c<T, C>(tab + (long)grid_index<D>(a.gridtype, a.align, hs, res, pl) * C, v);
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] += w * v[c];
  }
  VecC<T, C> o;
#pragma unroll
  for (int c = 0; c < C; ++c) o.v[c] = (T)acc[c];
  if ((a.so_b % C) == 0 && (a.so_l % C) == 0) *reinterpret_cast<VecC<T, C>*>(out) = o;
  else {
#pragma unroll
    for (int c = 0; c < C; ++c) out[c] = o.v[c];
  }
  if (dd) {
#pragma unroll
    for (int gd = 0; gd < D; ++gd) {
      A g[C];
#pragma unroll
      for (int c = 0; c < C; ++c) g[c] = 0.f;
#pragma unroll
      for (int idx = 0; idx < (1 << (D - 1)); ++idx) {
        float w = scale;
        uint32_t pl[D];
#pragma unroll
        for (int nd = 0; nd < D - 1; ++nd) {
          const int d = nd >= gd ? nd + 1 : nd;
          if (idx & (1 << nd)) { w *= pos[d]; pl[d] = pg[d] + 1; } else { w *= 1.f - pos[d]; pl[d] = pg[d]; }
        }
        A vl[C], vr[C];
        pl[gd] = pg[gd];
        load_c<T, C>(tab + (long)grid_index<D>(a.gridtype, a.align, hs, res, pl) * C, vl);
        pl[gd] = pg[gd] + 1;
        load_c<T, C>(tab + (long)grid_index<D>(a.gridtype, a.align, hs, res, pl) * C, vr);
#pragma unroll
        for (int c = 0; c < C; ++c) g[c] += w * (vr[c] - vl[c]) * pder[gd];
      }
#pragma unroll
      for (int c = 0; c < C; ++c) dd[gd * C + c] = (T)g[c];
    }
  }
}

// ---- backward: table gradient --------------------------------------------------------------------------------
struct GridBwdArgs {
  const void* grad; const float* inputs; const int* offsets; void* grad_table;
  long sg_l, sg_b;
  int B, L; float S; int H; int gridtype, align, interp;
};

__device__ __forceinline__ void atomic_add_T(float* p, float v) { atomicAdd(p, v); }
__device__ __forceinline__ void atomic_add_T(double* p, double v) { atomicAdd(p, v); }
__device__ __forceinline__ void atomic_add_T(__half* p, float v) {
  // single fp16 atomic (only C == 1 reaches here; even C uses the packed form below)
  unsigned int* base = (unsigned int*)((uintptr_t)p & ~(uintptr_t)3);
  const bool hi = ((uintptr_t)p & 2) != 0;
  unsigned int old = *base, assumed;
  do {
    assumed = old;
    __half cur = __ushort_as_half((unsigned short)(hi ? (assumed >> 16) : (assumed & 0xffffu)));
    const unsigned short nv = __half_as_ushort(__float2half(__half2float(cur) + v));
    const unsigned int repl = hi ? ((assumed & 0xffffu) | ((unsigned int)nv << 16)) : ((assumed & 0xffff0000u) | nv);
    old = atomicCAS(base, assumed, repl);
  } while (old != assumed);
}

template <typename T, int D, int C>
__global__ __launch_bounds__(256) void grid_bwd_kernel(GridBwdArgs a) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= a.B) return;
  const int level = blockIdx.y;
  float x[D];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    x[d] = a.inputs[(long)b * D + d];
    if (x[d] < 0.f || x[d] > 1.f) return;   // the gradient buffer arrives zeroed
  }
  T* gt = (T*)a.grad_table + (long)a.offsets[level] * C;
  const T* gin = (const T*)a.grad + level * a.sg_l + (long)b * a.sg_b;
  const uint32_t hs = a.offsets[level + 1] - a.offsets[level];
  const float scale = exp2f(level * a.S) * a.H - 1.0f;
  const uint32_t res = (uint32_t)ceilf(scale) + 1;
  float pos[D];
  uint32_t pg[D];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    float p;
    grid_cell(x[d], scale, a.align ? 0.f : 0.5f, &pg[d], &p);
    if (a.interp == 1) p = p * p * (3.f - 2.f * p);
    pos[d] = p;
  }
  typename Acc<T>::type g[C];
#pragma unroll
  for (int c = 0; c < C; ++c) g[c] = (typename Acc<T>::type)gin[c];
#pragma unroll
  for (int idx = 0; idx < (1 << D); ++idx) {
    float w = 1.f;
    uint32_t pl[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (idx & (1 << d)) { w *= pos[d]; pl[d] = pg[d] + 1; } else { w *= 1.f - pos[d]; pl[d] = pg[d]; }
    }
    T* dst = gt + (long)grid_index<D>(a.gridtype, a.align, hs, res, pl) * C;
    if constexpr (sizeof(T) == 2 && (C % 2) == 0) {
#pragma unroll
      for (int c = 0; c < C; c += 2) unsafeAtomicAdd((__half2*)(dst + c), __floats2half2_rn(w * g[c], w * g[c + 1]));
    } else {
#pragma unroll
      for (int c = 0; c < C; ++c) atomic_add_T(dst + c, w * g[c]);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void grid_input_bwd_kernel(const T* __restrict__ grad, long sg_l, long sg_b, const T* __restrict__ dy_dx,
                                                             T* __restrict__ grad_inputs, int B, int D, int C, int L) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long)B * D) return;
  const long b = t / D;
  const int d = (int)(t - b * D);
  const T* dd = dy_dx + b * L * D * C;
  typedef typename Acc<T>::type A;
  A r = 0.f;
  for (int l = 0; l < L; ++l)
    for (int c = 0; c < C; ++c) r += (A)grad[l * sg_l + b * sg_b + c] * (A)dd[(l * D + d) * C + c];
  grad_inputs[t] = (T)r;
}

// ---- total variation ------------------------------------------------------------------------------------------
template <typename T, int D, int C>
__global__ __launch_bounds__(256) void grid_tv_kernel(const float* __restrict__ inputs, const T* __restrict__ table, T* __restrict__ grad,
                                                      const int* __restrict__ offsets, float weight, int B, float S, int H, int gridtype,
                                                      int align) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  const int level = blockIdx.y;
  float x[D];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    x[d] = inputs[(long)b * D + d];
    if (x[d] < 0.f || x[d] > 1.f) return;
  }
  const T* tab = table + (long)offsets[level] * C;
  T* gt = grad + (long)offsets[level] * C;
  const uint32_t hs = offsets[level + 1] - offsets[level];
  const float scale = exp2f(level * S) * H - 1.0f;
  const uint32_t res = (uint32_t)ceilf(scale) + 1;
  uint32_t pg[D];
#pragma unroll
  for (int d = 0; d < D; ++d) { float fr_unused; grid_cell(x[d], scale, align ? 0.f : 0.5f, &pg[d], &fr_unused); }
  const uint32_t index = grid_index<D>(gridtype, align, hs, res, pg);
  typedef typename Acc<T>::type A;
  A v0[C], results[C], idelta[C];
  load_c<T, C>(tab + (long)index * C, v0);
#pragma unroll
  for (int c = 0; c < C; ++c) { results[c] = 0.f; idelta[c] = 0.f; }
  const float w = weight / (2 * D);
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const uint32_t cur = pg[d];
    A vn[C];
    if (cur < res) {
      pg[d] = cur + 1;
      load_c<T, C>(tab + (long)grid_index<D>(gridtype, align, hs, res, pg) * C, vn);
#pragma unroll
      for (int c = 0; c < C; ++c) { const A gv = v0[c] - vn[c]; results[c] += gv; idelta[c] += gv * gv; }
    }
    if (cur > 0) {
      pg[d] = cur - 1;
      load_c<T, C>(tab + (long)grid_index<D>(gridtype, align, hs, res, pg) * C, vn);
#pragma unroll
      for (int c = 0; c < C; ++c) { const A gv = v0[c] - vn[c]; results[c] += gv; idelta[c] += gv * gv; }
    }
    pg[d] = cur;
  }
#pragma unroll
  for (int c = 0; c < C; ++c) {
    if constexpr (sizeof(T) == 8) atomic_add_T(gt + (long)index * C + c, (double)w * results[c] * rsqrt(idelta[c] + 1e-9));
    else atomic_add_T(gt + (long)index * C + c, w * results[c] * rsqrtf(idelta[c] + 1e-9f));
  }
}

// ---- dispatch ---------------------------------------------------------------------------------------------------
#define GRID_DISPATCH_C(T, D, FN, ...)                                      \
  switch (C) {                                                              \
    case 1: FN<T, D, 1> __VA_ARGS__; break;                                 \
    case 2: FN<T, D, 2> __VA_ARGS__; break;                                 \
    case 4: FN<T, D, 4> __VA_ARGS__; break;                                 \
    case 8: FN<T, D, 8> __VA_ARGS__; break;                                 \
    default: return SNERF_ERR_ARG;                                          \
  }
#define GRID_DISPATCH_D(T, FN, ...)                                         \
  switch (D) {                                                              \
    case 2: { GRID_DISPATCH_C(T, 2, FN, __VA_ARGS__) } break;               \
    case 3: { GRID_DISPATCH_C(T, 3, FN, __VA_ARGS__) } break;               \
    case 4: { GRID_DISPATCH_C(T, 4, FN, __VA_ARGS__) } break;               \
    case 5: { GRID_DISPATCH_C(T, 5, FN, __VA_ARGS__) } break;               \
    default: return SNERF_ERR_ARG;                                          \
  }
// D = 2..5 and float / double / half: the instantiations of the reference (gridencoder.cu:376-399, AT_DISPATCH_FLOATING_TYPES_AND_HALF)
#define GRID_DISPATCH(FN, ...)                                              \
  if (dtype == SNERF_DT_F32) { GRID_DISPATCH_D(float, FN, __VA_ARGS__) }    \
  else if (dtype == SNERF_DT_F16) { GRID_DISPATCH_D(__half, FN, __VA_ARGS__) } \
  else if (dtype == SNERF_DT_F64) { GRID_DISPATCH_D(double, FN, __VA_ARGS__) } \
  else return SNERF_ERR_ARG;

template <typename T, int D, int C> static void launch_fwd(const GridArgs& a, hipStream_t s) {
  hipLaunchKernelGGL((grid_fwd_kernel<T, D, C>), dim3((a.B + 255) / 256, a.L), dim3(256), 0, s, a);
}
template <typename T, int D, int C> static void launch_bwd(const GridBwdArgs& a, hipStream_t s) {
  hipLaunchKernelGGL((grid_bwd_kernel<T, D, C>), dim3((a.B + 255) / 256, a.L), dim3(256), 0, s, a);
}
template <typename T, int D, int C> static void launch_tv(const float* in, const void* tab, void* grad, const int* off, float w, int B, int L,
                                                          float S, int H, int gt, int al, hipStream_t s) {
  hipLaunchKernelGGL((grid_tv_kernel<T, D, C>), dim3((B + 255) / 256, L), dim3(256), 0, s, in, (const T*)tab, (T*)grad, off, w, B, S, H, gt, al);
}

int g3_fwd_launch(const float* inputs, const void* table, const int* offsets, void* outputs, long B, int C, int L, float S, int H, int dtype,
                  long s_l, long s_b, hipStream_t s);                       // zip.hip
static int grid_fwd_entry(const float* inputs, const void* embeddings, const int* offsets, void* outputs, int B, int D, int C,
                          int L, float S, int H, void* dy_dx, int gridtype, int align_corners, int interp, int dtype,
                          long out_stride_l, long out_stride_b, void* stream, bool fast) {
  if (B <= 0) return SNERF_OK;
  if (L <= 0 || inputs == nullptr || embeddings == nullptr || offsets == nullptr || outputs == nullptr) return SNERF_ERR_ARG;
  // D = 3, hash, linear, float / half, no dy_dx (what zipnerf constructs, at every channel count): the pair-loading gather of zip.hip
  if (fast && D == 3 && (C == 1 || C == 2 || C == 4 || C == 8) && gridtype == 0 && !align_corners && interp == 0 && dy_dx == nullptr &&
      (dtype == SNERF_DT_F32 || dtype == SNERF_DT_F16))
    return g3_fwd_launch(inputs, embeddings, offsets, outputs, B, C, L, S, H, dtype, out_stride_l, out_stride_b, (hipStream_t)stream);
  GridArgs a{inputs, embeddings, offsets, outputs, dy_dx, out_stride_l, out_stride_b, B, L, S, H, gridtype, align_corners, interp};
  GRID_DISPATCH(launch_fwd, (a, (hipStream_t)stream))
  return snerf_check_launch();
}

extern "C" int snerf_grid_encode_fwd(const float* inputs, const void* embeddings, const int* offsets, void* outputs, int B, int D, int C,
                                     int L, float S, int H, void* dy_dx, int gridtype, int align_corners, int interp, int dtype,
                                     long out_stride_l, long out_stride_b, void* stream) {
  return grid_fwd_entry(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners, interp, dtype, out_stride_l, out_stride_b, stream, true);
}
// the same operator in kernel_grid's own form (one thread per (point, level), eight independent row loads) for EVERY instantiation: the
// A/B partner of the measurement legs and the parity tests -- a second entry point instead of a process-wide switch, so that two
// threads comparing the forms never race on library state
extern "C" int snerf_grid_encode_fwd_ref(const float* inputs, const void* embeddings, const int* offsets, void* outputs, int B, int D, int C,
                                         int L, float S, int H, void* dy_dx, int gridtype, int align_corners, int interp, int dtype,
                                         long out_stride_l, long out_stride_b, void* stream) {
  return grid_fwd_entry(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners, interp, dtype, out_stride_l, out_stride_b, stream, false);
}

extern "C" int snerf_grid_encode_bwd(const void* grad, const float* inputs, const void* embeddings, const int* offsets, void* grad_embeddings,
                                     int B, int D, int C, int L, float S, int H, const void* dy_dx, void* grad_inputs, int gridtype,
                                     int align_corners, int interp, int dtype, long grad_stride_l, long grad_stride_b, void* stream) {
  (void)embeddings;
  if (B <= 0) return SNERF_OK;
  if (L <= 0 || grad == nullptr || inputs == nullptr || offsets == nullptr || grad_embeddings == nullptr) return SNERF_ERR_ARG;
  GridBwdArgs a{grad, inputs, offsets, grad_embeddings, grad_stride_l, grad_stride_b, B, L, S, H, gridtype, align_corners, interp};
  GRID_DISPATCH(launch_bwd, (a, (hipStream_t)stream))
  if (dy_dx != nullptr && grad_inputs != nullptr) {
    const long n = (long)B * D;
    if (dtype == SNERF_DT_F32)
      hipLaunchKernelGGL(grid_input_bwd_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)grad,
                         grad_stride_l, grad_stride_b, (const float*)dy_dx, (float*)grad_inputs, B, D, C, L);
    else if (dtype == SNERF_DT_F64)
      hipLaunchKernelGGL(grid_input_bwd_kernel<double>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const double*)grad,
                         grad_stride_l, grad_stride_b, (const double*)dy_dx, (double*)grad_inputs, B, D, C, L);
    else
      hipLaunchKernelGGL(grid_input_bwd_kernel<__half>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const __half*)grad,
                         grad_stride_l, grad_stride_b, (const __half*)dy_dx, (__half*)grad_inputs, B, D, C, L);
  }
  return snerf_check_launch();
}

extern "C" int snerf_grid_tv_grad(const float* inputs, const void* embeddings, void* grad, const int* offsets, float weight, int B, int D,
                                  int C, int L, float S, int H, int gridtype, int align_corners, int dtype, void* stream) {
  if (B <= 0) return SNERF_OK;
  if (L <= 0 || inputs == nullptr || embeddings == nullptr || grad == nullptr || offsets == nullptr) return SNERF_ERR_ARG;
  GRID_DISPATCH(launch_tv, (inputs, embeddings, grad, offsets, weight, B, L, S, H, gridtype, align_corners, (hipStream_t)stream))
  return snerf_check_launch();
}
