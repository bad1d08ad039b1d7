// Shared device/host helpers for libsnerf_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SNERF_OK 0
#define SNERF_ERR_ARG 1
#define SNERF_ERR_LAUNCH 2

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 h16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define SNERF_DT_F32 0
#define SNERF_DT_BF16 1
#define SNERF_DT_F16 2
#define SNERF_DT_F64 3   // hash-grid tables of the stand-alone GridEncoder operator only
#define SNERF_DT_BF16X3 4  // GEMM entries only: split-bf16 operands (hi = bf16(x), lo = bf16(x - hi); three MFMA passes), gemm.hip
#define SNERF_DT_F16F8 5   // snerf_linear_fwd only: fp16 + fp8 split operands (fp16 tiles + e4m3 correction tiles on the block-scaled MFMA: two pass-equivalents), gemm.hip

extern int g_snerf_last_hip_error;   // (elementwise.hip) the hipError_t behind the most recent SNERF_ERR_LAUNCH: snerf_last_hip_error()
static inline int snerf_check_launch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) g_snerf_last_hip_error = (int)e;
  return e == hipSuccess ? SNERF_OK : SNERF_ERR_LAUNCH;
}

// Bijective XCD-aware remap of a 1-D block id (MI355X: 8 XCDs, block b runs on
// XCD b % 8).  Blocks that land on one XCD get a contiguous range of logical
// ids so that neighbouring tiles share that XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// inclusive prefix sum across the 64 lanes of a wave
__device__ __forceinline__ float wave_incl_scan_add(float v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    float t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}

// inclusive suffix sum: lane i gets sum over lanes >= i
__device__ __forceinline__ float wave_incl_rscan_add(float v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    float t = __shfl_down(v, o, 64);
    if (lane + o < 64) v += t;
  }
  return v;
}

__device__ __forceinline__ float wave_incl_scan_mul(float v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    float t = __shfl_up(v, o, 64);
    if (lane >= o) v *= t;
  }
  return v;
}

// four fp32 -> four OCP e4m3 bytes of x * mul, saturating at the format's +-448 (the fp16 + fp8 split operands of gemm.hip, GemmNT::split == 2)
__device__ __forceinline__ unsigned snerf_e4m3x4(float x0, float x1, float x2, float x3, float mul) {
  x0 = __builtin_amdgcn_fmed3f(x0 * mul, -448.f, 448.f); x1 = __builtin_amdgcn_fmed3f(x1 * mul, -448.f, 448.f);
  x2 = __builtin_amdgcn_fmed3f(x2 * mul, -448.f, 448.f); x3 = __builtin_amdgcn_fmed3f(x3 * mul, -448.f, 448.f);
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(x0, x1, 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(x2, x3, w, true);
  return (unsigned)w;
}
// power-of-two scales of those bytes: activations carry e4m3((x - fp16(x)) 2^13) and e4m3(x 2^2), weights e4m3(w 2^9) and e4m3((w - fp16(w)) 2^20);
// 13 + 9 = 2 + 20 = 22 = what the two E8M0 scale bytes (116 = 2^-11 each) of the block-scaled MFMA take out again
#define SNERF_F8_ACT_RES 8192.f
#define SNERF_F8_ACT_VAL 4.f
#define SNERF_F8_W_VAL 512.f
#define SNERF_F8_W_RES 1048576.f

template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ __bf16 from_f32<__bf16>(float v) { return (__bf16)v; }
template <> __device__ __forceinline__ _Float16 from_f32<_Float16>(float v) { return (_Float16)v; }   // (the fp16 compute mode, dtype SNERF_DT_F16)
template <typename T> __device__ __forceinline__ float to_f32(T v) { return (float)v; }
