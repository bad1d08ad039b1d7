// S-NeRF++ / zipnerf background model: sampling and the fused multisample hash-grid featurisation (gfx950).
//
//   snerf_zip_resample    stepfun.max_dilate_weights (stepfun.py:75-105) + the annealed logits of Model.forward
//                         (models.py:196-203) + stepfun.sample_intervals (stepfun.py:251-294, 175-218, 154-161, 108-128;
//                         math.sorted_interp math.py:88-107) + the s -> t ray warp (coord.py:103-162), one launch per level.
//                         The reference materialises [R, 3S+2, S] and [R, S+1, n] masks; here one lane walks one ray.
//   snerf_zip_encode_fwd  render.cast_rays (render.py:129-168: 7 multisamples on a 3-turn helix) + coord.contract_mean_std
//                         (coord.py:51-63) + /2 + GridEncoder forward (gridencoder.cu:87-245) + the erf down-weighting and
//                         the mean over the multisamples of MLP.predict_density (models.py:488-497), written straight into
//                         the density MLP's operand buffer.  The [R,S,7,3] means and the [R*S*7, L*C] per-multisample
//                         features never exist in memory.
//   snerf_zip_encode_bwd  the matching scatter-add into the fp32 table gradient (gridencoder.cu:248-340 composed with the
//                         mean / erf weights).
//   snerf_zip_composite_* render.compute_alpha_weights + volumetric_rendering (render.py:170-233) fused with the density /
//                         colour activations of MLP.forward (models.py:586, 689-703).
#include "common.h"
#include <stdlib.h>
#include <hip/hip_fp16.h>

#define ZIP_LANES 64

// ------------------------------------------------------------------------------------------------------------------
// resample (lane per ray)
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float zip_pow_t(float x, float lam) {            // coord.py:103-108
  const float l1 = fabsf(lam - 1.f);
  return l1 / lam * (powf(x / l1 + 1.f, lam) - 1.f);
}
__device__ __forceinline__ float zip_inv_pow_t(float x, float lam) {        // coord.py:111-118
  const float l1 = fabsf(lam - 1.f);
  return (powf(x * lam / l1 + 1.f + 1.1920929e-07f, 1.f / lam) - 1.f) * l1;
}

struct ZipResample {
  const float* sdist; const float* weights; int S0;      // input step function: S0 intervals, S0+1 posts
  const float* u; long u_stride; int n;                  // n centres -> n+1 output posts
  const float* near; const float* far;
  long R; float dilation; int dilate; float anneal, pad, lam, dom0, dom1;
  float* sdist_out; float* tdist_out;
};

#define ZR_WAVES 4
__device__ __forceinline__ double zr_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double zr_wave_scan(double v, int lane) {     // inclusive prefix sum over the 64 lanes
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const double u = __shfl_up(v, o, 64); if (lane >= o) v += u; }
  return v;
}

// One wave per ray, four rays per workgroup: every stage of C2 + C3 is a data-parallel pass over the <= 3*S0+1 posts, staged in a
// per-wave LDS region (2.3 KB at S0 = n = 64, so the LDS no longer limits the occupancy as it did for the lane-per-ray walk):
//   merge of the three sorted post lists {t}, {t - d}, {t + d}  -> rank of every element by binary searches (ties: t - d, t, t + d,
//                                                                   the order of the sequential merge)
//   dilated pdf (max over the intervals whose dilated support covers the post) -> the covering range by two binary searches
//   normalisation / softmax / cdf                                 -> float64 wave reductions and scans, one rounding per value
//   inverse cdf at the n centres, fence posts, power warp          -> one lane per centre, coalesced stores
__global__ __launch_bounds__(64 * ZR_WAVES) void zip_resample_kernel(ZipResample a) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int S0 = a.S0, NPmax = 3 * S0 + 2;
  const int per_wave = 2 * (S0 + 1) + 2 * NPmax + a.n;
  float* t = lds + (size_t)wv * per_wave;               // [S0+1] input posts
  float* pdf = t + (S0 + 1);                            // [S0+1] input weights (-> pdf when dilating; -> the S0+1 cdf values when not)
  float* Tm = pdf + (S0 + 1);                           // [NPmax] merged posts
  float* Wm = Tm + NPmax;                               // [NPmax] dilated weights -> cdf
  float* cb = Wm + NPmax;                               // [n]    sampled centres
  long ray = (long)blockIdx.x * ZR_WAVES + wv;
  const bool live = ray < a.R;
  if (!live) ray = a.R - 1;                             // keeps the barriers uniform; nothing is stored
  const float eps = 1.1920929e-07f, d = a.dilation;
  const float* gt = a.sdist + ray * (S0 + 1);
  const float* gw = a.weights + ray * S0;
  for (int k = lane; k <= S0; k += 64) t[k] = gt[k];
  for (int k = lane; k < S0; k += 64) pdf[k] = gw[k];
  __syncthreads();
  const float* T;                                       // posts of the step function that gets sampled
  float* W;                                             // its weights, later its cdf
  int S;
  if (a.dilate) {
    const int NP = 3 * S0 + 1;
    // #{j : list_j (<= | <) x} for the three monotone lists A = t (S0+1 entries), B = t[:-1] - d, C = t[1:] + d (S0 entries each)
    auto cntA = [&](float x, bool le) __attribute__((always_inline)) { int lo = 0, hi = S0 + 1; while (lo < hi) { const int m = (lo + hi) >> 1; const float v = t[m]; if (le ? v <= x : v < x) lo = m + 1; else hi = m; } return lo; };
    auto cntB = [&](float x, bool le) __attribute__((always_inline)) { int lo = 0, hi = S0; while (lo < hi) { const int m = (lo + hi) >> 1; const float v = t[m] - d; if (le ? v <= x : v < x) lo = m + 1; else hi = m; } return lo; };
    auto cntC = [&](float x, bool le) __attribute__((always_inline)) { int lo = 0, hi = S0; while (lo < hi) { const int m = (lo + hi) >> 1; const float v = t[m + 1] + d; if (le ? v <= x : v < x) lo = m + 1; else hi = m; } return lo; };
    for (int e = lane; e < NP; e += 64) {
      float v; int rank;
      if (e <= S0) { v = t[e]; rank = e + cntB(v, true) + cntC(v, false); }
      else if (e <= 2 * S0) { const int i = e - S0 - 1; v = t[i] - d; rank = i + cntA(v, false) + cntC(v, false); }
      else { const int i = e - 2 * S0 - 1; v = t[i + 1] + d; rank = i + cntA(v, true) + cntB(v, true); }
      Tm[rank] = fminf(fmaxf(v, a.dom0), a.dom1);
    }
    __syncthreads();
    float wkeep[4];                                      // this lane's w / pdf values stay in registers across the in-place update
    {
      int c = 0;
      for (int k = lane; k < S0; k += 64) wkeep[c++ & 3] = pdf[k] / fmaxf(t[k + 1] - t[k], eps);
      __syncthreads();
      c = 0;
      for (int k = lane; k < S0; k += 64) pdf[k] = wkeep[c++ & 3];
    }
    __syncthreads();
    double part = 0.0;
    for (int k = lane; k < NP - 1; k += 64) {
      const float tk = Tm[k];
      const int lo = cntC(tk, true), hi = cntB(tk, true) - 1;
      float p = 0.f;
      for (int i = lo; i <= hi; ++i)
        if (t[i] - d <= tk && t[i + 1] + d > tk) p = fmaxf(p, pdf[i]);
      const float wd = p * (Tm[k + 1] - tk);
      Wm[k] = wd;
      part += (double)wd;
    }
    const float sum = fmaxf((float)zr_wave_sum(part), eps);
    __syncthreads();
    // caller trims [1:-1] (models.py:187-188): posts 1 .. NP-2, intervals 1 .. NP-3
    S = NP - 3;
    T = Tm + 1;
    W = Wm + 1;
    for (int k = lane; k < S; k += 64) W[k] = W[k] / sum;
    __syncthreads();
  } else {
    S = S0;
    T = t;
    W = pdf;
  }
  // annealed logits + softmax (models.py:196-203, stepfun.py:157); S <= 4 * 64
  float lg[4];
  float mx = -INFINITY;
  {
    int c = 0;
    for (int k = lane; k < S; k += 64) { const float l = T[k + 1] > T[k] ? a.anneal * logf(W[k] + a.pad) : -INFINITY; lg[c++ & 3] = l; mx = fmaxf(mx, l); }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  double part = 0.0;
  {
    int c = 0;
    for (int k = lane; k < S; k += 64) { const float e = expf(lg[c & 3] - mx); lg[c++ & 3] = e; part += (double)e; }
  }
  const float esum = (float)zr_wave_sum(part);
  __syncthreads();
  // cdf in place: C[0] = 0, C[k+1] = min(1, cumsum), C[S] = 1 (stepfun.py:108-128), stored as W[k] = C[k], k = 0 .. S
  {
    double carry = 0.0;
    int c = 0;
    for (int k0 = 0; k0 < S; k0 += 64) {
      const int k = k0 + lane;
      const double v = k < S ? (double)(lg[c++ & 3] / esum) : 0.0;
      const double inc = zr_wave_scan(v, lane) + carry;
      if (k < S) W[k + 1] = k == S - 1 ? 1.f : fminf(1.f, (float)inc);
      carry = __shfl(inc, 63, 64);
    }
    if (lane == 0) W[0] = 0.f;
  }
  __syncthreads();
  // invert the cdf at the n centres
  const float* ur = a.u + ray * a.u_stride;
  for (int j = lane; j < a.n; j += 64) {
    const float uj = ur[j];
    int lo = 0, hi = S + 1;
    while (lo < hi) { const int m = (lo + hi) >> 1; if (W[m] <= uj) lo = m + 1; else hi = m; }
    int i0 = lo - 1; i0 = i0 < 0 ? 0 : (i0 > S ? S : i0);
    const int i1 = i0 + 1 > S ? S : i0 + 1;
    float off = (uj - W[i0]) / (W[i1] - W[i0]);
    if (off != off) off = 0.f;
    off = fminf(fmaxf(off, 0.f), 1.f);
    cb[j] = T[i0] + off * (T[i1] - T[i0]);
  }
  __syncthreads();
  // fence posts = midpoints, end posts reflected and clamped (stepfun.py:281-294), warped to metric distances
  if (live) {
    const float nr = a.near[ray], fr = a.far[ray];
    const float s_near = zip_pow_t(nr * 2.f, a.lam), s_far = zip_pow_t(fr * 2.f, a.lam);
    float* so = a.sdist_out + ray * (a.n + 1);
    float* to = a.tdist_out + ray * (a.n + 1);
    for (int j = lane; j <= a.n; j += 64) {
      float s;
      if (j == 0) s = fmaxf(2.f * cb[0] - (cb[1] + cb[0]) / 2.f, a.dom0);
      else if (j == a.n) s = fminf(2.f * cb[a.n - 1] - (cb[a.n - 1] + cb[a.n - 2]) / 2.f, a.dom1);
      else s = (cb[j] + cb[j - 1]) / 2.f;
      so[j] = s;
      to[j] = zip_inv_pow_t(s * s_far + (1.f - s) * s_near, a.lam) / 2.f;
    }
  }
}

extern "C" int snerf_zip_resample(const float* sdist, const float* weights, int S0, const float* u, long u_stride, int n,
                                  const float* near, const float* far, long R, float dilation, int dilate, float anneal,
                                  float resample_padding, float lam, float dom0, float dom1, float* sdist_out, float* tdist_out,
                                  void* stream) {
  if (R <= 0) return SNERF_OK;
  if (S0 < 1 || n < 2 || (dilate && S0 < 2)) return SNERF_ERR_ARG;
  if ((dilate ? 3 * S0 - 2 : S0) > 256) return SNERF_ERR_ARG;                 // per-lane register slots: 4 x 64 intervals
  const size_t lds = (size_t)ZR_WAVES * (2 * (S0 + 1) + 2 * (3 * S0 + 2) + n) * sizeof(float);
  if (lds > 64 * 1024) return SNERF_ERR_ARG;
  ZipResample a{sdist, weights, S0, u, u_stride, n, near, far, R, dilation, dilate, anneal, resample_padding, lam, dom0, dom1, sdist_out, tdist_out};
  hipLaunchKernelGGL(zip_resample_kernel, dim3((unsigned)((R + ZR_WAVES - 1) / ZR_WAVES)), dim3(64 * ZR_WAVES), lds, (hipStream_t)stream, a);
  return snerf_check_launch();
}

// ------------------------------------------------------------------------------------------------------------------
// fused multisample featurisation
// ------------------------------------------------------------------------------------------------------------------
struct ZipEnc {
  const float* tdist; const float* origins; const float* directions; const float* radii; const float* base_x; const float* base_y;
  const float* deg_jitter;            // [R,S,n] or null
  const void* table; const int* offsets; const int* grid_sizes;   // grid_sizes[l] = GridEncoder.grid_sizes (resolution + 1)
  void* feat; long ld;                // forward: output [P, ld]; backward: gradient input
  float* grad_table;                  // backward only (fp32)
  void* grad_table16;                 // backward, optional: bf16 [entries, C] for the levels that scatter with global atomics (C even)
  long R; int S, L, n, m; float Sl; int H; float std_scale;
  int level_begin;                    // first level handled by the generic kernel
  long slab_row0, slab_rows;          // LDS-privatised backward: the row range this workgroup accumulates
  const int* lds_slab_level; int lds_nslab;   // slab s covers level lds_slab_level[2s], rows from lds_slab_level[2s+1]
  int lf;                             // (workgroup, level) grid launched 1-D with the level fastest: the number of level blocks; 0 = 2-D grid (x = workgroup, y = level)
};
// logical block coordinates of the (workgroup, level) grids under either launch order
#define ZIP_BX(a) ((a).lf ? blockIdx.x / (unsigned)(a).lf : blockIdx.x)
#define ZIP_BY(a) ((a).lf ? blockIdx.x % (unsigned)(a).lf : blockIdx.y)
#define ZIP_GX(a) ((a).lf ? gridDim.x / (unsigned)(a).lf : gridDim.x)

__device__ __forceinline__ uint32_t zip_hash3(const uint32_t* p) { return p[0] ^ (p[1] * 2654435761u) ^ (p[2] * 805459861u); }

__device__ __forceinline__ uint32_t zip_grid_index(uint32_t hs, uint32_t res, const uint32_t* pg) {   // gridencoder.cu:66-84, D = 3, hash type
  // The reference walks `if (stride <= hs) { index += pg[d] * stride; stride *= res + 1; }` over d and hashes when the final stride exceeds
  // hs.  Which of the two a level is depends on (hs, res) only -- the same for every lane wherever a wave works on one level -- so the
  // decision is taken on those (scalar) values first and each lane evaluates ONE form: the featurisation and record kernels are VALU-issue
  // bound (profiles/r6_zz_pathC_gather_bound_pmc.txt) and evaluating both forms plus a runtime modulo per corner was a third of their
  // instructions (u32 multiplies are quarter rate).  Same uint32 arithmetic, same wrap-around, same result for every input.
  const uint32_t s1 = res + 1u, s2 = s1 * s1;
  const bool dense = s1 <= hs && s2 <= hs && s2 * s1 <= hs;        // every guard of the walk passes and the final stride does not exceed hs
  uint32_t index;
  if (dense) {
    index = pg[0] + pg[1] * s1 + pg[2] * s2;
    while (index >= hs) index -= hs;                               // (= index % hs; coordinates beyond res only: a three-instruction loop instead of a modulo expansion per corner -- these kernels are as long as the instruction cache)
  } else {
    index = zip_hash3(pg);
    index = (hs & (hs - 1u)) == 0u ? (index & (hs - 1u)) : index % hs;
  }
  return index;
}

template <typename TT, int C> struct alignas(sizeof(TT) * C) ZVec { TT v[C]; };

// Grid position of one coordinate on a level: ps = x * scale + 0.5 (gridencoder.cu:148), its cell and fraction.  The forward's count
// pass and the backward's record writers evaluate this in DIFFERENT kernels and must land in the same cell (a record more or less than
// counted would overrun a reserved range), so the product and the sum are rounded separately whatever the surrounding code is
// contracted into -- as the CPU oracle's separate fp32 ops do.
__device__ __forceinline__ void zip_cell(float x01, float scale, uint32_t* pg, float* fr) {
#pragma clang fp contract(off)
  const float ps = x01 * scale + 0.5f;
  const float fl = floorf(ps);
  *pg = (uint32_t)fl;
  *fr = ps - fl;
}

// position, contracted + halved, and its erf weight for one multisample
__device__ __forceinline__ void zip_sample_point(const ZipEnc& a, long ray, int i, int j, float t0, float t1, const float* o, const float* d,
                                                 const float* bx, const float* by, float rad, float* x01, float* sd) {
#pragma clang fp contract(off)        // (same positions in every kernel that calls this: see zip_cell)
  const float t = t0 + (t1 - t0) * ((float)j + 0.5f) / (float)a.n;
  float deg = 2.f * 3.14159265358979f * (float)a.m * (float)j / (float)a.n;
  if (a.deg_jitter != nullptr) deg += a.deg_jitter[(ray * a.S + i) * a.n + j] * 3.14159265358979f * 2.f;
  float sn, cs;
  sincosf(deg, &sn, &cs);
  const float lx = rad * t * cs / 2.f, ly = rad * t * sn / 2.f;
  float x[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) x[k] = lx * bx[k] + ly * by[k] + t * d[k] + o[k];
  float std = a.std_scale * rad * t;
  const float msq = fmaxf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2], 1.1920929e-07f);   // coord.py:51-63
  if (!(msq <= 1.f)) {
    const float mag = sqrtf(msq);
    const float sc = (2.f * mag - 1.f) / msq;
    const float q = 2.f / mag - 1.f / msq;
    const float det = (1.f / msq) * (q * q);
    std = cbrtf(det) * std;
#pragma unroll
    for (int k = 0; k < 3; ++k) x[k] = sc * x[k];
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) x01[k] = (x[k] / 2.f + 1.f) / 2.f;   // /2 (models.py:488-490), then (x + bound) / (2 bound) (grid.py:162)
  *sd = std / 2.f;
}

// MODE 0: forward gather.  MODE 1: backward, fp32 atomics straight into the table gradient.  MODE 2: backward for SMALL DENSE
// levels: the whole level's gradient table is privatised in LDS (ds_add_f32), a few hundred persistent workgroups stride over
// the points and flush once -- at resolution 17..33 every cell receives thousands of contributions per step, and same-address
// global atomics serialise (measured: level 0 alone cost 63 ms of a 107 ms backward before this path).
template <typename TT, typename OT, int C, int MODE>
__device__ __forceinline__ void zip_point_level(const ZipEnc& a, long p, int level, float* lds_tab) {
  const long ray = p / a.S;
  const int i = (int)(p - ray * a.S);
  const float t0 = a.tdist[ray * (a.S + 1) + i], t1 = a.tdist[ray * (a.S + 1) + i + 1];
  float o[3], d[3], bx[3], by[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { o[k] = a.origins[ray * 3 + k]; d[k] = a.directions[ray * 3 + k]; bx[k] = a.base_x[ray * 3 + k]; by[k] = a.base_y[ray * 3 + k]; }
  const float rad = a.radii[ray];
  const uint32_t hs = a.offsets[level + 1] - a.offsets[level];
  const float scale = exp2f(level * a.Sl) * a.H - 1.0f;
  const uint32_t res = (uint32_t)ceilf(scale) + 1;
  const TT* tab = (const TT*)a.table + (long)a.offsets[level] * C;
  float acc[C], g[C];
#pragma unroll
  for (int c = 0; c < C; ++c) { acc[c] = 0.f; g[c] = 0.f; }
  if (MODE != 0) {
    const OT* gi = (const OT*)a.feat + p * a.ld + level * C;
#pragma unroll
    for (int c = 0; c < C; ++c) g[c] = (float)gi[c] / (float)a.n;
  }
  // backward: contributions of consecutive multisamples that fall into the same cell are merged in registers first
  uint32_t cur[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu};
  float wsum[8];
#pragma unroll
  for (int idx = 0; idx < 8; ++idx) wsum[idx] = 0.f;
  auto flush = [&]() {
    if (cur[0] == 0xffffffffu) return;
    if constexpr (C == 1 && MODE == 1) {
      // single-channel grids with the bf16 gradient table: on the hashed levels the two x-neighbours of a corner pair share one
      // aligned 32-bit word whenever x is even (see the forward), so ONE packed bf16 atomic carries both; an unpaired corner still
      // costs one atomic (its value next to a zero).  Dense levels stay in fp32: their cells collect hundreds of addends each.
      if (a.grad_table16 != nullptr && (uint64_t)(res + 1) * (res + 1) * (res + 1) > (uint64_t)hs) {
        typedef short s16x2 __attribute__((ext_vector_type(2)));
        typedef __bf16 b16x2 __attribute__((ext_vector_type(2)));
        short* dst = (short*)a.grad_table16 + (long)a.offsets[level];
        auto add2 = [&](long base, float lo, float hi) __attribute__((always_inline)) {
          const b16x2 v = {(__bf16)lo, (__bf16)hi};
          __builtin_amdgcn_global_atomic_fadd_v2bf16((__attribute__((address_space(1))) s16x2*)(dst + base), __builtin_bit_cast(s16x2, v));
        };
#pragma unroll
        for (int yz = 0; yz < 4; ++yz) {
          uint32_t pl[3] = {cur[0], cur[1] + (yz & 1), cur[2] + (yz >> 1)};
          const long r0 = zip_grid_index(hs, res, pl);
          pl[0] = cur[0] + 1;
          const long r1 = zip_grid_index(hs, res, pl);
          const float v0 = wsum[2 * yz] * g[0], v1 = wsum[2 * yz + 1] * g[0];
          if ((r0 ^ r1) == 1) add2(r0 & ~1L, (r0 & 1) ? v1 : v0, (r0 & 1) ? v0 : v1);
          else {
            add2(r0 & ~1L, (r0 & 1) ? 0.f : v0, (r0 & 1) ? v0 : 0.f);
            add2(r1 & ~1L, (r1 & 1) ? 0.f : v1, (r1 & 1) ? v1 : 0.f);
          }
          wsum[2 * yz] = 0.f; wsum[2 * yz + 1] = 0.f;
        }
        return;
      }
    }
#pragma unroll
    for (int idx = 0; idx < 8; ++idx) {
      uint32_t pl[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) pl[k] = cur[k] + ((idx >> k) & 1);
      const long row = zip_grid_index(hs, res, pl);
      if (MODE == 2) {
        const long lr = row - a.slab_row0;                 // this workgroup owns rows [slab_row0, slab_row0 + slab_rows)
        if (lr >= 0 && lr < a.slab_rows) {
#pragma unroll
          for (int c = 0; c < C; ++c) atomicAdd(lds_tab + lr * C + c, wsum[idx] * g[c]);
        }
      } else if (C % 2 == 0 && a.grad_table16 != nullptr) {
        // the fp32 atomic rate is a hardware constant (profiles/r1_q): a packed bf16 pair halves the number of atomics
        typedef short s16x2 __attribute__((ext_vector_type(2)));
        typedef __bf16 b16x2 __attribute__((ext_vector_type(2)));
        short* dst = (short*)a.grad_table16 + ((long)a.offsets[level] + row) * C;
#pragma unroll
        for (int c = 0; c < C; c += 2) {
          const b16x2 v = {(__bf16)(wsum[idx] * g[c]), (__bf16)(wsum[idx] * g[c + 1])};
          __builtin_amdgcn_global_atomic_fadd_v2bf16((__attribute__((address_space(1))) s16x2*)(dst + c), __builtin_bit_cast(s16x2, v));
        }
      } else {
        float* dst = a.grad_table + ((long)a.offsets[level] + row) * C;
#pragma unroll
        for (int c = 0; c < C; ++c) atomicAdd(dst + c, wsum[idx] * g[c]);
      }
      wsum[idx] = 0.f;
    }
  };
  float pair_acc[8];
  for (int j = 0; j < a.n; ++j) {
    float x01[3], sd;
    zip_sample_point(a, ray, i, j, t0, t1, o, d, bx, by, rad, x01, &sd);
    if (x01[0] < 0.f || x01[0] > 1.f || x01[1] < 0.f || x01[1] > 1.f || x01[2] < 0.f || x01[2] > 1.f) continue;   // encoder returns zeros
    // erf(1 / sqrt(8 std^2 gs^2)) with gs = GridEncoder.grid_sizes[level] (models.py:494)
    const float gs = (float)a.grid_sizes[level];
    const float we = erff(1.f / sqrtf(8.f * sd * sd * gs * gs));
    float fr[3];
    uint32_t pg[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      zip_cell(x01[k], scale, &pg[k], &fr[k]);
    }
    if (MODE != 0 && (pg[0] != cur[0] || pg[1] != cur[1] || pg[2] != cur[2])) {
      flush();
      cur[0] = pg[0]; cur[1] = pg[1]; cur[2] = pg[2];
    }
    if constexpr (MODE == 0 && C == 1 && sizeof(TT) == 2) {
      // single-channel 2-byte tables (the proposal grids): the two x-neighbours of a corner pair are adjacent entries whenever their
      // rows differ only in bit 0 -- always for an even row of a dense level, and for every even x of a hashed level (the hash
      // multiplies x by 1, so x ^ 1 flips bit 0 of the row and nothing else) -- and then ONE aligned 32-bit load serves both.
      // The gathers are request-bound, not byte-bound: a quarter fewer requests.
#pragma unroll
      for (int yz = 0; yz < 4; ++yz) {
        uint32_t pl[3] = {pg[0], pg[1] + (yz & 1), pg[2] + (yz >> 1)};
        const float wyz = ((yz & 1) ? fr[1] : 1.f - fr[1]) * ((yz >> 1) ? fr[2] : 1.f - fr[2]);
        const long r0 = zip_grid_index(hs, res, pl);
        pl[0] = pg[0] + 1;
        const long r1 = zip_grid_index(hs, res, pl);
        float v0, v1;
        if ((r0 ^ r1) == 1) {
          const uint32_t word = *reinterpret_cast<const uint32_t*>(tab + (r0 & ~1L));
          const uint16_t lo16 = (uint16_t)(word & 0xffffu), hi16 = (uint16_t)(word >> 16);
          const uint16_t b0 = (r0 & 1) ? hi16 : lo16, b1 = (r0 & 1) ? lo16 : hi16;
          v0 = (float)__builtin_bit_cast(TT, b0);
          v1 = (float)__builtin_bit_cast(TT, b1);
        } else {
          v0 = (float)tab[r0];
          v1 = (float)tab[r1];
        }
        // same products, same order as the generic loop below (corner index = x + 2 y + 4 z)
        float wa = 1.f - fr[0], wb = fr[0];
        wa *= (yz & 1) ? fr[1] : 1.f - fr[1]; wb *= (yz & 1) ? fr[1] : 1.f - fr[1];
        wa *= (yz >> 1) ? fr[2] : 1.f - fr[2]; wb *= (yz >> 1) ? fr[2] : 1.f - fr[2];
        (void)wyz;
        pair_acc[2 * yz] = (wa * we) * v0;
        pair_acc[2 * yz + 1] = (wb * we) * v1;
      }
#pragma unroll
      for (int idx = 0; idx < 8; ++idx) acc[0] += pair_acc[idx];
    } else {
#pragma unroll
    for (int idx = 0; idx < 8; ++idx) {
      float w = 1.f;
      uint32_t pl[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        if (idx & (1 << k)) { w *= fr[k]; pl[k] = pg[k] + 1; } else { w *= 1.f - fr[k]; pl[k] = pg[k]; }
      }
      if (MODE == 0) {
        const long row = zip_grid_index(hs, res, pl);
        const ZVec<TT, C> r = *reinterpret_cast<const ZVec<TT, C>*>(tab + row * C);
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] += (w * we) * (float)r.v[c];
      } else {
        wsum[idx] += w * we;
      }
    }
    }
  }
  if (MODE == 0) {
    OT* out = (OT*)a.feat + p * a.ld + level * C;
#pragma unroll
    for (int c = 0; c < C; ++c) out[c] = (OT)(acc[c] / (float)a.n);
  } else {
    flush();
  }
}

template <typename TT, typename OT, int C, bool BWD>
__global__ __launch_bounds__(256) void zip_encode_kernel(ZipEnc a) {
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p >= a.R * a.S) return;
  zip_point_level<TT, OT, C, BWD ? 1 : 0>(a, p, blockIdx.y + a.level_begin, nullptr);
}

// (bins of the binned table gradient, described further down: the training forward below can already count the records per bin)
#define ZB_NBMAX 1024                      // bins per level (row ranges x replicas)
// bin of a table row: (row >> bshift) * K + rep.  Row ranges and replicas are both at most ZB_NBMAX, so the 24-bit multiply is exact -- and
// issues at the full VALU rate where v_mul_lo_u32 takes four slots (these kernels are VALU-issue bound and form 56 bins per thread)
__device__ __forceinline__ int zb_bin(uint32_t row, int bshift, int K, int rep) { return (int)__umul24(row >> bshift, (unsigned)K) + rep; }
#define ZB_HEAD 34                         // the largest |grad_feat| entry maps below 2^ZB_HEAD: 2^27 records cannot overflow 63 bits

struct ZipBin {
  int bshift;                              // log2(rows per bin)
  int* counts;                             // [L, ZB_NBMAX] records per bin
  unsigned* wg_offsets;                    // [L, workgroups, ZB_NBMAX] offset of a workgroup's record range inside a bin (pass 0 -> pass 1)
  const long* starts;                      // [L, ZB_NBMAX] bin offsets (accumulate pass)
  int ksplit[16];                          // replicas per row range, per level
  unsigned short* rec_row; float* rec_val; long capacity;   // records: C = 1: rec_val holds {row, value} pairs (8 B); else row + C floats
  long long* g64; long g64_rows;           // int64 image of table rows [0, g64_rows) for the replicated levels
  const int* scale_exp;                    // device: the launch's fixed-point scale is 2^scale_exp[0] (snerf_zip_bin_scale)
  // count-forward of the single-channel grids: blockIdx.y enumerates GROUPS of up to two levels evaluated by one thread on one
  // evaluation of the interval's multisamples (ngrp = 0: one level per thread, blockIdx.y = level); see snerf_zip_encode_fwd_count
  int ngrp; signed char grp[16][2];
};
// HALF RECORDS (HREC; passes 5 / 6 / 7): the record's values travel as fp16 of value * 2^(scale_exp - ZB_HALF_SHIFT) -- the largest
// |grad_feat| entry lands in [2^14, 2^15), fp16's subnormals reach 2^-39 of it, below the 2^-34 fixed-point grid -- 10 instead of 18
// bytes per record of a four-channel grid ({row u16} + {4 x fp16}), 4 instead of 8 for a single-channel one ({row u16 | fp16} in one
// word).  Every contribution is rounded to 11 significant bits once (round to nearest: unbiased), the sums stay exact and
// order-independent.  The reference scatters __half2 atomics into an fp16 table's gradient (gridencoder.cu:302-315: every ADDITION rounds
// to 11 bits), so for the tables it halves this is tighter than the reference's own gradient; for fp32 tables it is an option.
#define ZB_HALF_SHIFT (ZB_HEAD - 15)
typedef _Float16 zb_h4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned zb_half_bits(float v) { return (unsigned)__builtin_bit_cast(unsigned short, (_Float16)v); }

// Forward featurisation, one thread per interval for ALL levels: the n <= 8 helix multisamples (sincos, contraction, cbrt) are
// evaluated once and kept in registers instead of once per (interval, level) as in the per-level grid above -- for the
// single-channel proposal grids that arithmetic, not the gathers, was the larger half of the kernel.
// COUNT (training with the binned table gradient, one level per thread): the kernel is also PASS 0 of that gradient -- it has the 8 rows
// of every cell in hand for its gathers, so it counts the records the backward will emit (8 per run of consecutive in-bounds
// multisamples in one cell, exactly zip_emit_level's merging) in the workgroup's LDS histogram and reserves the workgroup's ranges like
// zip_bin_emit_kernel<.., 0> does; its grid (intervals / 256, levels) is the grid of the backward's record pass.  Saves the separate
// count sweep (the multisamples' sincos / contraction / cbrt once more per level).
#ifndef ZIP_PAIR_F32
#define ZIP_PAIR_F32 1                                   // fp32 single-channel tables: aligned x-neighbour pairs as one 8-byte load
#endif
template <typename TT, typename OT, int C, bool COUNT>
__device__ __forceinline__ void zip_fwd_all_body(const ZipEnc& a, const ZipBin& b, const long p, int* cnt) {
  const long ray = p / a.S;
  const int i = (int)(p - ray * a.S);
  const float t0 = a.tdist[ray * (a.S + 1) + i], t1 = a.tdist[ray * (a.S + 1) + i + 1];
  float o[3], d[3], bx[3], by[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { o[k] = a.origins[ray * 3 + k]; d[k] = a.directions[ray * 3 + k]; bx[k] = a.base_x[ray * 3 + k]; by[k] = a.base_y[ray * 3 + k]; }
  const float rad = a.radii[ray];
  float X[8][3], SDI[8];                                 // positions in [0,1]^3 and 1 / (sqrt(8) std) per multisample
  unsigned inb = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (j < a.n) {
      float sd;
      zip_sample_point(a, ray, i, j, t0, t1, o, d, bx, by, rad, X[j], &sd);
      SDI[j] = sd;
      if (!(X[j][0] < 0.f || X[j][0] > 1.f || X[j][1] < 0.f || X[j][1] > 1.f || X[j][2] < 0.f || X[j][2] > 1.f)) inb |= 1u << j;
    }
  }
  OT* out = (OT*)a.feat + p * a.ld;
  const bool grouped = COUNT && C == 1 && b.ngrp > 0;    // (groups exist for the single-channel grids only: compile-time false elsewhere)
  const int lbeg = grouped ? 0 : (int)ZIP_BY(a) * a.level_begin;
  const int lend = grouped ? 2 : min(a.L, (int)(ZIP_BY(a) + 1) * a.level_begin);       // level_begin = levels per thread in this kernel
  for (int it = lbeg; it < lend; ++it) {
    const int level = grouped ? (int)b.grp[ZIP_BY(a)][it] : it;
    if (level < 0) break;
    int* cntl = cnt + (grouped ? it * ZB_NBMAX : 0);     // (a group's levels count into their own bins)
    const uint32_t hs = a.offsets[level + 1] - a.offsets[level];
    const float scale = exp2f(level * a.Sl) * a.H - 1.0f;
    const uint32_t res = (uint32_t)ceilf(scale) + 1;
    const TT* tab = (const TT*)a.table + (long)a.offsets[level] * C;
    const float gs = (float)a.grid_sizes[level];
    float acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 0.f;
    // cell of the current run of multisamples: consecutive multisamples of an interval often sit in ONE cell (always on the coarse
    // levels), whose 8 corner entries are then kept in registers instead of being gathered again (same rows, same values: identical
    // features); COUNT tallies a record per corner of every run, exactly zip_emit_level's merging
    uint32_t cur[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu};
    const int K = COUNT ? b.ksplit[level] : 1, rep = COUNT ? (int)(ZIP_BX(a) % (unsigned)K) : 0;
    float cv[(C == 1) ? 8 : 1];                          // C = 1: the cell's corner values (corner = x + 2 y + 4 z)
    ZVec<TT, C> ce[(C == 1) ? 1 : 8];                    // C > 1: the cell's corner entries
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j >= a.n || !((inb >> j) & 1u)) continue;     // outside [0,1]^3 the encoder returns zeros
      const float sd = SDI[j];
      const float we = erff(1.f / sqrtf(8.f * sd * sd * gs * gs));
      float fr[3];
      uint32_t pg[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        zip_cell(X[j][k], scale, &pg[k], &fr[k]);
      }
      const bool newcell = pg[0] != cur[0] || pg[1] != cur[1] || pg[2] != cur[2];
      cur[0] = pg[0]; cur[1] = pg[1]; cur[2] = pg[2];
      auto tally = [&](long row) __attribute__((always_inline)) {
        if constexpr (COUNT) { if (newcell) atomicAdd(cntl + zb_bin((uint32_t)row, b.bshift, K, rep), 1); }
      };
      if constexpr (C == 1 && sizeof(TT) == 4 && ZIP_PAIR_F32) {
        // fp32 single-channel table (the proposal grids under the reference's table policy): the two x-neighbours of a corner pair are
        // adjacent entries whenever their rows differ only in bit 0 -- one aligned 8-byte load instead of two 4-byte ones (same
        // products, same order as the generic loop: corner index = x + 2 y + 4 z)
        float pa[8];
#pragma unroll
        for (int yz = 0; yz < 4; ++yz) {
          if (newcell) {
            uint32_t pl[3] = {pg[0], pg[1] + (yz & 1), pg[2] + (yz >> 1)};
            const long r0 = zip_grid_index(hs, res, pl);
            pl[0] = pg[0] + 1;
            const long r1 = zip_grid_index(hs, res, pl);
            tally(r0); tally(r1);
            if ((r0 ^ r1) == 1) {
              const float2 both = *reinterpret_cast<const float2*>(tab + (r0 & ~1L));
              cv[2 * yz] = (r0 & 1) ? both.y : both.x;
              cv[2 * yz + 1] = (r0 & 1) ? both.x : both.y;
            } else {
              cv[2 * yz] = (float)tab[r0];
              cv[2 * yz + 1] = (float)tab[r1];
            }
          }
          const float v0 = cv[2 * yz], v1 = cv[2 * yz + 1];
          float wa = 1.f - fr[0], wb = fr[0];
          wa *= (yz & 1) ? fr[1] : 1.f - fr[1]; wb *= (yz & 1) ? fr[1] : 1.f - fr[1];
          wa *= (yz >> 1) ? fr[2] : 1.f - fr[2]; wb *= (yz >> 1) ? fr[2] : 1.f - fr[2];
          pa[2 * yz] = (wa * we) * v0;
          pa[2 * yz + 1] = (wb * we) * v1;
        }
#pragma unroll
        for (int idx = 0; idx < 8; ++idx) acc[0] += pa[idx];
      } else if constexpr (C == 1 && sizeof(TT) == 2) {
        float pa[8];
#pragma unroll
        for (int yz = 0; yz < 4; ++yz) {
          if (newcell) {
            uint32_t pl[3] = {pg[0], pg[1] + (yz & 1), pg[2] + (yz >> 1)};
            const long r0 = zip_grid_index(hs, res, pl);
            pl[0] = pg[0] + 1;
            const long r1 = zip_grid_index(hs, res, pl);
            tally(r0); tally(r1);
            if ((r0 ^ r1) == 1) {                           // adjacent entries of one aligned 32-bit word (see zip_point_level)
              const uint32_t word = *reinterpret_cast<const uint32_t*>(tab + (r0 & ~1L));
              const uint16_t lo16 = (uint16_t)(word & 0xffffu), hi16 = (uint16_t)(word >> 16);
              const uint16_t b0 = (r0 & 1) ? hi16 : lo16, b1 = (r0 & 1) ? lo16 : hi16;
              cv[2 * yz] = (float)__builtin_bit_cast(TT, b0);
              cv[2 * yz + 1] = (float)__builtin_bit_cast(TT, b1);
            } else {
              cv[2 * yz] = (float)tab[r0];
              cv[2 * yz + 1] = (float)tab[r1];
            }
          }
          const float v0 = cv[2 * yz], v1 = cv[2 * yz + 1];
          float wa = 1.f - fr[0], wb = fr[0];
          wa *= (yz & 1) ? fr[1] : 1.f - fr[1]; wb *= (yz & 1) ? fr[1] : 1.f - fr[1];
          wa *= (yz >> 1) ? fr[2] : 1.f - fr[2]; wb *= (yz >> 1) ? fr[2] : 1.f - fr[2];
          pa[2 * yz] = (wa * we) * v0;
          pa[2 * yz + 1] = (wb * we) * v1;
        }
#pragma unroll
        for (int idx = 0; idx < 8; ++idx) acc[0] += pa[idx];
      } else if constexpr (C == 4 && sizeof(TT) == 2) {
        // 8-byte entries (the NeRF grid's fp16 table): the two x-neighbours of a corner pair are adjacent entries whenever their rows
        // differ only in bit 0 (every even x: the hash multiplies x by 1) -- ONE aligned 16-byte load then serves both, half of a 32-byte
        // sector instead of a quarter.  Same products in the same order as the generic loop (corner index = x + 2 y + 4 z).
#pragma unroll
        for (int yz = 0; yz < 4; ++yz) {
          if (newcell) {
            uint32_t pl[3] = {pg[0], pg[1] + (yz & 1), pg[2] + (yz >> 1)};
            const long r0 = zip_grid_index(hs, res, pl);
            pl[0] = pg[0] + 1;
            const long r1 = zip_grid_index(hs, res, pl);
            tally(r0); tally(r1);
            if ((r0 ^ r1) == 1) {
              const ZVec<TT, 8> both = *reinterpret_cast<const ZVec<TT, 8>*>(tab + (r0 & ~1L) * 4);
#pragma unroll
              for (int c = 0; c < 4; ++c) { ce[2 * yz].v[c] = both.v[(r0 & 1) * 4 + c]; ce[2 * yz + 1].v[c] = both.v[(r1 & 1) * 4 + c]; }
            } else {
              ce[2 * yz] = *reinterpret_cast<const ZVec<TT, 4>*>(tab + r0 * 4);
              ce[2 * yz + 1] = *reinterpret_cast<const ZVec<TT, 4>*>(tab + r1 * 4);
            }
          }
          const ZVec<TT, 4> e0 = ce[2 * yz], e1 = ce[2 * yz + 1];
          float wa = 1.f - fr[0], wb = fr[0];
          wa *= (yz & 1) ? fr[1] : 1.f - fr[1]; wb *= (yz & 1) ? fr[1] : 1.f - fr[1];
          wa *= (yz >> 1) ? fr[2] : 1.f - fr[2]; wb *= (yz >> 1) ? fr[2] : 1.f - fr[2];
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[c] += (wa * we) * (float)e0.v[c];
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[c] += (wb * we) * (float)e1.v[c];
        }
      } else {
#pragma unroll
        for (int idx = 0; idx < 8; ++idx) {
          float w = 1.f;
          uint32_t pl[3];
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            if (idx & (1 << k)) { w *= fr[k]; pl[k] = pg[k] + 1; } else { w *= 1.f - fr[k]; pl[k] = pg[k]; }
          }
          if (newcell) {
            const long row = zip_grid_index(hs, res, pl);
            tally(row);
            if constexpr (C == 1) cv[idx] = (float)tab[row];
            else ce[idx] = *reinterpret_cast<const ZVec<TT, C>*>(tab + row * C);
          }
#pragma unroll
          for (int c = 0; c < C; ++c) acc[c] += (w * we) * (C == 1 ? cv[idx] : (float)ce[C == 1 ? 0 : idx].v[c]);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) out[level * C + c] = (OT)(acc[c] / (float)a.n);
  }
}

template <typename TT, typename OT, int C, bool COUNT = false>
__global__ __launch_bounds__(256) void zip_encode_fwd_all_kernel(ZipEnc a, ZipBin b) {
  constexpr int NQ = COUNT && C == 1 ? 2 : 1;            // levels a thread may take (see ZipBin::grp)
  __shared__ int cnt[COUNT ? NQ * ZB_NBMAX : 1];
  const long p = (long)ZIP_BX(a) * 256 + threadIdx.x;
  const bool live = p < a.R * a.S;
  if constexpr (COUNT) {
    for (int k = threadIdx.x; k < NQ * ZB_NBMAX; k += 256) cnt[k] = 0;
    __syncthreads();
  }
  if (live) zip_fwd_all_body<TT, OT, C, COUNT>(a, b, p, cnt);
  if constexpr (COUNT) {
    __syncthreads();
    const bool grouped = NQ == 2 && b.ngrp > 0;
    for (int q = 0; q < (grouped ? 2 : 1); ++q) {       // (one level per thread, or a group of up to two)
      const int level = grouped ? (int)b.grp[ZIP_BY(a)][q] : (int)ZIP_BY(a);
      if (level < 0) break;
      unsigned* wgo = b.wg_offsets + ((long)level * ZIP_GX(a) + ZIP_BX(a)) * ZB_NBMAX;
      const int* cq = cnt + q * ZB_NBMAX;
      for (int k = threadIdx.x; k < ZB_NBMAX; k += 256)
        if (cq[k] != 0) wgo[k] = (unsigned)atomicAdd(b.counts + level * ZB_NBMAX + k, cq[k]);
    }
  }
}

// Inference of a proposal level in ONE kernel: featurisation of the single-channel grid (all levels per thread, as above) followed by
// the proposal MLP itself -- Linear(L -> hidden) + ReLU + Linear(hidden -> 1) = 448 MACs per interval at L = 6 (internal/models.py:425-427,
// 481-519 with disable_rgb) -- on the features still in registers.  Replaces feature store + two HBM-bound GEMM launches (M = 4.2 M
// rows, N = 64 / 1) per level and chunk.  `rnd` reproduces the bf16 GEMM path's roundings (features, weights and the hidden layer
// in bf16, fp32 accumulation); without it everything is fp32 (parity mode).  Weights live in LDS and are read as broadcasts.
struct ZipPropMlp { const float *w1, *b1, *w2, *b2; int hidden, rnd; float* raw_density; };

__device__ __forceinline__ float zip_rbf(float v, int rnd) { return rnd == 1 ? (float)(__bf16)v : (rnd == 2 ? (float)(_Float16)v : v); }   // rounding mode: 0 none, 1 bf16, 2 fp16

template <typename TT>
__global__ __launch_bounds__(256) void zip_encode_prop_kernel(ZipEnc a, ZipPropMlp w) {
  extern __shared__ float lw[];                          // [hidden][L] | b1[hidden] | w2[hidden] | b2
  const int nw1 = w.hidden * a.L;
  for (int k = threadIdx.x; k < nw1; k += 256) lw[k] = zip_rbf(w.w1[k], w.rnd);
  for (int k = threadIdx.x; k < w.hidden; k += 256) { lw[nw1 + k] = w.b1[k]; lw[nw1 + w.hidden + k] = zip_rbf(w.w2[k], w.rnd); }
  if (threadIdx.x == 0) lw[nw1 + 2 * w.hidden] = w.b2[0];
  __syncthreads();
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p >= a.R * a.S) return;
  const long ray = p / a.S;
  const int i = (int)(p - ray * a.S);
  const float t0 = a.tdist[ray * (a.S + 1) + i], t1 = a.tdist[ray * (a.S + 1) + i + 1];
  float o[3], d[3], bx[3], by[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { o[k] = a.origins[ray * 3 + k]; d[k] = a.directions[ray * 3 + k]; bx[k] = a.base_x[ray * 3 + k]; by[k] = a.base_y[ray * 3 + k]; }
  const float rad = a.radii[ray];
  float X[8][3], SDI[8];
  unsigned inb = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (j < a.n) {
      float sd;
      zip_sample_point(a, ray, i, j, t0, t1, o, d, bx, by, rad, X[j], &sd);
      SDI[j] = sd;
      if (!(X[j][0] < 0.f || X[j][0] > 1.f || X[j][1] < 0.f || X[j][1] > 1.f || X[j][2] < 0.f || X[j][2] > 1.f)) inb |= 1u << j;
    }
  }
  float feat[16];
#pragma unroll
  for (int level = 0; level < 16; ++level) {
    feat[level] = 0.f;
    if (level >= a.L) continue;
    const uint32_t hs = a.offsets[level + 1] - a.offsets[level];
    const float scale = exp2f(level * a.Sl) * a.H - 1.0f;
    const uint32_t res = (uint32_t)ceilf(scale) + 1;
    const TT* tab = (const TT*)a.table + (long)a.offsets[level];
    const float gs = (float)a.grid_sizes[level];
    float acc = 0.f;
    uint32_t cur[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu};   // the cell whose corner values are in cv (see zip_fwd_all_body)
    float cv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j >= a.n || !((inb >> j) & 1u)) continue;
      const float sd = SDI[j];
      const float we = erff(1.f / sqrtf(8.f * sd * sd * gs * gs));
      float fr[3];
      uint32_t pg[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        zip_cell(X[j][k], scale, &pg[k], &fr[k]);
      }
      const bool newcell = pg[0] != cur[0] || pg[1] != cur[1] || pg[2] != cur[2];
      cur[0] = pg[0]; cur[1] = pg[1]; cur[2] = pg[2];
      float pa[8];
#pragma unroll
      for (int yz = 0; yz < 4; ++yz) {
        if (newcell) {
          uint32_t pl[3] = {pg[0], pg[1] + (yz & 1), pg[2] + (yz >> 1)};
          const long r0 = zip_grid_index(hs, res, pl);
          pl[0] = pg[0] + 1;
          const long r1 = zip_grid_index(hs, res, pl);
          // (fp32 tables: the aligned-pair 8-byte load that helps the training forward, zip_fwd_all_body, costs this kernel 20 % on the
          // 8-level proposal grid -- 5.04 -> 6.06 ms per 4.2 M intervals, round 3 -- and stays out)
          if (sizeof(TT) == 2 && (r0 ^ r1) == 1) {
            const uint32_t word = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint16_t*>(tab) + (r0 & ~1L));
            const uint16_t lo16 = (uint16_t)(word & 0xffffu), hi16 = (uint16_t)(word >> 16);
            const uint16_t b0 = (r0 & 1) ? hi16 : lo16, b1 = (r0 & 1) ? lo16 : hi16;
            if constexpr (sizeof(TT) == 2) { cv[2 * yz] = (float)__builtin_bit_cast(TT, b0); cv[2 * yz + 1] = (float)__builtin_bit_cast(TT, b1); }
          } else {
            cv[2 * yz] = (float)tab[r0];
            cv[2 * yz + 1] = (float)tab[r1];
          }
        }
        const float v0 = cv[2 * yz], v1 = cv[2 * yz + 1];
        float wa = 1.f - fr[0], wb = fr[0];
        wa *= (yz & 1) ? fr[1] : 1.f - fr[1]; wb *= (yz & 1) ? fr[1] : 1.f - fr[1];
        wa *= (yz >> 1) ? fr[2] : 1.f - fr[2]; wb *= (yz >> 1) ? fr[2] : 1.f - fr[2];
        pa[2 * yz] = (wa * we) * v0;
        pa[2 * yz + 1] = (wb * we) * v1;
      }
#pragma unroll
      for (int idx = 0; idx < 8; ++idx) acc += pa[idx];
    }
    feat[level] = zip_rbf(acc / (float)a.n, w.rnd);
  }
  // ---- proposal MLP on the registers
  const float* b1 = lw + nw1;
  const float* w2 = b1 + w.hidden;
  float out = 0.f;
  for (int h = 0; h < w.hidden; ++h) {
    const float* wr = lw + h * a.L;
    float acc = 0.f;
#pragma unroll
    for (int l = 0; l < 16; ++l) acc += l < a.L ? feat[l] * wr[l] : 0.f;
    acc += b1[h];
    out += zip_rbf(fmaxf(acc, 0.f), w.rnd) * w2[h];
  }
  w.raw_density[p] = out + lw[nw1 + 2 * w.hidden];
}

extern "C" int snerf_zip_encode_prop_fwd(const float* tdist, const float* origins, const float* directions, const float* radii, const float* base_x,
                                         const float* base_y, const float* deg_jitter, const void* table, const int* offsets, const int* grid_sizes,
                                         long R, int S, int L, int n, int m, float Sl, int H, float std_scale, int table_dtype, const float* w1,
                                         const float* b1, const float* w2, const float* b2, int hidden, int round_bf16, float* raw_density,
                                         void* stream) {
  if (R <= 0) return SNERF_OK;
  if (S <= 0 || L <= 0 || L > 16 || n <= 0 || n > 8 || hidden <= 0 || hidden > 1024 || table == nullptr || grid_sizes == nullptr || w1 == nullptr ||
      b1 == nullptr || w2 == nullptr || b2 == nullptr || raw_density == nullptr)
    return SNERF_ERR_ARG;
  ZipEnc a{tdist, origins, directions, radii, base_x, base_y, deg_jitter, table, offsets, grid_sizes, nullptr, 0, nullptr, nullptr, R, S, L, n, m, Sl, H, std_scale};
  ZipPropMlp w{w1, b1, w2, b2, hidden, round_bf16, raw_density};
  const dim3 grid((unsigned)((R * S + 255) / 256)), blk(256);
  const size_t lds = (size_t)(hidden * L + 2 * hidden + 1) * sizeof(float);
  if (table_dtype == SNERF_DT_F32) hipLaunchKernelGGL(zip_encode_prop_kernel<float>, grid, blk, lds, (hipStream_t)stream, a, w);
  else if (table_dtype == 2) hipLaunchKernelGGL(zip_encode_prop_kernel<__half>, grid, blk, lds, (hipStream_t)stream, a, w);
  else return SNERF_ERR_ARG;
  return snerf_check_launch();
}

template <typename OT, int C>
__global__ __launch_bounds__(256) void zip_encode_bwd_lds_kernel(ZipEnc a, int lds_rows) {
  extern __shared__ float lds_tab[];
  // blockIdx.y enumerates (level, slab) pairs of the LDS-privatised levels: slab s of a level covers rows [s*lds_rows, ...)
  int level = 0, slab = blockIdx.y;
  for (;; ++level) {
    const int rows = a.offsets[level + 1] - a.offsets[level];
    const int ns = (rows + lds_rows - 1) / lds_rows;
    if (slab < ns) break;
    slab -= ns;
  }
  const int rows = a.offsets[level + 1] - a.offsets[level];
  a.slab_row0 = (long)slab * lds_rows;
  a.slab_rows = min((long)lds_rows, rows - a.slab_row0);
  const int cells = (int)a.slab_rows * C;
  for (int k = threadIdx.x; k < cells; k += 256) lds_tab[k] = 0.f;
  __syncthreads();
  const long P = a.R * a.S;
  // A slab of a dense level is a range of z-layers (row = x + y (res+1) + z (res+1)^2).  When the level needs several slabs every
  // slab pass walks all intervals, so an interval is first tested against the slab with ONE contracted point: its 7 multisamples lie
  // within rho = |d| (t1 - t0) / 2 + radius * t1 / 2 of the axis midpoint, the contraction is 1-Lipschitz and the [0,1]^3 mapping
  // scales by 1/4, so their z-cells lie in a known interval; intervals that cannot touch the slab (+ one layer of margin) skip the
  // full featurisation (sincos / cbrt / erf per multisample).
  const float scale = exp2f(level * a.Sl) * a.H - 1.0f;
  const uint32_t res = (uint32_t)ceilf(scale) + 1;
  const long layer = (long)(res + 1) * (res + 1);
  const bool dense = layer * (res + 1) <= (long)rows;            // hashed levels small enough for the LDS path have no z-order
  const bool whole = !dense || (a.slab_row0 == 0 && a.slab_rows >= rows);
  const long z_lo = a.slab_row0 / layer, z_hi = (a.slab_row0 + a.slab_rows - 1) / layer;
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < P; p += (long)gridDim.x * 256) {
    if (!whole) {
      const long ray = p / a.S;
      const int i = (int)(p - ray * a.S);
      const float t0 = a.tdist[ray * (a.S + 1) + i], t1 = a.tdist[ray * (a.S + 1) + i + 1];
      const float tm = 0.5f * (t0 + t1);
      float x[3], dn = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) { const float dk = a.directions[ray * 3 + k]; x[k] = a.origins[ray * 3 + k] + dk * tm; dn += dk * dk; }
      const float rho = sqrtf(dn) * 0.5f * fabsf(t1 - t0) + a.radii[ray] * fmaxf(fabsf(t0), fabsf(t1)) * 0.5f;
      const float msq = x[0] * x[0] + x[1] * x[1] + x[2] * x[2];
      float zc = x[2];
      if (msq > 1.f) { const float mag = sqrtf(msq); zc = (2.f * mag - 1.f) / msq * x[2]; }
      const float z01 = (zc / 2.f + 1.f) / 2.f, r01 = rho * 0.25f * 1.0001f + 1e-6f;
      const float c_lo = floorf((z01 - r01) * scale + 0.5f) - 1.f, c_hi = floorf((z01 + r01) * scale + 0.5f) + 2.f;   // corner layers, +-1 margin
      if (c_hi < (float)z_lo || c_lo > (float)z_hi) continue;
    }
    zip_point_level<float, OT, C, 2>(a, p, level, lds_tab);
  }
  __syncthreads();
  float* dst = a.grad_table + ((long)a.offsets[level] + a.slab_row0) * C;
  for (int k = threadIdx.x; k < cells; k += 256) {
    const float v = lds_tab[k];
    if (v != 0.f) atomicAdd(dst + k, v);
  }
}

template <typename TT, typename OT, bool BWD>
static int zip_enc_launch(ZipEnc a, int C, int lds_levels, size_t lds_bytes, int lds_slabs, hipStream_t s) {
  const dim3 blk(256);
  const int lds_rows = (int)(lds_bytes / (C * 4));
  if (BWD && lds_levels > 0) {
    const dim3 grid(256, lds_slabs);
    switch (C) {
      case 1: (void)hipFuncSetAttribute((const void*)zip_encode_bwd_lds_kernel<OT, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
              hipLaunchKernelGGL((zip_encode_bwd_lds_kernel<OT, 1>), grid, blk, lds_bytes, s, a, lds_rows); break;
      case 2: (void)hipFuncSetAttribute((const void*)zip_encode_bwd_lds_kernel<OT, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
              hipLaunchKernelGGL((zip_encode_bwd_lds_kernel<OT, 2>), grid, blk, lds_bytes, s, a, lds_rows); break;
      case 4: (void)hipFuncSetAttribute((const void*)zip_encode_bwd_lds_kernel<OT, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
              hipLaunchKernelGGL((zip_encode_bwd_lds_kernel<OT, 4>), grid, blk, lds_bytes, s, a, lds_rows); break;
      case 8: (void)hipFuncSetAttribute((const void*)zip_encode_bwd_lds_kernel<OT, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
              hipLaunchKernelGGL((zip_encode_bwd_lds_kernel<OT, 8>), grid, blk, lds_bytes, s, a, lds_rows); break;
      default: return SNERF_ERR_ARG;
    }
  }
  if (!BWD && a.n <= 8) {
    const int lpt = a.level_begin > 0 ? min(a.level_begin, a.L) : 1;        // levels per thread (the caller's locality hint)
    a.level_begin = lpt;
    const dim3 grid((unsigned)((a.R * a.S + 255) / 256), (a.L + lpt - 1) / lpt);
    switch (C) {
      case 1: hipLaunchKernelGGL((zip_encode_fwd_all_kernel<TT, OT, 1>), grid, blk, 0, s, a, ZipBin{}); break;
      case 2: hipLaunchKernelGGL((zip_encode_fwd_all_kernel<TT, OT, 2>), grid, blk, 0, s, a, ZipBin{}); break;
      case 4: hipLaunchKernelGGL((zip_encode_fwd_all_kernel<TT, OT, 4>), grid, blk, 0, s, a, ZipBin{}); break;
      case 8: hipLaunchKernelGGL((zip_encode_fwd_all_kernel<TT, OT, 8>), grid, blk, 0, s, a, ZipBin{}); break;
      default: return SNERF_ERR_ARG;
    }
    return snerf_check_launch();
  }
  a.level_begin = BWD ? lds_levels : 0;
  const int nl = a.L - a.level_begin;
  if (nl > 0) {
    const dim3 grid((unsigned)((a.R * a.S + 255) / 256), nl);
    switch (C) {
      case 1: hipLaunchKernelGGL((zip_encode_kernel<TT, OT, 1, BWD>), grid, blk, 0, s, a); break;
      case 2: hipLaunchKernelGGL((zip_encode_kernel<TT, OT, 2, BWD>), grid, blk, 0, s, a); break;
      case 4: hipLaunchKernelGGL((zip_encode_kernel<TT, OT, 4, BWD>), grid, blk, 0, s, a); break;
      case 8: hipLaunchKernelGGL((zip_encode_kernel<TT, OT, 8, BWD>), grid, blk, 0, s, a); break;
      default: return SNERF_ERR_ARG;
    }
  }
  return snerf_check_launch();
}

template <bool BWD>
static int zip_enc_dispatch(const ZipEnc& a, int C, int table_dtype, int feat_dtype, int lds_levels, size_t lds_bytes, int lds_slabs, hipStream_t s) {
  if (table_dtype == SNERF_DT_F32 && feat_dtype == SNERF_DT_F32) return zip_enc_launch<float, float, BWD>(a, C, lds_levels, lds_bytes, lds_slabs, s);
  if (table_dtype == SNERF_DT_F32 && feat_dtype == SNERF_DT_BF16) return zip_enc_launch<float, __bf16, BWD>(a, C, lds_levels, lds_bytes, lds_slabs, s);
  if (table_dtype == 2 && feat_dtype == SNERF_DT_F32) return zip_enc_launch<__half, float, BWD>(a, C, lds_levels, lds_bytes, lds_slabs, s);
  if (table_dtype == 2 && feat_dtype == SNERF_DT_BF16) return zip_enc_launch<__half, __bf16, BWD>(a, C, lds_levels, lds_bytes, lds_slabs, s);
  if (table_dtype == SNERF_DT_F32 && feat_dtype == SNERF_DT_F16) return zip_enc_launch<float, _Float16, BWD>(a, C, lds_levels, lds_bytes, lds_slabs, s);
  if (table_dtype == 2 && feat_dtype == SNERF_DT_F16) return zip_enc_launch<__half, _Float16, BWD>(a, C, lds_levels, lds_bytes, lds_slabs, s);
  return SNERF_ERR_ARG;
}

extern "C" int snerf_zip_encode_fwd(const float* tdist, const float* origins, const float* directions, const float* radii,
                                    const float* base_x, const float* base_y, const float* deg_jitter, const void* table,
                                    const int* offsets, const int* grid_sizes, void* feat, long ld, long R, int S, int L, int C, int n,
                                    int m, float Sl, int H, float std_scale, int table_dtype, int feat_dtype, int levels_per_thread, void* stream) {
  if (R <= 0) return SNERF_OK;
  if (S <= 0 || L <= 0 || n <= 0 || ld < (long)L * C || table == nullptr || feat == nullptr || grid_sizes == nullptr || levels_per_thread < 0) return SNERF_ERR_ARG;
  ZipEnc a{tdist, origins, directions, radii, base_x, base_y, deg_jitter, table, offsets, grid_sizes, feat, ld, nullptr, nullptr, R, S, L, n, m, Sl, H, std_scale};
  a.level_begin = levels_per_thread;
  return zip_enc_dispatch<false>(a, C, table_dtype, feat_dtype, 0, 0, 0, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------------------------
// Proposal MLP of a TRAINING step as two launches (round 3): Linear(L -> hidden <= 64) + ReLU + Linear(hidden -> 1) on the grid
// features (internal/models.py:425-427, 481-519 with disable_rgb) -- 1.2 kFLOP per interval, which the per-layer GEMM route turned into
// eight HBM-bound launches over [P, 64]-wide padded buffers per level and step (features, hidden activations and both their
// gradients: ~4 GB).  Here the features live in a compact [P, ld >= L] buffer, a thread owns an interval, the weights sit in LDS
// (read as broadcasts), and the hidden activations never leave the registers: the backward recomputes them from the features.
//   forward:  raw[p] = b2 + sum_h rb(relu(b1[h] + sum_l F[p, l] rb(W1[h, l]))) rb(w2[h])          (rb = bf16 rounding in bf16 mode:
//             the same rounding points as the GEMM route -- bf16 weights, bf16 stored activations, fp32 accumulation)
//   backward: g = rb(d raw[p]);  dh[h] = H[h] > 0 ? rb(g rb(w2[h])) : 0;  dF[p, l] = rb(sum_h dh[h] rb(W1[h, l]));
//             dW1[h, l] = sum_p dh[h] F[p, l];  db1[h] = sum_p dh[h];  dw2[h] = sum_p g H[h];  db2 = sum_p d raw[p]
// The weight-gradient sums: a workgroup walks tiles of 256 intervals; per tile the threads leave dh / H / F / g in LDS, then thread
// (h = tid & 63, q = tid >> 6) adds the 64 intervals of quarter q into its register sums (the interval's F and g are broadcast reads);
// at the end the four quarters meet in LDS and the workgroup stores ONE partial row; zip_prop_fold_kernel adds the rows in workgroup
// order: bit-reproducible, no atomics.
// ------------------------------------------------------------------------------------------------------------------
#define ZPM_H 64                                         // hidden units the kernels are laid out for (fewer: idle lanes)
#define ZPM_L 16                                         // features per interval at most
struct ZipPropTrain {
  const void* F; long ldf;                               // features [P, ldf] (T)
  const float *w1, *b1, *w2, *b2;                        // parameters, fp32 (density_layer.0 / .2)
  int L, hidden, rnd;
  long P;
  float* raw;                                            // forward: [P] fp32
  const float* d_raw;                                    // backward: [P] fp32
  void* dF; long lddf;                                   // backward: [P, lddf] (T); columns >= L are written as zeros up to lddf
  float* ws;                                             // backward: [workgroups, hidden (L + 2) + 1] partial sums
};

typedef float zpm_f2 __attribute__((ext_vector_type(2)));
template <int RND> __device__ __forceinline__ float zpm_rb(float v) {          // rounding mode of the compute dtype: 0 none, 1 bf16, 2 fp16
  if constexpr (RND == 1) return (float)(__bf16)v; else if constexpr (RND == 2) return (float)(_Float16)v; else return v;
}
// weights in LDS, ZERO-PADDED to ZPM_H hidden units x LM feature slots: [h][LM] W1 | b1[ZPM_H] | w2[ZPM_H] | b2 -- a padded unit has
// b1 = w2 = 0 (its activation and gradient are 0), a padded feature slot multiplies zeros: the kernels' loops have fixed trip counts
// and no L / hidden predicates
template <int LM, int RND>
__device__ __forceinline__ void zpm_load_weights(const ZipPropTrain& w, float* lw) {
  for (int k = threadIdx.x; k < ZPM_H * LM; k += 256) {
    const int h = k / LM, l = k - h * LM;
    lw[k] = (h < w.hidden && l < w.L) ? zpm_rb<RND>(w.w1[h * w.L + l]) : 0.f;
  }
  for (int k = threadIdx.x; k < ZPM_H; k += 256) {
    lw[ZPM_H * LM + k] = k < w.hidden ? w.b1[k] : 0.f;
    lw[ZPM_H * LM + ZPM_H + k] = k < w.hidden ? zpm_rb<RND>(w.w2[k]) : 0.f;
  }
  if (threadIdx.x == 0) lw[ZPM_H * LM + 2 * ZPM_H] = w.b2[0];
}

// an interval's features: one 16- / 32-byte load per 8 bf16 / fp32 when the buffer allows it (`vec`: ld a multiple of 8 and >= LM,
// 16-byte aligned base; columns >= L of such a buffer are zeros or hit zero weights), element-wise otherwise
template <typename T, int LM>
__device__ __forceinline__ void zpm_load_row(const T* f, int L, bool vec, float* feat) {
  if (vec) {
#pragma unroll
    for (int c = 0; c < LM; c += 8) {
      if constexpr (sizeof(T) == 2) {
        typedef __attribute__((ext_vector_type(8))) T vec8;
        const vec8 v = *(const vec8*)(f + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) feat[c + e] = (float)v[e];
      } else {
        const f32x4 v0 = *(const f32x4*)(f + c), v1 = *(const f32x4*)(f + c + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { feat[c + e] = v0[e]; feat[c + 4 + e] = v1[e]; }
      }
    }
    // (padding columns are never trusted: a NaN there would survive the zero weights)
#pragma unroll
    for (int l = 0; l < LM; ++l) feat[l] = l < L ? feat[l] : 0.f;
  } else {
#pragma unroll
    for (int l = 0; l < LM; ++l) feat[l] = l < L ? to_f32(f[l]) : 0.f;
  }
}

template <typename T, int LM, int RND>
__global__ __launch_bounds__(256) void zip_prop_mlp_fwd_kernel(ZipPropTrain w) {
  __shared__ __attribute__((aligned(16))) float lw[ZPM_H * LM + 2 * ZPM_H + 1];
  zpm_load_weights<LM, RND>(w, lw);
  __syncthreads();
  const float* b1 = lw + ZPM_H * LM;
  const float* w2 = b1 + ZPM_H;
  const bool vec = (w.ldf % 8 == 0) && w.ldf >= LM && ((uintptr_t)w.F % 16 == 0);
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < w.P; p += (long)gridDim.x * 256) {
    float feat[LM];
    zpm_load_row<T, LM>((const T*)w.F + p * w.ldf, w.L, vec, feat);
    zpm_f2 f2[LM / 2];
#pragma unroll
    for (int k = 0; k < LM / 2; ++k) f2[k] = zpm_f2{feat[2 * k], feat[2 * k + 1]};
    float out = 0.f;
#pragma unroll 4
    for (int h = 0; h < ZPM_H; ++h) {
      const zpm_f2* wr = (const zpm_f2*)(lw + h * LM);
      zpm_f2 a2 = {b1[h], 0.f};
#pragma unroll
      for (int k = 0; k < LM / 2; ++k) a2 = __builtin_elementwise_fma(f2[k], wr[k], a2);      // (packed fp32 FMA: two slots per instruction)
      out = __builtin_fmaf(zpm_rb<RND>(fmaxf(a2[0] + a2[1], 0.f)), w2[h], out);
    }
    w.raw[p] = out + lw[ZPM_H * LM + 2 * ZPM_H];
  }
}

// LM: feature slots per interval (8 or 16 >= L).  The tile arrays hold T: in bf16 mode every value in them is a rounded bf16 already
// (two workgroups per CU); the fp32 mode keeps floats (one workgroup per CU).
template <typename T, int LM, int RND>
__global__ __launch_bounds__(256) void zip_prop_mlp_bwd_kernel(ZipPropTrain w) {
  constexpr int HP = ZPM_H + (sizeof(T) == 2 ? 2 : 1);   // row pitch of the tile arrays (written row-wise, read column-wise)
  __shared__ __attribute__((aligned(16))) float lw[ZPM_H * LM + 2 * ZPM_H + 1];
  __shared__ __attribute__((aligned(16))) T s_dh[256 * HP];
  __shared__ __attribute__((aligned(16))) T s_h[256 * HP];
  __shared__ __attribute__((aligned(16))) float s_f[256 * LM];
  __shared__ float s_g[256];
  static_assert(sizeof(T) * 256 * HP >= sizeof(float) * 4 * 64 * (LM + 2), "s_dh doubles as the quarter-reduction buffer");
  zpm_load_weights<LM, RND>(w, lw);
  const float* b1 = lw + ZPM_H * LM;
  const float* w2 = b1 + ZPM_H;
  const int tid = threadIdx.x, hh = tid & 63, q = tid >> 6;
  const bool vec = (w.ldf % 8 == 0) && w.ldf >= LM && ((uintptr_t)w.F % 16 == 0);
  const bool vec_out = sizeof(T) == 2 && w.lddf == LM && LM == 8 && ((uintptr_t)w.dF % 16 == 0);
  zpm_f2 aw1[LM / 2];
  float ab1 = 0.f, aw2 = 0.f, ab2 = 0.f;
#pragma unroll
  for (int k = 0; k < LM / 2; ++k) aw1[k] = zpm_f2{0.f, 0.f};
  const long tiles = (w.P + 255) >> 8;
  for (long t = blockIdx.x; t < tiles; t += gridDim.x) {
    __syncthreads();                                     // (weights loaded / the previous tile's sums done)
    const long p = (t << 8) + tid;
    const bool live = p < w.P;
    float feat[LM];
    zpm_load_row<T, LM>((const T*)w.F + (live ? p : 0) * w.ldf, w.L, vec, feat);
    const float draw = live ? w.d_raw[p] : 0.f;
    const float g = live ? zpm_rb<RND>(draw) : 0.f;
    ab2 += draw;
    zpm_f2 f2[LM / 2], df2[LM / 2];
#pragma unroll
    for (int k = 0; k < LM / 2; ++k) { f2[k] = zpm_f2{feat[2 * k], feat[2 * k + 1]}; df2[k] = zpm_f2{0.f, 0.f}; }
#pragma unroll 4
    for (int h = 0; h < ZPM_H; ++h) {
      const zpm_f2* wr = (const zpm_f2*)(lw + h * LM);
      zpm_f2 a2 = {b1[h], 0.f};
#pragma unroll
      for (int k = 0; k < LM / 2; ++k) a2 = __builtin_elementwise_fma(f2[k], wr[k], a2);
      const float hr = zpm_rb<RND>(fmaxf(a2[0] + a2[1], 0.f));
      const float dh = zpm_rb<RND>(g * w2[h]) * (hr > 0.f ? 1.f : 0.f);          // (a select, not a branch around the LDS read)
      const zpm_f2 dh2 = {dh, dh};
#pragma unroll
      for (int k = 0; k < LM / 2; ++k) df2[k] = __builtin_elementwise_fma(dh2, wr[k], df2[k]);
      s_dh[tid * HP + h] = from_f32<T>(dh);
      s_h[tid * HP + h] = from_f32<T>(hr);
    }
    float df[LM];
#pragma unroll
    for (int k = 0; k < LM / 2; ++k) { df[2 * k] = df2[k][0]; df[2 * k + 1] = df2[k][1]; }
#pragma unroll
    for (int l = 0; l < LM; ++l) s_f[tid * LM + l] = (live && l < w.L) ? feat[l] : 0.f;
    s_g[tid] = g;
    if (live) {
      T* o = (T*)w.dF + p * w.lddf;
      if (vec_out) {
        if constexpr (sizeof(T) == 2 && LM == 8) {
          typedef __attribute__((ext_vector_type(8))) T vec8;
          vec8 v;
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (T)df[e];
          *(vec8*)o = v;                               // (columns >= L: sums over zero weights = the zeros the layout asks for)
        }
      } else {
#pragma unroll
        for (int l = 0; l < LM; ++l) if (l < (int)w.lddf) o[l] = from_f32<T>(l < w.L ? df[l] : 0.f);
        for (int l = LM; l < (int)w.lddf; ++l) o[l] = from_f32<T>(0.f);
      }
    }
    __syncthreads();
#pragma unroll 4
    for (int r = q * 64; r < q * 64 + 64; ++r) {
      const float dh = to_f32(s_dh[r * HP + hh]);
      ab1 += dh;
      aw2 = __builtin_fmaf(s_g[r], to_f32(s_h[r * HP + hh]), aw2);
      const zpm_f2 dh2 = {dh, dh};
      const zpm_f2* fr = (const zpm_f2*)(s_f + r * LM);
#pragma unroll
      for (int k = 0; k < LM / 2; ++k) aw1[k] = __builtin_elementwise_fma(dh2, fr[k], aw1[k]);
    }
  }
  // the four quarters of every hidden unit meet in LDS (s_dh reused), in quarter order; one partial row per workgroup
  __syncthreads();
  float* red = (float*)s_dh;                             // [4][64][LM + 2]
  {
    float* mine = red + (q * 64 + hh) * (LM + 2);
#pragma unroll
    for (int k = 0; k < LM / 2; ++k) { mine[2 * k] = aw1[k][0]; mine[2 * k + 1] = aw1[k][1]; }
    mine[LM] = ab1; mine[LM + 1] = aw2;
  }
  // d b2: wave sums, then the four waves in order
  float s = ab2;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (hh == 0) s_g[q] = s;
  __syncthreads();
  float* row = w.ws + (long)blockIdx.x * (w.hidden * (w.L + 2) + 1);
  for (int e = tid; e < w.hidden * (w.L + 2); e += 256) {
    const int h = e / (w.L + 2), c = e - h * (w.L + 2);
    const int src = c < w.L ? c : LM + (c - w.L);        // (W1 row | b1 | w2) of hidden unit h
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) v += red[(k * 64 + h) * (LM + 2) + src];
    row[e] = v;
  }
  if (tid == 0) row[w.hidden * (w.L + 2)] = ((s_g[0] + s_g[1]) + s_g[2]) + s_g[3];
}

// g_w1 [hidden, L], g_b1 [hidden], g_w2 [hidden], g_b2 [1] += the workgroups' partial rows: one wave per element, lane i adds rows i,
// i + 64, ... in order, then the fixed butterfly over the lanes -- the same sum whatever the schedule
__global__ __launch_bounds__(256) void zip_prop_fold_kernel(const float* __restrict__ ws, int rows, int hidden, int L, float* __restrict__ g_w1,
                                                            float* __restrict__ g_b1, float* __restrict__ g_w2, float* __restrict__ g_b2) {
  const int n = hidden * (L + 2) + 1;
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (e >= n) return;
  float s = 0.f;
  for (int r = lane; r < rows; r += 64) s += ws[(long)r * n + e];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane != 0) return;
  if (e == n - 1) { g_b2[0] += s; return; }
  const int h = e / (L + 2), c = e - h * (L + 2);
  if (c < L) g_w1[h * L + c] += s;
  else if (c == L) g_b1[h] += s;
  else g_w2[h] += s;
}

extern "C" int snerf_zip_prop_mlp_ws_floats(int L, int hidden, long P) {
  const long tiles = (P + 255) / 256;
  const long wgs = tiles < 1024 ? (tiles < 1 ? 1 : tiles) : 1024;
  return (int)(wgs * (hidden * (L + 2) + 1));
}

extern "C" int snerf_zip_prop_mlp_fwd(const void* F, long ldf, long P, int L, const float* w1, const float* b1, const float* w2, const float* b2,
                                      int hidden, int round_bf16, int feat_dtype, float* raw, void* stream) {
  if (P <= 0) return SNERF_OK;
  if (F == nullptr || L <= 0 || L > ZPM_L || ldf < L || hidden <= 0 || hidden > ZPM_H || w1 == nullptr || b1 == nullptr || w2 == nullptr ||
      b2 == nullptr || raw == nullptr)
    return SNERF_ERR_ARG;
  ZipPropTrain w{F, ldf, w1, b1, w2, b2, L, hidden, round_bf16, P, raw, nullptr, nullptr, 0, nullptr};
  const long blocks = (P + 255) / 256;
  const dim3 grid((unsigned)(blocks < 8192 ? blocks : 8192)), blk(256);
  hipStream_t s = (hipStream_t)stream;
#define ZPM_F(T, RN) do { if (L <= 8) hipLaunchKernelGGL((zip_prop_mlp_fwd_kernel<T, 8, RN>), grid, blk, 0, s, w); \
                          else hipLaunchKernelGGL((zip_prop_mlp_fwd_kernel<T, 16, RN>), grid, blk, 0, s, w); } while (0)
  if (feat_dtype == SNERF_DT_BF16) { if (round_bf16 == 1) ZPM_F(__bf16, 1); else if (round_bf16 == 0) ZPM_F(__bf16, 0); else return SNERF_ERR_ARG; }
  else if (feat_dtype == SNERF_DT_F16) { if (round_bf16 == 2) ZPM_F(_Float16, 2); else return SNERF_ERR_ARG; }
  else if (feat_dtype == SNERF_DT_F32) { if (round_bf16 == 1) ZPM_F(float, 1); else if (round_bf16 == 0) ZPM_F(float, 0); else return SNERF_ERR_ARG; }
  else return SNERF_ERR_ARG;
#undef ZPM_F
  return snerf_check_launch();
}

// backward of snerf_zip_prop_mlp_fwd: dF [P, lddf] (columns L .. lddf - 1 zero) and g_* += the parameter gradients (fp32, the
// layouts of the parameters); ws: snerf_zip_prop_mlp_ws_floats(L, hidden, P) floats of scratch
extern "C" int snerf_zip_prop_mlp_bwd(const void* F, long ldf, const float* d_raw, long P, int L, const float* w1, const float* b1,
                                      const float* w2, const float* b2, int hidden, int round_bf16, int feat_dtype, void* dF, long lddf,
                                      float* g_w1, float* g_b1, float* g_w2, float* g_b2, float* ws, long ws_floats, void* stream) {
  if (P <= 0) return SNERF_OK;
  if (F == nullptr || d_raw == nullptr || L <= 0 || L > ZPM_L || ldf < L || lddf < L || lddf > 64 || hidden <= 0 || hidden > ZPM_H || w1 == nullptr ||
      b1 == nullptr || w2 == nullptr || b2 == nullptr || dF == nullptr || g_w1 == nullptr || g_b1 == nullptr || g_w2 == nullptr || g_b2 == nullptr ||
      ws == nullptr || ws_floats < snerf_zip_prop_mlp_ws_floats(L, hidden, P))
    return SNERF_ERR_ARG;
  ZipPropTrain w{F, ldf, w1, b1, w2, b2, L, hidden, round_bf16, P, nullptr, d_raw, dF, lddf, ws};
  const long tiles = (P + 255) / 256;
  const int wgs = (int)(tiles < 1024 ? tiles : 1024);
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(wgs), blk(256);
#define ZPM_B(T, RN) do { if (L <= 8) hipLaunchKernelGGL((zip_prop_mlp_bwd_kernel<T, 8, RN>), grid, blk, 0, s, w); \
                          else hipLaunchKernelGGL((zip_prop_mlp_bwd_kernel<T, 16, RN>), grid, blk, 0, s, w); } while (0)
  if (feat_dtype == SNERF_DT_BF16) { if (round_bf16 == 1) ZPM_B(__bf16, 1); else if (round_bf16 == 0) ZPM_B(__bf16, 0); else return SNERF_ERR_ARG; }
  else if (feat_dtype == SNERF_DT_F16) { if (round_bf16 == 2) ZPM_B(_Float16, 2); else return SNERF_ERR_ARG; }
  else if (feat_dtype == SNERF_DT_F32) { if (round_bf16 == 1) ZPM_B(float, 1); else if (round_bf16 == 0) ZPM_B(float, 0); else return SNERF_ERR_ARG; }
  else return SNERF_ERR_ARG;
#undef ZPM_B
  const int n = hidden * (L + 2) + 1;
  hipLaunchKernelGGL(zip_prop_fold_kernel, dim3((n + 3) / 4), dim3(256), 0, s, ws, wgs, hidden, L, g_w1, g_b1, g_w2, g_b2);
  return snerf_check_launch();
}

// The training forward of the binned table gradient: snerf_zip_encode_fwd with one level per thread + pass 0 of
// snerf_zip_encode_bwd_binned in the same sweep (counts [L, 1024] zeroed by the caller, wg_offsets [L, workgroups, 1024]); the backward
// then starts at its record pass.
extern "C" int snerf_zip_encode_fwd_count(const float* tdist, const float* origins, const float* directions, const float* radii,
                                          const float* base_x, const float* base_y, const float* deg_jitter, const void* table,
                                          const int* offsets, const int* grid_sizes, void* feat, long ld, long R, int S, int L, int C, int n,
                                          int m, float Sl, int H, float std_scale, int table_dtype, int feat_dtype, const int* ksplit_host,
                                          const int* level_rows_host, int* counts, void* wg_offsets, void* stream) {
  if (R <= 0) return SNERF_OK;
  if (S <= 0 || L <= 0 || L > 16 || n <= 0 || n > 8 || (C != 1 && C != 4) || ld < (long)L * C || table == nullptr || feat == nullptr ||
      grid_sizes == nullptr || ksplit_host == nullptr || level_rows_host == nullptr || counts == nullptr || wg_offsets == nullptr)
    return SNERF_ERR_ARG;
  ZipBin b{};
  b.bshift = C == 4 ? 12 : 14;
  for (int l = 0; l < L; ++l) {                         // (the same bound as snerf_zip_encode_bwd_binned)
    const long rowbins = ((long)level_rows_host[l] + (1L << b.bshift) - 1) >> b.bshift;
    if (level_rows_host[l] <= 0 || ksplit_host[l] < 1 || rowbins * ksplit_host[l] > ZB_NBMAX) return SNERF_ERR_ARG;
    b.ksplit[l] = ksplit_host[l];
  }
  b.counts = counts; b.wg_offsets = (unsigned*)wg_offsets;
  ZipEnc a{tdist, origins, directions, radii, base_x, base_y, deg_jitter, table, offsets, grid_sizes, feat, ld, nullptr, nullptr, R, S, L, n, m, Sl, H, std_scale};
  a.level_begin = 1;
  // (2-D grid, level-major launch order: the workgroups in flight gather from ONE level's table.  Launched level-fastest the same kernel
  // takes 4.3 instead of 3.2 ms per proposal level and 5.5 instead of 4.8 on the NeRF level: profiles/r5_z_pathC_launch_order_ab.txt)
  int ny = L;
  if (C == 1 && L >= 2) {
    // The single-channel kernel is VALU-issue bound and close to half of a thread's instructions are the interval's multisample geometry
    // (profiles/r6_zz_pathC_gather_bound_pmc.txt): a thread takes a DENSE coarse level (a table of at most a few MB that stays in L2) together
    // with a HASHED one, so the geometry is evaluated once for both while the blocks in flight still gather from one large table.
    int dense[16], hashed[16], nd = 0, nh = 0;
    for (int l = 0; l < L; ++l) {
      const float scale = exp2f(l * Sl) * H - 1.0f;
      const uint32_t s1 = (uint32_t)ceilf(scale) + 2u, s2 = s1 * s1, hs = (uint32_t)level_rows_host[l];
      if (s1 <= hs && s2 <= hs && s2 * s1 <= hs) dense[nd++] = l; else hashed[nh++] = l;    // (zip_grid_index's rule; only the pairing depends on it)
    }
    const int np = nd < nh ? nd : nh;
    int g = 0;
    for (int i = 0; i < np; ++i, ++g) { b.grp[g][0] = (signed char)dense[i]; b.grp[g][1] = (signed char)hashed[i]; }
    for (int i = np; i < nd; ++i, ++g) { b.grp[g][0] = (signed char)dense[i]; b.grp[g][1] = -1; }
    for (int i = np; i < nh; ++i, ++g) { b.grp[g][0] = (signed char)hashed[i]; b.grp[g][1] = -1; }
    if (np > 0) { b.ngrp = g; ny = g; }
  }
  const dim3 grid((unsigned)((R * S + 255) / 256), ny), blk(256);
  hipStream_t s = (hipStream_t)stream;
#define ZFC(TT, OT) do { if (C == 4) hipLaunchKernelGGL((zip_encode_fwd_all_kernel<TT, OT, 4, true>), grid, blk, 0, s, a, b); \
                         else hipLaunchKernelGGL((zip_encode_fwd_all_kernel<TT, OT, 1, true>), grid, blk, 0, s, a, b); } while (0)
  if (table_dtype == SNERF_DT_F32 && feat_dtype == SNERF_DT_F32) ZFC(float, float);
  else if (table_dtype == SNERF_DT_F32 && feat_dtype == SNERF_DT_BF16) ZFC(float, __bf16);
  else if (table_dtype == 2 && feat_dtype == SNERF_DT_F32) ZFC(__half, float);
  else if (table_dtype == 2 && feat_dtype == SNERF_DT_BF16) ZFC(__half, __bf16);
  else if (table_dtype == SNERF_DT_F32 && feat_dtype == SNERF_DT_F16) ZFC(float, _Float16);
  else if (table_dtype == 2 && feat_dtype == SNERF_DT_F16) ZFC(__half, _Float16);
  else return SNERF_ERR_ARG;
#undef ZFC
  return snerf_check_launch();
}

extern "C" int snerf_zip_encode_bwd(const float* tdist, const float* origins, const float* directions, const float* radii,
                                    const float* base_x, const float* base_y, const float* deg_jitter, const int* offsets,
                                    const int* grid_sizes, const void* grad_feat, long ld, float* grad_table, long R, int S, int L, int C,
                                    int n, int m, float Sl, int H, float std_scale, int feat_dtype, int lds_levels, long lds_cells,
                                    int lds_slabs, void* grad_table_bf16, void* stream) {
  if (R <= 0) return SNERF_OK;
  if (S <= 0 || L <= 0 || n <= 0 || ld < (long)L * C || grad_feat == nullptr || grad_table == nullptr || grid_sizes == nullptr) return SNERF_ERR_ARG;
  if (grad_table_bf16 != nullptr && ((C % 2 != 0 && C != 1) || ((uintptr_t)grad_table_bf16) % 4 != 0)) return SNERF_ERR_ARG;
  ZipEnc a{tdist, origins, directions, radii, base_x, base_y, deg_jitter, nullptr, offsets, grid_sizes, (void*)grad_feat, ld, grad_table, grad_table_bf16, R, S, L, n, m, Sl, H, std_scale};
  // the first lds_levels levels take the LDS-privatised path in slabs of lds_cells rows (lds_cells * C * 4 bytes <= 160 KB);
  // lds_slabs = sum over those levels of ceil(rows / lds_cells) (the host knows the level sizes)
  if (lds_levels < 0 || lds_levels > L || (lds_levels > 0 && (lds_cells <= 0 || lds_cells * C * 4 > 160 * 1024 || lds_slabs < lds_levels))) return SNERF_ERR_ARG;
  return zip_enc_dispatch<true>(a, C, SNERF_DT_F32, feat_dtype, lds_levels, (size_t)lds_cells * C * 4, lds_slabs, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------------------------
// Table gradient WITHOUT the atomic wall: bin, then accumulate in LDS ("binned" mode).
//
// The scatter of gridencoder.cu:248-340 issues one fp32 atomic per (multisample, corner, channel): 3.3 G per 65 536-ray step for the
// NeRF grid alone, against a hardware rate of 20.6 G/s that no scope / layout choice moves (profiles/r1_q) -- 79 % of the step, and
// the order of the additions (hence the low bits of the gradient) changes from run to run.  Here the same contributions are first
// written out as RECORDS (row inside its bin: 2 bytes, value: C floats), partitioned by destination -- bin = (level, range of 4096 /
// 16384 table rows (C = 4 / 1), replica) -- and then ONE workgroup per bin adds its records into an LDS image of those rows with
// 64-bit FIXED-POINT integer atomics (scale 2^e chosen per launch from max |grad_feat| so that the largest contribution maps to
// < 2^34: a relative resolution of 2^-34 of the largest entry whatever the loss scale -- round 2's fixed 2^36 turned every
// contribution below 7e-12 into zero; rounding to that grid is the only difference to exact real-number sums): LDS atomics are ~2 orders of magnitude cheaper than L2 atomics, every
// table row is written back by exactly one workgroup, and integer addition is associative, so the gradient is BIT-IDENTICAL run to
// run whatever order the records arrive in.  Levels whose rows are few but hot (the dense levels: 4913 rows take 118 M records)
// are split into K replicas that meet in a small global int64 image (again order-independent).
//   pass 0  zip_bin_emit_kernel<.., 0>  count the records per bin (per-workgroup LDS histogram, one global atomic per non-empty bin,
//           whose return value reserves the workgroup's range inside the bin)
//           (host: exclusive scan of the counts -> bin offsets)
//   pass 1  zip_bin_emit_kernel<.., 1>  write the records into the ranges pass 0 reserved per (workgroup, bin)
//   pass 2  zip_bin_accumulate_kernel   one workgroup per bin: LDS fixed-point accumulation, write-back (+= into the fp32 gradient)
//   pass 3  zip_bin_finish_kernel       fold the replicated levels' int64 image into the gradient
// ------------------------------------------------------------------------------------------------------------------
// the records of one (interval, level): same merging of consecutive multisamples in one cell as the atomic path
template <typename OT, int C, bool WRITE, bool HREC = false>
__device__ __forceinline__ void zip_emit_level(const ZipEnc& a, const ZipBin& b, long p, int level, int* lds_cnt, const long* lds_base) {
  const long ray = p / a.S;
  const int i = (int)(p - ray * a.S);
  const float t0 = a.tdist[ray * (a.S + 1) + i], t1 = a.tdist[ray * (a.S + 1) + i + 1];
  float o[3], d[3], bx[3], by[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { o[k] = a.origins[ray * 3 + k]; d[k] = a.directions[ray * 3 + k]; bx[k] = a.base_x[ray * 3 + k]; by[k] = a.base_y[ray * 3 + k]; }
  const float rad = a.radii[ray];
  const uint32_t hs = a.offsets[level + 1] - a.offsets[level];
  const float scale = exp2f(level * a.Sl) * a.H - 1.0f;
  const uint32_t res = (uint32_t)ceilf(scale) + 1;
  const int K = b.ksplit[level], rep = (int)(blockIdx.x % (unsigned)K);
  float g[C];
  const OT* gi = (const OT*)a.feat + p * a.ld + level * C;
#pragma unroll
  for (int c = 0; c < C; ++c) g[c] = (float)gi[c] / (float)a.n;
  float hmul = 1.f;
  if constexpr (HREC && WRITE) hmul = exp2f((float)(b.scale_exp[0] - ZB_HALF_SHIFT));
  uint32_t cur[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu};
  float wsum[8];
#pragma unroll
  for (int idx = 0; idx < 8; ++idx) wsum[idx] = 0.f;
  auto flush = [&]() {
    if (cur[0] == 0xffffffffu) return;
#pragma unroll
    for (int idx = 0; idx < 8; ++idx) {
      uint32_t pl[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) pl[k] = cur[k] + ((idx >> k) & 1);
      const uint32_t row = zip_grid_index(hs, res, pl);
      const int bin = zb_bin((uint32_t)row, b.bshift, K, rep);
      const int slot = atomicAdd(lds_cnt + bin, 1);
      if (WRITE) {
        const long r = lds_base[bin] + slot;
        if (r < b.capacity) {
          const unsigned lrow = row & ((1u << b.bshift) - 1u);
          if constexpr (HREC && C == 1) {              // one 4-byte record {row in bin | fp16 value}
            ((unsigned*)b.rec_val)[r] = lrow | (zb_half_bits((wsum[idx] * g[0]) * hmul) << 16);
          } else if constexpr (HREC && C == 4) {       // row + the four channels as one 8-byte store
            b.rec_row[r] = (unsigned short)lrow;
            const zb_h4 v4 = {(_Float16)((wsum[idx] * g[0]) * hmul), (_Float16)((wsum[idx] * g[1]) * hmul), (_Float16)((wsum[idx] * g[2]) * hmul), (_Float16)((wsum[idx] * g[3]) * hmul)};
            *(zb_h4*)(b.rec_val + r * 2) = v4;
          } else if constexpr (C == 1) {               // one 8-byte record {row in bin, value}: one store instead of a 2- and a 4-byte one
            const uint2 rv = {lrow, __float_as_uint(wsum[idx] * g[0])};
            *(uint2*)(b.rec_val + r * 2) = rv;
          } else if constexpr (C == 4) {               // the four channels as one 16-byte store
            b.rec_row[r] = (unsigned short)lrow;
            const f32x4 v4 = {wsum[idx] * g[0], wsum[idx] * g[1], wsum[idx] * g[2], wsum[idx] * g[3]};
            *(f32x4*)(b.rec_val + r * 4) = v4;
          } else {
            b.rec_row[r] = (unsigned short)lrow;
#pragma unroll
            for (int c = 0; c < C; ++c) b.rec_val[r * C + c] = wsum[idx] * g[c];
          }
        }
      }
      wsum[idx] = 0.f;
    }
  };
  for (int j = 0; j < a.n; ++j) {
    float x01[3], sd;
    zip_sample_point(a, ray, i, j, t0, t1, o, d, bx, by, rad, x01, &sd);
    if (x01[0] < 0.f || x01[0] > 1.f || x01[1] < 0.f || x01[1] > 1.f || x01[2] < 0.f || x01[2] > 1.f) continue;
    const float gs = (float)a.grid_sizes[level];
    const float we = erff(1.f / sqrtf(8.f * sd * sd * gs * gs));
    float fr[3];
    uint32_t pg[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      zip_cell(x01[k], scale, &pg[k], &fr[k]);
    }
    if (pg[0] != cur[0] || pg[1] != cur[1] || pg[2] != cur[2]) {
      flush();
      cur[0] = pg[0]; cur[1] = pg[1]; cur[2] = pg[2];
    }
#pragma unroll
    for (int idx = 0; idx < 8; ++idx) {
      float w = 1.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) w *= (idx & (1 << k)) ? fr[k] : 1.f - fr[k];
      wsum[idx] += w * we;
    }
  }
  flush();
}

template <typename OT, int C, int PASS, bool HREC = false>
__global__ __launch_bounds__(256) void zip_bin_emit_kernel(ZipEnc a, ZipBin b) {
  __shared__ int cnt[ZB_NBMAX];
  __shared__ long base[ZB_NBMAX];
  const int level = blockIdx.y;
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  const bool live = p < a.R * a.S;
  for (int k = threadIdx.x; k < ZB_NBMAX; k += 256) cnt[k] = 0;
  if (PASS == 1) {
    // ranges reserved by pass 0 (entries of bins this workgroup does not touch are never read)
    const unsigned* wgo1 = b.wg_offsets + ((long)level * gridDim.x + blockIdx.x) * ZB_NBMAX;
    for (int k = threadIdx.x; k < ZB_NBMAX; k += 256) base[k] = b.starts[level * ZB_NBMAX + k] + (long)wgo1[k];
    __syncthreads();
    if (live) zip_emit_level<OT, C, true, HREC>(a, b, p, level, cnt, base);
    return;
  }
  __syncthreads();
  if (live) zip_emit_level<OT, C, false>(a, b, p, level, cnt, nullptr);
  __syncthreads();
  // PASS 0 above (shared code): the workgroup's histogram.  It reserves the workgroup's range inside every bin it touches right away --
  // the bin's running count IS the offset of the range relative to the bin's start (known only after the host's scan) -- and leaves
  // it in wg_offsets[level, workgroup, bin]; pass 1 then needs no second count sweep.
  unsigned* wgo = b.wg_offsets + ((long)level * gridDim.x + blockIdx.x) * ZB_NBMAX;
  if (PASS == 0) {
    for (int k = threadIdx.x; k < ZB_NBMAX; k += 256)
      if (cnt[k] != 0) wgo[k] = (unsigned)atomicAdd(b.counts + level * ZB_NBMAX + k, cnt[k]);
    return;
  }
}

// PASS 1, direct writer, ALL LEVELS PER THREAD (round 4; the single-channel grids): the interval's multisample positions (sincos,
// contraction, cbrt: about half of the per-level writer's arithmetic) are evaluated once and kept in registers while the workgroup walks
// the levels one after the other -- the per-level ranges pass 0 reserved are reloaded between two barriers per level.  Same records in
// the same ranges as zip_bin_emit_kernel<.., 1> with grid.y = levels.
template <typename OT, int C, bool HREC>
__global__ __launch_bounds__(256) void zip_bin_emit_all_kernel(ZipEnc a, ZipBin b) {
  __shared__ int cnt[ZB_NBMAX];
  __shared__ long base[ZB_NBMAX];
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  const bool live = p < a.R * a.S;
  float X[8][3], SDI[8];
  unsigned inb = 0;
  if (live) {
    const long ray = p / a.S;
    const int i = (int)(p - ray * a.S);
    const float t0 = a.tdist[ray * (a.S + 1) + i], t1 = a.tdist[ray * (a.S + 1) + i + 1];
    float o[3], d[3], bx[3], by[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { o[k] = a.origins[ray * 3 + k]; d[k] = a.directions[ray * 3 + k]; bx[k] = a.base_x[ray * 3 + k]; by[k] = a.base_y[ray * 3 + k]; }
    const float rad = a.radii[ray];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j < a.n) {
        zip_sample_point(a, ray, i, j, t0, t1, o, d, bx, by, rad, X[j], &SDI[j]);
        if (!(X[j][0] < 0.f || X[j][0] > 1.f || X[j][1] < 0.f || X[j][1] > 1.f || X[j][2] < 0.f || X[j][2] > 1.f)) inb |= 1u << j;
      }
    }
  }
  float hmul = 1.f;
  if constexpr (HREC) hmul = exp2f((float)(b.scale_exp[0] - ZB_HALF_SHIFT));
  for (int level = 0; level < a.L; ++level) {
    __syncthreads();
    const unsigned* wgo1 = b.wg_offsets + ((long)level * gridDim.x + blockIdx.x) * ZB_NBMAX;
    for (int k = threadIdx.x; k < ZB_NBMAX; k += 256) { base[k] = b.starts[level * ZB_NBMAX + k] + (long)wgo1[k]; cnt[k] = 0; }
    __syncthreads();
    if (!live) continue;
    const uint32_t hs = a.offsets[level + 1] - a.offsets[level];
    const float scale = exp2f(level * a.Sl) * a.H - 1.0f;
    const uint32_t res = (uint32_t)ceilf(scale) + 1;
    const int K = b.ksplit[level], rep = (int)(blockIdx.x % (unsigned)K);
    const float gs = (float)a.grid_sizes[level];
    float g[C];
    const OT* gi = (const OT*)a.feat + p * a.ld + level * C;
#pragma unroll
    for (int c = 0; c < C; ++c) g[c] = (float)gi[c] / (float)a.n;
    uint32_t cur[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu};
    float wsum[8];
#pragma unroll
    for (int idx = 0; idx < 8; ++idx) wsum[idx] = 0.f;
    auto flush = [&]() __attribute__((always_inline)) {
      if (cur[0] == 0xffffffffu) return;
#pragma unroll
      for (int idx = 0; idx < 8; ++idx) {
        uint32_t pl[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) pl[k] = cur[k] + ((idx >> k) & 1);
        const uint32_t row = zip_grid_index(hs, res, pl);
        const int bin = zb_bin((uint32_t)row, b.bshift, K, rep);
        const int slot = atomicAdd(cnt + bin, 1);
        const long r = base[bin] + slot;
        if (r < b.capacity) {
          const unsigned lrow = row & ((1u << b.bshift) - 1u);
          if constexpr (HREC && C == 1) {
            ((unsigned*)b.rec_val)[r] = lrow | (zb_half_bits((wsum[idx] * g[0]) * hmul) << 16);
          } else if constexpr (C == 1) {
            const uint2 rv = {lrow, __float_as_uint(wsum[idx] * g[0])};
            *(uint2*)(b.rec_val + r * 2) = rv;
          } else if constexpr (HREC) {
            b.rec_row[r] = (unsigned short)lrow;
            const zb_h4 v4 = {(_Float16)((wsum[idx] * g[0]) * hmul), (_Float16)((wsum[idx] * g[1]) * hmul), (_Float16)((wsum[idx] * g[2]) * hmul), (_Float16)((wsum[idx] * g[3]) * hmul)};
            *(zb_h4*)(b.rec_val + r * 2) = v4;
          } else {
            b.rec_row[r] = (unsigned short)lrow;
            const f32x4 v4 = {wsum[idx] * g[0], wsum[idx] * g[1], wsum[idx] * g[2], wsum[idx] * g[3]};
            *(f32x4*)(b.rec_val + r * 4) = v4;
          }
        }
        wsum[idx] = 0.f;
      }
    };
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j >= a.n || !((inb >> j) & 1u)) continue;
      const float sd = SDI[j];
      const float we = erff(1.f / sqrtf(8.f * sd * sd * gs * gs));
      float fr[3];
      uint32_t pg[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) zip_cell(X[j][k], scale, &pg[k], &fr[k]);
      if (pg[0] != cur[0] || pg[1] != cur[1] || pg[2] != cur[2]) {
        flush();
        cur[0] = pg[0]; cur[1] = pg[1]; cur[2] = pg[2];
      }
#pragma unroll
      for (int idx = 0; idx < 8; ++idx) {
        float w = 1.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) w *= (idx & (1 << k)) ? fr[k] : 1.f - fr[k];
        wsum[idx] += w * we;
      }
    }
    flush();
  }
}

// PASS 1 with the records STAGED IN LDS (round 3).  Written straight from the emitting threads (kernel above), a workgroup's 16- /
// 8-byte records go to ~hundreds of bins in thread order: every store instruction touches 64 different lines, partially, and the
// lines are evicted from L2 long before the workgroup's other records of the same bin arrive -- the NeRF level wrote 27.6 GB for
// 13.4 GB of records at ~2 TB/s (profiles/r2_j).  Here the workgroup evaluates its 256 intervals ONCE (cell, fractions and erf weight of
// every multisample kept in registers), then runs NSUB = 4 sub-passes over corner pairs: the records of a sub-pass are counted per
// bin (LDS atomics), the counts prefix-summed, the records formed a second time and placed at offset[bin] + slot (a second round of
// the same atomics hands out the slots) of a 32 KB staging area as {row-in-bin | bin | thread, weight sum}, and streamed out in that
// order -- consecutive lanes write consecutive records of ONE (workgroup, bin) run.  The value = weight sum x the interval's feature
// gradient is formed on the way out from a 4 KB table of the workgroup's gradients.  Same records (bit for bit) in the ranges pass 0
// reserved; only their order inside a run differs, which the fixed-point accumulation does not see.  53 KB of LDS and 155 registers:
// three workgroups per CU -- occupancy decides here: measured per NeRF-level launch (65 536 rays, gpurun_out/r3v) direct writer 11.05
// ms; 4 sub-passes 7.26 ms; 8 sub-passes (121 registers, four workgroups, half the run length) 9.06; 2 sub-passes (85 KB: one
// workgroup per CU, twice the run length) 10.87; 4 sub-passes with the records kept in registers between count and placement (202
// registers: two workgroups) 8.37.  What it can buy is bounded by the run length: a hashed level of 2^21 rows has 512 row ranges, a
// workgroup's 3 584 records of a sub-pass make runs of ~7 records (112 B) -- nine runs per store instruction instead of 64 lines.
// The single-channel proposal grids (8-byte records, 128 row ranges) are faster with the direct writer (4.5 vs 5.4 ms per level).
#define ZS_NMAX 8
__device__ __forceinline__ int zs_wave_incl_scan(int v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int u = __shfl_up(v, o, 64);
    if (lane >= o) v += u;
  }
  return v;
}

template <typename OT, int C, int NSUB, bool HREC = false>
__global__ __launch_bounds__(256, NSUB == 8 ? 4 : 3) void zip_bin_write_staged_kernel(ZipEnc a, ZipBin b) {
  constexpr int CPS = 8 / NSUB;                       // corners per sub-pass
  __shared__ int cnt[ZB_NBMAX];
  __shared__ int off[ZB_NBMAX];
  __shared__ long base[ZB_NBMAX];
  __shared__ uint2 stage[256 * ZS_NMAX * CPS];
  __shared__ float gtab[256 * C];
  __shared__ int wtot[4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int level = ZIP_BY(a);
  const long p = (long)ZIP_BX(a) * 256 + tid;
  const bool live = p < a.R * a.S;
  const unsigned* wgo1 = b.wg_offsets + ((long)level * ZIP_GX(a) + ZIP_BX(a)) * ZB_NBMAX;
  for (int k = tid; k < ZB_NBMAX; k += 256) {      // (entries of bins this workgroup does not touch are garbage and never used)
    base[k] = b.starts[level * ZB_NBMAX + k] + (long)wgo1[k];
    cnt[k] = 0;
  }
  const uint32_t hs = a.offsets[level + 1] - a.offsets[level];
  const float scale = exp2f(level * a.Sl) * a.H - 1.0f;
  const uint32_t res = (uint32_t)ceilf(scale) + 1;
  const int K = b.ksplit[level], rep = (int)(ZIP_BX(a) % (unsigned)K);
  const unsigned rmask = (1u << b.bshift) - 1u;
  float hmul = 1.f;
  if constexpr (HREC) hmul = exp2f((float)(b.scale_exp[0] - ZB_HALF_SHIFT));
  // ---- the interval's multisamples, once
  uint32_t pg[ZS_NMAX][3];
  float fr[ZS_NMAX][3], we[ZS_NMAX];
  unsigned inb = 0, ends = 0;                       // bit j: multisample j is inside the grid / is the last of its run of equal cells
  if (live) {
    const long ray = p / a.S;
    const int i = (int)(p - ray * a.S);
    const float t0 = a.tdist[ray * (a.S + 1) + i], t1 = a.tdist[ray * (a.S + 1) + i + 1];
    float o[3], d[3], bx[3], by[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { o[k] = a.origins[ray * 3 + k]; d[k] = a.directions[ray * 3 + k]; bx[k] = a.base_x[ray * 3 + k]; by[k] = a.base_y[ray * 3 + k]; }
    const float rad = a.radii[ray];
    const OT* gi = (const OT*)a.feat + p * a.ld + level * C;
#pragma unroll
    for (int c = 0; c < C; ++c) gtab[tid * C + c] = (float)gi[c] / (float)a.n;
    const float gs = (float)a.grid_sizes[level];
    int prev = -1;
#pragma unroll
    for (int j = 0; j < ZS_NMAX; ++j) {
      if (j >= a.n) break;
      float x01[3], sd;
      zip_sample_point(a, ray, i, j, t0, t1, o, d, bx, by, rad, x01, &sd);
      if (x01[0] < 0.f || x01[0] > 1.f || x01[1] < 0.f || x01[1] > 1.f || x01[2] < 0.f || x01[2] > 1.f) continue;
      inb |= 1u << j;
      we[j] = erff(1.f / sqrtf(8.f * sd * sd * gs * gs));
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        zip_cell(x01[k], scale, &pg[j][k], &fr[j][k]);
      }
      // the previous in-bounds multisample ends its run here if the cell changes (the merging of zip_emit_level)
#pragma unroll
      for (int q = 0; q < ZS_NMAX; ++q)
        if (q == prev && (pg[q][0] != pg[j][0] || pg[q][1] != pg[j][1] || pg[q][2] != pg[j][2])) ends |= 1u << q;
      prev = j;
    }
    if (prev >= 0) ends |= 1u << prev;
  } else {
#pragma unroll
    for (int c = 0; c < C; ++c) gtab[tid * C + c] = 0.f;
  }
  __syncthreads();
  // one record of the sub-pass: corner (e, y, z) of the run that ends at multisample j, weight sum r
  auto record = [&](int j, int e, int y, int z, float r, bool place) __attribute__((always_inline)) {
    const uint32_t pl[3] = {pg[j][0] + (uint32_t)e, pg[j][1] + (uint32_t)y, pg[j][2] + (uint32_t)z};
    const uint32_t row = zip_grid_index(hs, res, pl);
    const int bin = zb_bin((uint32_t)row, b.bshift, K, rep);
    const int slot = atomicAdd(cnt + bin, 1);         // count phase: the count; place phase: the slot inside the bin's run (any order will do)
    if (place) stage[off[bin] + slot] = uint2{(row & rmask) | ((unsigned)bin << 14) | ((unsigned)tid << 24), __float_as_uint(r)};
  };
  // the records of corners [sp CPS, sp CPS + CPS) (corner = x + 2 y + 4 z) of this thread's interval; same products in the same order
  // as zip_emit_level: w = 1; w *= (x ? fr0 : 1 - fr0); w *= (y ? fr1 : 1 - fr1); w *= (z ? fr2 : 1 - fr2); wsum += w * we
  auto walk = [&](int sp, bool place) __attribute__((always_inline)) {
    float r[CPS];
#pragma unroll
    for (int q = 0; q < CPS; ++q) r[q] = 0.f;
#pragma unroll
    for (int j = 0; j < ZS_NMAX; ++j) {
      if (!((inb >> j) & 1u)) continue;
#pragma unroll
      for (int q = 0; q < CPS; ++q) {
        const int idx = sp * CPS + q;
        float w = 1.f;
        w *= (idx & 1) ? fr[j][0] : 1.f - fr[j][0];
        w *= (idx & 2) ? fr[j][1] : 1.f - fr[j][1];
        w *= (idx & 4) ? fr[j][2] : 1.f - fr[j][2];
        r[q] += w * we[j];
      }
      if ((ends >> j) & 1u) {
        if constexpr (CPS == 2) {
          // the sub-pass's two corners are x-neighbours (corner = x + 2 y + 4 z): on a hashed level with even x their rows differ in bit 0 only, on a
          // dense level they are adjacent -- the same bin either way, so ONE counter update hands out both slots (a lane whose two rows fall into
          // different bins takes the second update alone).  Same records in the same ranges: the order inside a run is all that changes.
          const uint32_t pl0[3] = {pg[j][0], pg[j][1] + (uint32_t)(sp & 1), pg[j][2] + (uint32_t)(sp >> 1)};
          const uint32_t pl1[3] = {pg[j][0] + 1u, pl0[1], pl0[2]};
          const uint32_t row0 = zip_grid_index(hs, res, pl0), row1 = zip_grid_index(hs, res, pl1);
          const int bin0 = zb_bin((uint32_t)row0, b.bshift, K, rep);
          int bin1 = zb_bin((uint32_t)row1, b.bshift, K, rep);
          const bool same = bin0 == bin1;
          const int slot0 = atomicAdd(cnt + bin0, same ? 2 : 1);
          int slot1 = slot0 + 1;
          if (!same) slot1 = atomicAdd(cnt + bin1, 1);
          if (place) {
            const int o0 = off[bin0];
            const int o1 = same ? o0 : off[bin1];
            stage[o0 + slot0] = uint2{(row0 & rmask) | ((unsigned)bin0 << 14) | ((unsigned)tid << 24), __float_as_uint(r[0])};
            stage[o1 + slot1] = uint2{(row1 & rmask) | ((unsigned)bin1 << 14) | ((unsigned)tid << 24), __float_as_uint(r[1])};
          }
          r[0] = 0.f; r[1] = 0.f;
        } else {
#pragma unroll
        for (int q = 0; q < CPS; ++q) { const int idx = sp * CPS + q; record(j, idx & 1, (idx >> 1) & 1, idx >> 2, r[q], place); r[q] = 0.f; }
        }
      }
    }
  };
  for (int z = 0; z < NSUB; ++z) {                    // sub-pass z: CPS corners
    walk(z, false);
    __syncthreads();
    // exclusive prefix of the 1024 counts: thread t owns bins 4 t .. 4 t + 3 (and zeroes them for the placement counters)
    const int c0 = cnt[4 * tid], c1 = cnt[4 * tid + 1], c2 = cnt[4 * tid + 2], c3 = cnt[4 * tid + 3];
    const int mine = c0 + c1 + c2 + c3;
    const int incl = zs_wave_incl_scan(mine, lane);
    if (lane == 63) wtot[wv] = incl;
    cnt[4 * tid] = 0; cnt[4 * tid + 1] = 0; cnt[4 * tid + 2] = 0; cnt[4 * tid + 3] = 0;
    __syncthreads();
    int pre = incl - mine;
#pragma unroll
    for (int q = 0; q < 4; ++q) if (q < wv) pre += wtot[q];
    const int total = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    off[4 * tid] = pre; off[4 * tid + 1] = pre + c0; off[4 * tid + 2] = pre + c0 + c1; off[4 * tid + 3] = pre + c0 + c1 + c2;
    __syncthreads();
    walk(z, true);
    __syncthreads();
    for (int sidx = tid; sidx < total; sidx += 256) {
      const uint2 rec = stage[sidx];
      const int bin = (int)((rec.x >> 14) & 1023u);
      const long r = base[bin] + (sidx - off[bin]);
      if (r < b.capacity) {
        const unsigned lrow = rec.x & 0x3fffu;
        const float wsum = __uint_as_float(rec.y);
        const float* g = gtab + (rec.x >> 24) * C;
        if constexpr (HREC && C == 1) {
          ((unsigned*)b.rec_val)[r] = lrow | (zb_half_bits((wsum * g[0]) * hmul) << 16);
        } else if constexpr (HREC) {
          b.rec_row[r] = (unsigned short)lrow;
          const zb_h4 v4 = {(_Float16)((wsum * g[0]) * hmul), (_Float16)((wsum * g[1]) * hmul), (_Float16)((wsum * g[2]) * hmul), (_Float16)((wsum * g[3]) * hmul)};
          *(zb_h4*)(b.rec_val + r * 2) = v4;
        } else if constexpr (C == 1) {
          const uint2 rv = {lrow, __float_as_uint(wsum * g[0])};
          *(uint2*)(b.rec_val + r * 2) = rv;
        } else {
          b.rec_row[r] = (unsigned short)lrow;
          const f32x4 v4 = {wsum * g[0], wsum * g[1], wsum * g[2], wsum * g[3]};
          *(f32x4*)(b.rec_val + r * 4) = v4;
        }
      }
    }
    __syncthreads();
    base[4 * tid] += c0; base[4 * tid + 1] += c1; base[4 * tid + 2] += c2; base[4 * tid + 3] += c3;
    cnt[4 * tid] = 0; cnt[4 * tid + 1] = 0; cnt[4 * tid + 2] = 0; cnt[4 * tid + 3] = 0;
    __syncthreads();
  }
}

// the records [s0, s0 + n) of one bin added into the bin's LDS image in 64-bit fixed point (shared by the zipnerf path's accumulate
// kernel and the stand-alone GridEncoder's chunked one below).  Record formats: HREC C = 1 one word {row | fp16 << 16}; HREC C = 4 a
// 2-byte row plane + an 8-byte plane of four halves; fp32 C = 1 {row, value} pairs; otherwise a row plane + C floats.
// SKIPZ: records whose values are all zero are skipped before the conversions (the stand-alone encoder's runs are padded with such records)
template <int C, bool HREC, bool SKIPZ = false>
__device__ __forceinline__ void zb_acc_records(long long* zb_acc, const unsigned short* __restrict__ rec_row, const float* __restrict__ rec_val,
                                               const long s0, const int n, const float fix, const float lim) {
  // 16 waves x 4 records per thread in flight: the record stream is latency-bound (one record per thread and trip took a bin of
  // 2 M records 4 ms whatever the LDS did)
  // (round 3: 16 per thread for both record sizes -- train step 52.3-52.9 ms at 4, 51.5-52.2 at 8 / 16, 51.3-51.6 at 16 / 16;
  // 16 waves per CU leave 128 registers per thread)
#ifndef ZB_ACC_U4
#define ZB_ACC_U4 16
#endif
#ifndef ZB_ACC_U1
#define ZB_ACC_U1 16
#endif
  constexpr int U = C == 1 ? ZB_ACC_U1 : (C == 8 ? 8 : ZB_ACC_U4);        // (C = 8: 8 records x 8 channels in flight per thread: 128-register budget)
  for (int r0 = threadIdx.x; r0 < n; r0 += 1024 * U) {
    int row[U];
    float val[U][C];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int r = r0 + u * 1024;
      const long q = s0 + (r < n ? r : r0);
      if constexpr (HREC && C == 1) {
        const unsigned rv = ((const unsigned*)rec_val)[q];
        row[u] = r < n ? (int)(rv & 0xffffu) : -1;
        val[u][0] = (float)__builtin_bit_cast(_Float16, (unsigned short)(rv >> 16));
        continue;
      }
      if constexpr (HREC && C == 4) {
        row[u] = r < n ? (int)rec_row[q] : -1;
        const zb_h4 v4 = *(const zb_h4*)(rec_val + q * 2);
        val[u][0] = (float)v4[0]; val[u][1] = (float)v4[1]; val[u][2] = (float)v4[2]; val[u][3] = (float)v4[3];
        continue;
      }
      if constexpr (C == 1) {
        const uint2 rv = *(const uint2*)(rec_val + q * 2);
        row[u] = r < n ? (int)rv.x : -1;
        val[u][0] = __uint_as_float(rv.y);
        continue;
      }
      row[u] = r < n ? (int)rec_row[q] : -1;
      if constexpr (C == 4) {
        const f32x4 v4 = *(const f32x4*)(rec_val + q * 4);
        val[u][0] = v4[0]; val[u][1] = v4[1]; val[u][2] = v4[2]; val[u][3] = v4[3];
      } else {
#pragma unroll
        for (int c = 0; c < C; ++c) val[u][c] = rec_val[q * C + c];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (row[u] < 0) continue;
      if constexpr (SKIPZ) {
        bool nz = false;
#pragma unroll
        for (int c = 0; c < C; ++c) nz |= val[u][c] != 0.f;
        if (!nz) continue;
      }
      // C = 4: channel c of row r sits in slot c ^ ((r >> 3) & 3) of the row's four 8-byte cells.  Unswizzled, one instruction (a fixed
      // channel of 64 random rows) can only reach the 8 bank pairs 4 (r mod 8) + c of the 32: an 8-way conflict whatever the rows are;
      // with the swizzle the 64 lanes spread over all 32 pairs (round 4: the LDS atomics were 2.0 of the kernel's 3.8 ms)
      const int sw = C == 4 ? (row[u] >> 3) & 3 : 0;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        // (a record is at most max |grad_feat| in magnitude -- 8 x that for the stand-alone encoder's merged runs, whose scale leaves 3
        // more head bits: the clamp only stops non-finite values)
        const float v = HREC ? fminf(fmaxf(val[u][c], -65536.f), 65536.f) * (float)(1 << ZB_HALF_SHIFT) : fminf(fmaxf(val[u][c], -lim), lim) * fix;   // half records carry value * 2^(se - ZB_HALF_SHIFT)
        atomicAdd((unsigned long long*)(zb_acc + row[u] * C + (c ^ sw)), (unsigned long long)__float2ll_rn(v));
      }
    }
  }
}

// GT: type of the gradient table the rows are added to (fp32 in the zipnerf path)
template <int C, bool HREC = false, typename GT = float>
__global__ __launch_bounds__(1024) void zip_bin_accumulate_kernel(ZipEnc a, ZipBin b) {
  extern __shared__ long long zb_acc[];
  const int level = blockIdx.y, bin = blockIdx.x;
  const int K = b.ksplit[level];
  const long rows_l = a.offsets[level + 1] - a.offsets[level];
  const long row0 = (long)(bin / K) << b.bshift;
  if (row0 >= rows_l) return;                              // bins past the level's last row range
  const int n = b.counts[level * ZB_NBMAX + bin];
  const int se = b.scale_exp[0];
  const float fix = exp2f((float)se), lim = exp2f((float)(ZB_HEAD + 1 - se));
  const double unfix = exp2((double)-se);
  const int cells = (int)min((long)(1 << b.bshift), rows_l - row0) * C;
  for (int k = threadIdx.x; k < cells; k += 1024) zb_acc[k] = 0;
  __syncthreads();
  zb_acc_records<C, HREC>(zb_acc, b.rec_row, b.rec_val, b.starts[level * ZB_NBMAX + bin], n, fix, lim);
  __syncthreads();
  const long grow = (long)a.offsets[level] + row0;
  if (K == 1) {                                             // the only workgroup that owns these rows
    GT* dst = (GT*)a.grad_table + grow * C;
    for (int k = threadIdx.x; k < cells; k += 1024) {
      const long long v = zb_acc[k];
      const int kk = C == 4 ? (k ^ ((k >> 5) & 3)) : k;     // (undo the slot swizzle: row = k >> 2, its (row >> 3) & 3 = (k >> 5) & 3)
      if (v != 0) dst[kk] = (GT)((float)dst[kk] + (float)((double)v * unfix));
    }
  } else {
    long long* dst = b.g64 + grow * C;
    for (int k = threadIdx.x; k < cells; k += 1024) {
      const long long v = zb_acc[k];
      const int kk = C == 4 ? (k ^ ((k >> 5) & 3)) : k;
      if (v != 0) atomicAdd((unsigned long long*)(dst + kk), (unsigned long long)v);
    }
  }
}

template <typename GT = float>
__global__ __launch_bounds__(256) void zip_bin_finish_kernel(const long long* __restrict__ g64, long n, GT* __restrict__ grad, const int* __restrict__ scale_exp) {
  const double unfix = exp2((double)-scale_exp[0]);
  for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < n; k += (long)gridDim.x * 256) {
    const long long v = g64[k];
    if (v != 0) grad[k] = (GT)((float)grad[k] + (float)((double)v * unfix));
  }
}

// scale_exp[0] = e with 2^e * max |grad_feat| in [2^(ZB_HEAD - 1), 2^ZB_HEAD) (clamped to the normal range of a float; 36, round 2's
// constant, when the gradient is all zeros or not finite); scale_exp[1]: scratch (bits of the running maximum)
template <typename OT>
__global__ __launch_bounds__(256) void zip_bin_absmax_kernel(const OT* __restrict__ g, long ld, long rows, int cols, unsigned* __restrict__ mx) {
  float m = 0.f;
  const long total = rows * cols;
  if (ld == cols && (((uintptr_t)g) & 15) == 0) {
    // contiguous (the stand-alone GridEncoder's [B, L*C] / [L, B, C] gradients): 16-byte loads, no index division -- the strided form
    // below took 0.75 ms for 1.2 GB
    constexpr int E = 16 / (int)sizeof(OT);
    const long nv = total / E;
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < nv; v += (long)gridDim.x * 256) {
      const uint4 q = ((const uint4*)g)[v];
      const OT* e = (const OT*)&q;
#pragma unroll
      for (int k = 0; k < E; ++k) { const float x = fabsf((float)e[k]); m = fmaxf(m, x != x ? __builtin_inff() : x); }
    }
    for (long e = nv * E + (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
      const float x = fabsf((float)g[e]);
      m = fmaxf(m, x != x ? __builtin_inff() : x);
    }
  } else
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    // (rows at a pitch: a wave per row with lanes over 40 two-byte columns was tried and is 2.5x slower than this element-per-thread form)
    const long r = e / cols;
    const float x = fabsf((float)g[r * ld + (e - r * cols)]);
    m = fmaxf(m, x != x ? __builtin_inff() : x);                                   // (a NaN counts as an overflow: zip_bin_overflow_mark_kernel)
  }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(mx, __float_as_uint(m));      // (non-negative floats order like their bit patterns)
}
// an overflowed feature gradient (fp16 compute mode: +-Inf / NaN in grad_feat) cannot be scaled into fixed point: the level's records
// are meaningless.  Mark the table gradient by a NaN in its first element so that whoever checks the gradients for overflow (a
// dynamic loss scaler: snerf_nonfinite_flag, or torch's GradScaler.unscale_) skips the step; the static policy drops it in Adam.
template <typename GT = float>
__global__ void zip_bin_overflow_mark_kernel(GT* __restrict__ grad, const int* __restrict__ scale_exp) {
  if ((unsigned)scale_exp[1] >= 0x7f800000u) grad[0] = (GT)__builtin_nanf("");
}
// head: extra head bits (a record of the stand-alone encoder is the sum of up to 8 merged contributions: 3)
__global__ void zip_bin_scale_kernel(int* scale_exp, int head) {
  const unsigned bits = (unsigned)scale_exp[1];
  int e = 36 - head;
  if (bits != 0 && bits < 0x7f800000u) {
    const int ex = (int)(bits >> 23) - 127;                  // max < 2^(ex + 1) (a denormal maximum counts as 2^-127)
    e = ZB_HEAD - head - (ex + 1);
    e = e > 127 ? 127 : (e < -100 ? -100 : e);
  }
  scale_exp[0] = e;
}

extern "C" int snerf_zip_bin_scale(const void* grad_feat, long ld, long rows, int cols, int feat_dtype, int* scale_exp, void* stream) {
  if (scale_exp == nullptr || (rows > 0 && (grad_feat == nullptr || cols <= 0 || ld < cols))) return SNERF_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  (void)hipMemsetAsync(scale_exp, 0, 2 * sizeof(int), s);
  if (rows > 0) {
    const long total = rows * cols;
    const int grid = (int)(total / 256 / 8 + 1 < 2048 ? total / 256 / 8 + 1 : 2048);
    if (feat_dtype == SNERF_DT_BF16) hipLaunchKernelGGL(zip_bin_absmax_kernel<__bf16>, dim3(grid), dim3(256), 0, s, (const __bf16*)grad_feat, ld, rows, cols, (unsigned*)(scale_exp + 1));
    else if (feat_dtype == SNERF_DT_F16) hipLaunchKernelGGL(zip_bin_absmax_kernel<_Float16>, dim3(grid), dim3(256), 0, s, (const _Float16*)grad_feat, ld, rows, cols, (unsigned*)(scale_exp + 1));
    else if (feat_dtype == SNERF_DT_F32) hipLaunchKernelGGL(zip_bin_absmax_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)grad_feat, ld, rows, cols, (unsigned*)(scale_exp + 1));
    else return SNERF_ERR_ARG;
  }
  hipLaunchKernelGGL(zip_bin_scale_kernel, dim3(1), dim3(1), 0, s, scale_exp, 0);
  return snerf_check_launch();
}

// pass 2 of the binned table gradient: one workgroup per bin, the replicated levels' int64 image folded, the overflow mark
static int zb_accumulate_launch(const ZipEnc& a, const ZipBin& b, int C, int L, bool hrec, hipStream_t s) {
  const size_t lds = (size_t)(1 << b.bshift) * C * 8;
  const dim3 grid(ZB_NBMAX, L);
#define ZBA(CC, HH) do { (void)hipFuncSetAttribute((const void*)zip_bin_accumulate_kernel<CC, HH, float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                         hipLaunchKernelGGL((zip_bin_accumulate_kernel<CC, HH, float>), grid, dim3(1024), lds, s, a, b); } while (0)
  if (hrec && C == 4) ZBA(4, true);
  else if (hrec && C == 1) ZBA(1, true);
  else if (C == 4) ZBA(4, false);
  else if (C == 1) ZBA(1, false);
  else return SNERF_ERR_ARG;
#undef ZBA
  if (b.g64 != nullptr && b.g64_rows > 0)
    hipLaunchKernelGGL(zip_bin_finish_kernel<float>, dim3(1024), dim3(256), 0, s, (const long long*)b.g64, b.g64_rows * C, a.grad_table, b.scale_exp);
  hipLaunchKernelGGL(zip_bin_overflow_mark_kernel<float>, dim3(1), dim3(1), 0, s, a.grad_table, b.scale_exp);
  return snerf_check_launch();
}

// pass: 0 = count (+ reserve), 1 = write records, 2 = accumulate (+ finish).  The host zeroes counts / g64 and scans counts into starts.
extern "C" int snerf_zip_encode_bwd_binned(int pass, const float* tdist, const float* origins, const float* directions, const float* radii,
                                           const float* base_x, const float* base_y, const float* deg_jitter, const int* offsets,
                                           const int* grid_sizes, const void* grad_feat, long ld, float* grad_table, long R, int S, int L, int C,
                                           int n, int m, float Sl, int H, float std_scale, int feat_dtype, const int* ksplit_host,
                                           const int* level_rows_host, int* counts, void* wg_offsets, const long* starts, void* rec_row,
                                           float* rec_val, long capacity, void* g64, long g64_rows, const int* scale_exp, void* stream) {
  if (R <= 0) return SNERF_OK;
  if (S <= 0 || L <= 0 || L > 16 || n <= 0 || (C != 1 && C != 4) || ksplit_host == nullptr || level_rows_host == nullptr || counts == nullptr)
    return SNERF_ERR_ARG;
  // a level's bins = row ranges x replicas must fit the ZB_NBMAX-entry histograms of the kernels (LDS cnt / base, counts [L, ZB_NBMAX],
  // the accumulate grid): a table past 2^22 (C = 4) / 2^24 (C = 1) rows per level has more row ranges than that -- refuse it here
  // instead of corrupting LDS and dropping gradients (the caller falls back to the atomic scatter)
  for (int l = 0; l < L; ++l) {
    const long rowbins = ((long)level_rows_host[l] + (1L << (C == 4 ? 12 : 14)) - 1) >> (C == 4 ? 12 : 14);
    if (level_rows_host[l] <= 0 || ksplit_host[l] < 1 || rowbins * ksplit_host[l] > ZB_NBMAX) return SNERF_ERR_ARG;
  }
  ZipEnc a{tdist, origins, directions, radii, base_x, base_y, deg_jitter, nullptr, offsets, grid_sizes, (void*)grad_feat, ld, grad_table, nullptr, R, S, L, n, m, Sl, H, std_scale};
  ZipBin b{};
  b.bshift = C == 4 ? 12 : 14;
  b.counts = counts; b.wg_offsets = (unsigned*)wg_offsets; b.starts = starts;
  for (int l = 0; l < L; ++l) { b.ksplit[l] = ksplit_host[l]; if (b.ksplit[l] < 1) return SNERF_ERR_ARG; }
  b.rec_row = (unsigned short*)rec_row; b.rec_val = rec_val; b.capacity = capacity; b.g64 = (long long*)g64; b.g64_rows = g64_rows;
  b.scale_exp = scale_exp;
  hipStream_t s = (hipStream_t)stream;
  const dim3 blk(256);
  // passes 5 / 6 / 7 = 1 / 3 / 2 with HALF records (see ZB_HALF_SHIFT): the writers then need scale_exp too
  // passes 8 / 9 (probes): the all-levels direct writer whatever C, with fp32 / half records
  const bool hrec = (pass >= 5 && pass <= 7) || pass == 9;
  const bool force_all = pass == 8 || pass == 9;
  if (force_all) pass = 1;
  else if (hrec) pass = pass == 5 ? 1 : (pass == 6 ? 3 : 2);
  if (pass == 0 || pass == 1 || pass == 3 || pass == 4) {
    if (grad_feat == nullptr || wg_offsets == nullptr || (pass != 0 && (starts == nullptr || rec_row == nullptr || rec_val == nullptr)) || (hrec && scale_exp == nullptr))
      return SNERF_ERR_ARG;
    const dim3 grid((unsigned)((R * S + 255) / 256), L);
    // pass 1: records staged in LDS and written run by run (zip_bin_write_staged_kernel); pass 3 (A/B probes, or more than 8
    // multisamples): every thread writes its records where they fall
    // (C = 1: 8-byte records in 128 row ranges per level -- the direct writer is faster there: 4.6 vs 5.8 ms per proposal level)
    const bool staged = ((pass == 1 && C == 4 && !force_all) || pass == 4) && n <= ZS_NMAX;     // (pass 4: probe -- staged for C = 1 too)
    // pass 1 at C = 1: the direct writer with all levels per thread (pass 3 keeps the one-level-per-thread form for A/B runs and tests)
    const bool all_levels = pass == 1 && (C == 1 || force_all) && n <= 8;
    const dim3 grid1((unsigned)((R * S + 255) / 256), 1);
    // the staged writer gathers nothing: it is launched LEVEL-FASTEST (1-D grid), so that the L workgroups of one block of intervals run
    // together and the feature-gradient rows they all read (8-byte pieces of [P, ld]) and the rays' geometry come from HBM once:
    // 6.0 -> 5.2 ms per NeRF-level launch (profiles/r5_z_pathC_launch_order_ab.txt)
    ZipEnc a_st = a;
    dim3 grid_st = grid;
    static int lf_on = -1;
    if (lf_on < 0) { const char* e = getenv("SNERF_ZIP_WRITER_2D"); lf_on = (e != nullptr && e[0] == '1') ? 0 : 1; }   // (A/B switch)
    if (lf_on) { a_st.lf = L; grid_st = dim3(grid.x * (unsigned)L); }
#define ZBE(OT, CC) do { if (pass == 0) hipLaunchKernelGGL((zip_bin_emit_kernel<OT, CC, 0>), grid, blk, 0, s, a, b); \
                         else if (all_levels && hrec) hipLaunchKernelGGL((zip_bin_emit_all_kernel<OT, CC, true>), grid1, blk, 0, s, a, b); \
                         else if (all_levels) hipLaunchKernelGGL((zip_bin_emit_all_kernel<OT, CC, false>), grid1, blk, 0, s, a, b); \
                         else if (staged && hrec) hipLaunchKernelGGL((zip_bin_write_staged_kernel<OT, CC, 4, true>), grid_st, blk, 0, s, a_st, b); \
                         else if (staged) hipLaunchKernelGGL((zip_bin_write_staged_kernel<OT, CC, 4>), grid_st, blk, 0, s, a_st, b); \
                         else if (hrec) hipLaunchKernelGGL((zip_bin_emit_kernel<OT, CC, 1, true>), grid, blk, 0, s, a, b); \
                         else hipLaunchKernelGGL((zip_bin_emit_kernel<OT, CC, 1>), grid, blk, 0, s, a, b); } while (0)
    if (feat_dtype == SNERF_DT_BF16) { if (C == 4) ZBE(__bf16, 4); else ZBE(__bf16, 1); }
    else if (feat_dtype == SNERF_DT_F16) { if (C == 4) ZBE(_Float16, 4); else ZBE(_Float16, 1); }
    else if (feat_dtype == SNERF_DT_F32) { if (C == 4) ZBE(float, 4); else ZBE(float, 1); }
    else return SNERF_ERR_ARG;
#undef ZBE
    return snerf_check_launch();
  }
  if (pass != 2 || starts == nullptr || rec_row == nullptr || rec_val == nullptr || grad_table == nullptr || scale_exp == nullptr) return SNERF_ERR_ARG;
  return zb_accumulate_launch(a, b, C, L, hrec, s);
}

// ------------------------------------------------------------------------------------------------------------------
// The stand-alone GridEncoder (the reference's only native FFI: gridencoder/src/bindings.cpp:5-9) on the same machinery, for the
// instantiations zipnerf constructs (internal/models.py:413-421): D = 3, hash grid type, linear interpolation, align_corners = False,
// C = 4 (NeRF grid) / 1 (proposal grids), float or half table.  Arbitrary points [B, 3] in [0, 1]^3 instead of ray geometry:
//   forward   g3_fwd_kernel       one thread per (point, level), x-neighbour corners as ONE 4- / 8- / 16-byte load where their rows are
//             adjacent; point-major thread order for 8-byte entries so that [B, L*C] leaves in contiguous runs -- kernel_grid
//             (gridencoder.cu:87-245) fetches 8 rows per (point, level) whatever the neighbours did;
//   backward  g3_count / g3_write_staged / g3_accumulate: the binned table gradient above fed from points: a run of consecutive points
//             in one cell becomes 8 RECORDS (row, sum of w x grad over the run); level by level and chunk by chunk, count -> device scan ->
//             LDS-staged write -> LDS fixed-point accumulation, in ONE C-ABI call on a caller-provided BOUNDED workspace (no atomics on
//             the table, bit-reproducible) -- kernel_grid_backward (gridencoder.cu:248-340) issues 8 x C / 2 half2 atomics per
//             (point, level): 20.6 G atomics/s on this part.
// Everything else (D != 3, tiled, smoothstep, align_corners, double, dy_dx) stays in grid.hip.
// ------------------------------------------------------------------------------------------------------------------
struct G3Args {
  const float* inputs; long B;
  const void* table; const int* offsets;
  void* io; long s_l, s_b;                 // forward: outputs; backward: the incoming gradient; strides (elements) of the level / point axes
  int L; float Sl; int H;
};

template <typename TT, int C, int G>
__global__ __launch_bounds__(256) void g3_fwd_kernel(G3Args a) {
#pragma clang fp contract(off)        // every product and sum rounded on its own: the same bits as grid.hip's grid_fwd_kernel whatever the compiler fuses elsewhere
  // G >= 1: level-major grid (blockIdx.y = level: the blocks in flight gather from one level's table), G consecutive points per thread.
  // G == 0: point-major -- thread t = point * L + level, so that a workgroup's outputs are ONE contiguous run of [B, L*C] (the
  // level-major form writes 8-byte pieces 80 bytes apart: 4x the bytes at the memory side, profiles/r5_x_grid_encoder_pmc.txt)
  constexpr int GP = G == 0 ? 1 : G;
  const long t0 = (long)blockIdx.x * 256 + threadIdx.x;
  const int level = G == 0 ? (int)(t0 % a.L) : (int)blockIdx.y;
  const long p0 = G == 0 ? t0 / a.L : t0 * GP;
  if (p0 >= a.B) return;
  const uint32_t hs = a.offsets[level + 1] - a.offsets[level];
  const float scale = exp2f(level * a.Sl) * a.H - 1.0f;
  const uint32_t res = (uint32_t)ceilf(scale) + 1;
  const TT* tab = (const TT*)a.table + (long)a.offsets[level] * C;
  uint32_t cur[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu};
  ZVec<TT, C> ce[8];
#pragma unroll
  for (int j = 0; j < GP; ++j) {
    const long p = p0 + j;
    if (p >= a.B) break;
    TT* out = (TT*)a.io + level * a.s_l + p * a.s_b;
    float x[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) x[k] = a.inputs[p * 3 + k];
    if (x[0] < 0.f || x[0] > 1.f || x[1] < 0.f || x[1] > 1.f || x[2] < 0.f || x[2] > 1.f) {   // (a NaN coordinate passes, as in kernel_grid)
#pragma unroll
      for (int c = 0; c < C; ++c) out[c] = (TT)0.f;
      continue;
    }
    float fr[3];
    uint32_t pg[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) zip_cell(x[k], scale, &pg[k], &fr[k]);
    const bool newcell = GP == 1 || pg[0] != cur[0] || pg[1] != cur[1] || pg[2] != cur[2];
    cur[0] = pg[0]; cur[1] = pg[1]; cur[2] = pg[2];
    if (newcell) {
#pragma unroll
      for (int yz = 0; yz < 4; ++yz) {
        uint32_t pl[3] = {pg[0], pg[1] + (yz & 1), pg[2] + (yz >> 1)};
        const long r0 = zip_grid_index(hs, res, pl);
        pl[0] = pg[0] + 1;
        const long r1 = zip_grid_index(hs, res, pl);
        // x-neighbours in adjacent rows (always on dense levels with an even row, every even x of a hashed level: the hash multiplies x
        // by 1) arrive as ONE aligned load of both entries
        if ((r0 ^ r1) == 1 && sizeof(TT) * C <= 8) {
          const ZVec<TT, 2 * C> both = *reinterpret_cast<const ZVec<TT, 2 * C>*>(tab + (r0 & ~1L) * C);
#pragma unroll
          for (int c = 0; c < C; ++c) { ce[2 * yz].v[c] = both.v[(r0 & 1) * C + c]; ce[2 * yz + 1].v[c] = both.v[(r1 & 1) * C + c]; }
        } else {
          ce[2 * yz] = *reinterpret_cast<const ZVec<TT, C>*>(tab + r0 * C);
          ce[2 * yz + 1] = *reinterpret_cast<const ZVec<TT, C>*>(tab + r1 * C);
        }
      }
    }
    float acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 0.f;
#pragma unroll
    for (int idx = 0; idx < 8; ++idx) {          // kernel_grid's order (gridencoder.cu:160-185): corner = x + 2 y + 4 z, w = product over d
      float w = 1.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) w *= (idx & (1 << k)) ? fr[k] : 1.f - fr[k];
#pragma unroll
      for (int c = 0; c < C; ++c) acc[c] += w * (float)ce[idx].v[c];
    }
    ZVec<TT, C> o;
#pragma unroll
    for (int c = 0; c < C; ++c) o.v[c] = (TT)acc[c];
    if ((a.s_b % C) == 0 && (a.s_l % C) == 0) *reinterpret_cast<ZVec<TT, C>*>(out) = o;
    else {
#pragma unroll
      for (int c = 0; c < C; ++c) out[c] = o.v[c];
    }
  }
}

// the fast forward for grid.hip's snerf_grid_encode_fwd (D = 3, C = 1 / 2 / 4 / 8, hash, linear, no align_corners, float / half, no dy_dx)
int g3_fwd_launch(const float* inputs, const void* table, const int* offsets, void* outputs, long B, int C, int L, float S, int H, int dtype,
                  long s_l, long s_b, hipStream_t s) {
  G3Args a{inputs, B, table, offsets, outputs, s_l, s_b, L, S, H};
  // measured at 14.7 M points, L = 10 (profiles/r5_y_grid_encoder_mapping_ab.txt): 8-byte entries (C = 4, half) 6.2 ms point-major vs 7.7
  // level-major on ray-ordered points, equal on random ones; 2- / 4-byte entries (C = 1) are faster level-major (4.2 vs 5.6 ms).
  // (2 / 4 / 8 consecutive points per thread with the cell's corners kept in registers: slower, profiles/r5_a_grid_encoder_leg_and_sweep.txt)
  const bool pm = (size_t)C * (dtype == SNERF_DT_F16 ? 2 : 4) >= 8 && s_l == C && s_b == (long)L * C;   // point-major needs the [B, L*C] layout to pay
  const dim3 grid0((unsigned)((B * L + 255) / 256)), grid((unsigned)((B + 255) / 256), L), blk(256);
#define G3F(TT, CC) do { if (pm) hipLaunchKernelGGL((g3_fwd_kernel<TT, CC, 0>), grid0, blk, 0, s, a); \
                         else hipLaunchKernelGGL((g3_fwd_kernel<TT, CC, 1>), grid, blk, 0, s, a); } while (0)
  if (dtype == SNERF_DT_F16) { if (C == 4) G3F(_Float16, 4); else if (C == 1) G3F(_Float16, 1); else if (C == 2) G3F(_Float16, 2); else if (C == 8) G3F(_Float16, 8); else return SNERF_ERR_ARG; }
  else if (dtype == SNERF_DT_F32) { if (C == 4) G3F(float, 4); else if (C == 1) G3F(float, 1); else if (C == 2) G3F(float, 2); else if (C == 8) G3F(float, 8); else return SNERF_ERR_ARG; }
  else return SNERF_ERR_ARG;
#undef G3F
  return snerf_check_launch();
}

// ---- table gradient of the stand-alone encoder (round 6): chunked, LDS-staged, bounded workspace ----------------------------------
// Same record / bin / LDS-fixed-point scheme as the zipnerf path above, reorganised so that the workspace does not grow with B and every
// record store is a full, aligned line:
//   * LEVELS OUTER, POINT CHUNKS INNER.  One level's records of one chunk of Pc points are counted, written and accumulated before the
//     next chunk reuses the record buffer; the partial sums of a level meet in ONE int64 image of that level (exact: integer sums,
//     one rounding at the end, bit-reproducible whatever the chunking).  Workspace = records of one (level, chunk) + the image of one
//     level + the offsets of one level, whatever B is (the round-5 form kept B x 8 x L records: 11-30 GB at 14.7 M points).
//   * The gradient is read LEVEL-MAJOR ([L, B, C]: the layout the reference's own backward builds, grid.py:74, and its FFI takes);
//     a point-major [B, L*C] gradient is transposed by g3_transpose_kernel in groups of as many levels as the workspace holds (the
//     same sweep finds max |grad| for the fixed-point scale), so that a level's pass reads B x C contiguous values instead of 8 bytes of
//     every 80-byte row.
//   * STAGED WRITER (g3_write_staged_kernel): a workgroup of WT threads x PPT consecutive points forms its <= WT x PPT x 8 records
//     (runs of consecutive points in one cell merged, as before), and moves them through an LDS stage SORTED BY BIN in windows of NSW
//     records: consecutive lanes store consecutive records of one (workgroup, bin) run.  Every run is padded to GR records (zero
//     records: row 0, value 0) so that runs start and end on 32-byte sectors in both planes -- no partial-sector stores, no
//     read-for-ownership (the direct writer wrote 18.2 GB and fetched 15.7 GB for 13.3 GB of records, profiles/r5_x_grid_encoder_pmc.txt).
//     A bin that takes more than a quarter of the stage from one workgroup (dense coarse levels) is written directly: its records
//     arrive in consecutive slots anyway.  The count pass (g3_count_kernel) leaves {offset, count} per (workgroup, bin), so the writer
//     starts from its histogram; each (run, corner)'s window is classified once (4 bits) and a window's walk only hashes its own records.
//   * Merged runs can reach G3 PPT x max |grad|: the scale keeps log2(8) = 3 more head bits (zip_bin_scale_kernel's `head`).
#define G3_DIRECT 0xffffffffu
#ifndef G3_WT_
#define G3_WT_ 1024      // (tools/probes/g3_variants.sh builds the alternatives)
#define G3_PPT_ 4
#define G3_WPE_ 4       // waves per SIMD the register budget is set for
#endif
template <int C, bool HREC> struct G3Cfg {
  static constexpr int VW = HREC ? (C == 1 ? 1 : C / 2) : C;            // 32-bit words of a staged value
  static constexpr int VWG = (C == 1 && !HREC) ? 2 : VW;                // 32-bit words of a record in the global value plane (C = 1: the row rides in it)
  static constexpr bool ROWPLANE = C != 1;
  static constexpr int GRLOG = ROWPLANE ? 4 : 3;                        // runs padded to 16 records (2-byte row plane: 32 B) / 8 (4- / 8-byte records)
  // ONE workgroup of 1024 threads x 4 points per CU: 144 KB of stage (a workgroup's <= 32 768 records pass in 3-4 windows at 12 bytes per
  // staged record), 128 registers per thread.  Measured alternatives (profiles/r6_a_g3_writer_variants.txt): 512 threads x 8 points with
  // two workgroups per CU (10 windows, 97 spilled registers); 768 / 512 threads with a thread's keys in registers too; 8 instead of 4
  // consecutive points per thread would merge 7 % more records on the bench's points.
  static constexpr int STAGE_BYTES = 147456;
  static constexpr int NS = (STAGE_BYTES / (4 + 4 * VW)) / 64 * 64;     // stage capacity (records): 18432 / 12288 / 7360 / 4096
  static constexpr int BIG = NS / 4, NSW = NS - BIG;                    // runs longer than BIG go direct; a window takes the bins that START inside NSW
  static constexpr int WT = G3_WT_, PPT = G3_PPT_;
  static constexpr int REC_BYTES = (ROWPLANE ? 2 : 0) + 4 * VWG;
  static constexpr size_t LDS = 3 * ZB_NBMAX * 4 + (size_t)NS * 4 * (1 + VW) + 256;    // (+ 64 + 64 + 64 bytes of scan / window scalars)
};
#define G3_WT G3_WT_
#define G3_PPT G3_PPT_

struct G3W {
  const float* inputs; long B; int in_vec;              // points [B, 3]; in_vec: 16-byte loads allowed
  const void* grad; long g_sb;                          // this level's gradient: value c of point p at grad[p * g_sb + c]
  const int* offsets; int level; float Sl; int H;
  unsigned wg0; long wgs_per_chunk;                     // writer: first workgroup of the chunk; count: workgroups per chunk
  uint2* wgo;                                           // [workgroups of all points][ZB_NBMAX] {offset of the run inside its bin, records}
  int* counts;                                          // count: [chunks][ZB_NBMAX] padded records per bin
  const unsigned* starts;                               // writer: [ZB_NBMAX] bin offsets of this (level, chunk)
  int bshift, K;
  unsigned short* rec_row; unsigned* rec_val; unsigned cap;
  const int* scale_exp;
};

__device__ __forceinline__ uint32_t g3_grid_index(uint32_t hs, uint32_t res, bool pow2, const uint32_t* pg) {   // = zip_grid_index: one form per level, the modulo as a mask where it is one
  (void)pow2;
  return zip_grid_index(hs, res, pg);
}

template <int PPT> struct G3Pts { uint32_t pg[PPT][3]; float fr[PPT][3]; unsigned inb, ends; };
// cells and fractions of PPT consecutive points; bit j of `ends`: point j is the last of its run of in-bounds points in one cell
template <int PPT>
__device__ __forceinline__ void g3_decode(const G3W& a, long p0, float scale, G3Pts<PPT>& t) {
  float xs[PPT * 3];
  if (PPT * 3 % 4 == 0 && a.in_vec && p0 + PPT <= a.B) {
    const float4* v = reinterpret_cast<const float4*>(a.inputs + p0 * 3);
#pragma unroll
    for (int q = 0; q < PPT * 3 / 4; ++q) { const float4 f = v[q]; xs[4 * q] = f.x; xs[4 * q + 1] = f.y; xs[4 * q + 2] = f.z; xs[4 * q + 3] = f.w; }
  } else {
#pragma unroll
    for (int q = 0; q < PPT * 3; ++q) xs[q] = p0 * 3 + q < a.B * 3 ? a.inputs[p0 * 3 + q] : -1.f;
  }
  t.inb = 0; t.ends = 0;
  int prev = -1;
  uint32_t pc[3] = {0u, 0u, 0u};
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const float* x = xs + 3 * j;
    if (p0 + j >= a.B || x[0] < 0.f || x[0] > 1.f || x[1] < 0.f || x[1] > 1.f || x[2] < 0.f || x[2] > 1.f) continue;   // (a NaN coordinate passes, as in kernel_grid)
#pragma unroll
    for (int k = 0; k < 3; ++k) zip_cell(x[k], scale, &t.pg[j][k], &t.fr[j][k]);
    if (prev >= 0 && (t.pg[j][0] != pc[0] || t.pg[j][1] != pc[1] || t.pg[j][2] != pc[2])) t.ends |= 1u << prev;
    pc[0] = t.pg[j][0]; pc[1] = t.pg[j][1]; pc[2] = t.pg[j][2];
    prev = j;
    t.inb |= 1u << j;
  }
  if (prev >= 0) t.ends |= 1u << prev;
}

// count pass of one level, all chunks: per-workgroup histogram -> padded reservation inside every touched bin of the workgroup's chunk
template <int WT, int PPT, int GRLOG>
__global__ __launch_bounds__(WT) void g3_count_kernel(G3W a) {
  __shared__ int cnt[ZB_NBMAX];
  const int tid = threadIdx.x;
  for (int k = tid; k < ZB_NBMAX; k += WT) cnt[k] = 0;
  __syncthreads();
  const unsigned wg = blockIdx.x;
  const long p0 = ((long)wg * WT + tid) * PPT;
  const uint32_t hs = a.offsets[a.level + 1] - a.offsets[a.level];
  const float scale = exp2f(a.level * a.Sl) * a.H - 1.0f;
  const uint32_t res = (uint32_t)ceilf(scale) + 1;
  const bool pow2 = (hs & (hs - 1u)) == 0u;
  const int rep = (int)(wg % (unsigned)a.K);
  if (p0 < a.B) {
    G3Pts<PPT> t;
    g3_decode<PPT>(a, p0, scale, t);
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      if (!((t.ends >> j) & 1u)) continue;
#pragma unroll
      for (int idx = 0; idx < 8; ++idx) {
        const uint32_t pl[3] = {t.pg[j][0] + (idx & 1), t.pg[j][1] + ((idx >> 1) & 1), t.pg[j][2] + (idx >> 2)};
        const uint32_t row = g3_grid_index(hs, res, pow2, pl);
        atomicAdd(cnt + zb_bin((uint32_t)row, a.bshift, a.K, rep), 1);
      }
    }
  }
  __syncthreads();
  int* counts = a.counts + (long)(wg / (unsigned long)a.wgs_per_chunk) * ZB_NBMAX;
  uint2* wrow = a.wgo + (long)wg * ZB_NBMAX;
  for (int k = tid; k < ZB_NBMAX; k += WT) {
    const int c = cnt[k];
    unsigned o = 0;
    if (c != 0) o = (unsigned)atomicAdd(counts + k, (c + (1 << GRLOG) - 1) & ~((1 << GRLOG) - 1));
    wrow[k] = uint2{o, (unsigned)c};
  }
}

// starts[c][bin] = exclusive scan over the bins of counts[c][bin], one block per chunk (32-bit: a chunk holds < 2^32 records)
__global__ __launch_bounds__(1024) void g3_scan_chunks_kernel(const int* __restrict__ counts, unsigned* __restrict__ starts) {
  __shared__ unsigned wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const unsigned mine = (unsigned)counts[(long)blockIdx.x * ZB_NBMAX + tid];
  unsigned incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const unsigned u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
  if (lane == 63) wsum[wv] = incl;
  __syncthreads();
  unsigned pre = incl - mine;
  for (int q = 0; q < wv; ++q) pre += wsum[q];
  starts[(long)blockIdx.x * ZB_NBMAX + tid] = pre;
}

template <int C, bool HREC>
__device__ __forceinline__ void g3_pack(const float* v, float hmul, unsigned* w) {
  if constexpr (HREC && C == 1) w[0] = zb_half_bits(v[0] * hmul);
  else if constexpr (HREC) {
    const zb_h4 v4 = {(_Float16)(v[0] * hmul), (_Float16)(v[1] * hmul), (_Float16)(v[2] * hmul), (_Float16)(v[3] * hmul)};
    const uint2 hv = __builtin_bit_cast(uint2, v4);
    w[0] = hv.x; w[1] = hv.y;
  } else {
#pragma unroll
    for (int c = 0; c < C; ++c) w[c] = __float_as_uint(v[c]);
  }
}
template <int C, bool HREC>
__device__ __forceinline__ void g3_store_one(unsigned short* rec_row, unsigned* rec_val, unsigned r, unsigned lrow, const unsigned* w) {
  constexpr int VW = G3Cfg<C, HREC>::VW;
  if constexpr (HREC && C == 1) rec_val[r] = lrow | (w[0] << 16);
  else if constexpr (C == 1) *reinterpret_cast<uint2*>(rec_val + 2l * r) = uint2{lrow, w[0]};
  else {
    rec_row[r] = (unsigned short)lrow;
    unsigned* d = rec_val + (long)r * VW;
    if constexpr (VW == 2) *reinterpret_cast<uint2*>(d) = uint2{w[0], w[1]};
    else {
#pragma unroll
      for (int k = 0; k < VW; k += 4) *reinterpret_cast<uint4*>(d + k) = uint4{w[k], w[k + 1], w[k + 2], w[k + 3]};
    }
  }
}
// records r (even) and r + 1 of one run
template <int C, bool HREC>
__device__ __forceinline__ void g3_store_pair(unsigned short* rec_row, unsigned* rec_val, unsigned r, unsigned l0, unsigned l1, const unsigned* w0, const unsigned* w1) {
  constexpr int VW = G3Cfg<C, HREC>::VW;
  if constexpr (HREC && C == 1) *reinterpret_cast<uint2*>(rec_val + r) = uint2{l0 | (w0[0] << 16), l1 | (w1[0] << 16)};
  else if constexpr (C == 1) *reinterpret_cast<uint4*>(rec_val + 2l * r) = uint4{l0, w0[0], l1, w1[0]};
  else {
    *reinterpret_cast<unsigned*>(rec_row + r) = l0 | (l1 << 16);
    unsigned* d = rec_val + (long)r * VW;
    if constexpr (VW == 2) *reinterpret_cast<uint4*>(d) = uint4{w0[0], w0[1], w1[0], w1[1]};
    else {
#pragma unroll
      for (int k = 0; k < VW; k += 4) *reinterpret_cast<uint4*>(d + k) = uint4{w0[k], w0[k + 1], w0[k + 2], w0[k + 3]};
#pragma unroll
      for (int k = 0; k < VW; k += 4) *reinterpret_cast<uint4*>(d + VW + k) = uint4{w1[k], w1[k + 1], w1[k + 2], w1[k + 3]};
    }
  }
}

#ifdef G3_PROF          // probe builds only (tools/probes/g3_prof.sh): cycles per phase of the staged writer, summed over workgroups by thread 0
__device__ unsigned long long g3_prof_cycles[16];
extern "C" int snerf_g3_prof_read(unsigned long long* host16, int reset) {
  if (hipMemcpyFromSymbol(host16, HIP_SYMBOL(g3_prof_cycles), sizeof(unsigned long long) * 16) != hipSuccess) return SNERF_ERR_LAUNCH;
  if (reset) { unsigned long long z[16] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g3_prof_cycles), z, sizeof(z)); }
  return SNERF_OK;
}
#define G3_TICK(i) do { __builtin_amdgcn_s_waitcnt(0); if (threadIdx.x == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); atomicAdd(&g3_prof_cycles[i], now_ - t_prev_); t_prev_ = now_; } } while (0)
#define G3_TICK0() unsigned long long t_prev_ = __builtin_amdgcn_s_memtime()
#else
#define G3_TICK(i) do { } while (0)
#define G3_TICK0() do { } while (0)
#endif
template <typename GT, int C, bool HREC>
__global__ __launch_bounds__(G3_WT, G3_WPE_) void g3_write_staged_kernel(G3W a) {
  using Cfg = G3Cfg<C, HREC>;
  constexpr int WT = Cfg::WT, PPT = Cfg::PPT, VW = Cfg::VW, NS = Cfg::NS, NSW = Cfg::NSW, BIG = Cfg::BIG, GR = 1 << Cfg::GRLOG;
  constexpr int BPT = (ZB_NBMAX + WT - 1) / WT, NWV = WT / 64;       // thread t owns bins t, t + WT, ... (the order of the bins in the stage is free)
  extern __shared__ __attribute__((aligned(16))) unsigned g3_lds[];
  unsigned* sval = g3_lds;                                   // [NS * VW] staged values (16-byte aligned)
  unsigned* skey = sval + NS * VW;                           // [NS] staged record: row in bin | bin << 14
  int* cnt = (int*)(skey + NS);                              // [ZB_NBMAX] placement counters
  unsigned* off = (unsigned*)cnt + ZB_NBMAX;                 // [ZB_NBMAX] offset of the bin's padded run in the sorted order of the staged bins; G3_DIRECT: written directly
  unsigned* base = off + ZB_NBMAX;                           // [ZB_NBMAX] record offset of this workgroup's run inside the chunk's record buffer
  unsigned* wtot = base + ZB_NBMAX;                          // [NWV]
  unsigned* wend = wtot + 16;                                // [16] end of the staged records of window w (stage positions)
  unsigned* wbeg = wend + 16;                                // [16] their begin: the bin that straddles the window boundary belongs to the window before
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const unsigned wg = a.wg0 + blockIdx.x;
  const long p0 = ((long)wg * WT + tid) * PPT;
  const uint32_t hs = a.offsets[a.level + 1] - a.offsets[a.level];
  const float scale = exp2f(a.level * a.Sl) * a.H - 1.0f;
  const uint32_t res = (uint32_t)ceilf(scale) + 1;
  const bool pow2 = (hs & (hs - 1u)) == 0u;
  const int rep = (int)(wg % (unsigned)a.K);
  const unsigned rmask = (1u << a.bshift) - 1u;
  float hmul = 1.f;
  if constexpr (HREC) hmul = exp2f((float)(a.scale_exp[0] - ZB_HALF_SHIFT));
  G3_TICK0();
  // ---- the thread's points and their gradients
  G3Pts<PPT> t;
  t.inb = 0; t.ends = 0;
  ZVec<GT, C> g[PPT];
  if (p0 < a.B) {
    g3_decode<PPT>(a, p0, scale, t);
#pragma unroll
    for (int j = 0; j < PPT; ++j)
      if ((t.inb >> j) & 1u) g[j] = *reinterpret_cast<const ZVec<GT, C>*>((const GT*)a.grad + (p0 + j) * a.g_sb);
  }
  // ---- CACHE: every record's VALUE once -- (run, corner) -> packed words in registers (64 at C = 4); the fractions and gradients die here.
  // (A window that re-walks the points to form its values -- the form kept below for the wide fp32 records -- costs 8.7 instead of 8.3 ms
  // per backward; keeping the 32 KEYS in registers too needs 96 + registers per thread: at 1024 threads they spill to scratch (19.7 ms
  // per backward instead of 15.2), at 512 threads x 256 registers the phases of the single resident workgroup stop overlapping (16.7):
  // profiles/r6_a_g3_writer_variants.txt.)
  constexpr bool CACHE_ = G3Cfg<C, HREC>::VW <= 2;
  unsigned rv_[CACHE_ ? PPT * 8 : 1][G3Cfg<C, HREC>::VW];
  if constexpr (CACHE_) {
#pragma unroll
    for (int idx = 0; idx < 8; ++idx) {
      float acc[C];
#pragma unroll
      for (int c = 0; c < C; ++c) acc[c] = 0.f;
#pragma unroll
      for (int j = 0; j < PPT; ++j) {
#pragma unroll
        for (int k = 0; k < G3Cfg<C, HREC>::VW; ++k) rv_[j * 8 + idx][k] = 0u;
        if (!((t.inb >> j) & 1u)) continue;
        float wgt = 1.f;                             // kernel_grid_backward's weight (gridencoder.cu:291-300): product over d in order
#pragma unroll
        for (int k = 0; k < 3; ++k) wgt *= (idx & (1 << k)) ? t.fr[j][k] : 1.f - t.fr[j][k];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] += wgt * (float)g[j].v[c];
        if (!((t.ends >> j) & 1u)) continue;
        g3_pack<C, HREC>(acc, hmul, rv_[j * 8 + idx]);
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = 0.f;
      }
    }
  }
  G3_TICK(0);
  // ---- the workgroup's histogram (count pass) -> sorted offsets of the staged bins, windows
  if (tid < 16) { wend[tid] = 0; wbeg[tid] = 0xffffffffu; }
  const uint2* wrow = a.wgo + (long)wg * ZB_NBMAX;
  unsigned cb[BPT], pb[BPT], ob[BPT];
  unsigned mine = 0;
#pragma unroll
  for (int q = 0; q < BPT; ++q) {
    const int bin = tid + q * WT;
    cb[q] = 0; pb[q] = 0;
    if (bin >= ZB_NBMAX) continue;
    const uint2 e = wrow[bin];
    cb[q] = e.y;
    pb[q] = (e.y + GR - 1) & ~(unsigned)(GR - 1);
    base[bin] = a.starts[bin] + e.x;
    cnt[bin] = 0;
    if (pb[q] <= (unsigned)BIG) mine += pb[q];
  }
  unsigned incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const unsigned u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
  if (lane == 63) wtot[wv] = incl;
  __syncthreads();
  unsigned pre = incl - mine, total = 0;
#pragma unroll
  for (int q = 0; q < NWV; ++q) { if (q < wv) pre += wtot[q]; total += wtot[q]; }
  int nwin = (int)((total + NSW - 1) / NSW);
  const bool alldirect = nwin > 15;           // (cannot happen with the padded capacity of the shipped configurations; kept correct)
  if (alldirect) nwin = 0;
#pragma unroll
  for (int q = 0; q < BPT; ++q) {
    const int bin = tid + q * WT;
    ob[q] = G3_DIRECT;
    if (bin >= ZB_NBMAX) continue;
    if (alldirect || pb[q] > (unsigned)BIG) { off[bin] = G3_DIRECT; continue; }
    ob[q] = pre; off[bin] = pre;
    if (pb[q] != 0) { atomicMax(wend + pre / NSW, pre % NSW + pb[q]); atomicMin(wbeg + pre / NSW, pre % NSW); }
    pre += pb[q];
  }
  __syncthreads();
  G3_TICK(1);
  constexpr bool CACHE = CACHE_;
  auto& rv = rv_;
  // ---- window of every (run, corner): 4 bits (15 = direct)
  unsigned wr[PPT];
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    wr[j] = 0;
    if (!((t.ends >> j) & 1u)) continue;
#pragma unroll
    for (int idx = 0; idx < 8; ++idx) {
      const uint32_t pl[3] = {t.pg[j][0] + (idx & 1), t.pg[j][1] + ((idx >> 1) & 1), t.pg[j][2] + (idx >> 2)};
      const uint32_t row = g3_grid_index(hs, res, pow2, pl);
      const unsigned o = off[zb_bin((uint32_t)row, a.bshift, a.K, rep)];
      wr[j] |= (o == G3_DIRECT ? 15u : o / NSW) << (4 * idx);
    }
  }
  G3_TICK(2);
  // ---- windows: place, stream out
  for (int w = 0; w < (nwin > 0 ? nwin : 1); ++w) {
    // padding records of the bins of this window (row 0, value 0)
#pragma unroll
    for (int q = 0; q < BPT; ++q) {
      if (ob[q] == G3_DIRECT || pb[q] == cb[q] || (int)(ob[q] / NSW) != w) continue;
      const unsigned s0 = ob[q] - w * NSW;
      for (unsigned s = cb[q]; s < pb[q]; ++s) {
        skey[s0 + s] = (unsigned)(tid + q * WT) << 14;
#pragma unroll
        for (int k = 0; k < VW; ++k) sval[(s0 + s) * VW + k] = 0u;
      }
    }
    G3_TICK(5);
    if constexpr (CACHE) {
#pragma unroll
      for (int j = 0; j < PPT; ++j) {
        if (!((t.ends >> j) & 1u)) continue;
#pragma unroll
        for (int k = 0; k < 3; ++k) asm volatile("" : "+v"(t.pg[j][k]));      // (keeps the 32 hashes out of the loop preheader)
#pragma unroll
        for (int idx = 0; idx < 8; ++idx) {
          const unsigned wid = (wr[j] >> (4 * idx)) & 15u;
          if ((int)wid == w || (wid == 15u && w == 0)) {
            const uint32_t pl[3] = {t.pg[j][0] + (idx & 1), t.pg[j][1] + ((idx >> 1) & 1), t.pg[j][2] + (idx >> 2)};
            const uint32_t row = g3_grid_index(hs, res, pow2, pl);
            const int bin = zb_bin((uint32_t)row, a.bshift, a.K, rep);
            const unsigned slot = (unsigned)atomicAdd(cnt + bin, 1);
            if (wid == 15u) {
              const unsigned r = base[bin] + slot;
              if (r < a.cap) g3_store_one<C, HREC>(a.rec_row, a.rec_val, r, row & rmask, rv[j * 8 + idx]);
            } else {
              const unsigned s_ = off[bin] - w * NSW + slot;
              skey[s_] = (row & rmask) | ((unsigned)bin << 14);
#pragma unroll
              for (int k = 0; k < VW; ++k) sval[s_ * VW + k] = rv[j * 8 + idx][k];
            }
          }
        }
      }
    } else {
    // (the cells and fractions are loop-invariant: without this the compiler hoists all 64 hashes and weights out of the window loop
    // and spills several hundred registers)
#pragma unroll
    for (int j = 0; j < PPT; ++j)
#pragma unroll
      for (int k = 0; k < 3; ++k) { asm volatile("" : "+v"(t.pg[j][k])); asm volatile("" : "+v"(t.fr[j][k])); }
    // corner by corner (one accumulator of C values live instead of eight): the run sums of corner idx, each emitted at its run's end
#pragma unroll
    for (int idx = 0; idx < 8; ++idx) {
      float acc[C];
#pragma unroll
      for (int c = 0; c < C; ++c) acc[c] = 0.f;
#pragma unroll
      for (int j = 0; j < PPT; ++j) {
        if (!((t.inb >> j) & 1u)) continue;
        float wgt = 1.f;                             // kernel_grid_backward's weight (gridencoder.cu:291-300): product over d in order
#pragma unroll
        for (int k = 0; k < 3; ++k) wgt *= (idx & (1 << k)) ? t.fr[j][k] : 1.f - t.fr[j][k];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] += wgt * (float)g[j].v[c];
        if (!((t.ends >> j) & 1u)) continue;
        const unsigned wid = (wr[j] >> (4 * idx)) & 15u;
        if ((int)wid == w || (wid == 15u && w == 0)) {
          const uint32_t pl[3] = {t.pg[j][0] + (idx & 1), t.pg[j][1] + ((idx >> 1) & 1), t.pg[j][2] + (idx >> 2)};
          const uint32_t row = g3_grid_index(hs, res, pow2, pl);
          const int bin = zb_bin((uint32_t)row, a.bshift, a.K, rep);
          const unsigned slot = (unsigned)atomicAdd(cnt + bin, 1);
          unsigned wv_[VW];
          g3_pack<C, HREC>(acc, hmul, wv_);
          if (wid == 15u) {
            const unsigned r = base[bin] + slot;
            if (r < a.cap) g3_store_one<C, HREC>(a.rec_row, a.rec_val, r, row & rmask, wv_);
          } else {
            const unsigned s = off[bin] - w * NSW + slot;
            skey[s] = (row & rmask) | ((unsigned)bin << 14);
#pragma unroll
            for (int k = 0; k < VW; ++k) sval[s * VW + k] = wv_[k];
          }
        }
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = 0.f;
      }
    }
    }
    G3_TICK(6);
    __syncthreads();
    G3_TICK(3);
    const unsigned n = nwin > 0 ? wend[w] : 0u, n0 = nwin > 0 ? wbeg[w] : 0u;   // padded records staged in this window: [n0, n); pairs never straddle a run (GR is even)
    for (unsigned s = n0 + 2u * tid; s < n; s += 2u * WT) {
      const unsigned k0 = skey[s], k1 = skey[s + 1];
      const int bin = (int)(k0 >> 14);
      const unsigned r = base[bin] + (s - (off[bin] - w * NSW));
      if (r + 1 < a.cap) g3_store_pair<C, HREC>(a.rec_row, a.rec_val, r, k0 & 0x3fffu, k1 & 0x3fffu, sval + s * VW, sval + (s + 1) * VW);
    }
    __syncthreads();
    G3_TICK(4);
  }
  // ---- padding of the directly written runs
#pragma unroll
  for (int q = 0; q < BPT; ++q) {
    if (ob[q] != G3_DIRECT || tid + q * WT >= ZB_NBMAX) continue;
    const unsigned zero[VW] = {};
    for (unsigned s = cb[q]; s < pb[q]; ++s) {
      const unsigned r = base[tid + q * WT] + s;
      if (r < a.cap) g3_store_one<C, HREC>(a.rec_row, a.rec_val, r, 0u, zero);
    }
  }
}

// one chunk's records of one bin -> the bin's rows: LDS fixed-point sums (+ the level image of the earlier chunks), written as the
// gradient after the level's last chunk.  K > 1 (replicated row ranges of the hot dense levels): every chunk adds into the zeroed image
// with integer atomics and zip_bin_finish_kernel converts it after the last one.
struct G3A {
  const int* counts; const unsigned* starts; const unsigned short* rec_row; const float* rec_val; unsigned cap;
  int bshift, K; long rows_l;
  long long* g64;                            // image of this level's rows (null: single chunk, K = 1)
  void* out;                                 // this level's rows of grad_embeddings
  const int* scale_exp; int first, last;
};
template <int C, bool HREC, typename GT>
__global__ __launch_bounds__(1024) void g3_accumulate_kernel(G3A a) {
  extern __shared__ long long zb_acc[];
  const int bin = blockIdx.x;
  const long row0 = (long)(bin / a.K) << a.bshift;
  if (row0 >= a.rows_l) return;
  const unsigned s0 = a.starts[bin];
  int n = a.counts[bin];
  if (s0 >= a.cap) n = 0; else if ((unsigned)n > a.cap - s0) n = (int)(a.cap - s0);
  const int se = a.scale_exp[0];
  const float fix = exp2f((float)se), lim = exp2f((float)(ZB_HEAD + 1 - se));
  const double unfix = exp2((double)-se);
  const int cells = (int)min((long)(1 << a.bshift), a.rows_l - row0) * C;
  for (int k = threadIdx.x; k < cells; k += 1024) zb_acc[k] = 0;
  __syncthreads();
  zb_acc_records<C, HREC, true>(zb_acc, a.rec_row, a.rec_val, (long)s0, n, fix, lim);
  __syncthreads();
  if (a.K == 1) {
    GT* dst = (GT*)a.out + row0 * C;
    long long* img = a.g64 != nullptr ? a.g64 + row0 * C : nullptr;
    for (int k = threadIdx.x; k < cells; k += 1024) {
      long long v = zb_acc[k];
      const int kk = C == 4 ? (k ^ ((k >> 5) & 3)) : k;     // (undo the slot swizzle)
      if (!a.first) v += img[kk];
      if (a.last) { if (v != 0) dst[kk] = (GT)((float)dst[kk] + (float)((double)v * unfix)); }
      else img[kk] = v;
    }
  } else {
    long long* img = a.g64 + row0 * C;
    for (int k = threadIdx.x; k < cells; k += 1024) {
      const long long v = zb_acc[k];
      const int kk = C == 4 ? (k ^ ((k >> 5) & 3)) : k;
      if (v != 0) atomicAdd((unsigned long long*)(img + kk), (unsigned long long)v);
    }
  }
}

// [B, L*C] -> [nl, B, C] for levels [l0, l0 + nl) through an LDS tile of TP points (rows read whole and coalesced); mx != null: max |grad|
// over ALL columns on the way (bits of the non-negative maximum; NaN counts as Inf)
template <typename GT>
__global__ __launch_bounds__(256) void g3_transpose_kernel(const GT* __restrict__ grad, long B, int LC, int C, int l0, int nl, GT* __restrict__ tg,
                                                           unsigned* __restrict__ mx, int TP) {
  extern __shared__ __attribute__((aligned(16))) unsigned char g3_tile[];
  const int tid = threadIdx.x;
  const long p0 = (long)blockIdx.x * TP;
  const int np = (int)min((long)TP, B - p0);
  const long nbytes = (long)np * LC * (long)sizeof(GT);
  const unsigned char* src = (const unsigned char*)grad + p0 * LC * (long)sizeof(GT);
  // max |x| on the BIT patterns (sign cleared, unsigned compare: non-negative floats order like their bits, a NaN sorts above Inf): two
  // integer ops per 32-bit word (the float form -- convert, fabs, NaN test per element -- made this sweep 2.6 ms instead of 0.3)
  typedef unsigned short g3_us2 __attribute__((ext_vector_type(2)));
  unsigned mb = 0u;                                  // fp32: bits of the maximum; fp16: two 16-bit maxima
  auto upd = [&](unsigned wbits) __attribute__((always_inline)) {
    if constexpr (sizeof(GT) == 2) mb = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(g3_us2, mb), __builtin_bit_cast(g3_us2, wbits & 0x7fff7fffu)));
    else { const unsigned v = wbits & 0x7fffffffu; mb = v > mb ? v : mb; }
  };
  const long nv = (((uintptr_t)src) & 15) == 0 ? nbytes / 16 : 0;
  for (long v = tid; v < nv; v += 256) {
    const uint4 q = reinterpret_cast<const uint4*>(src)[v];
    reinterpret_cast<uint4*>(g3_tile)[v] = q;
    if (mx != nullptr) { upd(q.x); upd(q.y); upd(q.z); upd(q.w); }
  }
  for (long e = nv * (16 / (long)sizeof(GT)) + tid; e < nbytes / (long)sizeof(GT); e += 256) {
    const GT gv = ((const GT*)src)[e];
    ((GT*)g3_tile)[e] = gv;
    if constexpr (sizeof(GT) == 2) upd((unsigned)__builtin_bit_cast(unsigned short, gv)); else upd(__builtin_bit_cast(unsigned, gv));
  }
  if (mx != nullptr) {
    float m;
    if constexpr (sizeof(GT) == 2) {
      const unsigned h = (mb & 0xffffu) > (mb >> 16) ? (mb & 0xffffu) : (mb >> 16);
      m = h >= 0x7c00u ? __builtin_inff() : (float)__builtin_bit_cast(_Float16, (unsigned short)h);
    } else m = mb >= 0x7f800000u ? __builtin_inff() : __uint_as_float(mb);
    m = fmaxf(m, __shfl_xor(m, 32, 64));
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    // (one same-address atomic per wave of 57 000 workgroups serialises in L2: 2.3 ms; the maximum only grows, so a wave whose value is
    // not above what it can already see skips it)
    if ((tid & 63) == 0 && m > 0.f && __float_as_uint(m) > __atomic_load_n(mx, __ATOMIC_RELAXED)) atomicMax(mx, __float_as_uint(m));
  }
  __syncthreads();
  const GT* tile = (const GT*)g3_tile;
  for (int i = tid; i < nl * np; i += 256) {
    const int li = i / np, pt = i - li * np;
    const GT* s = tile + (long)pt * LC + (l0 + li) * C;
    GT* d = tg + ((long)li * B + p0 + pt) * C;
    if ((C * (int)sizeof(GT)) % 4 == 0) {
      for (int k = 0; k < C * (int)sizeof(GT) / 4; ++k) ((unsigned*)d)[k] = ((const unsigned*)s)[k];
    } else {
      for (int c = 0; c < C; ++c) d[c] = s[c];
    }
  }
}

// exclusive scan of the [L x ZB_NBMAX] bin counts into record offsets, on the device (the zipnerf trainer does it with torch.cumsum)
__global__ __launch_bounds__(1024) void g3_scan_kernel(const int* __restrict__ counts, long* __restrict__ starts, int L) {
  __shared__ long wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  long mine = 0;
  for (int k = 0; k < L; ++k) mine += counts[tid * L + k];            // thread t owns elements [t L, t L + L) of the flat array
  long incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const long u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
  if (lane == 63) wsum[wv] = incl;
  __syncthreads();
  long pre = incl - mine;
  for (int q = 0; q < wv; ++q) pre += wsum[q];
  for (int k = 0; k < L; ++k) { starts[tid * L + k] = pre; pre += counts[tid * L + k]; }
}

// starts = exclusive scan of the [L x ZB_NBMAX] bin counts (int32 -> int64 record offsets): one launch instead of torch's cast + cumsum + subtract
extern "C" int snerf_zip_bin_scan(const int* counts, long* starts, int L, void* stream) {
  if (counts == nullptr || starts == nullptr || L <= 0 || L > 16) return SNERF_ERR_ARG;
  hipLaunchKernelGGL(g3_scan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, counts, starts, L);
  return snerf_check_launch();
}

// ---- host plan: chunk size, levels per transposed group, workspace layout ----
struct G3Plan {
  int wt, ppt, grlog, rec_bytes, vwg, rowplane, bshift;
  long wgpts, nwg_total, pc, wgs_per_chunk;          // points per workgroup / chunk
  int nck, lt, groups;                               // chunks per level; levels per transposed group (0: the gradient is level-major already)
  unsigned cap;                                      // records a chunk's buffer holds (padded upper bound)
  int ks[16]; long rows[16]; long g64_rows;          // replicas per row range (at the chunk's record count); rows of the level image (0: none)
  size_t o_scale, o_counts, o_starts, o_wgo, o_g64, o_tg, o_row, o_val, total, zero_bytes;
};
static inline size_t g3_al(size_t v) { return (v + 255) & ~(size_t)255; }
// padded upper bound of the records of pc points of one level: 8 per point + (GR - 1) per non-empty (workgroup, bin) pair
static long g3_cap(long pc, long wgpts, int grlog, int maxbins) {
  const long rec = pc * 8, pairs = ((pc + wgpts - 1) / wgpts) * (long)maxbins;
  return rec + ((1L << grlog) - 1) * (rec < pairs ? rec : pairs) + 64;
}
static int g3_make_plan(long B, int C, int L, const int* oh, bool hrec, bool point_major, int gsz, long ws_bytes, G3Plan* pl, long* need_min) {
  G3Plan p{};
  const int vw = hrec ? (C == 1 ? 1 : C / 2) : C;
  p.vwg = (C == 1 && !hrec) ? 2 : vw;
  p.rowplane = C != 1; p.grlog = p.rowplane ? 4 : 3;
  p.wt = G3_WT; p.ppt = G3_PPT;
  p.rec_bytes = (p.rowplane ? 2 : 0) + 4 * p.vwg;
  p.bshift = C == 8 ? 11 : (C == 4 ? 12 : (C == 2 ? 13 : 14));          // 128 KB of 64-bit cells per bin
  p.wgpts = (long)p.wt * p.ppt;
  p.nwg_total = (B + p.wgpts - 1) / p.wgpts;
  long rowbins[16], max_rows = 0;
  for (int l = 0; l < L; ++l) {
    p.rows[l] = (long)oh[l + 1] - oh[l];
    if (p.rows[l] <= 0) return SNERF_ERR_ARG;
    rowbins[l] = (p.rows[l] + (1L << p.bshift) - 1) >> p.bshift;
    if (rowbins[l] > ZB_NBMAX) return SNERF_ERR_ARG;
    if (p.rows[l] > max_rows) max_rows = p.rows[l];
  }
  const long target = 2000000;                                           // records per bin the replicas aim at
  auto layout = [&](long nwg_chunk, int lt, G3Plan& q) -> bool {        // everything for chunks of nwg_chunk workgroups
    q.wgs_per_chunk = nwg_chunk; q.pc = nwg_chunk * q.wgpts;
    q.nck = (int)((q.nwg_total + nwg_chunk - 1) / nwg_chunk);
    q.lt = lt; q.groups = lt > 0 ? (L + lt - 1) / lt : 1;
    int maxbins = 1; q.g64_rows = 0;
    const long pts = q.pc < B ? q.pc : B;
    for (int l = 0; l < L; ++l) {
      // replicas per row range: a level's accumulate launch is on its own here (levels outer), so it needs >= ~512 bins to fill the
      // part -- level 0's 4913 rows are 2 row ranges -- as long as a bin keeps >= 8192 records (below that zeroing and folding its
      // 128 KB image costs more than its records); never more than `target` records per bin
      long k = (pts * 8 + rowbins[l] * target - 1) / (rowbins[l] * target);
      const long kfill = (512 + rowbins[l] - 1) / rowbins[l], kmax = pts * 8 / (rowbins[l] * 8192);
      if (kfill > k) k = kfill < kmax ? kfill : kmax;
      k = k < 1 ? 1 : (k > ZB_NBMAX / rowbins[l] ? ZB_NBMAX / rowbins[l] : k);
      q.ks[l] = (int)k;
      if (rowbins[l] * k > maxbins) maxbins = (int)(rowbins[l] * k);
      if ((k > 1 || q.nck > 1) && q.rows[l] > q.g64_rows) q.g64_rows = q.rows[l];
    }
    const long cap = g3_cap(pts, q.wgpts, q.grlog, maxbins);
    if (cap >= (1L << 32) - 64) return false;
    q.cap = (unsigned)cap;
    size_t o = 0;
    q.o_scale = o; o = g3_al(o + 8);
    q.o_counts = o; o = g3_al(o + (size_t)L * q.nck * ZB_NBMAX * 4);    // scale .. counts: zeroed at the start of a call
    q.zero_bytes = o;
    q.o_starts = o; o = g3_al(o + (size_t)q.nck * ZB_NBMAX * 4);
    q.o_wgo = o; o = g3_al(o + (size_t)q.nwg_total * ZB_NBMAX * 8);
    q.o_g64 = o; o = g3_al(o + (size_t)q.g64_rows * C * 8);
    q.o_tg = o; o = g3_al(o + (lt > 0 ? (size_t)lt * B * C * gsz : 0));
    q.o_row = o; o = g3_al(o + (q.rowplane ? (size_t)cap * 2 : 0));
    q.o_val = o; o = g3_al(o + (size_t)cap * 4 * q.vwg);
    q.total = o;
    return true;
  };
  // candidates: levels per transposed group x chunk size; the cheapest estimate that fits (transposition sweeps: the whole gradient read
  // once per group; chunks: launch gaps + the level image read and written once per extra chunk)
  double best = 1e30; bool found = false; G3Plan bestp{};
  long minneed = -1;
  for (int lt = point_major ? L : 0; lt >= (point_major ? 1 : 0); --lt) {
    if (point_major && lt > 1 && (L + lt - 1) / lt == (L + lt - 2) / (lt - 1)) continue;      // (lt - 1 makes the same number of sweeps with a smaller copy)
    G3Plan q = p;
    const long minwg = 1;                                              // (a workspace too small for more still runs, one workgroup per chunk)
    if (layout(minwg, lt, q) && (minneed < 0 || (long)q.total < minneed)) minneed = (long)q.total;
    if (layout(p.nwg_total, lt, q) && (minneed < 0 || (long)q.total < minneed)) minneed = (long)q.total;
    // the largest chunk that fits: one chunk for the whole level needs no level image (checked first); below that the size is monotone
    long lo = minwg, hi = p.nwg_total - 1, fit = -1;
    if (layout(p.nwg_total, lt, q) && (long)q.total <= ws_bytes) fit = p.nwg_total;
    else {
      while (lo <= hi) {
        const long mid = (lo + hi) / 2;
        if (layout(mid, lt, q) && (long)q.total <= ws_bytes) { fit = mid; lo = mid + 1; } else hi = mid - 1;
      }
    }
    if (fit < 0) { if (lt == 0) break; continue; }
    // balance the chunks: the same number of chunks with equal sizes
    const long nck = (p.nwg_total + fit - 1) / fit, even = (p.nwg_total + nck - 1) / nck;
    layout(even <= fit ? even : fit, lt, q);
    const double gbytes = (double)B * L * C * gsz;
    const double cost = (lt > 0 ? q.groups * (gbytes + (double)lt * B * C * gsz) / 4e12 : 0.0) +
                        (double)L * q.nck * (20e-6 + (q.nck > 1 ? 2.0 * max_rows * C * 8 / 4e12 : 0.0));
    if (cost < best) { best = cost; bestp = q; found = true; }
    if (lt == 0) break;
  }
  if (need_min != nullptr) *need_min = minneed;
  if (!found) return SNERF_ERR_ARG;
  *pl = bestp;
  return SNERF_OK;
}

#define G3_WS_DEFAULT 1000000000L
// recommended workspace: everything in one chunk if that takes less than 1 GB, else 1 GB (or the smallest feasible layout beyond it)
extern "C" long snerf_grid_encode_bwd_binned_ws_bytes(long B, int C, int L, const int* offsets_host, int half_records) {
  if (B <= 0) return 0;
  if (L <= 0 || L > 16 || (C != 1 && C != 2 && C != 4 && C != 8) || offsets_host == nullptr) return -1;
  const bool hrec = half_records != 0 && (C == 1 || C == 4);
  G3Plan pl; long need_min = -1;
  // (sized for a point-major fp32 gradient: the worst case of the layouts the call accepts)
  const int rc = g3_make_plan(B, C, L, offsets_host, hrec, true, 4, G3_WS_DEFAULT, &pl, &need_min);
  if (rc == SNERF_OK) return (long)pl.total;
  if (need_min < 0) return -1;
  return need_min;
}

// the plan a call with this workspace would run: out[0] = chunks per level, out[1] = points per chunk, out[2] = levels per transposed
// group (0: none), out[3] = kernel launches, out[4] = record capacity of a chunk, out[5] = bytes used
extern "C" int snerf_grid_encode_bwd_binned_plan(long B, int C, int L, const int* offsets_host, int half_records, int grad_dtype, int level_major,
                                                 long ws_bytes, long* out) {
  if (B <= 0 || L <= 0 || L > 16 || (C != 1 && C != 2 && C != 4 && C != 8) || offsets_host == nullptr || out == nullptr) return SNERF_ERR_ARG;
  G3Plan pl;
  const int rc = g3_make_plan(B, C, L, offsets_host, half_records != 0 && (C == 1 || C == 4), level_major == 0, grad_dtype == SNERF_DT_F16 ? 2 : 4, ws_bytes, &pl, nullptr);
  if (rc != SNERF_OK) return rc;
  int kx = 0;
  for (int l = 0; l < L; ++l) kx += pl.ks[l] > 1 ? 1 : 0;
  out[0] = pl.nck; out[1] = pl.pc; out[2] = pl.lt; out[3] = (pl.lt > 0 ? pl.groups : 1) + 2 + (long)L * (2 + 2 * pl.nck) + kx + 1;
  out[4] = pl.cap; out[5] = (long)pl.total;
  return SNERF_OK;
}

extern "C" int snerf_grid_encode_bwd_binned(const void* grad, const float* inputs, const int* offsets, const int* offsets_host, void* grad_embeddings,
                                            long B, int C, int L, float S, int H, int grad_dtype, int out_dtype, long grad_stride_l,
                                            long grad_stride_b, int half_records, void* ws, long ws_bytes, void* stream) {
  if (B <= 0) return SNERF_OK;
  if (L <= 0 || L > 16 || (C != 1 && C != 2 && C != 4 && C != 8) || grad == nullptr || inputs == nullptr || offsets == nullptr || offsets_host == nullptr ||
      grad_embeddings == nullptr || ws == nullptr || ((uintptr_t)ws & 255) || (grad_dtype != SNERF_DT_F32 && grad_dtype != SNERF_DT_F16) ||
      (out_dtype != SNERF_DT_F32 && out_dtype != SNERF_DT_F16))
    return SNERF_ERR_ARG;
  const bool hrec = half_records != 0 && (C == 1 || C == 4);              // (C = 2 / 8: fp32 records whatever the gradient's dtype)
  const bool point_major = grad_stride_l == C && grad_stride_b == (long)L * C, level_major = grad_stride_b == C && grad_stride_l == B * C;
  if (!point_major && !level_major) return SNERF_ERR_ARG;
  const int gsz = grad_dtype == SNERF_DT_F16 ? 2 : 4;
  G3Plan pl;
  if (g3_make_plan(B, C, L, offsets_host, hrec, !level_major, gsz, ws_bytes, &pl, nullptr) != SNERF_OK) return SNERF_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  char* base = (char*)ws;
  (void)hipMemsetAsync(base + pl.o_scale, 0, pl.zero_bytes, s);
  int* scale = (int*)(base + pl.o_scale);
  int* counts = (int*)(base + pl.o_counts);
  unsigned* starts = (unsigned*)(base + pl.o_starts);
  long long* g64 = pl.g64_rows > 0 ? (long long*)(base + pl.o_g64) : nullptr;
  const size_t esz = out_dtype == SNERF_DT_F16 ? 2 : 4;
  if (level_major) {                                                     // the reference's own layout: max |grad| by the stream kernel
    if (gsz == 2) hipLaunchKernelGGL(zip_bin_absmax_kernel<_Float16>, dim3(2048), dim3(256), 0, s, (const _Float16*)grad, (long)C, B * L, C, (unsigned*)(scale + 1));
    else hipLaunchKernelGGL(zip_bin_absmax_kernel<float>, dim3(2048), dim3(256), 0, s, (const float*)grad, (long)C, B * L, C, (unsigned*)(scale + 1));
    hipLaunchKernelGGL(zip_bin_scale_kernel, dim3(1), dim3(1), 0, s, scale, 3);
  }
  const int RB = L * C * gsz;
  const int TP = RB <= 256 ? 256 : ((65536 / RB) & ~63);
  G3W a{};
  a.inputs = inputs; a.B = B; a.in_vec = (((uintptr_t)inputs) & 15) == 0 ? 1 : 0;
  a.offsets = offsets; a.Sl = S; a.H = H;
  a.wgo = (uint2*)(base + pl.o_wgo); a.bshift = pl.bshift;
  a.rec_row = (unsigned short*)(base + pl.o_row); a.rec_val = (unsigned*)(base + pl.o_val); a.cap = pl.cap; a.scale_exp = scale;
  for (int l = 0; l < L; ++l) {
    if (!level_major && l % pl.lt == 0) {                                // the next group of levels, transposed (the first sweep also finds max |grad|)
      const int nl = L - l < pl.lt ? L - l : pl.lt;
      const dim3 tgrid((unsigned)((B + TP - 1) / TP));
      const size_t lds = (size_t)TP * RB;
      if (gsz == 2) hipLaunchKernelGGL(g3_transpose_kernel<_Float16>, tgrid, dim3(256), lds, s, (const _Float16*)grad, B, L * C, C, l, nl, (_Float16*)(base + pl.o_tg), l == 0 ? (unsigned*)(scale + 1) : nullptr, TP);
      else hipLaunchKernelGGL(g3_transpose_kernel<float>, tgrid, dim3(256), lds, s, (const float*)grad, B, L * C, C, l, nl, (float*)(base + pl.o_tg), l == 0 ? (unsigned*)(scale + 1) : nullptr, TP);
      if (l == 0) hipLaunchKernelGGL(zip_bin_scale_kernel, dim3(1), dim3(1), 0, s, scale, 3);
    }
    a.level = l; a.K = pl.ks[l];
    a.grad = level_major ? (const void*)((const char*)grad + (size_t)l * B * C * gsz) : (const void*)(base + pl.o_tg + (size_t)(l % pl.lt) * B * C * gsz);
    a.g_sb = C;
    int* counts_l = counts + (long)l * pl.nck * ZB_NBMAX;
    // count (all chunks of the level) -> scan
    a.counts = counts_l; a.wgs_per_chunk = pl.wgs_per_chunk; a.wg0 = 0;
    const dim3 cgrid((unsigned)pl.nwg_total);
    if (pl.grlog == 4) hipLaunchKernelGGL((g3_count_kernel<G3_WT, G3_PPT, 4>), cgrid, dim3(G3_WT), 0, s, a);
    else hipLaunchKernelGGL((g3_count_kernel<G3_WT, G3_PPT, 3>), cgrid, dim3(G3_WT), 0, s, a);
    hipLaunchKernelGGL(g3_scan_chunks_kernel, dim3(pl.nck), dim3(1024), 0, s, counts_l, starts);
    const bool image = pl.ks[l] > 1 || pl.nck > 1;
    if (pl.ks[l] > 1) (void)hipMemsetAsync(g64, 0, (size_t)pl.rows[l] * C * 8, s);
    G3A ac{};
    ac.rec_row = a.rec_row; ac.rec_val = (const float*)a.rec_val; ac.cap = pl.cap; ac.bshift = pl.bshift; ac.K = pl.ks[l]; ac.rows_l = pl.rows[l];
    ac.g64 = image ? g64 : nullptr; ac.out = (char*)grad_embeddings + (size_t)offsets_host[l] * C * esz; ac.scale_exp = scale;
    const size_t lds = (size_t)(1 << pl.bshift) * C * 8;
    for (int c = 0; c < pl.nck; ++c) {
      a.wg0 = (unsigned)(c * pl.wgs_per_chunk);
      a.starts = starts + (long)c * ZB_NBMAX;
      const long nwg_c = pl.nwg_total - (long)c * pl.wgs_per_chunk < pl.wgs_per_chunk ? pl.nwg_total - (long)c * pl.wgs_per_chunk : pl.wgs_per_chunk;
      const dim3 wgrid((unsigned)nwg_c);
#define G3WR(GT, CC, HH) do { (void)hipFuncSetAttribute((const void*)g3_write_staged_kernel<GT, CC, HH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G3Cfg<CC, HH>::LDS); \
                              hipLaunchKernelGGL((g3_write_staged_kernel<GT, CC, HH>), wgrid, dim3(G3_WT), (G3Cfg<CC, HH>::LDS), s, a); } while (0)
      if (gsz == 2) {
        if (C == 4) { if (hrec) G3WR(_Float16, 4, true); else G3WR(_Float16, 4, false); }
        else if (C == 1) { if (hrec) G3WR(_Float16, 1, true); else G3WR(_Float16, 1, false); }
        else if (C == 2) G3WR(_Float16, 2, false); else G3WR(_Float16, 8, false);
      } else {
        if (C == 4) { if (hrec) G3WR(float, 4, true); else G3WR(float, 4, false); }
        else if (C == 1) { if (hrec) G3WR(float, 1, true); else G3WR(float, 1, false); }
        else if (C == 2) G3WR(float, 2, false); else G3WR(float, 8, false);
      }
#undef G3WR
      ac.counts = counts_l + (long)c * ZB_NBMAX; ac.starts = a.starts; ac.first = c == 0; ac.last = c == pl.nck - 1;
#define G3AC(CC, HH, GT) do { (void)hipFuncSetAttribute((const void*)g3_accumulate_kernel<CC, HH, GT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                              hipLaunchKernelGGL((g3_accumulate_kernel<CC, HH, GT>), dim3(ZB_NBMAX), dim3(1024), lds, s, ac); } while (0)
      if (out_dtype == SNERF_DT_F16) {
        if (C == 4) { if (hrec) G3AC(4, true, _Float16); else G3AC(4, false, _Float16); }
        else if (C == 1) { if (hrec) G3AC(1, true, _Float16); else G3AC(1, false, _Float16); }
        else if (C == 2) G3AC(2, false, _Float16); else G3AC(8, false, _Float16);
      } else {
        if (C == 4) { if (hrec) G3AC(4, true, float); else G3AC(4, false, float); }
        else if (C == 1) { if (hrec) G3AC(1, true, float); else G3AC(1, false, float); }
        else if (C == 2) G3AC(2, false, float); else G3AC(8, false, float);
      }
#undef G3AC
    }
    if (pl.ks[l] > 1) {
      if (out_dtype == SNERF_DT_F16) hipLaunchKernelGGL(zip_bin_finish_kernel<_Float16>, dim3(256), dim3(256), 0, s, (const long long*)g64, pl.rows[l] * C, (_Float16*)ac.out, scale);
      else hipLaunchKernelGGL(zip_bin_finish_kernel<float>, dim3(256), dim3(256), 0, s, (const long long*)g64, pl.rows[l] * C, (float*)ac.out, scale);
    }
  }
  if (out_dtype == SNERF_DT_F16) hipLaunchKernelGGL(zip_bin_overflow_mark_kernel<_Float16>, dim3(1), dim3(1), 0, s, (_Float16*)grad_embeddings, scale);
  else hipLaunchKernelGGL(zip_bin_overflow_mark_kernel<float>, dim3(1), dim3(1), 0, s, (float*)grad_embeddings, scale);
  return snerf_check_launch();
}

// ------------------------------------------------------------------------------------------------------------------
// GLO modulation of the NeRF MLP's bottleneck (internal/models.py:620-630, num_glo_features > 0: configs/360_glo*.gin): the per-ray
// (scale, shift) pair SS [R, 2 B] -- the output of lin_glo_0 / lin_glo_1 on the ray's GLO vector -- applied to the B bottleneck
// columns of every sample of the ray: xm = x * exp(scale) + shift.  Backward (one workgroup per ray, a thread per column, the ray's
// samples in order: deterministic): d x = d xm * exp(scale) + d head (the raw-density / semantic-logit gradients that enter x
// directly, columns < n_head), d scale = sum_s d xm * x * exp(scale), d shift = sum_s d xm, and the per-ray column sums of d x (the
// bias gradient of density_layer.2 is their sum over the rays).
// ------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void zip_glo_modulate_kernel(const T* __restrict__ X, long ldx, const float* __restrict__ SS, long ldss, int S, long P,
                                                              int B, T* __restrict__ out, long ldo) {
  const long total = P * B;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long p = e / B;
    const int c = (int)(e - p * B);
    const float* ss = SS + (p / S) * ldss;
    out[p * ldo + c] = from_f32<T>(to_f32(X[p * ldx + c]) * expf(ss[c]) + ss[B + c]);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void zip_glo_modulate_bwd_kernel(const T* __restrict__ dXm, long lddxm, const T* __restrict__ X, long ldx,
                                                                  const float* __restrict__ SS, long ldss, const float* __restrict__ d_head, long ldh,
                                                                  int n_head, int S, int B, T* __restrict__ dX, long lddx, float* __restrict__ dSS,
                                                                  long lddss, float* __restrict__ dxsum, long ldsum) {
  const long ray = blockIdx.x;
  for (int c = threadIdx.x; c < B; c += 256) {
    const float e = expf(SS[ray * ldss + c]);
    float gs = 0.f, gh = 0.f, gx = 0.f;
    for (int i = 0; i < S; ++i) {
      const long p = ray * S + i;
      const float g = to_f32(dXm[p * lddxm + c]);
      gs += (g * to_f32(X[p * ldx + c])) * e;
      gh += g;
      float dx = g * e;
      if (c < n_head) dx += d_head[p * ldh + c];
      const T o = from_f32<T>(dx);
      dX[p * lddx + c] = o;
      gx += to_f32(o);
    }
    dSS[ray * lddss + c] = gs;
    dSS[ray * lddss + B + c] = gh;
    dxsum[ray * ldsum + c] = gx;
  }
}

extern "C" int snerf_zip_glo_modulate(const void* X, long ldx, const float* SS, long ldss, long R, int S, int B, void* out, long ldo, int dtype,
                                      void* stream) {
  if (R <= 0) return SNERF_OK;
  if (X == nullptr || SS == nullptr || out == nullptr || S <= 0 || B <= 0 || ldx < B || ldo < B || ldss < 2 * B) return SNERF_ERR_ARG;
  const long P = R * S;
  const long blocks = (P * B + 255) / 256;
  const dim3 grid((unsigned)(blocks < 65536 * 8 ? blocks : 65536 * 8));
  if (dtype == SNERF_DT_F32) hipLaunchKernelGGL(zip_glo_modulate_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)X, ldx, SS, ldss, S, P, B, (float*)out, ldo);
  else if (dtype == SNERF_DT_BF16) hipLaunchKernelGGL(zip_glo_modulate_kernel<__bf16>, grid, dim3(256), 0, (hipStream_t)stream, (const __bf16*)X, ldx, SS, ldss, S, P, B, (__bf16*)out, ldo);
  else if (dtype == SNERF_DT_F16) hipLaunchKernelGGL(zip_glo_modulate_kernel<_Float16>, grid, dim3(256), 0, (hipStream_t)stream, (const _Float16*)X, ldx, SS, ldss, S, P, B, (_Float16*)out, ldo);
  else return SNERF_ERR_ARG;
  return snerf_check_launch();
}

extern "C" int snerf_zip_glo_modulate_bwd(const void* dXm, long lddxm, const void* X, long ldx, const float* SS, long ldss, const float* d_head,
                                          long ldh, int n_head, long R, int S, int B, void* dX, long lddx, float* dSS, long lddss, float* dxsum,
                                          long ldsum, int dtype, void* stream) {
  if (R <= 0) return SNERF_OK;
  if (dXm == nullptr || X == nullptr || SS == nullptr || dX == nullptr || dSS == nullptr || dxsum == nullptr || S <= 0 || B <= 0 || n_head < 0 ||
      n_head > B || (n_head > 0 && (d_head == nullptr || ldh < n_head)) || lddxm < B || ldx < B || lddx < B || ldss < 2 * B || lddss < 2 * B ||
      ldsum < B || R >= (1L << 31))
    return SNERF_ERR_ARG;
  const dim3 grid((unsigned)R), blk(256);
  if (dtype == SNERF_DT_F32)
    hipLaunchKernelGGL(zip_glo_modulate_bwd_kernel<float>, grid, blk, 0, (hipStream_t)stream, (const float*)dXm, lddxm, (const float*)X, ldx, SS, ldss, d_head,
                       ldh, n_head, S, B, (float*)dX, lddx, dSS, lddss, dxsum, ldsum);
  else if (dtype == SNERF_DT_BF16)
    hipLaunchKernelGGL(zip_glo_modulate_bwd_kernel<__bf16>, grid, blk, 0, (hipStream_t)stream, (const __bf16*)dXm, lddxm, (const __bf16*)X, ldx, SS, ldss,
                       d_head, ldh, n_head, S, B, (__bf16*)dX, lddx, dSS, lddss, dxsum, ldsum);
  else if (dtype == SNERF_DT_F16)
    hipLaunchKernelGGL(zip_glo_modulate_bwd_kernel<_Float16>, grid, blk, 0, (hipStream_t)stream, (const _Float16*)dXm, lddxm, (const _Float16*)X, ldx, SS, ldss,
                       d_head, ldh, n_head, S, B, (_Float16*)dX, lddx, dSS, lddss, dxsum, ldsum);
  else return SNERF_ERR_ARG;
  return snerf_check_launch();
}

// ------------------------------------------------------------------------------------------------------------------
// Featurisation backward to the RAYS (pose refinement: `cal_input_grad`, internal/models.py:491, gridencoder/grid.py:65-89,
// gridencoder.cu:199-244 + 343-369 (dy_dx, kernel_input_backward); zipnerf/train.py:187-197 re-poses origins / directions / base_x /
// base_y with a learnable pose and back-propagates through the whole renderer).  One thread per interval, all levels: for each of
// the n multisamples the position gradient collects (1) the trilinear derivative of every level's features (dy/dx of the
// reference's encoder) and (2) the erf down-weighting's dependence on the contracted std -- which depends on the position through
// det(J)^(1/3) of the contraction (coord.py:51-63) --, is carried through the contraction's Jacobian and the helix construction
// (render.py:129-168: x = lx base_x + ly base_y + t d + o) and summed per ray.  The table is gathered again (no dy_dx buffer of
// [P n, L 3 C] values is ever stored).  grad_feat = d loss / d features [P, ld] in OT.
// ------------------------------------------------------------------------------------------------------------------
template <typename TT, typename OT, int C>
__global__ __launch_bounds__(256) void zip_encode_ray_bwd_kernel(ZipEnc a, float* __restrict__ g_o, float* __restrict__ g_d,
                                                                 float* __restrict__ g_bx, float* __restrict__ g_by) {
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p >= a.R * a.S) return;
  const long ray = p / a.S;
  const int i = (int)(p - ray * a.S);
  const float t0 = a.tdist[ray * (a.S + 1) + i], t1 = a.tdist[ray * (a.S + 1) + i + 1];
  float o[3], d[3], bx[3], by[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { o[k] = a.origins[ray * 3 + k]; d[k] = a.directions[ray * 3 + k]; bx[k] = a.base_x[ray * 3 + k]; by[k] = a.base_y[ray * 3 + k]; }
  const float rad = a.radii[ray];
  const OT* gi = (const OT*)a.feat + p * a.ld;
  float go[3] = {0.f, 0.f, 0.f}, gd[3] = {0.f, 0.f, 0.f}, gbx[3] = {0.f, 0.f, 0.f}, gby[3] = {0.f, 0.f, 0.f};
  for (int j = 0; j < a.n; ++j) {
    // forward of the multisample, intermediates kept (zip_sample_point)
    const float t = t0 + (t1 - t0) * ((float)j + 0.5f) / (float)a.n;
    float deg = 2.f * 3.14159265358979f * (float)a.m * (float)j / (float)a.n;
    if (a.deg_jitter != nullptr) deg += a.deg_jitter[(ray * a.S + i) * a.n + j] * 3.14159265358979f * 2.f;
    float sn, cs;
    sincosf(deg, &sn, &cs);
    const float lx = rad * t * cs / 2.f, ly = rad * t * sn / 2.f;
    float x[3], z[3], x01[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) x[k] = lx * bx[k] + ly * by[k] + t * d[k] + o[k];
    const float std_raw = a.std_scale * rad * t;
    const float msq_raw = x[0] * x[0] + x[1] * x[1] + x[2] * x[2];
    const float msq = fmaxf(msq_raw, 1.1920929e-07f);
    const bool contracted = !(msq <= 1.f);
    float mag = 1.f, sc = 1.f, q = 1.f, det = 1.f, cb = 1.f, sd = std_raw / 2.f;
    if (contracted) {
      mag = sqrtf(msq); sc = (2.f * mag - 1.f) / msq; q = 2.f / mag - 1.f / msq; det = (1.f / msq) * (q * q); cb = cbrtf(det);
      sd = cb * std_raw / 2.f;
    }
    bool inb = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) { z[k] = contracted ? sc * x[k] : x[k]; x01[k] = (z[k] / 2.f + 1.f) / 2.f; inb = inb && x01[k] >= 0.f && x01[k] <= 1.f; }
    if (!inb) continue;                                     // the encoder returns zeros (and no gradient) outside the grid
    float gx01[3] = {0.f, 0.f, 0.f}, gsd = 0.f;
    for (int level = 0; level < a.L; ++level) {
      const uint32_t hs = a.offsets[level + 1] - a.offsets[level];
      const float scale = exp2f(level * a.Sl) * a.H - 1.0f;
      const uint32_t res = (uint32_t)ceilf(scale) + 1;
      const TT* tab = (const TT*)a.table + (long)a.offsets[level] * C;
      const float gs = (float)a.grid_sizes[level];
      const float u = 1.f / sqrtf(8.f * sd * sd * gs * gs);
      const float we = erff(u);
      float fr[3];
      uint32_t pg[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        zip_cell(x01[k], scale, &pg[k], &fr[k]);
      }
      float gl[C];
#pragma unroll
      for (int c = 0; c < C; ++c) gl[c] = (float)gi[level * C + c] / (float)a.n;
      float s_fg = 0.f, dfx[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int idx = 0; idx < 8; ++idx) {
        uint32_t pl[3];
        float wk[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { const bool hi = idx & (1 << k); pl[k] = pg[k] + (hi ? 1u : 0u); wk[k] = hi ? fr[k] : 1.f - fr[k]; }
        const long row = zip_grid_index(hs, res, pl);
        const ZVec<TT, C> r = *reinterpret_cast<const ZVec<TT, C>*>(tab + row * C);
        float vg = 0.f;                                      // sum_c g_c v[corner][c]
#pragma unroll
        for (int c = 0; c < C; ++c) vg += gl[c] * (float)r.v[c];
        s_fg += (wk[0] * wk[1] * wk[2]) * vg;
        dfx[0] += ((idx & 1) ? vg : -vg) * (wk[1] * wk[2]);
        dfx[1] += ((idx & 2) ? vg : -vg) * (wk[0] * wk[2]);
        dfx[2] += ((idx & 4) ? vg : -vg) * (wk[0] * wk[1]);
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) gx01[k] += we * scale * dfx[k];
      gsd += s_fg * (1.1283791671f * expf(-u * u) * (-u / sd));   // d erf(u)/d sd, u = 1 / (sqrt(8) sd gs)
    }
    float dx[3];
    if (contracted) {
      const float dz[3] = {gx01[0] / 4.f, gx01[1] / 4.f, gx01[2] / 4.f};     // x01 = (z / 2 + 1) / 2
      const float dot = dz[0] * x[0] + dz[1] * x[1] + dz[2] * x[2];
      const float dsc = (1.f / mag - sc) / msq;                              // d sc / d |x|^2
      const float dq = -1.f / (mag * msq) + 1.f / (msq * msq);
      const float ddet = -(q * q) / (msq * msq) + (2.f * q / msq) * dq;
      const float dsd = std_raw * 0.5f * (cb / (3.f * det)) * ddet;           // d sd / d |x|^2
      const float coef = msq_raw > 1.1920929e-07f ? 2.f * (dsc * dot + gsd * dsd) : 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) dx[k] = sc * dz[k] + coef * x[k];
    } else {
#pragma unroll
      for (int k = 0; k < 3; ++k) dx[k] = gx01[k] / 4.f;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) { go[k] += dx[k]; gd[k] += t * dx[k]; gbx[k] += lx * dx[k]; gby[k] += ly * dx[k]; }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    atomicAdd(g_o + ray * 3 + k, go[k]); atomicAdd(g_d + ray * 3 + k, gd[k]);
    atomicAdd(g_bx + ray * 3 + k, gbx[k]); atomicAdd(g_by + ray * 3 + k, gby[k]);
  }
}

extern "C" int snerf_zip_encode_ray_bwd(const float* tdist, const float* origins, const float* directions, const float* radii,
                                        const float* base_x, const float* base_y, const float* deg_jitter, const void* table,
                                        const int* offsets, const int* grid_sizes, const void* grad_feat, long ld, long R, int S, int L, int C,
                                        int n, int m, float Sl, int H, float std_scale, int table_dtype, int feat_dtype, float* g_origins,
                                        float* g_directions, float* g_base_x, float* g_base_y, void* stream) {
  if (R <= 0) return SNERF_OK;
  if (S <= 0 || L <= 0 || n <= 0 || table == nullptr || grad_feat == nullptr || g_origins == nullptr || g_directions == nullptr ||
      g_base_x == nullptr || g_base_y == nullptr || (C != 1 && C != 4))
    return SNERF_ERR_ARG;
  ZipEnc a{};
  a.tdist = tdist; a.origins = origins; a.directions = directions; a.radii = radii; a.base_x = base_x; a.base_y = base_y;
  a.deg_jitter = deg_jitter; a.table = table; a.offsets = offsets; a.grid_sizes = grid_sizes; a.feat = (void*)grad_feat; a.ld = ld;
  a.R = R; a.S = S; a.L = L; a.n = n; a.m = m; a.Sl = Sl; a.H = H; a.std_scale = std_scale;
  const dim3 grid((unsigned)((R * S + 255) / 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
  // table: fp32 (0) or fp16 (2); gradient features: fp32 (0), bf16 (1) or fp16 (2)
#define ZRB(TT, OT, CC) hipLaunchKernelGGL((zip_encode_ray_bwd_kernel<TT, OT, CC>), grid, block, 0, s, a, g_origins, g_directions, g_base_x, g_base_y)
  if (table_dtype == 2 && feat_dtype == SNERF_DT_BF16) { if (C == 4) ZRB(__half, __bf16, 4); else ZRB(__half, __bf16, 1); }
  else if (table_dtype == 2 && feat_dtype == SNERF_DT_F32) { if (C == 4) ZRB(__half, float, 4); else ZRB(__half, float, 1); }
  else if (table_dtype == 0 && feat_dtype == SNERF_DT_BF16) { if (C == 4) ZRB(float, __bf16, 4); else ZRB(float, __bf16, 1); }
  else if (table_dtype == 0 && feat_dtype == SNERF_DT_F32) { if (C == 4) ZRB(float, float, 4); else ZRB(float, float, 1); }
  else if (table_dtype == 2 && feat_dtype == SNERF_DT_F16) { if (C == 4) ZRB(__half, _Float16, 4); else ZRB(__half, _Float16, 1); }
  else if (table_dtype == 0 && feat_dtype == SNERF_DT_F16) { if (C == 4) ZRB(float, _Float16, 4); else ZRB(float, _Float16, 1); }
  else return SNERF_ERR_ARG;
#undef ZRB
  return snerf_check_launch();
}

// ------------------------------------------------------------------------------------------------------------------
// compositing (wave per ray)
// ------------------------------------------------------------------------------------------------------------------
#define ZMAXSEG 8
__device__ __forceinline__ float z_softplus(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float z_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

struct ZipComp {
  const float* raw_rgb; long ld_rgb;        // [P,3] fp32 or null (proposal level: rgb = 0)
  const float* raw_density; long ld_den;    // [P] fp32 (strided)
  const float* tdist; const float* dirs;
  long R; int S; int opaque; float bg, rgb_padding, density_bias;
  float* rgb; float* depth; float* acc; float* weights;
  const float* g_rgb; const float* g_depth; const float* g_acc; const float* g_w;
  float* d_raw_rgb; long ld_drgb; float* d_raw_density; long ld_dden;
  float* g_dirs;                            // optional [R,3]: d loss / d directions through the interval lengths (t1 - t0) |d|
};

__global__ __launch_bounds__(256) void zip_composite_fwd_kernel(ZipComp a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long ray = (long)blockIdx.x * 4 + wave;
  if (ray >= a.R) return;
  const int S = a.S;
  const float* d = a.dirs + ray * 3;
  const float dnorm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  const float* td = a.tdist + ray * (S + 1);
  float carry = 0.f, s_rgb[3] = {0.f, 0.f, 0.f}, s_acc = 0.f, s_log = 0.f;
  for (int base = 0; base < S; base += 64) {
    const int i = base + lane;
    const bool ok = i < S;
    float dd = 0.f, lt = 0.f;
    bool inf_last = false;
    if (ok) {
      const float t0 = td[i], t1 = td[i + 1];
      lt = logf(0.5f * (t0 + t1));
      dd = z_softplus(a.raw_density[(ray * S + i) * a.ld_den] + a.density_bias) * ((t1 - t0) * dnorm);
      inf_last = a.opaque && i == S - 1;
    }
    const float incl = wave_incl_scan_add(inf_last ? 0.f : dd, lane);
    const float excl = carry + (incl - (inf_last ? 0.f : dd));
    const float alpha = inf_last ? 1.f : 1.f - expf(-dd);
    const float w = ok ? alpha * expf(-excl) : 0.f;
    if (ok) a.weights[ray * S + i] = w;
    s_acc += w;
    s_log += w * lt;
    if (a.raw_rgb != nullptr && ok) {
      const float* rr = a.raw_rgb + (ray * S + i) * a.ld_rgb;
#pragma unroll
      for (int c = 0; c < 3; ++c) s_rgb[c] += w * (z_sigmoid(rr[c]) * (1.f + 2.f * a.rgb_padding) - a.rgb_padding);
    }
    carry += __shfl(incl, 63, 64);
  }
  s_acc = wave_sum(s_acc); s_log = wave_sum(s_log);
#pragma unroll
  for (int c = 0; c < 3; ++c) s_rgb[c] = wave_sum(s_rgb[c]);
  if (lane == 0) {
    a.acc[ray] = s_acc;
    const float bgw = fmaxf(1.f - s_acc, 0.f);
#pragma unroll
    for (int c = 0; c < 3; ++c) a.rgb[ray * 3 + c] = s_rgb[c] + bgw * a.bg;
    float dep = expf(s_log / fmaxf(s_acc, 1.1920929e-07f));
    if (dep != dep) dep = INFINITY;
    a.depth[ray] = fminf(fmaxf(dep, td[0]), td[S]);
  }
}

// dL/d(dd_i) = g_i T_{i+1} - sum_{k>i} g_k w_k, g_i = dL/dw_i (see composite.hip); the opaque last interval has alpha == 1
// independent of its density (no gradient to it).
__global__ __launch_bounds__(256) void zip_composite_bwd_kernel(ZipComp a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long ray = (long)blockIdx.x * 4 + wave;
  if (ray >= a.R) return;
  const int S = a.S;
  const float* d = a.dirs + ray * 3;
  const float dnorm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  const float* td = a.tdist + ray * (S + 1);
  float grgb[3] = {0.f, 0.f, 0.f};
  if (a.g_rgb != nullptr) { grgb[0] = a.g_rgb[ray * 3]; grgb[1] = a.g_rgb[ray * 3 + 1]; grgb[2] = a.g_rgb[ray * 3 + 2]; }
  const float acc = a.acc[ray];
  float gacc = a.g_acc != nullptr ? a.g_acc[ray] : 0.f;
  if (1.f - acc > 0.f) gacc -= a.bg * (grgb[0] + grgb[1] + grgb[2]);          // bg_w = clamp_min(1 - acc, 0)
  // depth = clip(exp(E)), E = sum w log tmid / max(acc, eps): dE/dw_i = (log tmid_i - E) / acc for acc > eps
  float gdep = 0.f, E = 0.f, inv_acc = 0.f;
  if (a.g_depth != nullptr) {
    const float dep = a.depth[ray];
    float s_log = 0.f;
    for (int base = 0; base < S; base += 64) {
      const int i = base + lane;
      if (i < S) s_log += a.weights[ray * S + i] * logf(0.5f * (td[i] + td[i + 1]));
    }
    s_log = wave_sum(s_log);
    const float den = fmaxf(acc, 1.1920929e-07f);
    E = s_log / den;
    const float raw = expf(E);
    if (raw >= td[0] && raw <= td[S]) { gdep = a.g_depth[ray] * raw; inv_acc = 1.f / den; }
    if (!(acc > 1.1920929e-07f)) E = 0.f;   // clamp active: denominator constant
    (void)dep;
  }
  const int nseg = (S + 63) / 64;
  float segG[ZMAXSEG];
  auto gval = [&](int i) {
    float g = gacc;
    if (gdep != 0.f) g += gdep * (logf(0.5f * (td[i] + td[i + 1])) - E) * inv_acc;
    if (a.g_w != nullptr) g += a.g_w[ray * S + i];
    if (a.raw_rgb != nullptr) {
      const float* rr = a.raw_rgb + (ray * S + i) * a.ld_rgb;
#pragma unroll
      for (int c = 0; c < 3; ++c) g += grgb[c] * (z_sigmoid(rr[c]) * (1.f + 2.f * a.rgb_padding) - a.rgb_padding);
    }
    return g;
  };
#pragma unroll
  for (int seg = 0; seg < ZMAXSEG; ++seg) {
    float pg = 0.f;
    const int i = seg * 64 + lane;
    if (seg < nseg && i < S) pg = gval(i) * a.weights[ray * S + i];
    segG[seg] = seg < nseg ? wave_sum(pg) : 0.f;
  }
  float carry_dd = 0.f, g_norm = 0.f;
#pragma unroll
  for (int seg = 0; seg < ZMAXSEG; ++seg) {
    if (seg >= nseg) break;
    float later = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < ZMAXSEG; ++s2) later += s2 > seg ? segG[s2] : 0.f;
    const int i = seg * 64 + lane;
    const bool ok = i < S;
    float dd = 0.f, delta = 0.f, g = 0.f, w = 0.f, spg = 0.f;
    bool inf_last = false;
    if (ok) {
      delta = (td[i + 1] - td[i]) * dnorm;
      const float x = a.raw_density[(ray * S + i) * a.ld_den] + a.density_bias;
      dd = z_softplus(x) * delta;
      spg = x > 20.f ? 1.f : z_sigmoid(x);
      w = a.weights[ray * S + i];
      g = gval(i);
      inf_last = a.opaque && i == S - 1;
      if (a.raw_rgb != nullptr) {
        const float* rr = a.raw_rgb + (ray * S + i) * a.ld_rgb;
        float* dr = a.d_raw_rgb + (ray * S + i) * a.ld_drgb;
#pragma unroll
        for (int c = 0; c < 3; ++c) { const float sg = z_sigmoid(rr[c]); dr[c] = w * grgb[c] * (1.f + 2.f * a.rgb_padding) * sg * (1.f - sg); }
      }
    }
    const float gw = g * w;
    const float incl_dd = wave_incl_scan_add(inf_last ? 0.f : dd, lane);
    const float t_next = expf(-(carry_dd + incl_dd));
    const float suffix = later + (wave_incl_rscan_add(gw, lane) - gw);
    if (ok) a.d_raw_density[(ray * S + i) * a.ld_dden] = inf_last ? 0.f : (g * t_next - suffix) * delta * spg;
    // dd_i = density_i (t1 - t0) |d|: the same dL/d(dd_i) also reaches |d| (pose refinement, render.py:170-176)
    if (ok && !inf_last) g_norm += (g * t_next - suffix) * (dd / dnorm);
    carry_dd += __shfl(incl_dd, 63, 64);
  }
  if (a.g_dirs != nullptr) {
    g_norm = wave_sum(g_norm);
    if (lane < 3) a.g_dirs[ray * 3 + lane] = g_norm * d[lane] / dnorm;
  }
}

extern "C" int snerf_zip_composite_fwd(const float* raw_rgb, long ld_rgb, const float* raw_density, long ld_den, const float* tdist,
                                       const float* dirs, long R, int S, int opaque, float bg, float rgb_padding, float density_bias,
                                       float* rgb, float* depth, float* acc, float* weights, void* stream) {
  if (R <= 0) return SNERF_OK;
  if (S <= 0 || raw_density == nullptr || rgb == nullptr || depth == nullptr || acc == nullptr || weights == nullptr) return SNERF_ERR_ARG;
  ZipComp a{};
  a.raw_rgb = raw_rgb; a.ld_rgb = ld_rgb; a.raw_density = raw_density; a.ld_den = ld_den; a.tdist = tdist; a.dirs = dirs; a.R = R; a.S = S;
  a.opaque = opaque; a.bg = bg; a.rgb_padding = rgb_padding; a.density_bias = density_bias; a.rgb = rgb; a.depth = depth; a.acc = acc; a.weights = weights;
  hipLaunchKernelGGL(zip_composite_fwd_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
  return snerf_check_launch();
}

extern "C" int snerf_zip_composite_bwd(const float* raw_rgb, long ld_rgb, const float* raw_density, long ld_den, const float* tdist,
                                       const float* dirs, long R, int S, int opaque, float bg, float rgb_padding, float density_bias,
                                       const float* weights, const float* acc, const float* depth, const float* g_rgb, const float* g_depth,
                                       const float* g_acc, const float* g_w, float* d_raw_rgb, long ld_drgb, float* d_raw_density,
                                       long ld_dden, float* g_dirs, void* stream) {
  if (R <= 0) return SNERF_OK;
  if (S <= 0 || S > 64 * ZMAXSEG || raw_density == nullptr || weights == nullptr || acc == nullptr || d_raw_density == nullptr) return SNERF_ERR_ARG;
  if ((raw_rgb != nullptr && d_raw_rgb == nullptr) || (g_depth != nullptr && depth == nullptr)) return SNERF_ERR_ARG;
  ZipComp a{};
  a.raw_rgb = raw_rgb; a.ld_rgb = ld_rgb; a.raw_density = raw_density; a.ld_den = ld_den; a.tdist = tdist; a.dirs = dirs; a.R = R; a.S = S;
  a.opaque = opaque; a.bg = bg; a.rgb_padding = rgb_padding; a.density_bias = density_bias;
  a.weights = (float*)weights; a.acc = (float*)acc; a.depth = (float*)depth;
  a.g_rgb = g_rgb; a.g_depth = g_depth; a.g_acc = g_acc; a.g_w = g_w;
  a.d_raw_rgb = d_raw_rgb; a.ld_drgb = ld_drgb; a.d_raw_density = d_raw_density; a.ld_dden = ld_dden; a.g_dirs = g_dirs;
  hipLaunchKernelGGL(zip_composite_bwd_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
  return snerf_check_launch();
}

// ------------------------------------------------------------------------------------------------------------------
// semantic compositing, both flavours of the reference.  Wave per ray, one sample per lane and pass.
//   softmax = 1 (zipnerf: internal/models.py:594-597 + internal/render.py:237-241): semantic[r, c] = sum_i detach(w[r, i])
//       softmax(logits[r, i, :])[c]; backward d logits[i, c] = w_i p_c (g_c - sum_k p_k g_k), no gradient to the weights;
//   softmax = 0 (live mip path: s-nerf/model/mip.py:175-176): semantic[r, c] = sum_i w[r, i] raw[r, i, c]; backward
//       d raw[i, c] = w_i g_c and d w_i = sum_c g_c raw[i, c] (written to g_w_out, the weights are part of the graph).
// ------------------------------------------------------------------------------------------------------------------
#define ZSEM_MAXC 32

template <typename T, bool BWD>
__global__ __launch_bounds__(256) void semantic_composite_kernel(const float* __restrict__ weights, const T* __restrict__ logits, long ld, long R,
                                                                 int S, int C, int softmax, float* __restrict__ sem,
                                                                 const float* __restrict__ g_sem, float* __restrict__ d_logits, long ld_d,
                                                                 float* __restrict__ g_w_out, const int* __restrict__ row_index) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long ray = (long)blockIdx.x * 4 + wave;
  if (ray >= R) return;
  float acc[ZSEM_MAXC], g[ZSEM_MAXC];
#pragma unroll
  for (int c = 0; c < ZSEM_MAXC; ++c) { acc[c] = 0.f; g[c] = (BWD && c < C) ? g_sem[ray * C + c] : 0.f; }
  for (int base = 0; base < S; base += 64) {
    const int i = base + lane;
    if (i >= S) continue;
    const long p = ray * S + i;
    const float w = weights[p];
    // sample compaction (inference, csrc/ert.hip): the logits of sample p live in row row_index[p]; skipped samples (-1) carry no weight
    const long lrow = (!BWD && row_index != nullptr) ? (long)row_index[p] : p;
    if (lrow < 0) continue;
    const T* lg = logits + lrow * ld;
    float v[ZSEM_MAXC];
    if (softmax) {
      float mx = -INFINITY;
#pragma unroll
      for (int c = 0; c < ZSEM_MAXC; ++c) { v[c] = c < C ? (float)lg[c] : -INFINITY; mx = fmaxf(mx, v[c]); }
      float den = 0.f;
#pragma unroll
      for (int c = 0; c < ZSEM_MAXC; ++c) { v[c] = c < C ? expf(v[c] - mx) : 0.f; den += v[c]; }
      const float inv = 1.f / den;
#pragma unroll
      for (int c = 0; c < ZSEM_MAXC; ++c) v[c] *= inv;
    } else {
#pragma unroll
      for (int c = 0; c < ZSEM_MAXC; ++c) v[c] = c < C ? (float)lg[c] : 0.f;
    }
    if (BWD) {
      float dot = 0.f;
#pragma unroll
      for (int c = 0; c < ZSEM_MAXC; ++c) dot += v[c] * g[c];
#pragma unroll
      for (int c = 0; c < ZSEM_MAXC; ++c)
        if (c < C) d_logits[p * ld_d + c] = softmax ? w * v[c] * (g[c] - dot) : w * g[c];
      if (g_w_out != nullptr) g_w_out[p] = softmax ? 0.f : dot;
    } else {
#pragma unroll
      for (int c = 0; c < ZSEM_MAXC; ++c) acc[c] += w * v[c];
    }
  }
  if (!BWD) {
#pragma unroll
    for (int c = 0; c < ZSEM_MAXC; ++c) {
      const float t = wave_sum(acc[c]);
      if (lane == 0 && c < C) sem[ray * C + c] = t;
    }
  }
}

extern "C" int snerf_semantic_composite_fwd(const float* weights, const void* logits, long ld, int dtype, long R, int S, int C, int softmax,
                                            const int* row_index, float* sem, void* stream) {
  if (R <= 0) return SNERF_OK;
  if (S <= 0 || C <= 0 || C > ZSEM_MAXC || weights == nullptr || logits == nullptr || sem == nullptr) return SNERF_ERR_ARG;
  const dim3 grid((unsigned)((R + 3) / 4));
  if (dtype == SNERF_DT_F32) hipLaunchKernelGGL((semantic_composite_kernel<float, false>), grid, dim3(256), 0, (hipStream_t)stream, weights, (const float*)logits, ld, R, S, C, softmax, sem, nullptr, nullptr, 0, nullptr, row_index);
  else if (dtype == SNERF_DT_BF16) hipLaunchKernelGGL((semantic_composite_kernel<__bf16, false>), grid, dim3(256), 0, (hipStream_t)stream, weights, (const __bf16*)logits, ld, R, S, C, softmax, sem, nullptr, nullptr, 0, nullptr, row_index);
  else if (dtype == SNERF_DT_F16) hipLaunchKernelGGL((semantic_composite_kernel<_Float16, false>), grid, dim3(256), 0, (hipStream_t)stream, weights, (const _Float16*)logits, ld, R, S, C, softmax, sem, nullptr, nullptr, 0, nullptr, row_index);
  else return SNERF_ERR_ARG;
  return snerf_check_launch();
}

extern "C" int snerf_semantic_composite_bwd(const float* weights, const void* logits, long ld, int dtype, const float* g_sem, long R, int S,
                                            int C, int softmax, float* d_logits, long ld_d, float* g_w_out, void* stream) {
  if (R <= 0) return SNERF_OK;
  if (S <= 0 || C <= 0 || C > ZSEM_MAXC || weights == nullptr || logits == nullptr || g_sem == nullptr || d_logits == nullptr || ld_d < C) return SNERF_ERR_ARG;
  const dim3 grid((unsigned)((R + 3) / 4));
  if (dtype == SNERF_DT_F32) hipLaunchKernelGGL((semantic_composite_kernel<float, true>), grid, dim3(256), 0, (hipStream_t)stream, weights, (const float*)logits, ld, R, S, C, softmax, nullptr, g_sem, d_logits, ld_d, g_w_out, nullptr);
  else if (dtype == SNERF_DT_BF16) hipLaunchKernelGGL((semantic_composite_kernel<__bf16, true>), grid, dim3(256), 0, (hipStream_t)stream, weights, (const __bf16*)logits, ld, R, S, C, softmax, nullptr, g_sem, d_logits, ld_d, g_w_out, nullptr);
  else if (dtype == SNERF_DT_F16) hipLaunchKernelGGL((semantic_composite_kernel<_Float16, true>), grid, dim3(256), 0, (hipStream_t)stream, weights, (const _Float16*)logits, ld, R, S, C, softmax, nullptr, g_sem, d_logits, ld_d, g_w_out, nullptr);
  else return SNERF_ERR_ARG;
  return snerf_check_launch();
}

// ---------------------------------------------------------------------------
// Distance percentiles of compute_extras (render.py:255-267 -> stepfun.weighted_percentile :329-339): the histogram is extended by
// a fence post at t_far carrying the background weight, integrated to cw0 = [0, min(1, cumsum(w)), 1] (:106-126, the background
// weight itself never enters the cumsum) and t is interpolated at ps/100 (math.sorted_interp :88-107).  One lane per ray, one walk
// over the samples for all percentiles; prefix sums in the canonical order (float64, rounded once per emitted value).
// ---------------------------------------------------------------------------
#define ZIP_MAX_PCT 8
struct ZipPct { const float *tdist, *weights, *t_far; long R; int S, np; float x[ZIP_MAX_PCT]; float* out; };

__global__ __launch_bounds__(256) void zip_percentile_kernel(ZipPct a) {
  const long r = (long)blockIdx.x * 256 + threadIdx.x;
  if (r >= a.R) return;
  const float* t = a.tdist + r * (a.S + 1);
  const float* w = a.weights + r * a.S;
  float x0[ZIP_MAX_PCT], f0[ZIP_MAX_PCT], res[ZIP_MAX_PCT];
  unsigned done = 0;
#pragma unroll
  for (int j = 0; j < ZIP_MAX_PCT; ++j) { x0[j] = 0.f; f0[j] = t[0]; res[j] = 0.f; }
  double cum = 0.0;
  for (int k = 1; k <= a.S + 1; ++k) {
    float cw, tk;
    if (k <= a.S) { cum += (double)w[k - 1]; cw = fminf((float)cum, 1.f); tk = t[k]; }
    else { cw = 1.f; tk = a.t_far[r]; }
#pragma unroll
    for (int j = 0; j < ZIP_MAX_PCT; ++j) {
      if (j >= a.np || (done >> j) & 1u) continue;
      const float x = a.x[j];
      if (x >= cw) { x0[j] = cw; f0[j] = tk; }
      else {
        float off = (x - x0[j]) / (cw - x0[j]);
        off = off != off ? 0.f : fminf(fmaxf(off, 0.f), 1.f);
        res[j] = f0[j] + off * (tk - f0[j]);
        done |= 1u << j;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < ZIP_MAX_PCT; ++j)
    if (j < a.np) a.out[r * a.np + j] = (done >> j) & 1u ? res[j] : f0[j];
}

extern "C" int snerf_zip_percentiles(const float* tdist, const float* weights, const float* t_far, long R, int S, const float* ps_host, int np,
                                     float* out, void* stream) {
  if (R <= 0) return SNERF_OK;
  if (tdist == nullptr || weights == nullptr || t_far == nullptr || ps_host == nullptr || out == nullptr || S < 1 || np < 1 || np > ZIP_MAX_PCT)
    return SNERF_ERR_ARG;
  ZipPct a{tdist, weights, t_far, R, S, np, {}, out};
  for (int j = 0; j < np; ++j) a.x[j] = ps_host[j] / 100.f;
  hipLaunchKernelGGL(zip_percentile_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return snerf_check_launch();
}
