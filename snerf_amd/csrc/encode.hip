// Sample -> feature encoders that write straight into the MLP's A-operand
// buffers (row-major [M, ld], dtype fp32 or bf16, zero padded to the GEMM's
// K granularity) so that encodings never make a separate HBM round trip.
//
//   snerf_classic_embed : run_nerf_helpers.py:22-70 (Embedder) + the viewdir
//                         broadcast of run_network (run_nerf_helpers.py:465-469)
//   snerf_mip_encode    : mip.py:381-395 (sample2enc: Transform -> cast_rays ->
//                         contraction fn2 -> Jacobi_g -> J diag(c) J^T) fused with
//                         mip.py:94-118 (integrated_pos_enc, full-cov branch, of
//                         which only the diagonal is live) and mip.py:24-28.
//   snerf_mip_viewenc   : mip.py:12-21 (pos_enc of the view direction), tiled
//                         per sample like models.py:285-287.
//
// Compiled with -ffp-contract=off: products and sums round exactly like the
// CPU oracle's separate fp32 operations.
#include "common.h"

// Workgroup = CE_ROWS rows.  Phase 1: one work item per (row, frequency band, component) -- ONE sincosf feeds the sin and the cos
// column of the band (the element-per-thread form this replaces called sinf / cosf separately, divided a 64-bit index by the row
// width per element and stored 2 bytes per lane: 2.5 ms per 6.3 M rows, measured round 3) -- into an LDS tile; phase 2: the tile
// leaves in 16-byte stores to the up to three destinations (the embedding, its copy in the skip-concat buffer, the view embedding).
#define CE_ROWS 64
template <typename T>
__global__ __launch_bounds__(256) void classic_embed_kernel(const float* __restrict__ pts, const float* __restrict__ viewdirs,
                                                            int vd_stride, int S, long M, int L, int Lv, T* dst1, long ld1,
                                                            T* dst2, long ld2, int w_pts, T* dstv, long ldv, int w_views, int vec) {
  extern __shared__ __attribute__((aligned(16))) char ce_smem[];
  T* tile = (T*)ce_smem;
  const int wtot = w_pts + w_views;                      // tile row: [point embedding + pad | view embedding + pad]
  const long row0 = (long)blockIdx.x * CE_ROWS;
  const int rows = (int)(M - row0 < CE_ROWS ? M - row0 : CE_ROWS);
  const int np = 3 * (L + 1), nv = w_views > 0 ? 3 * (Lv + 1) : 0, per_row = np + nv;
  // padding columns
  const int pad_p = w_pts - (3 + 6 * L), pad_v = w_views > 0 ? w_views - (3 + 6 * Lv) : 0, pad = pad_p + pad_v;
  for (int i = threadIdx.x; i < rows * pad; i += 256) {
    const int r = i / pad, j = i - r * pad;
    tile[r * wtot + (j < pad_p ? 3 + 6 * L + j : w_pts + 3 + 6 * Lv + (j - pad_p))] = from_f32<T>(0.f);
  }
  for (int i = threadIdx.x; i < rows * per_row; i += 256) {
    const int r = i / per_row;
    int j = i - r * per_row;
    const long m = row0 + r;
    const bool is_view = j >= np;
    const float* src = is_view ? viewdirs + (m / S) * vd_stride : pts + m * 3;
    T* out = tile + r * wtot + (is_view ? w_pts : 0);
    if (is_view) j -= np;
    const int k = j / 3 - 1, c = j - 3 * (k + 1);
    const float x = src[c];
    if (k < 0) out[c] = from_f32<T>(x);
    else {
      float sn, cs;
      sincosf(x * (float)(1 << k), &sn, &cs);
      out[3 + 6 * k + c] = from_f32<T>(sn);
      out[3 + 6 * k + 3 + c] = from_f32<T>(cs);
    }
  }
  __syncthreads();
  if (vec) {                                             // every row piece is a whole number of aligned 16-byte chunks
    constexpr int E = 16 / (int)sizeof(T);
    const int cp = w_pts / E, cv = w_views / E, ct = cp + cv;
    for (int i = threadIdx.x; i < rows * ct; i += 256) {
      const int r = i / ct, ch = i - r * ct;
      const uint4 v = *(const uint4*)(tile + r * wtot + ch * E);
      const long m = row0 + r;
      if (ch < cp) {
        *(uint4*)(dst1 + m * ld1 + ch * E) = v;
        if (dst2 != nullptr) *(uint4*)(dst2 + m * ld2 + ch * E) = v;
      } else {
        *(uint4*)(dstv + m * ldv + (ch - cp) * E) = v;
      }
    }
  } else {
    for (int i = threadIdx.x; i < rows * wtot; i += 256) {
      const int r = i / wtot, col = i - r * wtot;
      const T v = tile[i];
      const long m = row0 + r;
      if (col < w_pts) {
        dst1[m * ld1 + col] = v;
        if (dst2 != nullptr) dst2[m * ld2 + col] = v;
      } else {
        dstv[m * ldv + (col - w_pts)] = v;
      }
    }
  }
}

template <typename T>
static void classic_embed_launch(const float* pts, const float* viewdirs, int vd_stride, int S, long M, int L, int Lv, void* dst1, long ld1,
                                 void* dst2, long ld2, int w_pts, void* dstv, long ldv, int w_views, hipStream_t s) {
  constexpr int E = 16 / (int)sizeof(T);
  auto ok = [&](const void* p, long ld, int w) { return p == nullptr || ((((uintptr_t)p) & 15) == 0 && (ld % E) == 0 && (w % E) == 0); };
  const int vec = ok(dst1, ld1, w_pts) && ok(dst2, ld2, w_pts) && ok(dstv, ldv, w_views);
  const int lds = CE_ROWS * (w_pts + w_views) * (int)sizeof(T);
  hipLaunchKernelGGL(classic_embed_kernel<T>, dim3((unsigned)((M + CE_ROWS - 1) / CE_ROWS)), dim3(256), lds, s, pts, viewdirs, vd_stride, S, M, L, Lv,
                     (T*)dst1, ld1, (T*)dst2, ld2, w_pts, (T*)dstv, ldv, w_views, vec);
}

extern "C" int snerf_classic_embed(const float* pts, const float* viewdirs, int vd_stride, int S, long M, int L, int Lv,
                                   void* dst1, long ld1, void* dst2, long ld2, int w_pts, void* dstv, long ldv, int w_views,
                                   int dtype, void* stream) {
  if (M <= 0) return SNERF_OK;
  if (w_pts < 3 + 6 * L || (viewdirs != nullptr && w_views < 3 + 6 * Lv) || S <= 0 || L < 0 || Lv < 0 || L > 24 || Lv > 24) return SNERF_ERR_ARG;
  if (viewdirs == nullptr) w_views = 0;
  if ((long)CE_ROWS * (w_pts + w_views) * 4 > 64 * 1024 || (M + CE_ROWS - 1) / CE_ROWS >= (1L << 31)) return SNERF_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == SNERF_DT_F32)
    classic_embed_launch<float>(pts, viewdirs, vd_stride, S, M, L, Lv, dst1, ld1, dst2, ld2, w_pts, dstv, ldv, w_views, s);
  else if (dtype == SNERF_DT_F16)
    classic_embed_launch<_Float16>(pts, viewdirs, vd_stride, S, M, L, Lv, dst1, ld1, dst2, ld2, w_pts, dstv, ldv, w_views, s);
  else if (dtype == SNERF_DT_BF16)
    classic_embed_launch<__bf16>(pts, viewdirs, vd_stride, S, M, L, Lv, dst1, ld1, dst2, ld2, w_pts, dstv, ldv, w_views, s);
  else
    return SNERF_ERR_ARG;
  return snerf_check_launch();
}

// ---------------------------------------------------------------------------
// mip path
// ---------------------------------------------------------------------------
__device__ __forceinline__ float mip_transform(float s, float near, float far, int idx) {
  if (idx == 0) return near * expf(s * logf(far / near));      // mip.py:8
  if (idx == 1) return 1.f / ((1.f - s) / near + s / far);     // mip.py:7
  return near * (1.f - s) + far * s;                           // mip.py:9
}

__device__ __forceinline__ float safe_sin(float x) {            // math_ops.py:6-12
  const float t = 314.15927f;                                   // float32(100 * pi)
  if (!(fabsf(x) < t)) {
    float r = fmodf(x, t);
    if (r != 0.f && r < 0.f) r += t;                            // python-sign remainder
    x = r;
  }
  return sinf(x);
}

struct MipEncArgs {
  const float* s_vals; const float* origins; const float* directions; const float* radii; const float* near; const float* far;
  int S;          // intervals per ray (s_vals row has S+1 posts)
  long M;         // N * S
  int cone, transform_idx, max_deg;
  void* dst1; long ld1; void* dst2; long ld2; int width;  // width >= 6*max_deg, zero padded
  float* means_out; float* covs_out;                       // optional [M,3] debug/parity outputs
  const int* sample_id;                                    // optional [M]: row j encodes sample sample_id[j] = ray * S + i (compacted rows)
  int fn_idx;                                              // 1: contraction (fn2 + Jacobi_g), 0: view-centred warp (fn1 + Jacobi_f)
  float viewc[3]; const float* far_max;                    // fn_idx 0: the mean camera centre; device scalar max(far) of the batch
};

template <typename T, int ROWS>
__global__ __launch_bounds__(256) void mip_encode_kernel(MipEncArgs a) {
  // ROWS rows per workgroup of 256 threads: 256 for large batches; 64 for small ones (phase 2 below is a chain of exp / sin evaluations, 64 per
  // thread at 256 rows: a 512-ray batch is then ONE wave per SIMD with nothing to hide their latency behind -- 47 us per launch whatever the
  // batch; at 64 rows the same work is four waves per SIMD, 16 evaluations each)
  __shared__ float sm[ROWS][6];
  const long mbase = (long)blockIdx.x * ROWS;
  const long m = mbase + threadIdx.x;
  if ((int)threadIdx.x < ROWS && m < a.M) {
    const long src = a.sample_id != nullptr ? (long)a.sample_id[m] : m;
    const long ray = src / a.S;
    const int i = (int)(src - ray * a.S);
    const float near = a.near[ray], far = a.far[ray];
    const float t0 = mip_transform(a.s_vals[ray * (a.S + 1) + i], near, far, a.transform_idx);
    const float t1 = mip_transform(a.s_vals[ray * (a.S + 1) + i + 1], near, far, a.transform_idx);
    const float rad = a.radii[ray];
    float t_mean, t_var, r_var;
    if (a.cone & 1) {  // conical_frustum_to_gaussian, stable form (mip.py:56-64)
      const float mu = (t0 + t1) / 2.f, hw = (t1 - t0) / 2.f;
      const float mu2 = mu * mu, hw2 = hw * hw, hw4 = hw2 * hw2;
      const float den = 3.f * mu2 + hw2;
      t_mean = mu + (2.f * mu * hw2) / den;
      t_var = hw2 / 3.f - (4.f / 15.f) * ((hw4 * (12.f * mu2 - hw2)) / (den * den));
      r_var = (rad * rad) * (mu2 / 4.f + (5.f / 12.f) * hw2 - (4.f / 15.f) * hw4 / den);
    } else {       // cylinder_to_gaussian (mip.py:73-77)
      t_mean = (t0 + t1) / 2.f;
      r_var = rad * rad / 4.f;
      t_var = (t1 - t0) * (t1 - t0) / 12.f;
    }
    float d[3], o[3], x[3], c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { d[k] = a.directions[ray * 3 + k]; o[k] = a.origins[ray * 3 + k]; }
    const float dmag = fmaxf(1e-10f, d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
#pragma unroll
    for (int k = 0; k < 3; ++k) {   // lift_gaussian diag (mip.py:31-46)
      x[k] = d[k] * t_mean + o[k];
      const float dd = d[k] * d[k];
      c[k] = t_var * dd + r_var * (1.f - dd / dmag);
    }
    const float nrm = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    float fm[3], fc[3];
    if (a.fn_idx == 0) {
      // view-centred warp fn1 (mip.py:368-369): (x - viewc) / sqrt(|x - viewc| far), and Jacobi_f (mip.py:323-340): J = (l I - x x^T)
      // / l^1.5 / sqrt(max far) with l = |x| + 1e-5 at the UNSHIFTED mean, as the reference has it; diag(J diag(c) J) as below
      float dx[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) dx[k] = x[k] - a.viewc[k];
      const float den = sqrtf(sqrtf(dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2]) * far);
#pragma unroll
      for (int k = 0; k < 3; ++k) fm[k] = dx[k] / den;
      const float lj = nrm + 1e-5f;
      const float l15 = powf(lj, 1.5f), sf = sqrtf(a.far_max[0]);
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float J = (((r == k ? lj : 0.f) - x[r] * x[k]) / l15) / sf;
          acc += (J * J) * c[k];
        }
        fc[r] = acc;
      }
    } else {
    // contraction fn2 (mip.py:371-374) and Jacobian (mip.py:343-364); radius hard-coded 3 (mip.py:386)
    const float l = nrm + 1e-8f;
#pragma unroll
    for (int k = 0; k < 3; ++k) fm[k] = l > 3.f ? (2.f - 3.f / l) * x[k] / l : x[k] / 3.f;
    const float lj = nrm + 1e-5f;
    if (lj >= 3.f) {
      const float ln = 1.f / lj, ln2 = ln * ln;
      const float p1 = -3.f * ln2 + 2.f * ln;
      const float p2 = 2.f * 3.f * (ln2 * ln2) - 2.f * (ln2 * ln);
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float J = (r == k ? p1 : 0.f) + p2 * (x[r] * x[k]);
          acc += (J * J) * c[k];
        }
        fc[r] = acc;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 3; ++r) fc[r] = ((1.f / 3.f) * (1.f / 3.f)) * c[r];
    }
    }
    if (a.cone & 2) {   // --disable_integration (models.py:132-133): samples = (means, zeros_like(covs)) -- the encoding loses its exp(-var / 2) factor
#pragma unroll
      for (int k = 0; k < 3; ++k) fc[k] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) { sm[threadIdx.x][k] = fm[k]; sm[threadIdx.x][3 + k] = fc[k]; }
    if (a.means_out != nullptr) {
#pragma unroll
      for (int k = 0; k < 3; ++k) { a.means_out[m * 3 + k] = fm[k]; a.covs_out[m * 3 + k] = fc[k]; }
    }
  }
  __syncthreads();
  // phase 2: one thread per (row, frequency band, axis) -- the sine and the cosine feature of a band share exp(-0.5 var) and differ by
  // the pi / 2 phase (mip.py:24-28, 105-118) -- 64 slots per row: 3 max_deg <= 48 live ones (max_deg <= 16; larger degrees take the
  // generic loop below), the rest write the zero padding.  Same fp32 operations per feature as one thread per output element, half the
  // exponentials and no 64-bit index divisions (round 2: 120 + 205 us per step for the two levels).
  const int nfeat = 6 * a.max_deg, half = 3 * a.max_deg;
  const int rows = (int)(a.M - mbase < ROWS ? a.M - mbase : ROWS);
  T* d1 = (T*)a.dst1; T* d2 = (T*)a.dst2;
  if (half <= 64) {
    for (int e = threadIdx.x; e < rows * 64; e += 256) {
      const int row = e >> 6, j = e & 63;
      T* o1 = d1 + (mbase + row) * a.ld1;
      T* o2 = d2 != nullptr ? d2 + (mbase + row) * a.ld2 : nullptr;
      if (j < half) {
        const int deg = j / 3, dim = j - deg * 3;
        const float sc = (float)(1 << deg);
        const float y = sm[row][dim] * sc;
        const float yv = (sm[row][3 + dim] * sc) * sc;
        const float ex = expf(-0.5f * yv);
        const T vs = from_f32<T>(ex * safe_sin(y)), vc = from_f32<T>(ex * safe_sin(y + 1.5707964f));   // float32(0.5 * pi)
        o1[j] = vs; o1[half + j] = vc;
        if (o2 != nullptr) { o2[j] = vs; o2[half + j] = vc; }
      }
      for (int col = nfeat + j; col < a.width; col += 64) {     // zero padding, spread over the row's 64 slots
        o1[col] = from_f32<T>(0.f);
        if (o2 != nullptr) o2[col] = from_f32<T>(0.f);
      }
    }
    return;
  }
  const long total = (long)rows * a.width;
  for (long e = threadIdx.x; e < total; e += 256) {
    const int row = (int)(e / a.width), col = (int)(e - (long)row * a.width);
    float v = 0.f;
    if (col < nfeat) {
      const int ph = col >= half, j = ph ? col - half : col, deg = j / 3, dim = j - deg * 3;
      const float sc = (float)(1 << deg);
      float y = sm[row][dim] * sc;
      const float yv = (sm[row][3 + dim] * sc) * sc;
      if (ph) y = y + 1.5707964f;   // float32(0.5 * pi)
      v = expf(-0.5f * yv) * safe_sin(y);
    }
    const T o = from_f32<T>(v);
    d1[(mbase + row) * a.ld1 + col] = o;
    if (d2 != nullptr) d2[(mbase + row) * a.ld2 + col] = o;
  }
}

static int mip_encode_launch(const float* s_vals, const float* origins, const float* directions, const float* radii, const float* near,
                             const float* far, long n_rays, int S, int cone, int transform_idx, int max_deg, void* dst1, long ld1, void* dst2,
                             long ld2, int width, float* means_out, float* covs_out, int dtype, const int* sample_id, long n_rows, int fn_idx,
                             float vx, float vy, float vz, const float* far_max, void* stream) {
  if (n_rays <= 0 || (sample_id != nullptr && n_rows <= 0)) return SNERF_OK;
  if (S <= 0 || width < 6 * max_deg || max_deg > 30 || dst1 == nullptr) return SNERF_ERR_ARG;
  if (sample_id != nullptr && n_rows > n_rays * (long)S) return SNERF_ERR_ARG;
  if ((fn_idx != 0 && fn_idx != 1) || (fn_idx == 0 && far_max == nullptr)) return SNERF_ERR_ARG;
  MipEncArgs a{s_vals, origins, directions, radii, near, far, S, sample_id != nullptr ? n_rows : n_rays * (long)S, cone, transform_idx, max_deg,
               dst1, ld1, dst2, ld2, width, means_out, covs_out, sample_id, fn_idx, {vx, vy, vz}, far_max};
  const bool small = a.M <= 131072;                       // (up to two workgroups of 256 rows per CU: take 64-row workgroups instead)
  const int blocks = (int)(small ? (a.M + 63) / 64 : (a.M + 255) / 256);
#define MIP_ENC(T_) do { if (small) hipLaunchKernelGGL((mip_encode_kernel<T_, 64>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, a); \
                         else hipLaunchKernelGGL((mip_encode_kernel<T_, 256>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, a); } while (0)
  if (dtype == SNERF_DT_F32) MIP_ENC(float);
  else if (dtype == SNERF_DT_F16) MIP_ENC(_Float16);
  else if (dtype == SNERF_DT_BF16) MIP_ENC(__bf16);
  else return SNERF_ERR_ARG;
#undef MIP_ENC
  return snerf_check_launch();
}

extern "C" int snerf_mip_encode(const float* s_vals, const float* origins, const float* directions, const float* radii,
                                const float* near, const float* far, long n_rays, int S, int cone, int transform_idx, int max_deg,
                                void* dst1, long ld1, void* dst2, long ld2, int width, float* means_out, float* covs_out,
                                int dtype, const int* sample_id, long n_rows, void* stream) {
  return mip_encode_launch(s_vals, origins, directions, radii, near, far, n_rays, S, cone, transform_idx, max_deg, dst1, ld1, dst2, ld2, width,
                           means_out, covs_out, dtype, sample_id, n_rows, 1, 0.f, 0.f, 0.f, nullptr, stream);
}

// the same with the warp selected: fn_idx 1 = the contraction above, 0 = the view-centred warp (mip.py:367-378: fn1 + Jacobi_f) around
// viewc = (vx, vy, vz); far_max: DEVICE scalar = max over the batch's rays.far (Jacobi_f divides by its square root, mip.py:340)
extern "C" int snerf_mip_encode_warp(const float* s_vals, const float* origins, const float* directions, const float* radii,
                                     const float* near, const float* far, long n_rays, int S, int cone, int transform_idx, int max_deg,
                                     void* dst1, long ld1, void* dst2, long ld2, int width, float* means_out, float* covs_out,
                                     int dtype, const int* sample_id, long n_rows, int fn_idx, float vx, float vy, float vz,
                                     const float* far_max, void* stream) {
  return mip_encode_launch(s_vals, origins, directions, radii, near, far, n_rays, S, cone, transform_idx, max_deg, dst1, ld1, dst2, ld2, width,
                           means_out, covs_out, dtype, sample_id, n_rows, fn_idx, vx, vy, vz, far_max, stream);
}

template <typename T>
__global__ __launch_bounds__(256) void mip_viewenc_kernel(const float* __restrict__ viewdirs, int S, long M, int deg, T* dst, long ld,
                                                          int width, const int* __restrict__ sample_id) {
  // [x, sin(2^i x) deg-major, sin(2^i x + pi/2)] per ray, replicated for each of the ray's S samples
  const long total = M * width;
  const int n3 = 3 * deg;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long m = e / width;
    const int col = (int)(e - m * width);
    const float* x = viewdirs + ((sample_id != nullptr ? (long)sample_id[m] : m) / S) * 3;
    float v = 0.f;
    if (col < 3) v = x[col];
    else if (col < 3 + 2 * n3) {
      int j = col - 3;
      const int ph = j >= n3;
      if (ph) j -= n3;
      float y = x[j % 3] * (float)(1 << (j / 3));
      if (ph) y = y + 1.5707964f;
      v = sinf(y);
    }
    dst[m * ld + col] = from_f32<T>(v);
  }
}

// dense rows (no sample_id): the encoding depends on the ray only, so one wave evaluates its ray's `width` values once (lane = column)
// and replicates them over the ray's S rows with row-contiguous stores -- instead of one sinf per output element.
template <typename T>
__global__ __launch_bounds__(256) void mip_viewenc_rays_kernel(const float* __restrict__ viewdirs, long N, int S, int deg, T* dst, long ld, int width) {
  const int lane = threadIdx.x & 63;
  const long ray = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= N) return;
  const int n3 = 3 * deg;
  const float* x = viewdirs + ray * 3;
  for (int c0 = 0; c0 < width; c0 += 64) {
    const int col = c0 + lane;
    float v = 0.f;
    if (col < 3) v = x[col];
    else if (col < 3 + 2 * n3) {
      int j = col - 3;
      const int ph = j >= n3;
      if (ph) j -= n3;
      float y = x[j % 3] * (float)(1 << (j / 3));
      if (ph) y = y + 1.5707964f;
      v = sinf(y);
    }
    if (col < width) {
      const T o = from_f32<T>(v);
      T* row = dst + (ray * S) * ld + col;
      for (int i = 0; i < S; ++i) row[(long)i * ld] = o;
    }
  }
}

extern "C" int snerf_mip_viewenc(const float* viewdirs, long n_rays, int S, int deg, void* dst, long ld, int width, int dtype,
                                 const int* sample_id, long n_rows, void* stream) {
  if (n_rays <= 0 || (sample_id != nullptr && n_rows <= 0)) return SNERF_OK;
  if (width < 3 + 6 * deg || S <= 0) return SNERF_ERR_ARG;
  if (sample_id == nullptr) {
    const dim3 g((unsigned)((n_rays + 3) / 4)), b(256);
    if (dtype == SNERF_DT_F32) hipLaunchKernelGGL(mip_viewenc_rays_kernel<float>, g, b, 0, (hipStream_t)stream, viewdirs, n_rays, S, deg, (float*)dst, ld, width);
    else if (dtype == SNERF_DT_F16) hipLaunchKernelGGL(mip_viewenc_rays_kernel<_Float16>, g, b, 0, (hipStream_t)stream, viewdirs, n_rays, S, deg, (_Float16*)dst, ld, width);
    else hipLaunchKernelGGL(mip_viewenc_rays_kernel<__bf16>, g, b, 0, (hipStream_t)stream, viewdirs, n_rays, S, deg, (__bf16*)dst, ld, width);
    return snerf_check_launch();
  }
  const long M = sample_id != nullptr ? n_rows : n_rays * (long)S, total = M * width;
  const int blocks = (int)((total + 255) / 256 < 262144 ? (total + 255) / 256 : 262144);
  if (dtype == SNERF_DT_F32)
    hipLaunchKernelGGL(mip_viewenc_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, viewdirs, S, M, deg, (float*)dst, ld, width, sample_id);
  else if (dtype == SNERF_DT_F16)
    hipLaunchKernelGGL(mip_viewenc_kernel<_Float16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, viewdirs, S, M, deg, (_Float16*)dst, ld, width, sample_id);
  else
    hipLaunchKernelGGL(mip_viewenc_kernel<__bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, viewdirs, S, M, deg, (__bf16*)dst, ld, width, sample_id);
  return snerf_check_launch();
}

// ---------------------------------------------------------------------------
// Gradient of the encoders w.r.t. the RAYS (the reference's pose refinement, s-nerf/utils/sample_utils.py:410-435: origins,
// directions and viewdirs are functions of the learnable camera pose and autograd carries the loss back through
// integrated_pos_enc (mip.py:105-118), sample2enc (contraction + Jacobian, mip.py:343-395) and cast_rays / lift_gaussian
// (mip.py:31-91)).  dE [M, ld] fp32 = d loss / d IPE features of every sample (the data gradient of the first MLP layer and of the
// skip layer, summed).  One wave per ray: its S samples are walked 64 at a time, every lane rebuilds its sample's Gaussian exactly as
// the forward does and chains the 6*max_deg feature gradients down to d origin / d direction; one wave reduction, one plain store per
// ray -- deterministic, no atomics.  The fence posts carry no ray gradient (level 0: constants; level 1: detached, mip.py:318).
// ---------------------------------------------------------------------------
__device__ __forceinline__ float safe_cos(float x) {            // d/dx of safe_sin: the remainder has unit slope
  const float t = 314.15927f;
  if (!(fabsf(x) < t)) {
    float r = fmodf(x, t);
    if (r != 0.f && r < 0.f) r += t;
    x = r;
  }
  return cosf(x);
}

struct MipEncBwd {
  const float* s_vals; const float* origins; const float* directions; const float* radii; const float* near; const float* far;
  long N; int S, cone, transform_idx, max_deg;
  const float* dE; long ld;
  float* g_origins; float* g_directions;
  int fn_idx; float viewc[3]; const float* far_max;        // fn_idx 0: the view-centred warp (mip.py:367-369 fn1 + Jacobi_f :323-340), as in MipEncArgs
};

__global__ __launch_bounds__(256) void mip_encode_bwd_kernel(MipEncBwd a) {
  const int lane = threadIdx.x & 63;
  const long ray = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= a.N) return;
  const float near = a.near[ray], far = a.far[ray], rad = a.radii[ray];
  float d[3], o[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { d[k] = a.directions[ray * 3 + k]; o[k] = a.origins[ray * 3 + k]; }
  const float dsq = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
  const float dmag = fmaxf(1e-10f, dsq);
  float go[3] = {0.f, 0.f, 0.f}, gd[3] = {0.f, 0.f, 0.f};
  const int half = 3 * a.max_deg;
  for (int i = lane; i < a.S; i += 64) {
    const float t0 = mip_transform(a.s_vals[ray * (a.S + 1) + i], near, far, a.transform_idx);
    const float t1 = mip_transform(a.s_vals[ray * (a.S + 1) + i + 1], near, far, a.transform_idx);
    float t_mean, t_var, r_var;
    if (a.cone & 1) {
      const float mu = (t0 + t1) / 2.f, hw = (t1 - t0) / 2.f;
      const float mu2 = mu * mu, hw2 = hw * hw, hw4 = hw2 * hw2;
      const float den = 3.f * mu2 + hw2;
      t_mean = mu + (2.f * mu * hw2) / den;
      t_var = hw2 / 3.f - (4.f / 15.f) * ((hw4 * (12.f * mu2 - hw2)) / (den * den));
      r_var = (rad * rad) * (mu2 / 4.f + (5.f / 12.f) * hw2 - (4.f / 15.f) * hw4 / den);
    } else {
      t_mean = (t0 + t1) / 2.f;
      r_var = rad * rad / 4.f;
      t_var = (t1 - t0) * (t1 - t0) / 12.f;
    }
    float x[3], c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      x[k] = d[k] * t_mean + o[k];
      const float dd = d[k] * d[k];
      c[k] = t_var * dd + r_var * (1.f - dd / dmag);
    }
    const float nrm = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    const float l = nrm + 1e-8f, lj = nrm + 1e-5f;
    float fm[3], fc[3], J[3][3];
    const bool warp0 = a.fn_idx == 0;
    const bool far_c = !warp0 && l > 3.f, far_j = !warp0 && lj >= 3.f;
    const float sl = far_c ? (2.f - 3.f / l) / l : 1.f / 3.f;
    float ln = 0.f, p1 = 1.f / 3.f, p2 = 0.f;
    float dxv[3] = {0.f, 0.f, 0.f}, rv = 0.f, denv = 1.f, l15 = 1.f, sf = 1.f;      // fn_idx 0
    if (warp0) {
      // fm = (x - viewc) / sqrt(|x - viewc| far); J = ((l I - x x^T) / l^1.5) / sqrt(max far), l = |x| + 1e-5 (the forward's arithmetic)
#pragma unroll
      for (int k = 0; k < 3; ++k) dxv[k] = x[k] - a.viewc[k];
      rv = sqrtf(dxv[0] * dxv[0] + dxv[1] * dxv[1] + dxv[2] * dxv[2]);
      denv = sqrtf(rv * far);
      l15 = powf(lj, 1.5f); sf = sqrtf(a.far_max[0]);
#pragma unroll
      for (int k = 0; k < 3; ++k) fm[k] = dxv[k] / denv;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) { J[r][k] = (((r == k ? lj : 0.f) - x[r] * x[k]) / l15) / sf; acc += (J[r][k] * J[r][k]) * c[k]; }
        fc[r] = acc;
      }
    } else {
#pragma unroll
    for (int k = 0; k < 3; ++k) fm[k] = far_c ? (2.f - 3.f / l) * x[k] / l : x[k] / 3.f;
    if (far_j) { ln = 1.f / lj; const float ln2 = ln * ln; p1 = -3.f * ln2 + 2.f * ln; p2 = 6.f * (ln2 * ln2) - 2.f * (ln2 * ln); }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) { J[r][k] = (r == k ? p1 : 0.f) + p2 * (x[r] * x[k]); acc += (J[r][k] * J[r][k]) * c[k]; }
      fc[r] = acc;
    }
    }
    if (a.cone & 2) {   // --disable_integration: the encoder saw zeros_like(covs), a constant -- no exp factor, no gradient through the covariance
#pragma unroll
      for (int k = 0; k < 3; ++k) fc[k] = 0.f;
    }
    // ---- feature gradients -> d fm, d fc
    const float* ge = a.dE + (ray * a.S + i) * a.ld;
    float gfm[3] = {0.f, 0.f, 0.f}, gfc[3] = {0.f, 0.f, 0.f};
    for (int deg = 0; deg < a.max_deg; ++deg) {
      const float sc = (float)(1 << deg);
#pragma unroll
      for (int dim = 0; dim < 3; ++dim) {
        const float y = fm[dim] * sc, yv = (fc[dim] * sc) * sc;
        const float e = expf(-0.5f * yv);
        const float gs = ge[deg * 3 + dim], gc = ge[half + deg * 3 + dim];
        const float y2 = y + 1.5707964f;
        gfm[dim] += sc * e * (safe_cos(y) * gs + safe_cos(y2) * gc);
        gfc[dim] += (-0.5f * sc * sc) * e * (safe_sin(y) * gs + safe_sin(y2) * gc);
      }
    }
    if (a.cone & 2) { gfc[0] = 0.f; gfc[1] = 0.f; gfc[2] = 0.f; }
    // ---- contraction and its Jacobian -> d x, d c
    float gx[3] = {0.f, 0.f, 0.f}, gcv[3] = {0.f, 0.f, 0.f};
    const float inv_n = nrm > 0.f ? 1.f / nrm : 0.f;                 // d|x|/dx = x / |x| (0 at the origin, as torch.norm's backward)
    if (warp0) {
      // d fm_k / d x_j = delta_kj / den - dx_k dx_j / (2 r^2 den)
      if (rv > 0.f) {
        const float dot = gfm[0] * dxv[0] + gfm[1] * dxv[1] + gfm[2] * dxv[2];
#pragma unroll
        for (int k = 0; k < 3; ++k) gx[k] += gfm[k] / denv - dot * dxv[k] / (2.f * rv * rv * denv);
      }
      // J_rk = (delta_rk l - x_r x_k) / (l^1.5 sf): through the outer product and through l = |x| + 1e-5
      float glj = 0.f;
      const float inv = 1.f / (l15 * sf);
#pragma unroll
      for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          gcv[k] += (J[r][k] * J[r][k]) * gfc[r];
          const float gJ = 2.f * J[r][k] * c[k] * gfc[r];
          gx[r] -= gJ * x[k] * inv;
          gx[k] -= gJ * x[r] * inv;
          glj += gJ * ((r == k ? inv : 0.f) - 1.5f * J[r][k] / lj);
        }
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) gx[k] += glj * x[k] * inv_n;
    } else {
    if (far_c) {
      const float dsl = -2.f / (l * l) + 6.f / (l * l * l);
      const float dot = gfm[0] * x[0] + gfm[1] * x[1] + gfm[2] * x[2];
#pragma unroll
      for (int k = 0; k < 3; ++k) gx[k] += sl * gfm[k] + dot * dsl * x[k] * inv_n;
    } else {
#pragma unroll
      for (int k = 0; k < 3; ++k) gx[k] += gfm[k] / 3.f;
    }
    if (far_j) {
      float gp1 = 0.f, gp2 = 0.f;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          gcv[k] += (J[r][k] * J[r][k]) * gfc[r];
          const float gJ = 2.f * J[r][k] * c[k] * gfc[r];
          if (r == k) gp1 += gJ;
          gp2 += gJ * (x[r] * x[k]);
          gx[r] += gJ * p2 * x[k];
          gx[k] += gJ * p2 * x[r];
        }
      }
      const float ln2 = ln * ln;
      const float gln = gp1 * (-6.f * ln + 2.f) + gp2 * (24.f * ln2 * ln - 6.f * ln2);
      const float glj = -ln2 * gln;
#pragma unroll
      for (int k = 0; k < 3; ++k) gx[k] += glj * x[k] * inv_n;
    } else {
#pragma unroll
      for (int k = 0; k < 3; ++k) gcv[k] += gfc[k] / 9.f;
    }
    }
    // ---- lift_gaussian -> d origin, d direction
    float gdm = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      go[k] += gx[k];
      gd[k] += gx[k] * t_mean + gcv[k] * (2.f * t_var * d[k] - r_var * 2.f * d[k] / dmag);
      gdm += gcv[k] * r_var * (d[k] * d[k]) / (dmag * dmag);
    }
    if (dsq > 1e-10f) {
#pragma unroll
      for (int k = 0; k < 3; ++k) gd[k] += gdm * 2.f * d[k];
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) { go[k] = wave_sum(go[k]); gd[k] = wave_sum(gd[k]); }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { a.g_origins[ray * 3 + k] = go[k]; a.g_directions[ray * 3 + k] = gd[k]; }
  }
}

extern "C" int snerf_mip_encode_bwd(const float* s_vals, const float* origins, const float* directions, const float* radii, const float* near,
                                    const float* far, long n_rays, int S, int cone, int transform_idx, int max_deg, const float* dE, long ld,
                                    float* g_origins, float* g_directions, void* stream) {
  if (n_rays <= 0) return SNERF_OK;
  if (S <= 0 || max_deg < 1 || max_deg > 30 || dE == nullptr || ld < 6 * max_deg || g_origins == nullptr || g_directions == nullptr) return SNERF_ERR_ARG;
  MipEncBwd a{s_vals, origins, directions, radii, near, far, n_rays, S, cone, transform_idx, max_deg, dE, ld, g_origins, g_directions, 1, {0.f, 0.f, 0.f}, nullptr};
  hipLaunchKernelGGL(mip_encode_bwd_kernel, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
  return snerf_check_launch();
}

// the same with the warp selected (snerf_mip_encode_warp's backward): fn_idx 0 = the view-centred warp around (vx, vy, vz), far_max = the
// device scalar max(far) of the batch -- pose refinement of a model built with fn = 0 (models.py:35; mip.py:367-378)
extern "C" int snerf_mip_encode_warp_bwd(const float* s_vals, const float* origins, const float* directions, const float* radii, const float* near,
                                         const float* far, long n_rays, int S, int cone, int transform_idx, int max_deg, const float* dE, long ld,
                                         float* g_origins, float* g_directions, int fn_idx, float vx, float vy, float vz, const float* far_max,
                                         void* stream) {
  if (n_rays <= 0) return SNERF_OK;
  if (S <= 0 || max_deg < 1 || max_deg > 30 || dE == nullptr || ld < 6 * max_deg || g_origins == nullptr || g_directions == nullptr ||
      (fn_idx != 0 && fn_idx != 1) || (fn_idx == 0 && far_max == nullptr)) return SNERF_ERR_ARG;
  MipEncBwd a{s_vals, origins, directions, radii, near, far, n_rays, S, cone, transform_idx, max_deg, dE, ld, g_origins, g_directions, fn_idx, {vx, vy, vz}, far_max};
  hipLaunchKernelGGL(mip_encode_bwd_kernel, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
  return snerf_check_launch();
}

// view-direction encoding [x, sin(2^i x), sin(2^i x + pi/2)] (mip.py:12-21) is shared by the S samples of a ray: sum its feature
// gradients over the samples, then chain through the sines.  One wave per ray.
__global__ __launch_bounds__(256) void mip_viewenc_bwd_kernel(const float* __restrict__ viewdirs, long N, int S, int deg, const float* __restrict__ dV,
                                                              long ld, float* __restrict__ g_viewdirs) {
  const int lane = threadIdx.x & 63;
  const long ray = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= N) return;
  const int n3 = 3 * deg, width = 3 + 2 * n3;
  float g[3] = {0.f, 0.f, 0.f};
  const float x[3] = {viewdirs[ray * 3], viewdirs[ray * 3 + 1], viewdirs[ray * 3 + 2]};
  for (int i = lane; i < S; i += 64) {
    const float* gv = dV + (ray * S + i) * ld;
    for (int col = 0; col < width; ++col) {
      const float v = gv[col];
      if (col < 3) g[col] += v;
      else {
        int j = col - 3;
        const int ph = j >= n3;
        if (ph) j -= n3;
        const float sc = (float)(1 << (j / 3));
        float y = x[j % 3] * sc;
        if (ph) y = y + 1.5707964f;
        g[j % 3] += v * sc * cosf(y);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) g[k] = wave_sum(g[k]);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) g_viewdirs[ray * 3 + k] = g[k];
  }
}

extern "C" int snerf_mip_viewenc_bwd(const float* viewdirs, long n_rays, int S, int deg, const float* dV, long ld, float* g_viewdirs, void* stream) {
  if (n_rays <= 0) return SNERF_OK;
  if (S <= 0 || deg < 0 || deg > 16 || dV == nullptr || ld < 3 + 6 * deg || g_viewdirs == nullptr) return SNERF_ERR_ARG;
  hipLaunchKernelGGL(mip_viewenc_bwd_kernel, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, (hipStream_t)stream, viewdirs, n_rays, S, deg, dV, ld,
                     g_viewdirs);
  return snerf_check_launch();
}
