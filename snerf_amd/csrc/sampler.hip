// Hierarchical samplers (index-producing; bit-exact against oracle/).
//
//   snerf_classic_sample_pdf : run_nerf_helpers.py:336-379 (sample_pdf) incl. the
//                              caller's z_vals_mid / weights[...,1:-1] slicing
//                              (render.py:378-380)
//   snerf_classic_merge_sort : render.py:383  torch.sort(cat([z_vals, z_samples]))
//   snerf_mip_resample       : mip.py:294-316 (blur-pool + padding) fused with
//                              math_ops.py:19-76 (sorted_piecewise_constant_pdf)
//   snerf_stratified         : render.py:330-352 / mip.py:268-288 jittered fence posts
//
// Exactness contract (oracle/common.py): row sums and prefix sums accumulate
// sequentially in fp64 and round once to fp32 per emitted value; everything
// else is a single correctly rounded fp32 operation.  Hence one LANE walks one
// ray sequentially, and this file is compiled with -ffp-contract=off.
#include "common.h"

#define LPR_THREADS 64  // one wave per workgroup; each lane owns one ray

// ---------------------------------------------------------------------------
// classic sample_pdf.  cdf has nc entries = number of bins; nc-1 weights.
// mid_mode 1 (how render_rays calls it, render.py:378-380): `bins` points at
// z_vals [N, nc+1] and bin j = 0.5 * (z[j+1] + z[j]); the caller passes
// weights + 1 so that w[j] = weights[j+1].  mid_mode 0: bins [N, nc] as given.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(LPR_THREADS) void classic_sample_pdf_kernel(const float* __restrict__ bins, long ld_bins, int mid_mode,
                                                                         const float* __restrict__ weights, long ld_w, int nc,
                                                                         const float* __restrict__ u, long u_stride, long N, int Nf,
                                                                         float* __restrict__ samples, int* __restrict__ inds,
                                                                         float* __restrict__ z_std) {
  extern __shared__ float lds[];
  const int stride = nc | 1;          // odd stride: lanes hit distinct banks
  float* cdf = lds + threadIdx.x * stride;
  const long ray = (long)blockIdx.x * LPR_THREADS + threadIdx.x;
  if (ray >= N) return;
  const float* w = weights + ray * ld_w;
  const float* b = bins + ray * ld_bins;
  double acc = 0.0;
  for (int j = 0; j < nc - 1; ++j) acc += (double)(w[j] + 1e-5f);
  const float wsum = (float)acc;
  acc = 0.0;
  cdf[0] = 0.f;
  for (int j = 0; j < nc - 1; ++j) {
    const float pdf = (w[j] + 1e-5f) / wsum;
    acc += (double)pdf;
    cdf[j + 1] = (float)acc;
  }
  const float* ur = u + ray * u_stride;
  double s1 = 0.0;
  for (int k = 0; k < Nf; ++k) {
    const float uk = ur[k];
    // inds = #(cdf <= u): upper bound by binary search over nc sorted entries
    int lo = 0, hi = nc;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cdf[mid] <= uk) lo = mid + 1; else hi = mid;
    }
    const int below = lo - 1 > 0 ? lo - 1 : 0;
    const int above = lo < nc - 1 ? lo : nc - 1;
    const float c0 = cdf[below], c1 = cdf[above];
    const float b0 = mid_mode ? 0.5f * (b[below + 1] + b[below]) : b[below];
    const float b1 = mid_mode ? 0.5f * (b[above + 1] + b[above]) : b[above];
    float denom = c1 - c0;
    if (denom < 1e-5f) denom = 1.f;
    const float t = (uk - c0) / denom;
    const float sv = b0 + t * (b1 - b0);
    samples[ray * Nf + k] = sv;
    s1 += (double)sv;
    if (inds != nullptr) inds[ray * Nf + k] = lo;
  }
  if (z_std != nullptr) {   // torch.std(z_samples, -1, unbiased=False), render.py:405
    const double mean = s1 / Nf;
    double s2 = 0.0;
    for (int k = 0; k < Nf; ++k) { const double dv = (double)samples[ray * Nf + k] - mean; s2 += dv * dv; }
    z_std[ray] = (float)sqrt(s2 / Nf);
  }
}

extern "C" int snerf_classic_sample_pdf(const float* bins, long ld_bins, int mid_mode, const float* weights, long ld_w, int nc,
                                        const float* u, long u_stride, long N, int Nf, float* samples, int* inds, float* z_std,
                                        void* stream) {
  if (N <= 0) return SNERF_OK;
  if (nc < 2 || Nf <= 0) return SNERF_ERR_ARG;
  const int stride = nc | 1;
  const size_t lds = (size_t)LPR_THREADS * stride * sizeof(float);
  if (lds > 160 * 1024) return SNERF_ERR_ARG;
  static bool attr = false;
  if (!attr) { hipFuncSetAttribute((const void*)classic_sample_pdf_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
  const int blocks = (int)((N + LPR_THREADS - 1) / LPR_THREADS);
  hipLaunchKernelGGL(classic_sample_pdf_kernel, dim3(blocks), dim3(LPR_THREADS), lds, (hipStream_t)stream, bins, ld_bins, mid_mode, weights,
                     ld_w, nc, u, u_stride, N, Nf, samples, inds, z_std);
  return snerf_check_launch();
}

// ---------------------------------------------------------------------------
// merge + sort: one wave per ray, rank sort in LDS (values only, so any
// stable order gives the bit pattern torch.sort returns).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void merge_sort_kernel(const float* __restrict__ a, int na, const float* __restrict__ b, int nb, long N,
                                                         float* __restrict__ out) {
  extern __shared__ float lds[];
  const int n = na + nb;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long ray = (long)blockIdx.x * 4 + wave;
  if (ray >= N) return;
  float* v = lds + wave * n;
  for (int i = lane; i < na; i += 64) v[i] = a[ray * na + i];
  for (int i = lane; i < nb; i += 64) v[na + i] = b[ray * nb + i];
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  for (int i = lane; i < n; i += 64) {
    const float x = v[i];
    int rank = 0;
    for (int j = 0; j < n; ++j) {
      const float y = v[j];
      rank += (y < x || (y == x && j < i)) ? 1 : 0;
    }
    out[ray * n + rank] = x;
  }
}

extern "C" int snerf_classic_merge_sort(const float* a, int na, const float* b, int nb, long N, float* out, void* stream) {
  if (N <= 0) return SNERF_OK;
  if (na < 0 || nb < 0 || na + nb <= 0) return SNERF_ERR_ARG;
  const size_t lds = (size_t)4 * (na + nb) * sizeof(float);
  if (lds > 64 * 1024) return SNERF_ERR_ARG;
  hipLaunchKernelGGL(merge_sort_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), lds, (hipStream_t)stream, a, na, b, nb, N, out);
  return snerf_check_launch();
}

// ---------------------------------------------------------------------------
// mip resample: s_vals [N,S+1], weights [N,S] -> new fence posts [N,Nf] + idx
// ---------------------------------------------------------------------------
#define MRS_THREADS 256    // 4 waves: wave 0 owns the serial prefix sums (lane per ray), all four the lookups (lane per output sample)
// MRS_RAYS: rays per workgroup.  64 for large batches; 16 up to 16 384 rays (a training batch: 4096 rays are 256 workgroups instead of 64 -- the
// kernel is a chain of dependent LDS reads, 44 us whatever the batch at 64 rays per workgroup -- and each wave looks up 4 rays instead of 16)
template <int MRS_RAYS>
__global__ __launch_bounds__(MRS_THREADS) void mip_resample_kernel(const float* __restrict__ s_vals, const float* __restrict__ weights,
                                                                   const float* __restrict__ u, long u_stride, long N, int S, int Nf,
                                                                   float padding_c, float* __restrict__ out, int* __restrict__ idx_out) {
  extern __shared__ float lds[];
  const int stride = (S + 1) | 1;
  float* const cdf_all = lds;                                  // [MRS_RAYS][stride]: weights, then blurred weights, then the CDF
  float* const bins_all = lds + MRS_RAYS * stride;             // [MRS_RAYS][S + 1] fence posts
  float* const u_all = bins_all + MRS_RAYS * (S + 1);          // [MRS_RAYS or 1][Nf]
  const long ray0 = (long)blockIdx.x * MRS_RAYS;
  const int nr = (int)min((long)MRS_RAYS, N - ray0);
  // stage the workgroup's rows with coalesced loads (round 2 read them lane-per-ray: a different row per lane)
  for (int e = threadIdx.x; e < nr * S; e += MRS_THREADS) cdf_all[(e / S) * stride + (e % S)] = weights[ray0 * S + e];
  for (int e = threadIdx.x; e < nr * (S + 1); e += MRS_THREADS) bins_all[e] = s_vals[ray0 * (S + 1) + e];
  if (u_stride == Nf) { for (int e = threadIdx.x; e < nr * Nf; e += MRS_THREADS) u_all[e] = u[ray0 * Nf + e]; }
  else if (u_stride == 0) { for (int e = threadIdx.x; e < Nf; e += MRS_THREADS) u_all[e] = u[e]; }
  __syncthreads();
  if ((int)threadIdx.x < nr) {
    float* cdf = cdf_all + threadIdx.x * stride;   // S+1 entries; slot j holds weight j until slot j is overwritten by the blurred weight j - 1
    // blur-pool (mip.py:296-306): wmax[i] = max(wp[i], wp[i+1]) with wp = [w0, w0..wS-1, wS-1];
    // blur[j] = 0.5 * (wmax[j] + wmax[j+1]) + padding
    double acc = 0.0;
    float wprev = cdf[0], wcur = cdf[0];
    float mx_lo = fmaxf(wprev, wcur);          // wmax[0] = max(wp[0], wp[1]) = w0
    for (int j = 0; j < S; ++j) {
      const float wnext = j + 1 < S ? cdf[j + 1] : wcur;   // (w[S - 1] = the current weight when j = S - 1)
      const float mx_hi = fmaxf(wcur, wnext);  // wmax[j+1]
      const float blur = 0.5f * (mx_lo + mx_hi) + padding_c;
      cdf[j + 1] = blur;                       // (weight j + 1 is already in `wnext`)
      acc += (double)blur;
      mx_lo = mx_hi; wcur = wnext;
    }
    // sorted_piecewise_constant_pdf (math_ops.py:31-46)
    float wsum = (float)acc;
    const float padding = fmaxf(0.f, 1e-5f - wsum);
    const float padw = padding / (float)S;
    wsum = wsum + padding;
    acc = 0.0;
    cdf[0] = 0.f;
    for (int j = 0; j < S - 1; ++j) {
      const float pdf = (cdf[j + 1] + padw) / wsum;
      acc += (double)pdf;
      cdf[j + 1] = fminf(1.f, (float)acc);
    }
    cdf[S] = 1.f;
  }
  __syncthreads();
  // lookups, lane per OUTPUT sample: wave w walks rays w, w + 4, ... with its 64 lanes searching that ray's CDF at once; everything it
  // reads is in LDS, the outputs leave row-contiguously.  (Round 2 kept the lane-per-ray mapping here too: ~1000 dependent steps per
  // lane, 110-125 us whatever the batch size.)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool u_lds = u_stride == Nf || u_stride == 0;
  for (int r = wave; r < nr; r += MRS_THREADS / 64) {
    const float* cdf_r = cdf_all + r * stride;
    const float* bins_r = bins_all + r * (S + 1);
    const long ray_r = ray0 + r;
    const float* ur = u_lds ? u_all + (u_stride == 0 ? 0 : r * Nf) : u + ray_r * u_stride;
    for (int k = lane; k < Nf; k += 64) {
      const float uk = ur[k];
      int lo = 0, hi = S + 1;                // #(cdf <= u) over S+1 sorted entries
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cdf_r[mid] <= uk) lo = mid + 1; else hi = mid;
      }
      int i0 = lo - 1;
      if (i0 < 0) i0 = 0;
      if (i0 > S - 1) i0 = S - 1;            // unreachable for u < 1; keeps loads in bounds
      const float c0 = cdf_r[i0], c1 = cdf_r[i0 + 1];
      const float b0 = bins_r[i0], b1 = bins_r[i0 + 1];
      float t = (uk - c0) / (c1 - c0);
      if (t != t) t = 0.f;                   // nan_to_num(., 0)
      t = fminf(fmaxf(t, 0.f), 1.f);         // clip also maps +-inf like nan_to_num + clip
      out[ray_r * Nf + k] = b0 + t * (b1 - b0);
      if (idx_out != nullptr) idx_out[ray_r * Nf + k] = i0;
    }
  }
}

extern "C" int snerf_mip_resample(const float* s_vals, const float* weights, const float* u, long u_stride, long N, int S, int Nf,
                                  float resample_padding, float* out, int* idx_out, void* stream) {
  if (N <= 0) return SNERF_OK;
  if (S < 2 || Nf <= 0) return SNERF_ERR_ARG;
  const int stride = (S + 1) | 1;
  const int rays = N <= 16384 ? 16 : 64;
  const size_t lds = (size_t)rays * (stride + (S + 1) + Nf) * sizeof(float);
  if ((size_t)64 * (stride + (S + 1) + Nf) * sizeof(float) > 160 * 1024) return SNERF_ERR_ARG;       // (one limit for every batch size)
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute((const void*)mip_resample_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)mip_resample_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  const int blocks = (int)((N + rays - 1) / rays);
  if (rays == 16) hipLaunchKernelGGL(mip_resample_kernel<16>, dim3(blocks), dim3(MRS_THREADS), lds, (hipStream_t)stream, s_vals, weights, u, u_stride, N, S,
                                     Nf, resample_padding, out, idx_out);
  else hipLaunchKernelGGL(mip_resample_kernel<64>, dim3(blocks), dim3(MRS_THREADS), lds, (hipStream_t)stream, s_vals, weights, u, u_stride, N, S,
                          Nf, resample_padding, out, idx_out);
  return snerf_check_launch();
}

// ---------------------------------------------------------------------------
// stratified fence posts.  mode 0: classic z = near*(1-t) + far*t (or disparity),
// mode 1: mip s = t.  Optional jitter between midpoints (render.py:338-352,
// mip.py:279-285).  base [P] is the host's torch.linspace(0, 1, P).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stratified_kernel(const float* __restrict__ base, const float* __restrict__ rnd,
                                                         const float* __restrict__ near, const float* __restrict__ far, int nf_stride,
                                                         long N, int P, int mode, int lindisp, float* __restrict__ out) {
  const long total = N * P;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long ray = e / P;
    const int i = (int)(e - ray * P);
    auto val = [&](int k) -> float {
      const float t = base[k];
      if (mode == 1) return t;
      const float nr = near[ray * nf_stride], fr = far[ray * nf_stride];
      if (!lindisp) return nr * (1.f - t) + fr * t;
      return 1.f / (1.f / nr * (1.f - t) + 1.f / fr * t);
    };
    float v = val(i);
    if (rnd != nullptr) {
      const float lower = i > 0 ? 0.5f * (v + val(i - 1)) : v;
      const float upper = i < P - 1 ? 0.5f * (val(i + 1) + v) : v;
      v = lower + (upper - lower) * rnd[e];
    }
    out[e] = v;
  }
}

extern "C" int snerf_stratified(const float* base, const float* rnd, const float* near, const float* far, int nf_stride, long N, int P,
                                int mode, int lindisp, float* out, void* stream) {
  if (N <= 0) return SNERF_OK;
  if (P <= 0 || (mode == 0 && (near == nullptr || far == nullptr))) return SNERF_ERR_ARG;
  const long total = N * P;
  const int blocks = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
  hipLaunchKernelGGL(stratified_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, base, rnd, near, far, nf_stride, N, P, mode,
                     lindisp, out);
  return snerf_check_launch();
}

// u[n, k] = min(k * s + jit[n, k], 1 - eps)  (math_ops.py:50-54: arange(num_samples) * s + uniform_(to = s - eps), clamped below 1), in
// place over the uniform draw: the five eager launches of that expression (arange, mul, add, ones_like - eps, minimum) as one.  Same
// roundings as the eager ops: float(k) * float(s) and the sum are rounded separately (no fma).
__global__ __launch_bounds__(256) void jitter_u_kernel(float* __restrict__ u, long N, int P, float s) {
  const long total = N * P;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int k = (int)(e % P);
    u[e] = fminf(__fadd_rn(__fmul_rn((float)k, s), u[e]), 0.99999988079071044921875f);
  }
}

extern "C" int snerf_jitter_u(float* u, long N, int P, float s, void* stream) {
  if (N <= 0) return SNERF_OK;
  if (u == nullptr || P <= 0) return SNERF_ERR_ARG;
  const long total = N * P;
  const int blocks = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
  hipLaunchKernelGGL(jitter_u_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, u, N, P, s);
  return snerf_check_launch();
}

// pts[n,s,:] = o[n] + d[n] * z[n,s]   (render.py:354, :385) -- separate multiply and add like the eager ops
__global__ __launch_bounds__(256) void classic_points_kernel(const float* __restrict__ rays, int ray_stride, const float* __restrict__ z,
                                                             long N, int S, float* __restrict__ pts) {
  const long total = N * S * 3;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long m = e / 3;
    const int c = (int)(e - m * 3);
    const long ray = m / S;
    const float prod = rays[ray * ray_stride + 3 + c] * z[m];
    pts[e] = rays[ray * ray_stride + c] + prod;
  }
}

extern "C" int snerf_classic_points(const float* rays, int ray_stride, const float* z_vals, long N, int S, float* pts, void* stream) {
  if (N <= 0) return SNERF_OK;
  if (S <= 0 || ray_stride < 6) return SNERF_ERR_ARG;
  const long total = N * S * 3;
  const int blocks = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
  hipLaunchKernelGGL(classic_points_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, rays, ray_stride, z_vals, N, S, pts);
  return snerf_check_launch();
}
