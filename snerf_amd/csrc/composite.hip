// Alpha compositing along rays: one wavefront per ray, lanes over samples,
// wave-level prefix scans for the running transmittance, butterfly
// reductions for the ray integrals.  Forward and backward.
//
//   snerf_mip_composite_fwd/bwd     : models.py:166-175 (activations: rgb = sigmoid*1.002-0.001,
//                                     density = softplus(raw - 1)) fused with mip.py:151-189
//                                     (real_volumetric_rendering)
//   snerf_classic_composite_fwd/bwd : run_nerf_helpers.py:381-424 (raw2outputs)
//
// raw network outputs are fp32 [M, ld] (the MLP heads store fp32).
#include "common.h"

#define MAXSEG 8   // backward kernels keep per-64-sample-segment partial sums in registers: S <= 512

__device__ __forceinline__ float transform_s(float s, float near, float far, int idx) {
  if (idx == 0) return near * expf(s * logf(far / near));
  if (idx == 1) return 1.f / ((1.f - s) / near + s / far);
  return near * (1.f - s) + far * s;
}
__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

struct MipComp {
  const float* raw_rgb; long ld_rgb;       // [M,3] or null (proposal level)
  const float* raw_density; long ld_den;   // [M] (strided)
  const float* noise;                      // [M] added to raw density, or null
  const float* s_vals;                     // [N,S+1]
  const float* dirs; const float* near; const float* far;
  long N; int S; int transform_idx; int white; float rgb_padding; float density_bias;
  // forward outputs
  float* comp_rgb; float* distance; float* acc; float* weights;
  // backward inputs / outputs
  const float* g_rgb; const float* g_dist; const float* g_acc; const float* g_w;   // [N,3],[N],[N],[N,S] (nullable)
  float* d_raw_rgb; long ld_drgb; float* d_raw_density; long ld_dden;
  const int* row_index;                    // forward, optional [N,S]: row of sample (ray, i) in the compacted raw arrays, -1 = not evaluated (empty)
  float* g_dirs;                           // backward, optional [N,3]: d loss / d directions through delta = (t1 - t0) |d| (mip.py:160-161)
};

// ---- forward -------------------------------------------------------------
__global__ __launch_bounds__(256) void mip_composite_fwd_kernel(MipComp a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long ray = (long)blockIdx.x * 4 + wave;
  if (ray >= a.N) return;
  const int S = a.S;
  const float near = a.near[ray], far = a.far[ray];
  const float dx = a.dirs[ray * 3], dy = a.dirs[ray * 3 + 1], dz = a.dirs[ray * 3 + 2];
  const float dnorm = sqrtf(dx * dx + dy * dy + dz * dz);
  const float* sv = a.s_vals + ray * (S + 1);
  float carry = 0.f;                 // sum of density*delta over previous 64-sample segments
  float s_rgb[3] = {0.f, 0.f, 0.f}, s_acc = 0.f, s_dist = 0.f;
  for (int base = 0; base < S; base += 64) {
    const int i = base + lane;
    const bool ok = i < S;
    float dd = 0.f, tmid = 0.f;
    long row = -1;
    if (ok) {
      const float t0 = transform_s(sv[i], near, far, a.transform_idx);
      const float t1 = transform_s(sv[i + 1], near, far, a.transform_idx);
      tmid = 0.5f * (t0 + t1);
      row = a.row_index != nullptr ? (long)a.row_index[ray * S + i] : ray * S + i;
      if (row >= 0) {
        float rd = a.raw_density[row * a.ld_den];
        if (a.noise != nullptr) rd += a.noise[ray * S + i];
        dd = softplus_f(rd + a.density_bias) * ((t1 - t0) * dnorm);
      }
    }
    const float incl = wave_incl_scan_add(dd, lane);
    const float excl = carry + (incl - dd);
    const float alpha = 1.f - expf(-dd);
    const float w = ok ? alpha * expf(-excl) : 0.f;
    if (ok) a.weights[ray * S + i] = w;
    s_acc += w;
    s_dist += w * tmid;
    if (a.raw_rgb != nullptr && ok && row >= 0) {
      const float* rr = a.raw_rgb + row * a.ld_rgb;
#pragma unroll
      for (int c = 0; c < 3; ++c) s_rgb[c] += w * (sigmoid_f(rr[c]) * (1.f + 2.f * a.rgb_padding) - a.rgb_padding);
    }
    carry += __shfl(incl, 63, 64);
  }
  s_acc = wave_sum(s_acc);
  s_dist = wave_sum(s_dist);
  if (a.raw_rgb != nullptr) {
#pragma unroll
    for (int c = 0; c < 3; ++c) s_rgb[c] = wave_sum(s_rgb[c]);
  }
  if (lane == 0) {
    const float tlo = transform_s(sv[0], near, far, a.transform_idx), thi = transform_s(sv[S], near, far, a.transform_idx);
    float d = s_dist;
    if (d != d) d = INFINITY;                       // nan_to_num(distance, inf)
    d = fminf(fmaxf(d, tlo), thi);                  // clip(., t[0], t[-1])
    a.distance[ray] = d;
    a.acc[ray] = s_acc;
    if (a.raw_rgb != nullptr) {
#pragma unroll
      for (int c = 0; c < 3; ++c) a.comp_rgb[ray * 3 + c] = s_rgb[c] + (a.white ? (1.f - s_acc) : 0.f);
    }
  }
}

// ---- backward ------------------------------------------------------------
// w_i = alpha_i T_i, T_i = exp(-sum_{j<i} dd_j).  With g_i = dL/dw_i:
//   dL/ddd_i = g_i T_{i+1} - sum_{k>i} g_k w_k
__global__ __launch_bounds__(256) void mip_composite_bwd_kernel(MipComp a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long ray = (long)blockIdx.x * 4 + wave;
  if (ray >= a.N) return;
  const int S = a.S;
  const float near = a.near[ray], far = a.far[ray];
  const float dx = a.dirs[ray * 3], dy = a.dirs[ray * 3 + 1], dz = a.dirs[ray * 3 + 2];
  const float dnorm = sqrtf(dx * dx + dy * dy + dz * dz);
  const float* sv = a.s_vals + ray * (S + 1);
  float grgb[3] = {0.f, 0.f, 0.f};
  if (a.g_rgb != nullptr) { grgb[0] = a.g_rgb[ray * 3]; grgb[1] = a.g_rgb[ray * 3 + 1]; grgb[2] = a.g_rgb[ray * 3 + 2]; }
  float gacc = a.g_acc != nullptr ? a.g_acc[ray] : 0.f;
  if (a.white && a.raw_rgb != nullptr) gacc -= grgb[0] + grgb[1] + grgb[2];
  const int nseg = (S + 63) / 64;
  // pass 1: per-segment sums of g_k w_k (split into the part without the distance gradient and sum w_k tmid_k).
  // sum_k w_k tmid_k is the UNclipped distance, which decides whether clip() passes the distance gradient
  // (torch.clip backward looks at its input).  Per-segment totals let pass 2 build suffix sums by direct
  // summation (reverse scan) instead of total - prefix, which would cancel catastrophically near the far end.
  float segA[MAXSEG], segB[MAXSEG];
#pragma unroll
  for (int seg = 0; seg < MAXSEG; ++seg) {
    float pa = 0.f, pb = 0.f;
    const int i = seg * 64 + lane;
    if (seg < nseg && i < S) {
      const float w = a.weights[ray * S + i];
      const float t0 = transform_s(sv[i], near, far, a.transform_idx), t1 = transform_s(sv[i + 1], near, far, a.transform_idx);
      float g = gacc;
      if (a.g_w != nullptr) g += a.g_w[ray * S + i];
      if (a.raw_rgb != nullptr) {
        const float* rr = a.raw_rgb + (ray * S + i) * a.ld_rgb;
#pragma unroll
        for (int c = 0; c < 3; ++c) g += grgb[c] * (sigmoid_f(rr[c]) * (1.f + 2.f * a.rgb_padding) - a.rgb_padding);
      }
      pa = g * w;
      pb = w * (0.5f * (t0 + t1));
    }
    segA[seg] = seg < nseg ? wave_sum(pa) : 0.f;
    segB[seg] = seg < nseg ? wave_sum(pb) : 0.f;
  }
  float raw_dist = 0.f;
#pragma unroll
  for (int seg = 0; seg < MAXSEG; ++seg) raw_dist += segB[seg];
  float gdist = 0.f;
  if (a.g_dist != nullptr) {
    const float tlo = transform_s(sv[0], near, far, a.transform_idx), thi = transform_s(sv[S], near, far, a.transform_idx);
    gdist = (raw_dist >= tlo && raw_dist <= thi) ? a.g_dist[ray] : 0.f;   // false for NaN as well (nan -> inf -> clipped)
  }
  // pass 2: per sample gradient
  float carry_dd = 0.f, g_norm = 0.f;
#pragma unroll
  for (int seg = 0; seg < MAXSEG; ++seg) {
    if (seg >= nseg) break;
    float later = 0.f;                                             // sum of g_k w_k over all later segments
#pragma unroll
    for (int s2 = 0; s2 < MAXSEG; ++s2) later += s2 > seg ? segA[s2] + gdist * segB[s2] : 0.f;
    const int i = seg * 64 + lane;
    const bool ok = i < S;
    float dd = 0.f, delta = 0.f, g = 0.f, w = 0.f, sp_grad = 0.f;
    if (ok) {
      const float t0 = transform_s(sv[i], near, far, a.transform_idx), t1 = transform_s(sv[i + 1], near, far, a.transform_idx);
      delta = (t1 - t0) * dnorm;
      float rd = a.raw_density[(ray * S + i) * a.ld_den];
      if (a.noise != nullptr) rd += a.noise[ray * S + i];
      const float x = rd + a.density_bias;
      dd = softplus_f(x) * delta;
      sp_grad = x > 20.f ? 1.f : sigmoid_f(x);
      w = a.weights[ray * S + i];
      g = gacc + gdist * (0.5f * (t0 + t1));
      if (a.g_w != nullptr) g += a.g_w[ray * S + i];
      if (a.raw_rgb != nullptr) {
        const float* rr = a.raw_rgb + (ray * S + i) * a.ld_rgb;
        float* dr = a.d_raw_rgb + (ray * S + i) * a.ld_drgb;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float sg = sigmoid_f(rr[c]);
          g += grgb[c] * (sg * (1.f + 2.f * a.rgb_padding) - a.rgb_padding);
          dr[c] = w * grgb[c] * (1.f + 2.f * a.rgb_padding) * sg * (1.f - sg);
        }
      }
    }
    const float gw = g * w;
    const float incl_dd = wave_incl_scan_add(dd, lane);
    const float t_next = expf(-(carry_dd + incl_dd));              // T_{i+1}
    const float suffix = later + (wave_incl_rscan_add(gw, lane) - gw);   // sum_{k>i} g_k w_k
    const float g_dd = g * t_next - suffix;                         // d loss / d (density * delta)
    if (ok) a.d_raw_density[(ray * S + i) * a.ld_dden] = g_dd * delta * sp_grad;
    if (ok) g_norm += g_dd * dd;                                    // d(dd)/d|d| = dd / |d|
    carry_dd += __shfl(incl_dd, 63, 64);
  }
  if (a.g_dirs != nullptr) {
    g_norm = wave_sum(g_norm);
    if (lane == 0) {
      const float k = dnorm > 0.f ? g_norm / (dnorm * dnorm) : 0.f;   // d|d|/dd_k = d_k / |d|
      a.g_dirs[ray * 3] = k * dx; a.g_dirs[ray * 3 + 1] = k * dy; a.g_dirs[ray * 3 + 2] = k * dz;
    }
  }
}

static MipComp make_mip(const float* raw_rgb, long ld_rgb, const float* raw_density, long ld_den, const float* noise, const float* s_vals,
                        const float* dirs, const float* near, const float* far, long N, int S, int transform_idx, int white,
                        float rgb_padding, float density_bias) {
  MipComp a{};
  a.raw_rgb = raw_rgb; a.ld_rgb = ld_rgb; a.raw_density = raw_density; a.ld_den = ld_den; a.noise = noise; a.s_vals = s_vals;
  a.dirs = dirs; a.near = near; a.far = far; a.N = N; a.S = S; a.transform_idx = transform_idx; a.white = white;
  a.rgb_padding = rgb_padding; a.density_bias = density_bias;
  return a;
}

extern "C" int snerf_mip_composite_fwd(const float* raw_rgb, long ld_rgb, const float* raw_density, long ld_den, const float* noise,
                                       const float* s_vals, const float* dirs, const float* near, const float* far, long N, int S,
                                       int transform_idx, int white, float rgb_padding, float density_bias, float* comp_rgb,
                                       float* distance, float* acc, float* weights, const int* row_index, void* stream) {
  if (N <= 0) return SNERF_OK;
  if (S <= 0 || raw_density == nullptr || weights == nullptr || distance == nullptr || acc == nullptr) return SNERF_ERR_ARG;
  if (raw_rgb != nullptr && comp_rgb == nullptr) return SNERF_ERR_ARG;
  MipComp a = make_mip(raw_rgb, ld_rgb, raw_density, ld_den, noise, s_vals, dirs, near, far, N, S, transform_idx, white, rgb_padding, density_bias);
  a.comp_rgb = comp_rgb; a.distance = distance; a.acc = acc; a.weights = weights; a.row_index = row_index;
  hipLaunchKernelGGL(mip_composite_fwd_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
  return snerf_check_launch();
}

extern "C" int snerf_mip_composite_bwd(const float* raw_rgb, long ld_rgb, const float* raw_density, long ld_den, const float* noise,
                                       const float* s_vals, const float* dirs, const float* near, const float* far, long N, int S,
                                       int transform_idx, int white, float rgb_padding, float density_bias, const float* weights,
                                       const float* distance, const float* g_rgb, const float* g_dist, const float* g_acc,
                                       const float* g_w, float* d_raw_rgb, long ld_drgb, float* d_raw_density, long ld_dden,
                                       float* g_dirs, void* stream) {
  if (N <= 0) return SNERF_OK;
  if (S <= 0 || S > 64 * MAXSEG || raw_density == nullptr || weights == nullptr || d_raw_density == nullptr) return SNERF_ERR_ARG;
  if (raw_rgb != nullptr && d_raw_rgb == nullptr) return SNERF_ERR_ARG;
  if (g_dist != nullptr && distance == nullptr) return SNERF_ERR_ARG;
  MipComp a = make_mip(raw_rgb, ld_rgb, raw_density, ld_den, noise, s_vals, dirs, near, far, N, S, transform_idx, white, rgb_padding, density_bias);
  a.weights = (float*)weights; a.distance = (float*)distance;
  a.g_rgb = g_rgb; a.g_dist = g_dist; a.g_acc = g_acc; a.g_w = g_w;
  a.d_raw_rgb = d_raw_rgb; a.ld_drgb = ld_drgb; a.d_raw_density = d_raw_density; a.ld_dden = ld_dden;
  a.g_dirs = g_dirs;
  hipLaunchKernelGGL(mip_composite_bwd_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
  return snerf_check_launch();
}

// ===========================================================================
// classic raw2outputs
// ===========================================================================
struct ClassicComp {
  const float* raw; long ld;               // [M, >=4]: rgb (3) + sigma (1)
  const float* noise;                      // [N,S] or null
  const float* z_vals; const float* rays_d; int rd_stride;
  long N; int S; int white;
  float* rgb_map; float* disp_map; float* acc_map; float* weights; float* depth_map;
  const float* g_rgb; const float* g_disp; const float* g_acc; const float* g_depth; const float* g_w;
  float* d_raw; long ld_draw;
};

__global__ __launch_bounds__(256) void classic_composite_fwd_kernel(ClassicComp a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long ray = (long)blockIdx.x * 4 + wave;
  if (ray >= a.N) return;
  const int S = a.S;
  const float* d = a.rays_d + ray * a.rd_stride;
  const float dnorm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  const float* z = a.z_vals + ray * S;
  float carry = 1.f;   // product of (1 - alpha + 1e-10) over previous segments
  float s_rgb[3] = {0.f, 0.f, 0.f}, s_acc = 0.f, s_depth = 0.f;
  for (int base = 0; base < S; base += 64) {
    const int i = base + lane;
    const bool ok = i < S;
    float q = 1.f, alpha = 0.f, zi = 0.f;
    if (ok) {
      zi = z[i];
      const float dist = (i + 1 < S ? z[i + 1] - zi : 1e10f) * dnorm;
      float sg = a.raw[(ray * S + i) * a.ld + 3];
      if (a.noise != nullptr) sg += a.noise[ray * S + i];
      alpha = 1.f - expf(-fmaxf(sg, 0.f) * dist);
      q = 1.f - alpha + 1e-10f;
    }
    const float incl = wave_incl_scan_mul(q, lane);
    float excl = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = 1.f;
    const float w = ok ? alpha * (carry * excl) : 0.f;
    if (ok) {
      a.weights[ray * S + i] = w;
      const float* rr = a.raw + (ray * S + i) * a.ld;
#pragma unroll
      for (int c = 0; c < 3; ++c) s_rgb[c] += w * sigmoid_f(rr[c]);
    }
    s_acc += w;
    s_depth += w * zi;
    carry *= __shfl(incl, 63, 64);
  }
  s_acc = wave_sum(s_acc);
  s_depth = wave_sum(s_depth);
#pragma unroll
  for (int c = 0; c < 3; ++c) s_rgb[c] = wave_sum(s_rgb[c]);
  if (lane == 0) {
    a.acc_map[ray] = s_acc;
    a.depth_map[ray] = s_depth;
    // 1 / max(1e-10, depth / acc); torch.max propagates NaN (0/0 when acc == 0)
    const float r = s_depth / s_acc;
    a.disp_map[ray] = 1.f / ((r != r) ? r : fmaxf(1e-10f, r));
#pragma unroll
    for (int c = 0; c < 3; ++c) a.rgb_map[ray * 3 + c] = s_rgb[c] + (a.white ? (1.f - s_acc) : 0.f);
  }
}

// w_i = alpha_i P_i, P_i = prod_{j<i} q_j, q_j = 1 - alpha_j + 1e-10
//   dL/dalpha_i = g_i P_i - (sum_{k>i} g_k w_k) / q_i
__global__ __launch_bounds__(256) void classic_composite_bwd_kernel(ClassicComp a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long ray = (long)blockIdx.x * 4 + wave;
  if (ray >= a.N) return;
  const int S = a.S;
  const float* d = a.rays_d + ray * a.rd_stride;
  const float dnorm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  const float* z = a.z_vals + ray * S;
  float grgb[3] = {0.f, 0.f, 0.f};
  if (a.g_rgb != nullptr) { grgb[0] = a.g_rgb[ray * 3]; grgb[1] = a.g_rgb[ray * 3 + 1]; grgb[2] = a.g_rgb[ray * 3 + 2]; }
  float gacc = a.g_acc != nullptr ? a.g_acc[ray] : 0.f;
  float gdepth = a.g_depth != nullptr ? a.g_depth[ray] : 0.f;
  if (a.white) gacc -= grgb[0] + grgb[1] + grgb[2];
  if (a.g_disp != nullptr) {
    // disp = 1 / max(1e-10, depth/acc): d disp/d depth = -disp^2 / acc, d disp/d acc = disp^2 depth / acc^2 (if not clamped)
    const float acc = a.acc_map[ray], depth = a.depth_map[ray];
    const float r = depth / acc;
    if (r > 1e-10f) {
      const float disp = 1.f / r, gd = a.g_disp[ray];
      gdepth += gd * (-disp * disp / acc);
      gacc += gd * (disp * disp * depth / (acc * acc));
    }
  }
  const int nseg = (S + 63) / 64;
  float segG[MAXSEG];                                               // per-segment sums of g_k w_k
#pragma unroll
  for (int seg = 0; seg < MAXSEG; ++seg) {
    float pg = 0.f;
    const int i = seg * 64 + lane;
    if (seg < nseg && i < S) {
      const float* rr = a.raw + (ray * S + i) * a.ld;
      float g = gacc + gdepth * z[i];
      if (a.g_w != nullptr) g += a.g_w[ray * S + i];
#pragma unroll
      for (int c = 0; c < 3; ++c) g += grgb[c] * sigmoid_f(rr[c]);
      pg = g * a.weights[ray * S + i];
    }
    segG[seg] = seg < nseg ? wave_sum(pg) : 0.f;
  }
  float carry_p = 1.f;
#pragma unroll
  for (int seg = 0; seg < MAXSEG; ++seg) {
    if (seg >= nseg) break;
    float later = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < MAXSEG; ++s2) later += s2 > seg ? segG[s2] : 0.f;
    const int i = seg * 64 + lane;
    const bool ok = i < S;
    float q = 1.f, g = 0.f, w = 0.f, dist = 0.f, sg = 0.f;
    if (ok) {
      const float zi = z[i];
      dist = (i + 1 < S ? z[i + 1] - zi : 1e10f) * dnorm;
      const float* rr = a.raw + (ray * S + i) * a.ld;
      sg = rr[3];
      if (a.noise != nullptr) sg += a.noise[ray * S + i];
      const float alpha = 1.f - expf(-fmaxf(sg, 0.f) * dist);
      q = 1.f - alpha + 1e-10f;
      w = a.weights[ray * S + i];
      g = gacc + gdepth * zi;
      if (a.g_w != nullptr) g += a.g_w[ray * S + i];
      float* dr = a.d_raw + (ray * S + i) * a.ld_draw;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float sgm = sigmoid_f(rr[c]);
        g += grgb[c] * sgm;
        dr[c] = w * grgb[c] * sgm * (1.f - sgm);
      }
    }
    const float gw = g * w;
    const float incl_q = wave_incl_scan_mul(q, lane);
    float excl_q = __shfl_up(incl_q, 1, 64);
    if (lane == 0) excl_q = 1.f;
    const float P = carry_p * excl_q;
    const float suffix = later + (wave_incl_rscan_add(gw, lane) - gw);
    if (ok) {
      const float dalpha = g * P - suffix / q;
      // alpha = 1 - exp(-relu(sg) dist): d alpha / d sg = dist * exp(-sg dist) for sg > 0
      a.d_raw[(ray * S + i) * a.ld_draw + 3] = sg > 0.f ? dalpha * dist * expf(-sg * dist) : 0.f;
    }
    carry_p *= __shfl(incl_q, 63, 64);
  }
}

extern "C" int snerf_classic_composite_fwd(const float* raw, long ld, const float* noise, const float* z_vals, const float* rays_d,
                                           int rd_stride, long N, int S, int white, float* rgb_map, float* disp_map, float* acc_map,
                                           float* weights, float* depth_map, void* stream) {
  if (N <= 0) return SNERF_OK;
  if (S <= 0 || ld < 4 || raw == nullptr || rgb_map == nullptr || disp_map == nullptr || acc_map == nullptr || weights == nullptr ||
      depth_map == nullptr)
    return SNERF_ERR_ARG;
  ClassicComp a{};
  a.raw = raw; a.ld = ld; a.noise = noise; a.z_vals = z_vals; a.rays_d = rays_d; a.rd_stride = rd_stride; a.N = N; a.S = S; a.white = white;
  a.rgb_map = rgb_map; a.disp_map = disp_map; a.acc_map = acc_map; a.weights = weights; a.depth_map = depth_map;
  hipLaunchKernelGGL(classic_composite_fwd_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
  return snerf_check_launch();
}

extern "C" int snerf_classic_composite_bwd(const float* raw, long ld, const float* noise, const float* z_vals, const float* rays_d,
                                           int rd_stride, long N, int S, int white, const float* weights, const float* acc_map,
                                           const float* depth_map, const float* g_rgb, const float* g_disp, const float* g_acc,
                                           const float* g_depth, const float* g_w, float* d_raw, long ld_draw, void* stream) {
  if (N <= 0) return SNERF_OK;
  if (S <= 0 || S > 64 * MAXSEG || ld < 4 || ld_draw < 4 || raw == nullptr || weights == nullptr || d_raw == nullptr) return SNERF_ERR_ARG;
  if (g_disp != nullptr && (acc_map == nullptr || depth_map == nullptr)) return SNERF_ERR_ARG;
  ClassicComp a{};
  a.raw = raw; a.ld = ld; a.noise = noise; a.z_vals = z_vals; a.rays_d = rays_d; a.rd_stride = rd_stride; a.N = N; a.S = S; a.white = white;
  a.weights = (float*)weights; a.acc_map = (float*)acc_map; a.depth_map = (float*)depth_map;
  a.g_rgb = g_rgb; a.g_disp = g_disp; a.g_acc = g_acc; a.g_depth = g_depth; a.g_w = g_w; a.d_raw = d_raw; a.ld_draw = ld_draw;
  hipLaunchKernelGGL(classic_composite_bwd_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
  return snerf_check_launch();
}
