// Early ray termination + sample compaction for inference (north star: "wavefront ballot / prefix-sum for early ray
// termination and sample compaction").  NOT part of the reference: an opt-in mode of the mip path whose error is bounded by the
// proposal histogram.  After the proposal level, every fine interval i of a ray has a predicted transmittance at its start,
// T_i = 1 - W(s1[i]), and a predicted weight m_i = W(s1[i+1]) - W(s1[i]), where W is the cumulative proposal weight
// (piecewise linear inside the proposal's intervals).  Intervals with T_i <= eps_t (behind the surface) or m_i <= eps_w (empty
// space) are not evaluated: the NeRF MLP runs on the compacted rows only and the compositing kernel treats the others as empty.
//   select  : wave per ray; wave scans build W, one ballot per 64 intervals gives the keep mask, popcount the row count
//   scan    : exclusive prefix sum of the per-ray counts (single workgroup)
//   assign  : wave per ray; row = offset[ray] + popcount(mask below the lane) -> row_index [N,S1] (-1 = skipped), sample_id [rows]
#include "common.h"

#define ERT_MAXP 512   // fence posts per level held in LDS per wave

struct ErtArgs {
  const float* s0; const float* w0; const float* s1;
  long N; int S0, S1; float eps_t, eps_w;
  unsigned long long* masks;   // [N, ceil(S1/64)]
  int* counts;                 // [N] -> exclusive offsets after the scan
  int* row_index; int* sample_id; long* total;
};

__global__ __launch_bounds__(256) void ert_select_kernel(ErtArgs a) {
  __shared__ float sp[4][ERT_MAXP + 1], cum[4][ERT_MAXP + 1], wc[4][ERT_MAXP + 1];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long ray = (long)blockIdx.x * 4 + wave;
  if (ray >= a.N) return;
  const int S0 = a.S0, S1 = a.S1;
  const float* s0 = a.s0 + ray * (S0 + 1);
  const float* w0 = a.w0 + ray * S0;
  const float* s1 = a.s1 + ray * (S1 + 1);
  // cumulative proposal weight at the proposal's fence posts
  float carry = 0.f;
  if (lane == 0) cum[wave][0] = 0.f;
  for (int base = 0; base < S0; base += 64) {
    const int k = base + lane;
    const float w = k < S0 ? w0[k] : 0.f;
    const float incl = wave_incl_scan_add(w, lane);
    if (k < S0) cum[wave][k + 1] = carry + incl;
    carry += __shfl(incl, 63, 64);
  }
  for (int k = lane; k <= S0; k += 64) sp[wave][k] = s0[k];
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_wave_barrier();
  // W at the fine fence posts
  for (int p = lane; p <= S1; p += 64) {
    const float s = s1[p];
    int lo = 0, hi = S0;                         // largest k in [0, S0-1] with sp[k] <= s
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (sp[wave][mid] <= s) lo = mid; else hi = mid;
    }
    const float a0 = sp[wave][lo], a1 = sp[wave][lo + 1];
    float fr = a1 > a0 ? (s - a0) / (a1 - a0) : 0.f;
    fr = fminf(fmaxf(fr, 0.f), 1.f);
    wc[wave][p] = cum[wave][lo] + fr * (cum[wave][lo + 1] - cum[wave][lo]);
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_wave_barrier();
  int count = 0;
  const int nchunk = (S1 + 63) >> 6;
  for (int c = 0; c < nchunk; ++c) {
    const int i = c * 64 + lane;
    bool keep = false;
    if (i < S1) {
      const float t = 1.f - wc[wave][i], m = wc[wave][i + 1] - wc[wave][i];
      keep = t > a.eps_t && m > a.eps_w;
    }
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(keep);
    if (lane == 0) a.masks[ray * nchunk + c] = mask;
    count += __builtin_popcountll(mask);
  }
  if (lane == 0) a.counts[ray] = count;
}

__global__ __launch_bounds__(1024) void ert_scan_kernel(int* counts, long N, long* total) {
  __shared__ long wsum[16];
  __shared__ long carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long base = 0; base < N; base += 1024) {
    const long r = base + threadIdx.x;
    const int v = r < N ? counts[r] : 0;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    long before = carry_s;
    for (int w = 0; w < wave; ++w) before += wsum[w];
    if (r < N) counts[r] = (int)(before + incl - v);        // exclusive offset
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = before + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry_s;
}

__global__ __launch_bounds__(256) void ert_assign_kernel(ErtArgs a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long ray = (long)blockIdx.x * 4 + wave;
  if (ray >= a.N) return;
  const int S1 = a.S1, nchunk = (S1 + 63) >> 6;
  int row = a.counts[ray];
  for (int c = 0; c < nchunk; ++c) {
    const unsigned long long mask = a.masks[ray * nchunk + c];
    const int i = c * 64 + lane;
    if (i < S1) {
      const bool keep = (mask >> lane) & 1ull;
      const int r = row + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
      a.row_index[ray * S1 + i] = keep ? r : -1;
      if (keep) a.sample_id[r] = (int)(ray * S1 + i);
    }
    row += __builtin_popcountll(mask);
  }
}

extern "C" int snerf_ert_compact(const float* s0, const float* w0, const float* s1, long N, int S0, int S1, float eps_t, float eps_w,
                                 void* masks, int* counts, int* row_index, int* sample_id, long* total, void* stream) {
  if (N <= 0) return SNERF_OK;
  if (S0 < 1 || S1 < 1 || S0 > ERT_MAXP || S1 > ERT_MAXP || N * (long)S1 >= (1L << 31)) return SNERF_ERR_ARG;
  if (s0 == nullptr || w0 == nullptr || s1 == nullptr || masks == nullptr || counts == nullptr || row_index == nullptr || sample_id == nullptr || total == nullptr)
    return SNERF_ERR_ARG;
  ErtArgs a{s0, w0, s1, N, S0, S1, eps_t, eps_w, (unsigned long long*)masks, counts, row_index, sample_id, total};
  const dim3 grid((unsigned)((N + 3) / 4));
  hipLaunchKernelGGL(ert_select_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
  hipLaunchKernelGGL(ert_scan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, counts, N, total);
  hipLaunchKernelGGL(ert_assign_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
  return snerf_check_launch();
}

// ------------------------------------------------------------------------------------------------------------------
// Front-to-back early ray termination on the FINE network's own densities (round 4).  The selection above trusts the proposal
// histogram, and on a fitted scene it finds little to skip: the fine samples are importance-sampled FROM that histogram, so almost
// all of them sit where it predicts weight (profiles/r4_i_ert_scene.txt: 94-95 % kept at eps = 1e-4).  What a converged scene does
// offer is opacity: once a ray's transmittance has fallen below eps_t, everything behind contributes at most eps_t to acc and rgb.
// The fine level is therefore evaluated in groups of G consecutive samples, front to back; after a group, one wave per ray adds the
// group's optical depth  sum softplus(raw + bias) (t1 - t0) |d|  (the compositing kernel's terms) to the ray's running tau, and rays
// with exp(-tau) <= eps_t leave: ballot-free here (one flag per ray), compaction by the same scan + popcount-free assign as above,
// since a surviving ray contributes a whole group of rows.  The bound is exact: the skipped samples' weights sum to <= exp(-tau).
//   step(prev group evaluated, next group to assign): update tau from the previous group's raw densities, flag the survivors,
//   exclusive scan of their row counts, write row_index / sample_id of the next group.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float ert_transform_s(float s, float near, float far, int idx) {       // composite.hip transform_s
  if (idx == 0) return near * expf(s * logf(far / near));
  if (idx == 1) return 1.f / ((1.f - s) / near + s / far);
  return near * (1.f - s) + far * s;
}
struct ErtF2B {
  const float* raw_d; long ld_den; long prev_base;       // raw densities of the previous group's rows (row_index - prev_base), or null
  const float* s1; const float* dirs; const float* near; const float* far;
  long N; int S1, g0, G, next_g0, next_G, tidx; float density_bias, eps_t;
  float* tau; int* flags; int* counts; int* row_index; int* sample_id; long row_base;
};
__global__ __launch_bounds__(256) void ert_f2b_update_kernel(ErtF2B a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long ray = (long)blockIdx.x * 4 + wave;
  if (ray >= a.N) return;
  float tau = a.tau[ray];
  if (a.raw_d != nullptr && a.G > 0) {
    const float near = a.near[ray], far = a.far[ray];
    const float dx = a.dirs[ray * 3], dy = a.dirs[ray * 3 + 1], dz = a.dirs[ray * 3 + 2];
    const float dnorm = sqrtf(dx * dx + dy * dy + dz * dz);
    const float* sv = a.s1 + ray * (a.S1 + 1);
    float part = 0.f;
    for (int k = lane; k < a.G; k += 64) {
      const int i = a.g0 + k;
      const int row = a.row_index[ray * a.S1 + i];
      if (row >= 0) {
        const float t0 = ert_transform_s(sv[i], near, far, a.tidx), t1 = ert_transform_s(sv[i + 1], near, far, a.tidx);
        const float x = a.raw_d[(row - a.prev_base) * a.ld_den] + a.density_bias;
        part += (x > 20.f ? x : log1pf(expf(x))) * ((t1 - t0) * dnorm);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
    tau += part;
    if (lane == 0) a.tau[ray] = tau;
  }
  if (lane == 0) {
    const int alive = expf(-tau) > a.eps_t ? 1 : 0;
    a.flags[ray] = alive;
    a.counts[ray] = alive ? a.next_G : 0;
  }
}
__global__ __launch_bounds__(256) void ert_f2b_assign_kernel(ErtF2B a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long ray = (long)blockIdx.x * 4 + wave;
  if (ray >= a.N) return;
  const int alive = a.flags[ray], off = a.counts[ray];     // (exclusive offset after the scan)
  for (int k = lane; k < a.next_G; k += 64) {
    const int i = a.next_g0 + k;
    a.row_index[ray * a.S1 + i] = alive ? (int)(a.row_base + off + k) : -1;
    if (alive) a.sample_id[off + k] = (int)(ray * a.S1 + i);
  }
}

extern "C" int snerf_ert_f2b_step(const float* prev_raw_d, long ld_den, long prev_base, const float* s1, const float* dirs, const float* near,
                                  const float* far, long N, int S1, int g0, int G, int next_g0, int next_G, int transform_idx, float density_bias,
                                  float eps_t, float* tau, int* flags, int* counts, int* row_index, int* sample_id, long row_base, long* total,
                                  void* stream) {
  if (N <= 0) return SNERF_OK;
  if (S1 < 1 || G < 0 || next_G < 1 || g0 < 0 || g0 + G > S1 || next_g0 < 0 || next_g0 + next_G > S1 || N * (long)S1 >= (1L << 31)) return SNERF_ERR_ARG;
  if (s1 == nullptr || dirs == nullptr || near == nullptr || far == nullptr || tau == nullptr || flags == nullptr || counts == nullptr ||
      row_index == nullptr || sample_id == nullptr || total == nullptr || (G > 0 && prev_raw_d == nullptr))
    return SNERF_ERR_ARG;
  ErtF2B a{prev_raw_d, ld_den, prev_base, s1, dirs, near, far, N, S1, g0, G, next_g0, next_G, transform_idx, density_bias, eps_t,
           tau, flags, counts, row_index, sample_id, row_base};
  const dim3 grid((unsigned)((N + 3) / 4));
  hipLaunchKernelGGL(ert_f2b_update_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
  hipLaunchKernelGGL(ert_scan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, counts, N, total);
  hipLaunchKernelGGL(ert_f2b_assign_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
  return snerf_check_launch();
}

// ------------------------------------------------------------------------------------------------------------------
// The same front-to-back termination for the CLASSIC path's fine pass (round 5; snerf_amd/classic.py _fine_pass_front_to_back): the
// fine network of render_rays evaluates the sorted union of the 64 uniform coarse positions and the 128 importance samples
// (render.py:380-389) -- the uniform ones behind the first surface are what can be skipped (35 % of the fine evaluations of a fitted
// street scene at eps 1e-4, tools/ert_classic_analysis.py).  One step after every evaluated group [g0, g0 + G) of the surviving rays:
//   update  wave per surviving ray: the group's raw outputs go to their places in raw_full [N, S, C] (unevaluated samples keep raw = 0:
//           alpha 0, weight 0), the group's optical depth  sum relu(sigma) (z[i+1] - z[i]) |d|  (raw2outputs, run_nerf_helpers.py:394-414)
//           multiplies the ray's transmittance, the ray stays if T > eps_t
//   scan    exclusive prefix sum of the keep flags (ert_scan_kernel)
//   assign  the survivors' ray ids, compacted in order -> the rows of the next group
// and the positions + view directions of a group's rows (classic_ert_points_kernel: render.py:354 on the compacted rows).
// ------------------------------------------------------------------------------------------------------------------
struct ClassicErt {
  const float* raw_g; int C;
  const int* alive; long n;
  const float* z; int S;
  const float* rays; long ld_rays;
  int g0, G; float eps_t;
  float* T; float* raw_full;
  int* keep; int* offs; int* alive_next;
};
__global__ __launch_bounds__(256) void classic_ert_update_kernel(ClassicErt a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long i = (long)blockIdx.x * 4 + wave;
  if (i >= a.n) return;
  const long ray = a.alive != nullptr ? (long)a.alive[i] : i;
  const float* rd = a.rays + ray * a.ld_rays + 3;
  const float dn = sqrtf(rd[0] * rd[0] + rd[1] * rd[1] + rd[2] * rd[2]);
  const float* zr = a.z + ray * a.S;
  float part = 0.f;
  for (int k = lane; k < a.G; k += 64) {
    const int s = a.g0 + k;
    const float* src = a.raw_g + (i * a.G + k) * a.C;
    float* dst = a.raw_full + (ray * a.S + s) * a.C;
    for (int c = 0; c < a.C; ++c) dst[c] = src[c];
    if (s + 1 < a.S) part += fmaxf(src[3], 0.f) * ((zr[s + 1] - zr[s]) * dn);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
  if (lane == 0) {
    const float T = a.T[ray] * expf(-part);
    a.T[ray] = T;
    const int keep = (a.g0 + a.G < a.S && T > a.eps_t) ? 1 : 0;
    a.keep[i] = keep;
  }
}
// compaction in three small launches instead of one workgroup walking all n flags (0.45 ms at n = 512 k): sums of 1024-flag blocks ->
// ert_scan_kernel over the <= n / 1024 sums -> block-local scan + ordered write
__global__ __launch_bounds__(256) void classic_ert_block_sums_kernel(const int* __restrict__ keep, long n, int* __restrict__ partial) {
  __shared__ int ws[4];
  const long e0 = (long)blockIdx.x * 1024 + threadIdx.x * 4;
  int c = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) c += (e0 + k < n) ? keep[e0 + k] : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
__global__ __launch_bounds__(256) void classic_ert_assign_kernel(ClassicErt a) {
  __shared__ int ws[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long e0 = (long)blockIdx.x * 1024 + threadIdx.x * 4;
  int k4[4], c = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) { k4[k] = (e0 + k < a.n) ? a.keep[e0 + k] : 0; c += k4[k]; }
  int incl = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
  if (lane == 63) ws[wave] = incl;
  __syncthreads();
  int base = a.offs[blockIdx.x] + incl - c;              // (offs: the exclusive scan of the block sums)
  for (int w = 0; w < wave; ++w) base += ws[w];
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (k4[k]) a.alive_next[base++] = a.alive != nullptr ? a.alive[e0 + k] : (int)(e0 + k);
}
extern "C" int snerf_classic_ert_step(const float* raw_g, int C, const int* alive, long n, const float* z_all, int S, const float* rays, long ld_rays,
                                      int g0, int G, float eps_t, float* T, float* raw_full, int* keep, int* offs, int* alive_next, long* total,
                                      void* stream) {
  if (n <= 0) return SNERF_OK;
  if (raw_g == nullptr || C < 4 || z_all == nullptr || S < 1 || rays == nullptr || ld_rays < 6 || g0 < 0 || G < 1 || g0 + G > S || T == nullptr ||
      raw_full == nullptr || keep == nullptr || offs == nullptr || alive_next == nullptr || total == nullptr || n >= (1L << 31))
    return SNERF_ERR_ARG;
  ClassicErt a{raw_g, C, alive, n, z_all, S, rays, ld_rays, g0, G, eps_t, T, raw_full, keep, offs, alive_next};
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(classic_ert_update_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, a);
  const long nb = (n + 1023) / 1024;
  hipLaunchKernelGGL(classic_ert_block_sums_kernel, dim3((unsigned)nb), dim3(256), 0, s, keep, n, offs);
  hipLaunchKernelGGL(ert_scan_kernel, dim3(1), dim3(1024), 0, s, offs, nb, total);
  hipLaunchKernelGGL(classic_ert_assign_kernel, dim3((unsigned)nb), dim3(256), 0, s, a);
  return snerf_check_launch();
}

__global__ __launch_bounds__(256) void classic_ert_points_kernel(const float* __restrict__ rays, long ld_rays, const float* __restrict__ z, int S,
                                                                 const int* __restrict__ alive, long n, int g0, int G, float* __restrict__ pts,
                                                                 float* __restrict__ vd, int vd_col) {
#pragma clang fp contract(off)        // o + d * z with the product rounded on its own, like classic_points_kernel (sampler.hip) and the eager ops
  const long total = n * G * 3;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long m = e / 3;
    const int c = (int)(e - m * 3);
    const long i = m / G;
    const int k = (int)(m - i * G);
    const long ray = alive != nullptr ? (long)alive[i] : i;
    const float prod = rays[ray * ld_rays + 3 + c] * z[ray * S + g0 + k];
    pts[e] = rays[ray * ld_rays + c] + prod;
    if (vd != nullptr && k == 0) vd[i * 3 + c] = rays[ray * ld_rays + vd_col + c];
  }
}
extern "C" int snerf_classic_ert_points(const float* rays, long ld_rays, const float* z_all, int S, const int* alive, long n, int g0, int G,
                                        float* pts, float* viewdirs, int viewdir_col, void* stream) {
  if (n <= 0) return SNERF_OK;
  if (rays == nullptr || ld_rays < 6 || z_all == nullptr || S < 1 || g0 < 0 || G < 1 || g0 + G > S || pts == nullptr ||
      (viewdirs != nullptr && (viewdir_col < 0 || viewdir_col + 3 > ld_rays)))
    return SNERF_ERR_ARG;
  const long total = n * G * 3;
  const int blocks = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
  hipLaunchKernelGGL(classic_ert_points_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, rays, ld_rays, z_all, S, alive, n, g0, G, pts, viewdirs, viewdir_col);
  return snerf_check_launch();
}
