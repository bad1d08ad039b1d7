// The two steps either side of the render path: ray generation (before) and the per-ray loss tail
// (after).  Compiled with -ffp-contract=off: every operation rounds like the reference's separate fp32 torch ops.
#include "common.h"

// ---------------------------------------------------------------------------
// Pinhole rays.  s-nerf/utils/sample_utils.py:286-345 (get_rays_single_img: whole frame, half-pixel centres) and :92-211
// (sample_single_img: selected pixels; directions from run_nerf_helpers.get_rays_by_coord :300-312, i.e. WITHOUT the
// half-pixel offset, radii still from the half-pixel grid), no-NDC branch.  One lane per ray.
// ---------------------------------------------------------------------------
struct RayGen {
  const int* coords;      // [N,2] (row, col) or null: pixel index first + n, row-major over W
  long first;
  int W, H, training;
  float p[12];            // pose [3,4] row-major
  float cx, cy, fx, fy, near, far;
  long N;
  float *origins, *directions, *viewdirs, *radii, *near_out, *far_out;
};

__device__ __forceinline__ void cam_dir(const RayGen& a, float jj, float ii, float f, float off, float* d) {
  const float c0 = (ii - a.cx + off) / f, c1 = -((jj - a.cy + off) / f), c2 = -1.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) d[c] = (c0 * a.p[4 * c + 0] + c1 * a.p[4 * c + 1]) + c2 * a.p[4 * c + 2];
}

__global__ __launch_bounds__(256) void pinhole_rays_kernel(RayGen a) {
  const long r = (long)blockIdx.x * 256 + threadIdx.x;
  if (r >= a.N) return;
  int row, col;
  if (a.coords != nullptr) { row = a.coords[2 * r]; col = a.coords[2 * r + 1]; }
  else { const long pix = a.first + r; row = (int)(pix / a.W); col = (int)(pix - (long)row * a.W); }
  const float f = (a.fx + a.fy) / 2.f;
  const float jj = (float)row, ii = (float)col;
  float d[3];
  if (a.training) {
    // get_rays_by_coord: i = (col - cx) / focal, j = -(row - cy) / focal
    const float c0 = (ii - a.cx) / f, c1 = -((jj - a.cy) / f);
#pragma unroll
    for (int c = 0; c < 3; ++c) d[c] = (c0 * a.p[4 * c + 0] + c1 * a.p[4 * c + 1]) + (-1.f) * a.p[4 * c + 2];
  } else {
    cam_dir(a, jj, ii, f, 0.5f, d);
  }
  // radius: distance to the neighbour one row down on the half-pixel grid; the reference appends dx[-2:-1] for the last image
  // row, i.e. the value of row H-3 (sample_utils.py:309-310), not that of row H-2
  const float ja = row >= a.H - 1 ? (float)(a.H - 3) : jj;
  float d0[3], d1[3];
  cam_dir(a, ja, ii, f, 0.5f, d0);
  cam_dir(a, ja + 1.f, ii, f, 0.5f, d1);
  const float e0 = d0[0] - d1[0], e1 = d0[1] - d1[1], e2 = d0[2] - d1[2];
  const float dx = sqrtf((e0 * e0 + e1 * e1) + e2 * e2);
  const float nrm = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    a.origins[3 * r + c] = a.p[4 * c + 3];
    a.directions[3 * r + c] = d[c];
    a.viewdirs[3 * r + c] = d[c] / nrm;
  }
  a.radii[r] = dx * 2.f / 3.4641016151377544f;          // dx[..., None] * 2 / np.sqrt(12)
  a.near_out[r] = a.near;
  a.far_out[r] = a.far;
}

extern "C" int snerf_pinhole_rays(const int* coords, long first_pixel, int W, int H, const float* pose_host, float cx, float cy, float fx,
                                  float fy, int training, float near, float far, long N, float* origins, float* directions,
                                  float* viewdirs, float* radii, float* near_out, float* far_out, void* stream) {
  if (N <= 0) return SNERF_OK;
  if (pose_host == nullptr || W <= 0 || H < 3) return SNERF_ERR_ARG;
  RayGen a{coords, first_pixel, W, H, training, {}, cx, cy, fx, fy, near, far, N, origins, directions, viewdirs, radii, near_out, far_out};
  for (int k = 0; k < 12; ++k) a.p[k] = pose_host[k];
  hipLaunchKernelGGL(pinhole_rays_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return snerf_check_launch();
}

// ---------------------------------------------------------------------------
// Per-ray loss tail of the mip path (s-nerf/train.py:150-208): RgbLoss (loss_factory.py:5-11), calc_depth_loss with DepthLoss
// and per-ray confidence (confidence.py:209-224, loss_factory.py:26-37) and ProposalLoss (loss_factory.py:59-74), forward
// value AND the gradients w.r.t. the renderer outputs in one pass.  One lane per ray; prefix sums in the canonical order
// (fp64 accumulate, one rounding per emitted fp32 value), which is what torch.cumsum does on the CPU.
// out[0..3] = {#valid depth rays, rgb loss, depth loss (x depth_lambda), proposal loss (x proposal_lambda)}.
// ---------------------------------------------------------------------------
struct LossTail {
  const float *rgb, *tgt, *dist1, *dist0, *tdepth, *conf;
  const float *s_f, *w_f, *s_c, *w_c;
  long N;
  int Pf, Sc, disparity;
  float depth_lambda, coarse_mult, prop_lambda;
  float* out;
  float *g_rgb, *g_dist1, *g_dist0, *g_wc;
};

__global__ __launch_bounds__(1024) void loss_prepare_kernel(const float* __restrict__ tdepth, long N, float* __restrict__ out) {
  __shared__ int part[16];
  int cnt = 0;
  if (tdepth != nullptr)
    for (long r = threadIdx.x; r < N; r += 1024) cnt += tdepth[r] != 0.f;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < 16; ++w) t += part[w];
    out[0] = (float)t; out[1] = 0.f; out[2] = 0.f; out[3] = 0.f;
  }
}

__global__ __launch_bounds__(64) void loss_tail_kernel(LossTail a) {
  const long r = (long)blockIdx.x * 64 + threadIdx.x;
  float l_rgb = 0.f, l_dep = 0.f, l_prop = 0.f;
  if (r < a.N) {
    // ---- RGB: mean over N*3 of (pred - tgt)^2
    const float inv = 1.f / (3.f * (float)a.N);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float d = a.rgb[3 * r + c] - a.tgt[3 * r + c];
      l_rgb += d * d;
      a.g_rgb[3 * r + c] = d * (2.f * inv);
    }
    l_rgb *= inv;
    // ---- depth: masked mean over the rays with a LiDAR target
    if (a.tdepth != nullptr) {
      const float t = a.tdepth[r];
      float g1 = 0.f, g0 = 0.f;
      if (t != 0.f) {
        const float nv = a.out[0];
        const float cw = (a.conf != nullptr ? a.conf[r] : 1.f);
        const float d1 = a.dist1[r], d0 = a.dist0[r];
        float e1, e0, s1, s0;
        if (a.disparity) { e1 = 1.f / d1 - 1.f / t; e0 = 1.f / d0 - 1.f / t; s1 = -1.f / (d1 * d1); s0 = -1.f / (d0 * d0); }
        else { e1 = d1 - t; e0 = d0 - t; s1 = 1.f; s0 = 1.f; }
        const float k = cw * a.depth_lambda / nv;
        l_dep = (fabsf(e1) + a.coarse_mult * fabsf(e0)) * k;
        g1 = (e1 > 0.f ? 1.f : (e1 < 0.f ? -1.f : 0.f)) * s1 * k;
        g0 = (e0 > 0.f ? 1.f : (e0 < 0.f ? -1.f : 0.f)) * s0 * k * a.coarse_mult;
      }
      a.g_dist1[r] = g1;
      a.g_dist0[r] = g0;
    }
    // ---- proposal: the coarse histogram must bound the (detached) fine weights from above
    if (a.s_c != nullptr) {
      const float* sf = a.s_f + r * a.Pf;
      const float* wf = a.w_f + r * (a.Pf - 1);
      const float* sc = a.s_c + r * (a.Sc + 1);
      const float* wc = a.w_c + r * a.Sc;
      float* g = a.g_wc + r * a.Sc;
      for (int j = 0; j < a.Sc; ++j) g[j] = 0.f;
      const int cap = min(a.Pf - 2, a.Sc - 1);          // the reference clamps `right` with the FINE interval count
      const float scale = a.prop_lambda / (float)a.N;
      // walk both sorted fence-post lists once; p = #(s_c <= s_f[k]) (searchsorted right=True); C = cumsum(w_c)
      int p = 0;
      double cum = 0.0;
      float Wlast = 0.f;                                 // C[min(p-1, Sc-1)]
      float Wcap = 0.f;                                  // C[cap], known once p-1 >= cap
      const float W0 = wc[0];                            // C[0]: the reference clamps the left index at 0 (not "empty prefix")
      float left = 0.f;
      int li = 0;
      for (int k = 0; k < a.Pf; ++k) {
        const float s = sf[k];
        while (p <= a.Sc && sc[p] <= s) {
          if (p < a.Sc) { cum += (double)wc[p]; Wlast = (float)cum; if (p == cap) Wcap = Wlast; }
          ++p;
        }
        const int idx = p - 1;                           // inds - 1 of this fence post
        if (k > 0) {
          const int ri = idx > cap ? cap : max(idx, 0);
          const float right = idx > cap ? Wcap : (idx <= 0 ? W0 : Wlast);
          const float w = wf[k - 1];
          const float over = w - (right - left);
          if (over > 0.f) {
            l_prop += over * over / (w + 1e-8f);
            const float gb = -2.f * over / (w + 1e-8f) * scale;
            g[ri] += gb;
            g[li] -= gb;
          }
        }
        li = idx <= 0 ? 0 : min(idx, a.Sc - 1);          // (an index past the last interval raises in the reference)
        left = idx <= 0 ? W0 : Wlast;
      }
      l_prop *= scale;
      // d/dw_c[j] = sum_{k >= j} dW[k]
      double acc = 0.0;
      for (int j = a.Sc - 1; j >= 0; --j) { acc += (double)g[j]; g[j] = (float)acc; }
    }
  }
  l_rgb = wave_sum(l_rgb); l_dep = wave_sum(l_dep); l_prop = wave_sum(l_prop);
  if (threadIdx.x == 0) {
    atomicAdd(a.out + 1, l_rgb);
    if (a.tdepth != nullptr) atomicAdd(a.out + 2, l_dep);
    if (a.s_c != nullptr) atomicAdd(a.out + 3, l_prop);
  }
}

extern "C" int snerf_mip_loss_tail(const float* rgb, const float* tgt, const float* dist1, const float* dist0, const float* tdepth,
                                   const float* conf, const float* s_f, const float* w_f, const float* s_c, const float* w_c, long N,
                                   int Pf, int Sc, int disparity, float depth_lambda, float coarse_mult, float prop_lambda, float* out,
                                   float* g_rgb, float* g_dist1, float* g_dist0, float* g_wc, void* stream) {
  if (N <= 0) return SNERF_OK;
  if (rgb == nullptr || tgt == nullptr || out == nullptr || g_rgb == nullptr) return SNERF_ERR_ARG;
  if (tdepth != nullptr && (dist1 == nullptr || dist0 == nullptr || g_dist1 == nullptr || g_dist0 == nullptr)) return SNERF_ERR_ARG;
  if (s_c != nullptr && (s_f == nullptr || w_f == nullptr || w_c == nullptr || g_wc == nullptr || Pf < 2 || Sc < 1)) return SNERF_ERR_ARG;
  hipLaunchKernelGGL(loss_prepare_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, tdepth, N, out);
  LossTail a{rgb, tgt, dist1, dist0, tdepth, conf, s_f, w_f, s_c, w_c, N, Pf, Sc, disparity, depth_lambda, coarse_mult, prop_lambda, out,
             g_rgb, g_dist1, g_dist0, g_wc};
  hipLaunchKernelGGL(loss_tail_kernel, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, (hipStream_t)stream, a);
  return snerf_check_launch();
}
