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
// Classic-path ray front end: get_rays (s-nerf/model/run_nerf_helpers.py:247-258), ndc_rays (:314-332) and the
// ray-batch assembly of render() (s-nerf/model/render.py:50-77) -- pinhole directions or given rays, unit view directions from
// the PRE-NDC directions, the optional static camera that replaces the rays but not the view directions, the NDC warp, and the
// row layout [o3, d3, near, far, (depth), (viewdir3)] -- as ONE lane-per-ray launch writing the [N, ld] rows render_rays reads.
// Every operation rounds like the reference's separate fp32 torch ops (-ffp-contract=off, explicit association).
// ---------------------------------------------------------------------------
struct ClassicRays {
  const float *rays_o, *rays_d;   // given rays [n,3] (contiguous) or null: pinhole rays of the H x W frame from c2w
  long n;
  int H, W;
  float focal, cx, cy;            // fp32 roundings of the reference's python scalars
  float c2w[12], c2w_static[12];  // [3,4] row-major
  int has_c2w, has_static, ndc, use_viewdirs;
  float ndc_near, near, far;
  float ndc_cw, ndc_ch;           // -1/(W/(2 focal)), -1/(H/(2 focal)) computed in double on the host like the python scalars
  const float* depths;            // optional extra column
  float* rows; int ld;            // ray batch rows, or null
  float *o_out, *d_out;           // separate [n,3] outputs (get_rays / ndc_rays), or null
};

__device__ __forceinline__ void classic_pinhole(const float* c2w, int row, int col, float focal, float cx, float cy, float* o, float* d) {
  // dirs = [((i + 0.5) - cx) / focal, -((j + 0.5) - cy) / focal, -1];  rays_d = sum(dirs[None, :] * c2w[:3, :3], -1)
  const float c0 = (((float)col + 0.5f) - cx) / focal, c1 = -((((float)row + 0.5f) - cy) / focal), c2 = -1.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    d[c] = (c0 * c2w[4 * c + 0] + c1 * c2w[4 * c + 1]) + c2 * c2w[4 * c + 2];
    o[c] = c2w[4 * c + 3];
  }
}

__device__ __forceinline__ void classic_ndc(const ClassicRays& a, float* o, float* d) {
  const float t = -(a.ndc_near + o[2]) / d[2];
#pragma unroll
  for (int c = 0; c < 3; ++c) o[c] = o[c] + t * d[c];
  const float o0 = a.ndc_cw * o[0] / o[2], o1 = a.ndc_ch * o[1] / o[2], o2 = 1.f + (2.f * a.ndc_near) / o[2];
  const float d0 = a.ndc_cw * (d[0] / d[2] - o[0] / o[2]), d1 = a.ndc_ch * (d[1] / d[2] - o[1] / o[2]), d2 = (-2.f * a.ndc_near) / o[2];
  o[0] = o0; o[1] = o1; o[2] = o2;
  d[0] = d0; d[1] = d1; d[2] = d2;
}

__global__ __launch_bounds__(256) void classic_rays_kernel(ClassicRays a) {
  const long r = (long)blockIdx.x * 256 + threadIdx.x;
  if (r >= a.n) return;
  const int row = (int)(r / a.W), col = (int)(r - (long)row * a.W);
  float o[3], d[3], v[3];
  if (a.has_c2w) classic_pinhole(a.c2w, row, col, a.focal, a.cx, a.cy, o, d);
  else {
#pragma unroll
    for (int c = 0; c < 3; ++c) { o[c] = a.rays_o[3 * r + c]; d[c] = a.rays_d[3 * r + c]; }
  }
  if (a.use_viewdirs) {
    const float nrm = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = d[c] / nrm;
    // c2w_staticcam: the rays come from the static camera (principal point at the image centre: render.py:59 passes no
    // ori_points), the view directions stay those of c2w
    if (a.has_static) classic_pinhole(a.c2w_static, row, col, a.focal, (float)(a.W * 0.5), (float)(a.H * 0.5), o, d);
  }
  if (a.ndc) classic_ndc(a, o, d);
  if (a.o_out != nullptr) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { a.o_out[3 * r + c] = o[c]; a.d_out[3 * r + c] = d[c]; }
  }
  if (a.rows != nullptr) {
    float* q = a.rows + r * a.ld;
    q[0] = o[0]; q[1] = o[1]; q[2] = o[2]; q[3] = d[0]; q[4] = d[1]; q[5] = d[2];
    q[6] = a.near; q[7] = a.far;                       // near * ones_like(d[..., :1]): exact
    int k = 8;
    if (a.depths != nullptr) q[k++] = a.depths[r];
    if (a.use_viewdirs) { q[k] = v[0]; q[k + 1] = v[1]; q[k + 2] = v[2]; }
  }
}

static int classic_rays_launch(ClassicRays& a, void* stream) {
  if (a.n <= 0) return SNERF_OK;
  hipLaunchKernelGGL(classic_rays_kernel, dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return snerf_check_launch();
}

extern "C" int snerf_classic_get_rays(int H, int W, double focal, double cx, double cy, const float* c2w_host, float* rays_o, float* rays_d,
                                      void* stream) {
  if (H <= 0 || W <= 0 || c2w_host == nullptr || rays_o == nullptr || rays_d == nullptr) return SNERF_ERR_ARG;
  ClassicRays a{};
  a.n = (long)H * W; a.H = H; a.W = W; a.focal = (float)focal; a.cx = (float)cx; a.cy = (float)cy; a.has_c2w = 1;
  for (int k = 0; k < 12; ++k) a.c2w[k] = c2w_host[k];
  a.o_out = rays_o; a.d_out = rays_d;
  return classic_rays_launch(a, stream);
}

extern "C" int snerf_classic_ndc_rays(int H, int W, double focal, float near, const float* rays_o, const float* rays_d, long n, float* o_out,
                                      float* d_out, void* stream) {
  if (H <= 0 || W <= 0 || rays_o == nullptr || rays_d == nullptr || o_out == nullptr || d_out == nullptr) return SNERF_ERR_ARG;
  ClassicRays a{};
  a.n = n; a.H = H; a.W = W; a.rays_o = rays_o; a.rays_d = rays_d; a.ndc = 1; a.ndc_near = near;
  a.ndc_cw = (float)(-1.0 / ((double)W / (2.0 * focal))); a.ndc_ch = (float)(-1.0 / ((double)H / (2.0 * focal)));
  a.o_out = o_out; a.d_out = d_out;
  return classic_rays_launch(a, stream);
}

extern "C" int snerf_classic_ray_batch(int H, int W, double focal, double cx, double cy, const float* c2w_host, const float* c2w_static_host,
                                       const float* rays_o, const float* rays_d, long n, int ndc, float near, float far,
                                       const float* depths, int use_viewdirs, float* rows, int ld, void* stream) {
  const int need = 8 + (depths != nullptr ? 1 : 0) + (use_viewdirs ? 3 : 0);
  if (H <= 0 || W <= 0 || rows == nullptr || ld < need) return SNERF_ERR_ARG;
  if (c2w_host == nullptr && (rays_o == nullptr || rays_d == nullptr)) return SNERF_ERR_ARG;
  if (c2w_host != nullptr && n != (long)H * W) return SNERF_ERR_ARG;
  if (c2w_static_host != nullptr && n != (long)H * W) return SNERF_ERR_ARG;
  ClassicRays a{};
  a.n = n; a.H = H; a.W = W; a.focal = (float)focal; a.cx = (float)cx; a.cy = (float)cy;
  a.rays_o = rays_o; a.rays_d = rays_d;
  if (c2w_host != nullptr) { a.has_c2w = 1; for (int k = 0; k < 12; ++k) a.c2w[k] = c2w_host[k]; }
  if (c2w_static_host != nullptr) { a.has_static = 1; for (int k = 0; k < 12; ++k) a.c2w_static[k] = c2w_static_host[k]; }
  a.ndc = ndc; a.ndc_near = 1.f;                       // render.py:66 passes near = 1. to ndc_rays
  a.ndc_cw = (float)(-1.0 / ((double)W / (2.0 * focal))); a.ndc_ch = (float)(-1.0 / ((double)H / (2.0 * focal)));
  a.near = near; a.far = far; a.depths = depths; a.use_viewdirs = use_viewdirs; a.rows = rows; a.ld = ld;
  return classic_rays_launch(a, stream);
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
    out[0] = (float)t; out[1] = 0.f; out[2] = 0.f; out[3] = 0.f; out[4] = 0.f;
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
    atomicAdd(a.out + 4, (l_rgb + l_dep) + l_prop);     // the total: the step's loss scalar without a fold launch behind the tail
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

// ---------------------------------------------------------------------------
// zipnerf (path C) ray generation: s-nerfpp/zipnerf/internal/camera_utils.py:453-563 (pixels_to_rays), perspective camera, no
// distortion, no NDC.  numpy promotes to float64 there (integer pixels + .5); the dataset casts the result to fp32.  One lane
// per ray, float64 arithmetic, one rounding per emitted value.
// ---------------------------------------------------------------------------
struct ZipRayGen {
  const int *pix_x, *pix_y, *cam_idx;
  const float *pixtocams, *camtoworlds;          // [ncam,3,3], [ncam,3,4]
  long N;
  int ncam;
  float *origins, *directions, *viewdirs, *radii, *imageplane, *base_x, *base_y;
};

__device__ __forceinline__ void zip_cast(const double* k, const double* p, double x, double y, double* cam, double* d) {
  const double px = x + .5, py = y + .5;
#pragma unroll
  for (int c = 0; c < 3; ++c) cam[c] = (k[3 * c] * px + k[3 * c + 1] * py) + k[3 * c + 2];
  cam[1] = -cam[1]; cam[2] = -cam[2];                 // OpenCV -> OpenGL (:527)
#pragma unroll
  for (int c = 0; c < 3; ++c) d[c] = (p[4 * c] * cam[0] + p[4 * c + 1] * cam[1]) + p[4 * c + 2] * cam[2];
}

__global__ __launch_bounds__(256) void zip_rays_kernel(ZipRayGen a) {
  const long r = (long)blockIdx.x * 256 + threadIdx.x;
  if (r >= a.N) return;
  int cam = a.cam_idx != nullptr ? a.cam_idx[r] : 0;
  cam = min(max(cam, 0), a.ncam - 1);
  double k[9], p[12];
#pragma unroll
  for (int i = 0; i < 9; ++i) k[i] = (double)a.pixtocams[9 * cam + i];
#pragma unroll
  for (int i = 0; i < 12; ++i) p[i] = (double)a.camtoworlds[12 * cam + i];
  const double x = (double)a.pix_x[r], y = (double)a.pix_y[r];
  double c0[3], cx[3], cy[3], d[3], dx[3], dy[3];
  zip_cast(k, p, x, y, c0, d);
  zip_cast(k, p, x + 1.0, y, cx, dx);
  zip_cast(k, p, x, y + 1.0, cy, dy);
  double nd = 0, nx = 0, ny = 0;
#pragma unroll
  for (int c = 0; c < 3; ++c) { dx[c] -= d[c]; dy[c] -= d[c]; nd += d[c] * d[c]; nx += dx[c] * dx[c]; ny += dy[c] * dy[c]; }
  nd = sqrt(nd); nx = sqrt(nx); ny = sqrt(ny);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    a.origins[3 * r + c] = (float)p[4 * c + 3];
    a.directions[3 * r + c] = (float)d[c];
    a.viewdirs[3 * r + c] = (float)(d[c] / nd);
    a.base_x[3 * r + c] = (float)(dx[c] / nx);
    a.base_y[3 * r + c] = (float)(dy[c] / ny);
  }
  a.radii[r] = (float)((0.5 * (nx + ny)) * 2.0 / 3.4641016151377544);
  if (a.imageplane != nullptr) { a.imageplane[2 * r] = (float)c0[0]; a.imageplane[2 * r + 1] = (float)c0[1]; }
}

extern "C" int snerf_zip_pixels_to_rays(const int* pix_x, const int* pix_y, const int* cam_idx, const float* pixtocams, const float* camtoworlds,
                                        int ncam, long N, float* origins, float* directions, float* viewdirs, float* radii, float* imageplane,
                                        float* base_x, float* base_y, void* stream) {
  if (N <= 0) return SNERF_OK;
  if (pix_x == nullptr || pix_y == nullptr || pixtocams == nullptr || camtoworlds == nullptr || ncam < 1 || origins == nullptr ||
      directions == nullptr || viewdirs == nullptr || radii == nullptr || base_x == nullptr || base_y == nullptr)
    return SNERF_ERR_ARG;
  ZipRayGen a{pix_x, pix_y, cam_idx, pixtocams, camtoworlds, N, ncam, origins, directions, viewdirs, radii, imageplane, base_x, base_y};
  hipLaunchKernelGGL(zip_rays_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return snerf_check_launch();
}

// ---------------------------------------------------------------------------
// Per-ray loss tail of the zipnerf training step (s-nerfpp/zipnerf/train.py:250-311), value AND gradients w.r.t. the renderer
// outputs in one pass:
//   data        train_utils.py:62-90   sum(lossmult * sqrt(resid^2 + pad^2)) / sum(lossmult)   ('charb'; 'mse' = resid^2)
//   depth       train.py:252-255,277   depth_lambda * masked mean |1/(depth+1e-5) - 1/(1e-5+target)|
//   d_complete  train.py:260-272       the same under the second mask, x 0.2
//   sem         train.py:294-298       -sem_mult * masked mean log(semantic[r, label] + 1e-6)
//   interlevel  train_utils.py:132-164 (anti_interlevel_loss): NeRF histogram blurred by the level's pulse width
//               (stepfun.py:425-433), integrated to a piecewise-quadratic CDF, resampled on the proposal intervals
//               (math.py:133-156); mean(clamp(w_s - wp, 0)^2 / (wp + 1e-5)); gradient reaches the proposal weights only
//   distortion  stepfun.py:297-307     mean_r [sum_ij w_i w_j |u_i - u_j| + sum_i w_i^2 d_i / 3]; gradient w.r.t. the NeRF weights
// grid = (ceil(R/64), 3): y = 0 per-ray terms + distortion, y = 1, 2 the two proposal levels.  One lane per ray.  The blurred
// histogram is never materialised: its 2(S+1) knots are the merge of the two sorted lists {c - r} and {c + r}, walked once
// together with the (sorted) proposal fence posts; prefix sums run in float64 (the reference's fp32 cumsum of the +/- steps
// carries ~1e-4 relative noise, tests/test_oracle_callers_golden.py) and every emitted value is rounded once.
// out[0..3] = {3 sum lossmult, sum depth mask, sum complete mask, sum semantic mask};
// out[4..10] = {data, mse, depth, d_complete, sem, interlevel, distortion} (already multiplied by their weights).
// ---------------------------------------------------------------------------
struct ZipLoss {
  const float *rgb, *tgt, *lossmult;
  const float *depth, *tdepth, *dmask, *cmask;
  const float* sem; const int* labels; const float* smask;
  const float *s0, *w0, *s1, *w1, *s2, *w2;
  long R;
  int C, S0, S1, S2, mse;
  float pad, data_mult, depth_lambda, com_mult, sem_mult, pw0, pw1, inter_mult, dist_mult;
  float* out;
  float *g_rgb, *g_depth, *g_sem, *g_w0, *g_w1, *g_w2;
};

__global__ __launch_bounds__(1024) void zip_loss_prepare_kernel(ZipLoss a) {
  __shared__ float part[4][16];
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  for (long r = threadIdx.x; r < a.R; r += 1024) {
    s[0] += a.lossmult != nullptr ? a.lossmult[r] : 1.f;
    if (a.depth != nullptr && a.dmask != nullptr) s[1] += a.dmask[r];
    if (a.depth != nullptr && a.cmask != nullptr) s[2] += a.cmask[r];
    if (a.sem != nullptr) s[3] += a.smask != nullptr ? a.smask[r] : 1.f;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    s[k] = wave_sum(s[k]);
    if ((threadIdx.x & 63) == 0) part[k][threadIdx.x >> 6] = s[k];
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += part[threadIdx.x][w];
    a.out[threadIdx.x] = threadIdx.x == 0 ? 3.f * t : t;
  }
  if (threadIdx.x >= 4 && threadIdx.x < 11) a.out[threadIdx.x] = 0.f;
}

// anti-interlevel term of one ray against one proposal level -> sum_j clamp(w_s - wp, 0)^2 / (wp + 1e-5); writes d/d wp
// (`g` may be `wp` itself: g[j] is written after the last read of wp[j] -- the LDS-staged caller overwrites the weights in place)
__device__ __forceinline__ float zip_interlevel_ray(const float* __restrict__ c, const float* __restrict__ w, int S, const float* __restrict__ cp,
                                                    const float* wp, int Sp, float r, float gscale, float* g) {
  const int n = S + 1;
  const double r2 = 2.0 * (double)r;
  // y1[k] = (wn[k] - wn[k-1]) / (2r) with wn = w / (c[k+1] - c[k]) and zeros outside (stepfun.py:427-428); each of the two knot streams
  // visits k = 0 .. S in order, so wn[k-1] is the value its previous visit computed (`last`): one division per knot
  auto y1 = [&](int k, double& last) __attribute__((always_inline)) {
    const double hi = k < S ? (double)w[k] / ((double)c[k + 1] - (double)c[k]) : 0.0;
    const double v = (hi - last) / r2;
    last = hi;
    return v;
  };
  double wn_a = 0.0, wn_b = 0.0;
  int ia = 0, ib = 0, q = 0;
  float xa = c[0] - r, xb = c[0] + r;
  double inner = 0.0, yrun = 0.0, cdf = 0.0;
  double xprev = 0.0, yprev = 0.0, cprev = 0.0;
  double ci_prev = 0.0;
  double loss = 0.0;
  auto emit = [&](double ci) __attribute__((always_inline)) {
    if (q > 0) {
      const double ws = ci - ci_prev;
      const double p = (double)wp[q - 1];
      const double over = ws > p ? ws - p : 0.0;
      const double den = p + 1e-5;
      loss += over * over / den;
      g[q - 1] = (float)((double)gscale * (-2.0 * over / den - over * over / (den * den)));
    }
    ci_prev = ci;
    ++q;
  };
  float xq = cp[0];
  for (int k = 0; k < 2 * n; ++k) {
    double xk, val;
    if (ib >= n || (ia < n && xa <= xb)) { xk = (double)xa; val = y1(ia, wn_a); ++ia; xa = ia < n ? c[ia] - r : 0.f; }
    else { xk = (double)xb; val = -y1(ib, wn_b); ++ib; xb = ib < n ? c[ib] + r : 0.f; }
    double yk = 0.0;
    if (k > 0) {
      const double dx = xk - xprev;
      yrun += dx * inner;
      yk = yrun > 0.0 ? yrun : 0.0;
      cdf += 0.5 * (yk + yprev) * dx;
    }
    // proposal fence posts in [x_{k-1}, x_k): idx = k knots are <= x (math.py:141-148)
    while (q <= Sp && (double)xq < xk) {
      double ci = 0.0;
      if (k > 0) {
        const double t = (double)xq - xprev;
        double off = t / (xk - xprev);
        off = off < 0.0 ? 0.0 : (off > 1.0 ? 1.0 : off);
        ci = cprev + t * (yprev + yk * off + yprev * (1.0 - off)) / 2.0;
      }
      emit(ci);
      xq = q <= Sp ? cp[q] : 0.f;
    }
    xprev = xk; yprev = yk; cprev = cdf;
    inner += val;
  }
  while (q <= Sp) {                                   // at or past the last knot: offset = 1 (or nan -> 0 when equal), both give this
    emit(cprev + ((double)xq - xprev) * yprev);
    xq = q <= Sp ? cp[q] : 0.f;
  }
  return (float)loss;
}

// The anti-interlevel term with a workgroup's 64 rays staged in LDS: the per-lane walks over (c, w, cp, wp) are 130 dependent steps of
// row-strided reads -- from global memory every one of them is 64 cache lines per wave instruction (573 us for 65 536 rays x 2 levels);
// staged through coalesced loads into odd-strided LDS rows (conflict-free for lane = ray) they are LDS latency.  The gradient overwrites
// the staged proposal weights in place and leaves through coalesced stores.  grid = (ceil(R / 64), 2 levels); same arithmetic, same
// order, same results as the global-memory walk (which stays for interval counts whose rows do not fit 160 KB).
__global__ __launch_bounds__(64) void zip_interlevel_lds_kernel(ZipLoss a) {
  extern __shared__ float zil_sm[];
  const int lvl = blockIdx.y;
  const float* sp = lvl == 0 ? a.s0 : a.s1;
  if (sp == nullptr) return;
  const int Sp = lvl == 0 ? a.S0 : a.S1, S = a.S2;
  const int stc = (S + 1) | 1, stw = S | 1, stp = (Sp + 1) | 1, stq = Sp | 1;
  float* sc = zil_sm;
  float* sw = sc + 64 * stc;
  float* scp = sw + 64 * stw;
  float* swp = scp + 64 * stp;
  const long r0 = (long)blockIdx.x * 64;
  const int nr = (int)(a.R - r0 < 64 ? a.R - r0 : 64);
  const int tid = threadIdx.x;
  auto stage = [&](const float* __restrict__ src, int len, float* dst, int stride) __attribute__((always_inline)) {
    for (int i = tid; i < nr * len; i += 64) { const int ray = i / len; dst[ray * stride + (i - ray * len)] = src[i]; }
  };
  stage(a.s2 + r0 * (S + 1), S + 1, sc, stc);
  stage(a.w2 + r0 * S, S, sw, stw);
  stage(sp + r0 * (Sp + 1), Sp + 1, scp, stp);
  stage((lvl == 0 ? a.w0 : a.w1) + r0 * Sp, Sp, swp, stq);
  __syncthreads();
  float l = 0.f;
  const float scale = a.inter_mult / ((float)a.R * (float)Sp);
  if (tid < nr)
    l = zip_interlevel_ray(sc + tid * stc, sw + tid * stw, S, scp + tid * stp, swp + tid * stq, Sp, lvl == 0 ? a.pw0 : a.pw1, scale, swp + tid * stq) * scale;
  __syncthreads();
  float* g = (lvl == 0 ? a.g_w0 : a.g_w1) + r0 * Sp;
  for (int i = tid; i < nr * Sp; i += 64) { const int ray = i / Sp; g[i] = swp[ray * stq + (i - ray * Sp)]; }
  l = wave_sum(l);
  if (tid == 0) atomicAdd(a.out + 9, l);
}

__global__ __launch_bounds__(64) void zip_loss_tail_kernel(ZipLoss a) {
  const long r = (long)blockIdx.x * 64 + threadIdx.x;
  const int task = blockIdx.y;
  const bool on = r < a.R;
  if (task > 0) {
    if (a.s2 == nullptr || a.inter_mult <= 0.f) return;
    const int lvl = task - 1;
    const float* sp = lvl == 0 ? a.s0 : a.s1;
    if (sp == nullptr) return;
    const int Sp = lvl == 0 ? a.S0 : a.S1;
    float l = 0.f;
    if (on) {
      const float* wp = (lvl == 0 ? a.w0 : a.w1) + r * Sp;
      float* g = (lvl == 0 ? a.g_w0 : a.g_w1) + r * Sp;
      const float sc = a.inter_mult / ((float)a.R * (float)Sp);
      l = zip_interlevel_ray(a.s2 + r * (a.S2 + 1), a.w2 + r * a.S2, a.S2, sp + r * (Sp + 1), wp, Sp, lvl == 0 ? a.pw0 : a.pw1, sc, g) * sc;
    }
    l = wave_sum(l);
    if (threadIdx.x == 0) atomicAdd(a.out + 9, l);
    return;
  }
  float l_data = 0.f, l_mse = 0.f, l_dep = 0.f, l_com = 0.f, l_sem = 0.f, l_dist = 0.f;
  if (on) {
    // ---- data term
    const float lm = a.lossmult != nullptr ? a.lossmult[r] : 1.f;
    const float inv = 1.f / a.out[0];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float d = a.rgb[3 * r + c] - a.tgt[3 * r + c];
      const float r2 = d * d;
      l_mse += lm * r2;
      if (a.mse) { l_data += lm * r2; a.g_rgb[3 * r + c] = a.data_mult * lm * 2.f * d * inv; }
      else {
        const float root = sqrtf(r2 + a.pad * a.pad);
        l_data += lm * root;
        a.g_rgb[3 * r + c] = a.data_mult * lm * (d / root) * inv;
      }
    }
    l_data *= a.data_mult * inv; l_mse *= inv;
    // ---- disparity L1 under the two masks
    if (a.depth != nullptr) {
      const float dp = a.depth[r] + 1e-5f;
      const float e = 1.f / dp - 1.f / (1e-5f + a.tdepth[r]);
      const float sg = (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f)) * (-1.f / (dp * dp));
      float g = 0.f;
      if (a.dmask != nullptr && a.out[1] > 0.f) { const float k = a.depth_lambda * a.dmask[r] / a.out[1]; l_dep = k * fabsf(e); g += k * sg; }
      if (a.cmask != nullptr && a.out[2] > 0.f) { const float k = a.depth_lambda * a.com_mult * a.cmask[r] / a.out[2]; l_com = k * fabsf(e); g += k * sg; }
      a.g_depth[r] = g;
    }
    // ---- semantic NLL
    if (a.sem != nullptr) {
      const int lab = min(max(a.labels[r], 0), a.C - 1);
      const float m = a.smask != nullptr ? a.smask[r] : 1.f;
      const float p = a.sem[r * a.C + lab] + 1e-6f;
      const float k = a.out[3] > 0.f ? a.sem_mult * m / a.out[3] : 0.f;
      l_sem = -k * logf(p);
      for (int c = 0; c < a.C; ++c) a.g_sem[r * a.C + c] = c == lab ? -k / p : 0.f;
    }
    // ---- distortion: sum_ij w_i w_j |u_i - u_j| through prefix sums over the sorted midpoints
    if (a.s2 != nullptr && a.dist_mult > 0.f) {
      const float* c = a.s2 + r * (a.S2 + 1);
      const float* w = a.w2 + r * a.S2;
      float* g = a.g_w2 + r * a.S2;
      double W = 0.0, WU = 0.0;
      for (int k = 0; k < a.S2; ++k) { const double u = (double)((c[k + 1] + c[k]) / 2.f); W += (double)w[k]; WU += (double)w[k] * u; }
      double Wl = 0.0, WUl = 0.0, acc = 0.0;
      const float sc = a.dist_mult / (float)a.R;
      for (int k = 0; k < a.S2; ++k) {
        const double u = (double)((c[k + 1] + c[k]) / 2.f), wk = (double)w[k], dk = (double)(c[k + 1] - c[k]);
        const double Wg = W - Wl - wk, WUg = WU - WUl - wk * u;
        const double inner = u * (Wl - Wg) - (WUl - WUg);
        acc += wk * inner + wk * wk * dk / 3.0;
        g[k] = sc * (float)(2.0 * inner + 2.0 * wk * dk / 3.0);
        Wl += wk; WUl += wk * u;
      }
      l_dist = sc * (float)acc;
    }
  }
  l_data = wave_sum(l_data); l_mse = wave_sum(l_mse); l_dep = wave_sum(l_dep); l_com = wave_sum(l_com); l_sem = wave_sum(l_sem);
  l_dist = wave_sum(l_dist);
  if (threadIdx.x == 0) {
    atomicAdd(a.out + 4, l_data); atomicAdd(a.out + 5, l_mse);
    if (a.depth != nullptr) { atomicAdd(a.out + 6, l_dep); atomicAdd(a.out + 7, l_com); }
    if (a.sem != nullptr) atomicAdd(a.out + 8, l_sem);
    if (a.s2 != nullptr && a.dist_mult > 0.f) atomicAdd(a.out + 10, l_dist);
  }
}

extern "C" int snerf_zip_loss_tail(const float* rgb, const float* tgt, const float* lossmult, const float* depth, const float* tdepth,
                                   const float* dmask, const float* cmask, const float* sem, const int* labels, const float* smask, int C,
                                   const float* s0, const float* w0, int S0, const float* s1, const float* w1, int S1, const float* s2,
                                   const float* w2, int S2, long R, int mse, float pad, float data_mult, float depth_lambda, float com_mult,
                                   float sem_mult, float pw0, float pw1, float inter_mult, float dist_mult, float* out, float* g_rgb,
                                   float* g_depth, float* g_sem, float* g_w0, float* g_w1, float* g_w2, void* stream) {
  if (R <= 0) return SNERF_OK;
  if (rgb == nullptr || tgt == nullptr || out == nullptr || g_rgb == nullptr) return SNERF_ERR_ARG;
  if (depth != nullptr && (tdepth == nullptr || g_depth == nullptr)) return SNERF_ERR_ARG;
  if (sem != nullptr && (labels == nullptr || g_sem == nullptr || C < 1)) return SNERF_ERR_ARG;
  if (s2 != nullptr) {
    if (w2 == nullptr || S2 < 1) return SNERF_ERR_ARG;
    if (dist_mult > 0.f && g_w2 == nullptr) return SNERF_ERR_ARG;
    if (inter_mult > 0.f) {
      if (s0 != nullptr && (w0 == nullptr || g_w0 == nullptr || S0 < 1 || !(pw0 > 0.f))) return SNERF_ERR_ARG;
      if (s1 != nullptr && (w1 == nullptr || g_w1 == nullptr || S1 < 1 || !(pw1 > 0.f))) return SNERF_ERR_ARG;
    }
  }
  ZipLoss a{rgb, tgt, lossmult, depth, tdepth, dmask, cmask, sem, labels, smask, s0, w0, s1, w1, s2, w2, R, C, S0, S1, S2, mse,
            pad, data_mult, depth_lambda, com_mult, sem_mult, pw0, pw1, inter_mult, dist_mult, out, g_rgb, g_depth, g_sem, g_w0, g_w1, g_w2};
  hipLaunchKernelGGL(zip_loss_prepare_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a);
  const bool inter = s2 != nullptr && inter_mult > 0.f && (s0 != nullptr || s1 != nullptr);
  // the interlevel walks from LDS when a workgroup's rows fit (the shipped interval counts: 49 KB); else from global memory (y = 1, 2)
  const int Spm = (s0 != nullptr ? S0 : 0) > (s1 != nullptr ? S1 : 0) ? S0 : S1;
  const size_t lds = (size_t)64 * 4 * (((S2 + 1) | 1) + (S2 | 1) + ((Spm + 1) | 1) + (Spm | 1));
  const bool staged = inter && lds <= 160 * 1024;
  hipLaunchKernelGGL(zip_loss_tail_kernel, dim3((unsigned)((R + 63) / 64), inter && !staged ? 3 : 1), dim3(64), 0, (hipStream_t)stream, a);
  if (staged) {
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)zip_interlevel_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(zip_interlevel_lds_kernel, dim3((unsigned)((R + 63) / 64), 2), dim3(64), lds, (hipStream_t)stream, a);
  }
  return snerf_check_launch();
}

// ---------------------------------------------------------------------------
// Frame quantisation for the S-NeRF++ wire format (s-nerfpp/zipnerf/random_render_waymo_seq.py:214-227):
//   rgb    u8  = trunc(clip(nan_to_num(rgb), 0, 1) * 255)                       internal/utils.py:111-116 (save_img_u8)
//   depth  u16 = (depth * 256 / scale_factor).astype(uint16)                    :218-219 (float32 product and quotient, truncation;
//                                                                                values past 65535 wrap like the x86 int32 cast numpy emits)
//   label  u8  = argmax_c semantic[., c] (first maximum)                        :222-223
//   paint  u8x3 = color_map[label]                                              :225
// One pass over the rendered buffers on the device: the host copy shrinks from (3 + 1 + C) floats to 9 bytes per pixel.
// ---------------------------------------------------------------------------
struct FrameQ {
  const float *rgb, *depth, *sem; long ld_sem; int C; const unsigned char* cmap; long P; float scale_factor;
  unsigned char* rgb8; unsigned short* depth16; unsigned char* label8; unsigned char* paint8;
};

__global__ __launch_bounds__(256) void frame_quantize_kernel(FrameQ a) {
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p >= a.P) return;
  if (a.rgb != nullptr) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float v = a.rgb[3 * p + c];
      v = v != v ? 0.f : v;                                   // nan_to_num: nan -> 0, +/-inf -> +/-max (then clipped)
      v = fminf(fmaxf(v, 0.f), 1.f) * 255.f;
      a.rgb8[3 * p + c] = (unsigned char)(int)v;
    }
  }
  if (a.depth != nullptr) {
    const float v = a.depth[p] * 256.f / a.scale_factor;
    int q;
    if (!(v == v)) q = (int)0x80000000;                       // cvttss2si's "integer indefinite" for nan / out-of-range
    else if (v >= 2147483648.f || v < -2147483648.f) q = (int)0x80000000;
    else q = (int)v;
    a.depth16[p] = (unsigned short)(q & 0xFFFF);
  }
  if (a.sem != nullptr) {
    const float* s = a.sem + p * a.ld_sem;
    int best = 0;
    float bv = s[0];
    for (int c = 1; c < a.C; ++c) {
      const float v = s[c];
      if (v > bv || (v != v && bv == bv)) { bv = v; best = c; }   // np.argmax: first maximum; a nan is a maximum
    }
    a.label8[p] = (unsigned char)best;
    if (a.paint8 != nullptr) {
#pragma unroll
      for (int c = 0; c < 3; ++c) a.paint8[3 * p + c] = a.cmap[3 * best + c];
    }
  }
}

extern "C" int snerf_frame_quantize(const float* rgb, const float* depth, const float* sem, long ld_sem, int C, const void* color_map, long P,
                                    float scale_factor, void* rgb_u8, void* depth_u16, void* label_u8, void* paint_u8, void* stream) {
  if (P <= 0) return SNERF_OK;
  if (rgb != nullptr && rgb_u8 == nullptr) return SNERF_ERR_ARG;
  if (depth != nullptr && (depth_u16 == nullptr || !(scale_factor != 0.f))) return SNERF_ERR_ARG;
  if (sem != nullptr && (label_u8 == nullptr || C < 1 || C > 256 || ld_sem < C)) return SNERF_ERR_ARG;
  if (sem != nullptr && paint_u8 != nullptr && color_map == nullptr) return SNERF_ERR_ARG;
  FrameQ a{rgb, depth, sem, ld_sem, C, (const unsigned char*)color_map, P, scale_factor, (unsigned char*)rgb_u8, (unsigned short*)depth_u16,
           (unsigned char*)label_u8, (unsigned char*)paint_u8};
  hipLaunchKernelGGL(frame_quantize_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return snerf_check_launch();
}

// ---------------------------------------------------------------------------
// Hash-grid weight decay of the zipnerf training step (s-nerfpp/zipnerf/internal/train_utils.py:184-203, hash_decay_loss, on by default:
// configs.py:74 hash_decay_mults = 0.1): per encoder, mult * mean_{l,c}( mean over the rows of level l of table[row, c]^2 ) -- the
// reference computes the per-level means with torch_scatter.segment_coo(param ** 2, idx, reduce='mean').  Value and gradient
// (grad += 2 mult p / (rows_l L C)) in one pass over the table; grid.y = level.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hash_decay_kernel(const float* __restrict__ table, float* __restrict__ grad, const int* __restrict__ offsets,
                                                         int L, int C, float mult, float* __restrict__ loss) {
  const int level = blockIdx.y;
  const long b = (long)offsets[level] * C, e = (long)offsets[level + 1] * C;
  const float k = mult / ((float)(offsets[level + 1] - offsets[level]) * (float)L * (float)C);
  float part = 0.f;
  for (long i = b + (long)blockIdx.x * 256 + threadIdx.x; i < e; i += (long)gridDim.x * 256) {
    const float p = table[i];
    part += p * p;
    grad[i] += 2.f * k * p;
  }
  part = wave_sum(part);
  __shared__ float ws[4];
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0 && loss != nullptr) atomicAdd(loss, k * (ws[0] + ws[1] + ws[2] + ws[3]));
}

extern "C" int snerf_hash_decay(const float* table, float* grad, const int* offsets, int L, int C, float mult, float* loss, void* stream) {
  if (L <= 0 || mult == 0.f) return SNERF_OK;
  if (table == nullptr || grad == nullptr || offsets == nullptr || C < 1) return SNERF_ERR_ARG;
  hipLaunchKernelGGL(hash_decay_kernel, dim3(512, L), dim3(256), 0, (hipStream_t)stream, table, grad, offsets, L, C, mult, loss);
  return snerf_check_launch();
}
