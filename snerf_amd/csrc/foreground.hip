// Image-space foreground composite of S-NeRF++ stage 1: byte / integer work, bit-exact.
//   snerf_fg_paste        <- s-nerfpp/stage1_code/utils_render.py:826-1005 handle_occlusion_paste (given the mesh depth per pixel)
//   snerf_fg_bound        <- utils_render.py:306-324 get_bound_im (cv2.dilate XOR cv2.erode, rect kernel) + ip_utils.py:10-19 set_diff
//   snerf_fg_accumulate   <- utils_render.py:338-361 fuse_bound + generate_images.py:161 mask union
//   snerf_fg_blank        <- utils_render.py:327-335 fuse_bound_and_im
// One lane per pixel, HBM-bound: every image is read once and written once.
#include "common.h"

struct FgPaste {
  unsigned char* bg; const unsigned char* fg; unsigned char* mask; float* depth; unsigned char* sem; const float* fg_depth;
  long P; int class_id, person; int* counters;
};

__global__ __launch_bounds__(256) void fg_paste_kernel(FgPaste a) {
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  int masked = 0, valid = 0;
  if (p < a.P && a.mask[3 * p] > 0) {
    masked = 1;
    const float fd = a.person ? -1.f : a.fg_depth[p];
    const int s = a.sem[p];
    valid = (fd < a.depth[p]) || s == 0 || s == 1 || s == 8;
    if (valid) {
#pragma unroll
      for (int c = 0; c < 3; ++c) a.bg[3 * p + c] = a.fg[3 * p + c];
      a.depth[p] = fd;
      a.sem[p] = (unsigned char)a.class_id;
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) a.mask[3 * p + c] = 0;
    }
  }
  const unsigned long long bm = __ballot(masked), bv = __ballot(valid);
  if ((threadIdx.x & 63) == 0) {
    if (bm) atomicAdd(a.counters, __popcll(bm));
    if (bv) atomicAdd(a.counters + 1, __popcll(bv));
  }
}

extern "C" int snerf_fg_paste(void* bg_im, const void* fg_im, void* mask_im, float* depth, void* semantic, const float* fg_depth, long P,
                              int class_id, int person, int* counters, void* stream) {
  if (P <= 0) return SNERF_OK;
  if (bg_im == nullptr || fg_im == nullptr || mask_im == nullptr || depth == nullptr || semantic == nullptr || counters == nullptr ||
      (fg_depth == nullptr && !person) || class_id < 0 || class_id > 255)
    return SNERF_ERR_ARG;
  hipMemsetAsync(counters, 0, 2 * sizeof(int), (hipStream_t)stream);
  FgPaste a{(unsigned char*)bg_im, (const unsigned char*)fg_im, (unsigned char*)mask_im, depth, (unsigned char*)semantic, fg_depth, P, class_id,
            person, counters};
  hipLaunchKernelGGL(fg_paste_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return snerf_check_launch();
}

// boundary band: max and min of channel 0 over the r x r window [x - r/2, x - r/2 + r) (OpenCV's rect kernel with its default anchor;
// pixels outside the image never win), band = (max != 0) xor (min != 0); mask_out = set_diff(mask, band) per channel
__global__ __launch_bounds__(256) void fg_bound_kernel(const unsigned char* __restrict__ mask, int H, int W, int r, unsigned char* __restrict__ bound,
                                                       unsigned char* __restrict__ mask_out) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= W || y >= H) return;
  const int a = r / 2;
  int mx = 0, mn = 255;
  for (int dy = 0; dy < r; ++dy) {
    const int yy = y - a + dy;
    if (yy < 0 || yy >= H) continue;
    for (int dx = 0; dx < r; ++dx) {
      const int xx = x - a + dx;
      if (xx < 0 || xx >= W) continue;
      const int v = mask[3 * ((long)yy * W + xx)];
      mx = max(mx, v); mn = min(mn, v);
    }
  }
  const unsigned char b = ((mx != 0) != (mn != 0)) ? 255 : 0;
  const long p = 3 * ((long)y * W + x);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    bound[p + c] = b;
    if (mask_out != nullptr) mask_out[p + c] = (mask[p + c] > 0 && !b) ? 255 : 0;
  }
}

extern "C" int snerf_fg_bound(const void* mask_im, int H, int W, int r, void* bound_im, void* mask_out, void* stream) {
  if (H <= 0 || W <= 0) return SNERF_OK;
  if (mask_im == nullptr || bound_im == nullptr || mask_im == mask_out || r < 1 || r > 255) return SNERF_ERR_ARG;
  hipLaunchKernelGGL(fg_bound_kernel, dim3((W + 63) / 64, (H + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)mask_im, H, W, r,
                     (unsigned char*)bound_im, (unsigned char*)mask_out);
  return snerf_check_launch();
}

// per byte: total_bound <- (bound \ total_mask) | (total_bound \ mask); total_mask <- total_mask | mask; all as 0 / 255
__global__ __launch_bounds__(256) void fg_accumulate_kernel(unsigned char* __restrict__ total_mask, unsigned char* __restrict__ total_bound,
                                                            const unsigned char* __restrict__ bound, const unsigned char* __restrict__ mask, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const bool tm = total_mask[i] > 0, tb = total_bound[i] > 0, b = bound[i] > 0, m = mask[i] > 0;
  total_bound[i] = ((b && !tm) || (tb && !m)) ? 255 : 0;
  total_mask[i] = (m || tm) ? 255 : 0;
}

extern "C" int snerf_fg_accumulate(void* total_mask, void* total_bound, const void* bound, const void* mask, long nbytes, void* stream) {
  if (nbytes <= 0) return SNERF_OK;
  if (total_mask == nullptr || total_bound == nullptr || bound == nullptr || mask == nullptr) return SNERF_ERR_ARG;
  hipLaunchKernelGGL(fg_accumulate_kernel, dim3((unsigned)((nbytes + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (unsigned char*)total_mask,
                     (unsigned char*)total_bound, (const unsigned char*)bound, (const unsigned char*)mask, nbytes);
  return snerf_check_launch();
}

__global__ __launch_bounds__(256) void fg_blank_kernel(unsigned char* __restrict__ im, const unsigned char* __restrict__ bound, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n && bound[i] > 0) im[i] = 0;
}

extern "C" int snerf_fg_blank(void* im, const void* bound, long nbytes, void* stream) {
  if (nbytes <= 0) return SNERF_OK;
  if (im == nullptr || bound == nullptr) return SNERF_ERR_ARG;
  hipLaunchKernelGGL(fg_blank_kernel, dim3((unsigned)((nbytes + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (unsigned char*)im,
                     (const unsigned char*)bound, nbytes);
  return snerf_check_launch();
}
