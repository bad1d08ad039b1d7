/* libsnerf_hip -- C ABI of the MI355X (gfx950) volumetric-render hot path.
 *
 * Drop-in boundary for the S-NeRF background renderer.
 * The reference has exactly one native interface -- the pybind module of
 * s-nerfpp/zipnerf/gridencoder (src/bindings.cpp:5-9, src/gridencoder.h:12-15)
 * -- and otherwise runs PyTorch eager ops; every entry point below names the
 * reference function (file:line under /root/reference) it replaces.
 *
 * Conventions
 *   - plain C: raw DEVICE pointers, explicit sizes/strides (in elements),
 *     `stream` is a hipStream_t passed as void* (0 = default stream);
 *   - the caller allocates everything, the callee writes in place, nothing is
 *     returned but a status: 0 ok, 1 bad argument, 2 launch failure; no
 *     exceptions cross the ABI, no hidden global state, thread-safe per stream;
 *   - `dtype`: 0 = fp32 activations/weights (exact-parity mode, fp32 MFMA),
 *     1 = bf16 activations/weights with fp32 accumulation;
 *   - all launches are asynchronous on `stream`.
 */
#ifndef SNERF_HIP_H
#define SNERF_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

#define SNERF_DT_F32 0
#define SNERF_DT_BF16 1
#define SNERF_DT_F16 2 /* hash-grid tables; the fp16 compute mode of path C (GEMMs: fp16 operands on the f16 MFMA, fp32 accumulation; features, activations, packed weights) */
#define SNERF_DT_F64 3 /* stand-alone GridEncoder operator only (the reference dispatches float / double / half) */
#define SNERF_DT_BF16X3 4 /* snerf_linear_fwd / snerf_linear_wgrad only: split-bf16 operands (the fp32-parity mode at bf16 MFMA rates) */
#define SNERF_ACT_NONE 0
#define SNERF_ACT_RELU 1
#define SNERF_ACT_MASK 2 /* y = aux > 0 ? y : 0 : ReLU backward fused into the data-gradient GEMM */
#define SNERF_ACT_RELU_BITS 3 /* ReLU; `aux` (uint32 words, OUTPUT) also receives one bit per element (y > 0) */
#define SNERF_ACT_MASK_BITS 4 /* as SNERF_ACT_MASK with `aux` = the words an ACT_RELU_BITS launch of the same [M, N] wrote */

int snerf_version(void);
/* the hipError_t (as int) that the most recent "kernel launch failure" status (2) of any entry stood for; 0 if there was none */
int snerf_last_hip_error(void);
/* Test utility (tests/test_stale_lds.py): fills the LDS of every CU with a seeded pseudo-random pattern.  LDS content survives from one
 * kernel to the next; a kernel that reads LDS before its own write / DMA has landed shows up as a result that depends on `seed`. */
int snerf_debug_lds_scribble(int seed, void* stream);

/* ---- tiny-MLP layers (MFMA GEMMs) ------------------------------------------------------------
 * Y[M, n_store] = act(A[M,K] . W[N,K]^T + bias).  Replaces nn.Linear(+ReLU):
 *   s-nerf/model/models.py:200-214 (DenseBlock), :232-255 (MLP), :307-315 (proposal),
 *   s-nerf/model/run_nerf_helpers.py:86-126 (NeRF), called through run_network :460-474.
 * W is the packed weight [N (multiple of 128), K] in `dtype`; K a multiple of 64 (bf16) / 32 (fp32)
 * with zero padding; lda/ldw multiples of 8/4; Y is `dtype` or fp32 (out_f32).  With W := W^T the
 * same entry computes the data gradient; ACT_MASK applies the ReLU mask from `aux` and `colsum`
 * (fp32 [n_store], accumulated) receives the column sums = bias gradient of the layer below; `colsum_ws`
 * (fp32 [max(2*ceil(M/128), 2*min(ceil(M/256)*(N/256), 1024)), N], contents irrelevant) lets that reduction run without atomics:
 * one partial row per 128-row slab, or -- persistent kernel -- one per (workgroup, wave row), kept in registers across its tiles.
 * The two *_BITS activations carry the ReLU mask of a training step as 1 bit per element (8*ceil(M/256) * N/64 blocks of 64
 * words; word l of block (32-row block, 64-column group) = rows 8*it + l/8, columns 8*(l%8) + e at bit 8*it + e): 1/16 of the
 * bytes of the activation and DMA-able ahead of use.  Only the persistent kernel implements them (bf16, variant 8,
 * N % 256 == 0, K >= 128, 16-byte aligned Y rows); other launches return SNERF_ERR_ARG.
 * variant (low nibble): 0 = 128x128 tile, 1 = 256x256 tile, 4 = 256x256 8-phase, 8 = 256x256 persistent 8-phase (bf16,
 * N % 256 == 0, K >= 128, 16-byte epilogue; other shapes fall back to 4, then 0); higher bits = ablation switches, except
 * bit 14 (0x4000; dtype bf16, ACT_MASK): `aux` is an activation a split-bf16 (SNERF_DT_BF16X3) forward saved -- interleaved hi / lo per 64
 * columns -- while A, W and Y are plain bf16: the single-pass data gradient behind a three-pass forward (compute="bf16x3_fwd"). */
int snerf_linear_fwd(const void* A, long lda, const void* W, long ldw, const float* bias, void* Y, long ldy,
                     const void* aux, long ldaux, float* colsum, float* colsum_ws, int M, int N, int K, int n_store,
                     int act, int dtype, int out_f32, int variant, void* stream);

/* dW[n_valid, k_valid] (fp32, ldw) += dZ[M,N]^T . X[M,K]  -- weight gradient of the same layers
 * (autograd of nn.Linear in the reference; train.py:213 loss.backward()).  `zeros` = >=16 bytes of
 * device zeros (source for rows past M).  variant bit 1 (bf16, N and K multiples of 128): transposing LDS reads; bit 2 (bf16,
 * N % 256 == 0, K >= 256): the 256 x 256 8-phase kernel; bits 16 / 32 / 64: probe only -- 1024 / 2048 / 4096 M-slices for the
 * 128 x 128 kernel instead of its default of about 512 (tools/gemm_tn_slices_probe.py); bit 14 (0x4000; dtype bf16): X is an activation a
 * split-bf16 forward saved ([M, >= 2 K], hi / lo interleaved per 64 columns) and its hi half is multiplied, K = its LOGICAL width, dZ plain bf16. */
int snerf_linear_wgrad(const void* Z, long ldz, const void* X, long ldx, float* dW, long ldw, const void* zeros,
                       int M, int N, int K, int n_valid, int k_valid, int dtype, int variant, void* stream);

/* ---- encoders ---------------------------------------------------------------------------------
 * Classic positional encoding written straight into MLP operand buffers.
 *   run_nerf_helpers.py:22-70 (Embedder.embed: [x, sin(2^k x), cos(2^k x)]_k) and the per-sample
 *   broadcast of the view direction in run_network (:465-469).
 * pts [M,3]; viewdirs [N,3] rows vd_stride apart (NULL = no view branch); S samples per ray.
 * dst1/dst2 (dst2 optional) get w_pts columns (3+6L values + zero pad); dstv gets w_views columns. */
int snerf_classic_embed(const float* pts, const float* viewdirs, int vd_stride, int S, long M, int L, int Lv,
                        void* dst1, long ld1, void* dst2, long ld2, int w_pts, void* dstv, long ldv, int w_views,
                        int dtype, void* stream);

/* mip path: s -> t (mip.py:7-9) -> cast_rays cone/cylinder (mip.py:80-91, 56-77, 31-53) -> contraction
 * fn2 + Jacobi_g (mip.py:343-374) -> diagonal of J diag(c) J^T (mip.py:381-395) -> integrated_pos_enc
 * (mip.py:94-118, 24-28; math_ops.py:6-12), `width` >= 6*max_deg columns (zero padded) into dst1 (and dst2).
 * means_out/covs_out: optional fp32 [N*S,3] copies of the contracted mean / covariance diagonal.
 * `cone`: bit 0 = ray shape (1 cone, 0 cylinder); bit 1 = --disable_integration (models.py:132-133: the covariances are replaced by zeros
 * before the encoding; the backward entries take the same value). */
int snerf_mip_encode(const float* s_vals, const float* origins, const float* directions, const float* radii,
                     const float* near, const float* far, long n_rays, int S, int cone, int transform_idx, int max_deg,
                     void* dst1, long ld1, void* dst2, long ld2, int width, float* means_out, float* covs_out,
                     int dtype, const int* sample_id, long n_rows, void* stream);

/* The same with the warp of sample2enc selected (mip.py:367-378 warp_fn, model argument `fn`): fn_idx 1 = the contraction above,
 * fn_idx 0 = the view-centred warp fn1 (x - viewc) / sqrt(|x - viewc| far) (mip.py:368-369) with Jacobi_f (mip.py:323-340:
 * (l I - x x^T) / l^1.5 / sqrt(max far), l = |x| + 1e-5 at the unshifted mean).  (vx, vy, vz) = viewc, the mean camera centre
 * (train.py:36, eval.py:50); far_max: DEVICE scalar = max over the batch of rays.far (ignored for fn_idx 1). */
int snerf_mip_encode_warp(const float* s_vals, const float* origins, const float* directions, const float* radii,
                          const float* near, const float* far, long n_rays, int S, int cone, int transform_idx, int max_deg,
                          void* dst1, long ld1, void* dst2, long ld2, int width, float* means_out, float* covs_out,
                          int dtype, const int* sample_id, long n_rows, int fn_idx, float vx, float vy, float vz,
                          const float* far_max, void* stream);

/* mip.py:12-21 pos_enc(viewdirs, 0, deg, append_identity) tiled per sample (models.py:285-287). */
int snerf_mip_viewenc(const float* viewdirs, long n_rays, int S, int deg, void* dst, long ld, int width, int dtype,
                      const int* sample_id, long n_rows, void* stream);

/* ---- samplers (bit-exact interval indices; fp64 sequential accumulation per ray) -----------------
 * run_nerf_helpers.py:336-379 sample_pdf.  nc = number of bins = cdf entries, nc-1 weights (rows ld_w apart).
 * mid_mode 1 = the render_rays call (render.py:378-380): `bins` is z_vals [N,nc+1] (bin j = mid of z[j],z[j+1]) and
 * `weights` points at weights[:,1]; mid_mode 0: bins [N,nc] as given.  u [N,Nf] rows u_stride apart (0 = shared row).
 * inds (int32, nullable) = searchsorted(cdf,u,right); z_std (nullable) = std(samples,-1,unbiased=False) (render.py:405). */
int snerf_classic_sample_pdf(const float* bins, long ld_bins, int mid_mode, const float* weights, long ld_w, int nc,
                             const float* u, long u_stride, long N, int Nf, float* samples, int* inds, float* z_std,
                             void* stream);
/* math_ops.py:50-54 (randomized branch of sorted_piecewise_constant_pdf): u = min(arange(P) * s + jit, 1 - eps) in place over the
 * uniform draw jit [N, P] (torch's own RNG launch stays the caller's, so the draws are the reference's); one launch, same roundings. */
int snerf_jitter_u(float* u, long N, int P, float s, void* stream);
/* render.py:354/:385  pts = rays_o[...,None,:] + rays_d[...,None,:] * z_vals[...,:,None]; rays rows = [o3,d3,...]. */
int snerf_classic_points(const float* rays, int ray_stride, const float* z_vals, long N, int S, float* pts, void* stream);
/* render.py:383 torch.sort(torch.cat([z_vals, z_samples], -1), -1) */
int snerf_classic_merge_sort(const float* a, int na, const float* b, int nb, long N, float* out, void* stream);
/* mip.py:294-316 blur-pool + resample_padding, then math_ops.py:19-76 sorted_piecewise_constant_pdf.
 * idx_out (int32, nullable) = #(u >= cdf) - 1. */
int snerf_mip_resample(const float* s_vals, const float* weights, const float* u, long u_stride, long N, int S, int Nf,
                       float resample_padding, float* out, int* idx_out, void* stream);
/* render.py:330-352 (mode 0: z = near(1-t)+far t or disparity) / mip.py:268-288 (mode 1: s = t) with optional
 * stratified jitter rnd [N,P]; base [P] = linspace(0,1,P). */
int snerf_stratified(const float* base, const float* rnd, const float* near, const float* far, int nf_stride, long N,
                     int P, int mode, int lindisp, float* out, void* stream);

/* ---- compositing --------------------------------------------------------------------------------
 * models.py:166-175 activations + mip.py:151-189 real_volumetric_rendering.  raw_rgb NULL = proposal level. 
 * g_dirs (nullable) [N,3] <- d loss / d directions through delta = (t1 - t0) * |d| (mip.py:160-161; the reference's pose refinement). */
int snerf_mip_composite_fwd(const float* raw_rgb, long ld_rgb, const float* raw_density, long ld_den, const float* noise,
                            const float* s_vals, const float* dirs, const float* near, const float* far, long N, int S,
                            int transform_idx, int white, float rgb_padding, float density_bias, float* comp_rgb,
                            float* distance, float* acc, float* weights, const int* row_index, void* stream);
int snerf_mip_composite_bwd(const float* raw_rgb, long ld_rgb, const float* raw_density, long ld_den, const float* noise,
                            const float* s_vals, const float* dirs, const float* near, const float* far, long N, int S,
                            int transform_idx, int white, float rgb_padding, float density_bias, const float* weights,
                            const float* distance, const float* g_rgb, const float* g_dist, const float* g_acc,
                            const float* g_w, float* d_raw_rgb, long ld_drgb, float* d_raw_density, long ld_dden,
                            float* g_dirs, void* stream);
/* run_nerf_helpers.py:381-424 raw2outputs */
int snerf_classic_composite_fwd(const float* raw, long ld, const float* noise, const float* z_vals, const float* rays_d,
                                int rd_stride, long N, int S, int white, float* rgb_map, float* disp_map, float* acc_map,
                                float* weights, float* depth_map, void* stream);
int snerf_classic_composite_bwd(const float* raw, long ld, const float* noise, const float* z_vals, const float* rays_d,
                                int rd_stride, long N, int S, int white, const float* weights, const float* acc_map,
                                const float* depth_map, const float* g_rgb, const float* g_disp, const float* g_acc,
                                const float* g_depth, const float* g_w, float* d_raw, long ld_draw, void* stream);

/* ---- multiresolution hash-grid encoder (S-NeRF++ / zipnerf) ------------------------------------------------------
 * Same entry points and argument order as the reference's pybind module (s-nerfpp/zipnerf/gridencoder/src/bindings.cpp:5-9,
 * src/gridencoder.h:12-15; kernels src/gridencoder.cu:87-245, :248-340, :343-369, :506-610):
 *   grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx?, gridtype, align_corners, interp)
 *   grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx?, grad_inputs?, ...)
 *   grad_total_variation(inputs, embeddings, grad, offsets, weight, B, D, C, L, S, H, gridtype, align_corners)
 * plus `dtype` of the table / outputs / gradients (SNERF_DT_F32, SNERF_DT_F16 or SNERF_DT_F64; the reference dispatches on the
 * tensor dtype, AT_DISPATCH_FLOATING_TYPES_AND_HALF gridencoder.cu:469), the strides (elements) of the level and point axes of `outputs` / `grad` (reference layout [L,B,C]:
 * stride_l = B*C, stride_b = C; [B, L*C] directly: stride_l = C, stride_b = L*C) and the stream (the reference always
 * uses the legacy default stream).  inputs fp32 [B,D] in [0,1] (out-of-bound points -> zeros); D in {2,3,4,5} (gridencoder.cu:376-399); C in {1,2,4,8};
 * gridtype 0 hash / 1 tiled; interp 0 linear / 1 smoothstep.  grad_embeddings / grad_inputs must arrive zeroed. */
int snerf_grid_encode_fwd(const float* inputs, const void* embeddings, const int* offsets, void* outputs, int B, int D, int C,
                          int L, float S, int H, void* dy_dx, int gridtype, int align_corners, int interp, int dtype,
                          long out_stride_l, long out_stride_b, void* stream);
/* snerf_grid_encode_fwd in kernel_grid's own form (one thread per (point, level), eight independent row loads: gridencoder.cu:87-245) for
 * EVERY instantiation -- the A/B partner of the measurement legs and parity tests.  Same arguments, same results bit for bit.  (A second
 * entry instead of a process-wide switch: no hidden state in the library.) */
int snerf_grid_encode_fwd_ref(const float* inputs, const void* embeddings, const int* offsets, void* outputs, int B, int D, int C,
                              int L, float S, int H, void* dy_dx, int gridtype, int align_corners, int interp, int dtype,
                              long out_stride_l, long out_stride_b, void* stream);
int snerf_grid_encode_bwd(const void* grad, const float* inputs, const void* embeddings, const int* offsets, void* grad_embeddings,
                          int B, int D, int C, int L, float S, int H, const void* dy_dx, void* grad_inputs, int gridtype,
                          int align_corners, int interp, int dtype, long grad_stride_l, long grad_stride_b, void* stream);
int snerf_grid_tv_grad(const float* inputs, const void* embeddings, void* grad, const int* offsets, float weight, int B, int D,
                       int C, int L, float S, int H, int gridtype, int align_corners, int dtype, void* stream);
/* The table gradient of grid_encode_backward (gridencoder.cu:248-340, bindings.cpp:7) WITHOUT atomics on the table, for the instantiations
 * zipnerf constructs (internal/models.py:413-421: D = 3, hash, linear, align_corners = False; C in {1, 2, 4, 8}): runs of consecutive points in one
 * cell become records (row, sum of w x grad), binned by destination row range through an LDS stage (full, aligned lines), accumulated per
 * bin in LDS in 64-bit fixed point (bit-reproducible; sums exact to 2^-31 of the largest |grad|).  ONE call, LEVEL by level and CHUNK of
 * points by chunk: count -> scan -> staged write -> accumulate into the level's int64 image, so that the workspace is bounded whatever B
 * is.  `ws`: 256-byte aligned; ANY size from the smallest feasible layout up works -- the call plans its chunks and (for a [B, L*C]
 * gradient) the number of levels it transposes per sweep from `ws_bytes`; snerf_grid_encode_bwd_binned_ws_bytes(...) returns the
 * recommended size: everything in one chunk when that takes less than 1 GB, else 1 GB (-1 = unsupported level layout: more than 1024
 * row ranges per level); snerf_grid_encode_bwd_binned_plan(...) reports what a given size would run (out[6]: chunks per level, points
 * per chunk, levels per transposed group (0 = none), kernel launches, record capacity of a chunk, bytes used).
 * grad [B, L*C] (stride_l = C, stride_b = L*C) or [L, B, C] (stride_l = B*C, stride_b = C: the layout the reference's backward builds,
 * grid.py:74 -- no transposition sweep) in grad_dtype (F32 / F16); grad_embeddings [sO, C] in out_dtype (F32 / F16), += like the reference
 * (arrives zeroed); offsets_host = the same int32 [L + 1] offsets in host memory (the plan is made on the host); half_records != 0
 * (C = 1 / 4 only): contributions rounded once to fp16 (the reference adds __half2 atomics there). */
/* starts[L, 1024] (int64) = exclusive scan of counts[L, 1024] (int32), flat over levels and bins: the record offsets of the binned table
 * gradient (between its count and write passes), on the device. */
int snerf_zip_bin_scan(const int* counts, long* starts, int L, void* stream);
long snerf_grid_encode_bwd_binned_ws_bytes(long B, int C, int L, const int* offsets_host, int half_records);
int snerf_grid_encode_bwd_binned_plan(long B, int C, int L, const int* offsets_host, int half_records, int grad_dtype, int level_major,
                                      long ws_bytes, long* out);
int snerf_grid_encode_bwd_binned(const void* grad, const float* inputs, const int* offsets, const int* offsets_host, void* grad_embeddings,
                                 long B, int C, int L, float S, int H, int grad_dtype, int out_dtype, long grad_stride_l,
                                 long grad_stride_b, int half_records, void* ws, long ws_bytes, void* stream);

/* ---- S-NeRF++ / zipnerf background model (s-nerfpp/zipnerf/internal) --------------------------------------------------
 * One launch per sampling level: stepfun.max_dilate_weights (stepfun.py:75-105; dilate = 0 skips it, level 0) with the
 * caller's [1:-1] trim (models.py:187-188), the annealed logits (models.py:196-203), stepfun.sample_intervals
 * (stepfun.py:251-294 -> 175-218 -> 154-161 -> 108-128, math.sorted_interp math.py:88-107) at the centres `u` [R,n]
 * (rows u_stride apart, 0 = shared) and the power-transformation ray warp s -> t (coord.py:103-162, lam).  sdist [R,S0+1],
 * weights [R,S0] -> sdist_out / tdist_out [R,n+1].  One wave per ray; at most 256 intervals after the dilation (3*S0 - 2 <= 256,
 * i.e. S0 <= 86; undilated S0 <= 256), SNERF_ERR_ARG beyond. */
int snerf_zip_resample(const float* sdist, const float* weights, int S0, const float* u, long u_stride, int n,
                       const float* near, const float* far, long R, float dilation, int dilate, float anneal,
                       float resample_padding, float lam, float dom0, float dom1, float* sdist_out, float* tdist_out,
                       void* stream);
/* render.cast_rays (render.py:129-168; n multisamples on an m-turn helix, deg_jitter [R,S,n] or NULL) + coord.contract_mean_std
 * (coord.py:51-63) + /2 + GridEncoder forward (gridencoder.cu:87-245, hash type, linear) + erf down-weighting and mean over
 * the multisamples (models.py:488-497) -> feat [R*S, ld] (columns level*C + c; feat_dtype fp32/bf16).  table fp32 or fp16
 * (SNERF_DT_F16); grid_sizes = GridEncoder.grid_sizes; Sl = log2(per_level_scale).  levels_per_thread: 0 / 1 = one thread per
 * (interval, level) (best for scattered training rays: most gathers in flight); L = one thread per interval evaluates its multisamples
 * once for all levels (best for the coherent rays of a frame); values in between group the levels.  Same results. */
int snerf_zip_encode_fwd(const float* tdist, const float* origins, const float* directions, const float* radii,
                         const float* base_x, const float* base_y, const float* deg_jitter, const void* table,
                         const int* offsets, const int* grid_sizes, void* feat, long ld, long R, int S, int L, int C, int n,
                         int m, float Sl, int H, float std_scale, int table_dtype, int feat_dtype, int levels_per_thread, void* stream);
/* snerf_zip_encode_fwd for a training step whose table gradient is the binned reduction (snerf_zip_encode_bwd_binned): one thread per
 * (interval, level), and the same sweep also IS that gradient's pass 0 -- it has the 8 rows of every multisample's cell in hand for its
 * gathers and counts the records the backward will emit per bin (counts [L, 1024], zeroed by the caller) and reserves each workgroup's
 * ranges (wg_offsets [L, ceil(R S / 256), 1024]); ksplit_host / level_rows_host as for snerf_zip_encode_bwd_binned.  The backward then
 * starts at its record pass (1 / 3) with these two buffers.  C = 1 or 4, n <= 8 multisamples. */
int snerf_zip_encode_fwd_count(const float* tdist, const float* origins, const float* directions, const float* radii,
                               const float* base_x, const float* base_y, const float* deg_jitter, const void* table,
                               const int* offsets, const int* grid_sizes, void* feat, long ld, long R, int S, int L, int C, int n,
                               int m, float Sl, int H, float std_scale, int table_dtype, int feat_dtype, const int* ksplit_host,
                               const int* level_rows_host, int* counts, void* wg_offsets, void* stream);
/* The proposal MLP of a zipnerf TRAINING step (internal/models.py:425-427, 481-519 with disable_rgb: Linear(L -> hidden) + ReLU +
 * Linear(hidden -> 1) on the grid features) as one launch each way instead of per-layer GEMMs over 64-column padded buffers: F [P, ldf]
 * (feat_dtype fp32 / bf16, L <= 16 feature columns, ldf >= L -- a compact buffer), parameters fp32 in the reference's layouts
 * (density_layer.0.weight [hidden, L], .bias [hidden], density_layer.2.weight [1, hidden], .bias [1]; hidden <= 64), raw [P] fp32.
 * round_bf16: rounding mode = the rounding points of the GEMM route in that compute dtype (weights and stored activations rounded, fp32
 * accumulation): 0 none (fp32), 1 bf16, 2 fp16 (requires feat_dtype SNERF_DT_F16).  The backward
 * recomputes the hidden activations from F, writes dF [P, lddf] (columns L .. lddf - 1 zero; lddf <= 64) and adds the parameter
 * gradients into g_* (fp32, same layouts) by per-workgroup partial sums folded in a fixed order (bit-reproducible);
 * ws: snerf_zip_prop_mlp_ws_floats(L, hidden, P) floats of scratch. */
int snerf_zip_prop_mlp_ws_floats(int L, int hidden, long P);
int snerf_zip_prop_mlp_fwd(const void* F, long ldf, long P, int L, const float* w1, const float* b1, const float* w2, const float* b2,
                           int hidden, int round_bf16, int feat_dtype, float* raw, void* stream);
int snerf_zip_prop_mlp_bwd(const void* F, long ldf, const float* d_raw, long P, int L, const float* w1, const float* b1,
                           const float* w2, const float* b2, int hidden, int round_bf16, int feat_dtype, void* dF, long lddf,
                           float* g_w1, float* g_b1, float* g_w2, float* g_b2, float* ws, long ws_floats, void* stream);
/* matching scatter-add of grad_feat [R*S, ld] into the fp32 table gradient (gridencoder.cu:248-340 composed with the mean /
 * erf weights); grad_table accumulates (fp32 atomics).  The first `lds_levels` levels (small dense tables) are accumulated
 * in LDS by persistent workgroups, in slabs of `lds_cells` rows (lds_cells*C*4 <= 160 KB; lds_slabs = sum of
 * ceil(rows_l / lds_cells) over those levels), and flushed once.  grad_table_bf16 (nullable, C even): a bf16 [entries, C]
 * buffer that receives the contributions of the remaining (hashed) levels as packed bf16-pair atomics instead -- half the
 * atomic operations (what the reference does with __half2 under autocast, gridencoder.cu:300-330); the caller adds it to the
 * fp32 gradient afterwards. */
int snerf_zip_encode_bwd(const float* tdist, const float* origins, const float* directions, const float* radii,
                         const float* base_x, const float* base_y, const float* deg_jitter, const int* offsets,
                         const int* grid_sizes, const void* grad_feat, long ld, float* grad_table, long R, int S, int L, int C,
                         int n, int m, float Sl, int H, float std_scale, int feat_dtype, int lds_levels, long lds_cells,
                         int lds_slabs, void* grad_table_bf16, void* stream);
/* render.compute_alpha_weights (render.py:170-189, opaque_background) + volumetric_rendering (render.py:192-233: rgb with
 * clamped background weight, depth = clip(exp(E_w[log t_mid]))) fused with density = softplus(raw + bias) (models.py:586)
 * and rgb = sigmoid(raw)(1 + 2 pad) - pad (models.py:689-703).  raw_rgb NULL = proposal level (rgb = 0). */
int snerf_zip_composite_fwd(const float* raw_rgb, long ld_rgb, const float* raw_density, long ld_den, const float* tdist,
                            const float* dirs, long R, int S, int opaque, float bg, float rgb_padding, float density_bias,
                            float* rgb, float* depth, float* acc, float* weights, void* stream);
int snerf_zip_composite_bwd(const float* raw_rgb, long ld_rgb, const float* raw_density, long ld_den, const float* tdist,
                            const float* dirs, long R, int S, int opaque, float bg, float rgb_padding, float density_bias,
                            const float* weights, const float* acc, const float* depth, const float* g_rgb, const float* g_depth,
                            const float* g_acc, const float* g_w, float* d_raw_rgb, long ld_drgb, float* d_raw_density,
                            long ld_dden, float* g_dirs, void* stream);

/* ---- training tail ------------------------------------------------------------------------------
 * torch.optim.Adam step over a flat fp32 arena (model_utils.py:23-34 builds Adam); grad_scale folds the
 * data-parallel 1/world_size; zero_grad clears g for the next step. */
int snerf_adam_step(float* p, float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, int step,
                    float grad_scale, int zero_grad, void* stream);
/* Appearance embedding (--encode_appearance; s-nerf/model/models.py:63-64, 153-159: condition = cat([view encoding, emb(rays.app)])):
 * snerf_app_embed writes emb[(int)app[ray], :dim] (fp32 table [n_vocab, dim], dim = 48 in the reference) into columns [0, dim) of dst
 * rows ray * S + i (dst: the condition block right of the 27 view-encoding columns; dtype fp32 / bf16; sample_id / rows as in
 * snerf_mip_viewenc); indices are clamped to the table.  snerf_app_embed_bwd accumulates d loss / d emb.weight (+=) from the fp32
 * gradient of those columns, dV [n_rays * S, ld]. */
int snerf_app_embed(const float* emb, const float* app, int n_vocab, long n_rays, int S, int dim, void* dst, long ld, int dtype,
                    const int* sample_id, long rows, void* stream);
int snerf_app_embed_bwd(const float* dV, long ld, const float* app, int n_vocab, long n_rays, int S, int dim, float* g_emb, void* stream);
/* the same accumulation in a fixed order (one workgroup per table row, rays in order; no atomics): bit-reproducible run to run */
int snerf_app_embed_bwd_det(const float* dV, long ld, const float* app, int n_vocab, long n_rays, int S, int dim, float* g_emb, void* stream);
/* bad[0] += number of entries of the float index vector idx [n] that torch.nn.Embedding(n_rows, .) would refuse after .long() (outside
 * [0, n_rows) or not finite); the kernels themselves clamp.  The host polls the counter later and raises (no sync inside the step). */
int snerf_index_check(const float* idx, long n, int n_rows, int* bad, void* stream);

/* Split-bf16 operands (dtype SNERF_DT_BF16X3 of snerf_linear_fwd / snerf_linear_wgrad): a value x is carried as hi = bf16(x) and
 * lo = bf16(x - hi) (16 mantissa bits) and a product evaluated as hi.hi + lo.hi + hi.lo on the bf16 MFMA path with fp32 accumulation
 * -- the arithmetic of the reference (fp32, s-nerf/model/models.py) to ~2^-16 relative per product at 1/3 of the bf16 rate instead
 * of 1/16 (the exact-fp32 MFMA).  Layout: activations [M, 2 K] with logical columns [64 j, 64 j + 64) at physical columns
 * [128 j, 128 j + 64) (hi) and [128 j + 64, 128 j + 128) (lo); weights [N, 3 K] = [hi_j | hi_j | lo_j] per 64 columns (built by
 * snerf_gather_pack: index bit 30 = the lo part).  snerf_linear_fwd then takes the LOGICAL K, writes bf16 outputs in the interleaved
 * layout (fp32 outputs plainly), sums hi + lo into the bias gradient; snerf_linear_wgrad takes the PHYSICAL N = dZ columns and
 * K = X columns (n_valid / k_valid logical) and maps its output indices back.  snerf_split_cast converts fp32 rows ([M, C], zero
 * padded to Cpad % 64 == 0 logical columns) into that layout. */
int snerf_split_cast(const float* src, long ld_src, long M, int C, int Cpad, void* dst, long ld_dst, void* stream);

/* fp16 + fp8 split operands (dtype SNERF_DT_F16F8 of snerf_linear_fwd): the same fp32-class contract in TWO pass-equivalents.  x = hi + r,
 * hi = fp16(x); a product is hi.hi on the fp16 MFMA plus the correction r.w + x.(w - fp16(w)), whose factors need 4 significant bits only and run as
 * OCP e4m3 on the block-scaled MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, twice the 16-bit rate; its E8M0 scale bytes undo the operands' power-of-two
 * scales).  Layout per 64 logical columns, 256 bytes: activations [fp16(x) x 64 | e4m3((x - hi) 2^13) x 64 | e4m3(x 2^2) x 64], weights
 * [fp16(w) x 64 | e4m3(w 2^9) x 64 | e4m3((w - hi) 2^20) x 64] (2-byte element strides, K the LOGICAL reduction length; e4m3 saturates at +-448:
 * |x| > 112 or |w| > 0.875 lose the clipped part of the correction only).  Forward activations only: ACT_NONE / ACT_RELU / ACT_RELU_BITS, no column
 * sums; the backward behind it runs on plain fp16 operands (bit 14 of the two GEMM entries' `variant` reads the fp16 halves).  Measured on fitted
 * weights: profiles/r6_c_two_pass_precision_with_split_schemes.txt.  snerf_split8_cast converts fp32 rows into either layout (weight = 0 / 1). */
#define SNERF_DT_F16F8 5
int snerf_split8_cast(const float* src, long ld_src, long M, int C, int Cpad, void* dst, long ld_dst, int weight, void* stream);

int snerf_colsum_f32(const float* x, long ld, long M, int C, float* out, void* stream);
int snerf_cast_pad(const float* src, long ld_src, long M, int C, int Cpad, void* dst, long ld_dst, int dtype, void* stream);
/* Refresh of the packed GEMM / fused-MLP operands after an optimiser step (the reference has no counterpart: torch.nn.Linear reads its
 * weights in place): dst[i] = flat[idx[i]] rounded to `dtype` (0 = fp32, 1 = bf16); idx -1 -> 0 (padding), -2 -> 1 (identity rows).  flat =
 * the fp32 parameter arena; idx (16-byte aligned) is built once per network by the host from the same slicing code that defines the
 * layouts (snerf_amd/mlp.py: _Net._build_plan). */
int snerf_gather_pack(const float* flat, const int* idx, long n, void* dst, int dtype, void* stream);
/* The same refresh with the transposed images (a packed W^T reads the arena at a stride of one weight row per element: one L2 request per
 * element in the gather) taken as 16 x 64 destination tiles: tiles int32 [n_tiles, 4] = {dst_off, src_base, src_stride, dst_ld} with
 * dst[dst_off + i * dst_ld + j] = flat[src_base + j * src_stride + i], i < 16, j < 64; src_base and src_stride multiples of 4, flat and tiles
 * 16-byte aligned.  The tiles' elements carry idx = -3 (skipped by the gather part of the launch).  One launch. */
int snerf_gather_pack_tiles(const float* flat, const int* idx, long n, void* dst, int dtype, const int* tiles, int n_tiles, void* stream);
/* snerf_gather_pack_tiles of a network's 16-bit operand pool and the plain gather of its (small) fp32 pool -- biases, fused-kernel tables:
 * dst32[i] = flat[idx32[i]], same index codes -- as ONE launch (tiles / n_tiles may be NULL / 0; n32 = 0: the 16-bit pool alone). */
int snerf_gather_pack_pair(const float* flat, const int* idx, long n, void* dst, int dtype, const int* tiles, int n_tiles, const int* idx32,
                           long n32, float* dst32, void* stream);

/* Semantic compositing, both flavours of the reference.  softmax = 1: zipnerf NerfMLP (internal/models.py:594-597) +
 * internal/render.py:237-241: semantic [R,C] = sum_i detach(weights [R,S]) softmax(logits [R*S, ld][:, :C]) (logits = columns
 * 1..C of the density network's output x); backward: d_logits only (g_w_out, if given, is zeroed).  softmax = 0: live mip path
 * (s-nerf/model/mip.py:175-176): semantic = sum_i weights raw_semantic; backward: d_logits [R*S, ld_d][:, :C] = w g and
 * g_w_out [R,S] = sum_c g_c raw (the weights are part of the graph there).  logits fp32 or bf16; gradients fp32.  row_index
 * (forward, may be NULL): int32 [R,S] from snerf_ert_compact -- the logits of sample (r, i) are row row_index[r, i] of `logits`, -1 =
 * sample not evaluated (contributes nothing): semantic rendering on compacted rows (inference extension, not in the reference). */
int snerf_semantic_composite_fwd(const float* weights, const void* logits, long ld, int dtype, long R, int S, int C, int softmax,
                                 const int* row_index, float* sem, void* stream);
int snerf_semantic_composite_bwd(const float* weights, const void* logits, long ld, int dtype, const float* g_sem, long R, int S, int C,
                                 int softmax, float* d_logits, long ld_d, float* g_w_out, void* stream);

/* ---- early ray termination + sample compaction (inference; NOT in the reference: opt-in, error bounded by the proposal) -----
 * From the proposal histogram (s0 [N,S0+1], w0 [N,S0]) and the resampled fence posts s1 [N,S1+1]: W = cumulative proposal
 * weight (piecewise linear), T_i = 1 - W(s1[i]), m_i = W(s1[i+1]) - W(s1[i]); fine interval i is kept iff T_i > eps_t and
 * m_i > eps_w.  Outputs: row_index [N,S1] (row of the kept sample in the compacted arrays, -1 = skipped), sample_id [rows]
 * (= ray * S1 + i of every row, rows in ray-major order) and *total = number of rows (device).  masks (8*N*ceil(S1/64) bytes)
 * and counts (int32 [N]) are scratch.  snerf_mip_encode / snerf_mip_viewenc take `sample_id, n_rows` (NULL = all samples) and
 * snerf_mip_composite_fwd takes `row_index` (NULL = dense rows) to run the fine level on the compacted rows. */
int snerf_ert_compact(const float* s0, const float* w0, const float* s1, long N, int S0, int S1, float eps_t, float eps_w,
                      void* masks, int* counts, int* row_index, int* sample_id, long* total, void* stream);
/* Front-to-back early ray termination on the fine network's OWN densities (inference; exact bound): the fine level is evaluated in groups
 * of consecutive samples.  One call = after group [g0, g0 + G) was evaluated (prev_raw_d: its raw densities, row r of it = row_index -
 * prev_base; G = 0 / NULL for the first call), add its optical depth sum softplus(raw + density_bias) (t1 - t0) |d| to tau [N] (in/out,
 * zeroed by the caller), flag the rays with exp(-tau) > eps_t (flags [N]), and assign the next group [next_g0, next_g0 + next_G):
 * row_index [N, S1] (row_base + running row, or -1 for the rays that left) and sample_id [rows]; counts [N] scratch; total[0] = rows.
 * The samples never assigned keep the -1 the caller initialised row_index with; their weights sum to at most eps_t per ray. */
int snerf_ert_f2b_step(const float* prev_raw_d, long ld_den, long prev_base, const float* s1, const float* dirs, const float* near,
                       const float* far, long N, int S1, int g0, int G, int next_g0, int next_G, int transform_idx, float density_bias,
                       float eps_t, float* tau, int* flags, int* counts, int* row_index, int* sample_id, long row_base, long* total,
                       void* stream);
/* Front-to-back early ray termination of the CLASSIC path's fine pass (render_rays(ert=...), inference extension, not in the reference):
 * after the fine network evaluated the group [g0, g0 + G) of the rays `alive` (int32 [n], NULL = rays 0 .. n-1; rows of raw_g [n*G, C >= 4]
 * in that order), scatter the raw outputs into raw_full [N, S, C], multiply the rays' transmittances T [N] by exp(-sum relu(sigma) dz |d|)
 * (raw2outputs, run_nerf_helpers.py:394-414; rays [N, ld_rays]: o3, d3, ...; z_all [N, S] sorted), and compact the rays with T > eps_t
 * into alive_next (in order; none when g0 + G == S); *total (device) = their number.  keep: int32 [n], offs: int32 [ceil(n / 1024)] scratch. */
int snerf_classic_ert_step(const float* raw_g, int C, const int* alive, long n, const float* z_all, int S, const float* rays, long ld_rays,
                           int g0, int G, float eps_t, float* T, float* raw_full, int* keep, int* offs, int* alive_next, long* total,
                           void* stream);
/* pts [n, G, 3] = o + d z (render.py:354) and viewdirs [n, 3] (columns viewdir_col .. +2 of the ray rows; NULL = none) of the rows
 * (alive[i], g0 + k) -- the compacted rows of the next group. */
int snerf_classic_ert_points(const float* rays, long ld_rays, const float* z_all, int S, const int* alive, long n, int g0, int G,
                             float* pts, float* viewdirs, int viewdir_col, void* stream);

/* ---- the callers either side of the path --------------------------------------------------------
 * Ray generation: s-nerf/utils/sample_utils.py:286-345 get_rays_single_img (training = 0: whole frame / any pixels, half-pixel
 * centres) and :92-211 sample_single_img (training = 1: directions by run_nerf_helpers.get_rays_by_coord :300-312 without the
 * half-pixel offset, radii from the half-pixel grid), no-NDC branch.  coords int32 [N,2] = (row, col), or NULL for the N pixels
 * first_pixel.. in row-major order.  pose_host: HOST pointer to the [3,4] camera-to-world matrix (12 floats, passed by value
 * to the kernel).  near / far are the caller's bounds (the reference scales them by 0.9 / 1.1 before).  Outputs are the Rays
 * fields of sample_utils.py:11-13 (lossmult = 1 and app = 0 are the caller's constants). */
int snerf_pinhole_rays(const int* coords, long first_pixel, int W, int H, const float* pose_host, float cx, float cy, float fx,
                       float fy, int training, float near, float far, long N, float* origins, float* directions, float* viewdirs,
                       float* radii, float* near_out, float* far_out, void* stream);
/* Per-ray loss tail of s-nerf/train.py:150-208, value and gradients w.r.t. the renderer outputs in one pass: RgbLoss
 * (model/loss_factory.py:5-11) on rgb/tgt [N,3]; calc_depth_loss (model/confidence.py:209-224) = DepthLoss
 * (loss_factory.py:26-37; disparity 1: |1/p - 1/t|) on dist1 (fine) + coarse_mult * dist0 (coarse), rays with tdepth == 0 masked
 * out, times conf [N] (nullable), mean over the valid rays, times depth_lambda (tdepth NULL = no depth loss); ProposalLoss
 * (loss_factory.py:59-74) of the detached fine histogram (s_f [N,Pf], w_f [N,Pf-1]) against the coarse one (s_c [N,Sc+1],
 * w_c [N,Sc]) times prop_lambda (s_c NULL = off).  out[5] = {#valid depth rays, rgb loss, depth loss, proposal loss, their sum};
 * g_rgb [N,3], g_dist1 / g_dist0 [N], g_wc [N,Sc] = d(total loss)/d(input). */
int snerf_mip_loss_tail(const float* rgb, const float* tgt, const float* dist1, const float* dist0, const float* tdepth,
                        const float* conf, const float* s_f, const float* w_f, const float* s_c, const float* w_c, long N, int Pf,
                        int Sc, int disparity, float depth_lambda, float coarse_mult, float prop_lambda, float* out, float* g_rgb,
                        float* g_dist1, float* g_dist0, float* g_wc, void* stream);

/* zipnerf (path C) ray generation: s-nerfpp/zipnerf/internal/camera_utils.py:453-563 pixels_to_rays, perspective camera, no
 * distortion, no NDC.  pix_x / pix_y int32 [N] integer pixel coordinates, cam_idx int32 [N] (NULL = camera 0) into pixtocams
 * [ncam,3,3] (inverse intrinsics) and camtoworlds [ncam,3,4], all device pointers.  float64 arithmetic like numpy's there, one
 * rounding per output.  origins / directions / viewdirs / base_x / base_y [N,3], radii [N], imageplane [N,2] (nullable). */
int snerf_zip_pixels_to_rays(const int* pix_x, const int* pix_y, const int* cam_idx, const float* pixtocams, const float* camtoworlds,
                             int ncam, long N, float* origins, float* directions, float* viewdirs, float* radii, float* imageplane,
                             float* base_x, float* base_y, void* stream);
/* Per-ray loss tail of s-nerfpp/zipnerf/train.py:250-311, value and gradients w.r.t. the renderer outputs in one pass:
 * data term (internal/train_utils.py:62-90; mse = 0: Charbonnier sqrt(resid^2 + pad^2), 1: resid^2) weighted by lossmult [R]
 * (nullable = 1) and normalised by its sum; disparity L1 |1/(depth+1e-5) - 1/(1e-5+tdepth)| as a masked mean under dmask [R]
 * (x depth_lambda, train.py:252-255,277) and cmask [R] (x depth_lambda * com_mult, :260-272), both nullable, empty mask = 0;
 * semantic NLL -log(sem[r, labels[r]] + 1e-6) as a masked mean under smask (x sem_mult, :294-298; sem NULL = off);
 * anti_interlevel_loss (train_utils.py:132-164: stepfun.blur_stepfun :425-433 with pulse widths pw0 / pw1, math.sorted_interp_quad
 * :133-156) of the NeRF histogram (s2 [R,S2+1], w2 [R,S2], detached) against the proposal levels (s0/w0, s1/w1; either NULL = skip)
 * x inter_mult; lossfun_distortion (stepfun.py:297-307) on the NeRF level x dist_mult (s2 NULL = no regularisers).
 * out[11]: [0..3] = {3 sum lossmult, sum dmask, sum cmask, sum smask}, [4..10] = {data, mse, depth, d_complete, sem, interlevel,
 * distortion}, weights applied.  g_rgb [R,3], g_depth [R], g_sem [R,C], g_w0 [R,S0], g_w1 [R,S1] (interlevel), g_w2 [R,S2]
 * (distortion) = d(sum of the terms)/d(input). */
int snerf_zip_loss_tail(const float* rgb, const float* tgt, const float* lossmult, const float* depth, const float* tdepth,
                        const float* dmask, const float* cmask, const float* sem, const int* labels, const float* smask, int C,
                        const float* s0, const float* w0, int S0, const float* s1, const float* w1, int S1, const float* s2,
                        const float* w2, int S2, long R, int mse, float pad, float data_mult, float depth_lambda, float com_mult,
                        float sem_mult, float pw0, float pw1, float inter_mult, float dist_mult, float* out, float* g_rgb,
                        float* g_depth, float* g_sem, float* g_w0, float* g_w1, float* g_w2, void* stream);

/* Distance percentiles of compute_extras (internal/render.py:255-267 -> stepfun.weighted_percentile :329-339, math.sorted_interp
 * :88-107): tdist [R,S+1], weights [R,S], t_far [R] (the extra fence post that carries the background weight); ps_host: HOST array of
 * np <= 8 percentiles in % (the reference uses 5, 50, 95); out [R,np]. */
int snerf_zip_percentiles(const float* tdist, const float* weights, const float* t_far, long R, int S, const float* ps_host, int np,
                          float* out, void* stream);

/* Frame quantisation for the S-NeRF++ wire format (s-nerfpp/zipnerf/random_render_waymo_seq.py:214-227): rgb [P,3] -> u8 as
 * internal/utils.py:111-116 save_img_u8 (clip(nan_to_num(x), 0, 1) * 255, truncated); depth [P] -> u16 = depth * 256 / scale_factor
 * truncated (:218-219); sem [P, ld_sem >= C] -> label u8 = first argmax over the C classes (:222-223) and paint u8 [P,3] =
 * color_map[label] (color_map: device u8 [C,3]; :225).  Any of rgb / depth / sem may be NULL (skipped); paint_u8 may be NULL. */
int snerf_frame_quantize(const float* rgb, const float* depth, const float* sem, long ld_sem, int C, const void* color_map, long P,
                         float scale_factor, void* rgb_u8, void* depth_u16, void* label_u8, void* paint_u8, void* stream);

/* ---- image-space foreground composite of S-NeRF++ stage 1 (s-nerfpp/stage1_code/) -----------------------
 * All images are device uint8 [H,W,3] (masks / bands hold 0 or 255, tested as > 0), depth float [H,W] (the uint16 depth PNG / 256,
 * exactly representable), semantic uint8 [H,W]; P = H*W.
 * snerf_fg_paste: utils_render.py:826-1005 handle_occlusion_paste given fg_depth [P] = the mesh depth along each pixel's ray (what the
 * reference's ray tracer returns, :913-947; ignored with person = 1, where the reference uses -1 :940-941).  In place: a pixel with
 * mask[...,0] > 0 takes the foreground colour, fg depth and class_id when fg_depth < depth or the background class is 0, 1 or 8;
 * otherwise its mask is cleared.  counters (device int[2]) <- {#masked, #pasted}: occlusion = 1 - pasted / (masked + 1) (:1002-1003). */
int snerf_fg_paste(void* bg_im, const void* fg_im, void* mask_im, float* depth, void* semantic, const float* fg_depth, long P,
                   int class_id, int person, int* counters, void* stream);
/* utils_render.py:306-324 get_bound_im: band = cv2.dilate(mask[...,0]) XOR cv2.erode(mask[...,0]) with the r x r rect kernel (default
 * anchor r/2, border never wins), written to all 3 channels of bound_im as 0 / 255; mask_out (nullable, must not alias mask_im)
 * <- ip_utils.py:10-19 set_diff(mask_im, bound_im) as the caller does next (generate_images.py:152-153). */
int snerf_fg_bound(const void* mask_im, int H, int W, int r, void* bound_im, void* mask_out, void* stream);
/* In place over nbytes bytes: total_bound <- utils_render.py:338-361 fuse_bound(total_mask, total_bound, bound, mask), then
 * total_mask <- (mask | total_mask) * 255 (generate_images.py:160-161). */
int snerf_fg_accumulate(void* total_mask, void* total_bound, const void* bound, const void* mask, long nbytes, void* stream);
/* utils_render.py:327-335 fuse_bound_and_im: im[bound > 0] = 0, per byte, in place. */
int snerf_fg_blank(void* im, const void* bound, long nbytes, void* stream);

/* Hash-grid weight decay (s-nerfpp/zipnerf/internal/train_utils.py:184-203 hash_decay_loss; torch_scatter.segment_coo(param**2, idx,
 * reduce='mean').mean() per encoder): table / grad fp32 [rows, C] (grad accumulated: += 2 mult p / (rows_l L C)), offsets device int32
 * [L+1]; *loss (device, nullable) += mult * mean over (l, c) of the per-level mean of table^2. */
int snerf_hash_decay(const float* table, float* grad, const int* offsets, int L, int C, float mult, float* loss, void* stream);

/* Gradients of the encoders w.r.t. the rays (pose refinement, s-nerf/utils/sample_utils.py:410-435: autograd through
 * integrated_pos_enc mip.py:105-118, sample2enc / Jacobi_g / fn2 mip.py:343-395 and cast_rays / lift_gaussian mip.py:31-91).
 * dE fp32 [n_rays*S, ld >= 6*max_deg] = d loss / d IPE features; g_origins / g_directions [n_rays,3] are WRITTEN (one wave per ray,
 * deterministic).  Fence posts carry no ray gradient (level 0 constants; level 1 detached, mip.py:318). */
int snerf_mip_encode_bwd(const float* s_vals, const float* origins, const float* directions, const float* radii, const float* near,
                         const float* far, long n_rays, int S, int cone, int transform_idx, int max_deg, const float* dE, long ld,
                         float* g_origins, float* g_directions, void* stream);
/* snerf_mip_encode_bwd with the warp selected, the backward of snerf_mip_encode_warp: fn_idx 1 = the contraction (above), 0 = the
 * view-centred warp fn1 + Jacobi_f (mip.py:323-341, 367-369) around (vx, vy, vz); far_max = device scalar max(far) of the batch. */
int snerf_mip_encode_warp_bwd(const float* s_vals, const float* origins, const float* directions, const float* radii, const float* near,
                              const float* far, long n_rays, int S, int cone, int transform_idx, int max_deg, const float* dE, long ld,
                              float* g_origins, float* g_directions, int fn_idx, float vx, float vy, float vz, const float* far_max,
                              void* stream);
/* dV fp32 [n_rays*S, ld >= 3 + 6*deg] = d loss / d view-direction encoding (mip.py:12-21) -> g_viewdirs [n_rays,3] (written). */
int snerf_mip_viewenc_bwd(const float* viewdirs, long n_rays, int S, int deg, const float* dV, long ld, float* g_viewdirs, void* stream);

/* Inference of one zipnerf PROPOSAL level in a single launch: snerf_zip_encode_fwd for the single-channel grid (C = 1) followed by the
 * proposal MLP on the features still in registers -- density_layer = Linear(L, hidden) + ReLU + Linear(hidden, 1)
 * (internal/models.py:425-427, 481-519 with disable_rgb).  w1 [hidden, L], b1 [hidden], w2 [hidden], b2 [1]: fp32 device pointers (the
 * state_dict tensors prop_mlp_i.density_layer.{0,2}.{weight,bias}); round_bf16 = 1 (bf16) / 2 (fp16) reproduces that GEMM path's roundings
 * (features, weights, hidden layer), 0 = fp32 throughout.  raw_density [R*S] fp32. */
int snerf_zip_encode_prop_fwd(const float* tdist, const float* origins, const float* directions, const float* radii, const float* base_x,
                              const float* base_y, const float* deg_jitter, const void* table, const int* offsets, const int* grid_sizes,
                              long R, int S, int L, int n, int m, float Sl, int H, float std_scale, int table_dtype, const float* w1,
                              const float* b1, const float* w2, const float* b2, int hidden, int round_bf16, float* raw_density,
                              void* stream);

/* ---- fused register-resident MLPs (csrc/fmlp.hip) ----------------------------------------------------------------------------
 * One launch for a whole 256-wide network; activations stay in registers from the first layer to the heads, the weights stream
 * through LDS as MFMA fragments.  wstream: n_frags x 1 KiB fragments in the kernel's consumption order, bias: n_blocks x 32 floats
 * (both built by snerf_amd.mlp.fmlp_pack from the reference's state_dict tensors; layout in the kernel header).  bf16 operands,
 * fp32 accumulation.
 * snerf_fmlp_classic_fwd  = NeRF.forward behind run_network (s-nerf/model/run_nerf_helpers.py:103-126, 460-474) for D = 8, W = 256,
 *   skips = [4], use_viewdirs: E [M, ldE] = embedded points (63 values, zero padded to 64), VE [M, ldVE] = embedded view directions
 *   (27 -> 32) -> raw [M,4] = (rgb, sigma).  n_frags = 1184, n_blocks = 78.
 * snerf_fmlp_proposal_fwd = proposal.forward (s-nerf/model/models.py:316-325), 4 x 256: E [M, ldE] = IPE (96) -> raw_density [M].
 *   n_frags = 448, n_blocks = 33. */
int snerf_fmlp_classic_fwd(const void* E, long ldE, const void* VE, long ldVE, const void* wstream, long n_frags, const float* bias,
                           int n_blocks, float* raw, long M, void* stream);
int snerf_fmlp_proposal_fwd(const void* E, long ldE, const void* wstream, long n_frags, const float* bias, int n_blocks,
                            float* raw_density, long M, void* stream);
/* snerf_fmlp_classic_fwd with the two positional encodings (Embedder, run_nerf_helpers.py:22-52) computed inside the kernel: the whole
 * run_network (:460-474) is this ONE launch.  pts [M,3] fp32 sample positions (row = ray * S + sample), viewdirs [M / S, ldvd] fp32.
 * sin / cos are evaluated in revolutions (two-term exact-product reduction + v_sin_f32, abs. error a few 1e-7) and rounded to bf16
 * like the output of snerf_classic_embed, which remains the bit-exact fp32 statement of the encoding.  M < 2^31. */
int snerf_fmlp_classic_pts_fwd(const float* pts, const float* viewdirs, long ldvd, int S, const void* wstream, long n_frags,
                               const float* bias, int n_blocks, float* raw, long M, void* stream);
/* Training forward of the two networks (autograd of NeRF.forward / proposal.forward in the reference): the same launch, which also
 * stores what the backward pass reads -- the bf16 outputs of the hidden layers and the ReLU bit masks of the 256-wide ones.
 * acts / act_ld: HOST arrays of device pointers / row strides (elements) -- classic: 10 = pts_linears.0 .. .7 (256 wide),
 * feature_linear (256), views_linears.0 (128); proposal: 4 = layers.0 .. .3 (256 wide).  Pointers 16-byte aligned, strides multiples
 * of 8.  bits: HOST array of 9 (classic: pts_linears.0 .. .7, then views_linears.0) / 4 (proposal) device pointers to
 * 4 * 8 * ceil(M / 256) * 4 * 64 bytes each (views_linears.0: half that), written in the layout snerf_linear_fwd's act = 4 (mask
 * bits) reads.  The per-layer snerf_linear_fwd (data gradient) /
 * snerf_linear_wgrad consume all of it unchanged. */
int snerf_fmlp_classic_train_fwd(const void* E, long ldE, const void* VE, long ldVE, const void* wstream, long n_frags,
                                 const float* bias, int n_blocks, float* raw, void* const* acts, const long* act_ld,
                                 void* const* bits, long M, void* stream);
int snerf_fmlp_proposal_train_fwd(const void* E, long ldE, const void* wstream, long n_frags, const float* bias, int n_blocks,
                                  float* raw_density, void* const* acts, const long* act_ld, void* const* bits, long M,
                                  void* stream);

/* NeRF MLP of the zipnerf path at INFERENCE as one launch (s-nerfpp/zipnerf/internal/models.py:462-479, 586-703; waymo.gin branch, no GLO
 * vectors): F [M, ldF >= 64] grid features (columns 40.. zero) and D [M, ldD >= 16] direction encoding (columns 9.. zero) in `dtype`
 * (SNERF_DT_BF16 or SNERF_DT_F16) -> raw_rgb [M, ld_rgb >= 3] and raw_d [M, ld_d >= 1] fp32.  density_layer.0 (+ReLU), density_layer.2 (x;
 * raw density = its output 0 before rounding), lin_second_stage_0 on cat([x, D]) (+ReLU), lin_second_stage_1 on cat([h, x, D]) (+ReLU),
 * rgb_layer -- activations stay in registers.  x32 (optional, [M, ld_x >= 32] in `dtype`): channels 0..31 of x, of which the semantic head
 * reads 1..C (models.py:594-597).  wstream (464 fragments of 1 KiB) / bias (35 blocks of 32 floats): the packing of
 * snerf_amd.mlp.ZipNerfNet._pack_fused_infer. */
int snerf_fmlp_zip_fwd(const void* F, long ldF, const void* D, long ldD, const void* wstream, long n_frags, const float* bias, int n_blocks,
                       float* raw_rgb, long ld_rgb, float* raw_d, long ld_d, void* x32, long ld_x, long M, int dtype, void* stream);
/* ... and as the TRAINING forward: the same outputs plus what the backward reads -- acts[0] = H1 [M, >= 64], acts[1] = x, acts[2] = h
 * (lin_second_stage_0), acts[3] = H3 (lin_second_stage_1) [M, >= 256] each (in `dtype`, row strides act_ld, 16-byte aligned) and bits[0..2] =
 * the ReLU bit masks of H1 ([M, 64]), h and H3 ([M, 256]) in snerf_linear_fwd's SNERF_ACT_RELU_BITS layout. */
int snerf_fmlp_zip_train_fwd(const void* F, long ldF, const void* D, long ldD, const void* wstream, long n_frags, const float* bias, int n_blocks,
                             float* raw_rgb, long ld_rgb, float* raw_d, long ld_d, void* const* acts, const long* act_ld, void* const* bits,
                             long M, int dtype, void* stream);
/* ... and its BACKWARD data-gradient chain in one launch: d_rgb [M, ld_rgb >= 3], d_den [M, ld_den >= den_cols] fp32 (d raw density, then the
 * gradients of the den_cols - 1 <= 31 semantic logits) -> dz[0] = d pre-activation of lin_second_stage_1, dz[1] = of lin_second_stage_0, dz[2] = d x
 * ([M, >= 256] each), dz[3] = d pre-activation of density_layer.0, dz[4] = d grid features ([M, >= 64]) in `dtype` -- the operands of the five
 * weight-gradient GEMMs and of the table gradient; bits[0..2] as written by snerf_fmlp_zip_train_fwd; the bias gradients of lin_second_stage_1,
 * lin_second_stage_0, density_layer.2, density_layer.0 are ADDED to g_bias[0..3] (not bit-reproducible); ws: snerf_fmlp_zip_chain_ws_floats(M)
 * floats.  wstream (448 fragments): snerf_amd.mlp.ZipNerfNet._pack_fused_chain. */
long snerf_fmlp_zip_chain_ws_floats(long M);
int snerf_fmlp_zip_chain_bwd(const float* d_rgb, long ld_rgb, const float* d_den, long ld_den, int den_cols, const void* wstream, long n_frags,
                             void* const* bits, void* const* dz, const long* dz_ld, float* const* g_bias, float* ws, long ws_floats, long M,
                             int dtype, void* stream);
/* Colour head of the live mip path's NeRF MLP, fused (s-nerf/model/models.py:283-296: cat([bottleneck, view encoding]) ->
 * cond_layers.0 .. .2 (Linear 128 + ReLU) -> rgb_layer; hidden 1024, 27 view-encoding columns).  Replaces four snerf_linear_fwd
 * launches forward and the four data-gradient launches backward.
 * snerf_fcolour_fwd: CB [M, ldCB] bf16 = [bottleneck 1024 | view encoding 27 | zeros up to column 1056] -> raw_rgb [M,3] fp32.
 *   wstream (336 fragments; cond_layers.0 in k-major order) / bias (13 blocks) from snerf_amd.mlp.fmlp_pack.  acts / act_ld / bits:
 *   HOST arrays of 3 (or all NULL for inference): the three hidden activations ([M, >= 128] bf16) and their ReLU bit masks
 *   (8 * ceil(M / 256) * 2 * 64 words each, snerf_linear_fwd's mask-bit layout for N = 128), stored for the backward pass.
 *   variant: 0; bit 0 selects the alternative input read-ahead depth (tools/fcolour_probe.py).
 * snerf_fcolour_bwd: d_raw_rgb [M,3] fp32 -> dC[0..2] = d pre-activation of cond_layers.2, .1, .0 ([M, >= 128] bf16; the weight-
 *   gradient GEMMs read them) and dB = d pre-activation of the bottleneck layer [M, >= 1024] bf16.  bits[0..3] = the bit masks of
 *   cond_layers.2, .1, .0 and of the bottleneck (N = 1024); wstream (336 fragments) = fmlp_pack of the TRANSPOSED weights; the four
 *   layers' bias gradients are ADDED to g_bias[0..3] (128, 128, 128, 1024 floats) in a fixed order (bit-reproducible).
 *   ws: snerf_fcolour_bwd_ws_floats(M) floats. */
int snerf_fcolour_fwd(const void* CB, long ldCB, const void* wstream, long n_frags, const float* bias, int n_blocks, float* raw_rgb,
                      void* const* acts, const long* act_ld, void* const* bits, long M, int variant, void* stream);
long snerf_fcolour_bwd_ws_floats(long M);
int snerf_fcolour_bwd(const float* d_raw_rgb, const void* wstream, long n_frags, void* const* bits, void* const* dC, const long* dC_ld,
                      void* dB, long dB_ld, float* const* g_bias, float* ws, long ws_floats, long M, void* stream);

/* Data-gradient chains of the two 256-wide networks, fused (autograd of NeRF.forward, run_nerf_helpers.py:83-139, and of the mip
 * path's proposal MLP, s-nerf/model/models.py:237-262): d raw -> d pre-activation of every layer in ONE launch; replaces ten
 * (classic) / four (proposal) snerf_linear_fwd data-gradient launches.  Every step's output is stored for snerf_linear_wgrad.
 * net 0 (classic): d_raw [M,4] fp32 (d rgb, d alpha); dz[0] = d views_linears.0 ([M, >= 128] bf16), dz[1] = d feature_linear output
 *   ([M, >= 256]), dz[2..9] = d pts_linears.7 .. .0; bits[0..7] = masks of pts_linears.0 .. .7, bits[8] = of views_linears.0;
 *   wstream: 1104 fragments.  net 1 (proposal): d_raw [M] (d raw density); dz[0..3] = d layers.3 .. .0; bits[0..3] = masks of
 *   layers.0 .. .3; wstream: 400 fragments.  wstream = snerf_amd.mlp.fmlp_pack of the transposed weights in chain order.
 * The bias gradient of step i is ADDED to g_bias[i] (width of dz[i] floats); the partial sums meet in LDS atomics, so the result is
 * NOT bit-reproducible run to run (the deterministic mode keeps the per-layer launches).  ws: snerf_fchain_bwd_ws_floats(net, M). */
long snerf_fchain_bwd_ws_floats(int net, long M);
int snerf_fchain_bwd(int net, const float* d_raw, const void* wstream, long n_frags, void* const* bits, void* const* dz, const long* dz_ld,
                     float* const* g_bias, float* ws, long ws_floats, long M, void* stream);

/* ---- deterministic mode ("deterministic mode for parity tests") ------------------------------------------
 * The weight gradient normally lands in dW by fp32 atomics from the M slices (order varies run to run).  snerf_linear_wgrad_det makes
 * every slice store its partial tile into `ws` (snerf_linear_wgrad_ws_floats(...) floats) and folds them in slice order: bit-identical
 * gradients run to run.  snerf_colsum_f32_det is the single-workgroup form of snerf_colsum_f32; snerf_linear_fwd takes variant bit 8
 * (| 256) to fold its bias-gradient partials in a fixed order. */
long snerf_linear_wgrad_ws_floats(int M, int N, int K, long ldz, long ldx, int dtype, int variant);
int snerf_linear_wgrad_det(const void* Z, long ldz, const void* X, long ldx, float* dW, long ldw, const void* zeros, int M, int N, int K,
                           int n_valid, int k_valid, int dtype, int variant, float* ws, long ws_floats, void* stream);
int snerf_colsum_f32_det(const float* x, long ld, long M, int C, float* out, void* stream);

/* ---- classic-path ray front end ---------------------------------------------------------------------
 * snerf_classic_get_rays  = get_rays (s-nerf/model/run_nerf_helpers.py:247-258): pinhole rays of an H x W frame, pixel centres at
 *   +0.5, principal point (cx, cy) = `ori_points` (default W/2, H/2); c2w_host: HOST pointer to the [3,4] camera-to-world matrix.
 * snerf_classic_ndc_rays  = ndc_rays (:314-332).
 * snerf_classic_ray_batch = the front half of render() (s-nerf/model/render.py:50-77) in one launch: rays from c2w_host (whole
 *   frame, n = H*W) or the given rays_o / rays_d [n,3]; unit view directions from the pre-NDC directions; c2w_static_host
 *   (c2w_staticcam) replaces the rays but not the view directions; NDC warp with near = 1; rows [n, ld] =
 *   [o3, d3, near, far, (depth), (viewdir3)] exactly as render_rays reads them (render.py:326-329). */
int snerf_classic_get_rays(int H, int W, double focal, double cx, double cy, const float* c2w_host, float* rays_o, float* rays_d, void* stream);
int snerf_classic_ndc_rays(int H, int W, double focal, float near, const float* rays_o, const float* rays_d, long n, float* o_out,
                           float* d_out, void* stream);
int snerf_classic_ray_batch(int H, int W, double focal, double cx, double cy, const float* c2w_host, const float* c2w_static_host,
                            const float* rays_o, const float* rays_d, long n, int ndc, float near, float far, const float* depths,
                            int use_viewdirs, float* rows, int ld, void* stream);

/* snerf_zip_encode_bwd without the atomic wall ("binned" table gradient): the contributions of gridencoder.cu:248-340 are written
 * out as records partitioned by destination (level, range of 4096 (C = 4) / 16384 (C = 1) table rows, replica), then one workgroup per
 * bin accumulates its records in LDS with 64-bit fixed-point integer atomics (scale from snerf_zip_bin_scale) and writes its rows back: no L2
 * atomics on the hashed levels, and a gradient that is BIT-IDENTICAL run to run (integer addition is order-independent).
 * Three calls: pass 0 counts (counts [L,1024] int32, zeroed by the caller) and reserves every workgroup's range inside the bins it
 * touches (wg_offsets: uint32 [L, ceil(R*S/256), 1024], need not be initialised); the caller scans the counts into `starts`
 * ([L,1024] int64, exclusive prefix sums over the flattened bins); pass 1 writes the records (rec_row uint16 [capacity], rec_val fp32
 * [capacity, max(C, 2)] -- for C = 1 a record is one 8-byte {row, value} pair in rec_val and rec_row is not touched; capacity >=
 * R*S*n*8*L is always enough) -- staged per workgroup in LDS and written run by run, so that a store instruction covers consecutive
 * records of ONE (workgroup, bin) range instead of 64 different lines (n <= 8; pass 3 is the same write with every thread storing its
 * records where they fall: more than 8 multisamples, A/B probes); pass 2 accumulates into grad_table (fp32, +=).  ksplit_host: HOST int[L],
 * replicas per row range (> 1 for levels with few, hot rows: they meet in g64, an int64 image of table rows [0, g64_rows) zeroed by
 * the caller); level_rows_host: HOST int[L], table rows per level -- a level needs ceil(rows / 4096 or 16384) * ksplit <= 1024 bins,
 * anything larger is refused with a bad-argument status (use snerf_zip_encode_bwd, the atomic scatter). */
int snerf_zip_encode_bwd_binned(int pass, const float* tdist, const float* origins, const float* directions, const float* radii,
                                const float* base_x, const float* base_y, const float* deg_jitter, const int* offsets,
                                const int* grid_sizes, const void* grad_feat, long ld, float* grad_table, long R, int S, int L, int C, int n,
                                int m, float Sl, int H, float std_scale, int feat_dtype, const int* ksplit_host, const int* level_rows_host,
                                int* counts, void* wg_offsets, const long* starts, void* rec_row, float* rec_val, long capacity, void* g64,
                                long g64_rows, const int* scale_exp, void* stream);
/* out[c] += sum over the M rows of x[m, c] (fp32, any number of columns C <= ld): the bias gradients behind the per-ray rows of the GLO
 * branch (snerf_colsum_f32 serves the heads: C <= 8).  deterministic != 0: one fixed summation order. */
int snerf_colsum_wide_f32(const float* x, long ld, long M, int C, float* out, int deterministic, void* stream);
/* GLO modulation of the zipnerf NeRF MLP's bottleneck (internal/models.py:620-630; Model.num_glo_features > 0, configs/360_glo*.gin):
 * out[p, c] = X[p, c] * exp(SS[p / S, c]) + SS[p / S, B + c] for the B bottleneck columns of the R * S samples; SS [R, >= 2 B] fp32 =
 * (scale | shift) per ray, the output of lin_glo_0 / lin_glo_1 on the ray's GLO vector.  X / out in `dtype` (fp32 or bf16).
 * _bwd: dXm = d loss / d out -> dX = dXm * exp(scale) + d_head (fp32 [R * S, >= n_head]: gradients that enter columns < n_head of x
 * directly -- raw density and the semantic logits, models.py:511,594), dSS [R, >= 2 B] = (sum_s dXm * X * exp(scale) | sum_s dXm), and
 * dxsum [R, >= B] = per-ray column sums of the stored dX (their sum over the rays is the bias gradient of density_layer.2).  One
 * workgroup per ray, samples in order: deterministic. */
int snerf_zip_glo_modulate(const void* X, long ldx, const float* SS, long ldss, long R, int S, int B, void* out, long ldo, int dtype, void* stream);
int snerf_zip_glo_modulate_bwd(const void* dXm, long lddxm, const void* X, long ldx, const float* SS, long ldss, const float* d_head, long ldh,
                               int n_head, long R, int S, int B, void* dX, long lddx, float* dSS, long lddss, float* dxsum, long ldsum,
                               int dtype, void* stream);
/* The fixed-point scale of one binned launch: scale_exp (device int[2]) [0] = e such that 2^e * max |grad_feat[:rows, :cols]| lies in
 * [2^33, 2^34) -- the accumulation grid follows the magnitude of the gradient (a loss-scaled 1e-12 gradient keeps 34 bits below its
 * largest entry); pass 2 reads it.  [1] = the bits of that maximum; a NaN / +-Inf in grad_feat (an fp16 overflow) leaves 0x7f800000
 * there and pass 2 then writes a NaN into grad_table[0]: the launch's records are meaningless and a found-inf check must see it. */
int snerf_zip_bin_scale(const void* grad_feat, long ld, long rows, int cols, int feat_dtype, int* scale_exp, void* stream);

/* Featurisation backward to the RAYS -- `cal_input_grad` of the reference (internal/models.py:491 -> gridencoder/grid.py:65-89 ->
 * gridencoder.cu:199-244, 343-369), needed by the pose refinement of zipnerf/train.py:187-197: d loss / d (origins, directions, base_x,
 * base_y) [R,3] each (ACCUMULATED: the caller zeroes them) from grad_feat = d loss / d features [R*S, ld] (feat_dtype), through the
 * trilinear derivative of every level, the erf down-weighting's dependence on the contracted std, the contraction's Jacobian
 * (coord.py:51-63) and the multisample helix (render.py:129-168).  table: the gather table of snerf_zip_encode_fwd (table_dtype 0 =
 * fp32, 2 = fp16).  snerf_zip_composite_bwd's `g_dirs` (optional, [R,3], written) is d loss / d directions through the interval
 * lengths (t1 - t0) |d| (render.py:170-176). */
int snerf_zip_encode_ray_bwd(const float* tdist, const float* origins, const float* directions, const float* radii, const float* base_x,
                             const float* base_y, const float* deg_jitter, const void* table, const int* offsets, const int* grid_sizes,
                             const void* grad_feat, long ld, long R, int S, int L, int C, int n, int m, float Sl, int H, float std_scale,
                             int table_dtype, int feat_dtype, float* g_origins, float* g_directions, float* g_base_x, float* g_base_y,
                             void* stream);

/* snerf_adam_step with the step count t in DEVICE memory: *step_dev is incremented, then used for the bias corrections -- no launch
 * argument depends on host state, so a whole training step can be captured in a hipGraph and replayed. */
int snerf_adam_step_dev(float* p, float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, int* step_dev,
                        float grad_scale, int zero_grad, void* stream);

/* The Adam step with the gradient hygiene of s-nerfpp/zipnerf/internal/train_utils.py:234-243 (clip_gradients, called every step at
 * zipnerf/train.py:336) folded into the same pass over the arena, in the reference's order: global-norm clip (clip_coef: device
 * scalar from snerf_grad_clip_coef, or NULL), value clip (grad_max_val > 0, torch.clamp semantics: NaN stays NaN), then `nonfinite`:
 * 0 = keep, 1 = NaN / +-Inf -> 0 (a poisoned gradient never reaches m, v or the parameters), 2 = torch.nan_to_num_ exactly (NaN -> 0,
 * +-Inf -> +-FLT_MAX).  step_dev != NULL: step count in device memory (incremented first); lr_dev != NULL: learning rate read from
 * device memory (a captured hipGraph then follows an lr schedule). */
int snerf_adam_step_ex(float* p, float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, int step,
                       int* step_dev, const float* lr_dev, float grad_scale, int zero_grad, int nonfinite, float grad_max_val,
                       const float* clip_coef, void* stream);
/* snerf_adam_step_ex with a counter: *dropped (device uint64, 8-byte aligned, never reset by the launch; NULL = off) += the number of
 * NaN / +-Inf gradient elements the launch saw, whatever `nonfinite` does with them.  The reference trains fp16 under GradScaler
 * (zipnerf/train.py:44,331), which skips such a step; with a static loss scale the drop would otherwise be silent. */
int snerf_adam_step_cnt(float* p, float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, int step,
                        int* step_dev, const float* lr_dev, float grad_scale, int zero_grad, int nonfinite, float grad_max_val,
                        const float* clip_coef, void* dropped, void* stream);
/* torch.nn.utils.clip_grad_norm_ coefficient over a flat gradient arena: out[0] = min(1, max_norm / (|grad_scale| * ||g||_2 + 1e-6)),
 * out[1] = the norm.  ws: >= 1024 doubles of device scratch.  Fixed reduction order (deterministic). */
int snerf_grad_clip_coef(const float* g, long n, float grad_scale, float max_norm, void* ws, float* out, void* stream);
/* The found-inf check of a dynamic loss scaler (torch.cuda.amp.GradScaler, which accelerate wraps around the reference's fp16 training:
 * s-nerfpp/zipnerf/train.py:44,215,331): *flag |= 1 when any of the n fp32 gradients is NaN or +-Inf.  The caller zeroes *flag
 * (device int); the binned table gradient marks an overflowed feature gradient by a NaN in its first element (snerf_zip_encode_bwd_binned). */
int snerf_nonfinite_flag(const float* g, long n, int* flag, void* stream);

#ifdef __cplusplus
}
#endif
#endif
