// MFMA GEMMs for the S-NeRF tiny-MLP layers (gfx950 / CDNA4).
//
//   snerf_linear_fwd : Y[M,n_store] = act( A[M,K] . W[N,K]^T + bias )        ("NT")
//                      used for every forward layer (reference: nn.Linear + ReLU,
//                      s-nerf/model/models.py:200-214, run_nerf_helpers.py:86-126) and,
//                      with W := W^T packed by the host, for the data gradient.
//   snerf_linear_wgrad: dW[N,K] += dZ[M,N]^T . X[M,K]   (fp32 atomics)         ("TN")
//
// Layout: every LDS tile row is 128 bytes (64 bf16 / 32 fp32 of the reduction
// axis) and is filled by `global_load_lds_dwordx4` (wave-uniform LDS base +
// lane*16).  The 16-byte chunk c of row r is stored at chunk position
// c ^ ((r>>1)&7); the permutation is applied on the per-lane GLOBAL source
// address (the LDS image stays lane-linear) and again on the ds_read_b128
// address, which makes every 16-lane read group of a 32x32 MFMA fragment hit
// 16 distinct 16-byte slots (conflict free).
//
// dtype f32 uses v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain: parity mode,
// 157 TF/s peak); dtype bf16 uses v_mfma_f32_32x32x16_bf16 (2.5 PF/s peak),
// fp32 accumulate in both.
#include "common.h"
#include <cstdlib>
#include <type_traits>

#define ACT_NONE 0
#define ACT_RELU 1
#define ACT_MASK 2  // y = (aux > 0) ? y : 0   (ReLU backward fused into dgrad)
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
#define ACT_RELU_BITS 3  // ReLU, and `aux` (uint32, OUTPUT) receives one bit per element: y > 0      (persistent kernel only)
#define ACT_MASK_BITS 4  // as ACT_MASK with `aux` = the bit mask an ACT_RELU_BITS launch of the same [M, N] wrote

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct GemmNT {
  const void* A; long lda;
  const void* W; long ldw;
  const float* bias;
  void* Y; long ldy;
  const void* aux; long ldaux;
  float* colsum;
  int M, N, K, n_store, act, out_f32, vec_store;
  float* colsum_ws; int fast_epi;
  int det;  // deterministic bias gradient: the per-slab column sums are folded by ONE thread per column (fixed order)
  int dbg;  // ablation bits, honoured only by a `make PROBE=1` build (tools/gemm_probe.py): 1 = no staging loads after tile 0, 2 = no LDS fragment reads, 4 = no stores
  int stagger;  // persistent kernel: workgroup start offsets, in units of KT x 64 clocks per slot (0 = all start together)
  // Split-bf16 mode (dtype SNERF_DT_BF16X3: the fp32-parity mode at MFMA-bf16 rates).  Every operand value x is the pair hi = bf16(x),
  // lo = bf16(x - hi) (16 mantissa bits), and a product is hi.hi + lo.hi + hi.lo (three bf16 MFMA passes, fp32 accumulation; lo.lo,
  // 2^-16 of 2^-16, is dropped).  Activations are stored k-tile interleaved -- logical columns [64 j, 64 j + 64) live at physical
  // columns [128 j, 128 j + 64) (hi) and [128 j + 64, 128 j + 128) (lo), so a logical column range at a multiple of 64 is a
  // contiguous physical range of twice the width -- and the weights as [hi_j | hi_j | lo_j] per k-tile, so that the reduction loop
  // simply walks 3 K / 64 VIRTUAL k-tiles: tile v reads W's physical tile v and A's physical tile 2 (v / 3) + (v % 3 == 1).
  // K in this struct is the virtual reduction length (3 x the logical one).
  //
  // split == 2 (dtype SNERF_DT_F16F8: the same contract in TWO pass-equivalents): x = hi + r with hi = fp16(x) (11 bits); the product is
  // hi.hi on the fp16 MFMA plus a correction r.w + x.(w - fp16(w)) whose operands need 4 bits only and run as e4m3 on the block-scaled
  // MFMA at twice the rate: activations [hi16 x 64 | e4m3(r 2^13) x 64 | e4m3(x 2^2) x 64] per 64 logical columns (256 bytes, the footprint
  // of the bf16 split layout), weights [hi16 x 64 | e4m3(w 2^9) x 64 | e4m3((w - hi) 2^20) x 64], so that virtual tile 2 j is the fp16 tile and
  // 2 j + 1 the 128-byte e4m3 tile of logical tile j for BOTH operands; the scale bytes of the MFMA take the 2^22 out again.  Values beyond
  // +-448 / scale saturate (|x| > 112, |w| > 0.875): the correction then loses what was clipped, i.e. the product falls back towards plain
  // fp16 for that element.  K = 2 x the logical reduction length.  Forward activations only (NONE / RELU / RELU_BITS, no column sums).
  int split;
  // Plain (bf16) data gradient behind a split-bf16 FORWARD (compute="bf16x3_fwd"): the ACT_MASK source `aux` is a saved activation in the
  // interleaved layout above -- logical column c is read at physical column (c >> 6) * 128 + (c & 63), its hi half.
  int aux_split;
};
__device__ __forceinline__ int split_a_tile(int v) { return 2 * (v / 3) + (v % 3 == 1 ? 1 : 0); }
#ifndef SNERF_PROBE
#define SNERF_PROBE 0   // the shipped library compiles the ablation branches out
#endif
#define DBG(p, bit) (SNERF_PROBE && ((p).dbg & (bit)))

template <typename T> struct Frag;
template <> struct Frag<float> { typedef f32x4 type; };
template <> struct Frag<__bf16> { typedef bf16x8 type; };

__device__ __forceinline__ void mma32(f32x16& acc, const bf16x8& a, const bf16x8& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma32(f32x16& acc, const f32x4& a, const f32x4& b) {
#pragma unroll
  for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
}

// fp16 flavour (dtype SNERF_DT_F16; BASELINE config 4's "fp16 MLP"): the kernels move their 16-bit operands through LDS and registers as opaque
// bit patterns (typed bf16x8 here), so a build flag F16 only selects the MFMA (v_mfma_f32_32x32x16_f16, same rate) and the fp32 <-> 16-bit
// conversions; ReLU by a packed signed max and the "is positive" tests on bit patterns hold for both formats.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <bool F16> __device__ __forceinline__ void mma32t(f32x16& acc, const bf16x8& a, const bf16x8& b) {
  if constexpr (F16) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
  else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}
template <bool F16> __device__ __forceinline__ void mma32t(f32x16& acc, const f32x4& a, const f32x4& b) { mma32(acc, a, b); }
// The CORRECTION tile of the fp16 + fp8 split mode (GemmNT::split == 2): 32 x 32 x 64 e4m3 products per instruction -- two 16-byte fragments
// per operand, read exactly like two bf16 fragments (the byte -> k map inside a lane is the same for both operands, so it does not matter)
// -- on the block-scaled MFMA with constant E8M0 scale bytes 116 = 2^-11 on both sides: the operands were scaled by 2^22 in total.
typedef int i32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void mma8(f32x16& acc, const bf16x8& a0, const bf16x8& a1, const bf16x8& b0, const bf16x8& b1) {
  const i32x8 a = __builtin_bit_cast(i32x8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15));
  const i32x8 b = __builtin_bit_cast(i32x8, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15));
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, 0, 0x74747474, 0, 0x74747474);
}
template <bool F16> __device__ __forceinline__ __bf16 cvt16(float v) {          // fp32 -> the launch's 16-bit format (as a bf16-typed bit pattern)
  if constexpr (F16) return __builtin_bit_cast(__bf16, (_Float16)v);
  else return (__bf16)v;
}
template <bool F16> __device__ __forceinline__ float up16(__bf16 h) {
  if constexpr (F16) return (float)__builtin_bit_cast(_Float16, h);
  else return (float)h;
}
template <bool F16> __device__ __forceinline__ float up16(float x) { return x; }
template <typename T, bool F16> __device__ __forceinline__ T down16(float v) {
  if constexpr (sizeof(T) == 4) return v;
  else return cvt16<F16>(v);
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__device__ __forceinline__ void glds16(const void* g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)lds_wave_base, 16, 0, 0);
}

// ---------------------------------------------------------------------------
// NT epilogue shared by the NT kernels.  acc[i][j] is the 32 x 32 MFMA tile at rows
// m0 + wm*WTM + 32 i, columns n0 + wn*WTN + 32 j of the block tile.
// ---------------------------------------------------------------------------
template <typename T, int BM, int BN, int WM, int WN, bool F16 = false>
__device__ __forceinline__ void nt_epilogue(const GemmNT& p, f32x16 (&acc)[BM / WM / 32][BN / WN / 32], char* smem,
                                            const int m0, const int n0, const int wave, const int lane) {
  typedef typename Frag<T>::type frag_t;
  constexpr int EPC = 16 / (int)sizeof(T);
  constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
  const int wm = wave / WN, wn = wave % WN;

  // epilogue: MFMA "A" operand = weights, "B" operand = activations, so D[i'][j'] has j' = lane&31 = local m and
  // i' = (r&3) + 8*(r>>2) + 4*(lane>>5) = local n: every lane owns ONE output row and, per register quad, FOUR consecutive
  // output columns.
  //
  // Fast path: 32-row x 128-byte slabs of the wave's tile are transposed through LDS (bias + activation + cast on the way
  // in) and leave as full 128-byte row segments, 16 bytes per lane (the same shape as the staging loads); the ReLU mask of
  // the data gradient is applied on the way out with 16-byte loads of the saved activation, and the bias gradient
  // (column sums) is reduced in registers + 3 shuffles and written, without atomics, to a [row-slab, N] workspace that a
  // second tiny kernel folds.
  if (p.fast_epi) {
    constexpr int CG = 128 / (int)sizeof(T);          // columns per 128-byte group
    constexpr int NCG = WTN / CG;                     // groups per wave tile (1 for bf16, 2 for fp32 at WTN = 64)
    constexpr int JPG = CG / 32;                      // 32-column MFMA tiles per group
    constexpr int PITCH = 144;                        // 128 + 16: keeps 16-byte alignment, staggers banks
    __syncthreads();                                  // every wave is done with the staging buffers
    char* my = smem + wave * (32 * PITCH);
    const T* __restrict__ auxp = (const T*)p.aux;
    const int prow = lane >> 3, pch = lane & 7;
    float cs[NCG][EPC];
#pragma unroll
    for (int g = 0; g < NCG; ++g)
#pragma unroll
      for (int e = 0; e < EPC; ++e) cs[g][e] = 0.f;
    // split-bf16 mode (bf16 only): two passes per slab -- hi = bf16(v), then lo = bf16(v - hi) -- to the two halves of the
    // 128-column physical group of these 64 logical columns; the mask source is the hi half of the saved activation
    const int parts = (sizeof(T) == 2 && p.split) ? 2 : 1;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int g = 0; g < NCG; ++g) {
        for (int part = 0; part < parts; ++part) {
        // phase 1: registers -> LDS slab [32 rows][CG columns]
#pragma unroll
        for (int jj = 0; jj < JPG; ++jj) {
          const int j = g * JPG + jj;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int cl = jj * 32 + 8 * q + 4 * (lane >> 5);                    // column inside the group
            float v[4];
            f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
            if (p.bias != nullptr) b4 = *(const f32x4*)(p.bias + n0 + wn * WTN + g * CG + cl);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[e] = acc[i][j][4 * q + e] + b4[e];
              if (p.act == ACT_RELU) v[e] = v[e] > 0.f ? v[e] : 0.f;
            }
            char* dst = my + (lane & 31) * PITCH + cl * (int)sizeof(T);
            if constexpr (sizeof(T) == 2) {
              bf16x4 o = {cvt16<F16>(v[0]), cvt16<F16>(v[1]), cvt16<F16>(v[2]), cvt16<F16>(v[3])};
              if (part == 1 && p.split == 2) {
                // fp16 + fp8 mode: the group's second 128 bytes = [e4m3(r 2^13) x 64 | e4m3(x 2^2) x 64], one byte per logical column
                char* d8 = my + (lane & 31) * PITCH + cl;
                *(unsigned*)d8 = snerf_e4m3x4(v[0] - up16<F16>(o[0]), v[1] - up16<F16>(o[1]), v[2] - up16<F16>(o[2]), v[3] - up16<F16>(o[3]), SNERF_F8_ACT_RES);
                *(unsigned*)(d8 + 64) = snerf_e4m3x4(v[0], v[1], v[2], v[3], SNERF_F8_ACT_VAL);
                continue;
              }
              if (part == 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (__bf16)(v[e] - (float)o[e]);
              }
              *(bf16x4*)dst = o;
            } else {
              f32x4 o = {v[0], v[1], v[2], v[3]};
              *(f32x4*)dst = o;
            }
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // phase 2: LDS -> global, 8 lanes per 128-byte row segment
        const int ncol = n0 + wn * WTN + g * CG + pch * EPC;
        const int pcol = parts == 2 ? ((ncol >> 6) << 7) + (ncol & 63) : ncol;     // physical column (hi half) of the logical column
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int row = it * 8 + prow;
          const int m = m0 + wm * WTM + i * 32 + row;
          frag_t val = *(const frag_t*)(my + row * PITCH + pch * 16);
          if (m < p.M && ncol < p.n_store && !DBG(p, 4)) {
            if (p.act == ACT_MASK) {
              const frag_t a8 = *(const frag_t*)(auxp + (long)m * p.ldaux + (p.aux_split ? ((ncol >> 6) << 7) + (ncol & 63) : pcol));
#pragma unroll
              for (int e = 0; e < EPC; ++e) if (!(up16<F16>(a8[e]) > 0.f)) val[e] = (T)0.f;
            }
            *(frag_t*)((T*)p.Y + (long)m * p.ldy + pcol + 64 * part) = val;
#pragma unroll
            for (int e = 0; e < EPC; ++e) cs[g][e] += up16<F16>(val[e]);
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
      }
    }
    if (p.colsum_ws != nullptr) {
      const int slab = (m0 / BM) * WM + wm;                                       // row of the workspace
#pragma unroll
      for (int g = 0; g < NCG; ++g) {
        const int ncol = n0 + wn * WTN + g * CG + pch * EPC;
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          float c = cs[g][e];
          c += __shfl_xor(c, 8, 64); c += __shfl_xor(c, 16, 64); c += __shfl_xor(c, 32, 64);
          if (lane < 8 && ncol + e < p.n_store) p.colsum_ws[(long)slab * p.N + ncol + e] = c;
        }
      }
    }
    return;
  }
  const T* __restrict__ aux = (const T*)p.aux;
  const bool vec_ok = p.vec_store != 0;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    float csum[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) csum[q][e] = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int nb = n0 + wn * WTN + j * 32 + 8 * q + 4 * (lane >> 5);
      float bv[4] = {0.f, 0.f, 0.f, 0.f};
      if (p.bias != nullptr) {
        const f32x4 b4 = *(const f32x4*)(p.bias + nb);      // bias is padded to N (multiple of 128)
        bv[0] = b4[0]; bv[1] = b4[1]; bv[2] = b4[2]; bv[3] = b4[3];
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * WTM + i * 32 + (lane & 31);
        if (m >= p.M || nb >= p.n_store || DBG(p, 4)) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = acc[i][j][4 * q + e] + bv[e];
          if (p.act == ACT_RELU) v[e] = v[e] > 0.f ? v[e] : 0.f;
        }
        const bool full = nb + 3 < p.n_store;
        if (p.act == ACT_MASK) {
          const T* ap = aux + (long)m * p.ldaux + (p.aux_split ? ((nb >> 6) << 7) + (nb & 63) : nb);
          if (full && vec_ok) {
            if constexpr (sizeof(T) == 2) {
              const bf16x4 a4 = *(const bf16x4*)ap;
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = up16<F16>(a4[e]) > 0.f ? v[e] : 0.f;
            } else {
              const f32x4 a4 = *(const f32x4*)ap;
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = a4[e] > 0.f ? v[e] : 0.f;
            }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (nb + e < p.n_store) v[e] = up16<F16>(ap[e]) > 0.f ? v[e] : 0.f;
          }
        }
        if (full && vec_ok) {
          if (p.out_f32 || sizeof(T) == 4) {
            f32x4 o = {v[0], v[1], v[2], v[3]};
            *(f32x4*)((float*)p.Y + (long)m * p.ldy + nb) = o;
          } else {
            bf16x4 o = {cvt16<F16>(v[0]), cvt16<F16>(v[1]), cvt16<F16>(v[2]), cvt16<F16>(v[3])};
            *(bf16x4*)((__bf16*)p.Y + (long)m * p.ldy + nb) = o;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (nb + e < p.n_store) {
              if (p.out_f32) ((float*)p.Y)[(long)m * p.ldy + nb + e] = v[e];
              else ((T*)p.Y)[(long)m * p.ldy + nb + e] = down16<T, F16>(v[e]);
            }
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) csum[q][e] += (nb + e < p.n_store) ? v[e] : 0.f;
      }
    }
    if (p.colsum != nullptr) {
      // column sums over this wave's rows: reduce across the 32 lanes of each half-wave (lanes = rows)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float c = csum[q][e];
#pragma unroll
          for (int o = 16; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
          const int n = n0 + wn * WTN + j * 32 + 8 * q + 4 * (lane >> 5) + e;
          if ((lane & 31) == 0 && n < p.n_store) atomicAdd(p.colsum + n, c);
        }
    }
  }
}

// ---------------------------------------------------------------------------
// NT kernel: BM x BN output tile, WM x WN waves, each wave (BM/WM) x (BN/WN).
// ---------------------------------------------------------------------------
template <typename T, int BM, int BN, int WM, int WN, bool F16 = false>
__global__ __launch_bounds__(64 * WM * WN) void gemm_nt_kernel(GemmNT p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef typename Frag<T>::type frag_t;
  constexpr int NW = WM * WN;
  constexpr int EPC = 16 / (int)sizeof(T);   // elements per 16-byte chunk
  constexpr int BKE = 128 / (int)sizeof(T);  // reduction elements per tile row
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
  constexpr int LA = BM / 8 / NW, LB = BN / 8 / NW;  // glds instructions per wave per tile
  static_assert(LA >= 1 && LB >= 1, "tile too small for the wave count");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int tiles_n = p.N / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int logical = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (logical / tiles_n) * BM, n0 = (logical % tiles_n) * BN;
  const T* __restrict__ A = (const T*)p.A;
  const T* __restrict__ W = (const T*)p.W;
  const int KT = p.K / BKE;

  // per-lane source coordinates of the staging loads (constant over k)
  const int lrow = lane >> 3, lsc = lane & 7;
  long a_off[LA], b_off[LB];
#pragma unroll
  for (int i = 0; i < LA; ++i) {
    const int row = (wave * LA + i) * 8 + lrow;
    const int c = lsc ^ ((row >> 1) & 7);
    int gr = m0 + row;
    gr = gr < p.M ? gr : p.M - 1;
    a_off[i] = (long)gr * p.lda + c * EPC;
  }
#pragma unroll
  for (int i = 0; i < LB; ++i) {
    const int row = (wave * LB + i) * 8 + lrow;
    const int c = lsc ^ ((row >> 1) & 7);
    b_off[i] = (long)(n0 + row) * p.ldw + c * EPC;
  }

  auto issue = [&](int kt, int stage) {
    char* sA = smem + stage * STAGE;
    char* sB = sA + BM * 128;
    const long k0 = (long)kt * BKE;
    const long ka = p.split == 1 ? (long)split_a_tile(kt) * BKE : k0;
#pragma unroll
    for (int i = 0; i < LA; ++i) glds16(A + a_off[i] + ka, sA + (wave * LA + i) * 1024);
#pragma unroll
    for (int i = 0; i < LB; ++i) glds16(W + b_off[i] + k0, sB + (wave * LB + i) * 1024);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment read offsets (bytes inside a tile), constant over k except the chunk index
  int ra[TM], rb[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) ra[i] = wm * WTM + i * 32 + (lane & 31);
#pragma unroll
  for (int j = 0; j < TN; ++j) rb[j] = wn * WTN + j * 32 + (lane & 31);
  const int chalf = lane >> 5;

  auto read_frags = [&](const char* sA, const char* sB, int ks, frag_t* a, frag_t* b) {
    const int c = 2 * ks + chalf;
#pragma unroll
    for (int i = 0; i < TM; ++i) a[i] = *(const frag_t*)(DBG(p, 2) ? smem + lane * 16 : sA + ra[i] * 128 + ((c ^ ((ra[i] >> 1) & 7)) << 4));
#pragma unroll
    for (int j = 0; j < TN; ++j) b[j] = *(const frag_t*)(DBG(p, 2) ? smem + lane * 16 + 1024 : sB + rb[j] * 128 + ((c ^ ((rb[j] >> 1) & 7)) << 4));
  };

  issue(0, 0);
  for (int kt = 0; kt < KT; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < KT && !DBG(p, 1)) issue(kt + 1, (kt + 1) & 1);
    const char* sA = smem + (kt & 1) * STAGE;
    const char* sB = sA + BM * 128;
    if constexpr (sizeof(T) == 2) {
      if (p.split == 2 && (kt & 1)) {                   // the e4m3 correction tile of the fp16 + fp8 mode
#pragma unroll
        for (int ks = 0; ks < 4; ks += 2) {
          frag_t a[2][TM], b[2][TN];
          read_frags(sA, sB, ks, a[0], b[0]);
          read_frags(sA, sB, ks + 1, a[1], b[1]);
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) mma8(acc[i][j], b[0][j], b[1][j], a[0][i], a[1][i]);
        }
        continue;
      }
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      frag_t a[TM], b[TN];
      read_frags(sA, sB, ks, a, b);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) mma32t<F16>(acc[i][j], b[j], a[i]);   // D = W-tile . X-tile^T: lanes own rows m
    }
  }
  nt_epilogue<T, BM, BN, WM, WN, F16>(p, acc, smem, m0, n0, wave, lane);
}

// ---------------------------------------------------------------------------
// NT kernel, 256 x 256 tile, bf16, 8 waves (2 x 4), BK = 64: the "8-phase" schedule.
//
// The block tile is staged as four HALF-tiles per k-tile (A0, A1, B0, B1: 128 rows x 128 B each, 16 KiB), double
// buffered (128 KiB).  The halves are interleaved over the waves so that every wave needs rows of every half and
// still owns a CONTIGUOUS 128 x 64 output tile (the epilogue is the common one):
//     A half h, local row lr  <->  tile row (lr>>6)*128 + h*64 + (lr&63)      (wave row wr = lr>>6)
//     B half g, local row lr  <->  tile col (lr>>5)*64  + g*32 + (lr&31)      (wave col wc = lr>>5)
// A k-tile is four phases, one 64 x 32 output quadrant (A half x B half, 8 MFMAs 32x32x16) each:
//     P1 A0.B0   P2 A0.B1   P3 A1.B1   P4 A1.B0
// and every phase stages ONE half-tile ahead (2 LDS-DMA pieces per wave):
//     P1 A1(t+1)   P2 A0(t+2)   P3 B0(t+2)   P4 B1(t+2), then s_waitcnt vmcnt(6)
// so three half-tiles are always in flight and the only VMEM wait of the loop (P4) retires everything tile t+1 needs.
// The two wave rows run half a phase apart (wr = 1 takes one extra barrier up front): while one wave of a SIMD issues
// its 8 MFMAs the other one fetches fragments and stages, and the two barriers per phase keep that alternation.
//
// Ordering rules (MI355X_MICROARCH.md, LDS-DMA): a staged half is read one phase (two barriers) after the vmcnt that
// retired it; a half is restaged at the earliest in the phase after its last fragment read, and every phase retires its
// fragment reads (lgkmcnt(0)) BEFORE its first barrier, so the restaging wave group cannot overtake them.
// ---------------------------------------------------------------------------
template <bool F16 = false>
__global__ __launch_bounds__(512) void gemm_nt8_kernel(GemmNT p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __bf16 T;
  constexpr int BM = 256, BN = 256, WM = 2, WN = 4;
  constexpr int HALF = 128 * 128;                     // bytes per half-tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int tiles_n = p.N / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int logical = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (logical / tiles_n) * BM, n0 = (logical % tiles_n) * BN;
  const T* __restrict__ A = (const T*)p.A;
  const T* __restrict__ W = (const T*)p.W;
  const int KT = p.K / 64;

  // staging sources: wave w owns pieces 2w, 2w+1 (8 rows x 128 B each) of every half-tile
  const int lrow = lane >> 3, lsc = lane & 7;
  long a_off[2][2], b_off[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int lr = (wave * 2 + i) * 8 + lrow;
    const int c = lsc ^ ((lr >> 1) & 7);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int gr = m0 + (lr >> 6) * 128 + h * 64 + (lr & 63);
      gr = gr < p.M ? gr : p.M - 1;
      a_off[h][i] = (long)gr * p.lda + c * 8;
      b_off[h][i] = (long)(n0 + (lr >> 5) * 64 + h * 32 + (lr & 31)) * p.ldw + c * 8;
    }
  }
  char* const my_piece = smem + wave * 2048;
  // which: 0 = A0, 1 = A1, 2 = B0, 3 = B1
  auto stage = [&](int kt, int db, int which) {
    char* dst = my_piece + (db * 4 + which) * HALF;
    const long k0 = (long)kt * 64;
    if (which < 2) {
      const long ka = p.split == 1 ? (long)split_a_tile(kt) * 64 : k0;
      glds16(A + a_off[which][0] + ka, dst);
      glds16(A + a_off[which][1] + ka, dst + 1024);
    } else {
      glds16(W + b_off[which - 2][0] + k0, dst);
      glds16(W + b_off[which - 2][1] + k0, dst + 1024);
    }
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment addresses inside a half-tile: row (wave part + lane&31), chunk (2 ks + lane>>5) ^ swizzle
  const int sw = ((lane & 31) >> 1) & 7;
  int cof[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) cof[ks] = ((2 * ks + (lane >> 5)) ^ sw) << 4;
  const int fa = (wr * 64 + (lane & 31)) * 128;       // + i2 * 32 * 128
  const int fb = (wc * 32 + (lane & 31)) * 128;

  bf16x8 a[2][4], b0[4], b1[4];
  auto read_a = [&](int db, int h) {
    const char* s = smem + (db * 4 + h) * HALF + fa;
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) a[i2][ks] = *(const bf16x8*)(s + i2 * 4096 + cof[ks]);
  };
  auto read_b = [&](int db, int g, bf16x8* b) {
    const char* s = smem + (db * 4 + 2 + g) * HALF + fb;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) b[ks] = *(const bf16x8*)(s + cof[ks]);
  };
  auto quadrant = [&](int h, int g, const bf16x8* b) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2) mma32t<F16>(acc[2 * h + i2][g], b[ks], a[i2][ks]);   // lanes own rows m
  };
  // end of a phase's load section / end of its MFMA section
  auto sync_loads = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto sync_mfma = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  if (DBG(p, 2) && blockIdx.x < 256) {
    // ablation: de-phase the CUs (first resident set of workgroups starts up to 7/8 of a tile time late)
    const int steps = ((blockIdx.x >> 3) & 7) * (p.K >> 3);
    for (int i = 0; i < steps; ++i) __builtin_amdgcn_s_sleep(8);
  }
  // prologue: all of tile 0, three halves of tile 1
  stage(0, 0, 0); stage(0, 0, 2); stage(0, 0, 3); stage(0, 0, 1);
  if (KT > 1) {
    stage(1, 1, 0); stage(1, 1, 2); stage(1, 1, 3);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();          // run half a phase behind wave row 0

  auto tile = [&](int t, auto db_tag) {
    constexpr int db = decltype(db_tag)::value;
    const bool more1 = t + 1 < KT && !DBG(p, 1), more2 = t + 2 < KT && !DBG(p, 1);
    // P1
    read_b(db, 0, b0);
    read_a(db, 0);
    if (more1) stage(t + 1, db ^ 1, 1);
    sync_loads();
    quadrant(0, 0, b0);
    sync_mfma();
    // P2
    read_b(db, 1, b1);
    if (more2) stage(t + 2, db, 0);
    sync_loads();
    quadrant(0, 1, b1);
    sync_mfma();
    // P3
    read_a(db, 1);
    if (more2) stage(t + 2, db, 2);
    sync_loads();
    quadrant(1, 1, b1);
    sync_mfma();
    // P4
    if (more2) {
      stage(t + 2, db, 3);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    sync_loads();
    quadrant(1, 0, b0);
    sync_mfma();
  };
  for (int t = 0; t < KT; t += 2) {
    tile(t, std::integral_constant<int, 0>{});
    if (t + 1 < KT) tile(t + 1, std::integral_constant<int, 1>{});
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();          // matches the extra barrier of wave row 1

  nt_epilogue<T, BM, BN, WM, WN, F16>(p, acc, smem, m0, n0, wave, lane);
}

// ---------------------------------------------------------------------------
// NT kernel, persistent 8-phase schedule ("variant 8"): 256 x 256 tiles, bf16, 8 waves, one workgroup per CU that walks
// its output tiles back to back.
//
// Same half-tile geometry and quadrant phases as gemm_nt8_kernel, with three changes aimed at what limited it
// (profiles/r1_f: the load section of a phase was longer than the 8-MFMA section it alternates with, and every output
// tile paid an exposed prologue and epilogue because only one workgroup fits a CU):
//  * fragments are PREFETCHED while the wave issues its MFMAs: the second B half during P1, A1 during P2 (each A0 fragment
//    is replaced in place right after its last MFMA), the first B half of t+1 during P3, A0(t+1) during P4; a load
//    section then holds only the staging DMA;
//  * the staging stream is continuous over (tile, k-tile): half-tiles are staged two k-tiles (8 halves) ahead in
//    consumption order Bfirst, A0, Bsecond, A1 and every phase ends with vmcnt(10) = five halves in flight; the next
//    tile's first k-tiles are therefore in LDS before the current tile is finished (no prologue after the first tile);
//  * the epilogue is cut into four 32-row units that ride in load sections: rows 0..63 of the wave tile are final after
//    P2 of the last k-tile and leave in its P3/P4, rows 64..127 leave in P1/P2 of the next tile's first k-tile, each
//    unit re-zeroing its accumulators.  The bias vector of the NEXT tile is DMA'd to LDS one tile ahead.
// The B half that is used first alternates with k-tile parity (even: B0, odd: B1) so that the two B register sets rotate.
// Ordering (MI355X_MICROARCH.md, LDS-DMA): a half retired by the vmcnt of load section p-1 is first read in MFMA section
// p (two barriers later for either wave row); a half last read in MFMA section p is restaged in load section p+2 at the
// earliest, after the lgkmcnt(0) of load section p+1 and its barrier.  Load sections of the two wave rows never overlap
// in time, so wave w and wave w+4 share one 4 KiB transposition slab.
// Requires N % 256 == 0, K >= 128, the fast (bf16, 16-byte aligned) epilogue.
// ---------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// SPLIT: 0 = plain operands, 1 = split-bf16 (three bf16 passes), 2 = fp16 + fp8 (GemmNT::split == 2: fp16 tiles and e4m3 correction tiles
// alternate -- virtual tile v is an fp16 tile for even v and lands in staging buffer v & 1, so the kind of a k-tile is its buffer index,
// a compile-time constant of each copy of the k-tile body)
template <int ACT, bool COLSUM, int SPLIT = 0, bool F16 = false, bool KT2 = false>
__global__ __launch_bounds__(512) void gemm_nt8p_kernel(GemmNT p) {
  static_assert(!(SPLIT && ACT == ACT_MASK), "split-bf16 data gradients take their ReLU masks from the bit masks");
  static_assert(SPLIT != 1 || !F16, "the three-pass split mode is a bf16 construction");
  static_assert(SPLIT != 2 || (F16 && !COLSUM && !KT2 && (ACT == ACT_NONE || ACT == ACT_RELU || ACT == ACT_RELU_BITS)),
                "fp16 + fp8: forward activations only (its backward runs on plain operands)");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __bf16 T;
  constexpr int HALF = 128 * 128;                      // bytes per half-tile
  constexpr int SCRATCH = 8 * HALF;                    // 4 x 4 KiB transposition slabs
  constexpr int BIAS = SCRATCH + 4 * 4096;             // 2 x 1 KiB bias vectors (tile parity)
  constexpr int MASKB = BIAS + 2048;                   // 8 waves x 4 units x 64 lanes x 4 B: ReLU bit masks of the current tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int tiles_n = p.N >> 8;
  const int tiles_m = (p.M + 255) >> 8;
  const int total = tiles_m * tiles_n;
  const int G = gridDim.x;
  const int n_my = (total - (int)blockIdx.x + G - 1) / G;      // >= 1 (the launch uses G <= total)
  const int KT = p.K >> 6;
  const long NT = (long)n_my * KT;
  const T* __restrict__ A = (const T*)p.A;
  const T* __restrict__ W = (const T*)p.W;

  auto origin = [&](int i, int& m0, int& n0) __attribute__((always_inline)) {
    const int logical = xcd_remap((int)blockIdx.x + i * G, total);
    m0 = (logical / tiles_n) << 8;
    n0 = (logical % tiles_n) << 8;
  };
  if constexpr (COLSUM) {
    // the column-sum partials accumulate into rows (2 blockIdx.x, 2 blockIdx.x + 1) of the workspace: the workgroup zeroes them itself,
    // long before its first flush (a whole tile later, behind that tile's barriers) -- instead of a memset launch in front of every
    // data-gradient GEMM (8 per training step).  The stores are older than every staging load, so the hand-counted vmcnt waits below only
    // get stricter while they are in flight.
    float* z = p.colsum_ws + (long)blockIdx.x * 2 * p.N;
    for (int i = tid; i < 2 * p.N; i += 512) z[i] = 0.f;
  }

  // ---- staging stream -------------------------------------------------------------------------------------------
  const int lrow = lane >> 3, lsc = lane & 7;
  int arel[2], brel[2];                                // byte offsets of this lane's 16 bytes inside the tile's rows
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int lr = (wave * 2 + i) * 8 + lrow;
    const int c = lsc ^ ((lr >> 1) & 7);
    arel[i] = (((lr >> 6) * 128 + (lr & 63)) * (int)p.lda + c * 8) * 2;
    brel[i] = (((lr >> 5) * 64 + (lr & 31)) * (int)p.ldw + c * 8) * 2;
  }
  // pieces 2 w and 2 w + 1 are 8 rows apart inside one 16-row group: same row block, swizzled chunk c ^ 4 (64 bytes up or down)
  const int rel_dc = ((lsc ^ ((((wave * 2) * 8 + lrow) >> 1) & 7)) & 4) ? -64 : 64;
  const int a_half = 64 * (int)p.lda * 2, b_half = 32 * (int)p.ldw * 2;
  __amdgpu_buffer_rsrc_t rA, rB;
  int s_i = 0, s_kt = 0;
  int s_ao = 0, s_m3 = 0;                              // SPLIT: byte offset of the A k-tile (hi, lo, hi of logical tile j = s_kt / 3), s_kt % 3
  auto stream_tile = [&](int i) __attribute__((always_inline)) {
    int m0, n0;
    origin(i < n_my ? i : n_my - 1, m0, n0);           // past the end: restage the last tile (never read)
    const int rows = min(p.M - m0, 256);
    rA = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (long)m0 * p.lda), 0, rows * (int)p.lda * 2, 0x00020000);
    rB = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (long)n0 * p.ldw), 0, 256 * (int)p.ldw * 2, 0x00020000);
  };
  auto stream_next = [&]() __attribute__((always_inline)) {
    if (++s_kt == KT) { s_kt = 0; s_ao = 0; s_m3 = 0; stream_tile(++s_i); }
    else if (SPLIT == 1) {                               // virtual tile v -> A's physical tile 2 (v / 3) + (v % 3 == 1): +1, -1, +2 tiles
      s_ao += s_m3 == 0 ? 128 : (s_m3 == 1 ? -128 : 256);
      s_m3 = s_m3 == 2 ? 0 : s_m3 + 1;
    }
  };
  // which: 0 = A0, 1 = A1, 2 = B0, 3 = B1 of the stream's k-tile; rows beyond M read as zeros (buffer bounds)
  auto stage = [&](int db, int which) __attribute__((always_inline)) {
    char* dst = smem + (db * 4 + which) * HALF + wave * 2048;
    const int soff = s_kt * 128;
    if (which < 2) {
      const int aoff = SPLIT == 1 ? s_ao : soff;            // (fp16 + fp8: A's physical tile IS the virtual tile, like W's)
      // SPLIT (at the register limit): the second piece's offset is derived from the first -- 8 rows further, 16-byte chunk c ^ 4 --
      // instead of being kept in a register of its own
      const int a1 = SPLIT ? arel[0] + 16 * (int)p.lda + rel_dc : arel[1];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr_t)dst, 16, arel[0] + which * a_half, aoff, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr_t)(dst + 1024), 16, a1 + which * a_half, aoff, 0, 0);
    } else {
      const int b1 = SPLIT ? brel[0] + 16 * (int)p.ldw + rel_dc : brel[1];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_ptr_t)dst, 16, brel[0] + (which - 2) * b_half, soff, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_ptr_t)(dst + 1024), 16, b1 + (which - 2) * b_half, soff, 0, 0);
    }
  };
  // bias of tile i -> LDS (one 1 KiB DMA by wave 0), double buffered on tile parity
  auto stage_bias = [&](int i) __attribute__((always_inline)) {
    if (wave == 0 && p.bias != nullptr && i < n_my) {
      int m0, n0;
      origin(i, m0, n0);
      const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bias + n0), 0, 1024, 0x00020000);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr_t)(smem + BIAS + (i & 1) * 1024), 16, lane * 16, 0, 0, 0);
    }
  };

  // ReLU bit masks (ACT_MASK_BITS): word `lane` of block (32-row block rb, 64-column group cg) holds, in byte it, the bits of
  // row 8 it + (lane >> 3), columns 8 (lane & 7) + e -- exactly what this lane masks in unit rb.  Each wave DMAs the four
  // words per lane it will need for tile i (wave-private, so its own vmcnt retirement is all the ordering there is).
  const int ncg = p.N >> 6;
  auto stage_mask = [&](int i) __attribute__((always_inline)) {
    if (ACT != ACT_MASK_BITS || i >= n_my) return;
    int m0, n0;
    origin(i, m0, n0);
    const long words = (long)(((p.M + 255) >> 8) * 8) * ncg * 64;
    const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void*)p.aux, 0, (int)(words * 4), 0x00020000);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int rb = (m0 >> 5) + wr * 4 + u, cg = (n0 >> 6) + wc;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rm, (lds_ptr_t)(smem + MASKB + (wave * 4 + u) * 256), 4, lane * 4, (rb * ncg + cg) * 256, 0, 0);
    }
  };

  // ---- fragments ------------------------------------------------------------------------------------------------
  const int sw = ((lane & 31) >> 1) & 7;
  int cof[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) cof[ks] = (lane & 31) * 128 + (((2 * ks + (lane >> 5)) ^ sw) << 4);
  const int fa = wr * 64 * 128, fb = wc * 32 * 128;
  bf16x8 aF[2][4], bS[2][4];                          // one A register set (replaced in place), two B sets
  auto lds_a = [&](int db, int h, int i2, int ks) __attribute__((always_inline)) {
    return *(const bf16x8*)(smem + (db * 4 + h) * HALF + fa + i2 * 4096 + cof[ks]);
  };
  auto lds_b = [&](int db, int g, int ks) __attribute__((always_inline)) {
    return *(const bf16x8*)(smem + (db * 4 + 2 + g) * HALF + fb + cof[ks]);
  };
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  int c_i = 0, c_kt = 0;                              // tile / k-tile the MFMAs are working on
  // ---- epilogue units --------------------------------------------------------------------------------------------
  char* const slab = smem + SCRATCH + (wave & 3) * 4096;
  const T* __restrict__ auxp = (const T*)p.aux;
  float cs[COLSUM ? 8 : 1];
#pragma unroll
  for (int e = 0; e < (COLSUM ? 8 : 1); ++e) cs[e] = 0.f;
  // unit u = rows [32u, 32u+32) of the wave tile of the tile at (em0, en0), bias buffer `par`
  auto unit = [&](int u, int em0, int en0, int par) __attribute__((always_inline)) {
    // everything lane-dependent is derived from a loop-variant spelling of the lane id, so that none of the unit's address
    // arithmetic is hoisted into (and kept alive across) the k-loop, where every register is spoken for
    int zero;
    asm volatile("s_lshr_b32 %0, %1, 30" : "=s"(zero) : "s"(c_kt));
    const int ln = lane | zero;
    const int hi = ln >> 5, row1 = ln & 31, prow = ln >> 3, pch = ln & 7;
    const int ncol = en0 + wc * 64 + pch * 8;
    const bool col_ok = ncol < p.n_store && !DBG(p, 4);
    const int mbase = em0 + wr * 128 + u * 32 + prow;
    unsigned mw = 0;                                     // ACT_MASK_BITS: this lane's 32 mask bits; ACT_RELU_BITS: the bits it produces
    if (ACT == ACT_MASK_BITS) {
      // the mask DMA was issued in P3 of the tile's first k-tile; with fewer than four k-tiles the vmcnt(10) of the load
      // sections in between has not necessarily retired it (wave-private data: this wave's own wait is all it takes)
      if (!SPLIT && KT < 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (split: K >= 128 logical columns = 6 virtual k-tiles)
      // read by inline asm: hipcc answers a ds_read of memory an LDS-DMA may have written with s_waitcnt vmcnt(0) -- a drain of the
      // whole staging stream per unit (13 per tile pass in the plain flavour; the split flavour's two passes re-materialised the read
      // into 43 of them: 9.2 ms instead of 2.7 per launch).  The word is waited for below, right before its first use.
      if (!SPLIT) asm volatile("ds_read_b32 %0, %1" : "=v"(mw) : "v"((unsigned)(size_t)(smem + MASKB + ((wave * 4 + u) * 64 + ln) * 4)));
    }
    // the ReLU mask source first: these loads sit behind the staging DMA in the (in-order) vmcnt queue
    bf16x8 a8[ACT == ACT_MASK ? 4 : 1];
    if (ACT == ACT_MASK) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        int m = mbase + it * 8;
        m = m < p.M ? m : p.M - 1;
        a8[it] = *(const bf16x8*)(auxp + (long)m * p.ldaux + (col_ok ? ncol : 0));
      }
    }
    // The accumulators were INITIALISED with the tile's bias (prologue / the previous tile's units), so a value leaves as it is: two
    // packed fp32 -> bf16 conversions per four values, ReLU as a packed signed 16-bit max on the rounded pair (a bf16 is negative iff
    // its bits are a negative int16; -0 -> +0), and the registers are re-initialised with the bias of the NEXT tile of this workgroup
    // (its buffer, the other parity, was staged at the start of the current tile; zeros without a bias; never used after the last tile).
    // Pass 1 reads the accumulators (pairs straight into v_cvt_pk_bf16_f32, no staging copies), pass 2 loads the next bias from LDS
    // INTO the accumulator registers (a ds_read_b128 per quad: no vector-ALU moves; without a bias the buffer holds zeros).
    const char* bl = smem + BIAS + (par ^ 1) * 1024 + wc * 256 + hi * 16;
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    // SPLIT: the unit runs twice over its accumulators -- part 0 emits hi = bf16(v), part 1 lo = bf16(v - hi) (zero where the ReLU
    // clamped v) -- to the two 64-column halves of the 128-column physical group of the wave's 64 logical columns; the accumulators
    // are re-initialised after the second pass.  The two slab round trips are serial (no registers to hold the lo values across the
    // first one), which the three-times-longer k-loop of this mode pays for.
    // physical column of this lane's 8 values (split: group g of 64 logical columns = physical columns [128 g, 128 g + 128))
    const int pcol = SPLIT ? 2 * (en0 + wc * 64) + pch * 8 : ncol;
    T* const yrow = (T*)p.Y + (long)mbase * p.ldy + pcol;      // one 64-bit address; the four row groups are 8 ldy apart
    const long ystep = 8 * p.ldy;
    constexpr bool AHEAD = ACT != ACT_MASK && !SPLIT;   // (the bf16-mask flavour and the split flavours have no registers to spare for this)
#pragma unroll
    for (int part = 0; part < (SPLIT ? 2 : 1); ++part) {
    if (SPLIT && part == 1) __builtin_amdgcn_sched_barrier(0);   // nothing of the second pass may be scheduled into the first (its values would be live across it)
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int jj = c >> 2, q = c & 3;
      const f32x2 v01 = {acc[u][jj][4 * q + 0], acc[u][jj][4 * q + 1]}, v23 = {acc[u][jj][4 * q + 2], acc[u][jj][4 * q + 3]};
      if constexpr (SPLIT == 2) {
        if (part == 1) {
          // fp16 + fp8: the second 128 bytes of the group, [e4m3(r 2^13) x 64 | e4m3(x 2^2) x 64], from the untouched accumulators (the values are
          // rounded and clamped once more: cheaper than carrying the first pass's results in registers this flavour does not have).  This lane's
          // four values are columns 8 c + 4 hi .. + 3 of the 64: bytes 8 c + 4 hi of either half = chunk c >> 1 (+ 4), offset 8 (c & 1) + 4 hi
          float x0 = v01[0], x1 = v01[1], x2 = v23[0], x3 = v23[1];
          if (ACT == ACT_RELU || ACT == ACT_RELU_BITS) { x0 = fmaxf(x0, 0.f); x1 = fmaxf(x1, 0.f); x2 = fmaxf(x2, 0.f); x3 = fmaxf(x3, 0.f); }
          const unsigned lo8 = snerf_e4m3x4(x0 - (float)(_Float16)x0, x1 - (float)(_Float16)x1, x2 - (float)(_Float16)x2, x3 - (float)(_Float16)x3, SNERF_F8_ACT_RES);
          const unsigned hi8 = snerf_e4m3x4(x0, x1, x2, x3, SNERF_F8_ACT_VAL);
          char* const d8 = slab + row1 * 128 + 8 * (c & 1) + 4 * hi;
          *(unsigned*)(d8 + (((c >> 1) ^ (row1 & 7)) << 4)) = lo8;
          *(unsigned*)(d8 + (((4 + (c >> 1)) ^ (row1 & 7)) << 4)) = hi8;
          continue;
        }
      }
      u32x2 o;
      if constexpr (F16) {
        typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
        o = u32x2{__builtin_bit_cast(unsigned, __builtin_convertvector(v01, f16x2)), __builtin_bit_cast(unsigned, __builtin_convertvector(v23, f16x2))};
      } else {
        o = u32x2{__builtin_bit_cast(unsigned, __builtin_convertvector(v01, bf16x2)), __builtin_bit_cast(unsigned, __builtin_convertvector(v23, bf16x2))};
      }
      if ((ACT == ACT_RELU || ACT == ACT_RELU_BITS) && part == 0) {
        // (the empty asm keeps the packed conversion; the max itself stays a compiler-visible instruction: see fmlp.hip to_frags)
        typedef short s16x4 __attribute__((ext_vector_type(4)));
        asm("" : "+v"(o));
        o = __builtin_bit_cast(u32x2, __builtin_elementwise_max(__builtin_bit_cast(s16x4, o), s16x4{0, 0, 0, 0}));
      }
      if (SPLIT == 1 && part == 0) {
        // the accumulators keep the RESIDUAL of the rounded (and clamped) value for the second pass -- hi's two bf16 are the high
        // halves of fp32 words; zero where the ReLU clamped -- so that pass costs no registers beyond the first one's
        const unsigned o0 = o[0], o1 = o[1];              // (named scalars: see the note at ACT_MASK_BITS)
        float r0 = v01[0] - __builtin_bit_cast(float, o0 << 16), r1 = v01[1] - __builtin_bit_cast(float, o0 & 0xffff0000u);
        float r2 = v23[0] - __builtin_bit_cast(float, o1 << 16), r3 = v23[1] - __builtin_bit_cast(float, o1 & 0xffff0000u);
        if (ACT == ACT_RELU || ACT == ACT_RELU_BITS) {
          r0 = v01[0] > 0.f ? r0 : 0.f; r1 = v01[1] > 0.f ? r1 : 0.f;
          r2 = v23[0] > 0.f ? r2 : 0.f; r3 = v23[1] > 0.f ? r3 : 0.f;
        }
        acc[u][jj][4 * q + 0] = r0; acc[u][jj][4 * q + 1] = r1; acc[u][jj][4 * q + 2] = r2; acc[u][jj][4 * q + 3] = r3;
      }
      *(u32x2*)(slab + row1 * 128 + ((c ^ (row1 & 7)) << 4) + 8 * hi) = o;
    }
    if (part == (SPLIT ? 1 : 0)) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const f32x4 b = *(const f32x4*)(bl + c * 32);           // columns 8c + 4 hi .. +3 of the next tile's bias
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[u][c >> 2][4 * (c & 3) + e] = b[e];
      }
    }
    // No wait between the slab writes and the read-back: the LDS executes one wave's instructions in order, so the four reads below
    // queue right behind the writes (and behind the bias reads) and their latencies overlap -- one exposed LDS round trip per unit
    // instead of five (the unit's length is what the other wave row's 8-MFMA section has to cover).  The empty asm pins the four
    // reads here: left alone, hipcc sinks each one into the store's branch, where it is waited for on its own.
    // (the split flavours fetch the word again in each pass instead of carrying it across the first one: they are at the register limit)
    if (ACT == ACT_MASK_BITS && SPLIT) asm volatile("ds_read_b32 %0, %1" : "=v"(mw) : "v"((unsigned)(size_t)(smem + MASKB + ((wave * 4 + u) * 64 + ln) * 4)));
    if (ACT == ACT_MASK_BITS && (SPLIT || part == 0)) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(mw));   // (LDS returns in order: the mask word is older than the slab read-backs)
    u32x4 vals[AHEAD ? 4 : 1];
    if constexpr (AHEAD) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + prow;
        vals[it] = *(const u32x4*)(slab + row * 128 + ((pch ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int it = 0; it < 4; ++it) asm volatile("" : "+v"(vals[it]));
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 8 + prow;
      const int m = mbase + it * 8;
      bf16x8 val;
      if constexpr (AHEAD) val = __builtin_bit_cast(bf16x8, vals[it]);
      else val = *(const bf16x8*)(slab + row * 128 + ((pch ^ (row & 7)) << 4));
      if (ACT == ACT_MASK) {
#pragma unroll
        for (int e = 0; e < 8; ++e) if (!(up16<F16>(a8[it][e]) > 0.f)) val[e] = (T)0.f;
      }
      if (ACT == ACT_MASK_BITS) {
        // byte `it` of the word masks these 8 values: u carries bit i of the byte at positions i and i + 15, so (u >> 2k) & 0x10001
        // has the bits of elements 2k / 2k + 1 in the low bit of each 16-bit half -- a packed multiply by 0 / 1 applies them
        const unsigned bt = (mw >> (8 * it)) & 0xffu, u = bt | (bt << 15);
        u32x4 raw = __builtin_bit_cast(u32x4, val);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          // (through named scalars: __builtin_bit_cast applied directly to a vector-element lvalue is miscompiled by this clang)
          const unsigned sel = (u >> (2 * k)) & 0x00010001u, x = raw[k];
          raw[k] = __builtin_bit_cast(unsigned, (u16x2)(__builtin_bit_cast(u16x2, x) * __builtin_bit_cast(u16x2, sel)));
        }
        val = __builtin_bit_cast(bf16x8, raw);
      }
      if (col_ok && m < p.M) {
        if (!DBG(p, 8)) *(bf16x8*)(yrow + it * ystep + (SPLIT ? 64 * part : 0)) = val;          // (non-temporal stores measure the same: profiles/r2_n)
        else asm volatile("" ::"v"(val));               // PROBE build, bit 8: everything but the store instruction itself
        if (ACT == ACT_RELU_BITS && part == 0) {
          // the value is a ReLU output (max(v, +0) rounded to bf16: never negative, never -0): > 0  <=>  its 16 bits are not all zero
          // min(half, 1) per 16-bit half = "is positive"; z gathers the even elements in bits 0, 2, 4, 6 and the odd ones 16 higher
          const u32x4 raw = __builtin_bit_cast(u32x4, val);
          // (the constant goes through an empty asm: knowing that it is 1, hipcc rewrites min(x, 1) as x != 0 and emits 16-bit
          // compares, selects and byte permutes -- 112 instead of 48 vector-ALU instructions per unit)
          unsigned one_bits = 0x00010001u;
          asm("" : "+v"(one_bits));
          const u16x2 one = __builtin_bit_cast(u16x2, one_bits);
          unsigned z = 0;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const unsigned x = raw[k];                      // (named scalar: see the note at ACT_MASK_BITS)
            z |= __builtin_bit_cast(unsigned, (u16x2)__builtin_elementwise_min(__builtin_bit_cast(u16x2, x), one)) << (2 * k);
          }
          mw |= ((z | (z >> 15)) & 0xffu) << (8 * it);
        }
        if constexpr (COLSUM) {
          // (not v_dot2c_f32_bf16 against (1, 0) / (0, 1), which would halve these instructions: tools/probes/dot2_probe.hip -- its
          // accumulate is not correctly rounded, inf x 0 poisons the neighbouring column, and hipcc 7.2 encodes the packed constant
          // (1, 0) as the inline constant 1.0, which the instruction reads as (0, 1))
#pragma unroll
          for (int e = 0; e < 8; ++e) cs[e] += up16<F16>(val[e]);
        }
      }
    }
    }
    if (ACT == ACT_RELU_BITS) {
      const long blk = (long)((em0 >> 5) + wr * 4 + u) * ncg + ((en0 >> 6) + wc);
      ((unsigned*)p.aux)[blk * 64 + ln] = mw;            // 256 contiguous bytes per wave
    }
  };
  // The column sums stay in registers across this workgroup's tiles: its tiles normally all lie in ONE column block (tile index
  // stride G is a multiple of tiles_n for the shapes of the step), so the partial sums are flushed once per workgroup -- row
  // (blockIdx.x, wave row) of the workspace (zeroed by the workgroup at its start) -- instead of once per tile.  A change of column block flushes early.
  auto flush_colsum = [&](int em0, int en0) __attribute__((always_inline)) {
    if constexpr (!COLSUM) return;
    (void)em0;
    int zero;
    asm volatile("s_lshr_b32 %0, %1, 30" : "=s"(zero) : "s"(c_kt));
    const int ln = lane | zero;
    const int pch = ln & 7;
    const int ncol = en0 + wc * 64 + pch * 8;
    const long srow = (long)((int)blockIdx.x * 2 + wr) * p.N;
#pragma unroll
    for (int e = 0; e < (COLSUM ? 8 : 0); ++e) {
      float c = cs[e];
      c += __shfl_xor(c, 8, 64); c += __shfl_xor(c, 16, 64); c += __shfl_xor(c, 32, 64);
      if (ln < 8 && ncol + e < p.n_store) p.colsum_ws[srow + ncol + e] += c;
      cs[e] = 0.f;
    }
  };

  // end of a load section (retire the DMA five halves back and this wave's LDS traffic) / of an MFMA section
  auto end_load = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto end_mfma = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- staggered start, a probe (variant bits 9..12, off by default; tools/gemm_stagger_probe.py): equal work per workgroup keeps the
  // tile boundaries of all CUs in step, so the chip's row stores arrive in bursts -- starting the workgroups of an XCD up to 7/8 of a
  // tile apart changed nothing on any shape (profiles/r2_n), i.e. the bursts are not what the stores cost
  if (p.stagger) {
    const int n = (((int)blockIdx.x >> 3) & 7) * KT * p.stagger;
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1);
  }
  // ---- prologue: k-tiles 0 and 1 of the stream, bias of tiles 0 and 1 --------------------------------------------
  if (p.bias == nullptr && tid < 128) *(f32x4*)(smem + BIAS + tid * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
  stream_tile(0);
  stage_bias(0);
  stage_mask(0);
  stage(0, 2); stage(0, 0); stage(0, 3); stage(0, 1);
  stream_next();
  stage(1, 3); stage(1, 0); stage(1, 2); stage(1, 1);
  stream_next();
  asm volatile("s_waitcnt vmcnt(10)" ::: "memory");     // B0, A0, B1 of k-tile 0 (and the bias) have landed
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    aF[0][ks] = lds_a(0, 0, 0, ks);
    aF[1][ks] = lds_a(0, 0, 1, ks);
    bS[0][ks] = lds_b(0, 0, ks);
  }
  {                                                      // accumulators start from the bias of tile 0 (see the epilogue units)
    const char* bl0 = smem + BIAS + wc * 256 + (lane >> 5) * 16;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const f32x4 b = *(const f32x4*)(bl0 + c * 32);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][c >> 2][4 * (c & 3) + e] = b[e];
    }
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();            // wave row 1 runs half a phase behind wave row 0

  int m0, n0;
  origin(0, m0, n0);
  int em0 = 0, en0 = 0, epar = 0;
  bool pending = false, pending_mask = false;

  auto ktile = [&](auto db_tag) __attribute__((always_inline)) {
    constexpr int db = decltype(db_tag)::value;
    constexpr int F = db, S = 1 - db;                   // B half used first / second in this k-tile
    constexpr bool F8T = SPLIT == 2 && db == 1;         // fp16 + fp8: the odd virtual tiles are the e4m3 correction tiles
    const bool last = c_kt == KT - 1;
    // P1: A0 x B_F; fetch B_S of this k-tile
    stage(db, 2 + F);
    // K = 128 (two k-tiles per tile): the next tile's bias (DMA'd in P4 of k-tile 0) and this tile's mask words (P3 of k-tile 0) are read by
    // unit 0 in P3 of THIS k-tile, six to nine DMAs later -- fewer than the ten the stream's vmcnt(10) leaves in flight, so nothing orders
    // them (round 4: about one launch in 200 took a stale bias into one 32 x 64 block when another persistent GEMM had run on the CU
    // before: tools/probes/gemm_k128_bias_race.py).  Retire everything but the two DMAs just issued; this section's barrier publishes
    // wave 0's bias to the other waves.  With three or more k-tiles the ordinary waits cover both.  KT2 is a template flag of the K = 128
    // launches (the wait as a run-time branch in this section cost every other launch 1-2 %: 25.93 -> 26.25 ms per step).
    if constexpr (KT2) { if (last) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
    if (pending) unit(2, em0, en0, epar);
    end_load();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) bS[S][ks] = lds_b(db, S, ks);
    if constexpr (F8T) {
#pragma unroll
      for (int ks = 0; ks < 4; ks += 2) {
        mma8(acc[0][F], bS[F][ks], bS[F][ks + 1], aF[0][ks], aF[0][ks + 1]);
        mma8(acc[1][F], bS[F][ks], bS[F][ks + 1], aF[1][ks], aF[1][ks + 1]);
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
    } else {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      mma32t<F16>(acc[0][F], bS[F][ks], aF[0][ks]);
      mma32t<F16>(acc[1][F], bS[F][ks], aF[1][ks]);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    }
    }
    end_mfma();
    // P2: A0 x B_S; every A0 fragment is replaced by the A1 fragment of the same position right after its last use
    stage(db, 0);
    if (pending) { unit(3, em0, en0, epar); if (en0 != n0) flush_colsum(em0, en0); pending = false; }
    end_load();
    if constexpr (F8T) {
#pragma unroll
      for (int ks = 0; ks < 4; ks += 2)
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {
          mma8(acc[i2][S], bS[S][ks], bS[S][ks + 1], aF[i2][ks], aF[i2][ks + 1]);
          aF[i2][ks] = lds_a(db, 1, i2, ks);
          aF[i2][ks + 1] = lds_a(db, 1, i2, ks + 1);
        }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      }
    } else {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2) {
        mma32t<F16>(acc[i2][S], bS[S][ks], aF[i2][ks]);
        aF[i2][ks] = lds_a(db, 1, i2, ks);
      }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    }
    end_mfma();
    // P3: A1 x B_S; B_S is replaced by the first B half of the next k-tile
    stage(db, 2 + S);
    if (last) unit(0, m0, n0, c_i & 1);
    if (pending_mask) { stage_mask(c_i); pending_mask = false; }   // its buffer was last read by units 2, 3 of the previous tile (P1, P2)
    end_load();
    if constexpr (F8T) {
#pragma unroll
      for (int ks = 0; ks < 4; ks += 2) {
        mma8(acc[2][S], bS[S][ks], bS[S][ks + 1], aF[0][ks], aF[0][ks + 1]);
        mma8(acc[3][S], bS[S][ks], bS[S][ks + 1], aF[1][ks], aF[1][ks + 1]);
        bS[S][ks] = lds_b(db ^ 1, S, ks);
        bS[S][ks + 1] = lds_b(db ^ 1, S, ks + 1);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      }
    } else {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      mma32t<F16>(acc[2][S], bS[S][ks], aF[0][ks]);
      mma32t<F16>(acc[3][S], bS[S][ks], aF[1][ks]);
      bS[S][ks] = lds_b(db ^ 1, S, ks);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    }
    end_mfma();
    // P4: A1 x B_F; A1 is replaced by A0 of the next k-tile
    stage(db, 1);
    stream_next();
    if (last) unit(1, m0, n0, c_i & 1);
    if (c_kt == 0) stage_bias(c_i + 1);                 // its buffer was last read by units 2, 3 of tile c_i - 1 (P1, P2)
    end_load();
    if constexpr (F8T) {
#pragma unroll
      for (int ks = 0; ks < 4; ks += 2)
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {
          mma8(acc[2 + i2][F], bS[F][ks], bS[F][ks + 1], aF[i2][ks], aF[i2][ks + 1]);
          aF[i2][ks] = lds_a(db ^ 1, 0, i2, ks);
          aF[i2][ks + 1] = lds_a(db ^ 1, 0, i2, ks + 1);
        }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      }
    } else {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2) {
        mma32t<F16>(acc[2 + i2][F], bS[F][ks], aF[i2][ks]);
        aF[i2][ks] = lds_a(db ^ 1, 0, i2, ks);
      }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    }
    end_mfma();
    if (last) {
      pending = true; pending_mask = true; em0 = m0; en0 = n0; epar = c_i & 1;
      c_kt = 0; ++c_i;
      origin(c_i < n_my ? c_i : n_my - 1, m0, n0);
    } else {
      ++c_kt;
    }
  };
  for (long t = 0; t < NT; t += 2) {
    ktile(std::integral_constant<int, 0>{});
    if (t + 1 < NT) ktile(std::integral_constant<int, 1>{});
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();            // matches the extra barrier of wave row 1
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // drain the read-ahead of the stream before the LDS is released
  // rows 64..127 of the last tile (wave rows take turns on the shared slabs)
  if (wr == 0) { unit(2, em0, en0, epar); unit(3, em0, en0, epar); flush_colsum(em0, en0); __builtin_amdgcn_s_waitcnt(0xC07F); }
  __builtin_amdgcn_s_barrier();
  if (wr == 1) { unit(2, em0, en0, epar); unit(3, em0, en0, epar); flush_colsum(em0, en0); }
}

// out[n] += sum_r ws[r, n]  (bias gradient from the per-slab column sums of the data-gradient epilogue)
__global__ __launch_bounds__(256) void colsum_reduce_kernel(const float* __restrict__ ws, int rows, int N, int n_store, float* __restrict__ out) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= n_store) return;
  const int chunk = (rows + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * chunk, r1 = min(rows, r0 + chunk);
  float acc = 0.f;
  int r = r0;
  for (; r + 4 <= r1; r += 4) {                            // four loads in flight; the order of the additions stays r0, r0 + 1, ...
    const float a = ws[(long)r * N + n], b = ws[(long)(r + 1) * N + n], c = ws[(long)(r + 2) * N + n], d = ws[(long)(r + 3) * N + n];
    acc += a; acc += b; acc += c; acc += d;
  }
  for (; r < r1; ++r) acc += ws[(long)r * N + n];
  atomicAdd(out + n, acc);
}

// out[n] += sum_m (hi + lo)[m, n] of a split-bf16 matrix Y [M, 2 N] (interleaved layout): the bias gradient of the split data-gradient
// launches of the persistent kernel, whose mask + column-sum flavour does not fit the register file next to the two-pass epilogue
// (13 spilled registers = scratch traffic in the k-loop; measured 14 ms per launch instead of 3).  One thread per 8 logical columns
// (two 16-byte loads per row, four rows in flight), 1024-row chunks per workgroup, one fp32 atomic per column and chunk (256-row
// chunks quadruple the same-address atomics: 3.3 -> 3.7 ms per 1024-wide launch, 0.68 -> 2.1 ms per 256-wide one, gpurun_out/r3s).
__global__ __launch_bounds__(256) void colsum_split_kernel(const __bf16* __restrict__ Y, long ldy, int M, int n_store, float* __restrict__ out) {
  const int g8 = (n_store + 7) >> 3;                       // column groups of 8
  const int gw = g8 < 256 ? g8 : 256;                      // column groups per workgroup
  const int per = 256 / gw;                                // rows handled side by side by one workgroup
  const int cg = threadIdx.x % gw + blockIdx.y * 256, rsub = threadIdx.x / gw;
  if (cg >= g8 || rsub >= per) return;
  const int c0 = cg * 8;
  const __bf16* src = Y + ((c0 >> 6) << 7) + (c0 & 63);
  const int r0 = blockIdx.x * 1024, r1 = min(M, r0 + 1024);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int r = r0 + rsub;
  for (; r + 3 * per < r1; r += 4 * per) {                 // eight 16-byte loads in flight per thread
    bf16x8 h[4], l[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { h[q] = *(const bf16x8*)(src + (long)(r + q * per) * ldy); l[q] = *(const bf16x8*)(src + (long)(r + q * per) * ldy + 64); }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += (float)h[q][e] + (float)l[q][e];
  }
  for (; r < r1; r += per) {
    const bf16x8 h = *(const bf16x8*)(src + (long)r * ldy), l = *(const bf16x8*)(src + (long)r * ldy + 64);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += (float)h[e] + (float)l[e];
  }
#pragma unroll
  for (int e = 0; e < 8; ++e)
    if (c0 + e < n_store) atomicAdd(out + c0 + e, acc[e]);
}

template <typename T, int BM, int BN, int WM, int WN, bool F16 = false>
static int launch_nt(const GemmNT& p, hipStream_t stream) {
  constexpr int LDS = 2 * (BM + BN) * 128;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)gemm_nt_kernel<T, BM, BN, WM, WN, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set = true;
  }
  const int tiles_m = (p.M + BM - 1) / BM;
  const int tiles = tiles_m * (p.N / BN);
  hipLaunchKernelGGL((gemm_nt_kernel<T, BM, BN, WM, WN, F16>), dim3(tiles), dim3(64 * WM * WN), LDS, stream, p);
  if (p.fast_epi && p.colsum_ws != nullptr) {
    const int rows = tiles_m * WM;
    int ychunks = rows / 8;
    ychunks = (ychunks < 1 || p.det) ? 1 : (ychunks > 64 ? 64 : ychunks);
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3((p.n_store + 255) / 256, ychunks), dim3(256), 0, stream, p.colsum_ws, rows, p.N, p.n_store, p.colsum);
  }
  return snerf_check_launch();
}

template <bool F16 = false>
static int launch_nt8(const GemmNT& p, hipStream_t stream) {
  constexpr int LDS = 8 * 128 * 128;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)gemm_nt8_kernel<F16>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set = true;
  }
  const int tiles_m = (p.M + 255) / 256;
  const int tiles = tiles_m * (p.N / 256);
  hipLaunchKernelGGL((gemm_nt8_kernel<F16>), dim3(tiles), dim3(512), LDS, stream, p);
  if (p.fast_epi && p.colsum_ws != nullptr) {
    const int rows = tiles_m * 2;
    int ychunks = rows / 8;
    ychunks = (ychunks < 1 || p.det) ? 1 : (ychunks > 64 ? 64 : ychunks);
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3((p.n_store + 255) / 256, ychunks), dim3(256), 0, stream, p.colsum_ws, rows, p.N, p.n_store, p.colsum);
  }
  return snerf_check_launch();
}

template <bool F16 = false>
static int launch_nt8p(const GemmNT& p, hipStream_t stream) {
  constexpr int LDS = 8 * 128 * 128 + 4 * 4096 + 2048 + 8192;
  static bool attr_set = false;
  static int n_cu = 256;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)gemm_nt8p_kernel<ACT_NONE, false, false, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    hipFuncSetAttribute((const void*)gemm_nt8p_kernel<ACT_RELU, false, false, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    hipFuncSetAttribute((const void*)gemm_nt8p_kernel<ACT_NONE, true, false, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    hipFuncSetAttribute((const void*)gemm_nt8p_kernel<ACT_MASK, false, false, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    hipFuncSetAttribute((const void*)gemm_nt8p_kernel<ACT_MASK, true, false, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    hipFuncSetAttribute((const void*)gemm_nt8p_kernel<ACT_RELU_BITS, false, false, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    hipFuncSetAttribute((const void*)gemm_nt8p_kernel<ACT_MASK_BITS, false, false, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    hipFuncSetAttribute((const void*)gemm_nt8p_kernel<ACT_MASK_BITS, true, false, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    // the K = 128 flavours (two k-tiles per tile: see KT2 in the kernel)
    hipFuncSetAttribute((const void*)gemm_nt8p_kernel<ACT_NONE, false, false, F16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    hipFuncSetAttribute((const void*)gemm_nt8p_kernel<ACT_RELU, false, false, F16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    hipFuncSetAttribute((const void*)gemm_nt8p_kernel<ACT_NONE, true, false, F16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    hipFuncSetAttribute((const void*)gemm_nt8p_kernel<ACT_MASK, false, false, F16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    hipFuncSetAttribute((const void*)gemm_nt8p_kernel<ACT_MASK, true, false, F16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    hipFuncSetAttribute((const void*)gemm_nt8p_kernel<ACT_RELU_BITS, false, false, F16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    hipFuncSetAttribute((const void*)gemm_nt8p_kernel<ACT_MASK_BITS, false, false, F16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    hipFuncSetAttribute((const void*)gemm_nt8p_kernel<ACT_MASK_BITS, true, false, F16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      n_cu = prop.multiProcessorCount;
    attr_set = true;
  }
  const int tiles_m = (p.M + 255) / 256;
  const int tiles = tiles_m * (p.N / 256);
  int grid = tiles < n_cu ? tiles : n_cu;
#if SNERF_PROBE
  // PROBE builds: SNERF_NT8P_GRID caps the number of persistent workgroups (tools/gemm_grid_probe.py: what a tile's row stores cost
  // when only a part of the chip is working)
  if (const char* e = getenv("SNERF_NT8P_GRID")) { const int g = atoi(e); if (g > 0 && g < grid) grid = g; }
#endif
  const bool cs = p.colsum_ws != nullptr;
  const dim3 g(grid), b(512);
  if (p.split == 2) {                                    // fp16 + fp8 flavours (forward activations only)
    if constexpr (F16) {
      static bool s8attr = false;
      if (!s8attr) {
        hipFuncSetAttribute((const void*)gemm_nt8p_kernel<ACT_NONE, false, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        hipFuncSetAttribute((const void*)gemm_nt8p_kernel<ACT_RELU, false, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        hipFuncSetAttribute((const void*)gemm_nt8p_kernel<ACT_RELU_BITS, false, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        s8attr = true;
      }
      if (cs) return SNERF_ERR_ARG;
      if (p.act == ACT_RELU_BITS) hipLaunchKernelGGL((gemm_nt8p_kernel<ACT_RELU_BITS, false, 2, true>), g, b, LDS, stream, p);
      else if (p.act == ACT_RELU) hipLaunchKernelGGL((gemm_nt8p_kernel<ACT_RELU, false, 2, true>), g, b, LDS, stream, p);
      else if (p.act == ACT_NONE) hipLaunchKernelGGL((gemm_nt8p_kernel<ACT_NONE, false, 2, true>), g, b, LDS, stream, p);
      else return SNERF_ERR_ARG;
      return snerf_check_launch();
    } else return SNERF_ERR_ARG;
  }
  if (p.split && F16) return SNERF_ERR_ARG;
  if (p.split) {                                         // split-bf16 flavours (no bf16-aux mask: the data gradients use the bit masks)
    static bool sattr = false;
    if (!sattr) {
      hipFuncSetAttribute((const void*)gemm_nt8p_kernel<ACT_NONE, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      hipFuncSetAttribute((const void*)gemm_nt8p_kernel<ACT_NONE, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      hipFuncSetAttribute((const void*)gemm_nt8p_kernel<ACT_RELU, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      hipFuncSetAttribute((const void*)gemm_nt8p_kernel<ACT_RELU_BITS, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      hipFuncSetAttribute((const void*)gemm_nt8p_kernel<ACT_MASK_BITS, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      sattr = true;
    }
    if (p.act == ACT_MASK_BITS) {
      // (with a bias gradient: the plain flavour, then colsum_split_kernel over the output -- see there)
      GemmNT q = p;
      q.colsum_ws = nullptr;
      hipLaunchKernelGGL((gemm_nt8p_kernel<ACT_MASK_BITS, false, true>), g, b, LDS, stream, q);
      if (cs) {
        const int g8 = (p.n_store + 7) / 8;
        hipLaunchKernelGGL(colsum_split_kernel, dim3((p.M + 1023) / 1024, (g8 + 255) / 256), dim3(256), 0, stream, (const __bf16*)p.Y, p.ldy, p.M, p.n_store, p.colsum);
      }
      return snerf_check_launch();
    } else if (p.act == ACT_RELU_BITS && !cs) hipLaunchKernelGGL((gemm_nt8p_kernel<ACT_RELU_BITS, false, true>), g, b, LDS, stream, p);
    else if (p.act == ACT_RELU && !cs) hipLaunchKernelGGL((gemm_nt8p_kernel<ACT_RELU, false, true>), g, b, LDS, stream, p);
    else if (p.act == ACT_NONE && cs) hipLaunchKernelGGL((gemm_nt8p_kernel<ACT_NONE, true, true>), g, b, LDS, stream, p);
    else if (p.act == ACT_NONE) hipLaunchKernelGGL((gemm_nt8p_kernel<ACT_NONE, false, true>), g, b, LDS, stream, p);
    else return SNERF_ERR_ARG;
  } else {
    const bool kt2 = (p.K >> 6) == 2;                    // K = 128: the flavour that orders the bias / mask DMAs by hand
#define NT8P(A_, C_) do { if (kt2) hipLaunchKernelGGL((gemm_nt8p_kernel<A_, C_, false, F16, true>), g, b, LDS, stream, p); \
                          else hipLaunchKernelGGL((gemm_nt8p_kernel<A_, C_, false, F16, false>), g, b, LDS, stream, p); } while (0)
    if (p.act == ACT_MASK) {
      if (cs) NT8P(ACT_MASK, true); else NT8P(ACT_MASK, false);
    } else if (p.act == ACT_MASK_BITS) {
      if (cs) NT8P(ACT_MASK_BITS, true); else NT8P(ACT_MASK_BITS, false);
    } else if (p.act == ACT_RELU_BITS) {
      if (cs) return SNERF_ERR_ARG;
      NT8P(ACT_RELU_BITS, false);
    } else if (cs) {
      if (p.act != ACT_NONE) return SNERF_ERR_ARG;
      NT8P(ACT_NONE, true);
    } else if (p.act == ACT_RELU) {
      NT8P(ACT_RELU, false);
    } else {
      NT8P(ACT_NONE, false);
    }
#undef NT8P
  }
  if (p.colsum_ws != nullptr) {
    const int rows = grid * 2;                            // one partial row per (workgroup, wave row)
    int ychunks = rows / 8;
    ychunks = (ychunks < 1 || p.det) ? 1 : (ychunks > 64 ? 64 : ychunks);
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3((p.n_store + 255) / 256, ychunks), dim3(256), 0, stream, p.colsum_ws, rows, p.N, p.n_store, p.colsum);
  }
  return snerf_check_launch();
}

extern "C" int snerf_linear_fwd(const void* A, long lda, const void* W, long ldw, const float* bias, void* Y, long ldy,
                                const void* aux, long ldaux, float* colsum, float* colsum_ws, int M, int N, int K, int n_store,
                                int act, int dtype, int out_f32, int variant, void* stream) {
  if (M <= 0) return SNERF_OK;
  if (N <= 0 || (N % 128) != 0 || K <= 0 || n_store <= 0 || n_store > N) return SNERF_ERR_ARG;
  // SNERF_DT_BF16X3: split-bf16 operands (see GemmNT::split).  A is [M, >= 2 K] (hi / lo interleaved per 64 columns), W [N, >= 3 K]
  // ([hi | hi | lo] per 64 columns), K the LOGICAL reduction length; a bf16 output is written in the interleaved layout (ldy >= 2 x
  // the logical width), an fp32 output (out_f32) plainly.
  // SNERF_DT_F16F8: the fp16 + fp8 form of the same contract (GemmNT::split == 2): A and W are [.., >= 2 K] 2-byte elements, K logical
  int split = dtype == SNERF_DT_BF16X3 ? 1 : (dtype == SNERF_DT_F16F8 ? 2 : 0);
  if (split) {
    if (K % 64 != 0) return SNERF_ERR_ARG;
    if (split == 2 && (act == ACT_MASK || act == ACT_MASK_BITS || colsum != nullptr)) return SNERF_ERR_ARG;   // forward activations only
    if (split == 2 && !out_f32 && n_store % 64 != 0) return SNERF_ERR_ARG;   // (a group's e4m3 bytes are 16 columns per store: whole 64-column groups only)
    dtype = split == 2 ? 2 : SNERF_DT_BF16;
    K *= split == 2 ? 2 : 3;
  }
  // SNERF_DT_F16: the bf16 kernels with the fp16 MFMA and fp16 conversions (same tiles, same layouts, same rate)
  const bool f16 = dtype == 2;
  if (f16) dtype = SNERF_DT_BF16;
  if (dtype != SNERF_DT_F32 && dtype != SNERF_DT_BF16) return SNERF_ERR_ARG;
  const int bke = dtype == SNERF_DT_F32 ? 32 : 64;
  if (K % bke != 0 || lda % (bke / 8) != 0 || ldw % (bke / 8) != 0) return SNERF_ERR_ARG;
  if ((act == ACT_MASK || act == ACT_RELU_BITS || act == ACT_MASK_BITS) && aux == nullptr) return SNERF_ERR_ARG;
  if ((act == ACT_MASK || act == ACT_MASK_BITS) && bias != nullptr) return SNERF_ERR_ARG;   // the masked flavours are data gradients: no bias
  if (act < ACT_NONE || act > ACT_MASK_BITS) return SNERF_ERR_ARG;
  // vector epilogue stores need 4-element alignment of the destination (and of the mask source)
  const long esz = (out_f32 || dtype == SNERF_DT_F32) ? 4 : 2;
  int vec = (ldy % 4 == 0) && (((uintptr_t)Y) % (4 * esz) == 0);
  if (act == ACT_MASK) vec = vec && (ldaux % 4 == 0) && (((uintptr_t)aux) % (dtype == SNERF_DT_F32 ? 16 : 8) == 0);
  if (act >= ACT_RELU_BITS && (((uintptr_t)aux) % 4 != 0)) return SNERF_ERR_ARG;
  // fast (LDS-transposed, 16-byte) epilogue: output in the compute dtype, 16-byte aligned row segments, whole chunks
  const int epc = dtype == SNERF_DT_F32 ? 4 : 8;
  int fast = !(out_f32 && dtype == SNERF_DT_BF16) && (ldy % epc == 0) && (((uintptr_t)Y) % 16 == 0) && (n_store % epc == 0);
  if (act == ACT_MASK) fast = fast && (ldaux % epc == 0) && (((uintptr_t)aux) % 16 == 0);
  if (colsum != nullptr && colsum_ws == nullptr) fast = 0;   // without a workspace the bias gradient uses the atomic path
  if ((variant >> 4) & 8) fast = 0;                          // ablation: force the direct-store epilogue
  const int det = (variant >> 8) & 1;                        // variant bit 8: deterministic fold of the bias-gradient partials
  if (det && colsum != nullptr && !fast) return SNERF_ERR_ARG;   // the direct-store epilogue adds its column sums with atomics
  if (split && !fast && !out_f32) return SNERF_ERR_ARG;          // the interleaved output exists only in the 16-byte epilogues
  if (split && out_f32 && (act == ACT_MASK || colsum != nullptr)) return SNERF_ERR_ARG;
  // variant bit 14: `aux` (ACT_MASK) is a split-bf16 activation, the operands are plain bf16 (GemmNT::aux_split)
  const int aux_split = (variant >> 14) & 1;
  if (aux_split && (split || dtype != SNERF_DT_BF16 || act != ACT_MASK)) return SNERF_ERR_ARG;     // (bf16, or fp16 behind an fp16 + fp8 forward)
  GemmNT p{A, lda, W, ldw, bias, Y, ldy, aux, ldaux, colsum, M, N, K, n_store, act, out_f32, vec, colsum_ws, fast, det, ((variant >> 4) & 7) | ((variant >> 10) & 8),
           (variant >> 9) & 15, split, aux_split};               // (PROBE builds: variant bit 13 = ablation bit 8; bits 9..12 = stagger)
  variant &= 15;
  hipStream_t s = (hipStream_t)stream;
  // variant: 0 = 128x128 block-issue, 1 = 256x256 block-issue,
  //          4 = 256x256 8-phase (bf16, N % 256 == 0; other shapes take the 128x128 kernel)
  //          8 = 256x256 persistent 8-phase (bf16, N % 256 == 0, K >= 128, 16-byte epilogue; else as variant 4)
  // the bit-mask activations exist only in the persistent kernel (the mask layout is its unit geometry)
  const bool p8 = dtype == SNERF_DT_BF16 && (variant & 8) && N % 256 == 0 && K >= 128 && fast && !(colsum != nullptr && act == ACT_RELU) &&
                  (lda * 2 * 256 < (1L << 31)) && (ldw * 2 * 256 < (1L << 31)) && !(split && (act == ACT_MASK || (colsum != nullptr && act == ACT_RELU_BITS))) &&
                  !aux_split &&                                   // (the persistent kernel's own bf16-mask loads know the plain layout only)
                  !(split == 2 && K < 256);                       // (fp16 + fp8: no K = 128 flavour)
  if (act >= ACT_RELU_BITS && !(p8 && (long)((M + 255) / 256) * 8 * (N / 64) * 256 < (1L << 31)))
    return SNERF_ERR_ARG;
  if (f16) {
    if (p8) return launch_nt8p<true>(p, s);
    if (split == 2) return launch_nt<__bf16, 128, 128, 2, 2, true>(p, s);       // (the other 8-phase kernel has no e4m3 tiles)
    if ((variant & 12) && N % 256 == 0) return launch_nt8<true>(p, s);
    if ((variant & 1) && N % 256 == 0) return launch_nt<__bf16, 256, 256, 2, 4, true>(p, s);
    return launch_nt<__bf16, 128, 128, 2, 2, true>(p, s);
  }
  if (p8) return launch_nt8p(p, s);
  if (dtype == SNERF_DT_BF16 && (variant & 12) && N % 256 == 0) return launch_nt8(p, s);
  if (dtype == SNERF_DT_F32) return launch_nt<float, 128, 128, 2, 2>(p, s);
  if ((variant & 1) && N % 256 == 0) return launch_nt<__bf16, 256, 256, 2, 4>(p, s);
  return launch_nt<__bf16, 128, 128, 2, 2>(p, s);
}

// ---------------------------------------------------------------------------
// TN kernel (weight gradient): dW[n,k] += sum_m dZ[m,n] * X[m,k]
// 128 x 128 output tile, 4 waves (2 x 2), M split into chunks across blockIdx.y.
// fp32: LDS rows of 128 floats, ds_read_b32 per MFMA operand (conflict free).
// bf16: LDS rows of 128 bf16, operands gathered with 16-bit LDS reads (the
//       reduction axis is the row axis of both operands, so a lane's 8
//       consecutive k values sit in 8 different rows).
// Rows >= M of dZ are sourced from a zero page so that they contribute 0.
// ---------------------------------------------------------------------------
struct GemmTN {
  const void* Z; long ldz;
  const void* X; long ldx;
  float* dW; long ldw;
  const void* zeros;  // >= 16 bytes of zeros in device memory
  int M, N, K, n_valid, k_valid, m_chunk;
  int tiles, slices;  // 128 x 128 kernel: output tiles and M slices of the launch (1-D grid, XCD-aware placement)
  // split-bf16 operands (SNERF_DT_BF16X3): dZ [M, 2 N] and X [M, 2 K] in the hi / lo interleaved layout; the kernels multiply the
  // PHYSICAL matrices (the 128 x 128 kernel: all four hi / lo combinations of every 64 x 64 block, lo.lo is noise-level and harmless; the
  // 8-phase kernel deals X's blocks so that the lo.lo quadrant is the same phase for every wave and skips it) and the output index
  // maps the physical (n', k') back to the logical (n, k) = ((n' >> 7) << 6 | n' & 63, ...): the atomics add the combinations up.
  int split;          // 1: as above; 2: dZ plain bf16, X a split-bf16 activation of which only the hi half is multiplied (the single-pass
                      //    weight gradient behind a split-bf16 forward, compute="bf16x3_fwd"): K = X's LOGICAL width, logical column c is read at
                      //    physical column (c >> 6) * 128 + (c & 63); everything else as the plain bf16 launch
  float* part;        // deterministic mode: every (tile, M slice) stores its partial tile at part + slice * part_stride + n * part_ld + k
  long part_stride;   // (no atomics); tn_fold_kernel adds the slices in a fixed order.  nullptr: fp32 atomics straight into dW
  int part_ld;
};

// TN_STAGES: slots of the staging ring of the 128 x 128 weight-gradient kernel (16 KiB per slot)
template <typename T, bool TR, int TN_STAGES, bool F16 = false>
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTN p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int EPC = 16 / (int)sizeof(T);
  constexpr int ROWB = 128 * (int)sizeof(T);         // bytes per LDS row (128 columns)
  constexpr int MT = sizeof(T) == 4 ? 16 : 32;       // reduction rows per stage
  constexpr int TILEB = MT * ROWB;                   // 8 KiB per operand per stage
  constexpr int RPI = 1024 / ROWB;                   // rows covered by one glds instruction (2 or 4)
  constexpr int LI = MT / RPI / 4;                   // instructions per wave per operand (2)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 1, wk = wave & 1;
  const int tiles_k = (p.K + 127) / 128;
  // 1-D grid of tiles x slices workgroups.  Workgroup b runs on XCD b % 8: ALL tiles of an M slice go to one XCD (consecutive
  // positions on it), so that the slice's rows of dZ and X -- which every tile of the slice re-reads -- are fetched into one L2
  // instead of eight (round 2: the N = 128, K = 1051 launch moved 2.4 GB through the fabric for 1.3 GB of operands).
  const int ntile = p.tiles;
  int slice, tile;
  {
    const int b = (int)blockIdx.x, xcd = b & 7, idx = b >> 3;
    const int full = p.slices >> 3;                     // slices every XCD owns; the first (slices & 7) XCDs own one more
    const int extra = p.slices & 7;
    const int mine = full + (xcd < extra ? 1 : 0);      // slices of this XCD
    if (idx >= mine * ntile) return;                    // (the grid is padded to 8 x the largest share)
    slice = (idx / ntile) * 8 + xcd;
    tile = idx % ntile;
  }
  const int n0 = (tile / tiles_k) * 128, k0 = (tile % tiles_k) * 128;
  const int mbeg = slice * p.m_chunk;
  const int mend = min(p.M, mbeg + p.m_chunk);
  const T* __restrict__ Z = (const T*)p.Z;
  const T* __restrict__ X = (const T*)p.X;
  const int steps = (mend - mbeg + MT - 1) / MT;

  // staging: lane -> (row within instruction, 16-byte chunk within row)
  constexpr int CPR = ROWB / 16;                     // chunks per row (32 fp32 / 16 bf16)
  const int lrow = lane / CPR, lch = lane % CPR;
  // clamp the column chunk so that partial tiles (N or K not a multiple of 128) stay in bounds
  // TR (bf16): the 16-byte chunk c of tile row r is stored at chunk position c ^ (4 * (r & 3)) so that the four rows a
  // transposing read touches fall into disjoint 64-byte bank spans; the permutation is applied to the SOURCE address
  const int sch = TR ? (lch ^ (4 * (lrow & 3))) : lch;
  int zc = n0 + sch * EPC; zc = zc < p.N ? zc : p.N - EPC;
  int xc = k0 + sch * EPC; xc = xc < p.K ? xc : p.K - EPC;
  if (p.split == 2) xc = ((xc >> 6) << 7) | (xc & 63);    // the hi half of a split-bf16 activation

  auto issue = [&](int st, int stage) {
    char* sZ = smem + stage * 2 * TILEB;
    char* sX = sZ + TILEB;
#pragma unroll
    for (int i = 0; i < LI; ++i) {
      const int r = (wave * LI + i) * RPI + lrow;
      const int m = mbeg + st * MT + r;
      const T* gz = m < mend ? Z + (long)m * p.ldz + zc : (const T*)p.zeros;
      const int mx = m < mend ? m : mend - 1;
      glds16(gz, sZ + (wave * LI + i) * 1024);
      glds16(X + (long)mx * p.ldx + xc, sX + (wave * LI + i) * 1024);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Ring of TN_STAGES slots, TN_STAGES - 1 stages in flight: these narrow launches are bound by HBM latency, not by their few MFMAs
  // (one 16 KiB stage in flight per workgroup = 32 KiB per CU: round 2 measured 2.5 TB/s on the N = 128, K = 1051 launch).
  // Stage st + TN_STAGES - 1 is issued after the barrier of iteration st into the slot stage st - 1 occupied (every wave finished
  // reading it before that barrier).  vmcnt(2 LI (TN_STAGES - 2)): the 2 LI loads per wave of each younger stage may stay in
  // flight; the last iterations, with fewer stages behind them, simply wait for everything.
#pragma unroll
  for (int s = 0; s < TN_STAGES - 1; ++s)
    if (s < steps) issue(s, s);
  for (int st = 0; st < steps; ++st) {
    if (st + TN_STAGES - 2 < steps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LI * (TN_STAGES - 2)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (st + TN_STAGES - 1 < steps) issue(st + TN_STAGES - 1, (st + TN_STAGES - 1) % TN_STAGES);
    const char* sZ = smem + (st % TN_STAGES) * 2 * TILEB;
    const char* sX = sZ + TILEB;
    if constexpr (sizeof(T) == 4) {
#pragma unroll
      for (int kk = 0; kk < MT / 2; ++kk) {
        const int row = 2 * kk + (lane >> 5);
        float a[2], b[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          a[t] = *(const float*)(sZ + row * ROWB + (wn * 64 + t * 32 + (lane & 31)) * 4);
          b[t] = *(const float*)(sX + row * ROWB + (wk * 64 + t * 32 + (lane & 31)) * 4);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    } else if constexpr (TR) {
      // ds_read_b64_tr_b16: per 16-lane group, lane p supplies the address of 4 consecutive bf16 of row (p>>2) and receives
      // column p of the 4x16 block -> 4 consecutive reduction indices for its own output row/column: two reads per fragment
      typedef __attribute__((address_space(3))) bf16x4* lds_b4;
      const int g = lane >> 4, pl = lane & 15;
      const int prow = pl >> 2;                                  // row inside the 4-row block == (row & 3)
#pragma unroll
      for (int ks = 0; ks < MT / 16; ++ks) {
        const int row0 = ks * 16 + 8 * (g >> 1) + prow;
        bf16x8 a[2], b[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int ca = (wn * 64 + t * 32 + 16 * (g & 1)) / 8 + ((pl & 3) >> 1);     // logical 16-byte chunk
          const int cb = (wk * 64 + t * 32 + 16 * (g & 1)) / 8 + ((pl & 3) >> 1);
          const int oa = ((ca ^ (4 * prow)) << 4) + ((pl & 1) << 3);
          const int ob = ((cb ^ (4 * prow)) << 4) + ((pl & 1) << 3);
          const bf16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4)(sZ + row0 * ROWB + oa));
          const bf16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4)(sZ + (row0 + 4) * ROWB + oa));
          const bf16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4)(sX + row0 * ROWB + ob));
          const bf16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4)(sX + (row0 + 4) * ROWB + ob));
          a[t] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
          b[t] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) mma32t<F16>(acc[i][j], a[i], b[j]);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < MT / 16; ++ks) {
        const int rbase = ks * 16 + 8 * (lane >> 5);
        union { bf16x8 v; unsigned short u[8]; } a[2], b[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int ca = (wn * 64 + t * 32 + (lane & 31)) * 2, cb = (wk * 64 + t * 32 + (lane & 31)) * 2;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            a[t].u[e] = *(const unsigned short*)(sZ + (rbase + e) * ROWB + ca);
            b[t].u[e] = *(const unsigned short*)(sX + (rbase + e) * ROWB + cb);
          }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) mma32t<F16>(acc[i][j], a[i].v, b[j].v);
      }
    }
  }
  // D[i = n][j = k]: j = lane&31, i = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int k = k0 + wk * 64 + j * 32 + (lane & 31);
      if (p.split == 1) k = ((k >> 7) << 6) | (k & 63);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int n = n0 + wn * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (p.split == 1) n = ((n >> 7) << 6) | (n & 63);
        if (n < p.n_valid && k < p.k_valid) {
          if (p.part != nullptr) p.part[(long)slice * p.part_stride + (long)n * p.part_ld + k] = acc[i][j][r];
          else atomicAdd(p.dW + (long)n * p.ldw + k, acc[i][j][r]);
        }
      }
    }
}

// ---------------------------------------------------------------------------
// TN kernel, 8-phase schedule (bf16): dW[n0:n0+256, k0:k0+256] += dZ[mbeg:mend, n-tile]^T . X[mbeg:mend, k-tile]
//
// Same machinery as gemm_nt8p_kernel (half-tiles, quadrant phases, fragments prefetched in place during the MFMA
// sections, staging two k-tiles ahead with five halves in flight, two wave rows in opposite section order), with the
// reduction axis on the ROWS of both operands: a k-tile is 64 rows of dZ and of X; a half-tile is 64 rows x 128 columns
// (256-byte LDS rows, 16-byte chunk c of row r at position c ^ 4 (r & 3)); MFMA fragments come from
// ds_read_b64_tr_b16 (two per fragment).  One workgroup owns one 256 x 256 tile of dW for one slice of M and adds it
// with fp32 atomics at the end; rows past the slice read as zeros through the buffer descriptor.
//     dZ half h, local column lc  <->  tile column (lc>>6)*128 + h*64 + (lc&63)     (wave row wr = lc>>6)
//     X  half g, local column lc  <->  tile column (lc>>5)*64  + g*32 + (lc&31)     (wave col wc = lc>>5)
// ---------------------------------------------------------------------------
template <bool SPLIT, bool F16 = false>
__global__ __launch_bounds__(512) void gemm_tn8_kernel(GemmTN p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __bf16 T;
  typedef __attribute__((address_space(3))) bf16x4* lds_b4;
  constexpr int HALF = 64 * 256;                       // bytes per half-tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int tiles_k = (p.K + 255) >> 8, tiles_n = (p.N + 255) >> 8;
  const int ntile = tiles_k * tiles_n;
  const int id = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int tile = id % ntile, chunk = id / ntile;     // workgroups of one M slice are neighbours on an XCD
  const int n0 = (tile / tiles_k) << 8, k0 = (tile % tiles_k) << 8;
  const int mbeg = chunk * p.m_chunk;
  const int rows = min(p.M, mbeg + p.m_chunk) - mbeg;
  const int KT = (rows + 63) >> 6;
  const T* __restrict__ Z = (const T*)p.Z + (long)mbeg * p.ldz + n0;
  const bool xhi = !SPLIT && p.split == 2;             // X: the hi half of a split-bf16 activation (logical K; see GemmTN::split)
  const T* __restrict__ X = (const T*)p.X + (long)mbeg * p.ldx + (xhi ? 2 * k0 : k0);
  // readable bytes from the tile's first element: up to the last VALID column of the slice's last row (the operands may be
  // column ranges of wider buffers, so nothing past that is known to be mapped)
  const int xcols = xhi ? ((p.K - 1) >> 6) * 128 + ((p.K - 1) & 63) + 1 - 2 * k0 : p.K - k0;    // physical columns from the tile's first one to the last valid one
  const int zbytes = (rows - 1) * (int)p.ldz * 2 + (p.N - n0) * 2, xbytes = (rows - 1) * (int)p.ldx * 2 + xcols * 2;
  const int zstep = 64 * (int)p.ldz * 2, xstep = 64 * (int)p.ldx * 2;

  // ---- staging stream: wave w owns pieces 2w, 2w+1 (4 rows x 256 B) of every half-tile -------------------------------
  const int lrow = lane >> 4, lch = lane & 15;
  const int lc = (lch ^ (4 * lrow)) * 8;               // logical column of this lane's 16 bytes inside the half
  // split-bf16 operands ([hi 64 | lo 64] per 128 physical columns): X's 32-column blocks are dealt so that EVERY wave column gets one hi
  // block (half 0) and the lo block of the same logical columns (half 1) -- tile column (wc, g, c) <-> physical column
  // (wc >> 1) * 128 + g * 64 + (wc & 1) * 32 + c -- while dZ's halves are hi / lo by themselves (half h = 64 columns).  The quadrant
  // dZ half 1 x X half 1 is then lo . lo for every wave and is skipped: 3/4 of the MFMAs instead of 4/4.
  constexpr int xhalf = SPLIT ? 128 : 64;              // byte offset of X half 1
  const int zcol = ((lc >> 6) * 128 + (lc & 63)) * 2;
  const int xcol = SPLIT ? (((lc >> 6) * 128) + ((lc >> 5) & 1) * 32 + (lc & 31)) * 2 : ((lc >> 5) * (xhi ? 128 : 64) + (lc & 31)) * 2;
  int zrel[2], xrel[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (wave * 2 + i) * 4 + lrow;
    zrel[i] = r * (int)p.ldz * 2 + zcol;
    xrel[i] = r * (int)p.ldx * 2 + xcol;
  }
  int s_kt = 0;
  // which: 0 = Z0, 1 = Z1, 2 = X0, 3 = X1 of the stream's k-tile
  auto stage = [&](int db, int which) __attribute__((always_inline)) {
    char* dst = smem + (db * 4 + which) * HALF + wave * 2048;
    if (which < 2) {
      const int off = s_kt * zstep;
      const int left = zbytes - off;
      const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)Z + off), 0, left > 0 ? left : 0, 0x00020000);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)dst, 16, zrel[0] + which * 128, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(dst + 1024), 16, zrel[1] + which * 128, 0, 0, 0);
    } else {
      const int off = s_kt * xstep;
      const int left = xbytes - off;
      const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)X + off), 0, left > 0 ? left : 0, 0x00020000);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)dst, 16, xrel[0] + (which - 2) * xhalf, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(dst + 1024), 16, xrel[1] + (which - 2) * xhalf, 0, 0, 0);
    }
  };

  // ---- fragments: per 16-lane group, lane pl supplies the address of 4 consecutive bf16 of row (pl>>2) of a 4-row block and
  // receives column pl of that 4 x 16 block; two reads (rows +0..3, +4..7) make the 8 reduction indices of an MFMA operand
  const int g4 = lane >> 4, pl = lane & 15, prow = pl >> 2;
  const int frow = (8 * (g4 >> 1) + prow) * 256;       // + ks * 16 * 256 (+ 4 * 256 for the second read)
  const int fsub = ((pl & 1) << 3);
  int zc[2], xc1;                                      // swizzled chunk byte offsets of the wave's 32-column blocks
#pragma unroll
  for (int i2 = 0; i2 < 2; ++i2) zc[i2] = ((((wr * 64 + i2 * 32 + 16 * (g4 & 1)) >> 3) + ((pl & 3) >> 1)) ^ (4 * prow)) << 4;
  xc1 = ((((wc * 32 + 16 * (g4 & 1)) >> 3) + ((pl & 3) >> 1)) ^ (4 * prow)) << 4;
  bf16x8 aF[2][4], bS[2][4];
  // The transposing reads are issued as inline asm: hipcc orders the builtin behind ALL outstanding LDS-DMA (vmcnt(0) in
  // every phase, which empties the staging pipeline).  Their results are first used after the lgkmcnt(0) + sched_barrier
  // that ends the NEXT load section; hardware returns LDS data in order.
  const unsigned lds0 = (unsigned)(size_t)smem;
  auto lds_frag = [&](unsigned addr) __attribute__((always_inline)) {
    bf16x4 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:1024" : "=&v"(lo), "=&v"(hi) : "v"(addr));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  const unsigned za[2] = {lds0 + frow + zc[0] + fsub, lds0 + frow + zc[1] + fsub};
  const unsigned xa = lds0 + frow + xc1 + fsub;
  auto lds_a = [&](int db, int h, int i2, int ks) __attribute__((always_inline)) {
    return lds_frag(za[i2] + (db * 4 + h) * HALF + ks * 4096);
  };
  auto lds_b = [&](int db, int g, int ks) __attribute__((always_inline)) {
    return lds_frag(xa + (db * 4 + 2 + g) * HALF + ks * 4096);
  };
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // One barrier per phase.  Wave row 1 runs [load section, MFMA section], wave row 0 [MFMA section, load section] between two
  // barriers, so on every SIMD one wave stages while the other multiplies, and they swap roles mid-interval without meeting.
  // A load section = stage this phase's half, vmcnt(10) (five halves in flight), lgkmcnt(0).  Ordering: a half retired in the
  // load section of phase j (by the end of interval j for both rows) is first read in MFMA section j+1; a half last read in
  // MFMA section p is restaged in load section p+2: row 1's reads (end of interval p) are retired by its lgkmcnt(0) at the
  // start of interval p+1, row 0's (start of interval p) by its lgkmcnt(0) at the end of interval p -- a barrier lies between
  // either and every DMA of phase p+2.
  auto wait_load = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto end_phase = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- prologue: k-tiles 0 and 1 (tiles past the slice read zeros) ------------------------------------------------------
  stage(0, 2); stage(0, 0); stage(0, 3); stage(0, 1);
  ++s_kt;
  stage(1, 3); stage(1, 0); stage(1, 2); stage(1, 1);
  ++s_kt;
  asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    aF[0][ks] = lds_a(0, 0, 0, ks);
    aF[1][ks] = lds_a(0, 0, 1, ks);
    bS[0][ks] = lds_b(0, 0, ks);
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();

  auto ktile = [&](auto db_tag) __attribute__((always_inline)) {
    constexpr int db = decltype(db_tag)::value;
    constexpr int F = db, S = 1 - db;
    // P1: Z0 x X_F; fetch X_S
    if (wr == 1) { stage(db, 2 + F); wait_load(); }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bS[S][ks] = lds_b(db, S, ks);
      mma32t<F16>(acc[0][F], aF[0][ks], bS[F][ks]);
      mma32t<F16>(acc[1][F], aF[1][ks], bS[F][ks]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (wr == 0) { stage(db, 2 + F); wait_load(); }
    end_phase();
    // P2: Z0 x X_S; Z0 fragments replaced by Z1 in place
    if (wr == 1) { stage(db, 0); wait_load(); }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2) {
        mma32t<F16>(acc[i2][S], aF[i2][ks], bS[S][ks]);
        aF[i2][ks] = lds_a(db, 1, i2, ks);
        __builtin_amdgcn_sched_barrier(0);
      }
    if (wr == 0) { stage(db, 0); wait_load(); }
    end_phase();
    // P3: Z1 x X_S; X_S replaced by the first X half of the next k-tile  (split operands: lo . lo when S = 1 -- skipped)
    if (wr == 1) { stage(db, 2 + S); wait_load(); }
    constexpr bool skip3 = S == 1 && SPLIT, skip4 = F == 1 && SPLIT;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if constexpr (!skip3) {
        mma32t<F16>(acc[2][S], aF[0][ks], bS[S][ks]);
        mma32t<F16>(acc[3][S], aF[1][ks], bS[S][ks]);
      }
      bS[S][ks] = lds_b(db ^ 1, S, ks);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (wr == 0) { stage(db, 2 + S); wait_load(); }
    end_phase();
    // P4: Z1 x X_F; Z1 replaced by Z0 of the next k-tile
    if (wr == 1) { stage(db, 1); ++s_kt; wait_load(); }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2) {
        if constexpr (!skip4) mma32t<F16>(acc[2 + i2][F], aF[i2][ks], bS[F][ks]);
        aF[i2][ks] = lds_a(db ^ 1, 0, i2, ks);
        __builtin_amdgcn_sched_barrier(0);
      }
    if (wr == 0) { stage(db, 1); ++s_kt; wait_load(); }
    end_phase();
  };
  for (int t = 0; t < KT; t += 2) {
    ktile(std::integral_constant<int, 0>{});
    if (t + 1 < KT) ktile(std::integral_constant<int, 1>{});
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // D[i' = n][j' = k]: j' = lane&31, i' = (r&3) + 8*(r>>2) + 4*(lane>>5); acc[i][j]: n block i, k block j of the wave tile
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int k = k0 + wc * 64 + j * 32 + (lane & 31);
      if constexpr (SPLIT) {                              // X's dealt 32-column blocks (see the staging): back to the physical, then the logical column
        if (i >= 2 && j == 1) continue;                   // the skipped lo . lo quadrant
        k = k0 + (wc >> 1) * 128 + j * 64 + (wc & 1) * 32 + (lane & 31);
        k = ((k >> 7) << 6) | (k & 63);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int n = n0 + wr * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if constexpr (SPLIT) n = ((n >> 7) << 6) | (n & 63);
        if (n < p.n_valid && k < p.k_valid) {
          if (p.part != nullptr) p.part[(long)chunk * p.part_stride + (long)n * p.part_ld + k] = acc[i][j][r];
          else atomicAdd(p.dW + (long)n * p.ldw + k, acc[i][j][r]);
        }
      }
    }
}

// dW[n, k] += sum over the M slices, in slice order, of the partial tiles (deterministic weight gradient; also the default for short
// reductions, ops.linear_wgrad).  VEC: four columns per thread (16-byte loads, the slices' loads of an unrolled group in flight together).
template <bool VEC>
__global__ __launch_bounds__(256) void tn_fold_kernel(const float* __restrict__ part, long part_stride, int part_ld, int slices, int n_valid,
                                                      int k_valid, float* __restrict__ dW, long ldw) {
  const int k = (blockIdx.x * 256 + threadIdx.x) * (VEC ? 4 : 1), n = blockIdx.y;
  if (k >= k_valid || n >= n_valid) return;
  const float* src = part + (long)n * part_ld + k;
  if constexpr (VEC) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int c = 0; c < slices; ++c) s += *(const f32x4*)(src + (long)c * part_stride);
    f32x4* dst = (f32x4*)(dW + (long)n * ldw + k);
    *dst += s;
  } else {
    float s = 0.f;
    for (int c = 0; c < slices; ++c) s += src[(long)c * part_stride];
    dW[(long)n * ldw + k] += s;
  }
}

// The same fold for MANY slices (the 128 x 128 kernel runs ~1024 workgroups per launch: 114 ... 1024 slices per output tile): a workgroup takes 64
// consecutive (VEC: 4-float) pieces of the valid output region and splits the slices over its G = 16 waves -- wave g adds slices g, g + G, ... in order,
// wave 0 then adds the 16 partial sums in wave order: a fixed order again (bit-reproducible), with 16 x more loads in flight than one thread per piece.
template <bool VEC>
__global__ __launch_bounds__(1024) void tn_fold_many_kernel(const float* __restrict__ part, long part_stride, int part_ld, int slices, int n_valid,
                                                            int k_valid, float* __restrict__ dW, long ldw) {
  constexpr int G = 16, W = VEC ? 4 : 1;
  __shared__ float sm[G][64][W];
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int kq = k_valid / W;                                     // pieces per output row
  const long e = (long)blockIdx.x * 64 + lane;
  const bool live = e < (long)n_valid * kq;
  const int n = live ? (int)(e / kq) : 0, k = live ? (int)(e - (long)n * kq) * W : 0;
  const float* src = part + (long)n * part_ld + k;
  float s[W];
#pragma unroll
  for (int w = 0; w < W; ++w) s[w] = 0.f;
  if (live) {
#pragma unroll 4
    for (int c = g; c < slices; c += G) {
      if constexpr (VEC) { const f32x4 v = *(const f32x4*)(src + (long)c * part_stride); s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3]; }
      else s[0] += src[(long)c * part_stride];
    }
  }
#pragma unroll
  for (int w = 0; w < W; ++w) sm[g][lane][w] = s[w];
  __syncthreads();
  if (g == 0 && live) {
    float* dst = dW + (long)n * ldw + k;
#pragma unroll
    for (int w = 0; w < W; ++w) {
      float t = sm[0][lane][w];
#pragma unroll
      for (int q = 1; q < G; ++q) t += sm[q][lane][w];
      dst[w] += t;
    }
  }
}

struct TnPlan {
  bool use8;          // 256 x 256 8-phase kernel, else 128 x 128
  int slices;         // M slices (workgroups per output tile)
  int m_chunk;        // rows per slice
  int part_ld;        // row length of a partial tile image (K rounded up to the tile)
  long part_stride;   // floats per slice in the deterministic workspace
};

// fold: the launch stores partial tiles (deterministic / default fold mode) instead of adding with atomics -- two rules that exist to amortise a
// workgroup's atomics then do not apply to SMALL M (<= 131 072 rows, where the fixed cost of a launch is what is left of it): the 128 x 128 kernel
// may cut slices of 256 instead of >= 1024 rows, and a launch the 256 x 256 kernel would cut into slices of fewer than eight k-tiles (one or two output
// tiles: M = 32 768 -> 256 slices of two k-tiles, a 256 KB partial tile each) goes to the 128 x 128 kernel instead.
static TnPlan tn_plan(int M, int N, int K, long ldz, long ldx, int dtype, int variant, bool fold = false) {
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    n_cu = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      n_cu = prop.multiProcessorCount;
  }
  TnPlan pl{};
  // variant 2 (bf16): 256 x 256 tiles, 8-phase schedule; M is cut into as many slices as keep every CU busy
  if ((variant & 2) && dtype == SNERF_DT_BF16 && N % 256 == 0 && K >= 256 && M >= 4096 && ldz * 2 * 65 < (1L << 31) && ldx * 2 * 65 < (1L << 31)) {
    const int t8 = (N / 256) * ((K + 255) / 256);
    int ch = n_cu / t8;
    ch = ch < 1 ? 1 : ch;
    long mc = ((long)M + ch - 1) / ch;
    mc = ((mc + 127) / 128) * 128;                    // whole k-tile pairs
    const long lim = (1L << 30) / ((ldz > ldx ? ldz : ldx) * 2);   // slice bytes stay inside 32-bit buffer offsets
    if (mc > lim) mc = lim / 128 * 128;
    if (!(fold && M <= 131072 && mc < 512)) {
      pl.use8 = true; pl.m_chunk = (int)mc; pl.slices = (int)((M + mc - 1) / mc);
      pl.part_ld = ((K + 255) / 256) * 256; pl.part_stride = (long)N * pl.part_ld;
      return pl;
    }
  }
  // split M so that every CU has work but each block still amortises its atomics
  const int tiles = ((N + 127) / 128) * ((K + 127) / 128);
  // about two workgroups per CU: every slice ends with one fp32 atomic per output element, and on the long-M launches of the step
  // 512 workgroups beat the few thousand of round 1 by 8-23 % (tools/gemm_tn_slices_probe.py; variant bits 16 / 32 / 64 select
  // 1024 / 2048 / 4096 for that probe)
  // round 3, with all tiles of an M slice placed on one XCD: 1024 workgroups beat 512 on the two longest launches (N = 1024, K = 96:
  // 247 vs 322 us; N = 1, K = 1024: 246 vs 295) and tie elsewhere (tools/gemm_tn_narrow_probe.py; variant bit 16 now selects 512)
  const int target = (variant & 64) ? 4096 : (variant & 32) ? 2048 : (variant & 16) ? 512 : 1024;
  int chunks = (target + tiles - 1) / tiles;
  int m_chunk = (M + chunks - 1) / chunks;
  m_chunk = ((m_chunk + 255) / 256) * 256;
  const int floor_rows = (fold && M <= 131072) ? 256 : 1024;
  if (m_chunk < floor_rows) m_chunk = floor_rows;
  pl.use8 = false; pl.m_chunk = m_chunk; pl.slices = (M + m_chunk - 1) / m_chunk;
  pl.part_ld = ((K + 127) / 128) * 128; pl.part_stride = (long)(((N + 127) / 128) * 128) * pl.part_ld;
  return pl;
}

static int wgrad_launch(const void* Z, long ldz, const void* X, long ldx, float* dW, long ldw, const void* zeros, int M, int N, int K,
                        int n_valid, int k_valid, int dtype, int variant, float* ws, long ws_floats, void* stream) {
  if (M <= 0) return SNERF_OK;
  // SNERF_DT_BF16X3: N, K are the PHYSICAL widths of the interleaved operands (2 x the logical ones), n_valid / k_valid logical
  const int split = dtype == SNERF_DT_BF16X3;
  if (split) {
    if (N % 128 != 0 || K % 128 != 0 || ws != nullptr) return SNERF_ERR_ARG;        // (no deterministic fold in this mode)
    dtype = SNERF_DT_BF16;
  }
  // variant bit 14 (dtype bf16): X is a split-bf16 activation [M, >= 2 K] of which the hi half is multiplied; K its LOGICAL width (GemmTN::split == 2)
  const int xhi = (variant >> 14) & 1;
  if (xhi && (split || (dtype != SNERF_DT_BF16 && dtype != 2))) return SNERF_ERR_ARG;     // (bf16, or fp16 behind an fp16 + fp8 forward)
  const bool f16 = dtype == 2;                        // SNERF_DT_F16: the bf16 kernels with the fp16 MFMA
  if (f16) dtype = SNERF_DT_BF16;
  const int epc = dtype == SNERF_DT_F32 ? 4 : 8;
  if (N < epc || K < epc || N % epc || K % epc || ldz % epc || ldx % epc || zeros == nullptr) return SNERF_ERR_ARG;
  const TnPlan pl = tn_plan(M, N, K, ldz, ldx, dtype, variant, ws != nullptr);
  if (ws != nullptr && ws_floats < pl.part_stride * pl.slices) return SNERF_ERR_ARG;
  GemmTN p{Z, ldz, X, ldx, dW, ldw, zeros, M, N, K, n_valid, k_valid, pl.m_chunk, 0, pl.slices, split ? 1 : (xhi ? 2 : 0), ws, pl.part_stride, pl.part_ld};
  if (pl.use8) {
    static bool attr_set = false;
    if (!attr_set) {
      hipFuncSetAttribute((const void*)gemm_tn8_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 64 * 256);
      hipFuncSetAttribute((const void*)gemm_tn8_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 64 * 256);
      attr_set = true;
    }
    const int t8 = (N / 256) * ((K + 255) / 256);
    if (f16) {
      static bool attr16 = false;
      if (!attr16) { hipFuncSetAttribute((const void*)gemm_tn8_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 64 * 256); attr16 = true; }
      hipLaunchKernelGGL((gemm_tn8_kernel<false, true>), dim3(t8 * pl.slices), dim3(512), 8 * 64 * 256, (hipStream_t)stream, p);
    } else if (split) hipLaunchKernelGGL(gemm_tn8_kernel<true>, dim3(t8 * pl.slices), dim3(512), 8 * 64 * 256, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(gemm_tn8_kernel<false>, dim3(t8 * pl.slices), dim3(512), 8 * 64 * 256, (hipStream_t)stream, p);
  } else {
    const int tiles = ((N + 127) / 128) * ((K + 127) / 128);
    p.tiles = tiles;
    // variant bit 128: a ring of four staging slots (three stages in flight) instead of the double buffer.  Measured with the XCD-aware
    // placement (round 3, tools/gemm_tn_narrow_probe.py): the ring LOSES (64 KiB of LDS per workgroup halve the residency: N = 128,
    // K = 1051: 472 vs 325 us; N = 1: 341 vs 295), so the double buffer stays the default and the ring is the probe's alternative.
    const bool deep = (variant & 128) != 0;
    const int lds = (deep ? 4 : 2) * 2 * 8192;
    static bool tn_attr_set = false;
    if (!tn_attr_set) {
      hipFuncSetAttribute((const void*)gemm_tn_kernel<float, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * 8192);
      hipFuncSetAttribute((const void*)gemm_tn_kernel<__bf16, true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * 8192);
      hipFuncSetAttribute((const void*)gemm_tn_kernel<__bf16, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * 8192);
      tn_attr_set = true;
    }
    dim3 grid(8 * ((pl.slices + 7) / 8) * tiles);        // every XCD gets room for the largest share of slices
    // variant 1 (bf16): operands via ds_read_b64_tr_b16 (needs whole 128-column tiles: the source swizzle permutes chunks
    // inside a 256-byte row); variant 0: 16-bit LDS gathers
    const bool tr = (variant & 1) && dtype == SNERF_DT_BF16 && (N % 128 == 0) && (K % 128 == 0);
    hipStream_t st = (hipStream_t)stream;
    if (f16) {
      static bool tn16_attr = false;
      if (!tn16_attr) {
        hipFuncSetAttribute((const void*)gemm_tn_kernel<__bf16, true, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * 8192);
        hipFuncSetAttribute((const void*)gemm_tn_kernel<__bf16, false, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * 8192);
        tn16_attr = true;
      }
      if (tr) { if (deep) hipLaunchKernelGGL((gemm_tn_kernel<__bf16, true, 4, true>), grid, dim3(256), lds, st, p);
                else hipLaunchKernelGGL((gemm_tn_kernel<__bf16, true, 2, true>), grid, dim3(256), lds, st, p); }
      else { if (deep) hipLaunchKernelGGL((gemm_tn_kernel<__bf16, false, 4, true>), grid, dim3(256), lds, st, p);
             else hipLaunchKernelGGL((gemm_tn_kernel<__bf16, false, 2, true>), grid, dim3(256), lds, st, p); }
    } else if (dtype == SNERF_DT_F32) {
      if (deep) hipLaunchKernelGGL((gemm_tn_kernel<float, false, 4>), grid, dim3(256), lds, st, p);
      else hipLaunchKernelGGL((gemm_tn_kernel<float, false, 2>), grid, dim3(256), lds, st, p);
    } else if (dtype == SNERF_DT_BF16 && tr) {
      if (deep) hipLaunchKernelGGL((gemm_tn_kernel<__bf16, true, 4>), grid, dim3(256), lds, st, p);
      else hipLaunchKernelGGL((gemm_tn_kernel<__bf16, true, 2>), grid, dim3(256), lds, st, p);
    } else if (dtype == SNERF_DT_BF16) {
      if (deep) hipLaunchKernelGGL((gemm_tn_kernel<__bf16, false, 4>), grid, dim3(256), lds, st, p);
      else hipLaunchKernelGGL((gemm_tn_kernel<__bf16, false, 2>), grid, dim3(256), lds, st, p);
    } else return SNERF_ERR_ARG;
  }
  if (ws != nullptr) {
    // (the same sums in the same order either way: the vector flavour only needs 16-byte aligned rows)
    const bool vec = k_valid % 4 == 0 && pl.part_ld % 4 == 0 && pl.part_stride % 4 == 0 && ldw % 4 == 0 && ((size_t)dW & 15) == 0 && ((size_t)ws & 15) == 0;
    if (pl.slices > 32) {                                  // (the 128 x 128 kernel's launches; the wide layers run 16 slices)
      const long pieces = (long)n_valid * (vec ? k_valid / 4 : k_valid);
      const dim3 fg((unsigned)((pieces + 63) / 64));
      if (vec) hipLaunchKernelGGL(tn_fold_many_kernel<true>, fg, dim3(1024), 0, (hipStream_t)stream, ws, pl.part_stride, pl.part_ld, pl.slices, n_valid, k_valid, dW, ldw);
      else hipLaunchKernelGGL(tn_fold_many_kernel<false>, fg, dim3(1024), 0, (hipStream_t)stream, ws, pl.part_stride, pl.part_ld, pl.slices, n_valid, k_valid, dW, ldw);
    } else if (vec)
      hipLaunchKernelGGL(tn_fold_kernel<true>, dim3((k_valid / 4 + 255) / 256, n_valid), dim3(256), 0, (hipStream_t)stream, ws, pl.part_stride,
                         pl.part_ld, pl.slices, n_valid, k_valid, dW, ldw);
    else
      hipLaunchKernelGGL(tn_fold_kernel<false>, dim3((k_valid + 255) / 256, n_valid), dim3(256), 0, (hipStream_t)stream, ws, pl.part_stride,
                         pl.part_ld, pl.slices, n_valid, k_valid, dW, ldw);
  }
  return snerf_check_launch();
}

extern "C" int snerf_linear_wgrad(const void* Z, long ldz, const void* X, long ldx, float* dW, long ldw, const void* zeros,
                                  int M, int N, int K, int n_valid, int k_valid, int dtype, int variant, void* stream) {
  return wgrad_launch(Z, ldz, X, ldx, dW, ldw, zeros, M, N, K, n_valid, k_valid, dtype, variant, nullptr, 0, stream);
}

// Deterministic weight gradient: the M slices store partial tiles into `ws` (snerf_linear_wgrad_ws_floats floats) and a second
// kernel folds them in slice order -- bit-reproducible run to run, no atomics ("deterministic mode").
extern "C" long snerf_linear_wgrad_ws_floats(int M, int N, int K, long ldz, long ldx, int dtype, int variant) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  if (dtype == 2) dtype = SNERF_DT_BF16;              // (SNERF_DT_F16 runs the bf16 kernels' plan)
  const TnPlan pl = tn_plan(M, N, K, ldz, ldx, dtype, variant, true);
  return pl.part_stride * pl.slices;
}

extern "C" int snerf_linear_wgrad_det(const void* Z, long ldz, const void* X, long ldx, float* dW, long ldw, const void* zeros, int M, int N,
                                      int K, int n_valid, int k_valid, int dtype, int variant, float* ws, long ws_floats, void* stream) {
  if (ws == nullptr) return SNERF_ERR_ARG;
  return wgrad_launch(Z, ldz, X, ldx, dW, ldw, zeros, M, N, K, n_valid, k_valid, dtype, variant, ws, ws_floats, stream);
}
