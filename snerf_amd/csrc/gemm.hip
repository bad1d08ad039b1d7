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
#include <type_traits>

#define ACT_NONE 0
#define ACT_RELU 1
#define ACT_MASK 2  // y = (aux > 0) ? y : 0   (ReLU backward fused into dgrad)

struct GemmNT {
  const void* A; long lda;
  const void* W; long ldw;
  const float* bias;
  void* Y; long ldy;
  const void* aux; long ldaux;
  float* colsum;
  int M, N, K, n_store, act, out_f32, vec_store;
  float* colsum_ws; int fast_epi;
  int dbg;  // ablation bits (tools/gemm_probe.py): 1 = no staging loads after tile 0, 2 = no LDS fragment reads, 4 = no stores
};

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

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__device__ __forceinline__ void glds16(const void* g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)lds_wave_base, 16, 0, 0);
}

// ---------------------------------------------------------------------------
// NT kernel: BM x BN output tile, WM x WN waves, each wave (BM/WM) x (BN/WN).
// ---------------------------------------------------------------------------
template <typename T, int BM, int BN, int WM, int WN, bool INTERLEAVE>
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
#pragma unroll
    for (int i = 0; i < LA; ++i) glds16(A + a_off[i] + k0, sA + (wave * LA + i) * 1024);
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

  // one staging piece (1 KiB per wave-instruction) of tile kt: pieces [0, LA) belong to A, [LA, LA+LB) to W
  auto issue_piece = [&](int kt, int stage, int pc) {
    char* sA = smem + stage * STAGE;
    char* sB = sA + BM * 128;
    const long k0 = (long)kt * BKE;
#pragma unroll
    for (int i = 0; i < LA; ++i) if (pc == i) glds16(A + a_off[i] + k0, sA + (wave * LA + i) * 1024);
#pragma unroll
    for (int i = 0; i < LB; ++i) if (pc == LA + i) glds16(W + b_off[i] + k0, sB + (wave * LB + i) * 1024);
  };
  auto read_frags = [&](const char* sA, const char* sB, int ks, frag_t* a, frag_t* b) {
    const int c = 2 * ks + chalf;
#pragma unroll
    for (int i = 0; i < TM; ++i) a[i] = *(const frag_t*)((p.dbg & 2) ? smem + lane * 16 : sA + ra[i] * 128 + ((c ^ ((ra[i] >> 1) & 7)) << 4));
#pragma unroll
    for (int j = 0; j < TN; ++j) b[j] = *(const frag_t*)((p.dbg & 2) ? smem + lane * 16 + 1024 : sB + rb[j] * 128 + ((c ^ ((rb[j] >> 1) & 7)) << 4));
  };

  issue(0, 0);
  if constexpr (!INTERLEAVE) {
    for (int kt = 0; kt < KT; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (kt + 1 < KT && !(p.dbg & 1)) issue(kt + 1, (kt + 1) & 1);
      const char* sA = smem + (kt & 1) * STAGE;
      const char* sB = sA + BM * 128;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        frag_t a[TM], b[TN];
        read_frags(sA, sB, ks, a, b);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) mma32(acc[i][j], b[j], a[i]);   // D = W-tile . X-tile^T: lanes own rows m
      }
    }
  } else {
    // Same double buffer, but the (LA+LB) staging instructions of the NEXT tile are spread between the MFMAs of this one
    // (an LDS-DMA piece costs ~60-100 issue cycles; issued as one block after the barrier they leave the matrix pipe
    // idle on every SIMD at once, since all waves of the workgroup are in the same phase), and the fragments of sub-step
    // ks+1 are fetched while the MFMAs of ks run.
    constexpr int NP = LA + LB, PPK = NP / 4;       // staging pieces per k sub-step
    static_assert(NP % 4 == 0, "pieces must split over the 4 k sub-steps");
    constexpr int NM = TM * TN;
    auto tile = [&](int kt, auto more_tag) {
      constexpr bool MORE = decltype(more_tag)::value;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const char* sA = smem + (kt & 1) * STAGE;
      const char* sB = sA + BM * 128;
      const int c0 = chalf;
      frag_t a[2][TM], b[2][TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[0][i] = *(const frag_t*)(sA + ra[i] * 128 + ((c0 ^ ((ra[i] >> 1) & 7)) << 4));
#pragma unroll
      for (int j = 0; j < TN; ++j) b[0][j] = *(const frag_t*)(sB + rb[j] * 128 + ((c0 ^ ((rb[j] >> 1) & 7)) << 4));
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (ks < 3) {
          const int c = 2 * (ks + 1) + chalf;
#pragma unroll
          for (int i = 0; i < TM; ++i) a[(ks + 1) & 1][i] = *(const frag_t*)(sA + ra[i] * 128 + ((c ^ ((ra[i] >> 1) & 7)) << 4));
#pragma unroll
          for (int j = 0; j < TN; ++j) b[(ks + 1) & 1][j] = *(const frag_t*)(sB + rb[j] * 128 + ((c ^ ((rb[j] >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int t = 0; t < NM; ++t) {
          const int i = t / TN, j = t % TN;
          mma32(acc[i][j], b[ks & 1][j], a[ks & 1][i]);
          // after every (NM/PPK)-th MFMA issue one staging piece of the next tile
          if constexpr (MORE) {
            if (((t + 1) % (NM / PPK)) == 0) issue_piece(kt + 1, (kt + 1) & 1, ks * PPK + (t + 1) / (NM / PPK) - 1);
          }
        }
      }
      // pin the software pipeline: fragments of sub-step ks+1 are fetched BEFORE the MFMAs of ks issue, staging pieces
      // sit between MFMA groups (masks: 0x008 MFMA, 0x010 VMEM, 0x100 DS read)
      __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (ks < 3) __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
#pragma unroll
        for (int g = 0; g < PPK; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, NM / PPK, 0);
          if constexpr (MORE) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        }
      }
    };
    for (int kt = 0; kt < KT - 1; ++kt) tile(kt, std::true_type{});
    tile(KT - 1, std::false_type{});
  }

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
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int g = 0; g < NCG; ++g) {
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
              bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
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
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int row = it * 8 + prow;
          const int m = m0 + wm * WTM + i * 32 + row;
          frag_t val = *(const frag_t*)(my + row * PITCH + pch * 16);
          if (m < p.M && ncol < p.n_store && !(p.dbg & 4)) {
            if (p.act == ACT_MASK) {
              const frag_t a8 = *(const frag_t*)(auxp + (long)m * p.ldaux + ncol);
#pragma unroll
              for (int e = 0; e < EPC; ++e) if (!((float)a8[e] > 0.f)) val[e] = (T)0.f;
            }
            *(frag_t*)((T*)p.Y + (long)m * p.ldy + ncol) = val;
#pragma unroll
            for (int e = 0; e < EPC; ++e) cs[g][e] += (float)val[e];
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
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
        if (m >= p.M || nb >= p.n_store || (p.dbg & 4)) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = acc[i][j][4 * q + e] + bv[e];
          if (p.act == ACT_RELU) v[e] = v[e] > 0.f ? v[e] : 0.f;
        }
        const bool full = nb + 3 < p.n_store;
        if (p.act == ACT_MASK) {
          const T* ap = aux + (long)m * p.ldaux + nb;
          if (full && vec_ok) {
            if constexpr (sizeof(T) == 2) {
              const bf16x4 a4 = *(const bf16x4*)ap;
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = (float)a4[e] > 0.f ? v[e] : 0.f;
            } else {
              const f32x4 a4 = *(const f32x4*)ap;
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = a4[e] > 0.f ? v[e] : 0.f;
            }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (nb + e < p.n_store) v[e] = to_f32(ap[e]) > 0.f ? v[e] : 0.f;
          }
        }
        if (full && vec_ok) {
          if (p.out_f32 || sizeof(T) == 4) {
            f32x4 o = {v[0], v[1], v[2], v[3]};
            *(f32x4*)((float*)p.Y + (long)m * p.ldy + nb) = o;
          } else {
            bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
            *(bf16x4*)((__bf16*)p.Y + (long)m * p.ldy + nb) = o;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (nb + e < p.n_store) {
              if (p.out_f32) ((float*)p.Y)[(long)m * p.ldy + nb + e] = v[e];
              else ((T*)p.Y)[(long)m * p.ldy + nb + e] = from_f32<T>(v[e]);
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

// out[n] += sum_r ws[r, n]  (bias gradient from the per-slab column sums of the data-gradient epilogue)
__global__ __launch_bounds__(256) void colsum_reduce_kernel(const float* __restrict__ ws, int rows, int N, int n_store, float* __restrict__ out) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= n_store) return;
  const int chunk = (rows + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * chunk, r1 = min(rows, r0 + chunk);
  float acc = 0.f;
  for (int r = r0; r < r1; ++r) acc += ws[(long)r * N + n];
  atomicAdd(out + n, acc);
}

template <typename T, int BM, int BN, int WM, int WN, bool IL>
static int launch_nt(const GemmNT& p, hipStream_t stream) {
  constexpr int LDS = 2 * (BM + BN) * 128;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)gemm_nt_kernel<T, BM, BN, WM, WN, IL>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set = true;
  }
  const int tiles_m = (p.M + BM - 1) / BM;
  const int tiles = tiles_m * (p.N / BN);
  hipLaunchKernelGGL((gemm_nt_kernel<T, BM, BN, WM, WN, IL>), dim3(tiles), dim3(64 * WM * WN), LDS, stream, p);
  if (p.fast_epi && p.colsum_ws != nullptr) {
    const int rows = tiles_m * WM;
    int ychunks = rows / 64;
    ychunks = ychunks < 1 ? 1 : (ychunks > 64 ? 64 : ychunks);
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3((p.n_store + 255) / 256, ychunks), dim3(256), 0, stream, p.colsum_ws, rows, p.N, p.n_store, p.colsum);
  }
  return snerf_check_launch();
}

extern "C" int snerf_linear_fwd(const void* A, long lda, const void* W, long ldw, const float* bias, void* Y, long ldy,
                                const void* aux, long ldaux, float* colsum, float* colsum_ws, int M, int N, int K, int n_store,
                                int act, int dtype, int out_f32, int variant, void* stream) {
  if (M <= 0) return SNERF_OK;
  if (N <= 0 || (N % 128) != 0 || K <= 0 || n_store <= 0 || n_store > N) return SNERF_ERR_ARG;
  if (dtype != SNERF_DT_F32 && dtype != SNERF_DT_BF16) return SNERF_ERR_ARG;
  const int bke = dtype == SNERF_DT_F32 ? 32 : 64;
  if (K % bke != 0 || lda % (bke / 8) != 0 || ldw % (bke / 8) != 0) return SNERF_ERR_ARG;
  if (act == ACT_MASK && aux == nullptr) return SNERF_ERR_ARG;
  // vector epilogue stores need 4-element alignment of the destination (and of the mask source)
  const long esz = (out_f32 || dtype == SNERF_DT_F32) ? 4 : 2;
  int vec = (ldy % 4 == 0) && (((uintptr_t)Y) % (4 * esz) == 0);
  if (act == ACT_MASK) vec = vec && (ldaux % 4 == 0) && (((uintptr_t)aux) % (dtype == SNERF_DT_F32 ? 16 : 8) == 0);
  // fast (LDS-transposed, 16-byte) epilogue: output in the compute dtype, 16-byte aligned row segments, whole chunks
  const int epc = dtype == SNERF_DT_F32 ? 4 : 8;
  int fast = !(out_f32 && dtype == SNERF_DT_BF16) && (ldy % epc == 0) && (((uintptr_t)Y) % 16 == 0) && (n_store % epc == 0);
  if (act == ACT_MASK) fast = fast && (ldaux % epc == 0) && (((uintptr_t)aux) % 16 == 0);
  if (colsum != nullptr && colsum_ws == nullptr) fast = 0;   // without a workspace the bias gradient uses the atomic path
  if ((variant >> 4) & 8) fast = 0;                          // ablation: force the direct-store epilogue
  GemmNT p{A, lda, W, ldw, bias, Y, ldy, aux, ldaux, colsum, M, N, K, n_store, act, out_f32, vec, colsum_ws, fast, (variant >> 4) & 7};
  variant &= 15;
  hipStream_t s = (hipStream_t)stream;
  // variant: 0 = 128x128 block-issue, 1 = 256x256 block-issue, 2 = 128x128 interleaved, 3 = 256x256 interleaved
  if (dtype == SNERF_DT_F32) return (variant & 2) ? launch_nt<float, 128, 128, 2, 2, true>(p, s) : launch_nt<float, 128, 128, 2, 2, false>(p, s);
  if ((variant & 1) && N % 256 == 0) return (variant & 2) ? launch_nt<__bf16, 256, 256, 2, 4, true>(p, s) : launch_nt<__bf16, 256, 256, 2, 4, false>(p, s);
  return (variant & 2) ? launch_nt<__bf16, 128, 128, 2, 2, true>(p, s) : launch_nt<__bf16, 128, 128, 2, 2, false>(p, s);
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
};

template <typename T, bool TR>
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
  const int n0 = (blockIdx.x / tiles_k) * 128, k0 = (blockIdx.x % tiles_k) * 128;
  const int mbeg = blockIdx.y * p.m_chunk;
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

  if (steps > 0) issue(0, 0);
  for (int st = 0; st < steps; ++st) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (st + 1 < steps) issue(st + 1, (st + 1) & 1);
    const char* sZ = smem + (st & 1) * 2 * TILEB;
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
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
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
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].v, b[j].v, acc[i][j], 0, 0, 0);
      }
    }
  }
  // D[i = n][j = k]: j = lane&31, i = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = k0 + wk * 64 + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (n < p.n_valid && k < p.k_valid) atomicAdd(p.dW + (long)n * p.ldw + k, acc[i][j][r]);
      }
    }
}

extern "C" int snerf_linear_wgrad(const void* Z, long ldz, const void* X, long ldx, float* dW, long ldw, const void* zeros,
                                  int M, int N, int K, int n_valid, int k_valid, int dtype, int variant, void* stream) {
  if (M <= 0) return SNERF_OK;
  const int epc = dtype == SNERF_DT_F32 ? 4 : 8;
  if (N < epc || K < epc || N % epc || K % epc || ldz % epc || ldx % epc || zeros == nullptr) return SNERF_ERR_ARG;
  // split M so that the grid has a few thousand blocks but each block still amortises its atomics
  const int tiles = ((N + 127) / 128) * ((K + 127) / 128);
  int chunks = (4096 + tiles - 1) / tiles;
  int m_chunk = (M + chunks - 1) / chunks;
  m_chunk = ((m_chunk + 255) / 256) * 256;
  if (m_chunk < 1024) m_chunk = 1024;
  chunks = (M + m_chunk - 1) / m_chunk;
  GemmTN p{Z, ldz, X, ldx, dW, ldw, zeros, M, N, K, n_valid, k_valid, m_chunk};
  const int lds = 2 * 2 * 8192;
  dim3 grid(tiles, chunks);
  // variant 1 (bf16): operands via ds_read_b64_tr_b16 (needs whole 128-column tiles: the source swizzle permutes chunks
  // inside a 256-byte row); variant 0: 16-bit LDS gathers
  const bool tr = (variant & 1) && dtype == SNERF_DT_BF16 && (N % 128 == 0) && (K % 128 == 0);
  if (dtype == SNERF_DT_F32) hipLaunchKernelGGL((gemm_tn_kernel<float, false>), grid, dim3(256), lds, (hipStream_t)stream, p);
  else if (dtype == SNERF_DT_BF16 && tr) hipLaunchKernelGGL((gemm_tn_kernel<__bf16, true>), grid, dim3(256), lds, (hipStream_t)stream, p);
  else if (dtype == SNERF_DT_BF16) hipLaunchKernelGGL((gemm_tn_kernel<__bf16, false>), grid, dim3(256), lds, (hipStream_t)stream, p);
  else return SNERF_ERR_ARG;
  return snerf_check_launch();
}
