// Fused multi-layer MLP for the 256-wide networks of the path (gfx950 / CDNA4): the whole network in ONE kernel, activations
// resident in REGISTERS from the first layer to the heads.
//
//   snerf_fmlp_classic_fwd  : NeRF 8 x 256 of the classic path (s-nerf/model/run_nerf_helpers.py:74-126 via run_network :460-474):
//                             pts embedding (63) -> 8 x [Linear 256 + ReLU], cat([pts, h]) after layer 4 -> alpha, feature,
//                             [feature | view embedding (27)] -> 128 ReLU -> rgb.  593 408 MAC per sample, 12 values out.
//   snerf_fmlp_proposal_fwd : proposal MLP of the live mip path (s-nerf/model/models.py:299-325): IPE (96) -> 4 x [256 + ReLU] -> 1.
//
// Why: as separate GEMM launches these layers are HBM-bound (K = N = 256: 128 FLOP per byte of activation traffic; measured
// 0.16 of the MFMA peak).  Here nothing but the encoded input (128 - 192 B) and the raw outputs (4 - 16 B) of a
// sample ever touches HBM.
//
// How (MI355X-first, not a chain of tiled GEMMs):
//  * a wave owns 32 samples (rows).  The MFMA runs "weights x activations": A operand = a 32 (outputs n) x 16 (k) block of W,
//    B operand = 16 (k) x 32 (rows) of the activations, D[n][row] accumulates in 16 VGPRs per 32 outputs.  Lane (row = lane & 31,
//    half = lane >> 5) then holds, for ITS row, outputs n = (r & 3) + 8 (r >> 2) + 4 half, r = 0..15 -- and the B operand of the
//    next layer wants, from that same lane, 8 reduction indices of that same row.  So after bias + ReLU + bf16 rounding the
//    accumulator registers r = 0..7 / 8..15 ARE the next layer's B fragments for two 16-wide k-steps; the only cost is that the
//    k-step's 16 reduction indices appear in the order {0-3, 8-11 | 4-7, 12-15} (lane halves), which the host bakes into the
//    packing of W (fmlp_perm in snerf_amd/mlp.py).  No LDS round trip, no transposition, no shuffles between layers.
//  * the weights are shared by the 8 waves of a workgroup (256 samples per tile): they stream through LDS as 1 KiB MFMA fragments
//    in exactly the order the code consumes them (packed once per parameter version on the host), 16 fragments per chunk, a ring
//    of chunks filled by `global_load_lds` (LDS-DMA) several chunks ahead; one s_barrier per chunk (16 MFMAs per wave) is the only
//    synchronisation.  A fragment is 64 lanes x 16 B contiguous: the DMA image is lane-linear and the ds_read_b128 of it is
//    conflict free without any swizzle.  The stream is continuous over the tiles a (persistent) workgroup walks.
//  * biases live in LDS for the whole kernel; an accumulator is INITIALISED with its bias by four broadcast ds_read_b128.
//  * per sample-tile the HBM traffic is the input fragments (16 B loads straight into the B-operand registers) and the raw head
//    outputs; the 1.2 MB weight stream of a tile comes from L2.
//
// Bound: MFMA (2.5 PFLOP/s dense bf16).  Algorithmic work: classic 1 186 816 FLOP / sample, proposal 442 880 FLOP / sample; the
// padded work the kernel executes is 1 212 416 / 458 752 (K and N rounded up to the 16 / 32 of the MFMA shape).
#include "common.h"
#include <utility>

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// Waves per workgroup: 8 (one workgroup per CU: ships) or 4 (two workgroups per CU, each with its own ring of half-size chunks, so that
// the two waves of a SIMD belong to different workgroups and do not reach their block ends together).  Measured (round 2, MI355X,
// 6.3 M rows): 4 waves 6.03 ms inference / 10.0 ms training forward, 8 waves 5.50 / 9.31 -- twice the L2 -> LDS weight traffic, half
// the read-ahead time and twice the barriers cost more than the desynchronisation gives.
#ifndef FM_WAVES
#define FM_WAVES 8
#endif
#define FM_CHUNK (2 * FM_WAVES)     // fragments per ring slot: every wave DMAs two of them
#define FM_SLOT (FM_CHUNK * 1024)   // bytes per ring slot
#define FM_RING 6                   // ring slots: five chunks in flight ahead of the one being consumed
#define FM_TILE_ROWS (32 * FM_WAVES)          // samples per workgroup tile
#define FM_WG_PER_CU (FM_WAVES == 4 ? 2 : 1)  // either way two waves per SIMD, 256 VGPRs each
#define FM_BIAS_MAX 128             // n-blocks of 32 outputs whose biases fit the LDS table (16 KiB)
#define FM_LOOK 4                   // weight fragments fetched from LDS ahead of the MFMA that uses them (a register queue)

struct FmlpArgs {
  const __bf16* E;  long ldE;       // encoded input rows [M, ldE] (pts embedding 63 -> 64, or IPE 96)
  const __bf16* VE; long ldVE;      // view-direction embedding rows [M, ldVE] (27 -> 32), classic network only
  const float* pts;                 // classic network, in-kernel embedding: sample positions [M,3] ...
  const float* viewdirs; long ldvd; // ... and per-ray view directions [M / S, ldvd]
  int S;                            // samples per ray
  __bf16* act[12]; long act_ld[12]; // training forward: where the output of layer i is stored (bf16 [M, >= width], row stride act_ld)
  unsigned* bits[9];                // ... and the ReLU bit masks of the 256-wide layers (layout of ACT_RELU_BITS in gemm.hip); classic: [8] = views_linears.0 (128 wide)
  const char* wstream;              // n_chunks x 16 KiB of MFMA fragments in consumption order
  const float* bias;                // n_blocks x 32 floats in consumption order
  float* out;                       // classic: raw [M,4] = (rgb, sigma); proposal: raw density [M]
  long M;
  int tiles, n_chunks, n_blocks;
};

// ---- the weight stream ---------------------------------------------------------------------------------------------------------
// g = chunks consumed so far by this workgroup (all tiles); chunk g of the stream lives in ring slot g % FM_RING and holds stream
// chunk g % n_chunks.  Every wave DMAs 2 of the 16 fragments of a chunk.
struct WStream {
  const char* src;          // this lane's source pointer inside chunk 0 (piece 2 * wave, + lane * 16)
  unsigned ring;            // LDS byte address of the ring
  unsigned my_piece;        // wave * 2048
  unsigned slot_off;        // byte offset of the slot being consumed
  int fill_slot, fill_chunk;  // ring slot / stream chunk of the next DMA
  int n_chunks;
};

template <int RING>
__device__ __forceinline__ void ws_issue(WStream& w, char* smem) {
  const char* g = w.src + (long)w.fill_chunk * FM_SLOT;
  char* dst = smem + w.fill_slot * FM_SLOT + w.my_piece;
  __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)dst, 16, 0, 0);
  __builtin_amdgcn_global_load_lds((gbl_ptr_t)(g + 1024), (lds_ptr_t)(dst + 1024), 16, 0, 0);
  w.fill_slot = w.fill_slot + 1 == RING ? 0 : w.fill_slot + 1;
  w.fill_chunk = w.fill_chunk + 1 == w.n_chunks ? 0 : w.fill_chunk + 1;
}

// chunk boundary: the chunk about to be read has landed for every wave, the chunk just finished is free for the DMA.
// vmcnt(2 (RING - 2)): of this wave's pieces only those of the RING - 2 youngest chunks may still be in flight, i.e. the
// pieces of the chunk we are about to read are in LDS (loads retire in order; other loads / stores in flight only make the wait
// stricter).  lgkmcnt(0): this wave's fragment reads of the finished chunk have returned.  The barrier then (a) extends the first
// fact to the other waves' pieces and (b) the second to the other waves' reads of the slot that is refilled right after it.
// One volatile asm with a memory clobber: no LDS access of the compiler's may move across it.
// (Withdrawn variants of this boundary -- a wave pair running half a chunk apart, arrival counters instead of the workgroup barrier, an
// extra vmcnt slack -- are kept with their measurements in tools/probes/fmlp_experiments.hip; DESIGN.md section 6b lists them.)
template <int RING>
__device__ __forceinline__ void ws_sync_issue(WStream& w, char* smem) {
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(2 * (RING - 2)) : "memory");
  ws_issue<RING>(w, smem);                              // chunk g + RING - 1 into the slot chunk g - 1 occupied
}
__device__ __forceinline__ void ws_cross(WStream& w, int ring) {
  w.slot_off = w.slot_off + FM_SLOT == ring * FM_SLOT ? 0 : w.slot_off + FM_SLOT;
}
template <int RING>
__device__ __forceinline__ void ws_advance(WStream& w, char* smem) {
  ws_sync_issue<RING>(w, smem);
  ws_cross(w, RING);
}

// ---- building blocks -----------------------------------------------------------------------------------------------------------
template <int RING, bool F16 = false>
struct CtxT {
  static constexpr int ring = RING;   // slots of the weight ring: FM_RING for the 256-wide networks, fewer where the LDS is needed elsewhere
  static constexpr bool f16 = F16;    // fp16 flavour (path C's compute="fp16"): the 16-bit operands are fp16 bit patterns, f16 MFMA and conversions
  char* smem;
  WStream ws;
  const char* frag_base;   // ring + lane * 16
  const char* bias_lds;    // bias table + (lane >> 5) * 16
  bf16x8 q[FM_LOOK];       // the next FM_LOOK fragments, already on their way from LDS
};
typedef CtxT<FM_RING> Ctx;

// Next weight fragment (A operand: 32 outputs x 16 reduction indices).  The ds_read of fragment F + FM_LOOK is issued when fragment
// F is handed out, so FM_LOOK - 1 MFMAs (and the partner wave's) cover the LDS latency; the queue runs across blocks, layers and
// tiles (the stream is one sequence).  The chunk boundary is taken when the READ-AHEAD crosses it.  F (the fragment's position in
// the network pass) and every index derived from it are template arguments: nothing here depends on the optimiser proving a
// counter constant.
// (Issuing these reads by hand with a hand-counted lgkmcnt gained 2 %: withdrawn, tools/probes/fmlp_experiments.hip.)
template <int F, typename C>
__device__ __forceinline__ bf16x8 next_frag(C& c) {
  bf16x8 w = c.q[F % FM_LOOK];
  constexpr int G = F + FM_LOOK;
  if constexpr ((G % FM_CHUNK) == 0) ws_advance<C::ring>(c.ws, c.smem);
  c.q[F % FM_LOOK] = *(const bf16x8*)(c.frag_base + c.ws.slot_off + (G % FM_CHUNK) * 1024);
  return w;
}

// accumulator of the 32-output block B of the pass, initialised with its bias: lane (row, half) owns outputs 8 q + 4 half + e
template <int B, typename C>
__device__ __forceinline__ f32x16 acc_init(const C& c) {
  f32x16 acc;
  const char* a = c.bias_lds + B * 128;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 b = *(const f32x4*)(a + q * 32);
    acc[4 * q + 0] = b[0]; acc[4 * q + 1] = b[1]; acc[4 * q + 2] = b[2]; acc[4 * q + 3] = b[3];
  }
  return acc;
}

template <int F, int NK, int... I, typename C>
__device__ __forceinline__ void mac_seq(C& c, f32x16& acc, const bf16x8 (&in)[NK], std::integer_sequence<int, I...>) {
  if constexpr (C::f16) {
    typedef _Float16 fm_f16x8 __attribute__((ext_vector_type(8)));
    ((acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(fm_f16x8, next_frag<F + I>(c)), __builtin_bit_cast(fm_f16x8, in[I]), acc, 0, 0, 0)), ...);
  } else
  ((acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(next_frag<F + I>(c), in[I], acc, 0, 0, 0)), ...);
}
template <int F, int NK, typename C>
__device__ __forceinline__ void mac(C& c, f32x16& acc, const bf16x8 (&in)[NK]) {
  mac_seq<F, NK>(c, acc, in, std::make_integer_sequence<int, NK>{});
}

// accumulator -> the two B fragments (k-steps 2 j, 2 j + 1) of the next layer
typedef unsigned fm_u32x4 __attribute__((ext_vector_type(4)));
template <bool RELU, bool F16 = false>
__device__ __forceinline__ void to_frags(const f32x16& acc, bf16x8& lo, bf16x8& hi) {
  typedef __attribute__((ext_vector_type(8))) float f32x8;
  const f32x8 a = {acc[0], acc[1], acc[2], acc[3], acc[4], acc[5], acc[6], acc[7]};
  const f32x8 b = {acc[8], acc[9], acc[10], acc[11], acc[12], acc[13], acc[14], acc[15]};
  if constexpr (F16) {                                   // (an fp16 is negative iff its 16 bits are a negative int16 too: the same packed max below)
    typedef _Float16 fm_f16x8 __attribute__((ext_vector_type(8)));
    lo = __builtin_bit_cast(bf16x8, __builtin_convertvector(a, fm_f16x8));
    hi = __builtin_bit_cast(bf16x8, __builtin_convertvector(b, fm_f16x8));
  } else {
  lo = __builtin_convertvector(a, bf16x8);               // v_cvt_pk_bf16_f32: two values per instruction
  hi = __builtin_convertvector(b, bf16x8);
  }
  if (RELU) {
    // ReLU on the rounded value (rounding is monotone and keeps the sign, so round-then-clamp == clamp-then-round): a bf16 is
    // negative iff its 16 bits are a negative int16, so a packed signed max with 0 clamps two values per instruction (-0 -> +0).
    // The max must be an instruction the COMPILER emits: its result is an MFMA operand, and a VALU write needs wait states before an
    // MFMA reads the register (tools/probes/mfma_war_probe.hip: zero gap = 100 % wrong results).  hipcc counts them for its own
    // instructions but put a single `s_nop 0` behind an inline-asm v_pk_max_i16 -- in the one instantiation whose schedule placed an
    // MFMA right there that was a rare, timing-dependent wrong fragment (round 2, tools/stress_fmlp_variants.py).  The EMPTY asm
    // only hides the conversion's provenance: without it hipcc converts every value separately and re-packs them with v_perm_b32.
    typedef short fm_s16x8 __attribute__((ext_vector_type(8)));
    const fm_s16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    fm_u32x4 ul = __builtin_bit_cast(fm_u32x4, lo), uh = __builtin_bit_cast(fm_u32x4, hi);
    asm("" : "+v"(ul), "+v"(uh));
    lo = __builtin_bit_cast(bf16x8, __builtin_elementwise_max(__builtin_bit_cast(fm_s16x8, ul), zero));
    hi = __builtin_bit_cast(bf16x8, __builtin_elementwise_max(__builtin_bit_cast(fm_s16x8, uh), zero));
  }
}

// Training forward: the layer outputs also go to HBM (the weight gradient reads them, the data gradient takes its ReLU mask from
// them).  A lane holds 4 x 4 columns of ITS row of a block -- stored from there, an instruction would write 16..32-byte pieces of 32
// rows (measured: the 8 x 256 network at 2.2..2.8 TB/s of stores, twice the time of the launch without them).  Instead every PAIR of
// blocks (64 columns) is transposed through a 4 KiB per-wave LDS slab, the way the GEMM epilogue does it: lane (r, half) writes its
// four 8-byte groups of both blocks (16-byte chunks XOR-swizzled by the row), then lane l reads the chunks (row 8 it + l / 8,
// chunk l % 8), it = 0..3 -- each store instruction covers 8 full 128-byte row segments.  In that layout the lane's four chunks are
// exactly word l of the ReLU bit-mask block ACT_MASK_BITS consumes (gemm.hip: block (row / 32, column / 64) of 64 words; byte `it` of
// word l = row 8 it + l / 8, columns 8 (l % 8) .. + 7), so the masks cost one 4-byte store per lane and pair.
// The slab is wave-private: LDS operations of a wave execute in order, no barrier is involved.
struct StoreTo {
  __bf16* y; long ld;        // output buffer of the layer, row stride
  unsigned* bits;            // its ReLU bit mask words (nullptr for the layers without one)
  long row0, M;              // first row of this wave's 32-row block (wave-uniform), rows of the launch
  char* slab;                // this wave's 4 KiB of LDS
  int lane;
  int ncg;                   // 64-column groups of the layer (its width / 64): row length of the bit-mask block grid
};

// The row stores are streaming (nt) stores -- nothing of a launch reads them again, and without the hint the 1.2 KB per row
// that pass through an XCD's L2 evict the weight stream every workgroup re-reads per tile (gemm.hip, the epilogue units' stores).
// Measured (round 3, A/B/A on one box): training forward 9.56 -> 9.31 ms per 6.3 M rows, classic gradient chain 10.4-10.9 -> 10.3 ms,
// path-B train step 47.9 / 48.2 -> 46.8 ms.
template <bool BITS, int J>
__device__ __forceinline__ void store_block(const StoreTo& st, const bf16x8& lo, const bf16x8& hi) {
  const int r = st.lane & 31, half = st.lane >> 5;
  char* w = st.slab + r * 128 + 8 * half;
  const fm_u32x4 l = __builtin_bit_cast(fm_u32x4, lo), h = __builtin_bit_cast(fm_u32x4, hi);
  typedef unsigned fm_u32x2 __attribute__((ext_vector_type(2)));
  constexpr int C0 = 4 * (J & 1);                                        // first 16-byte chunk of this block inside the 64-column pair
  *(fm_u32x2*)(w + (((C0 + 0) ^ (r & 7)) << 4)) = fm_u32x2{l[0], l[1]};
  *(fm_u32x2*)(w + (((C0 + 1) ^ (r & 7)) << 4)) = fm_u32x2{l[2], l[3]};
  *(fm_u32x2*)(w + (((C0 + 2) ^ (r & 7)) << 4)) = fm_u32x2{h[0], h[1]};
  *(fm_u32x2*)(w + (((C0 + 3) ^ (r & 7)) << 4)) = fm_u32x2{h[2], h[3]};
  if constexpr ((J & 1) == 1) {
    const int prow = st.lane >> 3, pch = st.lane & 7;
    __bf16* dst = st.y + (st.row0 + prow) * st.ld + 64 * (J >> 1) + 8 * pch;
    unsigned mw = 0;
    // the four read-backs queue right behind the writes (the LDS executes one wave's instructions in order: no wait in between) and
    // are pinned ahead of the row-bound branches, so that their latencies overlap instead of being paid one by one inside the branches
    fm_u32x4 vs[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = 8 * it + prow;
      vs[it] = *(const fm_u32x4*)(st.slab + row * 128 + ((pch ^ (row & 7)) << 4));
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) asm volatile("" : "+v"(vs[it]));
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = 8 * it + prow;
      const fm_u32x4 v = vs[it];
      if (st.row0 + row < st.M) {
        __builtin_nontemporal_store(v, (fm_u32x4*)(dst + (long)(8 * it) * st.ld));
        if constexpr (BITS) {
          // a ReLU output is > 0 iff its 16 bits are not 0: min(half word, 1), even elements gathered in bits 0, 2, 4, 6, odd ones 16 higher
          typedef unsigned short fm_u16x2 __attribute__((ext_vector_type(2)));
          const fm_u16x2 one = {1, 1};
          unsigned z = 0;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const unsigned x = v[k];                   // (named scalar: __builtin_bit_cast of a vector-element lvalue is miscompiled)
            z |= __builtin_bit_cast(unsigned, (fm_u16x2)__builtin_elementwise_min(__builtin_bit_cast(fm_u16x2, x), one)) << (2 * k);
          }
          mw |= ((z | (z >> 15)) & 0xffu) << (8 * it);
        }
      }
    }
    if constexpr (BITS) st.bits[((st.row0 >> 5) * st.ncg + (J >> 1)) * 64 + st.lane] = mw;   // 256 contiguous bytes per wave; rows >= M: zeros
  }
}

// (Cutting this store path into pieces that ride between the next block's MFMAs was built and measured 4 % slower: withdrawn,
// tools/probes/fmlp_experiments.hip.)

// One layer: out[32 NB] = act(W . [in0 | in1] + b) -- NB blocks of 32 outputs over one or two input segments (skip connections and
// concatenations are never formed).  F = first fragment, B = first bias block of the layer within the pass.
template <int F, int B, int NK0, int NK1, bool RELU, bool STORE, bool BITS, int J, bool MORE, int NOUT, typename C>
__device__ __forceinline__ void dense_block(C& c, f32x16& acc, const bf16x8 (&in0)[NK0], const bf16x8 (&in1)[NK1 > 0 ? NK1 : 1], bf16x8 (&out)[NOUT],
                                            const StoreTo& st) {
  bf16x8& lo = out[2 * J];
  bf16x8& hi = out[2 * J + 1];
  mac<F, NK0>(c, acc, in0);
  if constexpr (NK1 > 0) mac<F + NK0, NK1>(c, acc, in1);
  to_frags<RELU, C::f16>(acc, lo, hi);
  if constexpr (STORE) store_block<BITS, J>(st, lo, hi);
  // (issuing these bias reads BEFORE the stores, so that their LDS latency runs under them, measured 9.53 vs 9.43 ms: the sixteen
  // accumulator registers are then live across the store path of a kernel that already sits at the 256-register limit)
  if constexpr (MORE) acc = acc_init<B + 1>(c);
}
template <int F, int B, int NK0, int NK1, int NB, bool RELU, bool STORE, bool BITS, int... J, typename C>
__device__ __forceinline__ void dense_seq(C& c, const bf16x8 (&in0)[NK0], const bf16x8 (&in1)[NK1 > 0 ? NK1 : 1], bf16x8 (&out)[2 * NB],
                                          const StoreTo& st, std::integer_sequence<int, J...>) {
  static_assert(NB % 2 == 0, "the training stores work on pairs of blocks");
  f32x16 acc = acc_init<B>(c);
  (dense_block<F + J * (NK0 + NK1), B + J, NK0, NK1, RELU, STORE, BITS, J, (J + 1 < NB)>(c, acc, in0, in1, out, st), ...);
}
// BITS (training stores): the ReLU bit masks; by default for the 256-wide ReLU layers (st.ncg = 4), explicitly for the colour head's
template <int F, int B, int NK, int NB, bool RELU, bool STORE = false, bool BITS = (STORE && RELU && NB == 8), typename C>
__device__ __forceinline__ void dense(C& c, const bf16x8 (&in)[NK], bf16x8 (&out)[2 * NB], const StoreTo& st = StoreTo{}) {
  const bf16x8 none[1] = {};
  dense_seq<F, B, NK, 0, NB, RELU, STORE, BITS>(c, in, none, out, st, std::make_integer_sequence<int, NB>{});
}
template <int F, int B, int NK0, int NK1, int NB, bool RELU, bool STORE = false, bool BITS = (STORE && RELU && NB == 8), typename C>
__device__ __forceinline__ void dense2(C& c, const bf16x8 (&in0)[NK0], const bf16x8 (&in1)[NK1], bf16x8 (&out)[2 * NB], const StoreTo& st = StoreTo{}) {
  dense_seq<F, B, NK0, NK1, NB, RELU, STORE, BITS>(c, in0, in1, out, st, std::make_integer_sequence<int, NB>{});
}

// input fragments straight from HBM: lane (row, half) reads the 16 bytes [16 s + 8 half, +8) of its row (natural k order)
template <int NK>
__device__ __forceinline__ void load_rows(const __bf16* p, long ld, long row, int half, bf16x8 (&out)[NK]) {
  const __bf16* q = p + row * ld + half * 8;
#pragma unroll
  for (int s = 0; s < NK; ++s) out[s] = *(const bf16x8*)(q + 16 * s);
}

// ---- in-kernel positional encoding (run_nerf_helpers.py:22-52: [x, sin(2^0 x), cos(2^0 x), sin(2^1 x), cos(2^1 x), ...]) -------------
// Fragment s of the embedding in natural order: lane half h supplies features k = 16 s + 8 h + e, e = 0..7.  Which (frequency band,
// sin / cos, component) a feature is, is a compile-time fact per (s, e, half); the lane picks its half's constants with v_cndmask.
// sin(2^b x) is evaluated in REVOLUTIONS: 2^b x / (2 pi) as an exact-product two-term reduction (p = x * fl(2^b / 2pi), its rounding
// error recovered by an fma, plus the low word of 1 / 2pi), v_fract_f32, then the hardware's v_sin_f32 (cos = sin(. + 1/4)).  The
// absolute error is a few 1e-7, far below the half-ulp of the bf16 rounding the feature undergoes next (2e-3 at 1); a fraction of
// a per cent of the features land on the other side of a bf16 rounding boundary compared with the separate embedding kernel's
// sinf / cosf (that kernel stays the bit-exact fp32 reference, tests/test_gpu_kernels.py).
__device__ __forceinline__ constexpr int fe_comp(int k) { return k < 3 ? k : (k - 3) % 3; }
__device__ __forceinline__ constexpr int fe_band(int k) { return k < 3 ? 0 : (k - 3) / 6; }
__device__ __forceinline__ constexpr bool fe_cos(int k) { return k >= 3 && ((k - 3) % 6) >= 3; }

template <int K, int WIDTH>
__device__ __forceinline__ constexpr float fe_scale_hi() {   // fl(2^band / 2pi); 0 for padding features
  return K >= WIDTH ? 0.f : 0.15915494f * (float)(1 << fe_band(K));
}
template <int K, int WIDTH>
__device__ __forceinline__ constexpr float fe_scale_lo() {   // 2^band * (1/2pi - fl(1/2pi))
  return K >= WIDTH ? 0.f : 6.4206383e-9f * (float)(1 << fe_band(K));
}

struct Vec3 { float c0, c1, c2; };                              // scalars, not an array: a select between two array elements
template <int C>                                               // would become a dynamically indexed (scratch) access
__device__ __forceinline__ float pick(const Vec3& x) { return C == 0 ? x.c0 : (C == 1 ? x.c1 : x.c2); }

template <int K0, int WIDTH>
__device__ __forceinline__ float embed_feature(const Vec3& x, bool hi_half) {
  constexpr int K1 = K0 + 8;
  const float xs = hi_half ? pick<fe_comp(K1)>(x) : pick<fe_comp(K0)>(x);
  const float sh = hi_half ? fe_scale_hi<K1, WIDTH>() : fe_scale_hi<K0, WIDTH>();
  const float sl = hi_half ? fe_scale_lo<K1, WIDTH>() : fe_scale_lo<K0, WIDTH>();
  const float q = hi_half ? (fe_cos(K1) ? 0.25f : 0.f) : (fe_cos(K0) ? 0.25f : 0.f);
  const float p = xs * sh;
  const float err = __builtin_fmaf(xs, sh, -p);                 // exact rounding error of the product
  const float t = __builtin_amdgcn_fractf(p) + (__builtin_fmaf(xs, sl, err) + q);
  float v = __builtin_amdgcn_sinf(t);
  if (K0 < 3) v = hi_half ? v : pick<K0 < 3 ? K0 : 0>(x);       // identity features (half 0 of k-step 0 only)
  if (K1 >= WIDTH) v = hi_half ? 0.f : v;                       // zero padding (the packed weights are zero there too)
  if (K0 >= WIDTH) v = 0.f;
  return v;
}

template <int S, int WIDTH>
__device__ __forceinline__ bf16x8 embed_frag(const Vec3& x, bool hi_half) {
  typedef __attribute__((ext_vector_type(8))) float f32x8;
  const f32x8 v = {embed_feature<16 * S + 0, WIDTH>(x, hi_half), embed_feature<16 * S + 1, WIDTH>(x, hi_half),
                   embed_feature<16 * S + 2, WIDTH>(x, hi_half), embed_feature<16 * S + 3, WIDTH>(x, hi_half),
                   embed_feature<16 * S + 4, WIDTH>(x, hi_half), embed_feature<16 * S + 5, WIDTH>(x, hi_half),
                   embed_feature<16 * S + 6, WIDTH>(x, hi_half), embed_feature<16 * S + 7, WIDTH>(x, hi_half)};
  return __builtin_convertvector(v, bf16x8);
}

template <int NK, int WIDTH>
__device__ __forceinline__ void embed_frags(const Vec3& x, bool hi_half, bf16x8 (&out)[NK]) {
  out[0] = embed_frag<0, WIDTH>(x, hi_half);
  if constexpr (NK > 1) out[1] = embed_frag<1, WIDTH>(x, hi_half);
  if constexpr (NK > 2) out[2] = embed_frag<2, WIDTH>(x, hi_half);
  if constexpr (NK > 3) out[3] = embed_frag<3, WIDTH>(x, hi_half);
  static_assert(NK <= 4, "embedding wider than 64 features");
}

struct ClassicInputs { bf16x8 e[4], ve[2]; };
__device__ __forceinline__ ClassicInputs classic_embed_inputs(const float* pp, const float* vp, bool hi_half) {
  const Vec3 x{pp[0], pp[1], pp[2]}, v{vp[0], vp[1], vp[2]};
  ClassicInputs r;
  embed_frags<4, 63>(x, hi_half, r.e);                    // get_embedder(10): 63 features (+ 1 zero)
  embed_frags<2, 27>(v, hi_half, r.ve);                   // get_embedder(4): 27 features (+ 5 zeros)
  return r;
}

// kernel prologue shared by the fused kernels: stream state, biases into LDS, the first chunks into the ring, the fragment queue
template <typename C>
__device__ __forceinline__ void ctx_start(C& c, char* smem, const char* wstream, int n_chunks, const float* bias, int n_blocks, int tid,
                                          int wave, int lane) {
  constexpr int RING = C::ring;
  float* bias_tab = (float*)(smem + RING * FM_SLOT);
  c.smem = smem;
  c.ws.src = wstream + wave * 2048 + lane * 16;
  c.ws.ring = (unsigned)(size_t)smem;
  c.ws.my_piece = wave * 2048;
  c.ws.slot_off = (RING - 1) * FM_SLOT;                 // "chunk -1": the first boundary advances to slot 0
  c.ws.fill_slot = 0;
  c.ws.fill_chunk = 0;
  c.ws.n_chunks = n_chunks;
  c.frag_base = smem + lane * 16;
  c.bias_lds = (const char*)bias_tab + (lane >> 5) * 16;

  // prologue: biases into LDS (plain stores), the first RING - 1 chunks of the stream into the ring
  for (int i = tid; i < n_blocks * 32; i += 64 * FM_WAVES) bias_tab[i] = bias[i];
#pragma unroll
  for (int i = 0; i < RING - 1; ++i) ws_issue<RING>(c.ws, smem);
  // first boundary: chunk 0 has landed for everybody (and the bias stores are visible); start the fragment queue
  ws_advance<RING>(c.ws, smem);
#pragma unroll
  for (int i = 0; i < FM_LOOK; ++i) c.q[i] = *(const bf16x8*)(c.frag_base + c.ws.slot_off + i * 1024);
}

// 8 x 4 + 4 x 128 + 160 + 2 x 128 (trunk) + 16 (alpha) + 128 (feature) + 72 (views) + 8 (rgb) fragments; 64 + 1 + 8 + 4 + 1 blocks
#define FMLP_CLASSIC_FRAGS 1184
#define FMLP_CLASSIC_BLOCKS 78
#define FMLP_CLASSIC 0
#define FMLP_PROPOSAL 1

template <int NET, bool EMBED, bool STORE>
__global__ __launch_bounds__(64 * FM_WAVES, 2) void fmlp_kernel(FmlpArgs a) {   // (second argument: waves per SIMD -> at most 256 VGPRs)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;

  CtxT<FM_RING> c;
  ctx_start(c, smem, a.wstream, a.n_chunks, a.bias, a.n_blocks, tid, wave, lane);

  for (int tile = blockIdx.x; tile < a.tiles; tile += gridDim.x) {
    long row = (long)tile * FM_TILE_ROWS + wave * 32 + (lane & 31);
    const bool row_ok = row < a.M;
    row = row_ok ? row : a.M - 1;                       // tail rows: compute on a valid row, store nothing

    if constexpr (NET == FMLP_CLASSIC) {
      bf16x8 e[4], ve[2], p[16], q[16];
      if constexpr (EMBED) {
        const ClassicInputs in = classic_embed_inputs(a.pts + row * 3, a.viewdirs + (long)((unsigned)row / (unsigned)a.S) * a.ldvd, half != 0);   // (M < 2^31 rows, checked by the launcher)
#pragma unroll
        for (int i = 0; i < 4; ++i) e[i] = in.e[i];
        ve[0] = in.ve[0]; ve[1] = in.ve[1];
      } else {
        load_rows<4>(a.E, a.ldE, row, half, e);
        load_rows<2>(a.VE, a.ldVE, row, half, ve);
      }
      // fragment / bias-block offsets of the layers inside the pass (8 blocks x k-steps each)
      constexpr int F1 = 8 * 4, F2 = F1 + 128, F3 = F2 + 128, F4 = F3 + 128, F5 = F4 + 128, F6 = F5 + 8 * 20, F7 = F6 + 128;
      constexpr int FA = F7 + 128, FF = FA + 16, FV = FF + 128, FR = FV + 4 * 18;
      static_assert(FR + 8 == FMLP_CLASSIC_FRAGS, "classic network: fragment count");
      auto to = [&](int i) { return StoreTo{a.act[i], a.act_ld[i], i < 8 ? a.bits[i] : nullptr, (long)tile * FM_TILE_ROWS + wave * 32, a.M, smem + FM_RING * FM_SLOT + FM_BIAS_MAX * 128 + wave * 4096, lane, 4}; };
      dense<0, 0, 4, 8, true, STORE>(c, e, p, to(0));             // pts_linears.0
      dense<F1, 8, 16, 8, true, STORE>(c, p, q, to(1));           // .1
      dense<F2, 16, 16, 8, true, STORE>(c, q, p, to(2));          // .2
      dense<F3, 24, 16, 8, true, STORE>(c, p, q, to(3));          // .3
      dense<F4, 32, 16, 8, true, STORE>(c, q, p, to(4));          // .4  (skip: the next layer reads cat([pts, h]))
      dense2<F5, 40, 4, 16, 8, true, STORE>(c, e, p, q, to(5));   // .5
      dense<F6, 48, 16, 8, true, STORE>(c, q, p, to(6));          // .6
      dense<F7, 56, 16, 8, true, STORE>(c, p, q, to(7));          // .7
      f32x16 alpha = acc_init<64>(c);                             // alpha_linear: output 0 of one block
      mac<FA, 16>(c, alpha, q);
      const float sigma = alpha[0];
      dense<FF, 65, 16, 8, false, STORE>(c, q, p, to(8));         // feature_linear (no activation)
      bf16x8 hv[8];
      const StoreTo to_hv{a.act[9], a.act_ld[9], a.bits[8], (long)tile * FM_TILE_ROWS + wave * 32, a.M, smem + FM_RING * FM_SLOT + FM_BIAS_MAX * 128 + wave * 4096, lane, 2};
      dense2<FV, 73, 16, 2, 4, true, STORE, STORE>(c, p, ve, hv, to_hv); // views_linears.0 on cat([feature, views]) (masks: 2 column groups)
      f32x16 rgb = acc_init<77>(c);                     // rgb_linear: outputs 0..2
      mac<FR, 8>(c, rgb, hv);
      if (row_ok && half == 0) {
        const f32x4 o = {rgb[0], rgb[1], rgb[2], sigma};
        *(f32x4*)(a.out + row * 4) = o;
      }
    } else {
      bf16x8 e[6], p[16], q[16];
      load_rows<6>(a.E, a.ldE, row, half, e);
      auto to = [&](int i) { return StoreTo{a.act[i], a.act_ld[i], i < 8 ? a.bits[i] : nullptr, (long)tile * FM_TILE_ROWS + wave * 32, a.M, smem + FM_RING * FM_SLOT + FM_BIAS_MAX * 128 + wave * 4096, lane, 4}; };
      dense<0, 0, 6, 8, true, STORE>(c, e, p, to(0));             // layers.0
      dense<48, 8, 16, 8, true, STORE>(c, p, q, to(1));
      dense<48 + 128, 16, 16, 8, true, STORE>(c, q, p, to(2));
      dense<48 + 256, 24, 16, 8, true, STORE>(c, p, q, to(3));
      f32x16 d = acc_init<32>(c);                       // density_layer
      mac<48 + 384, 16>(c, d, q);
      if (row_ok && half == 0) a.out[row] = d[0];
    }
    // the fragment count of a network pass is a whole number of chunks (the host pads the stream), so the next tile starts on a
    // chunk boundary again -- and the queue already holds its first FM_LOOK fragments
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // drain the read-ahead before the LDS is released
}

// =================================================================================================================================
// NeRF MLP of the zipnerf path at inference (s-nerfpp/zipnerf/internal/models.py:462-479, 586-703 on the waymo.gin branch, deg_view = 1, no GLO
// vectors): grid features (40 -> 64 columns) -> Linear 64 + ReLU -> Linear 256 (x; channel 0 = raw density) -> cat([x, dir_enc 9])
// -> 256 ReLU -> cat([h, x, dir_enc]) -> 256 ReLU -> Linear 3.  As seven GEMM-shaped launches every layer is bound by the 0.5 - 2.4 GB of
// activations it moves per 65 536-ray chunk (2.2 ms per chunk of a 1920 x 1280 frame); here a row's 160 bytes of inputs and 16 bytes of
// outputs are all that touches HBM.  Built from the blocks above; two things are particular:
//   * the raw density is output 0 of density_layer.2 BEFORE its rounding to the compute dtype (the per-layer path evaluates that row once
//     more as an fp32 head): one extra 32-output block over the same 4 k-steps;
//   * the last hidden layer is never held: as soon as a block of 32 of its outputs is rounded, the two k-steps of the rgb layer that
//     consume it are multiplied into the rgb accumulator (the host interleaves the rgb weights block by block) -- x, the first hidden
//     layer and the direction encoding (33 fragments) are what a lane keeps while the 264 MFMAs of that layer run.
// 460 fragments (29 chunks), 35 bias blocks.  F16: fp16 operands (compute="fp16").
#define FZIP_FRAGS 460
#define FZIP_BLOCKS 35
struct FzipArgs {
  const __bf16* F; long ldF;        // grid features [M, >= 64] (columns 40..63 zero)
  const __bf16* D; long ldD;        // direction encoding [M, >= 16] (columns 9..15 zero)
  const char* wstream; const float* bias;
  float* raw_rgb; long ld_rgb;      // [M, >= 3] fp32
  float* raw_d; long ld_d;          // [M, >= 1] fp32
  __bf16* X32; long ld_x;           // optional [M, >= 32]: the first 32 channels of x in the compute dtype (the semantic head reads x[:, 1:1+C])
  long M; int tiles, n_chunks, n_blocks;
  // training forward (STORE): the four layer outputs the backward needs -- H1 [M, >= 64], x, h (lin_second_stage_0) and H3 (lin_second_stage_1)
  // [M, >= 256] each, row strides act_ld -- and the ReLU bit masks of H1 (one 64-column group), h and H3 (layout of ACT_RELU_BITS in gemm.hip)
  __bf16* act[4]; long act_ld[4]; unsigned* bits[3];
};

template <int F, int B, int J, bool STORE, typename C>
__device__ __forceinline__ void fzip_last_block(C& c, const bf16x8 (&h2)[16], const bf16x8 (&x)[16], const bf16x8 (&dv)[1], f32x16& rgb, const StoreTo& st, bf16x8 (&lh)[2]) {
  f32x16 acc = acc_init<B + 2 * J>(c);
  constexpr int F0 = F + J * 35;
  mac<F0, 16>(c, acc, h2);
  mac<F0 + 16, 16>(c, acc, x);
  mac<F0 + 32, 1>(c, acc, dv);
  to_frags<true, C::f16>(acc, lh[0], lh[1]);
  if constexpr (STORE) store_block<true, J>(st, lh[0], lh[1]);
  mac<F0 + 33, 2>(c, rgb, lh);
}
template <int F, int B, bool STORE, typename C, int... J>
__device__ __forceinline__ void fzip_last_seq(C& c, const bf16x8 (&h2)[16], const bf16x8 (&x)[16], const bf16x8 (&dv)[1], f32x16& rgb, const StoreTo& st,
                                              std::integer_sequence<int, J...>) {
  bf16x8 lh[2];
  (fzip_last_block<F, B, J, STORE>(c, h2, x, dv, rgb, st, lh), ...);
}

template <bool F16, bool STORE = false>
__global__ __launch_bounds__(64 * FM_WAVES, 2) void fzip_fwd_kernel(FzipArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  CtxT<FM_RING, F16> c;
  ctx_start(c, smem, a.wstream, a.n_chunks, a.bias, a.n_blocks, tid, wave, lane);
  for (int tile = blockIdx.x; tile < a.tiles; tile += gridDim.x) {
    long row = (long)tile * FM_TILE_ROWS + wave * 32 + (lane & 31);
    const bool row_ok = row < a.M;
    row = row_ok ? row : a.M - 1;
    bf16x8 f[4], dv[1], h1[4], x[16], h2[16];
    load_rows<4>(a.F, a.ldF, row, half, f);
    load_rows<1>(a.D, a.ldD, row, half, dv);
    constexpr int F1 = 2 * 4, FD = F1 + 8 * 4, F2 = FD + 4, F3 = F2 + 8 * 17;
    static_assert(F3 + 8 * 35 == FZIP_FRAGS, "zipnerf MLP: fragment count");
    auto to = [&](int i, int ncg) { return StoreTo{a.act[i], a.act_ld[i], i == 0 ? a.bits[0] : (i >= 2 ? a.bits[i - 1] : nullptr), (long)tile * FM_TILE_ROWS + wave * 32, a.M,
                                                   smem + FM_RING * FM_SLOT + FM_BIAS_MAX * 128 + wave * 4096, lane, ncg}; };
    dense<0, 0, 4, 2, true, STORE, STORE>(c, f, h1, to(0, 1));    // density_layer.0 (+ ReLU; training: with its bit mask, one column group)
    dense<F1, 2, 4, 8, false, STORE>(c, h1, x, to(1, 4));  // density_layer.2: x (bottleneck, no activation)
    if (a.X32 != nullptr && row_ok) {
      // block 0 of x as the lanes hold it: fragment 0 / 1 = outputs 0..15 / 16..31, lane half h owns {4 h .. 4 h + 3, 8 + 4 h .. 8 + 4 h + 3} of each
      typedef unsigned fz_u32x2 __attribute__((ext_vector_type(2)));
      __bf16* xr = a.X32 + row * a.ld_x + 4 * half;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const fm_u32x4 v = __builtin_bit_cast(fm_u32x4, x[s]);
        *(fz_u32x2*)(xr + 16 * s) = fz_u32x2{v[0], v[1]};
        *(fz_u32x2*)(xr + 16 * s + 8) = fz_u32x2{v[2], v[3]};
      }
    }
    f32x16 dh = acc_init<10>(c);                           // ... and its output 0 once more, unrounded: the raw density
    mac<FD, 4>(c, dh, h1);
    dense2<F2, 11, 16, 1, 8, true, STORE>(c, x, dv, h2, to(2, 4));   // lin_second_stage_0 on cat([x, dir_enc]) (+ ReLU)
    f32x16 rgb = acc_init<20>(c);                          // rgb_layer: its bias sits in the first of the interleaved pieces
    fzip_last_seq<F3, 19, STORE>(c, h2, x, dv, rgb, to(3, 4), std::make_integer_sequence<int, 8>{});   // lin_second_stage_1 on cat([h, x, dir_enc]) (+ ReLU) -> rgb
    // the pass is padded to whole chunks (464 fragments): take the four padding fragments out of the queue so that the next tile starts on
    // a chunk boundary like every other network's
    (void)next_frag<FZIP_FRAGS>(c); (void)next_frag<FZIP_FRAGS + 1>(c); (void)next_frag<FZIP_FRAGS + 2>(c); (void)next_frag<FZIP_FRAGS + 3>(c);
    static_assert((FZIP_FRAGS + 4) % FM_CHUNK == 0, "padding of the zipnerf pass");
    if (row_ok && half == 0) {
      a.raw_d[row * a.ld_d] = dh[0];
      float* o = a.raw_rgb + row * a.ld_rgb;
      o[0] = rgb[0]; o[1] = rgb[1]; o[2] = rgb[2];
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// =================================================================================================================================
// Colour head of the live mip path's NeRF MLP (s-nerf/model/models.py:283-296): cat([bottleneck, view encoding]) (1024 + 27) ->
// 3 x [Linear 128 + ReLU] -> Linear 3.  As four GEMM launches forward and eight backward these 128-wide layers are HBM- or
// epilogue-bound (round 2: 0.48 ms forward, 0.81 ms data gradients per 524 288 rows).  Fused:
//   * forward (fcolour_fwd_kernel): the first layer runs K-MAJOR -- four accumulators (128 outputs) live, the [bottleneck | view
//     encoding] row streamed ONCE from HBM as B fragments (16 bytes per lane and k-step, fetched four k-steps = one 128-byte line per
//     row at a time, two to three lines ahead), its weights streamed in (k-step, block) order; the two 128 x 128 layers and the rgb
//     head continue from the accumulator registers like every other fused layer.  Training stores the three hidden activations and
//     their ReLU bit masks (standard block layout, 2 column groups).
//   * backward (fcolour_bwd_kernel): d raw_rgb -> dC2 -> dC1 -> dC0 -> d bottleneck, the "weights x activations" chain on the
//     TRANSPOSED weights, every ReLU mask applied from the bit masks (sign-extended 1-bit fields ANDed onto the fp32 accumulators),
//     the four bias gradients reduced over the 32 rows of a wave by a register butterfly (DPP mirrors / quad permutes inside the
//     16-lane rows, one ds_swizzle across them: 47 instructions per 32 columns) and carried in one register per block across the tiles of a
//     workgroup; dC2, dC1, dC0 (the weight-gradient GEMMs read them) and d bottleneck leave through the transposition slabs.
// Bound: HBM (forward reads the 2176-byte rows once: 1.14 GB per 524 288 rows; backward writes 1.07 GB + 0.4 GB).
#define FC_NK0 66                                   // k-steps of cond_layers.0: 1024 bottleneck columns + 32 (27 view-encoding columns + zeros)
#define FC_FWD_FRAGS (4 * FC_NK0 + 32 + 32 + 8)     // 336 = 21 chunks
#define FC_FWD_BLOCKS 13                            // 4 + 4 + 4 + 1 bias blocks
#define FC_QD 12                                    // input fragments in flight (three 128-byte lines per row)
#define FC_QD_ALT 20                                // (variant bit 0 of snerf_fcolour_fwd: five lines)
#define FC_BWD_FRAGS 336                            // 4 (rgb^T) + 32 + 32 + 256 (cond_layers.0^T, bottleneck columns) + 12 padding
#define FC_BWD_COLS 1408                            // 3 x 128 + 1024 bias-gradient columns

struct ColourFwdArgs {
  const __bf16* CB; long ldCB;        // [M, >= 1056] = [bottleneck 1024 | view encoding 27 | zeros]
  const char* wstream; const float* bias;
  float* raw_rgb;                     // [M, 3]
  __bf16* act[3]; long act_ld[3];     // training: outputs of cond_layers.0 .. .2 ([M, >= 128] bf16)
  unsigned* bits[3];                  // ... and their ReLU bit masks
  long M;
  int tiles, n_chunks, n_blocks;
};

template <int F, int NB, int S, int... J, typename C>
__device__ __forceinline__ void kmajor_mfma(C& c, f32x16 (&acc)[NB], const bf16x8& in, std::integer_sequence<int, J...>) {
  ((acc[J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(next_frag<F + S * NB + J>(c), in, acc[J], 0, 0, 0)), ...);
}
// The input fragments are fetched by inline-asm loads the compiler does not track, and waited for with counted s_waitcnt: with the
// weight stream's `global_load_lds` pending, hipcc answers any vector-memory dependency of its own with s_waitcnt vmcnt(0) (an LDS-DMA is
// a FLAT operation that touches both memories; on gfx9 a pending one forces every wait to zero) -- measured on the first version of this
// kernel: one full drain of the input read-ahead AND of the weight stream every third line.
template <int NK, int QD, int K>
__device__ __forceinline__ void kmajor_fetch(bf16x8 (&q)[QD], const __bf16* src) {
  if constexpr (K < NK) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(q[K % QD]) : "v"(src), "n"(32 * K) : "memory");
}
__device__ __forceinline__ constexpr int kmajor_cnt(int a, int b, int nk) { return (b < nk ? b : nk) - (a < nk ? a : nk); }
// vector-memory operations issued after the loads of line L (k-steps 4 L .. 4 L + 3) and before its first use at k-step 4 L, with
// A = QD / 4 - 1 lines of read-ahead: the A lines fetched after it, and two DMA pieces per chunk boundary in between (a boundary falls
// in front of every k-step S = 3 mod 4 of a 4-block k-major layer that starts on a chunk boundary: fragment 4 S + FM_LOOK is a
// multiple of FM_CHUNK; line L > A is fetched at k-step 4 (L - A), the first A + 1 lines at the start of the tile).  "At most that
// many outstanding" therefore means line L has landed (loads retire in order).
__device__ __forceinline__ constexpr int kmajor_younger(int L, int nk, int qd) {
  const int A = qd / 4 - 1;
  int n = 2 * (L < A ? L : A);
  for (int i = 1; i <= A; ++i) n += kmajor_cnt(4 * (L + i), 4 * (L + i) + 4, nk);
  return n;
}
template <int F, int NB, int NK, int QD, int S, typename C>
__device__ __forceinline__ void kmajor_one(C& c, f32x16 (&acc)[NB], bf16x8 (&q)[QD], const __bf16* src) {
  static_assert(NB == 4 && QD % 4 == 0 && QD >= 8 && FM_LOOK == 4 && FM_CHUNK == 16 && F % FM_CHUNK == 0, "the wait counts assume this geometry");
  if constexpr (S % 4 == 0) {
    if constexpr (S >= 4) {                             // the line consumed last (k-steps S - 4 .. S - 1) is free: fetch the line QD - 4 ahead
      kmajor_fetch<NK, QD, S + QD - 4>(q, src); kmajor_fetch<NK, QD, S + QD - 3>(q, src);
      kmajor_fetch<NK, QD, S + QD - 2>(q, src); kmajor_fetch<NK, QD, S + QD - 1>(q, src);
    }
    // (the four registers are operands of the wait, so that no use of them can be scheduled in front of it)
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(q[S % QD]), "+v"(q[S % QD + 1]), "+v"(q[S % QD + 2]), "+v"(q[S % QD + 3]) : "n"(kmajor_younger(S / 4, NK, QD)));
  }
  kmajor_mfma<F, NB, S>(c, acc, q[S % QD], std::make_integer_sequence<int, NB>{});
}
template <int F, int NB, int NK, int QD, typename C, int... S>
__device__ __forceinline__ void kmajor_seq(C& c, f32x16 (&acc)[NB], bf16x8 (&q)[QD], const __bf16* src, std::integer_sequence<int, S...>) {
  (kmajor_one<F, NB, NK, QD, S>(c, acc, q, src), ...);
}
template <int NK, int QD, int... K>
__device__ __forceinline__ void kmajor_prefetch(bf16x8 (&q)[QD], const __bf16* src, std::integer_sequence<int, K...>) {
  (kmajor_fetch<NK, QD, K>(q, src), ...);
}

template <bool STORE, int QD>
__global__ __launch_bounds__(64 * FM_WAVES, 2) void fcolour_fwd_kernel(ColourFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  Ctx c;
  ctx_start(c, smem, a.wstream, a.n_chunks, a.bias, a.n_blocks, tid, wave, lane);
  char* const slab = smem + FM_RING * FM_SLOT + FM_BIAS_MAX * 128 + wave * 4096;

  for (int tile = blockIdx.x; tile < a.tiles; tile += gridDim.x) {
    long row = (long)tile * FM_TILE_ROWS + wave * 32 + (lane & 31);
    const bool row_ok = row < a.M;
    row = row_ok ? row : a.M - 1;                       // tail rows: compute on a valid row, store nothing
    const __bf16* src = a.CB + row * a.ldCB + half * 8;
    bf16x8 qin[QD];
    kmajor_prefetch<FC_NK0, QD>(qin, src, std::make_integer_sequence<int, QD>{});
    f32x16 acc[4] = {acc_init<0>(c), acc_init<1>(c), acc_init<2>(c), acc_init<3>(c)};
    kmajor_seq<0, 4, FC_NK0, QD>(c, acc, qin, src, std::make_integer_sequence<int, FC_NK0>{});
    auto to = [&](int i) { return StoreTo{a.act[i], a.act_ld[i], a.bits[i], (long)tile * FM_TILE_ROWS + wave * 32, a.M, slab, lane, 2}; };
    bf16x8 p[8], q[8];
    to_frags<true>(acc[0], p[0], p[1]);
    if constexpr (STORE) store_block<true, 0>(to(0), p[0], p[1]);
    to_frags<true>(acc[1], p[2], p[3]);
    if constexpr (STORE) store_block<true, 1>(to(0), p[2], p[3]);
    to_frags<true>(acc[2], p[4], p[5]);
    if constexpr (STORE) store_block<true, 2>(to(0), p[4], p[5]);
    to_frags<true>(acc[3], p[6], p[7]);
    if constexpr (STORE) store_block<true, 3>(to(0), p[6], p[7]);
    constexpr int F1 = 4 * FC_NK0, F2 = F1 + 32, FR = F2 + 32;
    static_assert(FR + 8 == FC_FWD_FRAGS, "colour head: fragment count");
    dense<F1, 4, 8, 4, true, STORE, STORE>(c, p, q, to(1));        // cond_layers.1
    dense<F2, 8, 8, 4, true, STORE, STORE>(c, q, p, to(2));        // cond_layers.2
    f32x16 rgb = acc_init<12>(c);                                  // rgb_layer: outputs 0..2
    mac<FR, 8>(c, rgb, p);
    if (row_ok && half == 0) {
      float* o = a.raw_rgb + row * 3;
      o[0] = rgb[0]; o[1] = rgb[1]; o[2] = rgb[2];
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // drain the read-ahead before the LDS is released
}

struct ColourBwdArgs {
  const float* d_rgb;                 // [M, 3] fp32: d loss / d raw_rgb
  const char* wstream;                // transposed weights: rgb_layer^T, cond_layers.2^T, .1^T, .0^T (bottleneck columns)
  const unsigned* bits[4];            // ReLU bit masks of cond_layers.2, .1, .0 (2 column groups) and of the bottleneck (16)
  __bf16* dC[3]; long dC_ld[3];       // d pre-activation of cond_layers.2, .1, .0 ([M, >= 128] bf16), read by the weight-gradient GEMMs
  __bf16* dB; long dB_ld;             // d pre-activation of the bottleneck layer [M, >= 1024]
  float* colsum_ws;                   // [gridDim.x, FC_BWD_COLS]: per-workgroup bias-gradient partials
  long M;
  int tiles, n_chunks;
};

// sum over the 32 rows (lanes 0..31 / 32..63 separately) of a 32 x 32 accumulator block; lane L ends up with the sum of register
// r = L & 15, i.e. of output column (r & 3) + 8 (r >> 2) + 4 (L >> 5) of the block (lanes L and L ^ 16 hold the same sum).  A reduce-
// scatter butterfly inside the 16-lane rows -- every step halves the registers a lane still carries: it keeps the half its lane bit
// selects and adds the partner's copy of that half (DPP row_mirror / row_half_mirror / quad permutes: 45 instructions) -- and one
// ds_swizzle (lane ^ 16) for the two rows.  (v_permlane16_swap would take the row bit first at half the cost, but hipcc 7.2 returns
// the first result of __builtin_amdgcn_permlane16_swap for BOTH members of its result pair: tools/probes/rows_sum_probe.hip.)
#define FC_DPP_ADD(keep, send, ctrl) ((keep) + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (send)), (ctrl), 0xf, 0xf, true)))
// one reduce-scatter exchange: the lane keeps x (bit clear) or y (bit set) and adds the partner's copy of the same register
#define FC_RS(x, y, bit, ctrl) FC_DPP_ADD((bit) ? (y) : (x), (bit) ? (x) : (y), ctrl)
__device__ __forceinline__ float rows_sum(const f32x16& a, int lane) {
  const bool k3 = (lane & 8) != 0, k2 = (lane & 4) != 0, k1 = (lane & 2) != 0, k0 = (lane & 1) != 0;
  // (literal register indices: with a loop index hipcc expands every vector element access into a 16-way select chain)
  const float b0 = FC_RS(a[0], a[8], k3, 0x140), b1 = FC_RS(a[1], a[9], k3, 0x140), b2 = FC_RS(a[2], a[10], k3, 0x140), b3 = FC_RS(a[3], a[11], k3, 0x140);   // lane bit 3:
  const float b4 = FC_RS(a[4], a[12], k3, 0x140), b5 = FC_RS(a[5], a[13], k3, 0x140), b6 = FC_RS(a[6], a[14], k3, 0x140), b7 = FC_RS(a[7], a[15], k3, 0x140); // row_mirror (15 - p)
  const float c0 = FC_RS(b0, b4, k2, 0x141), c1 = FC_RS(b1, b5, k2, 0x141), c2 = FC_RS(b2, b6, k2, 0x141), c3 = FC_RS(b3, b7, k2, 0x141);   // lane bit 2: row_half_mirror (7 - p)
  const float d0 = FC_RS(c0, c2, k1, 0x4E), d1 = FC_RS(c1, c3, k1, 0x4E);                  // lane bit 1: quad_perm [2,3,0,1]
  const float e = FC_RS(d0, d1, k0, 0xB1);                                                // lane bit 0: quad_perm [1,0,3,2]
  // the other 16-lane row of this half: ds_swizzle, bit mode (and 0x1f, or 0, xor 0x10)
  return e + __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, e), 0x401F));
}

// one 32-output block of a backward layer: acc = W^T-block . in, ReLU mask from the wave's LDS copy of the layer's bit-mask blocks
// (`mk`: [64-column group][64 words]), bias-gradient partial, bf16 fragments, store through the slab (pairs of blocks)
template <int F, int NK, int J, int CS, typename C>
__device__ __forceinline__ void bwd_block(C& c, const bf16x8 (&in)[NK], const char* mk, int lane, int sh, bool row_ok,
                                          float (&cs)[FC_BWD_COLS / 64], bf16x8& lo, bf16x8& hi, const StoreTo& st) {
  // this lane's 16 mask bits: words (row & 7) * 8 + 4 (J & 1) + q, q = 0..3, of column group J >> 1; byte row >> 3, nibble lane >> 5
  const fm_u32x4 w = *(const fm_u32x4*)(mk + ((J >> 1) * 64 + (lane & 7) * 8 + 4 * (J & 1)) * 4);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  mac<F, NK>(c, acc, in);
  // (literal register indices, as in rows_sum)
#define FC_MASK1(R, NQ, E) { const float v = acc[R]; acc[R] = __builtin_bit_cast(float, __builtin_bit_cast(int, v) & __builtin_amdgcn_sbfe(NQ, E, 1)); }
#define FC_MASK4(Q) { const unsigned wq = w[Q]; const int nq = row_ok ? (int)(wq >> sh) : 0;   /* rows past the end contribute nothing */ \
                      FC_MASK1(4 * Q + 0, nq, 0) FC_MASK1(4 * Q + 1, nq, 1) FC_MASK1(4 * Q + 2, nq, 2) FC_MASK1(4 * Q + 3, nq, 3) }
  FC_MASK4(0) FC_MASK4(1) FC_MASK4(2) FC_MASK4(3)
  // lanes L and L ^ 16 hold the same sum: the 16-lane rows of a half take turns (even blocks: row 0, odd blocks: row 1), so that one
  // register carries TWO blocks' partials across the tiles (44 registers would not fit beside the fragments)
  const float rs = rows_sum(acc, lane);
  cs[CS >> 1] += (((lane >> 4) & 1) == (CS & 1)) ? rs : 0.f;
  to_frags<false>(acc, lo, hi);
  store_block<false, J>(st, lo, hi);
}
template <int F, int NK, int CS0, int NB, typename C, int... J>
__device__ __forceinline__ void bwd_layer_seq(C& c, const bf16x8 (&in)[NK], const char* mk, int lane, int sh, bool row_ok,
                                              float (&cs)[FC_BWD_COLS / 64], bf16x8 (&out)[2 * NB], const StoreTo& st, std::integer_sequence<int, J...>) {
  (bwd_block<F + J * NK, NK, J, CS0 + J>(c, in, mk, lane, sh, row_ok, cs, out[2 * J], out[2 * J + 1], st), ...);
}
template <int F, int NK, int CS0, int NB, typename C>
__device__ __forceinline__ void bwd_layer(C& c, const bf16x8 (&in)[NK], const char* mk, int lane, int sh, bool row_ok,
                                          float (&cs)[FC_BWD_COLS / 64], bf16x8 (&out)[2 * NB], const StoreTo& st) {
  bwd_layer_seq<F, NK, CS0, NB>(c, in, mk, lane, sh, row_ok, cs, out, st, std::make_integer_sequence<int, NB>{});
}
// the wide last layer: the fragments are not needed again
template <int F, int NK, int CS0, typename C, int... J>
__device__ __forceinline__ void bwd_last_seq(C& c, const bf16x8 (&in)[NK], const char* mk, int lane, int sh, bool row_ok,
                                             float (&cs)[FC_BWD_COLS / 64], const StoreTo& st, std::integer_sequence<int, J...>) {
  bf16x8 lo, hi;
  (bwd_block<F + J * NK, NK, J, CS0 + J>(c, in, mk, lane, sh, row_ok, cs, lo, hi, st), ...);
}
template <int F, int N, typename C, int... I>
__device__ __forceinline__ void skip_frags(C& c, std::integer_sequence<int, I...>) {
  ((void)next_frag<F + I>(c), ...);
}

// LDS of the backward kernel: [weight ring, FC_BWD_RING slots][8 transposition slabs][8 x FC_MASK_BYTES: per wave the bottleneck's
// bit masks of its 32 rows (4 KiB), those of cond_layers.2 / .1 / .0 (512 B each) and its 32 rows of d raw_rgb (384 B)].
// Everything a tile reads from memory arrives by LDS-DMA one tile ahead, so that the loop holds NO vector-memory load the compiler
// tracks: with the weight stream's `global_load_lds` pending, hipcc answers any such dependency with s_waitcnt vmcnt(0) -- a full
// drain that also waits for the acknowledgement of every row store just issued (first version of this kernel, one drain per pair of
// blocks: 620-675 us per 524 288 rows).
#define FC_BWD_RING 5
#define FC_MASK_BYTES 6144
#define FC_BWD_LDS (FC_BWD_RING * FM_SLOT + FM_WAVES * 4096 + FM_WAVES * FC_MASK_BYTES)

__global__ __launch_bounds__(64 * FM_WAVES, 2) void fcolour_bwd_kernel(ColourBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  char* const slab = smem + FC_BWD_RING * FM_SLOT + wave * 4096;
  char* const mk = smem + FC_BWD_RING * FM_SLOT + FM_WAVES * 4096 + wave * FC_MASK_BYTES;
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)a.d_rgb, 0, (int)(a.M * 12), 0x00020000);   // rows >= M read as zeros
  // the bottleneck's masks of the wave's 32 rows: 16 column groups x 256 B, contiguous
  auto dma_wide = [&](long row0, int lane) __attribute__((always_inline)) {
    const char* g = (const char*)(a.bits[3] + (row0 >> 5) * (16 * 64)) + lane * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) __builtin_amdgcn_global_load_lds((gbl_ptr_t)(g + i * 1024), (lds_ptr_t)(mk + i * 1024), 16, 0, 0);
  };
  // the three narrow layers' masks (2 column groups = 512 B each: lanes 0..31) and d raw_rgb (32 rows x 12 B: lanes 0..23)
  auto dma_narrow = [&](long row0, int lane) __attribute__((always_inline)) {
    if (lane < 32) {
#pragma unroll
      for (int l = 0; l < 3; ++l)
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)((const char*)(a.bits[l] + (row0 >> 5) * (2 * 64)) + lane * 16), (lds_ptr_t)(mk + 4096 + l * 512), 16, 0, 0);
    }
    if (lane < 24) __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, (lds_ptr_t)(mk + 5632), 16, lane * 16, (int)(row0 * 12), 0, 0);
  };
  {
    const long row0 = (long)blockIdx.x * FM_TILE_ROWS + wave * 32;    // (the launch has gridDim.x <= tiles)
    dma_wide(row0, lane);
    dma_narrow(row0, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  CtxT<FC_BWD_RING> c;
  ctx_start(c, smem, a.wstream, a.n_chunks, nullptr, 0, tid, wave, lane);
  const int sh = 8 * ((lane & 31) >> 3) + 4 * half;     // this lane's nibble inside a mask word
  float cs[FC_BWD_COLS / 64];                            // bias-gradient partials: register i = blocks 2 i (lanes with bit 4 clear) / 2 i + 1
#pragma unroll
  for (int i = 0; i < FC_BWD_COLS / 64; ++i) cs[i] = 0.f;

  for (int tile = blockIdx.x; tile < a.tiles; tile += gridDim.x) {
    const long row0 = (long)tile * FM_TILE_ROWS + wave * 32;
    const bool row_ok = row0 + (lane & 31) < a.M;
    const bool more = tile + (int)gridDim.x < a.tiles;
    // every global address of the tile is derived from a loop-variant spelling of the lane id: hoisted out of the loop, the 64-bit
    // store / DMA addresses are spilled (the kernel sits at the 256-register limit) and every reload is a scratch load, i.e. one more
    // s_waitcnt vmcnt(0) drain per use
    int zero;
    asm volatile("s_lshr_b32 %0, %1, 31" : "=s"(zero) : "s"(tile));
    const int ln = lane | zero;
    const long next0 = row0 + (long)gridDim.x * FM_TILE_ROWS;
    // d raw_rgb as the B fragment of one k-step: lane half 0 supplies reduction indices 0..7 = (r, g, b, 0, ...), half 1 zeros
    bf16x8 g[1];
    {
      const float* dp = (const float*)(mk + 5632) + (lane & 31) * 3;
      const float x0 = dp[0], x1 = dp[1], x2 = dp[2];
      typedef __attribute__((ext_vector_type(8))) float f32x8;
      const f32x8 v = {half == 0 ? x0 : 0.f, half == 0 ? x1 : 0.f, half == 0 ? x2 : 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      g[0] = __builtin_convertvector(v, bf16x8);
    }
    auto to = [&](int i) { return StoreTo{a.dC[i], a.dC_ld[i], nullptr, row0, a.M, slab, ln, 2}; };
    bf16x8 p[8], q[8];
    bwd_layer<0, 1, 0, 4>(c, g, mk + 4096, lane, sh, row_ok, cs, p, to(0));           // dC2 = mask2 . (W_rgb^T d raw_rgb)
    bwd_layer<4, 8, 4, 4>(c, p, mk + 4608, lane, sh, row_ok, cs, q, to(1));           // dC1 = mask1 . (W_c2^T dC2)
    bwd_layer<36, 8, 8, 4>(c, q, mk + 5120, lane, sh, row_ok, cs, p, to(2));          // dC0 = mask0 . (W_c1^T dC1)
    // The bottleneck masks of THIS tile were DMA'd at the end of the previous one; since then the four chunk boundaries of fragments
    // 0..67 issued eight younger DMA operations, so "at most eight outstanding" means they have landed (loads retire in order;
    // stores in flight only make the wait stricter).  The narrow masks and d raw_rgb of this tile are consumed: fetch the next tile's.
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    if (more) dma_narrow(next0, ln);
    const StoreTo tb{a.dB, a.dB_ld, nullptr, row0, a.M, slab, ln, 16};
    bwd_last_seq<68, 8, 12>(c, p, mk, lane, sh, row_ok, cs, tb, std::make_integer_sequence<int, 32>{});   // d bottleneck
    skip_frags<324, 12>(c, std::make_integer_sequence<int, 12>{});                           // the stream's padding to whole chunks
    static_assert(324 + 12 == FC_BWD_FRAGS && FM_CHUNK == 16 && FM_LOOK == 4, "colour head backward: fragment count / boundary positions");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of the bottleneck masks have returned: refill
    if (more) dma_wide(next0, ln);
  }
  // the workgroup's eight waves fold their partials in LDS (wave order: fixed) and leave ONE row of the workspace
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's read-ahead of the stream has landed: the ring can be reused
  __syncthreads();
  float* red = (float*)smem;                            // [FM_WAVES][FC_BWD_COLS]
  {
    const int r = lane & 15;
    float* dst = red + wave * FC_BWD_COLS + 32 * ((lane >> 4) & 1) + (r & 3) + 8 * (r >> 2) + 4 * half;
#pragma unroll
    for (int i = 0; i < FC_BWD_COLS / 64; ++i) dst[64 * i] = cs[i];
  }
  __syncthreads();
  for (int col = tid; col < FC_BWD_COLS; col += 64 * FM_WAVES) {
    float t = red[col];
#pragma unroll
    for (int w = 1; w < FM_WAVES; ++w) t += red[w * FC_BWD_COLS + col];
    a.colsum_ws[(long)blockIdx.x * FC_BWD_COLS + col] = t;
  }
}

// =================================================================================================================================
// Fused DATA-GRADIENT CHAINS of the 256-wide networks (classic NeRF 8 x 256 of path B, proposal MLP 4 x 256 of path A): d raw -> ...
// -> d pre-activation of every layer in ONE launch, the mirror image of fmlp_kernel -- "weights x activations" on the transposed
// weights, the gradient of a layer being the next layer's B fragments as it leaves the accumulators.  Every layer's gradient is stored
// (bf16, through the transposition slabs) for the weight-gradient GEMMs, its ReLU mask comes from the bit masks the training forward
// wrote, its bias gradient is reduced over the wave's 32 rows by the register butterfly (rows_sum) and added to a workgroup-wide LDS
// table (ds_add_f32: the order of the eight waves' additions is not fixed: the deterministic mode keeps the per-layer kernels).
// As separate GEMM launches these layers ran at 0.18 of the MFMA peak (256 x 256 tiles, one activation row read per 256 outputs).
// Everything a tile reads arrives by LDS-DMA (no vector-memory load the compiler tracks: see fcolour_bwd_kernel): the bit masks of
// the masked steps alternate between two 1 KiB buffers per wave, each fetched one step ahead; d raw and the first mask of the NEXT
// tile are fetched while the current one runs.
#define FCH_RING 5
#define FCH_WAVE_BYTES 3072                      // per wave: two 1 KiB mask buffers + 1 KiB (narrow mask 512 B, d raw 512 B)
struct ChainArgs {
  const float* d_raw; int d_cols;                // [M, d_cols] fp32: classic (rgb, alpha) = 4, proposal = 1
  const char* wstream;
  const unsigned* bits[9];                       // classic: pts_linears.0..7, views_linears.0; proposal: layers.0..3
  __bf16* dz[10]; long dz_ld[10];                // the steps' outputs, in chain order
  float* colsum_ws;                              // [gridDim.x, n_cols]
  long M;
  int tiles, n_chunks, n_cols;
};

// Bias gradients: a register butterfly over the wave's 32 rows and one LDS atomic per column into the workgroup's table (chain_block).
// (The alternative -- column sums by two transposing LDS reads + two MFMAs per block on the slab -- has 35 % fewer instructions and is 10 %
// slower: withdrawn, tools/probes/fmlp_experiments.hip, profiles/r3_n_fchain_probe_mfma_colsum.txt.)
struct ColsumCtx {
  unsigned tab0;              // this lane's column of block 0 in the table (lanes with bit 4 clear: r = lane & 15)
};
__device__ __forceinline__ void colsum_start(ColsumCtx& k, const char* slab, const float* cs, int lane) {
  (void)slab;
  k.tab0 = (unsigned)(size_t)(cs + ((lane & 15) & 3) + 8 * ((lane & 15) >> 2) + 4 * (lane >> 5));
}

// one 32-output block of a chain step: acc = W^T-block . [in | extra], optional ReLU mask from `mk`, bf16 fragments, store through
// the slab, bias-gradient partial (pairs of blocks); G = the block's number in the chain, LAST: the chain's last block
template <int F, int NK, bool EXTRA, bool MASKED, int J, int G, bool LAST, typename C>
__device__ __forceinline__ void chain_block(C& c, const bf16x8 (&in)[NK], const bf16x8& extra, const char* mk, int lane, int sh, bool row_ok,
                                            ColsumCtx& k, bf16x8& lo, bf16x8& hi, const StoreTo& st) {
  fm_u32x4 w = {0u, 0u, 0u, 0u};
  if constexpr (MASKED) w = *(const fm_u32x4*)(mk + ((J >> 1) * 64 + (lane & 7) * 8 + 4 * (J & 1)) * 4);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  mac<F, NK>(c, acc, in);
  if constexpr (EXTRA) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(next_frag<F + NK>(c), extra, acc, 0, 0, 0);
#define FCH_MASK1(R, NQ, E) { const float v = acc[R]; acc[R] = __builtin_bit_cast(float, __builtin_bit_cast(int, v) & __builtin_amdgcn_sbfe(NQ, E, 1)); }
#define FCH_MASK4(Q) { const unsigned wq = w[Q]; const int nq = MASKED ? (row_ok ? (int)(wq >> sh) : 0) : (row_ok ? -1 : 0); \
                       FCH_MASK1(4 * Q + 0, nq, MASKED ? 0 : 0) FCH_MASK1(4 * Q + 1, nq, MASKED ? 1 : 0) FCH_MASK1(4 * Q + 2, nq, MASKED ? 2 : 0) FCH_MASK1(4 * Q + 3, nq, MASKED ? 3 : 0) }
  FCH_MASK4(0) FCH_MASK4(1) FCH_MASK4(2) FCH_MASK4(3)
  {
    // register butterfly over the wave's 32 rows, then one LDS atomic per column into the workgroup's table -- by hand: an LDS atomic
    // the compiler emits itself waits for vmcnt(0) while an LDS-DMA is in flight (it may alias) and the weight stream always has some
    // in flight; the table is disjoint from every DMA target.  Completion: lgkmcnt(0) at the tile's end
    const float rs = rows_sum(acc, lane);
    if ((lane & 16) == 0) {
      asm volatile("ds_add_f32 %0, %1 offset:%2" ::"v"(k.tab0), "v"(rs), "n"(32 * G * 4) : "memory");
    }
  }
  to_frags<false>(acc, lo, hi);
  store_block<false, J>(st, lo, hi);
}
template <int F, int NK, bool EXTRA, bool MASKED, int NB, int G0, bool LAST, typename C, int... J>
__device__ __forceinline__ void chain_step_seq(C& c, const bf16x8 (&in)[NK], const bf16x8& extra, const char* mk, int lane, int sh, bool row_ok,
                                               ColsumCtx& k, bf16x8 (&out)[2 * NB], const StoreTo& st, std::integer_sequence<int, J...>) {
  (chain_block<F + J * (NK + (EXTRA ? 1 : 0)), NK, EXTRA, MASKED, J, G0 + J, (LAST && J == NB - 1)>(c, in, extra, mk, lane, sh, row_ok, k, out[2 * J], out[2 * J + 1], st), ...);
}
// G0 = number of the step's first block in the chain (its bias gradients: columns 32 G0 .. of the table)
template <int F, int NK, bool EXTRA, bool MASKED, int NB, int G0, bool LAST = false, typename C>
__device__ __forceinline__ void chain_step(C& c, const bf16x8 (&in)[NK], const bf16x8& extra, const char* mk, int lane, int sh, bool row_ok,
                                           ColsumCtx& k, bf16x8 (&out)[2 * NB], const StoreTo& st) {
  chain_step_seq<F, NK, EXTRA, MASKED, NB, G0, LAST>(c, in, extra, mk, lane, sh, row_ok, k, out, st, std::make_integer_sequence<int, NB>{});
}

#define FCH_CLASSIC_FRAGS 1104                   // 4 + 64 + 8 x 17 + 7 x 128 = 1100, padded to whole chunks
#define FCH_CLASSIC_COLS (128 + 256 + 8 * 256)   // bias gradients: views_linears.0, feature_linear, pts_linears.7 .. .0
#define FCH_TAB(NCOLS) (((NCOLS) + 1023) / 1024 * 1024)      // the LDS table: whole groups of 32 blocks
#define FCH_PROPOSAL_FRAGS 400                   // 8 + 3 x 128 = 392, padded
#define FCH_PROPOSAL_COLS (4 * 256)              // layers.3 .. .0

template <int NET>
__global__ __launch_bounds__(64 * FM_WAVES, 2) void fchain_bwd_kernel(ChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  char* const slab = smem + FCH_RING * FM_SLOT + wave * 4096;
  char* const mk = smem + FCH_RING * FM_SLOT + FM_WAVES * 4096 + wave * FCH_WAVE_BYTES;      // [mask A 1 KiB][mask B 1 KiB][narrow mask 512][d raw 512]
  float* const cs = (float*)(smem + FCH_RING * FM_SLOT + FM_WAVES * 4096 + FM_WAVES * FCH_WAVE_BYTES);
  constexpr int NCOLS = NET == FMLP_CLASSIC ? FCH_CLASSIC_COLS : FCH_PROPOSAL_COLS;
  constexpr int DC = NET == FMLP_CLASSIC ? 4 : 1;             // columns of d raw
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)a.d_raw, 0, (int)(a.M * DC * 4), 0x00020000);   // rows >= M read as zeros
  // the 256-wide layer l's masks of the wave's 32 rows: 4 column groups x 256 B, contiguous; one DMA (64 lanes x 16 B)
  auto dma_mask = [&](int l, int buf, long row0, int lane) __attribute__((always_inline)) {
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)((const char*)(a.bits[l] + (row0 >> 5) * (4 * 64)) + lane * 16), (lds_ptr_t)(mk + buf * 1024), 16, 0, 0);
  };
  // d raw of the wave's 32 rows (+ classic: the 128-wide views layer's masks, 2 column groups = 512 B)
  auto dma_small = [&](long row0, int lane) __attribute__((always_inline)) {
    if (NET == FMLP_CLASSIC && lane < 32)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)((const char*)(a.bits[8] + (row0 >> 5) * (2 * 64)) + lane * 16), (lds_ptr_t)(mk + 2048), 16, 0, 0);
    if (lane < 8 * DC) __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, (lds_ptr_t)(mk + 2560), 16, lane * 16, (int)(row0 * DC * 4), 0, 0);
  };
  {
    const long row0 = (long)blockIdx.x * FM_TILE_ROWS + wave * 32;    // (the launch has gridDim.x <= tiles)
    dma_small(row0, lane);
    if (NET == FMLP_PROPOSAL) dma_mask(3, 0, row0, lane);
    for (int i = tid; i < FCH_TAB(NCOLS); i += 64 * FM_WAVES) cs[i] = 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  ColsumCtx k;
  colsum_start(k, slab, cs, lane);
  CtxT<FCH_RING> c;
  ctx_start(c, smem, a.wstream, a.n_chunks, nullptr, 0, tid, wave, lane);        // (ends with a barrier: the zeroed table is visible)
  const int sh = 8 * ((lane & 31) >> 3) + 4 * half;     // this lane's nibble inside a mask word

  for (int tile = blockIdx.x; tile < a.tiles; tile += gridDim.x) {
    const long row0 = (long)tile * FM_TILE_ROWS + wave * 32;
    const bool row_ok = row0 + (lane & 31) < a.M;
    const bool more = tile + (int)gridDim.x < a.tiles;
    const long next0 = row0 + (long)gridDim.x * FM_TILE_ROWS;
    int zero;                                            // (addresses from a loop-variant lane id: see fcolour_bwd_kernel)
    asm volatile("s_lshr_b32 %0, %1, 31" : "=s"(zero) : "s"(tile));
    const int ln = lane | zero;
    auto to = [&](int i, int ncg) { return StoreTo{a.dz[i], a.dz_ld[i], nullptr, row0, a.M, slab, ln, ncg}; };
    typedef __attribute__((ext_vector_type(8))) float f32x8;
    const bf16x8 none = {};
    bf16x8 p[16], q[16];
    // "at most eight outstanding": the mask fetched one step ago has landed (a step of >= 64 fragments issues >= 8 younger DMA
    // pieces at its chunk boundaries; stores in flight only make the wait stricter)
#define FCH_WAIT() asm volatile("s_waitcnt vmcnt(8)" ::: "memory")
    FCH_WAIT();                                          // what the previous tile fetched for this one (first tile: waited for above)
    if constexpr (NET == FMLP_CLASSIC) {
      const f32x4 dr = *(const f32x4*)(mk + 2560 + (ln & 31) * 16);
      const f32x8 v0 = {half == 0 ? dr[0] : 0.f, half == 0 ? dr[1] : 0.f, half == 0 ? dr[2] : 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const f32x8 v1 = {half == 0 ? dr[3] : 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      bf16x8 g[1] = {__builtin_convertvector(v0, bf16x8)};
      const bf16x8 ga = __builtin_convertvector(v1, bf16x8);
      bf16x8 hv[8];
      chain_step<0, 1, false, true, 4, 0>(c, g, none, mk + 2048, ln, sh, row_ok, k, hv, to(0, 2));                       // d views_linears.0 (masked by its output)
      dma_mask(7, 0, row0, ln);                                                                                      // -> A: pts_linears.7's masks
      chain_step<4, 8, false, false, 8, 4>(c, hv, none, nullptr, ln, sh, row_ok, k, p, to(1, 4));                 // d feature_linear (no activation)
      FCH_WAIT();
      dma_mask(6, 1, row0, ln);
      if (more) dma_small(next0, ln);                    // d raw and the views masks of the next tile (this tile's are consumed)
      chain_step<68, 16, true, true, 8, 12>(c, p, ga, mk, ln, sh, row_ok, k, q, to(2, 4));                        // d pts_linears.7 = mask . ([W_f | w_a]^T [dF; da])
      FCH_WAIT(); dma_mask(5, 0, row0, ln);
      chain_step<204, 16, false, true, 8, 20>(c, q, none, mk + 1024, ln, sh, row_ok, k, p, to(3, 4));             // .6
      FCH_WAIT(); dma_mask(4, 1, row0, ln);
      chain_step<332, 16, false, true, 8, 28>(c, p, none, mk, ln, sh, row_ok, k, q, to(4, 4));                    // .5
      FCH_WAIT(); dma_mask(3, 0, row0, ln);
      chain_step<460, 16, false, true, 8, 36>(c, q, none, mk + 1024, ln, sh, row_ok, k, p, to(5, 4));            // .4
      FCH_WAIT(); dma_mask(2, 1, row0, ln);
      chain_step<588, 16, false, true, 8, 44>(c, p, none, mk, ln, sh, row_ok, k, q, to(6, 4));                   // .3
      FCH_WAIT(); dma_mask(1, 0, row0, ln);
      chain_step<716, 16, false, true, 8, 52>(c, q, none, mk + 1024, ln, sh, row_ok, k, p, to(7, 4));            // .2
      FCH_WAIT(); dma_mask(0, 1, row0, ln);
      chain_step<844, 16, false, true, 8, 60>(c, p, none, mk, ln, sh, row_ok, k, q, to(8, 4));                   // .1
      FCH_WAIT();
      chain_step<972, 16, false, true, 8, 68, true>(c, q, none, mk + 1024, ln, sh, row_ok, k, p, to(9, 4));            // .0
      skip_frags<1100, 4>(c, std::make_integer_sequence<int, 4>{});
      static_assert(1100 + 4 == FCH_CLASSIC_FRAGS, "classic chain: fragment count");
    } else {
      const float dr = *(const float*)(mk + 2560 + (ln & 31) * 4);
      const f32x8 v0 = {half == 0 ? dr : 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      bf16x8 g[1] = {__builtin_convertvector(v0, bf16x8)};
      dma_mask(2, 1, row0, ln);                          // -> B: layers.2's masks (A holds layers.3's, fetched during the previous tile)
      chain_step<0, 1, false, true, 8, 0>(c, g, none, mk, ln, sh, row_ok, k, p, to(0, 4));                              // d layers.3
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the first step is 8 fragments long: too short for the counted wait)
      dma_mask(1, 0, row0, ln);
      if (more) dma_small(next0, ln);
      chain_step<8, 16, false, true, 8, 8>(c, p, none, mk + 1024, ln, sh, row_ok, k, q, to(1, 4));                // .2
      FCH_WAIT(); dma_mask(0, 1, row0, ln);
      chain_step<136, 16, false, true, 8, 16>(c, q, none, mk, ln, sh, row_ok, k, p, to(2, 4));                     // .1
      FCH_WAIT();
      if (more) dma_mask(3, 0, next0, ln);               // A is free: the next tile's first masks
      chain_step<264, 16, false, true, 8, 24, true>(c, p, none, mk + 1024, ln, sh, row_ok, k, q, to(3, 4));              // .0
      skip_frags<392, 8>(c, std::make_integer_sequence<int, 8>{});
      static_assert(392 + 8 == FCH_PROPOSAL_FRAGS, "proposal chain: fragment count");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = tid; i < NCOLS; i += 64 * FM_WAVES) a.colsum_ws[(long)blockIdx.x * NCOLS + i] = cs[i];
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Data-gradient chain of the zipnerf NeRF MLP (round 4; the backward of fzip_fwd_kernel<.., STORE>): d raw_rgb [M,3], d (raw density |
// semantic logits) [M, <= 32] ->
//   dH3 = mask(H3) . rgb_layer^T d rgb                               -> dz[0]  (the weight gradient of lin_second_stage_1 reads it)
//   dh  = mask(h)  . lin_second_stage_1[:, :256]^T dH3               -> dz[1]  (... of lin_second_stage_0)
//   dx  = lin_second_stage_0[:, :256]^T dh + lin_second_stage_1[:, 256:512]^T dH3 + [d density | d logits | 0]   -> dz[2]
//   dH1 = mask(H1) . density_layer.2^T dx                            -> dz[3]
//   dF  = density_layer.0^T dH1                                      -> dz[4]  (the gradient of the grid features: the table gradient's input)
// on the blocks of fchain_bwd_kernel (transposed weights streamed through the ring, masks from the forward's bit masks, bias gradients
// by the register butterfly into a workgroup table).  dx is never held: as soon as a block of 32 of its channels is rounded and on its way
// out, the two k-steps of density_layer.2^T that consume it are multiplied into the two dH1 accumulators (the host interleaves those
// fragments), so a lane keeps dh, dH3 and the head gradients (34 fragments) while the 272 MFMAs of the dx step run.
// Mask images per wave and tile: H3 and h 1 KiB each, H1 256 B, by LDS-DMA ONE TILE AHEAD (H3 / h right after the step that consumed them --
// 19 chunks of weight stream follow before the next tile reads them -- H1 after its step; the first tile's with a full wait).
// 448 fragments (28 chunks); bias-gradient table 896 columns (256 + 256 + 256 + 64, + 64 unused of the last step).
#define FZCH_FRAGS 448
#define FZCH_COLS 896
#define FZCH_WAVE_BYTES 2560
#define FZCH_LDS (FCH_RING * FM_SLOT + FM_WAVES * 4096 + FM_WAVES * FZCH_WAVE_BYTES + FCH_TAB(FZCH_COLS) * 4)
struct ZipChainArgs {
  const float* d_rgb; long ld_rgb;               // [M, >= 3] fp32
  const float* d_den; long ld_den; int den_cols;  // [M, den_cols <= 32] fp32: d raw density, then the semantic logits' gradients
  const char* wstream;
  const unsigned* bits[3];                       // ReLU bit masks of H1 (64 wide), h, H3 (256 wide)
  __bf16* dz[5]; long dz_ld[5];
  float* colsum_ws;                              // [gridDim.x, FZCH_COLS]
  long M;
  int tiles, n_chunks;
};

// what follows the MFMAs of a chain block: ReLU mask from the bit-mask words, bias-gradient partial, rounding, store through the slab
template <bool MASKED, int J, int G, typename C>
__device__ __forceinline__ void zc_finish(C& c, f32x16& acc, const char* mk, int lane, int sh, bool row_ok, ColsumCtx& k, bf16x8& lo, bf16x8& hi, const StoreTo& st) {
  (void)c;
  fm_u32x4 w = {0u, 0u, 0u, 0u};
  if constexpr (MASKED) w = *(const fm_u32x4*)(mk + ((J >> 1) * 64 + (lane & 7) * 8 + 4 * (J & 1)) * 4);
  FCH_MASK4(0) FCH_MASK4(1) FCH_MASK4(2) FCH_MASK4(3)
  const float rs = rows_sum(acc, lane);
  if ((lane & 16) == 0) {
    asm volatile("ds_add_f32 %0, %1 offset:%2" ::"v"(k.tab0), "v"(rs), "n"(32 * G * 4) : "memory");
  }
  to_frags<false, C::f16>(acc, lo, hi);
  store_block<false, J>(st, lo, hi);
}
template <int F, int NK, bool MASKED, int J, int G, typename C>
__device__ __forceinline__ void zc_block(C& c, const bf16x8 (&in)[NK], const char* mk, int lane, int sh, bool row_ok, ColsumCtx& k, bf16x8& lo, bf16x8& hi,
                                         const StoreTo& st) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  mac<F, NK>(c, acc, in);
  zc_finish<MASKED, J, G>(c, acc, mk, lane, sh, row_ok, k, lo, hi, st);
}
template <int F, int NK, bool MASKED, int NB, int G0, typename C, int... J>
__device__ __forceinline__ void zc_step(C& c, const bf16x8 (&in)[NK], const char* mk, int lane, int sh, bool row_ok, ColsumCtx& k, bf16x8 (&out)[2 * NB],
                                        const StoreTo& st, std::integer_sequence<int, J...>) {
  (zc_block<F + J * NK, NK, MASKED, J, G0 + J>(c, in, mk, lane, sh, row_ok, k, out[2 * J], out[2 * J + 1], st), ...);
}
// the dx step, block J: 34 k-steps over [dh | dH3 | head gradients], then its two fragments into the two dH1 accumulators
template <int F, int J, int G0, typename C>
__device__ __forceinline__ void zc_dx_block(C& c, const bf16x8 (&in)[34], int lane, int sh, bool row_ok, ColsumCtx& k, f32x16& h0, f32x16& h1, const StoreTo& st) {
  constexpr int F0 = F + J * 38;
  bf16x8 lh[2];
  zc_block<F0, 34, false, J, G0 + J>(c, in, nullptr, lane, sh, row_ok, k, lh[0], lh[1], st);
  mac<F0 + 34, 2>(c, h0, lh);
  mac<F0 + 36, 2>(c, h1, lh);
}
template <int F, int G0, typename C, int... J>
__device__ __forceinline__ void zc_dx_step(C& c, const bf16x8 (&in)[34], int lane, int sh, bool row_ok, ColsumCtx& k, f32x16& h0, f32x16& h1, const StoreTo& st,
                                           std::integer_sequence<int, J...>) {
  (zc_dx_block<F, J, G0>(c, in, lane, sh, row_ok, k, h0, h1, st), ...);
}

template <bool F16>
__global__ __launch_bounds__(64 * FM_WAVES, 2) void fzip_chain_bwd_kernel(ZipChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  char* const slab = smem + FCH_RING * FM_SLOT + wave * 4096;
  char* const mk = smem + FCH_RING * FM_SLOT + FM_WAVES * 4096 + wave * FZCH_WAVE_BYTES;      // [H3 masks 1 KiB][h masks 1 KiB][H1 masks 256 B]
  float* const cs = (float*)(smem + FCH_RING * FM_SLOT + FM_WAVES * 4096 + FM_WAVES * FZCH_WAVE_BYTES);
  auto dma_wide = [&](int l, int buf, long row0, int ln) __attribute__((always_inline)) {           // a 256-wide layer's masks of the wave's 32 rows
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)((const char*)(a.bits[l] + (row0 >> 5) * (4 * 64)) + ln * 16), (lds_ptr_t)(mk + buf * 1024), 16, 0, 0);
  };
  auto dma_h1 = [&](long row0, int ln) __attribute__((always_inline)) {                              // H1's (one column group: 256 B)
    if (ln < 16) __builtin_amdgcn_global_load_lds((gbl_ptr_t)((const char*)(a.bits[0] + (row0 >> 5) * 64) + ln * 16), (lds_ptr_t)(mk + 2048), 16, 0, 0);
  };
  {
    const long row0 = (long)blockIdx.x * FM_TILE_ROWS + wave * 32;    // (the launch has gridDim.x <= tiles)
    dma_wide(2, 0, row0, lane); dma_wide(1, 1, row0, lane); dma_h1(row0, lane);
    for (int i = tid; i < FCH_TAB(FZCH_COLS); i += 64 * FM_WAVES) cs[i] = 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  ColsumCtx k;
  colsum_start(k, slab, cs, lane);
  CtxT<FCH_RING, F16> c;
  ctx_start(c, smem, a.wstream, a.n_chunks, nullptr, 0, tid, wave, lane);        // (ends with a barrier: the zeroed table is visible)
  const int sh = 8 * ((lane & 31) >> 3) + 4 * half;     // this lane's nibble inside a mask word

  for (int tile = blockIdx.x; tile < a.tiles; tile += gridDim.x) {
    const long row0 = (long)tile * FM_TILE_ROWS + wave * 32;
    const bool row_ok = row0 + (lane & 31) < a.M;
    const bool more = tile + (int)gridDim.x < a.tiles;
    const long next0 = row0 + (long)gridDim.x * FM_TILE_ROWS;
    int zero;                                            // (addresses from a loop-variant lane id: see fcolour_bwd_kernel)
    asm volatile("s_lshr_b32 %0, %1, 31" : "=s"(zero) : "s"(tile));
    const int ln = lane | zero;
    auto to = [&](int i, int ncg) { return StoreTo{a.dz[i], a.dz_ld[i], nullptr, row0, a.M, slab, ln, ncg}; };
    typedef __attribute__((ext_vector_type(8))) float f32x8;
    // the head gradients of this lane's row, as B fragments in natural order: lane half h supplies reduction indices 8 h + e of a k-step
    const long row = row_ok ? row0 + (ln & 31) : 0;
    f32x8 vg = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, v0 = vg, v1 = vg;
    if (row_ok) {
      if (half == 0) { const float* pr = a.d_rgb + row * a.ld_rgb; vg[0] = pr[0]; vg[1] = pr[1]; vg[2] = pr[2]; }
      const float* pd = a.d_den + row * a.ld_den;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c0 = 8 * half + e, c1 = 16 + 8 * half + e;
        v0[e] = c0 < a.den_cols ? pd[c0] : 0.f;
        v1[e] = c1 < a.den_cols ? pd[c1] : 0.f;
      }
    }
    bf16x8 p[16], q[16], in[34];
    bf16x8 g[1];
    if constexpr (F16) {
      typedef _Float16 fz_f16x8 __attribute__((ext_vector_type(8)));
      g[0] = __builtin_bit_cast(bf16x8, __builtin_convertvector(vg, fz_f16x8));
      in[32] = __builtin_bit_cast(bf16x8, __builtin_convertvector(v0, fz_f16x8));
      in[33] = __builtin_bit_cast(bf16x8, __builtin_convertvector(v1, fz_f16x8));
    } else {
      g[0] = __builtin_convertvector(vg, bf16x8);
      in[32] = __builtin_convertvector(v0, bf16x8);
      in[33] = __builtin_convertvector(v1, bf16x8);
    }
    FCH_WAIT();                                          // the masks fetched during the previous tile have landed (first tile: waited for above)
    zc_step<0, 1, true, 8, 0>(c, g, mk, ln, sh, row_ok, k, p, to(0, 4), std::make_integer_sequence<int, 8>{});                 // dH3
    zc_step<8, 16, true, 8, 8>(c, p, mk + 1024, ln, sh, row_ok, k, q, to(1, 4), std::make_integer_sequence<int, 8>{});         // dh
    if (more) { dma_wide(2, 0, next0, ln); dma_wide(1, 1, next0, ln); }      // both buffers are consumed: the next tile's H3 / h masks
#pragma unroll
    for (int i = 0; i < 16; ++i) { in[i] = q[i]; in[16 + i] = p[i]; }
    f32x16 h0, h1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { h0[r] = 0.f; h1[r] = 0.f; }
    zc_dx_step<136, 16>(c, in, ln, sh, row_ok, k, h0, h1, to(2, 4), std::make_integer_sequence<int, 8>{});                     // dx (+ dH1's MFMAs)
    bf16x8 dh1[4];
    zc_finish<true, 0, 24>(c, h0, mk + 2048, ln, sh, row_ok, k, dh1[0], dh1[1], to(3, 1));                                   // dH1
    zc_finish<true, 1, 25>(c, h1, mk + 2048, ln, sh, row_ok, k, dh1[2], dh1[3], to(3, 1));
    if (more) dma_h1(next0, ln);
    bf16x8 df[4];
    zc_step<440, 4, false, 2, 26>(c, dh1, nullptr, ln, sh, row_ok, k, df, to(4, 1), std::make_integer_sequence<int, 2>{});     // dF
    static_assert(440 + 8 == FZCH_FRAGS && FZCH_FRAGS % FM_CHUNK == 0, "zipnerf chain: fragment count");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = tid; i < FZCH_COLS; i += 64 * FM_WAVES) a.colsum_ws[(long)blockIdx.x * FZCH_COLS + i] = cs[i];
}

// out[c] += sum over the workgroups' partial rows (fixed order)
struct ChainFoldTab { float* dst[10]; int first[10]; };
// (64 columns x 4 row groups per workgroup, four rows of a group in flight: a thread walking all 256 partial rows of its column alone
// took 65 us of load latency per launch)
__global__ __launch_bounds__(256) void fchain_colsum_fold_kernel(const float* __restrict__ ws, int rows, int ncols, ChainFoldTab tab) {
  __shared__ float part[4][64];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cl;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (col < ncols) {
    int r = rg;
    for (; r + 12 < rows; r += 16) {
      s0 += ws[(long)r * ncols + col]; s1 += ws[(long)(r + 4) * ncols + col];
      s2 += ws[(long)(r + 8) * ncols + col]; s3 += ws[(long)(r + 12) * ncols + col];
    }
    for (; r < rows; r += 4) s0 += ws[(long)r * ncols + col];
  }
  part[rg][cl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rg != 0 || col >= ncols) return;
  const float s = ((part[0][cl] + part[1][cl]) + part[2][cl]) + part[3][cl];
  int t = 0;
#pragma unroll
  for (int i = 1; i < 10; ++i) t += tab.first[i] <= col ? 1 : 0;
  tab.dst[t][col - tab.first[t]] += s;
}

// bias gradients += the per-workgroup partials, summed in a fixed order (bit-reproducible): 64 columns x 4 row groups per workgroup
__global__ __launch_bounds__(256) void fcolour_colsum_fold_kernel(const float* __restrict__ ws, int rows, float* g2, float* g1, float* g0, float* gb) {
  __shared__ float part[4][64];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cl;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;           // (four rows in flight)
  int r = rg;
  for (; r + 12 < rows; r += 16) {
    s0 += ws[(long)r * FC_BWD_COLS + col]; s1 += ws[(long)(r + 4) * FC_BWD_COLS + col];
    s2 += ws[(long)(r + 8) * FC_BWD_COLS + col]; s3 += ws[(long)(r + 12) * FC_BWD_COLS + col];
  }
  for (; r < rows; r += 4) s0 += ws[(long)r * FC_BWD_COLS + col];
  part[rg][cl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rg == 0) {
    const float t = ((part[0][cl] + part[1][cl]) + part[2][cl]) + part[3][cl];
    float* dst = col < 128 ? g2 + col : col < 256 ? g1 + (col - 128) : col < 384 ? g0 + (col - 256) : gb + (col - 384);
    *dst += t;
  }
}

template <int NET, bool EMBED, bool STORE = false>
static int fmlp_launch(const FmlpArgs& a, int expect_frags, int expect_blocks, long n_frags, void* stream) {
  if (a.M <= 0) return SNERF_OK;
  if (n_frags != expect_frags || a.n_blocks != expect_blocks || a.n_blocks > FM_BIAS_MAX || (n_frags % FM_CHUNK) != 0) return SNERF_ERR_ARG;
  if (a.wstream == nullptr || a.bias == nullptr || a.out == nullptr || (((uintptr_t)a.wstream) & 15)) return SNERF_ERR_ARG;
  if (!EMBED && (a.E == nullptr || (a.ldE % 8) != 0 || (((uintptr_t)a.E) & 15))) return SNERF_ERR_ARG;
  constexpr int LDS = FM_RING * FM_SLOT + FM_BIAS_MAX * 128 + (STORE ? FM_WAVES * 4096 : 0);   // + the transposition slabs of the training stores
  static bool attr_set = false;
  static int n_cu = 256;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)fmlp_kernel<NET, EMBED, STORE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      n_cu = prop.multiProcessorCount;
    attr_set = true;
  }
  const int grid = a.tiles < n_cu * FM_WG_PER_CU ? a.tiles : n_cu * FM_WG_PER_CU;
  hipLaunchKernelGGL((fmlp_kernel<NET, EMBED, STORE>), dim3(grid), dim3(64 * FM_WAVES), LDS, (hipStream_t)stream, a);
  return snerf_check_launch();
}

extern "C" int snerf_fmlp_classic_fwd(const void* E, long ldE, const void* VE, long ldVE, const void* wstream, long n_frags, const float* bias,
                                      int n_blocks, float* raw, long M, void* stream) {
  if (M <= 0) return SNERF_OK;
  if (VE == nullptr || (ldVE % 8) != 0 || (((uintptr_t)VE) & 15) || (((uintptr_t)raw) & 15)) return SNERF_ERR_ARG;
  FmlpArgs a{};
  a.E = (const __bf16*)E; a.ldE = ldE; a.VE = (const __bf16*)VE; a.ldVE = ldVE; a.S = 1; a.wstream = (const char*)wstream; a.bias = bias; a.out = raw;
  a.M = M; a.tiles = (int)((M + FM_TILE_ROWS - 1) / FM_TILE_ROWS); a.n_chunks = (int)(n_frags / FM_CHUNK); a.n_blocks = n_blocks;
  return fmlp_launch<FMLP_CLASSIC, false>(a, FMLP_CLASSIC_FRAGS, FMLP_CLASSIC_BLOCKS, n_frags, stream);
}

// Training forward of the same network: additionally stores the outputs of the ten hidden layers for the backward pass --
// acts[i] / act_ld[i] (HOST arrays of 10 device pointers / row strides in elements): pts_linears.0 .. .7 (256 wide), feature_linear
// (256), views_linears.0 (128); every pointer 16-byte aligned, every stride a multiple of 8 -- and bits[i] (HOST array of 9 device
// pointers): the ReLU bit masks of pts_linears.i (i < 8: 4 * 8 * ceil(M / 256) * 4 * 64 bytes each) and of views_linears.0 (i = 8:
// half that) in the layout snerf_linear_fwd's ACT_MASK_BITS reads.
extern "C" int snerf_fmlp_classic_train_fwd(const void* E, long ldE, const void* VE, long ldVE, const void* wstream, long n_frags,
                                            const float* bias, int n_blocks, float* raw, void* const* acts, const long* act_ld,
                                            void* const* bits, long M, void* stream) {
  if (M <= 0) return SNERF_OK;
  if (VE == nullptr || (ldVE % 8) != 0 || (((uintptr_t)VE) & 15) || (((uintptr_t)raw) & 15) || acts == nullptr || act_ld == nullptr || bits == nullptr)
    return SNERF_ERR_ARG;
  FmlpArgs a{};
  a.E = (const __bf16*)E; a.ldE = ldE; a.VE = (const __bf16*)VE; a.ldVE = ldVE; a.S = 1; a.wstream = (const char*)wstream; a.bias = bias; a.out = raw;
  a.M = M; a.tiles = (int)((M + FM_TILE_ROWS - 1) / FM_TILE_ROWS); a.n_chunks = (int)(n_frags / FM_CHUNK); a.n_blocks = n_blocks;
  for (int i = 0; i < 10; ++i) {
    if (acts[i] == nullptr || (((uintptr_t)acts[i]) & 15) || (act_ld[i] % 8) != 0) return SNERF_ERR_ARG;
    a.act[i] = (__bf16*)acts[i]; a.act_ld[i] = act_ld[i];
    if (i < 9) {
      if (bits[i] == nullptr || (((uintptr_t)bits[i]) & 15)) return SNERF_ERR_ARG;
      a.bits[i] = (unsigned*)bits[i];
    }
  }
  return fmlp_launch<FMLP_CLASSIC, false, true>(a, FMLP_CLASSIC_FRAGS, FMLP_CLASSIC_BLOCKS, n_frags, stream);
}

// the same network with the positional encodings computed in the kernel: pts [M,3] fp32 sample positions, viewdirs [M / S, ldvd]
extern "C" int snerf_fmlp_classic_pts_fwd(const float* pts, const float* viewdirs, long ldvd, int S, const void* wstream, long n_frags,
                                          const float* bias, int n_blocks, float* raw, long M, void* stream) {
  if (M <= 0) return SNERF_OK;
  if (pts == nullptr || viewdirs == nullptr || S <= 0 || ldvd < 3 || (((uintptr_t)raw) & 15) || M >= (1L << 31)) return SNERF_ERR_ARG;
  FmlpArgs a{};
  a.pts = pts; a.viewdirs = viewdirs; a.ldvd = ldvd; a.S = S; a.wstream = (const char*)wstream; a.bias = bias; a.out = raw;
  a.M = M; a.tiles = (int)((M + FM_TILE_ROWS - 1) / FM_TILE_ROWS); a.n_chunks = (int)(n_frags / FM_CHUNK); a.n_blocks = n_blocks;
  return fmlp_launch<FMLP_CLASSIC, true>(a, FMLP_CLASSIC_FRAGS, FMLP_CLASSIC_BLOCKS, n_frags, stream);
}

extern "C" int snerf_fmlp_proposal_fwd(const void* E, long ldE, const void* wstream, long n_frags, const float* bias, int n_blocks,
                                       float* raw_density, long M, void* stream) {
  FmlpArgs a{};
  a.E = (const __bf16*)E; a.ldE = ldE; a.S = 1; a.wstream = (const char*)wstream; a.bias = bias; a.out = raw_density;
  a.M = M; a.tiles = (int)((M + FM_TILE_ROWS - 1) / FM_TILE_ROWS); a.n_chunks = (int)(n_frags / FM_CHUNK); a.n_blocks = n_blocks;
  // 8 x 6 + 3 x 128 + 16 fragments; 32 + 1 blocks
  return fmlp_launch<FMLP_PROPOSAL, false>(a, 448, 33, n_frags, stream);
}

// training forward of the proposal MLP: acts[0..3] = outputs of layers.0 .. .3 (256 wide)
extern "C" int snerf_fmlp_proposal_train_fwd(const void* E, long ldE, const void* wstream, long n_frags, const float* bias, int n_blocks,
                                             float* raw_density, void* const* acts, const long* act_ld, void* const* bits, long M,
                                             void* stream) {
  if (M <= 0) return SNERF_OK;
  if (acts == nullptr || act_ld == nullptr || bits == nullptr) return SNERF_ERR_ARG;
  FmlpArgs a{};
  a.E = (const __bf16*)E; a.ldE = ldE; a.S = 1; a.wstream = (const char*)wstream; a.bias = bias; a.out = raw_density;
  a.M = M; a.tiles = (int)((M + FM_TILE_ROWS - 1) / FM_TILE_ROWS); a.n_chunks = (int)(n_frags / FM_CHUNK); a.n_blocks = n_blocks;
  for (int i = 0; i < 4; ++i) {
    if (acts[i] == nullptr || (((uintptr_t)acts[i]) & 15) || (act_ld[i] % 8) != 0) return SNERF_ERR_ARG;
    if (bits[i] == nullptr) return SNERF_ERR_ARG;
    a.act[i] = (__bf16*)acts[i]; a.act_ld[i] = act_ld[i]; a.bits[i] = (unsigned*)bits[i];
  }
  return fmlp_launch<FMLP_PROPOSAL, false, true>(a, 448, 33, n_frags, stream);
}

// ---- colour head (cond_layers.0..2 + rgb_layer of the mip path's NeRF MLP, hidden 1024) -------------------------------------------
// NeRF MLP of the zipnerf path, inference (fzip_fwd_kernel).  wstream / bias: ZipNerfNet._pack_fused_infer (snerf_amd/mlp.py) through fmlp_pack;
// dtype SNERF_DT_BF16 or SNERF_DT_F16 = the type of F, D and of the weight stream.
extern "C" int snerf_fmlp_zip_fwd(const void* F, long ldF, const void* D, long ldD, const void* wstream, long n_frags, const float* bias, int n_blocks,
                                  float* raw_rgb, long ld_rgb, float* raw_d, long ld_d, void* x32, long ld_x, long M, int dtype, void* stream) {
  if (M <= 0) return SNERF_OK;
  if (x32 != nullptr && (ld_x < 32 || (ld_x % 4) || (((uintptr_t)x32) & 7))) return SNERF_ERR_ARG;
  if (F == nullptr || D == nullptr || wstream == nullptr || bias == nullptr || raw_rgb == nullptr || raw_d == nullptr || ldF < 64 || ldD < 16 || (ldF % 8) ||
      (ldD % 8) || (((uintptr_t)F) & 15) || (((uintptr_t)D) & 15) || ld_rgb < 3 || ld_d < 1 || M >= (1L << 31) || n_blocks != FZIP_BLOCKS ||
      n_frags != ((FZIP_FRAGS + FM_CHUNK - 1) / FM_CHUNK) * FM_CHUNK || (dtype != SNERF_DT_BF16 && dtype != SNERF_DT_F16))
    return SNERF_ERR_ARG;
  FzipArgs a{(const __bf16*)F, ldF, (const __bf16*)D, ldD, (const char*)wstream, bias, raw_rgb, ld_rgb, raw_d, ld_d, (__bf16*)x32, ld_x, M, 0, (int)(n_frags / FM_CHUNK), n_blocks, {}, {}, {}};
  a.tiles = (int)((M + FM_TILE_ROWS - 1) / FM_TILE_ROWS);
  const int lds = FM_RING * FM_SLOT + FM_BIAS_MAX * 128;
  static bool attr = false;
  static int n_cu = 256;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)fzip_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)fzip_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) n_cu = prop.multiProcessorCount;
    attr = true;
  }
  const int grid = a.tiles < n_cu * FM_WG_PER_CU ? a.tiles : n_cu * FM_WG_PER_CU;
  if (dtype == SNERF_DT_F16) hipLaunchKernelGGL(fzip_fwd_kernel<true>, dim3(grid), dim3(64 * FM_WAVES), lds, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(fzip_fwd_kernel<false>, dim3(grid), dim3(64 * FM_WAVES), lds, (hipStream_t)stream, a);
  return snerf_check_launch();
}

// ... and the same network as the TRAINING forward of a ZipTrainer step: additionally stores what the backward reads -- acts[0] = H1 [M, >= 64]
// (density_layer.0), acts[1] = x, acts[2] = h (lin_second_stage_0), acts[3] = H3 (lin_second_stage_1) [M, >= 256] each (compute dtype, row
// strides act_ld, 16-byte aligned; x and h are normally column ranges of the [h | x | dir] operand of lin_second_stage_1's weight gradient) and
// bits[0] / bits[1] / bits[2] = the ReLU bit masks of H1 ([M, 64]) / h / H3 ([M, 256]; snerf_linear_fwd's ACT_RELU_BITS layout) -- through the per-wave transposition
// slabs of the other store-carrying fused kernels.  Replaces seven launches that move 10 GB per 65 536-ray step by one that writes 3.5 GB.
extern "C" int snerf_fmlp_zip_train_fwd(const void* F, long ldF, const void* D, long ldD, const void* wstream, long n_frags, const float* bias,
                                        int n_blocks, float* raw_rgb, long ld_rgb, float* raw_d, long ld_d, void* const* acts, const long* act_ld,
                                        void* const* bits, long M, int dtype, void* stream) {
  if (M <= 0) return SNERF_OK;
  if (F == nullptr || D == nullptr || wstream == nullptr || bias == nullptr || raw_rgb == nullptr || raw_d == nullptr || acts == nullptr || act_ld == nullptr ||
      bits == nullptr || ldF < 64 || ldD < 16 || (ldF % 8) || (ldD % 8) || (((uintptr_t)F) & 15) || (((uintptr_t)D) & 15) || ld_rgb < 3 || ld_d < 1 ||
      M >= (1L << 31) || n_blocks != FZIP_BLOCKS || n_frags != ((FZIP_FRAGS + FM_CHUNK - 1) / FM_CHUNK) * FM_CHUNK ||
      (dtype != SNERF_DT_BF16 && dtype != SNERF_DT_F16))
    return SNERF_ERR_ARG;
  FzipArgs a{(const __bf16*)F, ldF, (const __bf16*)D, ldD, (const char*)wstream, bias, raw_rgb, ld_rgb, raw_d, ld_d, nullptr, 0, M, 0, (int)(n_frags / FM_CHUNK), n_blocks, {}, {}, {}};
  for (int i = 0; i < 4; ++i) {
    if (acts[i] == nullptr || (((uintptr_t)acts[i]) & 15) || (act_ld[i] % 8) != 0 || act_ld[i] < (i == 0 ? 64 : 256)) return SNERF_ERR_ARG;
    a.act[i] = (__bf16*)acts[i]; a.act_ld[i] = act_ld[i];
  }
  for (int i = 0; i < 3; ++i) {
    if (bits[i] == nullptr || (((uintptr_t)bits[i]) & 15)) return SNERF_ERR_ARG;
    a.bits[i] = (unsigned*)bits[i];
  }
  a.tiles = (int)((M + FM_TILE_ROWS - 1) / FM_TILE_ROWS);
  const int lds = FM_RING * FM_SLOT + FM_BIAS_MAX * 128 + FM_WAVES * 4096;
  static bool attr = false;
  static int n_cu = 256;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)fzip_fwd_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)fzip_fwd_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) n_cu = prop.multiProcessorCount;
    attr = true;
  }
  const int grid = a.tiles < n_cu * FM_WG_PER_CU ? a.tiles : n_cu * FM_WG_PER_CU;
  if (dtype == SNERF_DT_F16) hipLaunchKernelGGL((fzip_fwd_kernel<true, true>), dim3(grid), dim3(64 * FM_WAVES), lds, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((fzip_fwd_kernel<false, true>), dim3(grid), dim3(64 * FM_WAVES), lds, (hipStream_t)stream, a);
  return snerf_check_launch();
}

static int fcolour_grid(int tiles) {
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    n_cu = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      n_cu = prop.multiProcessorCount;
  }
  return tiles < n_cu ? tiles : n_cu;
}

// raw_rgb [M,3] fp32 = rgb_layer(cond_layers.2(cond_layers.1(cond_layers.0(CB[:, :1051])))) (models.py:283-296).  CB [M, ldCB] bf16 =
// [bottleneck 1024 | view encoding 27 | zeros up to column 1056]; wstream / bias from mlp.fmlp_pack (cond_layers.0 k-major).
// acts / act_ld / bits (HOST arrays of 3; all nullptr for inference): where the three hidden activations ([M, >= 128] bf16) and their
// ReLU bit masks (snerf_linear_fwd's ACT_RELU_BITS layout for N = 128) are stored for the backward pass.
extern "C" int snerf_fcolour_fwd(const void* CB, long ldCB, const void* wstream, long n_frags, const float* bias, int n_blocks, float* raw_rgb,
                                 void* const* acts, const long* act_ld, void* const* bits, long M, int variant, void* stream) {
  if (M <= 0) return SNERF_OK;
  if (CB == nullptr || wstream == nullptr || bias == nullptr || raw_rgb == nullptr || (((uintptr_t)CB) & 15) || (((uintptr_t)wstream) & 15) ||
      (ldCB % 8) != 0 || ldCB < 16 * FC_NK0 || n_frags != FC_FWD_FRAGS || n_blocks != FC_FWD_BLOCKS)
    return SNERF_ERR_ARG;
  ColourFwdArgs a{};
  a.CB = (const __bf16*)CB; a.ldCB = ldCB; a.wstream = (const char*)wstream; a.bias = bias; a.raw_rgb = raw_rgb; a.M = M;
  a.tiles = (int)((M + FM_TILE_ROWS - 1) / FM_TILE_ROWS); a.n_chunks = (int)(n_frags / FM_CHUNK); a.n_blocks = n_blocks;
  const bool store = acts != nullptr;
  if (store) {
    if (act_ld == nullptr || bits == nullptr) return SNERF_ERR_ARG;
    for (int i = 0; i < 3; ++i) {
      if (acts[i] == nullptr || bits[i] == nullptr || (((uintptr_t)acts[i]) & 15) || (act_ld[i] % 8) != 0 || act_ld[i] < 128) return SNERF_ERR_ARG;
      a.act[i] = (__bf16*)acts[i]; a.act_ld[i] = act_ld[i]; a.bits[i] = (unsigned*)bits[i];
    }
  }
  const int lds = FM_RING * FM_SLOT + FM_BIAS_MAX * 128 + FM_WAVES * 4096;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)fcolour_fwd_kernel<false, FC_QD>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)fcolour_fwd_kernel<true, FC_QD>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)fcolour_fwd_kernel<false, FC_QD_ALT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)fcolour_fwd_kernel<true, FC_QD_ALT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  const int grid = fcolour_grid(a.tiles);
  const dim3 g(grid), b(64 * FM_WAVES);
  hipStream_t st = (hipStream_t)stream;
  if (variant & 1) {                                    // tools/fcolour_probe.py: the alternative read-ahead depth
    if (store) hipLaunchKernelGGL((fcolour_fwd_kernel<true, FC_QD_ALT>), g, b, lds, st, a);
    else hipLaunchKernelGGL((fcolour_fwd_kernel<false, FC_QD_ALT>), g, b, lds, st, a);
  } else {
    if (store) hipLaunchKernelGGL((fcolour_fwd_kernel<true, FC_QD>), g, b, lds, st, a);
    else hipLaunchKernelGGL((fcolour_fwd_kernel<false, FC_QD>), g, b, lds, st, a);
  }
  return snerf_check_launch();
}

// workspace floats snerf_fcolour_bwd needs for M rows
extern "C" long snerf_fcolour_bwd_ws_floats(long M) {
  if (M <= 0) return 0;
  return (long)fcolour_grid((int)((M + FM_TILE_ROWS - 1) / FM_TILE_ROWS)) * FC_BWD_COLS;
}

// Data-gradient chain of the colour head: d_raw_rgb [M,3] fp32 -> dC[0..2] = d pre-activation of cond_layers.2, .1, .0 ([M, >= 128]
// bf16 each) and dB = d pre-activation of the bottleneck layer ([M, >= 1024] bf16); bits[0..3] = ReLU bit masks of cond_layers.2, .1,
// .0 (N = 128) and of the bottleneck (N = 1024); wstream from mlp.fmlp_pack of the transposed weights; the bias gradients of the four
// layers are ADDED to g_bias[0..3] (cond_layers.2, .1, .0: 128 floats, bottleneck: 1024) in a fixed order.  ws: snerf_fcolour_bwd_ws_floats(M).
extern "C" int snerf_fcolour_bwd(const float* d_raw_rgb, const void* wstream, long n_frags, void* const* bits, void* const* dC, const long* dC_ld,
                                 void* dB, long dB_ld, float* const* g_bias, float* ws, long ws_floats, long M, void* stream) {
  if (M <= 0) return SNERF_OK;
  if (d_raw_rgb == nullptr || wstream == nullptr || bits == nullptr || dC == nullptr || dC_ld == nullptr || dB == nullptr || g_bias == nullptr ||
      ws == nullptr || n_frags != FC_BWD_FRAGS || (((uintptr_t)wstream) & 15) || (((uintptr_t)dB) & 15) || (dB_ld % 8) != 0 || dB_ld < 1024 ||
      ws_floats < snerf_fcolour_bwd_ws_floats(M))
    return SNERF_ERR_ARG;
  ColourBwdArgs a{};
  a.d_rgb = d_raw_rgb; a.wstream = (const char*)wstream; a.dB = (__bf16*)dB; a.dB_ld = dB_ld; a.colsum_ws = ws; a.M = M;
  a.tiles = (int)((M + FM_TILE_ROWS - 1) / FM_TILE_ROWS); a.n_chunks = (int)(n_frags / FM_CHUNK);
  for (int i = 0; i < 4; ++i) {
    if (bits[i] == nullptr || (((uintptr_t)bits[i]) & 15) || g_bias[i] == nullptr) return SNERF_ERR_ARG;
    a.bits[i] = (const unsigned*)bits[i];
  }
  for (int i = 0; i < 3; ++i) {
    if (dC[i] == nullptr || (((uintptr_t)dC[i]) & 15) || (dC_ld[i] % 8) != 0 || dC_ld[i] < 128) return SNERF_ERR_ARG;
    a.dC[i] = (__bf16*)dC[i]; a.dC_ld[i] = dC_ld[i];
  }
  if (M * 12 >= (1L << 31) || (((uintptr_t)d_raw_rgb) & 15)) return SNERF_ERR_ARG;      // d raw_rgb goes through a 32-bit buffer descriptor
  const int lds = FC_BWD_LDS;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)fcolour_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  const int grid = fcolour_grid(a.tiles);
  hipLaunchKernelGGL(fcolour_bwd_kernel, dim3(grid), dim3(64 * FM_WAVES), lds, (hipStream_t)stream, a);
  hipLaunchKernelGGL(fcolour_colsum_fold_kernel, dim3(FC_BWD_COLS / 64), dim3(256), 0, (hipStream_t)stream, ws, grid, g_bias[0], g_bias[1],
                     g_bias[2], g_bias[3]);
  return snerf_check_launch();
}

// ---- fused data-gradient chains of the 256-wide networks --------------------------------------------------------------------------
#define FCH_LDS(NCOLS) (FCH_RING * FM_SLOT + FM_WAVES * 4096 + FM_WAVES * FCH_WAVE_BYTES + FCH_TAB(NCOLS) * 4)
extern "C" long snerf_fchain_bwd_ws_floats(int net, long M) {
  if (M <= 0 || (net != FMLP_CLASSIC && net != FMLP_PROPOSAL)) return 0;
  const int ncols = net == FMLP_CLASSIC ? FCH_CLASSIC_COLS : FCH_PROPOSAL_COLS;
  return (long)fcolour_grid((int)((M + FM_TILE_ROWS - 1) / FM_TILE_ROWS)) * ncols + 64;       // + the fold's two small tables
}

// net 0 (classic NeRF, run_nerf_helpers.py:83-139): d_raw [M,4] fp32 (d rgb, d alpha) -> dz[0] = d pre-activation of views_linears.0
// ([M, >= 128] bf16), dz[1] = d feature_linear output ([M, >= 256]), dz[2..9] = d pre-activation of pts_linears.7 .. .0; bits[0..7] =
// the ReLU bit masks of pts_linears.0 .. .7, bits[8] = of views_linears.0 (snerf_fmlp_classic_train_fwd wrote them); n_steps = 10.
// net 1 (proposal MLP of the mip path, s-nerf/model/models.py:237-262 with the proposal widths): d_raw [M,1] (d raw density) -> dz[0..3]
// = d pre-activation of layers.3 .. .0; bits[0..3] of layers.0 .. .3; n_steps = 4.
// wstream: mlp.fmlp_pack of the transposed weights in chain order.  The bias gradient of step i is ADDED to g_bias[i] (the workgroups'
// partial sums are reduced in LDS in arrival order: NOT bit-reproducible -- the deterministic mode uses the per-layer kernels).
extern "C" int snerf_fchain_bwd(int net, const float* d_raw, const void* wstream, long n_frags, void* const* bits, void* const* dz, const long* dz_ld,
                                float* const* g_bias, float* ws, long ws_floats, long M, void* stream) {
  if (M <= 0) return SNERF_OK;
  if (net != FMLP_CLASSIC && net != FMLP_PROPOSAL) return SNERF_ERR_ARG;
  const bool classic = net == FMLP_CLASSIC;
  const int n_steps = classic ? 10 : 4, n_bits = classic ? 9 : 4, ncols = classic ? FCH_CLASSIC_COLS : FCH_PROPOSAL_COLS, dc = classic ? 4 : 1;
  if (d_raw == nullptr || wstream == nullptr || bits == nullptr || dz == nullptr || dz_ld == nullptr || g_bias == nullptr || ws == nullptr ||
      n_frags != (classic ? FCH_CLASSIC_FRAGS : FCH_PROPOSAL_FRAGS) || (((uintptr_t)wstream) & 15) || (((uintptr_t)d_raw) & 15) ||
      ws_floats < snerf_fchain_bwd_ws_floats(net, M) || M * dc * 4 >= (1L << 31))
    return SNERF_ERR_ARG;
  ChainArgs a{};
  a.d_raw = d_raw; a.d_cols = dc; a.wstream = (const char*)wstream; a.colsum_ws = ws; a.M = M; a.n_cols = ncols;
  a.tiles = (int)((M + FM_TILE_ROWS - 1) / FM_TILE_ROWS); a.n_chunks = (int)(n_frags / FM_CHUNK);
  for (int i = 0; i < n_bits; ++i) {
    if (bits[i] == nullptr || (((uintptr_t)bits[i]) & 15)) return SNERF_ERR_ARG;
    a.bits[i] = (const unsigned*)bits[i];
  }
  ChainFoldTab tab{};
  int col = 0;
  for (int i = 0; i < 10; ++i) {
    if (i < n_steps) {
      const int width = (classic && i == 0) ? 128 : 256;
      if (dz[i] == nullptr || (((uintptr_t)dz[i]) & 15) || (dz_ld[i] % 8) != 0 || dz_ld[i] < width || g_bias[i] == nullptr) return SNERF_ERR_ARG;
      a.dz[i] = (__bf16*)dz[i]; a.dz_ld[i] = dz_ld[i];
      tab.dst[i] = g_bias[i]; tab.first[i] = col; col += width;
    } else {
      tab.dst[i] = nullptr; tab.first[i] = 1 << 30;
    }
  }
  const int grid = fcolour_grid(a.tiles);
  const int lds = FCH_LDS(ncols);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)fchain_bwd_kernel<FMLP_CLASSIC>, hipFuncAttributeMaxDynamicSharedMemorySize, FCH_LDS(FCH_CLASSIC_COLS));
    (void)hipFuncSetAttribute((const void*)fchain_bwd_kernel<FMLP_PROPOSAL>, hipFuncAttributeMaxDynamicSharedMemorySize, FCH_LDS(FCH_PROPOSAL_COLS));
    attr_set = true;
  }
  hipStream_t st = (hipStream_t)stream;
  if (classic) hipLaunchKernelGGL(fchain_bwd_kernel<FMLP_CLASSIC>, dim3(grid), dim3(64 * FM_WAVES), lds, st, a);
  else hipLaunchKernelGGL(fchain_bwd_kernel<FMLP_PROPOSAL>, dim3(grid), dim3(64 * FM_WAVES), lds, st, a);
  hipLaunchKernelGGL(fchain_colsum_fold_kernel, dim3((ncols + 63) / 64), dim3(256), 0, st, (const float*)ws, grid, ncols, tab);
  return snerf_check_launch();
}

// Data-gradient chain of the zipnerf NeRF MLP (fzip_chain_bwd_kernel; the backward of snerf_fmlp_zip_train_fwd).  d_rgb [M, ld_rgb >= 3], d_den
// [M, ld_den >= den_cols], den_cols <= 32 (d raw density, then the semantic logits' gradients) fp32; bits[0..2] = the bit masks of H1, h, H3 the
// training forward wrote; dz[0] = dH3, dz[1] = dh, dz[2] = dx ([M, >= 256] each), dz[3] = dH1, dz[4] = dF ([M, >= 64]) in `dtype`; the bias
// gradients of lin_second_stage_1, lin_second_stage_0, density_layer.2, density_layer.0 are ADDED to g_bias[0..3] (arrival-order LDS atomics:
// not bit-reproducible; the deterministic mode keeps the per-layer kernels).  wstream: ZipNerfNet._pack_fused_chain (448 fragments).
extern "C" long snerf_fmlp_zip_chain_ws_floats(long M) {
  if (M <= 0) return 0;
  return (long)fcolour_grid((int)((M + FM_TILE_ROWS - 1) / FM_TILE_ROWS)) * FZCH_COLS + 64;
}
extern "C" int snerf_fmlp_zip_chain_bwd(const float* d_rgb, long ld_rgb, const float* d_den, long ld_den, int den_cols, const void* wstream, long n_frags,
                                        void* const* bits, void* const* dz, const long* dz_ld, float* const* g_bias, float* ws, long ws_floats, long M,
                                        int dtype, void* stream) {
  if (M <= 0) return SNERF_OK;
  if (d_rgb == nullptr || d_den == nullptr || wstream == nullptr || bits == nullptr || dz == nullptr || dz_ld == nullptr || g_bias == nullptr || ws == nullptr ||
      ld_rgb < 3 || den_cols < 1 || den_cols > 32 || ld_den < den_cols || n_frags != FZCH_FRAGS || (((uintptr_t)wstream) & 15) ||
      ws_floats < snerf_fmlp_zip_chain_ws_floats(M) || M >= (1L << 31) || (dtype != SNERF_DT_BF16 && dtype != SNERF_DT_F16))
    return SNERF_ERR_ARG;
  ZipChainArgs a{};
  a.d_rgb = d_rgb; a.ld_rgb = ld_rgb; a.d_den = d_den; a.ld_den = ld_den; a.den_cols = den_cols; a.wstream = (const char*)wstream; a.colsum_ws = ws; a.M = M;
  a.tiles = (int)((M + FM_TILE_ROWS - 1) / FM_TILE_ROWS); a.n_chunks = (int)(n_frags / FM_CHUNK);
  for (int i = 0; i < 3; ++i) {
    if (bits[i] == nullptr || (((uintptr_t)bits[i]) & 15)) return SNERF_ERR_ARG;
    a.bits[i] = (const unsigned*)bits[i];
  }
  const int grid = fcolour_grid(a.tiles);
  ChainFoldTab tab{};
  const int widths[5] = {256, 256, 256, 64, 64};
  int col = 0;
  for (int i = 0; i < 10; ++i) {
    if (i < 5) {
      if (dz[i] == nullptr || (((uintptr_t)dz[i]) & 15) || (dz_ld[i] % 8) != 0 || dz_ld[i] < widths[i] || (i < 4 && g_bias[i] == nullptr)) return SNERF_ERR_ARG;
      a.dz[i] = (__bf16*)dz[i]; a.dz_ld[i] = dz_ld[i];
      tab.dst[i] = i < 4 ? g_bias[i] : ws + (long)grid * FZCH_COLS;      // (the last step's column sums are not a bias gradient: into the workspace's tail)
      tab.first[i] = col; col += widths[i];
    } else {
      tab.dst[i] = nullptr; tab.first[i] = 1 << 30;
    }
  }
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)fzip_chain_bwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, FZCH_LDS);
    (void)hipFuncSetAttribute((const void*)fzip_chain_bwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, FZCH_LDS);
    attr_set = true;
  }
  hipStream_t st = (hipStream_t)stream;
  (void)hipMemsetAsync(ws + (long)grid * FZCH_COLS, 0, 64 * sizeof(float), st);
  if (dtype == SNERF_DT_F16) hipLaunchKernelGGL(fzip_chain_bwd_kernel<true>, dim3(grid), dim3(64 * FM_WAVES), FZCH_LDS, st, a);
  else hipLaunchKernelGGL(fzip_chain_bwd_kernel<false>, dim3(grid), dim3(64 * FM_WAVES), FZCH_LDS, st, a);
  hipLaunchKernelGGL(fchain_colsum_fold_kernel, dim3((FZCH_COLS + 63) / 64), dim3(256), 0, st, (const float*)ws, grid, FZCH_COLS, tab);
  return snerf_check_launch();
}
