// WITHDRAWN EXPERIMENT (round 3): the two-workgroups-per-CU NT kernel, as it stood in snerf_amd/csrc/gemm.hip (variant bit 14 of
// snerf_linear_fwd).  Not built: it uses gemm.hip's GemmNT / nt_epilogue definitions.  Measurements: profiles/r3_w_gemm_nt4_probe.txt,
// profiles/r3_w_gemm_nt4_pmc.txt (0.69 of the shipped kernel's speed).
// ---------------------------------------------------------------------------
// NT kernel, TWO independent 4-wave workgroups per CU (DESIGN r2 section 7.0b; variant bit 14 of snerf_linear_fwd): 128 x 256 tile,
// bf16, wave w owns all 128 rows x columns [64 w, 64 w + 64) (128 accumulator registers, 768 B of LDS fragments per MFMA as in the
// 8-phase kernels), BK = 32 so that a stage is 24 KiB and a ring of three fits twice into the 160 KiB of a CU (the epilogue's
// transposition slabs reuse the ring).  One tile per workgroup, one barrier per k-tile (16 MFMAs per wave); the latencies a
// workgroup cannot hide -- the ring's DMA two k-tiles ahead, the fragment reads, above all its epilogue's store drain -- are to be
// covered by the OTHER workgroup of the CU, whose waves share the SIMDs but none of the barriers.
// MEASURED (round 3, tools/gemm_nt4_probe.py, profiles/r3_w_*): correct (same result as the shipped kernel), and 0.69 of its speed --
// M = 524 288, N = K = 1024: 1390 us = 790 TFLOP/s against 950-1090 us = 1007-1162; K = 1152: 1524 vs 1062 us; M = 65 536: 168 vs 117 us.
// PMC against the shipped kernel: 1.65 x the issue cycles (ring bookkeeping and 64-bit DMA addresses per 16 instead of 32 MFMAs), 1.45 x
// the wait cycles, FETCH 2.09 vs 1.62 GB (W re-read per 128 instead of 256 rows), LDS bank conflicts 10 % of LDS-active cycles.  The
// structural part: 160 KiB per CU hold 2 x 48 KiB of operands in flight for a demand of 47 B per clock and CU = 2000 clocks of latency
// cover, where the 8-phase kernel holds 96 KiB for 32 B per clock = 3000 clocks -- the second workgroup pays for its independence with
// the prefetch depth.  Kept as an experiment (forward flavours only, no bias gradient); nothing ships through it.
// LDS layout of a stage: rows of 64 bytes, two per 128-byte line; 16-byte chunk c of tile row r sits in line r >> 1 at chunk position
// ((4 (r & 1) + c) ^ ((r >> 2) & 3)): the 16 lanes of a ds_read_b128 pass (16 consecutive rows, one k-chunk) hit 16 different
// 16-byte bank groups.  The DMA writes lane i of a 1 KiB piece to line i >> 3, position i & 7, so the swizzle is applied to the
// SOURCE address: that position holds row 2 line + (x >> 2), chunk x & 3 with x = (i & 7) ^ ((line >> 1) & 3).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void gemm_nt4_kernel(GemmNT p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __bf16 T;
  constexpr int BM = 128, BN = 256, STAGE = (BM + BN) * 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = p.N / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int logical = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (logical / tiles_n) * BM, n0 = (logical % tiles_n) * BN;
  const T* __restrict__ A = (const T*)p.A;
  const T* __restrict__ W = (const T*)p.W;
  const int KT = p.K / 32;
  // staging sources: A = pieces 0..7 (16 rows each), B = pieces 0..15; wave w takes A pieces 2 w, 2 w + 1 and B pieces 4 w .. 4 w + 3
  long a_off[2], b_off[4];
  {
    const int line = lane >> 3, pos = lane & 7;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int L = (wave * 2 + i) * 8 + line;                  // line of the A part of the stage = tile row >> 1
      const int x = pos ^ ((L >> 1) & 3);
      int gr = m0 + 2 * L + (x >> 2);
      gr = gr < p.M ? gr : p.M - 1;
      a_off[i] = (long)gr * p.lda + (x & 3) * 8;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int L = (wave * 4 + i) * 8 + line;
      const int x = pos ^ ((L >> 1) & 3);
      b_off[i] = (long)(n0 + 2 * L + (x >> 2)) * p.ldw + (x & 3) * 8;
    }
  }
  auto issue = [&](int kt, int stage) __attribute__((always_inline)) {
    char* sA = smem + stage * STAGE;
    char* sB = sA + BM * 64;
    const long k0 = (long)kt * 32;
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16(A + a_off[i] + k0, sA + (wave * 2 + i) * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(W + b_off[i] + k0, sB + (wave * 4 + i) * 1024);
  };
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // fragment addresses inside a stage for k-step 0 (k-step 1: chunk + 2, i.e. position ^ 2)
  int fa[4], fb[2];
  {
    const int chalf = lane >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = i * 32 + (lane & 31);
      fa[i] = (r >> 1) * 128 + ((((r & 1) << 2 | chalf) ^ ((r >> 2) & 3)) << 4);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = wave * 64 + j * 32 + (lane & 31);
      fb[j] = BM * 64 + (r >> 1) * 128 + ((((r & 1) << 2 | chalf) ^ ((r >> 2) & 3)) << 4);
    }
  }
  bf16x8 a0[4], b0[2], a1[4], b1[2];
  auto read = [&](int stage, int ks, bf16x8 (&a)[4], bf16x8 (&b)[2]) __attribute__((always_inline)) {
    const char* base = smem + stage * STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = *(const bf16x8*)(base + (fa[i] ^ (ks << 5)));
#pragma unroll
    for (int j = 0; j < 2; ++j) b[j] = *(const bf16x8*)(base + (fb[j] ^ (ks << 5)));
  };
  auto mfma8 = [&](const bf16x8 (&a)[4], const bf16x8 (&b)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) mma32(acc[i][j], b[j], a[i]);   // D = W-tile . X-tile^T: lanes own rows m
  };
  issue(0, 0);
  if (KT > 1) issue(1, 1);
  if (KT > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  read(0, 0, a0, b0);
  int st = 0;                                                   // stage of k-tile kt
  for (int kt = 0; kt < KT; ++kt) {
    const int st1 = st == 2 ? 0 : st + 1, st2 = st1 == 2 ? 0 : st1 + 1;
    if (kt + 2 < KT) issue(kt + 2, st2);                        // (its previous content, k-tile kt - 1, was last read before the barrier below of the previous iteration)
    read(st, 1, a1, b1);
    mfma8(a0, b0);
    if (kt + 2 < KT) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // k-tile kt + 1 has landed (this wave's pieces)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 1 < KT) read(st1, 0, a0, b0);
    mfma8(a1, b1);
    st = st1;
  }
  nt_epilogue<T, BM, BN, 1, 4>(p, acc, smem, m0, n0, wave, lane);
}

