// Hazards between VALU writes and MFMA SrcA / SrcB reads on gfx950, with two waves per SIMD issuing MFMAs in lockstep.  Inline asm on
// FIXED physical registers (v200-v223), so neither the compiler's hazard recogniser nor its register allocator is in the picture.
// One trip:  v_mov  op0 <- X           [RAW gap]   K x v_mfma (same operand registers)   [WAR gap]   v_mov op0 <- garbage   [160 cycles]
// with X alternating between two real values.  Reference = both gaps 160 cycles.  A differing result read op0 too early or too late.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_war_probe mfma_war_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

#define NOP20 "s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n"
#define NOP1 "s_nop 7\n"
#define NOP0 ""
#define MF "v_mfma_f32_32x32x16_bf16 v[208:223], v[200:203], v[204:207], v[208:223]\n"
#define TRIP(OP, X, PRE, MFS, POST) "v_mov_b32 " OP ", " X "\n" PRE MFS POST "v_mov_b32 " OP ", %[garb]\n" NOP20
#define KERNEL_BODY(OP, PRE, MFS, POST)                                                                                                    \
  asm volatile(                                                                                                                            \
      "v_mov_b32 v200, %[a0]\n v_mov_b32 v201, %[a1]\n v_mov_b32 v202, %[a2]\n v_mov_b32 v203, %[a3]\n"                                     \
      "v_mov_b32 v204, %[b0]\n v_mov_b32 v205, %[b1]\n v_mov_b32 v206, %[b2]\n v_mov_b32 v207, %[b3]\n"                                     \
      "v_mov_b32 v208, 0\n v_mov_b32 v209, 0\n v_mov_b32 v210, 0\n v_mov_b32 v211, 0\n v_mov_b32 v212, 0\n v_mov_b32 v213, 0\n v_mov_b32 v214, 0\n v_mov_b32 v215, 0\n" \
      "v_mov_b32 v216, 0\n v_mov_b32 v217, 0\n v_mov_b32 v218, 0\n v_mov_b32 v219, 0\n v_mov_b32 v220, 0\n v_mov_b32 v221, 0\n v_mov_b32 v222, 0\n v_mov_b32 v223, 0\n" \
      "s_mov_b32 s20, %[n]\n s_barrier\n"                                                                                                   \
      "1:\n" TRIP(OP, "%[x0]", PRE, MFS, POST) TRIP(OP, "%[x1]", PRE, MFS, POST)                                                            \
      "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n" NOP20                                                            \
      "v_add_f32 %[o], v208, v209\n v_add_f32 %[o], %[o], v210\n v_add_f32 %[o], %[o], v215\n v_add_f32 %[o], %[o], v223\n"               \
      : [o] "=&v"(o)                                                                                                                       \
      : [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3), [b0] "v"(b0), [b1] "v"(b1), [b2] "v"(b2), [b3] "v"(b3), [x0] "v"(x0), [x1] "v"(x1), \
        [garb] "v"(garb), [n] "s"(iters)                                                                                                   \
      : "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", \
        "v218", "v219", "v220", "v221", "v222", "v223", "s20", "scc", "memory")

// OPB: the written register is B[0] (v204) instead of A[0] (v200)
template <bool OPB, int CFG>
__global__ __launch_bounds__(512) void probe(const unsigned* __restrict__ src, float* __restrict__ out, int iters) {
  const int tid = threadIdx.x;
  const unsigned a0 = src[tid * 4 + 0], a1 = src[tid * 4 + 1], a2 = src[tid * 4 + 2], a3 = src[tid * 4 + 3];
  const unsigned b0 = src[4096 + tid * 4 + 0], b1 = src[4096 + tid * 4 + 1], b2 = src[4096 + tid * 4 + 2], b3 = src[4096 + tid * 4 + 3];
  const unsigned x0 = OPB ? b0 : a0, x1 = x0 ^ 0x00400040u, garb = 0x40004000u;
  float o;
  if (!OPB) {
    if (CFG == 0) KERNEL_BODY("v200", NOP20, MF, NOP20);
    else if (CFG == 1) KERNEL_BODY("v200", NOP0, MF, NOP20);
    else if (CFG == 3) KERNEL_BODY("v200", NOP20, MF, NOP0);
    else if (CFG == 4) KERNEL_BODY("v200", NOP20, MF MF, NOP0);
    else if (CFG == 5) KERNEL_BODY("v200", NOP20, MF MF, NOP1);
    else if (CFG == 6) KERNEL_BODY("v200", NOP20, MF MF MF, NOP0);
    else if (CFG == 7) KERNEL_BODY("v200", NOP20, MF MF, NOP20);
    else KERNEL_BODY("v200", NOP20, MF MF MF, NOP20);
  } else {
    if (CFG == 0) KERNEL_BODY("v204", NOP20, MF, NOP20);
    else if (CFG == 1) KERNEL_BODY("v204", NOP0, MF, NOP20);
    else if (CFG == 3) KERNEL_BODY("v204", NOP20, MF, NOP0);
    else if (CFG == 4) KERNEL_BODY("v204", NOP20, MF MF, NOP0);
    else if (CFG == 6) KERNEL_BODY("v204", NOP20, MF MF MF, NOP0);
    else if (CFG == 7) KERNEL_BODY("v204", NOP20, MF MF, NOP20);
    else KERNEL_BODY("v204", NOP20, MF MF MF, NOP20);
  }
  out[blockIdx.x * 512 + tid] = o;
}

template <bool OPB, int CFG, int REF>
static void run(const char* name, const unsigned* src, float* out, std::vector<float>& ref, std::vector<float>& got) {
  long bad = 0;
  for (int rep = 0; rep < 10; ++rep) {
    hipLaunchKernelGGL((probe<OPB, REF>), dim3(256), dim3(512), 0, 0, src, out, 500);
    hipMemcpy(ref.data(), out, ref.size() * 4, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL((probe<OPB, CFG>), dim3(256), dim3(512), 0, 0, src, out, 500);
    hipMemcpy(got.data(), out, got.size() * 4, hipMemcpyDeviceToHost);
    for (size_t i = 0; i < got.size(); ++i) if (memcmp(&got[i], &ref[i], 4) != 0) ++bad;
  }
  printf("%s %-58s wrong results: %8ld of %zu\n", OPB ? "SrcB" : "SrcA", name, bad, 10 * got.size());
}

int main() {
  std::vector<unsigned> h(8192);
  srand(3);
  for (auto& v : h) {
    float f0 = (rand() / (float)RAND_MAX - 0.5f) * 0.25f, f1 = (rand() / (float)RAND_MAX - 0.5f) * 0.25f;
    unsigned u0, u1; memcpy(&u0, &f0, 4); memcpy(&u1, &f1, 4);
    v = (u0 >> 16) | (u1 & 0xffff0000u);
  }
  unsigned* src; float* out;
  hipMalloc(&src, 8192 * 4); hipMalloc(&out, 256 * 512 * 4);
  hipMemcpy(src, h.data(), 8192 * 4, hipMemcpyHostToDevice);
  std::vector<float> ref(256 * 512), got(256 * 512);
  run<false, 0, 0>("reference repeated", src, out, ref, got);
  run<false, 1, 0>("RAW: VALU write, MFMA right after", src, out, ref, got);
  run<false, 3, 0>("WAR: 1 MFMA, VALU write right after", src, out, ref, got);
  run<false, 4, 7>("WAR: 2 MFMAs (same operands), VALU write right after", src, out, ref, got);
  run<false, 5, 7>("WAR: 2 MFMAs, 8 cycles, VALU write", src, out, ref, got);
  run<false, 6, 8>("WAR: 3 MFMAs, VALU write right after", src, out, ref, got);
  run<true, 1, 0>("RAW: VALU write, MFMA right after", src, out, ref, got);
  run<true, 3, 0>("WAR: 1 MFMA, VALU write right after", src, out, ref, got);
  run<true, 4, 7>("WAR: 2 MFMAs (same operands), VALU write right after", src, out, ref, got);
  run<true, 6, 8>("WAR: 3 MFMAs, VALU write right after", src, out, ref, got);
  return 0;
}
