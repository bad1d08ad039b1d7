// What v_dot2c_f32_bf16 does with its operands on gfx950: inline constant 1.0 vs a literal, denormals, accumulate rounding.
// Asked because cs += (float)bf16 as dot2(pair, (1, 0), cs) would halve the vector-ALU work of the GEMM epilogue's column sums.
// Findings on MI355X (ROCm 7.2.0), which is why the epilogue does NOT use it:
//  * hipcc folds the packed constant 0x00003f80 = (lo 1.0, hi 0) into the INLINE constant 1.0, and the instruction reads that inline
//    constant as 0x3f800000 = (lo 0, hi 1.0): both constant forms below return the HIGH element.  From a register it is right.
//  * the accumulate is not a correctly rounded fp32 add (last line: 1 ulp below c + lo), and 0 x inf = NaN reaches the other column.
//   hipcc --offload-arch=gfx950 -O3 -o dot2_probe dot2_probe.hip && ./dot2_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__global__ void k(const unsigned* a, const float* c, float* o) {
  const int t = threadIdx.x;
  const bf16x2 v = __builtin_bit_cast(bf16x2, a[t]);
  const bf16x2 lo = __builtin_bit_cast(bf16x2, 0x00003f80u), hi = __builtin_bit_cast(bf16x2, 0x3f800000u);
  unsigned lo_bits = 0x00003f80u;
  asm volatile("" : "+v"(lo_bits));                       // the same constant from a register
  o[4 * t + 0] = __builtin_amdgcn_fdot2_f32_bf16(v, lo, c[t], false);
  o[4 * t + 1] = __builtin_amdgcn_fdot2_f32_bf16(v, hi, c[t], false);
  o[4 * t + 2] = __builtin_amdgcn_fdot2_f32_bf16(v, __builtin_bit_cast(bf16x2, lo_bits), c[t], false);
  o[4 * t + 3] = c[t] + __builtin_bit_cast(float, a[t] << 16);
}
int main() {
  const unsigned ha[8] = {0x40003f80u /* (1, 2) */, 0xc0404000u /* (2, -3) */, 0x00013f80u /* hi = denormal */, 0x3f800001u /* lo = denormal */,
                          0x3f803f81u, 0x7f803f80u /* hi = inf */, 0x3f80ffc0u /* lo = nan */, 0x3dcd3e4du};
  const float hc[8] = {0.f, 10.f, 0.f, 0.f, 16777216.f, 1.f, 1.f, 1e-3f};
  unsigned* a; float *c, *o;
  hipMalloc(&a, 32); hipMalloc(&c, 32); hipMalloc(&o, 128);
  hipMemcpy(a, ha, 32, hipMemcpyHostToDevice); hipMemcpy(c, hc, 32, hipMemcpyHostToDevice);
  k<<<1, 8>>>(a, c, o);
  float ho[32];
  hipMemcpy(ho, o, 128, hipMemcpyDeviceToHost);
  for (int t = 0; t < 8; ++t) {
    unsigned b[4]; memcpy(b, ho + 4 * t, 16);
    printf("pair %08x c %g:  dot(1,0) %.9g [%08x]  dot(0,1) %.9g [%08x]  dot(1,0 from a register) %.9g [%08x]  c + lo %.9g [%08x]\n", ha[t], hc[t], ho[4 * t], b[0],
           ho[4 * t + 1], b[1], ho[4 * t + 2], b[2], ho[4 * t + 3], b[3]);
  }
  return 0;
}
