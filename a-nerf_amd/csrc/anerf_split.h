// anerf_split.h -- fp32 -> (hi, lo) bf16 operand split shared by the split-bf16 kernels (anerf_mlp_b3.hip,
// anerf_gemm.hip): x = hi + lo with both halves rounded to nearest, so that
//     a * b  ~=  a_hi*b_hi + a_hi*b_lo + a_lo*b_hi      (three v_mfma_f32_32x32x16_bf16, fp32 accumulate; the dropped
//                                                        lo*lo term is ~2^-18 relative).
#pragma once
#include <hip/hip_runtime.h>

namespace anerf {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
typedef float f2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// (a, b) -> packed bf16 pairs hi, lo with a = hi.x + lo.x (+ 2^-18 rel.), round-to-nearest both times
__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) {
  const bf2_t h = __builtin_convertvector(f2_t{a, b}, bf2_t);
  const f2_t hf = __builtin_convertvector(h, f2_t);
  const bf2_t l = __builtin_convertvector(f2_t{a - hf.x, b - hf.y}, bf2_t);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}

struct BOp {
  bf16x8 hi, lo;
};

__device__ __forceinline__ BOp split8(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7) {
  unsigned h0, h1, h2, h3, l0, l1, l2, l3;
  split2(v0, v1, h0, l0);
  split2(v2, v3, h1, l1);
  split2(v4, v5, h2, l2);
  split2(v6, v7, h3, l3);
  const u32x4 h = {h0, h1, h2, h3}, l = {l0, l1, l2, l3};
  BOp o;
  o.hi = __builtin_bit_cast(bf16x8, h);
  o.lo = __builtin_bit_cast(bf16x8, l);
  return o;
}

}  // namespace anerf
