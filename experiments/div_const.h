// div_const.h — a / b, correctly rounded, for the two constant divisors of the tiling and up-sampling kernels
// (b = 2000: `/= SCALE`, testing/supres.py:23-24; b = 30000: interp_patches, utils/patches.py:15) without the ten-instruction
// IEEE division sequence: q = a * r with r = RN(1 / b), then Markstein's correction q' = fma(fma(-q, b, a), r, q).
// q' is the IEEE quotient for EVERY float a whose biased exponent lies in [5, 254] — checked exhaustively over all 2^32 bit
// patterns for both divisors (experiments/probe_div_const.hip, which includes this file: 0 of 2^32 differ);
// outside [32, 222] (tiny quotients that round in the subnormal range, infinities, NaNs) and for any other divisor the
// IEEE division itself is used, so the function equals `a / b` for every input by construction.  ±0 keeps its sign (q).
#pragma once
#include <hip/hip_runtime.h>

namespace dsen2 {

__host__ __device__ inline bool div_const_verified(float b) { return b == 2000.0f || b == 30000.0f; }

// the inputs for which the correction step is not proved: biased exponent < 32 or > 222, zero excepted (its q is exact)
__device__ __forceinline__ bool div_const_needs_division(float a) {
  const unsigned m = __builtin_bit_cast(unsigned, a) & 0x7fffffffu;
  return m - (32u << 23) >= (191u << 23) && m != 0u;
}

// r must be 1.0f / b (correctly rounded); b must satisfy div_const_verified(b); a must not need the division
__device__ __forceinline__ float div_const_unguarded(float a, float b, float r) {
  const float q = __fmul_rn(a, r);
  const float e = __builtin_fmaf(-q, b, a);
  const float c = __builtin_fmaf(e, r, q);
  return a == 0.0f ? q : c;                        // +-0 keeps its sign
}

__device__ __forceinline__ float div_const(float a, float b, float r) {
  if (__builtin_expect(div_const_needs_division(a), 0)) return __fdiv_rn(a, b);
  return div_const_unguarded(a, b, r);
}

}  // namespace dsen2
