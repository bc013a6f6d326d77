// pnrf_ieee.h — fp32 operations with exactly one IEEE rounding each, for the arithmetic that has to reproduce torch's separate elementwise
// ops bit for bit (ray set-up, Pluecker encoding, the depth affine that feeds the sort, compositing sums).
//
// hipcc compiles device code with -ffp-contract=fast: `a * b + c` becomes one v_fma_f32 / v_fmac_f32 — and so does
// `__fadd_rn(__fmul_rn(a, b), c)`: without OCML_BASIC_ROUNDED_OPERATIONS (whose __ocml_*_rte_f32 entry points this ROCm does not ship for
// sqrt and division) <__clang_hip_math.h> defines __fmul_rn(x, y) as `x * y`, __fadd_rn as `x + y` and __fsqrt_rn as the 1-ulp
// v_sqrt_f32.  The helpers below switch contraction off inside their own bodies (the resulting fmul / fadd carry no `contract` flag, so
// they stay separate after inlining) and use the correctly rounded square root and division (hipcc's default
// -fhip-fp32-correctly-rounded-divide-sqrt: v_sqrt_f32 / v_rcp_f32 + refinement + fix-up).  ieee_fma is an explicit fused multiply-add.
#pragma once
#include <hip/hip_runtime.h>

namespace pnrf {

__device__ __forceinline__ float ieee_mul(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float ieee_add(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ float ieee_sub(float a, float b) {
#pragma clang fp contract(off)
  return a - b;
}
__device__ __forceinline__ float ieee_div(float a, float b) {
#pragma clang fp contract(off)
  return a / b;
}
__device__ __forceinline__ float ieee_sqrt(float a) { return __builtin_sqrtf(a); }
__device__ __forceinline__ float ieee_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

}  // namespace pnrf
