// pnrf_geom.h — per-ray geometry shared by the operator kernels (pnrf_ops.hip) and the fused MLP stages (pnrf_mlp_kernels.hip):
// unit direction, Pluecker moment, the neighbour projection and the bilinear tap set-up.  Arithmetic is written with explicit
// round-to-nearest intrinsics so that hipcc's FMA contraction cannot change roundings relative to the reference's separate torch ops.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pnrf {

// d / max(|d|, 1e-12)   (torch.nn.functional.normalize, run_nerf_helpers.py:630)
__device__ __forceinline__ void unit_dir(float dx, float dy, float dz, float& hx, float& hy, float& hz) {
  const float n2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
  const float den = fmaxf(__fsqrt_rn(n2), 1e-12f);
  hx = __fdiv_rn(dx, den); hy = __fdiv_rn(dy, den); hz = __fdiv_rn(dz, den);
}
// a x b   (torch.cross, run_nerf_helpers.py:631)
__device__ __forceinline__ void cross_rn(float ax, float ay, float az, float bx, float by, float bz, float& m0, float& m1, float& m2) {
  m0 = __fsub_rn(__fmul_rn(ay, bz), __fmul_rn(az, by));
  m1 = __fsub_rn(__fmul_rn(az, bx), __fmul_rn(ax, bz));
  m2 = __fsub_rn(__fmul_rn(ax, by), __fmul_rn(ay, bx));
}

// Bilinear fetch set-up, zero padding, align_corners=True: grid_sample's coordinate round trip (inverse_warp.py:607-608, then torch's
// un-normalisation) replayed in fp32.  Non-finite coordinates give x0 = y0 = -4 (every tap outside).
__device__ __forceinline__ void bilinear_setup(float X, float Y, int Hf, int Wf, int& x0, int& y0, float& wx0, float& wx1, float& wy0, float& wy1, bool& finite) {
  const float xn = __fsub_rn(__fdiv_rn(__fmul_rn(2.f, X), (float)(Wf - 1)), 1.f);
  const float yn = __fsub_rn(__fdiv_rn(__fmul_rn(2.f, Y), (float)(Hf - 1)), 1.f);
  const float ix = __fmul_rn(__fdiv_rn(__fadd_rn(xn, 1.f), 2.f), (float)(Wf - 1));
  const float iy = __fmul_rn(__fdiv_rn(__fadd_rn(yn, 1.f), 2.f), (float)(Hf - 1));
  finite = isfinite(ix) && isfinite(iy) && fabsf(ix) < 1e9f && fabsf(iy) < 1e9f;
  const float fx = floorf(ix), fy = floorf(iy);
  wx1 = __fsub_rn(ix, fx); wx0 = __fsub_rn(__fadd_rn(fx, 1.f), ix);
  wy1 = __fsub_rn(iy, fy); wy0 = __fsub_rn(__fadd_rn(fy, 1.f), iy);
  x0 = finite ? (int)fx : -4; y0 = finite ? (int)fy : -4;
}

// One neighbour-view colour sample of the inference path (run_S_eS_eN_alter_trt.py:637-655; inverse_warp.py:584-619): NDC depth dn ->
// metric depth 1 / (1 - dn - eps) -> world point w = o + e z -> pixel (X, Y) = (M w)_{0,1} / (M w)_2 with the 3x4 matrix M -> the four
// bilinear taps as texel indices into a [Hf, Wf] float4 image and their weights.  A tap outside the image gets weight 0 and a clamped
// (valid) index: t * 0 = 0 and x + 0 = x exactly, so fetching unconditionally gives the bits of the reference's zero padding without a
// divergent branch per tap.
struct Taps { uint32_t i00, i01, i10, i11; float a00, a01, a10, a11; };      // texel indices (clamped into the image: never negative)
__device__ __forceinline__ Taps project_taps(float o0, float o1, float o2, float e0, float e1, float e2, float dn, float eps, const float (&M)[12], int Hf, int Wf) {
  const float z3d = __fdiv_rn(1.f, __fsub_rn(__fsub_rn(1.f, dn), eps));                                                   // trt.py:637
  const float w0 = __fadd_rn(o0, __fmul_rn(e0, z3d)), w1 = __fadd_rn(o1, __fmul_rn(e1, z3d)), w2 = __fadd_rn(o2, __fmul_rn(e2, z3d));   // inverse_warp.py:600
  float p[3];
#pragma unroll
  for (int r = 0; r < 3; ++r)                                                                                           // :601 (homogeneous 1)
    p[r] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(M[r * 4], w0), __fmul_rn(M[r * 4 + 1], w1)), __fmul_rn(M[r * 4 + 2], w2)), M[r * 4 + 3]);
  const float X = __fdiv_rn(p[0], p[2]), Y = __fdiv_rn(p[1], p[2]);                                                     // :603-605
  int x0, y0; float wx0, wx1, wy0, wy1; bool fin;
  bilinear_setup(X, Y, Hf, Wf, x0, y0, wx0, wx1, wy0, wy1, fin);
  const bool okx0 = x0 >= 0 && x0 < Wf, okx1 = x0 + 1 >= 0 && x0 + 1 < Wf, oky0 = y0 >= 0 && y0 < Hf, oky1 = y0 + 1 >= 0 && y0 + 1 < Hf;
  const int xa = min(max(x0, 0), Wf - 1), xb = min(max(x0 + 1, 0), Wf - 1), ya = min(max(y0, 0), Hf - 1), yb = min(max(y0 + 1, 0), Hf - 1);
  Taps t;
  t.i00 = ya * Wf + xa; t.i01 = ya * Wf + xb; t.i10 = yb * Wf + xa; t.i11 = yb * Wf + xb;
  t.a00 = (oky0 && okx0) ? __fmul_rn(wx0, wy0) : 0.f; t.a01 = (oky0 && okx1) ? __fmul_rn(wx1, wy0) : 0.f;
  t.a10 = (oky1 && okx0) ? __fmul_rn(wx0, wy1) : 0.f; t.a11 = (oky1 && okx1) ? __fmul_rn(wx1, wy1) : 0.f;
  return t;
}
// The same projection for the fused refine head, where the 96 colours of a ray go straight into bf16 MFMA operands (8 significant bits):
// p(z) = M (o + e z) is linear in the metric depth, so per (ray, view) A = M[:, :3] o + M[:, 3] and B = M[:, :3] e are formed once and a
// sample costs three FMAs, one v_rcp_f32 and two multiplies; grid_sample's normalise / un-normalise round trip is the identity and is
// dropped.  Against project_taps this moves the pixel coordinate by a few fp32 ulps (~1e-4 px at X ~ 1e3), i.e. a colour by ~1e-4 of the
// local texel difference — 20x below the bf16 rounding applied to it next.  (The operator pnrf_refine_input_fwd keeps the exact sequence.)
struct ViewRay { float A0, A1, A2, B0, B1, B2; };
__device__ __forceinline__ ViewRay view_ray(float o0, float o1, float o2, float e0, float e1, float e2, const float (&M)[12]) {
  ViewRay v;
  v.A0 = fmaf(M[0], o0, fmaf(M[1], o1, fmaf(M[2], o2, M[3])));  v.B0 = fmaf(M[0], e0, fmaf(M[1], e1, M[2] * e2));
  v.A1 = fmaf(M[4], o0, fmaf(M[5], o1, fmaf(M[6], o2, M[7])));  v.B1 = fmaf(M[4], e0, fmaf(M[5], e1, M[6] * e2));
  v.A2 = fmaf(M[8], o0, fmaf(M[9], o1, fmaf(M[10], o2, M[11]))); v.B2 = fmaf(M[8], e0, fmaf(M[9], e1, M[10] * e2));
  return v;
}
__device__ __forceinline__ Taps project_taps_fast(const ViewRay& v, float z3d, int Hf, int Wf) {
  const float p2 = fmaf(z3d, v.B2, v.A2);
  const float r = __builtin_amdgcn_rcpf(p2);
  const float X = fmaf(z3d, v.B0, v.A0) * r, Y = fmaf(z3d, v.B1, v.A1) * r;
  const bool fin = fabsf(X) < 1e9f && fabsf(Y) < 1e9f;                 // false for NaN / inf as well (p2 = 0)
  const float fx = floorf(X), fy = floorf(Y);
  const float wx1 = X - fx, wy1 = Y - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
  const int x0 = fin ? (int)fx : -4, y0 = fin ? (int)fy : -4;
  const bool okx0 = x0 >= 0 && x0 < Wf, okx1 = x0 + 1 >= 0 && x0 + 1 < Wf, oky0 = y0 >= 0 && y0 < Hf, oky1 = y0 + 1 >= 0 && y0 + 1 < Hf;
  const int xa = min(max(x0, 0), Wf - 1), xb = min(max(x0 + 1, 0), Wf - 1), ya = min(max(y0, 0), Hf - 1), yb = min(max(y0 + 1, 0), Hf - 1);
  Taps t;
  t.i00 = ya * Wf + xa; t.i01 = ya * Wf + xb; t.i10 = yb * Wf + xa; t.i11 = yb * Wf + xb;
  t.a00 = (oky0 && okx0) ? wx0 * wy0 : 0.f; t.a01 = (oky0 && okx1) ? wx1 * wy0 : 0.f;
  t.a10 = (oky1 && okx0) ? wx0 * wy1 : 0.f; t.a11 = (oky1 && okx1) ? wx1 * wy1 : 0.f;
  return t;
}
__device__ __forceinline__ float blend4(float t00, float t01, float t10, float t11, const Taps& t) {
  return __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(t00, t.a00), __fmul_rn(t01, t.a01)), __fmul_rn(t10, t.a10)), __fmul_rn(t11, t.a11));
}

}  // namespace pnrf
