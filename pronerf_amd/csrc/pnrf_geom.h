// pnrf_geom.h — per-ray geometry shared by the operator kernels (pnrf_ops.hip) and the fused MLP stages (pnrf_mlp_kernels.hip):
// unit direction, Pluecker moment, the neighbour projection and the bilinear tap set-up.  Arithmetic that has to match the reference's
// separate torch ops goes through the one-rounding helpers of pnrf_ieee.h (hipcc's FMA contraction would otherwise change roundings).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pnrf_ieee.h"

namespace pnrf {

// d / max(|d|, 1e-12)   (torch.nn.functional.normalize, run_nerf_helpers.py:630)
__device__ __forceinline__ void unit_dir(float dx, float dy, float dz, float& hx, float& hy, float& hz) {
  // torch's 2-norm over the last axis (CPU kernel behind F.normalize / torch.norm) accumulates with fused multiply-adds:
  // sqrt(fma(z, z, fma(y, y, x x))) — checked bit for bit against torch 2.10 on the 762 048 rays of the Fern frame
  const float n2 = ieee_fma(dz, dz, ieee_fma(dy, dy, ieee_mul(dx, dx)));
  const float den = fmaxf(ieee_sqrt(n2), 1e-12f);
  hx = ieee_div(dx, den); hy = ieee_div(dy, den); hz = ieee_div(dz, den);
}
// a x b   (torch.cross, run_nerf_helpers.py:631).  torch's CPU kernel evaluates each component as fma(a1, b2, -(a2 b1)) (one product rounded,
// the other fused into the subtraction) — checked bit for bit against torch 2.10
__device__ __forceinline__ void cross_rn(float ax, float ay, float az, float bx, float by, float bz, float& m0, float& m1, float& m2) {
  m0 = ieee_fma(ay, bz, -ieee_mul(az, by));
  m1 = ieee_fma(az, bx, -ieee_mul(ax, bz));
  m2 = ieee_fma(ax, by, -ieee_mul(ay, bx));
}

// Bilinear fetch set-up, zero padding, align_corners=True: grid_sample's coordinate round trip (inverse_warp.py:607-608, then torch's
// un-normalisation) replayed in fp32.  Non-finite coordinates give x0 = y0 = -4 (every tap outside).
__device__ __forceinline__ void bilinear_setup(float X, float Y, int Hf, int Wf, int& x0, int& y0, float& wx0, float& wx1, float& wy0, float& wy1, bool& finite) {
  const float xn = ieee_sub(ieee_div(ieee_mul(2.f, X), (float)(Wf - 1)), 1.f);
  const float yn = ieee_sub(ieee_div(ieee_mul(2.f, Y), (float)(Hf - 1)), 1.f);
  const float ix = ieee_mul(ieee_div(ieee_add(xn, 1.f), 2.f), (float)(Wf - 1));
  const float iy = ieee_mul(ieee_div(ieee_add(yn, 1.f), 2.f), (float)(Hf - 1));
  finite = isfinite(ix) && isfinite(iy) && fabsf(ix) < 1e9f && fabsf(iy) < 1e9f;
  const float fx = floorf(ix), fy = floorf(iy);
  wx1 = ieee_sub(ix, fx); wx0 = ieee_sub(ieee_add(fx, 1.f), ix);
  wy1 = ieee_sub(iy, fy); wy0 = ieee_sub(ieee_add(fy, 1.f), iy);
  x0 = finite ? (int)fx : -4; y0 = finite ? (int)fy : -4;
}

// One neighbour-view colour sample of the inference path (run_S_eS_eN_alter_trt.py:637-655; inverse_warp.py:584-619): NDC depth dn ->
// metric depth 1 / (1 - dn - eps) -> world point w = o + e z -> pixel (X, Y) = (M w)_{0,1} / (M w)_2 with the 3x4 matrix M -> the four
// bilinear taps as texel indices into a [Hf, Wf] float4 image and their weights.  A tap outside the image gets weight 0 and a clamped
// (valid) index: t * 0 = 0 and x + 0 = x exactly, so fetching unconditionally gives the bits of the reference's zero padding without a
// divergent branch per tap.
struct Taps { uint32_t i00, i01, i10, i11; float a00, a01, a10, a11; };      // texel indices (clamped into the image: never negative)
__device__ __forceinline__ Taps project_taps(float o0, float o1, float o2, float e0, float e1, float e2, float dn, float eps, const float (&M)[12], int Hf, int Wf) {
  const float z3d = ieee_div(1.f, ieee_sub(ieee_sub(1.f, dn), eps));                                                   // trt.py:637
  const float w0 = ieee_add(o0, ieee_mul(e0, z3d)), w1 = ieee_add(o1, ieee_mul(e1, z3d)), w2 = ieee_add(o2, ieee_mul(e2, z3d));   // inverse_warp.py:600
  float p[3];
#pragma unroll
  for (int r = 0; r < 3; ++r)                                                                                           // :601 (homogeneous 1)
    p[r] = ieee_add(ieee_add(ieee_add(ieee_mul(M[r * 4], w0), ieee_mul(M[r * 4 + 1], w1)), ieee_mul(M[r * 4 + 2], w2)), M[r * 4 + 3]);
  const float X = ieee_div(p[0], p[2]), Y = ieee_div(p[1], p[2]);                                                     // :603-605
  int x0, y0; float wx0, wx1, wy0, wy1; bool fin;
  bilinear_setup(X, Y, Hf, Wf, x0, y0, wx0, wx1, wy0, wy1, fin);
  const bool okx0 = x0 >= 0 && x0 < Wf, okx1 = x0 + 1 >= 0 && x0 + 1 < Wf, oky0 = y0 >= 0 && y0 < Hf, oky1 = y0 + 1 >= 0 && y0 + 1 < Hf;
  const int xa = min(max(x0, 0), Wf - 1), xb = min(max(x0 + 1, 0), Wf - 1), ya = min(max(y0, 0), Hf - 1), yb = min(max(y0 + 1, 0), Hf - 1);
  Taps t;
  t.i00 = ya * Wf + xa; t.i01 = ya * Wf + xb; t.i10 = yb * Wf + xa; t.i11 = yb * Wf + xb;
  t.a00 = (oky0 && okx0) ? ieee_mul(wx0, wy0) : 0.f; t.a01 = (oky0 && okx1) ? ieee_mul(wx1, wy0) : 0.f;
  t.a10 = (oky1 && okx0) ? ieee_mul(wx0, wy1) : 0.f; t.a11 = (oky1 && okx1) ? ieee_mul(wx1, wy1) : 0.f;
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
  return ieee_add(ieee_add(ieee_add(ieee_mul(t00, t.a00), ieee_mul(t01, t.a01)), ieee_mul(t10, t.a10)), ieee_mul(t11, t.a11));
}

}  // namespace pnrf
