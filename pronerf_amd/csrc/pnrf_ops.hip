// pnrf_ops.hip — element-wise / gather operators of the render path, the render context and
// the whole-path entry point.  All kernels are HBM/L2-bound streaming kernels: one thread per
// output element group, coalesced 16-byte accesses where the layout allows, grid-stride loops.
#include "pnrf_common.h"
#include "pnrf_geom.h"

using namespace pnrf;

namespace {

constexpr int TPB = 256;
inline int grid_for(int64_t work, int per_block = TPB) {
  int64_t g = (work + per_block - 1) / per_block;
  const int64_t cap = 256 * 16;     // 256 CUs x 16 blocks: enough to fill the chip, grid-stride the rest
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

// ---------------------------------------------------------------- positional encoding (helpers:666-671)
__global__ void posenc_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t n3, int n_freq) {
  // one thread per input scalar; output row width = 3 + 6*n_freq
  const int width = 3 + 6 * n_freq;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n3; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / 3;
    const int c = (int)(i - row * 3);
    const float v = x[i];
    float* o = out + row * width;
    o[c] = v;
    for (int k = 0; k < n_freq; ++k) {
      const float arg = v * (float)(1u << k);        // x * 2^k is exact in fp32
      o[3 + 6 * k + c] = sinf(arg);
      o[3 + 6 * k + 3 + c] = cosf(arg);
    }
  }
}

// ---------------------------------------------------------------- Pluecker (helpers:629-632)
__global__ void plucker_kernel(const float* __restrict__ o, const float* __restrict__ d, float* __restrict__ out, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float hx, hy, hz, m0, m1, m2;
    unit_dir(d[i * 3], d[i * 3 + 1], d[i * 3 + 2], hx, hy, hz);
    cross_rn(o[i * 3], o[i * 3 + 1], o[i * 3 + 2], hx, hy, hz, m0, m1, m2);
    float* q = out + i * 6;
    q[0] = hx; q[1] = hy; q[2] = hz; q[3] = m0; q[4] = m1; q[5] = m2;
  }
}
// mm_input[n, 6*n_pts]: one thread per (ray, point)  (trt.py:274-277)
__global__ void ray_encode_kernel(const float* __restrict__ rays, const float* __restrict__ tvals, float* __restrict__ out, int64_t n, int n_pts) {
  const int64_t total = n * n_pts;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t ray = i / n_pts;
    const int p = (int)(i - ray * n_pts);
    const float* r = rays + ray * 11;
    const float t = tvals[p];
    float hx, hy, hz, m0, m1, m2;
    unit_dir(r[3], r[4], r[5], hx, hy, hz);
    const float px = ieee_add(r[0], ieee_mul(r[3], t)), py = ieee_add(r[1], ieee_mul(r[4], t)), pz = ieee_add(r[2], ieee_mul(r[5], t));
    cross_rn(px, py, pz, hx, hy, hz, m0, m1, m2);
    float* q = out + i * 6;        // [ray][p][6] == ray*6*n_pts + p*6
    q[0] = hx; q[1] = hy; q[2] = hz; q[3] = m0; q[4] = m1; q[5] = m2;
  }
}

// ---------------------------------------------------------------- frame rays (trt.py:245-271; helpers:2705-2714, 2776-2793)
// The NDC scale factors follow the reference drivers' types: K is a float64 numpy array there (run_S_eS_eN_alter_trt.py:742-747, :798), so
// -1./(W/(2.*focal)) is evaluated in double and rounded to fp32 once, when it meets the fp32 ray tensor (helpers:2781-2786).
static inline float ndc_scale(int extent, float focal) { return (float)(-1.0 / ((double)extent / (2.0 * (double)focal))); }

struct FrameArgs {
  float K00, K02, K11, K12;
  float sx, sy;              // ndc_scale(W, K00), ndc_scale(H, K00)
  float R[9], T[3];
  int H, W;
  float near, far, or_near, or_far;
  int64_t first, count;
  int64_t block, stride;     // output row q is pixel first + (q / block) * stride + q % block (one contiguous range: block = count; a rank's blocks of a
                             // block-cyclic partition: first = rank * block, stride = world * block)
};
__global__ void frame_rays_kernel(FrameArgs a, float* __restrict__ rays, float* __restrict__ or_rays) {
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < a.count; q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pix = a.first + (q / a.block) * a.stride + q % a.block;
    const int j = (int)(pix / a.W), i = (int)(pix - (int64_t)j * a.W);
    // dirs = ((i-cx)/fx, -(j-cy)/fy, -1);  rays_d[c] = sum_k dirs[k]*R[c][k]  (products, then a 3-term sum)
    const float d0 = ieee_div(ieee_sub((float)i, a.K02), a.K00);
    const float d1 = -ieee_div(ieee_sub((float)j, a.K12), a.K11);
    const float d2 = -1.f;
    float rd[3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
      rd[c] = ieee_add(ieee_add(ieee_mul(d0, a.R[c * 3]), ieee_mul(d1, a.R[c * 3 + 1])), ieee_mul(d2, a.R[c * 3 + 2]));
    const float ro[3] = {a.T[0], a.T[1], a.T[2]};
    const float nrm = ieee_sqrt(ieee_fma(rd[2], rd[2], ieee_fma(rd[1], rd[1], ieee_mul(rd[0], rd[0]))));      // torch.norm: an FMA chain (pnrf_geom.h, unit_dir)
    const float v0 = ieee_div(rd[0], nrm), v1 = ieee_div(rd[1], nrm), v2 = ieee_div(rd[2], nrm);
    float* orr = or_rays + q * 11;
    orr[0] = ro[0]; orr[1] = ro[1]; orr[2] = ro[2]; orr[3] = rd[0]; orr[4] = rd[1]; orr[5] = rd[2];
    orr[6] = a.or_near; orr[7] = a.or_far; orr[8] = v0; orr[9] = v1; orr[10] = v2;
    // ndc_rays(H, W, focal=K00, near=1.)
    const float nearp = 1.f;
    const float t = ieee_div(-ieee_add(nearp, ro[2]), rd[2]);
    const float ox = ieee_add(ro[0], ieee_mul(t, rd[0])), oy = ieee_add(ro[1], ieee_mul(t, rd[1])), oz = ieee_add(ro[2], ieee_mul(t, rd[2]));
    const float sx = a.sx, sy = a.sy;
    const float o0 = ieee_div(ieee_mul(sx, ox), oz);
    const float o1 = ieee_div(ieee_mul(sy, oy), oz);
    const float roz = ieee_div(1.f, oz);                               // python scalar / tensor is tensor.reciprocal() * scalar in torch
    const float o2 = ieee_add(1.f, ieee_mul(roz, 2.f * nearp));
    const float e0 = ieee_mul(sx, ieee_sub(ieee_div(rd[0], rd[2]), ieee_div(ox, oz)));
    const float e1 = ieee_mul(sy, ieee_sub(ieee_div(rd[1], rd[2]), ieee_div(oy, oz)));
    const float e2 = ieee_mul(roz, -2.f * nearp);
    float* r = rays + q * 11;
    r[0] = o0; r[1] = o1; r[2] = o2; r[3] = e0; r[4] = e1; r[5] = e2; r[6] = a.near; r[7] = a.far; r[8] = v0; r[9] = v1; r[10] = v2;
  }
}

// ndc_rays on arbitrary ray sets (helpers:2776-2793)
__global__ void ndc_rays_kernel(const float* __restrict__ o, const float* __restrict__ d, float sx, float sy, float nearp, float two_near,
                                float* __restrict__ oo, float* __restrict__ od, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float r0 = o[i * 3], r1 = o[i * 3 + 1], r2 = o[i * 3 + 2], d0 = d[i * 3], d1 = d[i * 3 + 1], d2 = d[i * 3 + 2];
    const float t = ieee_div(-ieee_add(nearp, r2), d2);
    const float ox = ieee_add(r0, ieee_mul(t, d0)), oy = ieee_add(r1, ieee_mul(t, d1)), oz = ieee_add(r2, ieee_mul(t, d2));
    const float roz = ieee_div(1.f, oz);                               // 2.*near / tensor = tensor.reciprocal() * (2.*near)
    oo[i * 3] = ieee_div(ieee_mul(sx, ox), oz);
    oo[i * 3 + 1] = ieee_div(ieee_mul(sy, oy), oz);
    oo[i * 3 + 2] = ieee_add(1.f, ieee_mul(roz, two_near));
    od[i * 3] = ieee_mul(sx, ieee_sub(ieee_div(d0, d2), ieee_div(ox, oz)));
    od[i * 3 + 1] = ieee_mul(sy, ieee_sub(ieee_div(d1, d2), ieee_div(oy, oz)));
    od[i * 3 + 2] = ieee_mul(roz, -two_near);
  }
}

// ---------------------------------------------------------------- warps (bilinear fetch, zero padding, align_corners=True: pnrf_geom.h)
// inverse_warp_rod1_rt2_coords_trt on planar images: one thread per (b, pixel)  (inverse_warp.py:584-619)
__global__ void warp_trt_kernel(const float* __restrict__ img, const float* __restrict__ depth, const float* __restrict__ ro1,
                                const float* __restrict__ rd1, int64_t ray_bstride, const float* __restrict__ w2c,
                                float* __restrict__ out, int B, int Hf, int Wf, int64_t n) {
  const int64_t total = (int64_t)B * n;
  const int64_t plane = (int64_t)Hf * Wf;
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(q / n);
    const int64_t i = q - (int64_t)b * n;
    const float dep = depth[q];
    float w[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) w[c] = ieee_add(ro1[b * ray_bstride + c * n + i], ieee_mul(rd1[b * ray_bstride + c * n + i], dep));   // :600
    const float* M = w2c + b * 12;
    float p[3];
#pragma unroll
    for (int r = 0; r < 3; ++r)                                                                     // :601 bmm, K=4
      p[r] = ieee_add(ieee_add(ieee_add(ieee_mul(M[r * 4], w[0]), ieee_mul(M[r * 4 + 1], w[1])), ieee_mul(M[r * 4 + 2], w[2])), ieee_mul(M[r * 4 + 3], w[3]));
    const float X = ieee_div(p[0], p[2]), Y = ieee_div(p[1], p[2]);                               // :603-605
    int x0, y0; float wx0, wx1, wy0, wy1; bool fin;
    bilinear_setup(X, Y, Hf, Wf, x0, y0, wx0, wx1, wy0, wy1, fin);
    const bool okx0 = x0 >= 0 && x0 < Wf, okx1 = x0 + 1 >= 0 && x0 + 1 < Wf, oky0 = y0 >= 0 && y0 < Hf, oky1 = y0 + 1 >= 0 && y0 + 1 < Hf;
    const float* im = img + (int64_t)b * 3 * plane;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* pc = im + c * plane;
      float acc = 0.f;
      if (oky0 && okx0) acc = ieee_add(acc, ieee_mul(pc[(int64_t)y0 * Wf + x0], ieee_mul(wx0, wy0)));
      if (oky0 && okx1) acc = ieee_add(acc, ieee_mul(pc[(int64_t)y0 * Wf + x0 + 1], ieee_mul(wx1, wy0)));
      if (oky1 && okx0) acc = ieee_add(acc, ieee_mul(pc[(int64_t)(y0 + 1) * Wf + x0], ieee_mul(wx0, wy1)));
      if (oky1 && okx1) acc = ieee_add(acc, ieee_mul(pc[(int64_t)(y0 + 1) * Wf + x0 + 1], ieee_mul(wx1, wy1)));
      out[((int64_t)b * 3 + c) * n + i] = acc;
    }
  }
}

// [nv,3,Hf,Wf] -> [nv,Hf,Wf,4] texel-interleaved (one 16-byte load per tap in the fused projection)
__global__ void images_pack_kernel(const float* __restrict__ in, float4* __restrict__ out, int nv, int64_t plane) {
  const int64_t total = (int64_t)nv * plane;
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
    const int v = (int)(q / plane);
    const int64_t i = q - (int64_t)v * plane;
    const float* p = in + (int64_t)v * 3 * plane + i;
    out[q] = make_float4(p[0], p[plane], p[2 * plane], 0.f);
  }
}

// Projection + sample Pluecker -> refine_in[n, 48 + 24 nb]   (trt.py:637-661; nb = 4: [n,144])
// Workgroup = 64 consecutive rays x 4 waves: lane = ray, wave = neighbour view k, k + 4 (nb <= 8), loop over the 8 samples.  Consecutive rays
// are consecutive pixels, so for a fixed (view, sample) the 64 lanes of a wave fetch adjacent texels (coalesced 16-byte taps, each
// cache line fetched once) — with one thread per (ray, view, sample) in ray-major order a wave touched 256 unrelated lines and
// the L2 hit rate was 63 %.  The [64 rays][144] tile is transposed through LDS and written out as one contiguous 36 KiB block.
constexpr int RI_TILE = 64;
__global__ __launch_bounds__(256) void refine_input_kernel(const float* __restrict__ rays, const float* __restrict__ or_rays, const float* __restrict__ depth_sorted,
                                    const float4* __restrict__ img4, const float* __restrict__ proj, int nb, int Hf, int Wf, float eps,
                                    float* __restrict__ out, int64_t n) {
  __shared__ float sM[8 * 12];
  __shared__ float sT[(48 + 24 * 8) * (RI_TILE + 1)];          // [feature][ray] (+1: conflict-free column writes)
  if ((int)threadIdx.x < 12 * nb) sM[threadIdx.x] = proj[threadIdx.x];
  const int F = 48 + 24 * nb;
  const int lane = threadIdx.x & 63, kw = threadIdx.x >> 6;
  const int64_t plane = (int64_t)Hf * Wf;
  const int64_t ntiles = (n + RI_TILE - 1) / RI_TILE;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    __syncthreads();                                  // sM ready / previous tile's sT fully written out
    const int64_t ray0 = tile * RI_TILE;
    const int64_t ray = ray0 + lane < n ? ray0 + lane : n - 1;
    const float* orr = or_rays + ray * 11;
    const float o0 = orr[0], o1 = orr[1], o2 = orr[2], e0 = orr[3], e1 = orr[4], e2 = orr[5];
    const float4 da = *(const float4*)(depth_sorted + ray * 8), db = *(const float4*)(depth_sorted + ray * 8 + 4);
    const float dn8[8] = {da.x, da.y, da.z, da.w, db.x, db.y, db.z, db.w};
    for (int k = kw; k < nb; k += 4) {
    const float* M = sM + k * 12;
    const float4* im = img4 + (int64_t)k * plane;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const float z3d = ieee_div(1.f, ieee_sub(ieee_sub(1.f, dn8[s]), eps));                 // trt.py:637
      const float w0 = ieee_add(o0, ieee_mul(e0, z3d)), w1 = ieee_add(o1, ieee_mul(e1, z3d)), w2 = ieee_add(o2, ieee_mul(e2, z3d));   // inverse_warp.py:600
      float p[3];
#pragma unroll
      for (int r = 0; r < 3; ++r)
        p[r] = ieee_add(ieee_add(ieee_add(ieee_mul(M[r * 4], w0), ieee_mul(M[r * 4 + 1], w1)), ieee_mul(M[r * 4 + 2], w2)), M[r * 4 + 3]);
      const float X = ieee_div(p[0], p[2]), Y = ieee_div(p[1], p[2]);
      int x0, y0; float wx0, wx1, wy0, wy1; bool fin;
      bilinear_setup(X, Y, Hf, Wf, x0, y0, wx0, wx1, wy0, wy1, fin);
      const bool okx0 = x0 >= 0 && x0 < Wf, okx1 = x0 + 1 >= 0 && x0 + 1 < Wf, oky0 = y0 >= 0 && y0 < Hf, oky1 = y0 + 1 >= 0 && y0 + 1 < Hf;
      const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 t00 = (oky0 && okx0) ? im[(int64_t)y0 * Wf + x0] : zero;
      const float4 t01 = (oky0 && okx1) ? im[(int64_t)y0 * Wf + x0 + 1] : zero;
      const float4 t10 = (oky1 && okx0) ? im[(int64_t)(y0 + 1) * Wf + x0] : zero;
      const float4 t11 = (oky1 && okx1) ? im[(int64_t)(y0 + 1) * Wf + x0 + 1] : zero;
      const float a00 = ieee_mul(wx0, wy0), a01 = ieee_mul(wx1, wy0), a10 = ieee_mul(wx0, wy1), a11 = ieee_mul(wx1, wy1);
      const int f = 48 + (k * 8 + s) * 3;                                                        // epi index (k*8+s)*3+c
      sT[(f + 0) * (RI_TILE + 1) + lane] = ieee_add(ieee_add(ieee_add(ieee_mul(t00.x, a00), ieee_mul(t01.x, a01)), ieee_mul(t10.x, a10)), ieee_mul(t11.x, a11));
      sT[(f + 1) * (RI_TILE + 1) + lane] = ieee_add(ieee_add(ieee_add(ieee_mul(t00.y, a00), ieee_mul(t01.y, a01)), ieee_mul(t10.y, a10)), ieee_mul(t11.y, a11));
      sT[(f + 2) * (RI_TILE + 1) + lane] = ieee_add(ieee_add(ieee_add(ieee_mul(t00.z, a00), ieee_mul(t01.z, a01)), ieee_mul(t10.z, a10)), ieee_mul(t11.z, a11));
    }
    }
    {                                                                                            // trt.py:656-658: wave k encodes samples 2k, 2k+1
      const int k = kw;
      const float* r = rays + ray * 11;
      float hx, hy, hz;
      unit_dir(r[3], r[4], r[5], hx, hy, hz);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int s = 2 * k + u;
        const float dn = s < 4 ? (s == 0 ? da.x : s == 1 ? da.y : s == 2 ? da.z : da.w) : (s == 4 ? db.x : s == 5 ? db.y : s == 6 ? db.z : db.w);
        float m0, m1, m2;
        const float px = ieee_add(r[0], ieee_mul(r[3], dn)), py = ieee_add(r[1], ieee_mul(r[4], dn)), pz = ieee_add(r[2], ieee_mul(r[5], dn));
        cross_rn(px, py, pz, hx, hy, hz, m0, m1, m2);
        float* q = sT + (s * 6) * (RI_TILE + 1) + lane;
        q[0] = hx; q[RI_TILE + 1] = hy; q[2 * (RI_TILE + 1)] = hz; q[3 * (RI_TILE + 1)] = m0; q[4 * (RI_TILE + 1)] = m1; q[5 * (RI_TILE + 1)] = m2;
      }
    }
    __syncthreads();
    const int64_t nvalid = (n - ray0 < RI_TILE ? n - ray0 : RI_TILE) * F;                         // contiguous block of the output
    float* dst = out + ray0 * F;
    for (int e = threadIdx.x; e < RI_TILE * F; e += 256)
      if (e < nvalid) dst[e] = sT[(e % F) * (RI_TILE + 1) + e / F];
  }
}

// ---------------------------------------------------------------- training-variant projection
// inverse_warp_rod1_rt2_coords (inverse_warp.py:515-581): c2 = R^T w - R^T t; c2 /= |c2.z| + 1e-8; c2.z = 1; c2.y = -c2.y;
// p = K c2; a sample whose normalised X or Y leaves [-1,1] yields 0; otherwise bilinear, zero padding, align_corners.
__device__ __forceinline__ bool project_train(const float* __restrict__ pose /*3x4 c2w*/, const float* __restrict__ K /*3x3*/,
                                              float w0, float w1, float w2, int Hf, int Wf, float& X, float& Y) {
  // R2_ = R^T; t2_ = -(R^T t)  (bmm, K = 3)
  float tt[3], c2[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float r0 = pose[0 * 4 + i], r1 = pose[1 * 4 + i], r2 = pose[2 * 4 + i];       // row i of R^T = column i of R
    tt[i] = -ieee_add(ieee_add(ieee_mul(r0, pose[3]), ieee_mul(r1, pose[7])), ieee_mul(r2, pose[11]));
    c2[i] = ieee_add(ieee_add(ieee_add(ieee_mul(r0, w0), ieee_mul(r1, w1)), ieee_mul(r2, w2)), tt[i]);
  }
  const float z = ieee_add(fabsf(c2[2]), 1e-8f);
  const float cx = ieee_div(c2[0], z), cy = -ieee_div(c2[1], z);
  X = ieee_add(ieee_add(ieee_mul(K[0], cx), ieee_mul(K[1], cy)), K[2]);
  Y = ieee_add(ieee_add(ieee_mul(K[3], cx), ieee_mul(K[4], cy)), K[5]);
  const float xn = ieee_sub(ieee_div(ieee_mul(2.f, X), (float)(Wf - 1)), 1.f);
  const float yn = ieee_sub(ieee_div(ieee_mul(2.f, Y), (float)(Hf - 1)), 1.f);
  return xn <= 1.f && xn >= -1.f && yn <= 1.f && yn >= -1.f;
}

// stand-alone operator on planar images: one thread per (b, pixel)
__global__ void warp_train_kernel(const float* __restrict__ img, const float* __restrict__ depth, const float* __restrict__ ro1,
                                  const float* __restrict__ rd1, int64_t ray_bstride, const float* __restrict__ c2w2,
                                  const float* __restrict__ Kmat, float* __restrict__ out, int B, int Hf, int Wf, int64_t n) {
  const int64_t total = (int64_t)B * n;
  const int64_t plane = (int64_t)Hf * Wf;
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(q / n);
    const int64_t i = q - (int64_t)b * n;
    const float dep = depth[q];
    const float* ro = ro1 + b * ray_bstride;
    const float* rd = rd1 + b * ray_bstride;
    const float w0 = ieee_add(ro[i], ieee_mul(rd[i], dep)), w1 = ieee_add(ro[n + i], ieee_mul(rd[n + i], dep)),
                w2 = ieee_add(ro[2 * n + i], ieee_mul(rd[2 * n + i], dep));
    float X, Y;
    const bool inside = project_train(c2w2 + b * 12, Kmat + b * 9, w0, w1, w2, Hf, Wf, X, Y);
    int x0, y0; float wx0, wx1, wy0, wy1; bool fin;
    bilinear_setup(X, Y, Hf, Wf, x0, y0, wx0, wx1, wy0, wy1, fin);
    const bool ok = inside && fin;
    const bool okx0 = ok && x0 >= 0 && x0 < Wf, okx1 = ok && x0 + 1 >= 0 && x0 + 1 < Wf, oky0 = y0 >= 0 && y0 < Hf, oky1 = y0 + 1 >= 0 && y0 + 1 < Hf;
    const float* im = img + (int64_t)b * 3 * plane;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* pc = im + c * plane;
      float acc = 0.f;
      if (oky0 && okx0) acc = ieee_add(acc, ieee_mul(pc[(int64_t)y0 * Wf + x0], ieee_mul(wx0, wy0)));
      if (oky0 && okx1) acc = ieee_add(acc, ieee_mul(pc[(int64_t)y0 * Wf + x0 + 1], ieee_mul(wx1, wy0)));
      if (oky1 && okx0) acc = ieee_add(acc, ieee_mul(pc[(int64_t)(y0 + 1) * Wf + x0], ieee_mul(wx0, wy1)));
      if (oky1 && okx1) acc = ieee_add(acc, ieee_mul(pc[(int64_t)(y0 + 1) * Wf + x0 + 1], ieee_mul(wx1, wy1)));
      out[((int64_t)b * 3 + c) * n + i] = acc;
    }
  }
}

// Training refine_in: per-ray source views ref_nos[n,4], training projection, valid-mask mean fill
// (refine2.py:570-634; base.py:607-673).  32 threads per ray, thread t = k*8+s; the 4 neighbours of a sample are the
// lanes s, 8+s, 16+s, 24+s of the ray's 32-lane group.  layout 0: epi index (k*8+s)*3+c (stage 2 / inference);
// layout 1: s*12+k*3+c (stage 1).
__global__ void refine_input_train_kernel(const float* __restrict__ rays, const float* __restrict__ or_rays, const float* __restrict__ depth_sorted,
                                          const float4* __restrict__ img4, const float* __restrict__ poses, const float* __restrict__ Kmat,
                                          const int64_t* __restrict__ ref_nos, int nv, int Hf, int Wf, float eps, int layout,
                                          float* __restrict__ out, int64_t n) {
  const int64_t total = n * 32;
  const int64_t plane = (int64_t)Hf * Wf;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t rounds = (total + stride - 1) / stride;         // every lane runs every round: the shuffles below need full groups
  for (int64_t it = 0; it < rounds; ++it) {
    const int64_t q = it * stride + blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const bool live = q < total;
    const int64_t ray = live ? q / 32 : n - 1;
    const int t = (int)(q & 31);
    const int k = t >> 3, s = t & 7;
    const float dn = depth_sorted[ray * 8 + s];
    const float* orr = or_rays + ray * 11;
    const float z3d = ieee_div(1.f, ieee_sub(ieee_sub(1.f, dn), eps));
    const float w0 = ieee_add(orr[0], ieee_mul(orr[3], z3d)), w1 = ieee_add(orr[1], ieee_mul(orr[4], z3d)),
                w2 = ieee_add(orr[2], ieee_mul(orr[5], z3d));
    int64_t view = ref_nos[ray * 4 + k];
    view = view < 0 ? 0 : (view >= nv ? nv - 1 : view);
    float X, Y;
    const bool inside = project_train(poses + view * 12, Kmat, w0, w1, w2, Hf, Wf, X, Y);
    int x0, y0; float wx0, wx1, wy0, wy1; bool fin;
    bilinear_setup(X, Y, Hf, Wf, x0, y0, wx0, wx1, wy0, wy1, fin);
    const bool ok = inside && fin;
    const bool okx0 = ok && x0 >= 0 && x0 < Wf, okx1 = ok && x0 + 1 >= 0 && x0 + 1 < Wf, oky0 = y0 >= 0 && y0 < Hf, oky1 = y0 + 1 >= 0 && y0 + 1 < Hf;
    const float4* im = img4 + view * plane;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 t00 = (oky0 && okx0) ? im[(int64_t)y0 * Wf + x0] : zero;
    const float4 t01 = (oky0 && okx1) ? im[(int64_t)y0 * Wf + x0 + 1] : zero;
    const float4 t10 = (oky1 && okx0) ? im[(int64_t)(y0 + 1) * Wf + x0] : zero;
    const float4 t11 = (oky1 && okx1) ? im[(int64_t)(y0 + 1) * Wf + x0 + 1] : zero;
    const float a00 = ieee_mul(wx0, wy0), a01 = ieee_mul(wx1, wy0), a10 = ieee_mul(wx0, wy1), a11 = ieee_mul(wx1, wy1);
    float v[3];
    v[0] = ieee_add(ieee_add(ieee_add(ieee_mul(t00.x, a00), ieee_mul(t01.x, a01)), ieee_mul(t10.x, a10)), ieee_mul(t11.x, a11));
    v[1] = ieee_add(ieee_add(ieee_add(ieee_mul(t00.y, a00), ieee_mul(t01.y, a01)), ieee_mul(t10.y, a10)), ieee_mul(t11.y, a11));
    v[2] = ieee_add(ieee_add(ieee_add(ieee_mul(t00.z, a00), ieee_mul(t01.z, a01)), ieee_mul(t10.z, a10)), ieee_mul(t11.z, a11));
    // valid = (sum_c rgb > 0); mean over the valid neighbours of this sample, in k order (refine2.py:622-624)
    const float valid = (ieee_add(ieee_add(v[0], v[1]), v[2]) > 0.f) ? 1.f : 0.f;
    const int base = (threadIdx.x & 63 & 32) + s;
    float cnt = 0.f, m[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const float vk = __shfl(valid, base + kk * 8);
      cnt = ieee_add(cnt, vk);
#pragma unroll
      for (int c = 0; c < 3; ++c) m[c] = ieee_add(m[c], ieee_mul(vk, __shfl(v[c], base + kk * 8)));
    }
    const float den = ieee_add(cnt, 1e-6f);
    if (live) {
      float* o = out + ray * 144 + 48 + (layout == 0 ? t * 3 : s * 12 + k * 3);
#pragma unroll
      for (int c = 0; c < 3; ++c) o[c] = ieee_add(ieee_mul(v[c], valid), ieee_mul(ieee_div(m[c], den), ieee_sub(1.f, valid)));
      if (k == 0) {
        const float* r = rays + ray * 11;
        float hx, hy, hz, m0, m1, m2;
        unit_dir(r[3], r[4], r[5], hx, hy, hz);
        const float px = ieee_add(r[0], ieee_mul(r[3], dn)), py = ieee_add(r[1], ieee_mul(r[4], dn)), pz = ieee_add(r[2], ieee_mul(r[5], dn));
        cross_rn(px, py, pz, hx, hy, hz, m0, m1, m2);
        float* pl = out + ray * 144 + s * 6;
        pl[0] = hx; pl[1] = hy; pl[2] = hz; pl[3] = m0; pl[4] = m1; pl[5] = m2;
      }
    }
  }
}

// ---------------------------------------------------------------- stage-1 exploration (base.py:689-729)
struct ExploreArgs {
  int n_mult, dir1, dir2, S;          // S = 8 * n_mult
  float mults[32];                    // torch.linspace(0, 1 - 1/n_mult, n_mult)
};
// One WAVE per ray, lane l = samples l, l + 64, ...: replicate each of the 8 refined depths n_mult times toward the next (dir1 > 0) or previous
// sample, sort, jitter toward the neighbour (dir2), and lift to query points o + d*z.  S <= 256 samples per ray.  The sort is a rank sort — rank
// = number of smaller values + number of equal values before it: the order a stable insertion sort leaves — over the wave's copy of the S values
// in LDS.  (A thread per ray with a 256-entry private array and an insertion sort: 54 us for 4096 rays x 64 samples.)
__global__ __launch_bounds__(256) void explore_kernel(ExploreArgs a, const float* __restrict__ z8, const float* __restrict__ rays, const float* __restrict__ jitter,
                               float* __restrict__ z_out, float* __restrict__ pts_out, int64_t n) {
  __shared__ float lds[4][2][256];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 4 + wv;
  if (i >= n) return;                                                          // wave-uniform
  float* zz = lds[wv][0];
  float* zs = lds[wv][1];
  const float* r = rays + i * 11;
  const float near = r[6], far = r[7];
  float z[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) z[s] = z8[i * 8 + s];
  const int S = a.S, nm = a.n_mult;
  for (int u = lane; u < S; u += 64) {
    float v;
    if (nm > 1) {
      const int s = u / nm, j = u - s * nm;
      float zc = z[0], nb = a.dir1 > 0 ? z[1] : near;
#pragma unroll
      for (int k = 1; k < 8; ++k)
        if (s == k) { zc = z[k]; nb = a.dir1 > 0 ? (k < 7 ? z[k < 7 ? k + 1 : 7] : far) : z[k - 1]; }
      const float diff = fabsf(ieee_sub(zc, nb));
      const float m = a.dir1 > 0 ? a.mults[j] : -a.mults[j];
      v = ieee_add(zc, ieee_mul(m, diff));
    } else {
      v = z[0];
#pragma unroll
      for (int k = 1; k < 8; ++k) if (u == k) v = z[k];
    }
    zz[u] = v;
  }
  if (nm > 1) {
    for (int u = lane; u < S; u += 64) {
      const float v = zz[u];
      int rank = 0;
      for (int k = 0; k < S; ++k) {
        const float o = zz[k];
        // a NaN ranks behind every number (and NaNs among themselves by index): every slot of zs[] is written exactly once, so a non-finite
        // refined depth (a diverged step) comes out as NaN samples, the same ones on every run, instead of uninitialised LDS
        rank += (o < v || (o == v && k < u) || (v != v && (o == o || k < u))) ? 1 : 0;
      }
      zs[rank] = v;
    }
  } else {
    for (int u = lane; u < S; u += 64) zs[u] = zz[u];
  }
  for (int u = lane; u < S; u += 64) {                                         // jitter uses the un-jittered neighbours (base.py:721-727)
    const float cur = zs[u];
    const float nb = a.dir2 > 0 ? (u + 1 < S ? zs[u + 1] : far) : (u > 0 ? zs[u - 1] : near);
    const float jv = jitter[i * S + u];
    const float zo = ieee_add(cur, ieee_mul(a.dir2 > 0 ? jv : -jv, fabsf(ieee_sub(cur, nb))));
    z_out[i * S + u] = zo;
    float* p = pts_out + (i * S + u) * 3;
    p[0] = ieee_add(r[0], ieee_mul(r[3], zo)); p[1] = ieee_add(r[1], ieee_mul(r[4], zo)); p[2] = ieee_add(r[2], ieee_mul(r[5], zo));
  }
}

// ---------------------------------------------------------------- raw2outputs (trt.py:564-597; base.py:501-551; refine2.py:475-522)
// One thread per ray: the form for many rays (from 65 536 on every SIMD has waves to hide the latencies with).
__global__ void composite_thread_kernel(const float* __restrict__ raw, const float* __restrict__ z, const float* __restrict__ rays_d, int d_stride,
                                 const float* __restrict__ add, const float* __restrict__ mul, const float* __restrict__ noise, float clampv,
                                 int white_bkgd, float* __restrict__ rgb, float* __restrict__ disp, float* __restrict__ acc_out,
                                 float* __restrict__ weights, float* __restrict__ depth, int64_t n, int S) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float* d = rays_d + i * d_stride;
    const float dn = ieee_sqrt(ieee_add(ieee_add(ieee_mul(d[0], d[0]), ieee_mul(d[1], d[1])), ieee_mul(d[2], d[2])));
    float T = 1.f, s0 = 0.f, s1 = 0.f, s2 = 0.f, sd = 0.f, sa = 0.f;
    // four samples' inputs in flight per thread, then the recurrence over them in order (one sample per loop iteration waited out a memory
    // latency per sample: 51 us for 4096 rays x 64 samples on a handful of waves); same operations in the same order
    for (int sb = 0; sb < S; sb += 4) {
      const int64_t e0 = i * S + sb;
      const int m = S - sb < 4 ? S - sb : 4;
      float4 rw[4];
      float zz[5], nz[4], ad[4], ml[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t e = e0 + (u < m ? u : m - 1);
        rw[u] = *(const float4*)(raw + e * 4);
        zz[u] = z[e];
        nz[u] = noise ? noise[e] : 0.f;
        ad[u] = add ? add[e] : 0.f;
        ml[u] = mul ? mul[e] : 0.f;
      }
      zz[4] = sb + 4 < S ? z[e0 + 4] : 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (u >= m) break;
        const int s = sb + u;
        float r0 = rw[u].x, r1 = rw[u].y, r2 = rw[u].z, r3 = rw[u].w;
        if (clampv > 0.f) {
          r0 = fminf(fmaxf(r0, -clampv), clampv); r1 = fminf(fmaxf(r1, -clampv), clampv);
          r2 = fminf(fmaxf(r2, -clampv), clampv); r3 = fminf(fmaxf(r3, -clampv), clampv);
        }
        const float zc = zz[u];
        float dist = (s + 1 < S) ? ieee_sub(zz[u + 1], zc) : 1e10f;
        dist = ieee_mul(dist, dn);
        float sg = r3;
        if (noise) sg = ieee_add(sg, nz[u]);
        if (add) sg = ieee_add(sg, ad[u]);
        sg = fmaxf(sg, 0.f);
        float alpha = ieee_sub(1.f, expf(ieee_mul(-sg, dist)));
        if (mul) alpha = ieee_mul(alpha, fmaxf(ml[u], 0.f));
        const float w = ieee_mul(alpha, T);
        T = ieee_mul(T, ieee_add(ieee_sub(1.f, alpha), 1e-10f));
        s0 = ieee_add(s0, ieee_mul(w, sigmoid_f(r0)));
        s1 = ieee_add(s1, ieee_mul(w, sigmoid_f(r1)));
        s2 = ieee_add(s2, ieee_mul(w, sigmoid_f(r2)));
        sd = ieee_add(sd, ieee_mul(w, zc));
        sa = ieee_add(sa, w);
        if (weights) weights[e0 + u] = w;
      }
    }
    if (white_bkgd) { const float bg = ieee_sub(1.f, sa); s0 = ieee_add(s0, bg); s1 = ieee_add(s1, bg); s2 = ieee_add(s2, bg); }
    if (rgb) { rgb[i * 3] = s0; rgb[i * 3 + 1] = s1; rgb[i * 3 + 2] = s2; }
    if (depth) depth[i] = sd;
    if (acc_out) acc_out[i] = sa;
    if (disp) disp[i] = ieee_div(1.f, fmaxf(1e-10f, ieee_div(sd, sa)));
  }
}

// One WAVE per ray, lane = sample (chunks of 64 for S > 64): the per-sample work (loads, exp, sigmoids) runs in parallel, the transmittance
// product and the five sums run over the lanes in sample order through v_readlane — the same operations in the same order as a thread walking
// the ray, without its S memory latencies (thread per ray: 41 us for 4096 rays x 64 samples; 4096 threads do not fill 1024 SIMDs).
__global__ __launch_bounds__(256) void composite_kernel(const float* __restrict__ raw, const float* __restrict__ z, const float* __restrict__ rays_d, int d_stride,
                                 const float* __restrict__ add, const float* __restrict__ mul, const float* __restrict__ noise, float clampv,
                                 int white_bkgd, float* __restrict__ rgb, float* __restrict__ disp, float* __restrict__ acc_out,
                                 float* __restrict__ weights, float* __restrict__ depth, int64_t n, int S) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;                                                          // wave-uniform
  const float* d = rays_d + i * d_stride;
  const float dn = ieee_sqrt(ieee_add(ieee_add(ieee_mul(d[0], d[0]), ieee_mul(d[1], d[1])), ieee_mul(d[2], d[2])));
  float T = 1.f, s0 = 0.f, s1 = 0.f, s2 = 0.f, sd = 0.f, sa = 0.f;            // uniform: every lane carries the same running values
  auto lane_val = [](float v, int j) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j)); };
  for (int c0 = 0; c0 < S; c0 += 64) {
    const int m = S - c0 < 64 ? S - c0 : 64;
    const int s = c0 + lane;
    const bool on = lane < m;
    const int64_t e = i * S + (on ? s : c0);
    const float4 rw = *(const float4*)(raw + e * 4);
    const float zc = z[e];
    const float zn = (on && s + 1 < S) ? z[e + 1] : 0.f;
    float r0 = rw.x, r1 = rw.y, r2 = rw.z, r3 = rw.w;
    if (clampv > 0.f) {
      r0 = fminf(fmaxf(r0, -clampv), clampv); r1 = fminf(fmaxf(r1, -clampv), clampv);
      r2 = fminf(fmaxf(r2, -clampv), clampv); r3 = fminf(fmaxf(r3, -clampv), clampv);
    }
    float dist = (s + 1 < S) ? ieee_sub(zn, zc) : 1e10f;
    dist = ieee_mul(dist, dn);
    float sg = r3;
    if (noise) sg = ieee_add(sg, noise[e]);
    if (add) sg = ieee_add(sg, add[e]);
    sg = fmaxf(sg, 0.f);
    float alpha = ieee_sub(1.f, expf(ieee_mul(-sg, dist)));
    if (mul) alpha = ieee_mul(alpha, fmaxf(mul[e], 0.f));
    const float x = ieee_add(ieee_sub(1.f, alpha), 1e-10f);
    float Tl = 0.f;                                                            // exclusive product up to this lane's sample
    for (int j = 0; j < m; ++j) {
      Tl = lane == j ? T : Tl;
      T = ieee_mul(T, lane_val(x, j));
    }
    const float w = ieee_mul(alpha, Tl);
    const float c0v = ieee_mul(w, sigmoid_f(r0)), c1v = ieee_mul(w, sigmoid_f(r1)), c2v = ieee_mul(w, sigmoid_f(r2)), cdv = ieee_mul(w, zc);
    for (int j = 0; j < m; ++j) {
      s0 = ieee_add(s0, lane_val(c0v, j));
      s1 = ieee_add(s1, lane_val(c1v, j));
      s2 = ieee_add(s2, lane_val(c2v, j));
      sd = ieee_add(sd, lane_val(cdv, j));
      sa = ieee_add(sa, lane_val(w, j));
    }
    if (weights && on) weights[e] = w;
  }
  if (white_bkgd) { const float bg = ieee_sub(1.f, sa); s0 = ieee_add(s0, bg); s1 = ieee_add(s1, bg); s2 = ieee_add(s2, bg); }
  if (lane == 0) {
    if (rgb) { rgb[i * 3] = s0; rgb[i * 3 + 1] = s1; rgb[i * 3 + 2] = s2; }
    if (depth) depth[i] = sd;
    if (acc_out) acc_out[i] = sa;
    if (disp) disp[i] = ieee_div(1.f, fmaxf(1e-10f, ieee_div(sd, sa)));
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------ C ABI
extern "C" int pnrf_posenc_fwd(const float* x, float* out, int64_t n, int n_freq, void* stream) {
  PNRF_REQUIRE(n >= 0 && n_freq >= 0 && n_freq <= 16 && (n == 0 || (x && out)), PNRF_E_ARG, "pnrf_posenc_fwd: bad arguments");
  if (n == 0) return 0;
  hipLaunchKernelGGL(posenc_kernel, dim3(grid_for(n * 3)), dim3(TPB), 0, (hipStream_t)stream, x, out, n * 3, n_freq);
  PNRF_LAUNCH_CHECK();
  return 0;
}

extern "C" int pnrf_plucker_fwd(const float* o, const float* d, float* out, int64_t n, void* stream) {
  PNRF_REQUIRE(n >= 0 && (n == 0 || (o && d && out)), PNRF_E_ARG, "pnrf_plucker_fwd: bad arguments");
  if (n == 0) return 0;
  hipLaunchKernelGGL(plucker_kernel, dim3(grid_for(n)), dim3(TPB), 0, (hipStream_t)stream, o, d, out, n);
  PNRF_LAUNCH_CHECK();
  return 0;
}

// small per-device cache of linspace tables for pnrf_ray_encode_fwd (n_pts <= 256)
#include <map>
#include <mutex>
static const float* tvals_for(int n_pts) {
  static std::mutex mu;
  static std::map<std::pair<int, int>, float*> cache;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  auto key = std::make_pair(dev, n_pts);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  std::vector<float> host(n_pts);
  pnrf_linspace(0.f, 1.f, n_pts, host.data());
  float* d = nullptr;
  if (hipMalloc((void**)&d, n_pts * sizeof(float)) != hipSuccess) return nullptr;
  if (hipMemcpy(d, host.data(), n_pts * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return nullptr; }
  cache[key] = d;
  return d;
}

extern "C" int pnrf_ray_encode_fwd(const float* rays, float* mm_input, int64_t n, int n_pts, void* stream) {
  PNRF_REQUIRE(n >= 0 && n_pts >= 1 && n_pts <= 256 && (n == 0 || (rays && mm_input)), PNRF_E_ARG, "pnrf_ray_encode_fwd: bad arguments");
  if (n == 0) return 0;
  const float* tv = tvals_for(n_pts);      // first call per (device, n_pts) allocates; later calls are launch-only
  PNRF_REQUIRE(tv, PNRF_E_STATE, "pnrf_ray_encode_fwd: could not allocate the ray-point table");
  hipLaunchKernelGGL(ray_encode_kernel, dim3(grid_for(n * n_pts)), dim3(TPB), 0, (hipStream_t)stream, rays, tv, mm_input, n, n_pts);
  PNRF_LAUNCH_CHECK();
  return 0;
}

extern "C" int pnrf_frame_rays_blocks_fwd(const float* K, const float* c2w, int H, int W, float near, float far, float or_near, float or_far,
                                          int64_t first, int64_t block, int64_t stride, int64_t count, float* rays, float* or_rays, void* stream) {
  PNRF_REQUIRE(K && c2w && H > 0 && W > 0 && first >= 0 && count >= 0 && block >= 1 && stride >= 0, PNRF_E_ARG,
               "pnrf_frame_rays_blocks_fwd: bad arguments (H=%d W=%d first=%lld block=%lld stride=%lld count=%lld)", H, W, (long long)first,
               (long long)block, (long long)stride, (long long)count);
  if (count == 0) return 0;
  {
    const int64_t last = first + ((count - 1) / block) * stride + (count - 1) % block;          // the largest pixel index addressed
    PNRF_REQUIRE(last < (int64_t)H * W && (stride == 0 ? count <= block : stride >= block), PNRF_E_ARG,
                 "pnrf_frame_rays_blocks_fwd: the blocks leave the %d x %d frame or overlap (last pixel %lld)", H, W, (long long)last);
  }
  PNRF_REQUIRE(rays && or_rays, PNRF_E_ARG, "pnrf_frame_rays_blocks_fwd: null output");
  FrameArgs a;
  a.K00 = K[0]; a.K02 = K[2]; a.K11 = K[4]; a.K12 = K[5];
  a.sx = ndc_scale(W, K[0]); a.sy = ndc_scale(H, K[0]);
  for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) a.R[r * 3 + c] = c2w[r * 4 + c]; a.T[r] = c2w[r * 4 + 3]; }
  a.H = H; a.W = W; a.near = near; a.far = far; a.or_near = or_near; a.or_far = or_far; a.first = first; a.count = count;
  a.block = block; a.stride = stride;
  hipLaunchKernelGGL(frame_rays_kernel, dim3(grid_for(count)), dim3(TPB), 0, (hipStream_t)stream, a, rays, or_rays);
  PNRF_LAUNCH_CHECK();
  return 0;
}

extern "C" int pnrf_frame_rays_fwd(const float* K, const float* c2w, int H, int W, float near, float far, float or_near,
                                   float or_far, int64_t first, int64_t count, float* rays, float* or_rays, void* stream) {
  PNRF_REQUIRE(K && c2w && H > 0 && W > 0 && first >= 0 && count >= 0 && first + count <= (int64_t)H * W, PNRF_E_ARG,
               "pnrf_frame_rays_fwd: bad arguments (H=%d W=%d first=%lld count=%lld)", H, W, (long long)first, (long long)count);
  if (count == 0) return 0;
  return pnrf_frame_rays_blocks_fwd(K, c2w, H, W, near, far, or_near, or_far, first, count, 0, count, rays, or_rays, stream);
}

extern "C" int pnrf_ndc_rays_fwd(const float* rays_o, const float* rays_d, int H, int W, float focal, float near,
                                 float* out_o, float* out_d, int64_t n, void* stream) {
  PNRF_REQUIRE(n >= 0 && H > 0 && W > 0, PNRF_E_ARG, "pnrf_ndc_rays_fwd: bad sizes");
  if (n == 0) return 0;
  PNRF_REQUIRE(rays_o && rays_d && out_o && out_d, PNRF_E_ARG, "pnrf_ndc_rays_fwd: null pointer");
  hipLaunchKernelGGL(ndc_rays_kernel, dim3(grid_for(n)), dim3(TPB), 0, (hipStream_t)stream, rays_o, rays_d, ndc_scale(W, focal), ndc_scale(H, focal), near,
                     (float)(2.0 * (double)near), out_o, out_d, n);
  PNRF_LAUNCH_CHECK();
  return 0;
}

extern "C" int pnrf_warp_trt_fwd(const float* img, const float* depth, const float* ro1, const float* rd1, int64_t ray_bstride,
                                 const float* w2c, float* out, int B, int Hf, int Wf, int64_t n, void* stream) {
  PNRF_REQUIRE(B >= 0 && n >= 0 && Hf >= 2 && Wf >= 2 && ray_bstride >= 0, PNRF_E_ARG, "pnrf_warp_trt_fwd: bad sizes");
  if (B == 0 || n == 0) return 0;
  PNRF_REQUIRE(img && depth && ro1 && rd1 && w2c && out, PNRF_E_ARG, "pnrf_warp_trt_fwd: null pointer");
  hipLaunchKernelGGL(warp_trt_kernel, dim3(grid_for((int64_t)B * n)), dim3(TPB), 0, (hipStream_t)stream, img, depth, ro1, rd1, ray_bstride, w2c, out, B, Hf, Wf, n);
  PNRF_LAUNCH_CHECK();
  return 0;
}

extern "C" int pnrf_images_pack(const float* img_nchw, float* out_nhwc4, int nv, int Hf, int Wf, void* stream) {
  PNRF_REQUIRE(nv >= 0 && Hf > 0 && Wf > 0, PNRF_E_ARG, "pnrf_images_pack: bad sizes");
  if (nv == 0) return 0;
  PNRF_REQUIRE(img_nchw && out_nhwc4, PNRF_E_ARG, "pnrf_images_pack: null pointer");
  const int64_t plane = (int64_t)Hf * Wf;
  hipLaunchKernelGGL(images_pack_kernel, dim3(grid_for((int64_t)nv * plane)), dim3(TPB), 0, (hipStream_t)stream, img_nchw, (float4*)out_nhwc4, nv, plane);
  PNRF_LAUNCH_CHECK();
  return 0;
}

extern "C" int pnrf_refine_input_fwd(const float* rays, const float* or_rays, const float* depth_sorted, const float* img4,
                                     const float* proj, int nb, int Hf, int Wf, float eps, float* refine_in, int64_t n, void* stream) {
  PNRF_REQUIRE(n >= 0 && nb >= 1 && nb <= 8 && Hf >= 2 && Wf >= 2, PNRF_E_ARG, "pnrf_refine_input_fwd: bad sizes (nb must be 1 .. 8, got %d)", nb);
  if (n == 0) return 0;
  PNRF_REQUIRE(rays && or_rays && depth_sorted && img4 && proj && refine_in, PNRF_E_ARG, "pnrf_refine_input_fwd: null pointer");
  hipLaunchKernelGGL(refine_input_kernel, dim3(grid_for((n + RI_TILE - 1) / RI_TILE, 1)), dim3(256), 0, (hipStream_t)stream, rays, or_rays, depth_sorted,
                     (const float4*)img4, proj, nb, Hf, Wf, eps, refine_in, n);
  PNRF_LAUNCH_CHECK();
  return 0;
}

extern "C" int pnrf_composite_fwd(const float* raw, const float* z, const float* rays_d, int d_stride, const float* add,
                                  const float* mul, const float* noise, float clampv, int white_bkgd, float* rgb, float* disp,
                                  float* acc, float* weights, float* depth, int64_t n, int s, void* stream) {
  PNRF_REQUIRE(n >= 0 && s >= 1 && d_stride >= 3, PNRF_E_ARG, "pnrf_composite_fwd: bad sizes");
  if (n == 0) return 0;
  PNRF_REQUIRE(raw && z && rays_d, PNRF_E_ARG, "pnrf_composite_fwd: null pointer");
  PNRF_REQUIRE(((uintptr_t)raw & 15) == 0, PNRF_E_ARG, "pnrf_composite_fwd: raw must be 16-byte aligned (the kernel reads [r, g, b, sigma] as one 16-byte load)");
  // few rays (a training batch): a wave per ray; many rays (a frame): a thread per ray.  Same operations in the same order either way.
  if (n < 65536)
    hipLaunchKernelGGL(composite_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, raw, z, rays_d, d_stride, add, mul, noise,
                       clampv, white_bkgd, rgb, disp, acc, weights, depth, n, s);
  else
    hipLaunchKernelGGL(composite_thread_kernel, dim3(grid_for(n)), dim3(TPB), 0, (hipStream_t)stream, raw, z, rays_d, d_stride, add, mul, noise,
                       clampv, white_bkgd, rgb, disp, acc, weights, depth, n, s);
  PNRF_LAUNCH_CHECK();
  return 0;
}

extern "C" int pnrf_warp_train_fwd(const float* img, const float* depth, const float* ro1, const float* rd1, int64_t ray_bstride,
                                   const float* c2w2, const float* K, float* out, int B, int Hf, int Wf, int64_t n, void* stream) {
  PNRF_REQUIRE(B >= 0 && n >= 0 && Hf >= 2 && Wf >= 2 && ray_bstride >= 0, PNRF_E_ARG, "pnrf_warp_train_fwd: bad sizes");
  if (B == 0 || n == 0) return 0;
  PNRF_REQUIRE(img && depth && ro1 && rd1 && c2w2 && K && out, PNRF_E_ARG, "pnrf_warp_train_fwd: null pointer");
  hipLaunchKernelGGL(warp_train_kernel, dim3(grid_for((int64_t)B * n)), dim3(TPB), 0, (hipStream_t)stream, img, depth, ro1, rd1, ray_bstride, c2w2, K,
                     out, B, Hf, Wf, n);
  PNRF_LAUNCH_CHECK();
  return 0;
}

extern "C" int pnrf_refine_input_train_fwd(const float* rays, const float* or_rays, const float* depth_sorted, const float* img4,
                                           const float* poses, const float* K, const int64_t* ref_nos, int nv, int nb, int Hf, int Wf,
                                           float eps, int layout, float* refine_in, int64_t n, void* stream) {
  PNRF_REQUIRE(n >= 0 && nb == 4 && nv >= 1 && Hf >= 2 && Wf >= 2 && (layout == 0 || layout == 1), PNRF_E_ARG,
               "pnrf_refine_input_train_fwd: bad sizes (nb must be 4, got %d; layout 0|1, got %d)", nb, layout);
  if (n == 0) return 0;
  PNRF_REQUIRE(rays && or_rays && depth_sorted && img4 && poses && K && ref_nos && refine_in, PNRF_E_ARG, "pnrf_refine_input_train_fwd: null pointer");
  hipLaunchKernelGGL(refine_input_train_kernel, dim3(grid_for(n * 32)), dim3(TPB), 0, (hipStream_t)stream, rays, or_rays, depth_sorted,
                     (const float4*)img4, poses, K, ref_nos, nv, Hf, Wf, eps, layout, refine_in, n);
  PNRF_LAUNCH_CHECK();
  return 0;
}

extern "C" int pnrf_explore_fwd(const float* z8, const float* rays, const float* jitter, int n_mult, int dir1, int dir2,
                                float* z_out, float* pts_out, int64_t n, void* stream) {
  PNRF_REQUIRE(n >= 0 && n_mult >= 1 && n_mult <= 32 && (dir1 == 1 || dir1 == -1) && (dir2 == 1 || dir2 == -1), PNRF_E_ARG,
               "pnrf_explore_fwd: bad arguments (n_mult 1..32, dir +-1)");
  if (n == 0) return 0;
  PNRF_REQUIRE(z8 && rays && jitter && z_out && pts_out, PNRF_E_ARG, "pnrf_explore_fwd: null pointer");
  ExploreArgs a;
  a.n_mult = n_mult; a.dir1 = dir1; a.dir2 = dir2; a.S = 8 * n_mult;
  pnrf_linspace(0.f, (float)(1.0 - 1.0 / n_mult), n_mult, a.mults);
  hipLaunchKernelGGL(explore_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a, z8, rays, jitter, z_out, pts_out, n);
  PNRF_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------ context + whole path
struct pnrf_ctx {
  const pnrf_mlp* sampler;
  const pnrf_mlp* refine;
  const pnrf_mlp* nerf;
  int64_t max_rays;
  float* ws;             // one allocation: depth[8] add[8] mul[8] z[8] pts[24] per ray, then the two-pass sampler's workspace
  void* sampler_ws;      // inside ws: pnrf_sampler_workspace_bytes(max_rays)
  bool sampler_ws_clean; // its counters are zero: cleared at creation, left at zero by every completed call (sampler_h16_kernel's last workgroup)
  int device;
  // per-stage timing (pnrf_ctx_profile_begin / _end): 4 events per profiled call, recorded on the caller's stream
  hipEvent_t* ev;
  int prof_cap, prof_n;
  bool prof_on;
  float kappa;           // two-pass sampler threshold (pnrf_ctx_set_sampler_kappa); < 0 = PNRF_SAMPLER_KAPPA
};
static constexpr int PROF_EVENTS = 4;
static constexpr int PROF_MAX_FRAMES = 4096;

static void profile_release(pnrf_ctx* c) {
  if (c->ev) {
    for (int i = 0; i < c->prof_cap * PROF_EVENTS; ++i) (void)hipEventDestroy(c->ev[i]);
    delete[] c->ev;
  }
  c->ev = nullptr; c->prof_cap = 0; c->prof_n = 0; c->prof_on = false;
}
static constexpr int WS_FLOATS_PER_RAY = 8 + 8 + 8 + 8 + 24;      // 224 B per ray

extern "C" int pnrf_ctx_create(const pnrf_mlp_t* sampler, const pnrf_mlp_t* refine, const pnrf_mlp_t* nerf, int64_t max_rays, pnrf_ctx_t** out) {
  PNRF_REQUIRE(sampler && refine && nerf && out && max_rays > 0, PNRF_E_ARG, "pnrf_ctx_create: bad arguments");
  PNRF_REQUIRE(sampler->net == PNRF_NET_SAMPLER && refine->net == PNRF_NET_REFINE && (nerf->net == PNRF_NET_NERF || nerf->net == PNRF_NET_NERFCLS),
               PNRF_E_ARG, "pnrf_ctx_create: handles must be (sampler, refine, nerf | nerf-class)");
  pnrf_ctx* c = new pnrf_ctx();
  c->sampler = sampler; c->refine = refine; c->nerf = nerf; c->max_rays = max_rays; c->ws = nullptr;
  c->ev = nullptr; c->prof_cap = 0; c->prof_n = 0; c->prof_on = false;
  c->kappa = -1.f; c->sampler_ws_clean = false;
  hipError_t e = hipGetDevice(&c->device);
  PNRF_REQUIRE(max_rays < ((int64_t)1 << 31), PNRF_E_ARG, "pnrf_ctx_create: at most 2^31 - 1 rays per context");
  const size_t ws_rays = (((size_t)max_rays * WS_FLOATS_PER_RAY * sizeof(float)) + 255) & ~(size_t)255;
  if (e == hipSuccess) e = hipMalloc((void**)&c->ws, ws_rays + (size_t)pnrf_sampler_workspace_bytes(max_rays));
  if (e == hipSuccess) {
    c->sampler_ws = (char*)c->ws + ws_rays;
    e = hipMemset(c->sampler_ws, 0, 64);
    c->sampler_ws_clean = e == hipSuccess;
  }
  if (e != hipSuccess) {
    set_error("pnrf_ctx_create: workspace allocation failed: %s", hipGetErrorString(e));
    delete c;
    return (int)e;
  }
  *out = c;
  return 0;
}

extern "C" int pnrf_ctx_free(pnrf_ctx_t* c) {
  if (!c) return 0;
  if (c->ws) (void)hipFree(c->ws);
  profile_release(c);
  delete c;
  return 0;
}

extern "C" int pnrf_render_rays_fwd(pnrf_ctx_t* c, const float* rays, const float* or_rays, const float* img4, const float* proj,
                                    int nb, int Hf, int Wf, float eps, float* rgbd, int64_t* sort_idx, int64_t n, void* stream) {
  PNRF_REQUIRE(c, PNRF_E_ARG, "pnrf_render_rays_fwd: null context");
  PNRF_REQUIRE(n >= 0 && n <= c->max_rays, PNRF_E_STATE, "pnrf_render_rays_fwd: %lld rays exceed the context's capacity %lld",
               (long long)n, (long long)c->max_rays);
  if (n == 0) return 0;
  PNRF_REQUIRE(rays && or_rays && img4 && proj && rgbd, PNRF_E_ARG, "pnrf_render_rays_fwd: null pointer");
  {
    int cur = -1;
    PNRF_HIP(hipGetDevice(&cur));
    PNRF_REQUIRE(cur == c->device, PNRF_E_STATE, "pnrf_render_rays_fwd: the context lives on device %d, the calling thread's current device is %d",
                 c->device, cur);
  }
  float* depth = c->ws;
  float* add = depth + c->max_rays * 8;
  float* mul = add + c->max_rays * 8;
  float* z = mul + c->max_rays * 8;
  float* pts = z + c->max_rays * 8;
  int rc;
  hipEvent_t* ev = (c->prof_on && c->prof_n < c->prof_cap) ? c->ev + (size_t)c->prof_n * PROF_EVENTS : nullptr;
  hipStream_t st = (hipStream_t)stream;
  if (ev) PNRF_HIP(hipEventRecord(ev[0], st));
  const bool clean = c->sampler_ws_clean;
  c->sampler_ws_clean = false;                       // a call that fails between the two passes leaves the counters set: the next one clears them
  if ((rc = pnrf_sampler_fwd_ws_impl(c->sampler, rays, n, depth, add, mul, sort_idx, nullptr, nullptr, c->sampler_ws,
                                     pnrf_sampler_workspace_bytes(c->max_rays), c->kappa, clean, stream))) return rc;       // trt.py:628-635
  c->sampler_ws_clean = true;
  if (ev) PNRF_HIP(hipEventRecord(ev[1], st));
  if ((rc = pnrf_refine_project_fwd(c->refine, rays, or_rays, depth, img4, proj, nb, Hf, Wf, eps, z, pts, n, stream))) return rc;   // :637-681
  if (ev) PNRF_HIP(hipEventRecord(ev[2], st));
  // NeRF stage with its batches handed out dynamically (words 8, 9 of the sampler workspace's 16-word header: zero since creation, left at zero
  // by every launch): a workgroup that another stream's kernel keeps off its CU for a while does not hold the frame up with its whole share
  if ((rc = pnrf_nerf_fwd_queue_impl(c->nerf, pts, rays, z, add, mul, nullptr, 0.f, 0, 8, rgbd, nullptr, n, (int*)c->sampler_ws + 8, stream))) return rc;   // :691-694
  if (ev) {
    PNRF_HIP(hipEventRecord(ev[3], st));
    c->prof_n += 1;
  }
  return 0;
}

extern "C" int pnrf_ctx_set_sampler_kappa(pnrf_ctx_t* c, float kappa) {
  PNRF_REQUIRE(c, PNRF_E_ARG, "pnrf_ctx_set_sampler_kappa: null context");
  PNRF_REQUIRE(kappa == kappa && kappa < 1e30f, PNRF_E_ARG, "pnrf_ctx_set_sampler_kappa: kappa must be a number below 1e30 (negative = default), got %g", (double)kappa);
  c->kappa = kappa;
  return 0;
}

extern "C" int pnrf_ctx_get_sampler_kappa(const pnrf_ctx_t* c, float* kappa) {
  PNRF_REQUIRE(c && kappa, PNRF_E_ARG, "pnrf_ctx_get_sampler_kappa: null argument");
  *kappa = c->kappa < 0.f ? PNRF_SAMPLER_KAPPA : c->kappa;
  return 0;
}

extern "C" int pnrf_ctx_sampler_stats(pnrf_ctx_t* c, int64_t* rays_second_pass) {
  PNRF_REQUIRE(c && rays_second_pass, PNRF_E_ARG, "pnrf_ctx_sampler_stats: null argument");
  int v[2] = {0, 0};
  PNRF_HIP(hipMemcpy(v, c->sampler_ws, sizeof(v), hipMemcpyDeviceToHost));       // synchronises with the device: diagnostics only
  *rays_second_pass = v[1];
  return 0;
}

extern "C" int pnrf_ctx_sampler_saturated(pnrf_ctx_t* c, int64_t* rays_third_pass) {
  PNRF_REQUIRE(c && rays_third_pass, PNRF_E_ARG, "pnrf_ctx_sampler_saturated: null argument");
  int v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  PNRF_HIP(hipMemcpy(v, c->sampler_ws, sizeof(v), hipMemcpyDeviceToHost));       // synchronises with the device: diagnostics only
  *rays_third_pass = v[5];
  return 0;
}

extern "C" int pnrf_ctx_profile_begin(pnrf_ctx_t* c, int max_frames) {
  PNRF_REQUIRE(c && max_frames > 0 && max_frames <= PROF_MAX_FRAMES, PNRF_E_ARG, "pnrf_ctx_profile_begin: need a context and 1..%d frames", PROF_MAX_FRAMES);
  if (max_frames > c->prof_cap) {
    profile_release(c);
    c->ev = new hipEvent_t[(size_t)max_frames * PROF_EVENTS];
    for (int i = 0; i < max_frames * PROF_EVENTS; ++i) {
      hipError_t e = hipEventCreate(&c->ev[i]);
      if (e != hipSuccess) {
        for (int j = 0; j < i; ++j) (void)hipEventDestroy(c->ev[j]);
        delete[] c->ev;
        c->ev = nullptr;
        set_error("pnrf_ctx_profile_begin: hipEventCreate failed: %s", hipGetErrorString(e));
        return (int)e;
      }
    }
    c->prof_cap = max_frames;
  }
  c->prof_n = 0;
  c->prof_on = true;
  return 0;
}

extern "C" int pnrf_ctx_profile_end(pnrf_ctx_t* c, float* ms, int* frames) {
  PNRF_REQUIRE(c && ms && frames, PNRF_E_ARG, "pnrf_ctx_profile_end: null argument");
  PNRF_REQUIRE(c->prof_on, PNRF_E_STATE, "pnrf_ctx_profile_end: no pnrf_ctx_profile_begin before it");
  c->prof_on = false;
  *frames = c->prof_n;
  for (int k = 0; k < PROF_EVENTS - 1; ++k) ms[k] = 0.f;
  if (c->prof_n == 0) return 0;
  PNRF_HIP(hipEventSynchronize(c->ev[(size_t)c->prof_n * PROF_EVENTS - 1]));
  double acc[PROF_EVENTS - 1] = {0, 0, 0};
  for (int i = 0; i < c->prof_n; ++i)
    for (int k = 0; k < PROF_EVENTS - 1; ++k) {
      float t = 0.f;
      PNRF_HIP(hipEventElapsedTime(&t, c->ev[(size_t)i * PROF_EVENTS + k], c->ev[(size_t)i * PROF_EVENTS + k + 1]));
      acc[k] += t;
    }
  for (int k = 0; k < PROF_EVENTS - 1; ++k) ms[k] = (float)(acc[k] / c->prof_n);
  return 0;
}
