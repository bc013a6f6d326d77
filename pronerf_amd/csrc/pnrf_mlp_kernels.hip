// pnrf_mlp_kernels.hip — the three fused MLP stages of the ProNeRF render path + their
// module-level (x -> y) variants.  gfx950 only.
//
//   sampler_kernel : [Pluecker ray encoding] -> fp32 MLP (v_mfma_f32_32x32x2_f32) -> sigmoid,
//                    depth affine, stable sort-8, add/mul permutation      (trt.py:628-635)
//   refine_kernel  : refine_in -> bf16 MLP (v_mfma_f32_32x32x16_bf16) -> sigmoid/tanh,
//                    interval refinement, query points                      (trt.py:668-681)
//   nerf_kernel    : positional encoding -> bf16 MLP -> alpha compositing   (trt.py:691-694)
//
// Workgroup = NW waves (4 = one per SIMD with the 512-register budget, 8 = two per SIMD with 256).
// A wave owns 32*NCB columns (rays or ray-samples) and keeps two layers of activations in
// registers (ping-pong, no copies); the workgroup streams each network's packed weights once per
// batch of NW*32*NCB columns through the LDS ring (pnrf_engine.h).
#include <type_traits>

#include "pnrf_common.h"
#include "pnrf_geom.h"

#ifndef PNRF_LAST_TMAX
#define PNRF_LAST_TMAX 1          // output layers of the NeRF nets: 4 (3) rows, all in the first 16-row tile of the pair
#endif
#ifndef PNRF_OLD_PRIO
#define PNRF_OLD_PRIO 0
#endif
#ifndef PNRF_YOUNG_PRIO
#define PNRF_YOUNG_PRIO 1
#endif

using namespace pnrf;

namespace {



// compile-time loop: f(std::integral_constant<int,I>) for I in [0,N) — keeps register-array indices static
template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<N, I + 1>(f);
  }
}

typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef short i16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }
// hardware exp2 / reciprocal (1 ulp each): for the fused bf16 kernels' epilogues
__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanh_fast(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * x) + 1.f); }
// DPP lane moves inside a row of 16 lanes; lanes without a source read 0 (bound_ctrl)
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// sum over each aligned group of 8 lanes, result in all 8: xor 1, xor 2 (quad_perm), then the mirrored half row
__device__ __forceinline__ float sum8_dpp(float v) {
  v = ieee_add(v, dpp_mov<0xB1>(v));
  v = ieee_add(v, dpp_mov<0x4E>(v));
  return ieee_add(v, dpp_mov<0x141>(v));
}

// Two waves per SIMD: the second-dispatched half of the workgroup (waves NW/2..NW-1) loses every VALU/MFMA
// arbitration to its older partner, finishes each tile late and makes the older half wait at the slot barrier
// (tools/diag_stamps.py: 109 vs 600 cycles of barrier wait per tile).  One static s_setprio for that half
// evens the pair out (cdna guide T5, static form).  The condition must be provably wave-uniform.
// A WIDE fused workgroup (8 waves, two per SIMD) claims the SIMDs' whole register file: 256 VGPRs per wave whatever the kernel uses.
// Round 6 (NOTEBOOK §22): with 240 + 240 registers allocated a 32-register wave of ANOTHER kernel fits beside the pair — and in that configuration (the
// register file exactly full) the refine stage's slower wave half returned wrong rows, a few calls in 10^4, in code whose two-wave-per-SIMD form is clean
// alone, clean at 232 + 232 (+ 32) registers and clean at 256 + 256: the third case of this family on this chip (round 4's paired workgroups, round 5's
// packed fp32).  Nothing can be co-resident on a SIMD whose two waves own all 512 registers: the configuration is excluded by construction, at no cost
// (the occupancy of these kernels is two waves per SIMD either way).  -DPNRF_WIDE_VGPRS=0 builds without the reservation (tools/wide_repro.py).
#ifndef PNRF_WIDE_VGPRS
#define PNRF_WIDE_VGPRS 256
#endif
template <int NW>
__device__ __forceinline__ void own_the_simd() {
#if PNRF_WIDE_VGPRS == 256
  if constexpr (NW == 8) asm volatile("; a wide fused workgroup owns its SIMDs' register file" ::: "v255");
#elif PNRF_WIDE_VGPRS == 248
  if constexpr (NW == 8) asm volatile("; reserve 248" ::: "v247");
#elif PNRF_WIDE_VGPRS == 240
  if constexpr (NW == 8) asm volatile("; reserve 240" ::: "v239");
#endif
}

template <int NW>
__device__ __forceinline__ void young_half_priority() {
  if (NW == 8 && PNRF_YOUNG_PRIO) {
    if (__builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(PNRF_YOUNG_PRIO);
#if PNRF_OLD_PRIO
    else __builtin_amdgcn_s_setprio(PNRF_OLD_PRIO);
#endif
  }
}

__device__ __forceinline__ bf16x8 pack_bf16(const float (&v)[8]) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (__bf16)v[j];
  return r;
}

__device__ __forceinline__ f16x8 pack_f16(const float (&v)[8]) {
  f16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (_Float16)v[j];
  return r;
}
// Operand type of the refine / NeRF stages.  The MFMA cycles are the same; fp16 keeps 11 significand bits where bf16 keeps 8 (the reference's
// own fast path runs FP16 TensorRT engines, trt_infer_v2.py).  fp16 has a finite range: a packed activation above 65 504 would be +inf and
// NaN one layer on.  The fp16 kernels therefore run with MODE.FP16_OVFL = 1 (set once per wave at kernel entry): an fp32 -> fp16 conversion
// that overflows then returns +-65 504 instead of +-inf, at no cost per activation (tools/fp16_ovfl_probe.hip: 1e6 -> 0x7bff, -1e6 -> 0xfbff
// on gfx950; tests/test_ops_gpu.py::test_fp16_operands_precision_and_saturation).
struct PrecBf16 {
  using v8 = bf16x8;
  static __device__ __forceinline__ void enter() {}
  static __device__ __forceinline__ v8 pack(const float (&v)[8]) { return pack_bf16(v); }
  static __device__ __forceinline__ int cvt_pk(float a, float b) {
    int pk;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(a), "v"(b));
    return pk;
  }
};
struct PrecF16 {
  using v8 = f16x8;
  static __device__ __forceinline__ void enter() { __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1); }      // hwreg(HW_REG_MODE, 23, 1) = FP16_OVFL
  static __device__ __forceinline__ v8 pack(const float (&v)[8]) {
    i32x4_t w;
#pragma unroll
    for (int d = 0; d < 4; ++d) w[d] = cvt_pk(v[2 * d], v[2 * d + 1]);
    return __builtin_bit_cast(v8, w);
  }
  // (not volatile: the operands of every conversion come from loads / MFMAs issued after enter(), so it cannot move in front of the mode
  // switch, and a volatile statement could not be interleaved with the engine's other asm statements)
  static __device__ __forceinline__ int cvt_pk(float a, float b) {
    int pk;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(a), "v"(b));
    return pk;
  }
};

// Pluecker moment of point p=o+t*d with unit direction hd, arithmetic un-fused like the
// reference's separate torch ops (trt.py:559-560 mul,add; helpers:630-631 cross).
__device__ __forceinline__ void moment(float ox, float oy, float oz, float dx, float dy, float dz, float t,
                                       float hx, float hy, float hz, float& m0, float& m1, float& m2) {
  const float px = ieee_add(ox, ieee_mul(dx, t)), py = ieee_add(oy, ieee_mul(dy, t)), pz = ieee_add(oz, ieee_mul(dz, t));
  cross_rn(px, py, pz, hx, hy, hz, m0, m1, m2);
}

// ------------------------------------------------------------------------------------------ sampler
// Diagnostic switch (tools/wide_repro.py): -DPNRF_FIXED_NHID compiles the Fern layer counts in instead of reading them from the launch arguments
#ifdef PNRF_FIXED_NHID
#define PNRF_NHID(runtime, fern) (fern)
#else
#define PNRF_NHID(runtime, fern) (runtime)
#endif
struct SamplerArgs {
  const void* blob; const float* bias; uint32_t nslots; int nbias;
  int nhid;                                        // hidden 256 -> 256 layers behind layer 0 (mmnetdepth - 1; the Fern configs: 5)
  int64_t n; int nbatch;
  const float* rays; const float* tvals;           // fused producer
  const float* x; const int* in0;                  // module-level producer
  float* depth_sorted; float* add_sorted; float* mul_sorted; int64_t* sort_idx; float* mm_rgb; float* depth_raw;
  float* y; const int* outmap; int head_act;       // module-level consumer
  // two-pass scheme: pass 1 (sampler_p1_kernel) appends the rays it cannot decide to list[] (count in counters[0]); pass 2
  // (sampler_h16_kernel with list != NULL) renders exactly those rays and leaves the count in counters[1]
  int* list; int* counters; const float* p1c; float kappa;
  // list_count: where the length of list[] is (pass 2: counters; pass 3: counters + 3).  sat_list (sampler_h16_kernel, NULL = off): rays with a
  // hidden activation at the fp16 limit (the conversion saturates at 65 504: MODE.FP16_OVFL) are appended here, count in counters[3], and
  // rendered again by the exact-fp32 kernel (pass 3, sampler_kernel<2> with list = sat_list)
  const int* list_count; int* sat_list;
};

#define PNRF_CSWAP(i, j)                                                              \
  {                                                                                   \
    const bool sw = (dep[i] > dep[j]) || (dep[i] == dep[j] && idx[i] > idx[j]);       \
    const float td = sw ? dep[j] : dep[i]; dep[j] = sw ? dep[i] : dep[j]; dep[i] = td; \
    const int ti = sw ? idx[j] : idx[i]; idx[j] = sw ? idx[i] : idx[j]; idx[i] = ti;   \
  }

// stable ascending sort of dep[8] with idx[8]: 19-comparator network on (value, index) keys, used by both sampler kernels
// (tests/test_abi_cpu.py checks this exact list exhaustively on 0/1 inputs and for stability)
#define PNRF_SORT8                                                       \
  PNRF_CSWAP(0, 1) PNRF_CSWAP(2, 3) PNRF_CSWAP(4, 5) PNRF_CSWAP(6, 7)  \
  PNRF_CSWAP(0, 2) PNRF_CSWAP(1, 3) PNRF_CSWAP(4, 6) PNRF_CSWAP(5, 7)  \
  PNRF_CSWAP(1, 2) PNRF_CSWAP(5, 6) PNRF_CSWAP(0, 4) PNRF_CSWAP(3, 7)  \
  PNRF_CSWAP(1, 5) PNRF_CSWAP(2, 6)                                     \
  PNRF_CSWAP(1, 4) PNRF_CSWAP(3, 6)                                     \
  PNRF_CSWAP(2, 4) PNRF_CSWAP(3, 5)                                     \
  PNRF_CSWAP(3, 4)

__device__ __forceinline__ float sel4(int q, float a, float b, float c, float d) {
  const float lo = q == 0 ? a : b, hi = q == 2 ? c : d;
  return q < 2 ? lo : hi;
}

// MODE 0: module-level (x -> y); 1: fused, full K=288 first layer; 2: fused, folded 6->256 first layer.
// 8 waves x 16 columns = 128 rays per workgroup batch, two waves per SIMD.
template <int MODE>
__global__ __launch_bounds__(512, 2) void sampler_kernel(SamplerArgs a) {
  constexpr bool FUSED = MODE != 0;
  constexpr int TPB = 512, NW = 8;
  // pass 3 of the sampler (MODE 2 with a list): the rays whose fp16 activations saturated in the split kernel.  Almost always none: leave
  // before anything is fetched
  int64_t total = a.n;
  int nbatch = a.nbatch;
  if (MODE == 2 && a.list) {
    const int cnt = __builtin_amdgcn_readfirstlane(*a.list_count);
    if (blockIdx.x == 0 && threadIdx.x == 0) a.counters[5] = cnt;              // kept for pnrf_ctx_sampler_stats
    if (cnt == 0) return;
    total = cnt; nbatch = (cnt + NW * 16 - 1) / (NW * 16);
  }
  constexpr int KS0 = MODE == 2 ? 4 * SF_KS4_0 : S_KS0;
  constexpr int KS4_0 = MODE == 2 ? SF_KS4_0 : S_KS4_0;
  constexpr int POS_H = MODE == 2 ? SF_POS_H : S_POS_H;
  constexpr int POS_LAST = MODE == 2 ? SF_POS_LAST : S_POS_LAST;
  constexpr int SLOTS_PAD = MODE == 2 ? SF_SLOTS_PAD : S_SLOTS_PAD;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* bias_lds = (float*)(smem + RING_BYTES);
  for (int i = threadIdx.x; i < a.nbias; i += TPB) bias_lds[i] = a.bias[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 15, q = lane >> 4;
  WStream<NW> st;
  st.init(a.blob, a.nslots, smem);
  st.prologue();
  own_the_simd<NW>();
  young_half_priority<NW>();
  const char* ringlane = smem + lane * 16;
  const float* biaslane = bias_lds + 4 * q;

  for (int batch = blockIdx.x; batch < nbatch; batch += gridDim.x) {
    const int64_t pos = (int64_t)batch * (NW * 16) + wave * 16 + col;
    const bool valid = pos < total;
    const int64_t row = (MODE == 2 && a.list) ? (int64_t)a.list[valid ? pos : total - 1] : pos;
    const int64_t rr = (valid || (MODE == 2 && a.list)) ? row : a.n - 1;
    float B0[KS0];
    float near = 0.f, far = 1.f;
    if (FUSED) {
      const float* r = a.rays + rr * 11;
      const float ox = r[0], oy = r[1], oz = r[2], dx = r[3], dy = r[4], dz = r[5];
      near = r[6]; far = r[7];
      float hx, hy, hz;
      unit_dir(dx, dy, dz, hx, hy, hz);
      if (MODE == 2) {           // one Pluecker 6-vector (moment at t = 0) against the folded weights
        float m0, m1, m2;
        moment(ox, oy, oz, dx, dy, dz, 0.f, hx, hy, hz, m0, m1, m2);
        B0[0] = sel4(q, hx, hy, hz, m0);
        B0[1] = sel4(q, m1, m2, 0.f, 0.f);
        B0[2] = 0.f; B0[3] = 0.f;
      } else {                   // feature f = 6p + c of mm_input (trt.py:274-277); k-step kk holds f = 4kk + q
        auto feat = [&](int f) {
          const int pnt = f / 6, c = f % 6;
          if (c == 0) return hx;
          if (c == 1) return hy;
          if (c == 2) return hz;
          float m0, m1, m2;
          moment(ox, oy, oz, dx, dy, dz, a.tvals[pnt], hx, hy, hz, m0, m1, m2);
          return c == 3 ? m0 : (c == 4 ? m1 : m2);
        };
#pragma unroll
        for (int kk = 0; kk < S_KS0; ++kk) B0[kk % KS0] = sel4(q, feat(4 * kk), feat(4 * kk + 1), feat(4 * kk + 2), feat(4 * kk + 3));
      }
    } else {
      const float* xr = a.x + rr * S_IN;
#pragma unroll
      for (int kk = 0; kk < KS0; ++kk) B0[kk] = xr[a.in0[kk * 4 + q]];
    }

    // activations ping-pong between X and Y (fp32 accumulators are the next layer's B operand as they stand);
    // `pend` = raw accumulators of the previous layer's last tile, whose ELU is deferred into the next layer
    f32x4 X[NT16_HID], Y[NT16_HID], pend;
    auto hidden = [&](f32x4(&in)[NT16_HID], f32x4(&out)[NT16_HID], int l) {
      f32x4 np;
      layer_f32<S_KS4_H, NT16_HID, POS_H>(
          st, ringlane, biaslane + (1 + l) * W_HID, [&](int kk) { return in[kk >> 2][kk & 3]; },
          [&](int to, int r, float v) { out[to][r] = act_f32(v, ACT_ELU); },
          [&](int r) { in[NT16_HID - 1][r] = act_f32(pend[r], ACT_ELU); }, np);
      pend = np;
    };
    layer_f32<KS4_0, NT16_HID, 0>(
        st, ringlane, biaslane, [&](int kk) { return B0[kk]; }, [&](int to, int r, float v) { X[to][r] = act_f32(v, ACT_ELU); }, [](int) {}, pend);
    // ping-pong X -> Y -> X ...: pairs of layers, then the odd one; the output layer reads Y (an even count ends in X: moved over)
    const int nhid = PNRF_NHID(a.nhid, S_NHID);
    for (int l = 0; l + 1 < nhid; l += 2) {
      hidden(X, Y, l);
      hidden(Y, X, l + 1);
    }
    if (nhid & 1) hidden(X, Y, nhid - 1);
    else {
#pragma unroll
      for (int t = 0; t < NT16_HID; ++t) Y[t] = X[t];
    }
    f32x4 fin0, fin1;
    layer_f32<S_KS4_H, S_NT_LAST, POS_LAST>(
        st, ringlane, biaslane + (1 + nhid) * W_HID, [&](int kk) { return Y[kk >> 2][kk & 3]; },
        [&](int, int r, float v) { fin0[r] = v; }, [&](int r) { Y[NT16_HID - 1][r] = act_f32(pend[r], ACT_ELU); }, fin1);
#pragma unroll
    for (int i = 0; i < SLOTS_PAD; ++i) st.begin();
    const float vals[8] = {fin0[0], fin0[1], fin0[2], fin0[3], fin1[0], fin1[1], fin1[2], fin1[3]};

    if (!FUSED) {
      if (valid) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int o = a.outmap[(i >> 2) * 16 + 4 * q + (i & 3)];
          // head_act: depth = sigmoid(y[0:8]), rgb = sigmoid(y[24:27]) (helpers:1502-1505)
          if (o >= 0) a.y[row * S_OUT + o] = (a.head_act && (o < 8 || o >= 24)) ? sigmoid_f(vals[i]) : vals[i];
        }
      }
      continue;
    }
    // ---- fused epilogue.  Quarter 0 of a column holds the 8 depth logits, quarter 1 add, quarter 2 mul,
    // quarter 3 rgb (sampler_out).  Quarter 0 sorts; the permutation goes to the other quarters as a word.
    float dep[8];
    int idx[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { dep[i] = sigmoid_f(vals[i]); idx[i] = i; }
    if (q == 0 && valid && a.depth_raw) {
      float4* p = (float4*)(a.depth_raw + row * 8);
      p[0] = make_float4(dep[0], dep[1], dep[2], dep[3]);
      p[1] = make_float4(dep[4], dep[5], dep[6], dep[7]);
    }
    const float span = ieee_sub(far, near);
#pragma unroll
    for (int i = 0; i < 8; ++i) dep[i] = ieee_add(ieee_mul(dep[i], span), near);      // trt.py:631
    // stable ascending sort (19-comparator network on (value, index) keys)          // trt.py:632-635
    PNRF_SORT8
    uint32_t word = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) word |= (uint32_t)idx[i] << (3 * i);
    const uint32_t w0 = __shfl(word, col);      // quarter 0's permutation for this column, in all quarters
    float perm[8];                              // own values permuted like the depths (trt.py:634-635)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = (w0 >> (3 * i)) & 7;
      float m = vals[0];
#pragma unroll
      for (int jj = 1; jj < 8; ++jj) m = (k == jj) ? vals[jj] : m;
      perm[i] = m;
    }
    if (valid) {
      if (q == 0) {
        float4* p = (float4*)(a.depth_sorted + row * 8);
        p[0] = make_float4(dep[0], dep[1], dep[2], dep[3]);
        p[1] = make_float4(dep[4], dep[5], dep[6], dep[7]);
        if (a.sort_idx) {
#pragma unroll
          for (int i = 0; i < 8; ++i) a.sort_idx[row * 8 + i] = idx[i];
        }
      } else if (q == 3) {
        if (a.mm_rgb) {
          a.mm_rgb[row * 3 + 0] = sigmoid_f(vals[0]);
          a.mm_rgb[row * 3 + 1] = sigmoid_f(vals[1]);
          a.mm_rgb[row * 3 + 2] = sigmoid_f(vals[2]);
        }
      } else {
        float4* p = (float4*)((q == 1 ? a.add_sorted : a.mul_sorted) + row * 8);
        p[0] = make_float4(perm[0], perm[1], perm[2], perm[3]);
        p[1] = make_float4(perm[4], perm[5], perm[6], perm[7]);
      }
    }
  }
  st.drain();
  if (MODE == 2 && a.list && threadIdx.x == 0) {          // pass 3: the last workgroup leaves its counters at zero for the next call
    if (atomicAdd(a.counters + 4, 1) == (int)gridDim.x - 1) { a.counters[3] = 0; a.counters[4] = 0; }
  }
}

// Sampler in split fp16 (layer_h16x2): same producer geometry (16 columns per wave, lane quarter q), same fused epilogue;
// the hidden activations live as two fp16 planes per 32-feature k-step (hi, lo*2^11).  Fused path, folded first layer only.
__device__ __forceinline__ void split_h16(const float (&v)[8], f16x8& hi, f16x8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const _Float16 h = (_Float16)v[j];
    hi[j] = h;
    lo[j] = (_Float16)((v[j] - (float)h) * H16_LO_SCALE);
  }
}

// NW = 8: 128 rays per workgroup batch, two waves per SIMD.  NW = 4: 64 rays per batch, a SIMD per wave: the same instruction stream per
// wave — results are bit-identical — at half the batch latency, for calls with at most one batch per CU (ray chunks, the short list of pass 2).
// One workgroup per CU in both shapes (stage_shape).
template <int NW>
__global__ __launch_bounds__(64 * NW, 2) void sampler_h16_kernel(SamplerArgs a) {
  constexpr int TPB = 64 * NW;
  PrecF16::enter();               // packed activations saturate at +-65 504 instead of overflowing to inf (see PrecF16)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* bias_lds = (float*)(smem + RING_BYTES);
  WStream<NW> st;
  st.init(a.blob, a.nslots, smem);
  st.prologue();                  // the first slots are on their way while the bias table is fetched
  // the handle's bias table is shared with the exact-fp32 kernels (true scale); this kernel's stream is packed for log2(e)-scaled
  // activations (elu_scaled): the biases of the six ELU layers are scaled here, the output layer's stay as they are
  for (int i = threadIdx.x; i < a.nbias; i += TPB) bias_lds[i] = i < (1 + a.nhid) * W_HID ? a.bias[i] * LOG2E : a.bias[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 15, q = lane >> 4;
  own_the_simd<NW>();
  young_half_priority<NW>();
  const char* ringlane = smem + lane * 16;
  const float* biaslane = bias_lds + 4 * q;
  constexpr float INV = 1.f / H16_LO_SCALE;
  // pass 2 of the two-pass scheme: the rays to render are list[0 .. counters[0]) (written by pass 1, the previous launch on the stream)
  int64_t total = a.n;
  int nbatch = a.nbatch;
  if (a.list) {
    const int cnt = __builtin_amdgcn_readfirstlane(*a.list_count);
    total = cnt; nbatch = (cnt + NW * 16 - 1) / (NW * 16);
    if (blockIdx.x == 0 && threadIdx.x == 0) a.counters[1] = cnt;            // kept for pnrf_ctx_sampler_stats
  }

  for (int batch = blockIdx.x; batch < nbatch; batch += gridDim.x) {
    const int64_t pos = (int64_t)batch * (NW * 16) + wave * 16 + col;
    const bool valid = pos < total;
    const int64_t row = a.list ? (int64_t)a.list[valid ? pos : total - 1] : pos;
    const int64_t rr = (valid || a.list) ? row : a.n - 1;
    const float* r = a.rays + rr * 11;
    const float ox = r[0], oy = r[1], oz = r[2], dx = r[3], dy = r[4], dz = r[5];
    const float near = r[6], far = r[7];
    float hx, hy, hz, m0, m1, m2;
    unit_dir(dx, dy, dz, hx, hy, hz);
    moment(ox, oy, oz, dx, dy, dz, 0.f, hx, hy, hz, m0, m1, m2);
    f16x8 P0h, P0l;                       // layer-0 B operand: group 0 holds the Pluecker 6-vector, groups 1-3 padding
    {
      const float z = 0.f;
      const float v[8] = {q == 0 ? hx : z, q == 0 ? hy : z, q == 0 ? hz : z, q == 0 ? m0 : z, q == 0 ? m1 : z, q == 0 ? m2 : z, z, z};
      split_h16(v, P0h, P0l);
    }
    // activations ping-pong between X and Y: per 32-feature k-step one hi and one lo plane
    f16x8 Xh[SH_KS_H], Xl[SH_KS_H], Yh[SH_KS_H], Yl[SH_KS_H];
    f32x4 pm[2], pc[2];                   // pending (deferred) tile pair of the previous layer
    // largest packed high plane seen, as 16-bit integers: hidden activations are >= -log2(e), so only the positive limit 0x7bff (65 504, where
    // the conversion saturates) can be reached — one v_pk_max_i16 per pair of activations
    i16x2_t amax = {0, 0};
    // pair (t, p) of tile pair tp: registers 2p, 2p+1 of tile t -> dword 2t + p of k-step tp of the next layer's
    // planes.  Per activation: combine (v_fma), ELU on the log2(e) scale (v_exp, v_fma, v_med3), then per PAIR one v_cvt_pk_f16_f32 for the
    // high plane and per value v_mul (x 2^11) + v_fma_mix{lo,hi}_f16 for the low plane ((v - hi) 2^11 in one fused step: hi 2^11 and v 2^11 are
    // exact).  Written with the instructions spelled out: left to the compiler this came out as v_cvt_f32_f16 round trips and SLP-packed
    // v_pk_*_f32, which cost more beside MFMAs than the scalar forms (MI355X_MICROARCH.md, issue-cost table).
    auto store_piece = [&](f16x8(&dh)[SH_KS_H], f16x8(&dl)[SH_KS_H], int tp, int pcx, f32x4(&mn)[2], f32x4(&cr)[2]) {
      int t, p;
      float v0, v1;
      if constexpr (H16_PIECES == 8) {          // one activation per piece: the even one waits, activated, in its accumulator register
        t = pcx >> 2; p = (pcx >> 1) & 1;
        const int r = pcx & 3;
        const float v = elu_scaled(fmaf(cr[t][r], INV, mn[t][r]));
        if (!(r & 1)) { mn[t][r] = v; return; }
        v0 = mn[t][r - 1]; v1 = v;
      } else {
        t = pcx >> 1; p = pcx & 1;
        v0 = elu_scaled(fmaf(cr[t][2 * p], INV, mn[t][2 * p])); v1 = elu_scaled(fmaf(cr[t][2 * p + 1], INV, mn[t][2 * p + 1]));
      }
      int hi, lo;
      asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(v0), "v"(v1));
      amax = __builtin_elementwise_max(amax, __builtin_bit_cast(i16x2_t, hi));
      const float s0 = v0 * H16_LO_SCALE, s1 = v1 * H16_LO_SCALE, sc = H16_LO_SCALE;
      asm("v_fma_mixlo_f16 %0, -%1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(sc), "v"(s0));
      asm("v_fma_mixhi_f16 %0, -%1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(sc), "v"(s1));
      i32x4_t wh = __builtin_bit_cast(i32x4_t, dh[tp]), wl = __builtin_bit_cast(i32x4_t, dl[tp]);
      wh[2 * t + p] = hi; wl[2 * t + p] = lo;
      dh[tp] = __builtin_bit_cast(f16x8, wh); dl[tp] = __builtin_bit_cast(f16x8, wl);
    };
    auto hidden = [&](f16x8(&ih)[SH_KS_H], f16x8(&il)[SH_KS_H], f16x8(&oh)[SH_KS_H], f16x8(&ol)[SH_KS_H], int l) {
      f32x4 nm[2], nc[2];
      layer_h16x2<SH_KS_H, SH_NTP_H, SH_POS_H>(
          st, ringlane, biaslane + (1 + l) * W_HID, [&](int ks, int pl) { return pl == 0 ? ih[ks] : il[ks]; },
          [&](int tp, int pcx, f32x4(&mn)[2], f32x4(&cr)[2]) { store_piece(oh, ol, tp, pcx, mn, cr); },
          [&](int pcx) { store_piece(ih, il, SH_NTP_H - 1, pcx, pm, pc); }, nm, nc);
#pragma unroll
      for (int t = 0; t < 2; ++t) { pm[t] = nm[t]; pc[t] = nc[t]; }
    };
    layer_h16x2<1, SH_NTP_H, 0>(
        st, ringlane, biaslane, [&](int, int pl) { return pl == 0 ? P0h : P0l; },
        [&](int tp, int pcx, f32x4(&mn)[2], f32x4(&cr)[2]) { store_piece(Xh, Xl, tp, pcx, mn, cr); }, [](int) {}, pm, pc);
    const int nhid = PNRF_NHID(a.nhid, S_NHID);
    for (int l = 0; l + 1 < nhid; l += 2) {
      hidden(Xh, Xl, Yh, Yl, l);
      hidden(Yh, Yl, Xh, Xl, l + 1);
    }
    if (nhid & 1) hidden(Xh, Xl, Yh, Yl, nhid - 1);
    else {                                  // an even count ends in X: the output layer reads Y
#pragma unroll
      for (int k = 0; k < SH_KS_H; ++k) { Yh[k] = Xh[k]; Yl[k] = Xl[k]; }
    }
    f32x4 fm[2], fc[2];
    layer_h16x2<SH_KS_H, 1, SH_POS_LAST>(
        st, ringlane, biaslane + (1 + nhid) * W_HID, [&](int ks, int pl) { return pl == 0 ? Yh[ks] : Yl[ks]; },
        [&](int, int, f32x4(&)[2], f32x4(&)[2]) {}, [&](int pcx) { store_piece(Yh, Yl, SH_NTP_H - 1, pcx, pm, pc); }, fm, fc);
#pragma unroll
    for (int i = 0; i < SH_SLOTS_PAD; ++i) st.begin();
    float vals[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) vals[i] = fmaf(fc[i >> 2][i & 3], INV, fm[i >> 2][i & 3]);

    // ---- fused epilogue.  Quarter 0 of a column holds the 8 depth logits, quarter 1 add, quarter 2 mul,
    // quarter 3 rgb (sampler_out).  Quarter 0 sorts; the permutation goes to the other quarters as a word.
    float dep[8];
    int idx[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { dep[i] = sigmoid_f(vals[i]); idx[i] = i; }
    if (q == 0 && valid && a.depth_raw) {
      float4* p = (float4*)(a.depth_raw + row * 8);
      p[0] = make_float4(dep[0], dep[1], dep[2], dep[3]);
      p[1] = make_float4(dep[4], dep[5], dep[6], dep[7]);
    }
    const float span = ieee_sub(far, near);
#pragma unroll
    for (int i = 0; i < 8; ++i) dep[i] = ieee_add(ieee_mul(dep[i], span), near);      // trt.py:631
    // stable ascending sort (19-comparator network on (value, index) keys)          // trt.py:632-635
    PNRF_SORT8
    uint32_t word = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) word |= (uint32_t)idx[i] << (3 * i);
    const uint32_t w0 = __shfl(word, col);      // quarter 0's permutation for this column, in all quarters
    float perm[8];                              // own values permuted like the depths (trt.py:634-635)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = (w0 >> (3 * i)) & 7;
      float m = vals[0];
#pragma unroll
      for (int jj = 1; jj < 8; ++jj) m = (k == jj) ? vals[jj] : m;
      perm[i] = m;
    }
    if (valid) {
      if (q == 0) {
        float4* p = (float4*)(a.depth_sorted + row * 8);
        p[0] = make_float4(dep[0], dep[1], dep[2], dep[3]);
        p[1] = make_float4(dep[4], dep[5], dep[6], dep[7]);
        if (a.sort_idx) {
#pragma unroll
          for (int i = 0; i < 8; ++i) a.sort_idx[row * 8 + i] = idx[i];
        }
      } else if (q == 3) {
        if (a.mm_rgb) {
          a.mm_rgb[row * 3 + 0] = sigmoid_f(vals[0]);
          a.mm_rgb[row * 3 + 1] = sigmoid_f(vals[1]);
          a.mm_rgb[row * 3 + 2] = sigmoid_f(vals[2]);
        }
      } else {
        float4* p = (float4*)((q == 1 ? a.add_sorted : a.mul_sorted) + row * 8);
        p[0] = make_float4(perm[0], perm[1], perm[2], perm[3]);
        p[1] = make_float4(perm[4], perm[5], perm[6], perm[7]);
      }
    }
    // rays with a saturated hidden activation (any of the column's four lane groups): to the exact-fp32 pass
    if (a.sat_list) {
      int sat = (amax[0] >= 0x7bff) | (amax[1] >= 0x7bff);
      sat |= __shfl_xor(sat, 16);
      sat |= __shfl_xor(sat, 32);
      const bool flag = q == 0 && valid && sat;
      const uint64_t m = __ballot(flag);
      if (m) {
        int base = 0;
        const int leader = __builtin_ctzll(m);
        if (lane == leader) base = atomicAdd(a.counters + 3, __builtin_popcountll(m));
        base = __shfl(base, leader);
        if (flag) a.sat_list[base + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = (int)row;
      }
    }
  }
  st.drain();
  // two-pass scheme: the last workgroup to finish leaves the counters at zero for the next call on this workspace (every workgroup has
  // read counters[0] by now; stream order makes the stores visible to the next launch) — no memset on the stream per call
  if (a.list && threadIdx.x == 0) {
    if (atomicAdd(a.counters + 2, 1) == (int)gridDim.x - 1) { a.counters[0] = 0; a.counters[2] = 0; }
  }
}

// ------------------------------------------------------------------------------------------ sampler, pass 1 of two
// The sampler's outputs are SORTED, so its products must be fp32-grade wherever two of a ray's eight depths are close — and only there.
// Pass 1 runs the net for every ray in plain fp16 (one v_mfma_f32_32x32x16_f16 per product instead of the three of layer_h16x2; the folded
// 6 -> 256 first layer stays split, it is 24 MFMAs) on the refine net's engine, together with a per-ray bound on the standard deviation of
// its own rounding error, and flags a ray when some adjacent sorted gap is not larger than kappa x that bound; the flagged rays are
// re-rendered by sampler_h16_kernel (pass 2), which overwrites their rows.  Error model (tools/sampler_twopass_model.py):
//   rounding an operand to fp16 is a zero-mean error of variance <= c x^2, c = 2^-22 / 3, independent between operands;
//   with S_l = |x_l|^2 and V_l = sum_i var(dx_l,i):  V_{l+1} <= C_l (2 c S_l + V_l),  C_l = max_j sum_i W_l[i,j]^2  (|ELU'| <= 1);
//   var(d logit_k) <= M (2 c S_L + V_L),  M = max_{k,j} W_out[k,j]^2;   std(d depth_k) = span d_k (1 - d_k) sqrt(var(d logit_k)).
// S_l is accumulated from the activations themselves (one v_fma per activation), C_l and M come from the packer (SamplerArgs::p1c).
__device__ __forceinline__ int cvt_pk_f16(float a, float b) {        // one v_cvt_pk_f16_f32 (round to nearest even; a in the low half)
  int pk;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(a), "v"(b));
  return pk;
}
// deferred hidden-layer epilogue of pass 1, one activation per piece (as HiddenEpi<.., 16>): ELU on the log2(e) scale, |x|^2, fp16 pack
struct P1Epi {
  f16x8 (&Bn)[KS_HID];
  float& ssq;
  __device__ __forceinline__ void operator()(int to, int pc, f32x16 (&acc)[1]) const {
    const float v = elu_scaled(acc[0][pc]);
    acc[0][pc] = v;
    ssq = fmaf(v, v, ssq);
    if (pc & 1) {
      f16x8& frag = Bn[2 * to + pc / 8];
      i32x4_t w = __builtin_bit_cast(i32x4_t, frag);
      w[(pc % 8) / 2] = cvt_pk_f16(acc[0][pc - 1], v);
      frag = __builtin_bit_cast(f16x8, w);
    }
  }
};

template <int NW>
__global__ __launch_bounds__(64 * NW, 2) void sampler_p1_kernel(SamplerArgs a) {
  constexpr int TPB = 64 * NW;
  PrecF16::enter();
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* bias_lds = (float*)(smem + RING_BYTES);
  WStream<NW> st;
  st.init(a.blob, a.nslots, smem);
  st.prologue();
  for (int i = threadIdx.x; i < a.nbias; i += TPB) bias_lds[i] = a.bias[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 31, h = lane >> 5;
  own_the_simd<NW>();
  young_half_priority<NW>();
  const char* ringlane = smem + lane * 16;
  const float* biaslane = bias_lds + h * 16;
  constexpr float INV = 1.f / H16_LO_SCALE;
  constexpr float C2 = 2.f * 7.947285970052083e-08f;                  // 2 c, c = 2^-22 / 3
  const float m_out = a.p1c[0];              // p1c: [0] output-layer constant, [1 + l] C_l of hidden layer l

  for (int batch = blockIdx.x; batch < a.nbatch; batch += gridDim.x) {
    const int64_t row = (int64_t)batch * (NW * 32) + wave * 32 + col;
    const bool valid = row < a.n;
    const int64_t rr = valid ? row : a.n - 1;
    const float* r = a.rays + rr * 11;
    const float ox = r[0], oy = r[1], oz = r[2], dx = r[3], dy = r[4], dz = r[5];
    const float near = r[6], far = r[7];
    float hx, hy, hz, m0, m1, m2;
    unit_dir(dx, dy, dz, hx, hy, hz);
    moment(ox, oy, oz, dx, dy, dz, 0.f, hx, hy, hz, m0, m1, m2);
    f16x8 P0h, P0l;                       // layer-0 B operand: half 0 holds the Pluecker 6-vector (split hi / lo 2^11), half 1 padding
    {
      const float z = 0.f;
      const float v[8] = {h == 0 ? hx : z, h == 0 ? hy : z, h == 0 ? hz : z, h == 0 ? m0 : z, h == 0 ? m1 : z, h == 0 ? m2 : z, z, z};
      split_h16(v, P0h, P0l);
    }
    f16x8 Bo[KS_HID], Bn[KS_HID];
    f32x16 pend[1];
    float sq_a = 0.f, sq_b = 0.f, V = 0.f;   // |x_l|^2 of the layer being written / the one before it (per-lane partial sums), error variance
    // an activation at the fp16 limit (the pack saturates at 65 504) shows in its layer's |x|^2 >= 65 504^2: such a ray is undecided whatever
    // its gaps (its pass-1 values are wrong; the split kernel and, if that saturates too, the exact-fp32 kernel render it)
    constexpr float OVF2 = 65504.f * 65504.f;
    bool ovf = false;

    // ---- layer 0: 8 tiles x (W_hi P_hi | W_hi P_lo + W_lo P_hi), one slot; tile t's activation runs behind tile t+1's MFMAs
    {
      st.wait_slot();
      const P1Epi epi{Bn, sq_a};
#pragma unroll
      for (int to = 0; to < NT_HID; ++to) {
        const f16x8 ahi = *(const f16x8*)(ringlane + (2 * to) * FRAG_BYTES), alo = *(const f16x8*)(ringlane + (2 * to + 1) * FRAG_BYTES);
        f32x16 mn, cr;
        {
          const f32x4* bp = (const f32x4*)(biaslane + to * 32);
          const f32x4 b0 = bp[0], b1 = bp[1], b2 = bp[2], b3 = bp[3];
#pragma unroll
          for (int i = 0; i < 4; ++i) { mn[i] = b0[i]; mn[4 + i] = b1[i]; mn[8 + i] = b2[i]; mn[12 + i] = b3[i]; }
#pragma unroll
          for (int i = 0; i < 16; ++i) cr[i] = 0.f;
        }
        mn = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, P0h, mn, 0, 0, 0);
        cr = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, P0l, cr, 0, 0, 0);
        cr = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, P0h, cr, 0, 0, 0);
        if (to == 0) st.slot_issue(0);
        if (to > 0) {
#pragma unroll
          for (int pc = 0; pc < 16; ++pc) epi(to - 1, pc, pend);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) pend[0][i] = fmaf(cr[i], INV, mn[i]);
      }
    }
    // ---- hidden layers (ping-pong Bn -> Bo -> Bn ...); the last tile of a layer is activated in the head of the next one (pre1), so
    // |x_l|^2 is complete when layer l's call returns: fold it into V there and hand the accumulator to layer l+1's outputs
    auto hidden = [&](f16x8(&in)[KS_HID], f16x8(&out)[KS_HID], int l, float& sq_in, float& sq_out) {
      f32x16 np[1];
      layer_bf16<1, KS_HID, NT_HID, P1_POS_H, 16, true>(st, ringlane, biaslane + (1 + l) * W_HID, [&](int, int ks) { return in[ks]; }, P1Epi{out, sq_out},
                                                         [&](int pc) { P1Epi{in, sq_in}(NT_HID - 1, pc, pend); }, np);
      pend[0] = np[0];
      ovf |= sq_in >= OVF2;
      V = a.p1c[1 + l] * fmaf(C2, sq_in, V);
      sq_in = 0.f;
    };
    const int nhid = PNRF_NHID(a.nhid, S_NHID);
    for (int l = 0; l + 1 < nhid; l += 2) {
      hidden(Bn, Bo, l, sq_a, sq_b);
      hidden(Bo, Bn, l + 1, sq_b, sq_a);
    }
    if (nhid & 1) hidden(Bn, Bo, nhid - 1, sq_a, sq_b);
    else {                                  // an even count ends in Bn (its last tile still pending, its |x|^2 in sq_a): the output layer reads Bo / sq_b
#pragma unroll
      for (int k = 0; k < KS_HID; ++k) Bo[k] = Bn[k];
      sq_b = sq_a;
    }
    f32x16 fin[1];
    layer_bf16<1, KS_HID, 1, P1_POS_LAST, 16, true>(st, ringlane, biaslane + (1 + nhid) * W_HID, [&](int, int ks) { return Bo[ks]; },
                                                     [&](int, int, f32x16(&)[1]) {}, [&](int pc) { P1Epi{Bo, sq_b}(NT_HID - 1, pc, pend); }, fin);
#pragma unroll
    for (int i = 0; i < P1_SLOTS_PAD; ++i) st.begin();
    ovf |= sq_b >= OVF2;
    ovf |= (bool)__shfl_xor((int)ovf, 32);
    // variance bound of a depth logit: both halves of a column hold partial sums over their rows
    float U = fmaf(C2, sq_b, V);
    U = ieee_add(U, __shfl_xor(U, 32));
    const float sd = sqrtf(m_out * U);

    // ---- epilogue.  Half 0: registers 0-7 depth logits, 8-15 add; half 1: 0-7 mul, 8-10 rgb (sampler_p1_out)
    float vals[8], oth[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { vals[i] = fin[0][i]; oth[i] = fin[0][8 + i]; }
    float dep[8];
    int idx[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { dep[i] = sigmoid_f(vals[i]); idx[i] = i; }
    if (h == 0 && valid && a.depth_raw) {
      float4* p = (float4*)(a.depth_raw + row * 8);
      p[0] = make_float4(dep[0], dep[1], dep[2], dep[3]);
      p[1] = make_float4(dep[4], dep[5], dep[6], dep[7]);
    }
    const float span = ieee_sub(far, near);
#pragma unroll
    for (int i = 0; i < 8; ++i) dep[i] = ieee_add(ieee_mul(dep[i], span), near);      // trt.py:631
    PNRF_SORT8                                                                             // trt.py:632-635
    // decidable?  every adjacent gap must exceed kappa x (std bound of the two depths) + an fp32 round-off allowance
    bool undecided = ovf;
    {
      const float inv_span = 1.f / span;
      float sdev[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float u = (dep[i] - near) * inv_span;
        sdev[i] = (dep[i] - near) * (1.f - u) * sd;                     // span d (1 - d) sd
      }
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const float thr = fmaf(a.kappa, sdev[i] + sdev[i + 1], 2e-6f * span);
        undecided |= !((dep[i + 1] - dep[i]) > thr);                     // also true for NaN / inf (fp16 overflow somewhere in the net)
      }
    }
    uint32_t word = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) word |= (uint32_t)idx[i] << (3 * i);
    const uint32_t w0 = __shfl(word, col);      // half 0's permutation for this column, in both halves
    // half 0 permutes add, half 1 mul (trt.py:634-635)
    float perm[8];
    {
      float src[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) src[i] = h == 0 ? oth[i] : vals[i];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = (w0 >> (3 * i)) & 7;
        float m = src[0];
#pragma unroll
        for (int jj = 1; jj < 8; ++jj) m = (k == jj) ? src[jj] : m;
        perm[i] = m;
      }
    }
    if (valid) {
      if (h == 0) {
        float4* p = (float4*)(a.depth_sorted + row * 8);
        p[0] = make_float4(dep[0], dep[1], dep[2], dep[3]);
        p[1] = make_float4(dep[4], dep[5], dep[6], dep[7]);
        float4* pa = (float4*)(a.add_sorted + row * 8);
        pa[0] = make_float4(perm[0], perm[1], perm[2], perm[3]);
        pa[1] = make_float4(perm[4], perm[5], perm[6], perm[7]);
        if (a.sort_idx) {
#pragma unroll
          for (int i = 0; i < 8; ++i) a.sort_idx[row * 8 + i] = idx[i];
        }
      } else {
        float4* pm = (float4*)(a.mul_sorted + row * 8);
        pm[0] = make_float4(perm[0], perm[1], perm[2], perm[3]);
        pm[1] = make_float4(perm[4], perm[5], perm[6], perm[7]);
        if (a.mm_rgb) {
          a.mm_rgb[row * 3 + 0] = sigmoid_f(oth[0]);
          a.mm_rgb[row * 3 + 1] = sigmoid_f(oth[1]);
          a.mm_rgb[row * 3 + 2] = sigmoid_f(oth[2]);
        }
      }
    }
    // ---- compact the undecided rays of this wave into list[] (one atomic per wave)
    {
      const bool flag = h == 0 && valid && undecided;
      const uint64_t m = __ballot(flag);
      if (m) {
        int base = 0;
        const int leader = __builtin_ctzll(m);
        if (lane == leader) base = atomicAdd(a.counters, __builtin_popcountll(m));
        base = __shfl(base, leader);
        if (flag) a.list[base + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = (int)row;
      }
    }
  }
  st.drain();
}

// ------------------------------------------------------------------------------------------ bf16 nets
// Deferred hidden-layer epilogue, one piece at a time: piece pc = accumulator registers 8pc..8pc+7 of tile
// `to` -> activation -> packed bf16 B fragment of k-step 2*to+pc of the next layer.
__device__ __forceinline__ int cvt_pk_bf16(float a, float b) {      // one v_cvt_pk_bf16_f32 for the pair (a in the low half)
  int pk;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(a), "v"(b));
  return pk;
}
// PIECES = 2: piece pc = registers 8pc..8pc+7 = one whole B fragment; PIECES = 8: registers 2pc, 2pc+1 = one dword of a fragment; PIECES = 16: register pc.
template <int NCB, int ACT, int PIECES = 2, class P = PrecBf16>
struct HiddenEpi {
  typename P::v8 (&Bn)[NCB][KS_HID];
  __device__ __forceinline__ void operator()(int to, int pc, f32x16 (&acc)[NCB]) const {
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      if constexpr (PIECES == 2) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = act_fast(acc[cb][8 * pc + j], ACT);
        Bn[cb][2 * to + pc] = P::pack(v);
      } else if constexpr (PIECES == 16) {      // one activation per piece: the even one waits, activated, in its accumulator register
        acc[cb][pc] = act_fast(acc[cb][pc], ACT);
        if (pc & 1) {
          typename P::v8& frag = Bn[cb][2 * to + pc / 8];
          i32x4_t w = __builtin_bit_cast(i32x4_t, frag);
          w[(pc % 8) / 2] = P::cvt_pk(acc[cb][pc - 1], acc[cb][pc]);
          frag = __builtin_bit_cast(typename P::v8, w);
        }
      } else {
        constexpr int E = 16 / PIECES;
        typename P::v8& frag = Bn[cb][2 * to + (E * pc) / 8];
        i32x4_t w = __builtin_bit_cast(i32x4_t, frag);
#pragma unroll
        for (int d = 0; d < E / 2; ++d)
          w[((E * pc) % 8) / 2 + d] = P::cvt_pk(act_fast(acc[cb][E * pc + 2 * d], ACT), act_fast(acc[cb][E * pc + 2 * d + 1], ACT));
        frag = __builtin_bit_cast(typename P::v8, w);
      }
    }
  }
};

#ifdef PNRF_DEBUG_EHEAD
// Diagnostic build only (tools/coresidency_repro.py): the fused refine epilogue records, per (ray, half), the ray / depth rows it holds in registers
// since the batch head + where it ran, keyed by the ray's index in the frame (rays_base = first row of the frame's ray tensor).
__device__ float* g_dbg_ehead;
__device__ const float* g_dbg_rays_base;
constexpr int DBG_STRIDE = 128;        // floats per (ray, half): [0,16) e_d / e_ray, [16,24) placement, [24,32) xor of the B operand after every layer, [32,48) last accumulators, [48,120) inputs
template <int CNT, class V, int NK>
__device__ __forceinline__ int dbg_xor(const V (&B)[NK]) {
  int x = 0;
#pragma unroll
  for (int k = 0; k < CNT; ++k) {
    const i32x4_t w = __builtin_bit_cast(i32x4_t, B[k]);
    x ^= w[0] ^ w[1] ^ w[2] ^ w[3];
  }
  return x;
}
extern "C" int pnrf_debug_set_ehead(float* buf, const float* rays_base) {
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_ehead), &buf, sizeof(buf));
  if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_rays_base), &rays_base, sizeof(rays_base));
  return (int)e;
}
#endif
struct RefineArgs {
  const void* blob; const float* bias; uint32_t nslots; int nbias;
  int nhid, nb;                                     // hidden 256 -> 256 layers behind layer 0 (mmnetdepth - 1; Fern: 5); neighbour views (num_neighbor; Fern: 4)
  int64_t n; int nbatch;
  const float* x;                                   // refine_in [n,144] (HEAD = 0)
  const float* or_rays; const float4* img4; const float* proj; int Hf, Wf; float eps;   // HEAD = 1: the projection runs in the kernel
  const float* rays; const float* depth_sorted;     // fused consumer
  float* z; float* pts;
  const float* jitter; int jitter_dir; float* rgb0;  // training mode: depth jitter [n,8] (>= 0), its direction, refine rgb head [n,3]
  float* y; const int* outmap; int head_act;        // module-level consumer
};

// MODE 0: module-level (x -> y); 1: fused inference epilogue; 2: fused training-time epilogue (depth jitter, refine rgb head)
// HEAD 0: the 144 inputs of a ray come from refine_in [n,144] in memory; 1: they are produced in the batch head — neighbour projection +
// bilinear colour fetch + sample Pluecker (run_S_eS_eN_alter_trt.py:637-661), each lane for its own two views and four samples
// (refine_in0): no [n,144] round trip through HBM, one kernel launch less per frame.
// NV: neighbour views per lane half = ceil(num_neighbor / 2) (Fern: 2).  A lane half holds 24 NV colour values + 24 Pluecker values = 3 NV + 3
// k-steps of layer 0 (refine_in0_nv); a view index >= a.nb is padding (zero weights in the stream; its inputs are fetched from the last real view).
template <int NCB, int NW, int MODE, int HEAD = 0, class P = PrecBf16, int NV = 2>
__global__ __launch_bounds__(64 * NW, NCB == 1 ? 2 : 1) void refine_kernel(RefineArgs a) {
  constexpr bool FUSED = MODE != 0;
  using RL = RefineL0<NV>;
  constexpr int KS0 = RL::KS0;
  static_assert(HEAD == 0 || (NCB == 1 && MODE == 1), "the projecting head exists for the fused inference stage");
  constexpr int TPB = 64 * NW;
  constexpr bool F16 = std::is_same<P, PrecF16>::value;
  using v8 = typename P::v8;
  P::enter();
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* bias_lds = (float*)(smem + RING_BYTES);
  WStream<NW> st;
  st.init(a.blob, a.nslots, smem);
  st.prologue();
  for (int i = threadIdx.x; i < a.nbias; i += TPB) bias_lds[i] = a.bias[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 31, h = lane >> 5;
  constexpr int COLS = 32 * NCB;
  own_the_simd<NW>();
  young_half_priority<NW>();
  const char* ringlane = smem + lane * 16;
  const float* biaslane = bias_lds + h * 16;

  // Batch order.  HEAD = 1 fetches the neighbour images: workgroups b and b + 8 share an XCD (its L2) under the round-robin placement the
  // dispatcher is observed to use, so the workgroups of residue x walk the x-th EIGHTH of the batches — one band of the frame, whose samples
  // project into one band of every neighbour image: each L2 fetches its band instead of all 48.8 MB (round 3: 503 MB of fetches per frame,
  // 406 MB of them the images once per XCD).  Speed only: any placement gives the same rows.  Plain grid-stride order otherwise.
  int jbase = 0, j0 = (int)blockIdx.x, jstep = (int)gridDim.x, jend = a.nbatch;
  if (HEAD == 1 && (gridDim.x & 7) == 0 && a.nbatch >= (int)gridDim.x) {
    const int per = (a.nbatch + 7) >> 3;
    jbase = (int)(blockIdx.x & 7) * per; j0 = (int)(blockIdx.x >> 3); jstep = (int)(gridDim.x >> 3);
    jend = a.nbatch - jbase < per ? a.nbatch - jbase : per;
  }
  for (int j = j0; j < jend; j += jstep) {
    const int batch = jbase + j;
    int64_t row[NCB];
    bool valid[NCB];
    v8 Bo[NCB][KS_HID], Bn[NCB][KS_HID];
    float e_ray[NCB][8];         // inputs of the fused epilogue, fetched with the batch's features (see nerf_kernel's RawIn)
    float4 e_d0[NCB], e_d1[NCB];
    static_for<NCB>([&](auto cbc) {
      constexpr int cb = decltype(cbc)::value;
      row[cb] = (int64_t)batch * (NW * COLS) + wave * COLS + cb * 32 + col;
      valid[cb] = row[cb] < a.n;
      if (MODE != 0) {
        const int64_t rr = valid[cb] ? row[cb] : a.n - 1;
        const float* r = a.rays + rr * 11;
#pragma unroll
        for (int i = 0; i < 8; ++i) e_ray[cb][i] = r[i];
        e_d0[cb] = *(const float4*)(a.depth_sorted + rr * 8); e_d1[cb] = *(const float4*)(a.depth_sorted + rr * 8 + 4);
      }
      if constexpr (HEAD == 0) {
        // refine_in0: a lane's 72 inputs are two contiguous runs of the natural row — the colours of its two views [48 + 48h, 96 + 48h) and
        // the Pluecker values of its four samples [24h, 24h + 24): 18 aligned 16-byte loads
        const float* xr = a.x + (valid[cb] ? row[cb] : a.n - 1) * (48 + 24 * a.nb);
        static_assert(refine_in0(0, 1, 0) == 96 && refine_in0(5, 1, 7) == 143 && refine_in0(6, 1, 0) == 24 && refine_in0(8, 0, 7) == 23, "runs of refine_in0");
        static_assert(refine_in0_nv(2, 4, 0, 1, 0) == 96 && refine_in0_nv(2, 4, 5, 1, 7) == 143 && refine_in0_nv(2, 4, 6, 1, 0) == 24 && refine_in0_nv(2, 3, 3, 1, 0) == -1, "refine_in0_nv");
        const float4* xp = (const float4*)(xr + 24 * h);
#pragma unroll
        for (int ks = 0; ks < KS0; ++ks) {
          float4 lo, hi;
          if (ks < 3 * NV) {                 // colours of view NV h + ks / 3 (a view beyond nb: zeros — its row does not hold it)
            const int view = NV * h + ks / 3;
            const float4* xc = (const float4*)(xr + 48 + 24 * (view < a.nb ? view : 0)) + 2 * (ks % 3);
            lo = xc[0]; hi = xc[1];
            if (view >= a.nb) { lo = make_float4(0.f, 0.f, 0.f, 0.f); hi = lo; }
          } else {
            lo = xp[2 * (ks - 3 * NV)]; hi = xp[2 * (ks - 3 * NV) + 1];
          }
          const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
          Bo[cb][ks] = P::pack(v);
        }
      } else {
        // lane (ray, h): views 2h, 2h+1 x 8 samples (48 colours) + Pluecker of samples 4h..4h+3 (24 values), in the order of refine_in0
        const int64_t rr = valid[cb] ? row[cb] : a.n - 1;
        const float* orr = a.or_rays + rr * 11;
        const float o0 = orr[0], o1 = orr[1], o2 = orr[2], w0 = orr[3], w1 = orr[4], w2 = orr[5];
        const float dn8[8] = {e_d0[cb].x, e_d0[cb].y, e_d0[cb].z, e_d0[cb].w, e_d1[cb].x, e_d1[cb].y, e_d1[cb].z, e_d1[cb].w};
        const int plane = a.Hf * a.Wf;
        float feat[8 * KS0];
        // 8 NV projections per lane (view vv = p / 8, sample p % 8) in a software pipeline: the four texel fetches of projection p + D are issued
        // before projection p is blended, so D projections (4 D x 16 B per lane) are in flight.  The fences keep the compiler from hoisting
        // all 64 fetches to the top of the batch (256 registers of texels: it spilled 187).
        constexpr int D = 4;
        float wt[D][4];
        float4 tx[D][4];
        ViewRay vr[NV];
        int view[NV];                    // a view index beyond nb - 1 is padding (zero weights): it projects into the last real view instead
        float z3d[8];
#pragma unroll
        for (int vv = 0; vv < NV; ++vv) {
          view[vv] = NV * h + vv < a.nb ? NV * h + vv : a.nb - 1;
          float M[12];
#pragma unroll
          for (int i = 0; i < 12; ++i) M[i] = a.proj[view[vv] * 12 + i];
          vr[vv] = view_ray(o0, o1, o2, w0, w1, w2, M);
        }
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) z3d[s8] = __builtin_amdgcn_rcpf(1.f - dn8[s8] - a.eps);                     // trt.py:637
        // texel addresses as 32-bit byte offsets from the uniform image base (4 views x Hf x Wf x 16 B < 2^31, checked by the entry point)
        const char* imb = (const char*)a.img4;
        auto texel = [&](uint32_t view_off, uint32_t idx) { return *(const float4*)(imb + ((view_off + idx) << 4)); };
        auto issue = [&](auto pc) {
          constexpr int p = decltype(pc)::value, vv = p / 8, sl = p % D;
          const uint32_t vo = (uint32_t)(view[vv] * plane);
          const Taps t = project_taps_fast(vr[vv], z3d[p % 8], a.Hf, a.Wf);
          tx[sl][0] = texel(vo, t.i00); tx[sl][1] = texel(vo, t.i01); tx[sl][2] = texel(vo, t.i10); tx[sl][3] = texel(vo, t.i11);
          wt[sl][0] = t.a00; wt[sl][1] = t.a01; wt[sl][2] = t.a10; wt[sl][3] = t.a11;
        };
        auto blend = [&](auto pc) {
          constexpr int p = decltype(pc)::value, sl = p % D;
          Taps t;
          t.a00 = wt[sl][0]; t.a01 = wt[sl][1]; t.a10 = wt[sl][2]; t.a11 = wt[sl][3];
          feat[3 * p + 0] = blend4(tx[sl][0].x, tx[sl][1].x, tx[sl][2].x, tx[sl][3].x, t);
          feat[3 * p + 1] = blend4(tx[sl][0].y, tx[sl][1].y, tx[sl][2].y, tx[sl][3].y, t);
          feat[3 * p + 2] = blend4(tx[sl][0].z, tx[sl][1].z, tx[sl][2].z, tx[sl][3].z, t);
        };
        static_for<D>([&](auto pc) { issue(pc); });
        __builtin_amdgcn_sched_barrier(0);
        static_for<8 * NV>([&](auto pc) {
          constexpr int p = decltype(pc)::value;
          blend(pc);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (p + D < 8 * NV) {
            issue(std::integral_constant<int, p + D>{});
            __builtin_amdgcn_sched_barrier(0);
          }
        });
        {                                                                                          // trt.py:656-658
          const float* r = e_ray[cb];
          float hx, hy, hz;
          unit_dir(r[3], r[4], r[5], hx, hy, hz);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float dn = h ? dn8[4 + u] : dn8[u];
            float m0, m1, m2;
            moment(r[0], r[1], r[2], r[3], r[4], r[5], dn, hx, hy, hz, m0, m1, m2);
            float* q = feat + 24 * NV + 6 * u;
            q[0] = hx; q[1] = hy; q[2] = hz; q[3] = m0; q[4] = m1; q[5] = m2;
          }
        }
#ifdef PNRF_DEBUG_EHEAD
        if (MODE == 1 && valid[cb] && g_dbg_ehead) {
          float* d = g_dbg_ehead + (((a.rays - g_dbg_rays_base) / 11 + row[cb]) * 2 + h) * DBG_STRIDE;
#pragma unroll
          for (int i = 0; i < 8 * KS0 && i < 72; ++i) d[48 + i] = feat[i];
        }
#endif
#pragma unroll
        for (int ks = 0; ks < KS0; ++ks) {
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = feat[8 * ks + j];
          Bo[cb][ks] = P::pack(v);
        }
      }
    });
    // ping-pong: layer 0 Bo -> Bn, then Bn -> Bo, Bo -> Bn, ... (5 hidden layers end in Bo);
    // `pend` = raw accumulators of the previous layer's last tile (its epilogue is deferred into the next layer)
    f32x16 pend[NCB];
#ifndef PNRF_REFINE_PIECES
#define PNRF_REFINE_PIECES 16
#endif
    constexpr int RP = PNRF_REFINE_PIECES;      // pieces of the deferred hidden-layer epilogue (2, 8 or 16)
#ifdef PNRF_DEBUG_EHEAD
    float* dbgp = (MODE == 1 && valid[0] && g_dbg_ehead) ? g_dbg_ehead + (((a.rays - g_dbg_rays_base) / 11 + row[0]) * 2 + h) * DBG_STRIDE : nullptr;
    auto dbg_cks = [&](int idx, int x) { if (dbgp) dbgp[24 + idx] = __int_as_float(x); };
#endif
    auto hidden = [&](v8(&in)[NCB][KS_HID], v8(&out)[NCB][KS_HID], int l) {
      f32x16 np[NCB];
      layer_bf16<NCB, KS_HID, NT_HID, RL::POS_H, RP, F16>(st, ringlane, biaslane + (1 + l) * W_HID, [&](int cb, int ks) { return in[cb][ks]; },
                                                   HiddenEpi<NCB, ACT_ELU, RP, P>{out}, [&](int pc) { HiddenEpi<NCB, ACT_ELU, RP, P>{in}(NT_HID - 1, pc, pend); }, np);
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) pend[cb] = np[cb];
#ifdef PNRF_DEBUG_EHEAD
      dbg_cks(1 + l, dbg_xor<KS_HID>(in[0]));        // `in` = the previous layer's output, complete since this call's first tile
#endif
    };
    layer_bf16<NCB, KS0, NT_HID, 0, RP, F16>(st, ringlane, biaslane, [&](int cb, int ks) { return Bo[cb][ks < KS0 ? ks : 0]; }, HiddenEpi<NCB, ACT_ELU, RP, P>{Bn}, [](int) {}, pend);
#ifdef PNRF_DEBUG_EHEAD
    dbg_cks(0, dbg_xor<(KS0 < KS_HID ? KS0 : KS_HID)>(Bo[0]));                 // the packed inputs
#endif
    const int nhid = PNRF_NHID(a.nhid, R_NHID);              // pairs of layers, then the odd one; the output layer reads Bo (an even count ends in Bn: moved over)
    for (int l = 0; l + 1 < nhid; l += 2) {
      hidden(Bn, Bo, l);
      hidden(Bo, Bn, l + 1);
    }
    if (nhid & 1) hidden(Bn, Bo, nhid - 1);
    else {
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int k = 0; k < KS_HID; ++k) Bo[cb][k] = Bn[cb][k];
    }
    const float* blast = biaslane + (1 + nhid) * W_HID;
    auto pre_last = [&](int pc) { HiddenEpi<NCB, ACT_ELU, 2, P>{Bo}(NT_HID - 1, pc, pend); };
    f32x16 fin[NCB];
    if (!FUSED) {
      auto store_tile = [&](int to, int pc, f32x16(&acc)[NCB]) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
          if (valid[cb]) {
#pragma unroll
            for (int g = 8 * pc; g < 8 * pc + 8; ++g) {
              const int o = a.outmap[(to * 2 + h) * 16 + g];
              // head_act: refine = sigmoid(y[0:8]), offsets = tanh(y[8:32]), rgb = sigmoid(y[32:35]) (helpers:1536-1538)
              const float v = acc[cb][g];
              if (o >= 0) a.y[row[cb] * R_OUT + o] = !a.head_act ? v : ((o < 8 || o >= 32) ? sigmoid_f(v) : tanhf(v));
            }
          }
      };
      layer_bf16<NCB, KS_HID, R_NT_LAST, RL::POS_LAST, BF16_PIECES, F16>(st, ringlane, blast, [&](int cb, int ks) { return Bo[cb][ks]; }, store_tile, pre_last, fin);
      store_tile(R_NT_LAST - 1, 0, fin);
      store_tile(R_NT_LAST - 1, 1, fin);
#pragma unroll
      for (int i = 0; i < RL::SLOTS_PAD; ++i) st.begin();
      continue;
    }
    if constexpr (MODE == 1) {
      layer_bf16<NCB, KS_HID, 1, RL::POS_LAST, BF16_PIECES, F16>(st, ringlane, blast, [&](int cb, int ks) { return Bo[cb][ks]; }, [&](int, int, f32x16(&)[NCB]) {}, pre_last, fin);
#ifdef PNRF_DEBUG_EHEAD
      dbg_cks(6, dbg_xor<KS_HID>(Bo[0]));              // the last hidden layer's output
#endif
#pragma unroll
      for (int i = 0; i < (R_SLOTS_LAST - 1) + RL::SLOTS_PAD; ++i) st.begin();     // rgb tile (unused at inference) + pad
    } else {                   // training: tile 0 = refine + offsets, tile 1 = rgb head (rgb_map0, refine2.py:637)
      f32x16 t1[NCB];
      layer_bf16<NCB, KS_HID, R_NT_LAST, RL::POS_LAST, BF16_PIECES, F16>(
          st, ringlane, blast, [&](int cb, int ks) { return Bo[cb][ks]; },
          [&](int, int pc, f32x16(&acc)[NCB]) {
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
              for (int j = 0; j < 8; ++j) fin[cb][8 * pc + j] = acc[cb][8 * pc + j];
          },
          pre_last, t1);
#pragma unroll
      for (int i = 0; i < RL::SLOTS_PAD; ++i) st.begin();
      if (a.rgb0 && h == 0) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
          if (valid[cb]) {
            a.rgb0[row[cb] * 3 + 0] = sigmoid_f(t1[cb][0]);
            a.rgb0[row[cb] * 3 + 1] = sigmoid_f(t1[cb][1]);
            a.rgb0[row[cb] * 3 + 2] = sigmoid_f(t1[cb][2]);
          }
      }
    }

    // ---- fused epilogue: lane (ray, h) owns samples 4h..4h+3: reg 4a = refine logit, 4a+1..3 = offset
    static_for<NCB>([&](auto cbc) {
      constexpr int cb = decltype(cbc)::value;
      const int64_t rr = valid[cb] ? row[cb] : a.n - 1;
      const float* r = e_ray[cb];
      const float ox = r[0], oy = r[1], oz = r[2], dx = r[3], dy = r[4], dz = r[5], near = r[6], far = r[7];
      const float4 d0 = e_d0[cb], d1 = e_d1[cb];
      // e = [near, d0..d7, far]; window w[i] = e[4h+i], i=0..5
      float w[6];
      w[0] = h ? d0.w : near; w[1] = h ? d1.x : d0.x; w[2] = h ? d1.y : d0.y;
      w[3] = h ? d1.z : d0.z; w[4] = h ? d1.w : d0.w; w[5] = h ? far : d1.x;
      float zz[4], pp[12];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const float lower = ieee_mul(0.5f, ieee_add(w[s4 + 1], w[s4]));       // trt.py:673-675
        const float upper = ieee_mul(0.5f, ieee_add(w[s4 + 2], w[s4 + 1]));
        const float rf = sigmoid_fast(fin[cb][4 * s4]);                           // bf16-grade logits: hardware exp / rcp are exact enough
        zz[s4] = ieee_add(lower, ieee_mul(ieee_sub(upper, lower), rf));     // :676
      }
      if (MODE == 2) {           // depth jitter toward the next / previous refined sample (refine2.py:646-662)
        const float o0 = __shfl_xor(zz[0], 32), o3 = __shfl_xor(zz[3], 32);     // the other half's first / last sample
        if (a.jitter) {
          const float4 jt = *(const float4*)(a.jitter + rr * 8 + 4 * h);
          const float jv[4] = {jt.x, jt.y, jt.z, jt.w};
          float zn[4];
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) {
            if (a.jitter_dir > 0) {
              const float nxt = s4 < 3 ? zz[s4 < 3 ? s4 + 1 : 3] : (h == 0 ? o0 : far);
              zn[s4] = ieee_add(zz[s4], ieee_mul(jv[s4], fabsf(ieee_sub(zz[s4], nxt))));
            } else {
              const float prv = s4 > 0 ? zz[s4 > 0 ? s4 - 1 : 0] : (h == 1 ? o3 : near);
              zn[s4] = ieee_add(zz[s4], ieee_mul(-jv[s4], fabsf(ieee_sub(zz[s4], prv))));
            }
          }
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) zz[s4] = zn[s4];
        }
      }
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const float zv = zz[s4];
        const float fx = tanh_fast(fin[cb][4 * s4 + 1]), fy = tanh_fast(fin[cb][4 * s4 + 2]), fz = tanh_fast(fin[cb][4 * s4 + 3]);
        pp[3 * s4 + 0] = ieee_add(ieee_add(ox, ieee_mul(dx, zv)), ieee_mul(1e-2f, fx));   // :679-681
        pp[3 * s4 + 1] = ieee_add(ieee_add(oy, ieee_mul(dy, zv)), ieee_mul(1e-2f, fy));
        pp[3 * s4 + 2] = ieee_add(ieee_add(oz, ieee_mul(dz, zv)), ieee_mul(1e-2f, fz));
      }
#ifdef PNRF_DEBUG_EHEAD
      if (MODE == 1 && valid[cb] && g_dbg_ehead) {
        const int64_t g = (a.rays - g_dbg_rays_base) / 11 + row[cb];
        float* d = g_dbg_ehead + (g * 2 + h) * DBG_STRIDE;
        d[0] = d0.x; d[1] = d0.y; d[2] = d0.z; d[3] = d0.w; d[4] = d1.x; d[5] = d1.y; d[6] = d1.z; d[7] = d1.w;
#pragma unroll
        for (int i = 0; i < 8; ++i) d[8 + i] = r[i];
        d[16] = __int_as_float((int)__builtin_amdgcn_s_getreg((31 << 11) | 4));      // HW_ID
        d[17] = __int_as_float((int)__builtin_amdgcn_s_getreg((31 << 11) | 20));     // XCC_ID
        d[18] = __int_as_float((int)blockIdx.x * 65536 + (int)threadIdx.x);
        d[19] = __int_as_float((int)(__builtin_readcyclecounter() >> 8));
        d[20] = __int_as_float(batch);
        d[21] = __int_as_float((int)__builtin_amdgcn_s_getreg((31 << 11) | 6));      // LDS_ALLOC (physical base / size of the workgroup's LDS)
        d[22] = __int_as_float((int)__builtin_amdgcn_s_getreg((31 << 11) | 5));      // GPR_ALLOC (physical VGPR / SGPR base and size of the wave)
#pragma unroll
        for (int i = 0; i < 16; ++i) d[32 + i] = fin[cb][i];
      }
#endif
      if (valid[cb]) {
        *(float4*)(a.z + row[cb] * 8 + 4 * h) = make_float4(zz[0], zz[1], zz[2], zz[3]);
        float4* pq = (float4*)(a.pts + row[cb] * 24 + 12 * h);
        pq[0] = make_float4(pp[0], pp[1], pp[2], pp[3]);
        pq[1] = make_float4(pp[4], pp[5], pp[6], pp[7]);
        pq[2] = make_float4(pp[8], pp[9], pp[10], pp[11]);
      }
    });
  }
  st.drain();
}

// ------------------------------------------------------------------------------------------ refine stage on the 16x16x32 engine (round 6)
// refine16_kernel: the fused inference refine stage (projection head or refine_in rows -> ELU MLP -> interval refinement + query points) on layer_e16:
// v_mfma_f32_16x16x32_f16, two blocks of 16 rays per wave, 8 (4) waves = 256 (128) rays per batch like refine_kernel<1, 8 (4)>.  Lane l: ray l & 15 of each
// block, group g = l >> 4.  The four lanes of a ray share its head and its epilogue: group g projects the ray's 8 samples into views NV4 g .. NV4 g + NV4 - 1
// and encodes samples 2g, 2g + 1 (refine16_in0), and the output pair lands as tile t, registers 0..3 of group g = [logit, offset xyz] of sample 2g + t
// (refine16_out): no exchange between lanes anywhere.  Same arithmetic per ray as refine_kernel<.., PrecF16> (the same fp16 operands, fp32 accumulation in
// another order).
template <class P>
struct HiddenEpiE16 {
  typename P::v8 (&Bn)[2][NB_KS_H];
  __device__ __forceinline__ void operator()(int tp, int p, f32x4 (&acc)[2][2]) const {
    const int t = p >> 3, cb = (p >> 2) & 1, r = p & 3;
    acc[t][cb][r] = elu_scaled(acc[t][cb][r]);          // the even one waits, activated, in its accumulator register
    if (r & 1) {
      i32x4_t w = __builtin_bit_cast(i32x4_t, Bn[cb][tp]);
      w[2 * t + (r >> 1)] = P::cvt_pk(acc[t][cb][r - 1], acc[t][cb][r]);
      Bn[cb][tp] = __builtin_bit_cast(typename P::v8, w);
    }
  }
};

template <int NW, int HEAD, int NV4>
__global__ __launch_bounds__(64 * NW, 2) void refine16_kernel(RefineArgs a) {
  using P = PrecF16;
  using v8 = typename P::v8;
  constexpr bool FOLD = HEAD == 1;          // the head that computes the Pluecker vector itself runs the folded first layer (pnrf_layout.h)
  using RL = RefineE16<NV4, FOLD>;
  using Epi = HiddenEpiE16<P>;
  constexpr int KS0 = RL::KS0, TPB = 64 * NW;
  static_assert(KS0 <= NB_KS_H, "layer 0 fits the hidden layers' operand registers");
  P::enter();
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* bias_lds = (float*)(smem + RING_BYTES);
  WStream<NW> st;
  st.init(a.blob, a.nslots, smem);
  st.prologue();
  for (int i = threadIdx.x; i < a.nbias; i += TPB) bias_lds[i] = a.bias[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c16 = lane & 15, g = lane >> 4;
  own_the_simd<NW>();
  young_half_priority<NW>();
  const char* ringlane = smem + lane * 16;
  const float* biaslane = bias_lds + 4 * g;

  // batch order: as refine_kernel (HEAD = 1: the workgroups of an XCD walk one band of the frame)
  int jbase = 0, j0 = (int)blockIdx.x, jstep = (int)gridDim.x, jend = a.nbatch;
  if (HEAD == 1 && (gridDim.x & 7) == 0 && a.nbatch >= (int)gridDim.x) {
    const int per = (a.nbatch + 7) >> 3;
    jbase = (int)(blockIdx.x & 7) * per; j0 = (int)(blockIdx.x >> 3); jstep = (int)(gridDim.x >> 3);
    jend = a.nbatch - jbase < per ? a.nbatch - jbase : per;
  }
  for (int j = j0; j < jend; j += jstep) {
    const int batch = jbase + j;
    int64_t row[2];
    bool valid[2];
    v8 Bo[2][NB_KS_H], Bn[2][NB_KS_H];
    float e_ray[2][6];           // origin, direction
    float e_w[2][4];             // the group's window of e = [near, d0 .. d7, far]: e[2g .. 2g + 3]
    static_for<2>([&](auto cbc) {
      constexpr int cb = decltype(cbc)::value;
      row[cb] = (int64_t)batch * (NW * 32) + wave * 32 + cb * 16 + c16;
      valid[cb] = row[cb] < a.n;
      const int64_t rr = valid[cb] ? row[cb] : a.n - 1;
      const float* r = a.rays + rr * 11;
#pragma unroll
      for (int i = 0; i < 6; ++i) e_ray[cb][i] = r[i];
      const float* ds = a.depth_sorted + rr * 8;
      const float wl = ds[g > 0 ? 2 * g - 1 : 0], wh = ds[g < 3 ? 2 * g + 2 : 7];
      e_w[cb][0] = g == 0 ? r[6] : wl; e_w[cb][1] = ds[2 * g]; e_w[cb][2] = ds[2 * g + 1]; e_w[cb][3] = g == 3 ? r[7] : wh;
      float feat[8 * KS0];
      if constexpr (HEAD == 0) {
        // a lane's inputs are runs of the natural row: the 24 colours of each of its views at 48 + 24 view, the Pluecker values of its two samples at 12 g
        const float* xr = a.x + rr * (48 + 24 * a.nb);
#pragma unroll
        for (int vv = 0; vv < NV4; ++vv) {
          const int view = NV4 * g + vv;
          const float4* xc = (const float4*)(xr + 48 + 24 * (view < a.nb ? view : 0));
#pragma unroll
          for (int q = 0; q < 6; ++q) {
            float4 t = xc[q];
            if (view >= a.nb) t = make_float4(0.f, 0.f, 0.f, 0.f);
            feat[24 * vv + 4 * q + 0] = t.x; feat[24 * vv + 4 * q + 1] = t.y; feat[24 * vv + 4 * q + 2] = t.z; feat[24 * vv + 4 * q + 3] = t.w;
          }
        }
        const float4* xp = (const float4*)(xr + 12 * g);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const float4 t = xp[q];
          feat[24 * NV4 + 4 * q + 0] = t.x; feat[24 * NV4 + 4 * q + 1] = t.y; feat[24 * NV4 + 4 * q + 2] = t.z; feat[24 * NV4 + 4 * q + 3] = t.w;
        }
      } else {
        // lane (ray, g): views NV4 g + vv x 8 samples (24 NV4 colours) + Pluecker of samples 2g, 2g + 1, in the order of refine16_in0; the projections in
        // refine_kernel's software pipeline (D projections' texel fetches in flight)
        const float* orr = a.or_rays + rr * 11;
        const float o0 = orr[0], o1 = orr[1], o2 = orr[2], w0 = orr[3], w1 = orr[4], w2 = orr[5];
        const float4 d0 = *(const float4*)ds, d1 = *(const float4*)(ds + 4);
        const float dn8[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
        const int plane = a.Hf * a.Wf;
        constexpr int D = 4;
        float wt[D][4];
        float4 tx[D][4];
        ViewRay vr[NV4];
        int view[NV4];
        float z3d[8];
#pragma unroll
        for (int vv = 0; vv < NV4; ++vv) {
          view[vv] = NV4 * g + vv < a.nb ? NV4 * g + vv : a.nb - 1;        // a view beyond nb - 1 is padding (zero weights): it projects into the last real view
          float M[12];
#pragma unroll
          for (int i = 0; i < 12; ++i) M[i] = a.proj[view[vv] * 12 + i];
          vr[vv] = view_ray(o0, o1, o2, w0, w1, w2, M);
        }
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) z3d[s8] = __builtin_amdgcn_rcpf(1.f - dn8[s8] - a.eps);                     // trt.py:637
        const char* imb = (const char*)a.img4;
        auto texel = [&](uint32_t view_off, uint32_t idx) { return *(const float4*)(imb + ((view_off + idx) << 4)); };
        auto issue = [&](auto pc) {
          constexpr int p = decltype(pc)::value, vv = p / 8, sl = p % D;
          const uint32_t vo = (uint32_t)(view[vv] * plane);
          const Taps t = project_taps_fast(vr[vv], z3d[p % 8], a.Hf, a.Wf);
          tx[sl][0] = texel(vo, t.i00); tx[sl][1] = texel(vo, t.i01); tx[sl][2] = texel(vo, t.i10); tx[sl][3] = texel(vo, t.i11);
          wt[sl][0] = t.a00; wt[sl][1] = t.a01; wt[sl][2] = t.a10; wt[sl][3] = t.a11;
        };
        auto blend = [&](auto pc) {
          constexpr int p = decltype(pc)::value, sl = p % D;
          Taps t;
          t.a00 = wt[sl][0]; t.a01 = wt[sl][1]; t.a10 = wt[sl][2]; t.a11 = wt[sl][3];
          feat[3 * p + 0] = blend4(tx[sl][0].x, tx[sl][1].x, tx[sl][2].x, tx[sl][3].x, t);
          feat[3 * p + 1] = blend4(tx[sl][0].y, tx[sl][1].y, tx[sl][2].y, tx[sl][3].y, t);
          feat[3 * p + 2] = blend4(tx[sl][0].z, tx[sl][1].z, tx[sl][2].z, tx[sl][3].z, t);
        };
        static_for<D>([&](auto pc) { issue(pc); });
        __builtin_amdgcn_sched_barrier(0);
        static_for<8 * NV4>([&](auto pc) {
          constexpr int p = decltype(pc)::value;
          blend(pc);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (p + D < 8 * NV4) {
            issue(std::integral_constant<int, p + D>{});
            __builtin_amdgcn_sched_barrier(0);
          }
        });
        {                                                                                          // trt.py:656-658, folded: [d^, o x d^] in group 0
          float hx, hy, hz, m0, m1, m2;
          unit_dir(r[3], r[4], r[5], hx, hy, hz);
          cross_rn(r[0], r[1], r[2], hx, hy, hz, m0, m1, m2);
          float* q = feat + 24 * NV4;
          q[0] = g == 0 ? hx : 0.f; q[1] = g == 0 ? hy : 0.f; q[2] = g == 0 ? hz : 0.f;
          q[3] = g == 0 ? m0 : 0.f; q[4] = g == 0 ? m1 : 0.f; q[5] = g == 0 ? m2 : 0.f;
        }
      }
#pragma unroll
      for (int i = 24 * NV4 + (FOLD ? 6 : 12); i < 8 * KS0; ++i) feat[i] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS0; ++ks) {
        float v[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) v[jj] = feat[8 * ks + jj];
        Bo[cb][ks] = P::pack(v);
      }
    });
    // ping-pong as refine_kernel: layer 0 Bo -> Bn, hidden layers in pairs, the output layer reads Bo
    f32x4 pend[2][2];
    auto hidden = [&](v8(&in)[2][NB_KS_H], v8(&out)[2][NB_KS_H], int l) {
      f32x4 np[2][2];
      layer_e16<NB_KS_H, NB_NTP_H, RL::POS_H, v8>(st, ringlane, biaslane + (1 + l) * W_HID, [&](int cb, int ks) { return in[cb][ks]; }, Epi{out},
                                                  [&](int p) { Epi{in}(NB_NTP_H - 1, p, pend); }, np);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) pend[t][cb] = np[t][cb];
    };
    layer_e16<KS0, NB_NTP_H, 0, v8>(st, ringlane, biaslane, [&](int cb, int ks) { return Bo[cb][ks < KS0 ? ks : 0]; }, Epi{Bn}, [](int) {}, pend);
    const int nhid = PNRF_NHID(a.nhid, R_NHID);
    for (int l = 0; l + 1 < nhid; l += 2) {
      hidden(Bn, Bo, l);
      hidden(Bo, Bn, l + 1);
    }
    if (nhid & 1) hidden(Bn, Bo, nhid - 1);
    else {
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int k = 0; k < NB_KS_H; ++k) Bo[cb][k] = Bn[cb][k];
    }
    f32x4 fin[2][2];
    layer_e16<NB_KS_H, 1, RL::POS_LAST, v8>(st, ringlane, biaslane + (1 + nhid) * W_HID, [&](int cb, int ks) { return Bo[cb][ks]; }, [&](int, int, f32x4(&)[2][2]) {},
                                            [&](int p) { Epi{Bo}(NB_NTP_H - 1, p, pend); }, fin);
#pragma unroll
    for (int i = 0; i < (R16_SLOTS_LAST - 1) + RL::SLOTS_PAD; ++i) st.begin();     // rgb pair (unused at inference) + pad

    // ---- fused epilogue: lane (ray, g) owns samples 2g, 2g + 1: tile t, reg 0 = refine logit, regs 1..3 = offset
    static_for<2>([&](auto cbc) {
      constexpr int cb = decltype(cbc)::value;
      const float* r = e_ray[cb];
      const float ox = r[0], oy = r[1], oz = r[2], dx = r[3], dy = r[4], dz = r[5];
      const float* w = e_w[cb];
      float zz[2], pp[6];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float lower = ieee_mul(0.5f, ieee_add(w[t + 1], w[t]));          // trt.py:673-675
        const float upper = ieee_mul(0.5f, ieee_add(w[t + 2], w[t + 1]));
        const float rf = sigmoid_fast(fin[t][cb][0]);
        zz[t] = ieee_add(lower, ieee_mul(ieee_sub(upper, lower), rf));        // :676
        const float fx = tanh_fast(fin[t][cb][1]), fy = tanh_fast(fin[t][cb][2]), fz = tanh_fast(fin[t][cb][3]);
        pp[3 * t + 0] = ieee_add(ieee_add(ox, ieee_mul(dx, zz[t])), ieee_mul(1e-2f, fx));   // :679-681
        pp[3 * t + 1] = ieee_add(ieee_add(oy, ieee_mul(dy, zz[t])), ieee_mul(1e-2f, fy));
        pp[3 * t + 2] = ieee_add(ieee_add(oz, ieee_mul(dz, zz[t])), ieee_mul(1e-2f, fz));
      }
      if (valid[cb]) {
        *(float2*)(a.z + row[cb] * 8 + 2 * g) = make_float2(zz[0], zz[1]);
        float2* pq = (float2*)(a.pts + row[cb] * 24 + 6 * g);
        pq[0] = make_float2(pp[0], pp[1]);
        pq[1] = make_float2(pp[2], pp[3]);
        pq[2] = make_float2(pp[4], pp[5]);
      }
    });
  }
  st.drain();
}

struct NerfArgs {
  const void* blob; const float* bias; uint32_t nslots; int nbias;
  int nhid;                                         // DoNeRFTRT: hidden 256 -> 256 layers behind layer 0 (netdepth - 2; the Fern configs: 6); unused by the class net
  int64_t n;                                        // fused: rays; module-level: rows
  int nbatch;
  const float* pts; const float* rays;              // fused producer
  const float* x; const float* xv; const int* in0; const int* inx;   // module-level producer
  const float* z; const float* add; const float* mul;                // fused consumer
  float* rgbd; float* raw;
  const float* noise; int white_bkgd;                                // training-time compositing: sigma noise [n,8], white background
  float clampv; int S;                                               // raw clamp (stage 1: 10), samples per ray (8; stage-1 exploration: 8..256)
  float* y; const int* outmap;                      // module-level consumer
  int* queue;                                       // nerf16_kernel: [0] batches handed out beyond the first round, [1] workgroups done; NULL = static stride
};

// sin / cos of scale * x, scale a power of two: hardware v_sin / v_cos on the fraction of the angle in revolutions (valid for any
// magnitude; error of the fraction 2^-24 * |scale x / 2 pi|, far below the bf16 rounding of the MLP input)
__device__ __forceinline__ void pe_sincos_scaled(float x, float scale, float& s, float& c) {
#ifdef PNRF_EXACT_SINCOS
  sincosf(x * scale, &s, &c);
#else
  const float rev = __builtin_amdgcn_fractf(x * (scale * 0.15915494309189535f));
  s = __builtin_amdgcn_sinf(rev); c = __builtin_amdgcn_cosf(rev);
#endif
}
__device__ __forceinline__ void pe_sincos(float x, float& s, float& c) {
#ifdef PNRF_EXACT_SINCOS
  sincosf(x, &s, &c);
#else
  s = __sinf(x); c = __cosf(x);
#endif
}

// CLS = false: DoNeRFTRT; CLS = true: the NeRF class (skip-concat at layer 5, feature/alpha heads, view branch)
template <int NCB, int NW, bool FUSED, bool CLS>
__global__ __launch_bounds__(64 * NW, NCB == 1 ? 2 : 1) void nerf_kernel(NerfArgs a) {
  constexpr int TPB = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* bias_lds = (float*)(smem + RING_BYTES);
  for (int i = threadIdx.x; i < a.nbias; i += TPB) bias_lds[i] = a.bias[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 31, h = lane >> 5;
  constexpr int COLS = 32 * NCB;
  const int64_t nrows = FUSED ? a.n * a.S : a.n;
  WStream<NW> st;
  st.init(a.blob, a.nslots, smem);
  st.prologue();
  own_the_simd<NW>();
  young_half_priority<NW>();
  const char* ringlane = smem + lane * 16;
  const float* biaslane = bias_lds + h * 16;

  // Raw per-column inputs of a batch (fused path): sample position, view direction, and what the compositing epilogue
  // needs — all fetched at the head of the batch: loaded where it is used, the epilogue's share costs a second exposed HBM
  // round trip with all 8 waves of the workgroup waiting behind the last layer (3.74 -> 3.66 ms per frame).  Fetching
  // one batch ahead from inside the last hidden layer was measured too and is a loss (3.73 ms): 13 more live registers
  // spill, and the loads sit in the same in-order vmcnt queue as the weight stream.
  struct RawIn { float x[3], v[3], d[3], z, add, mul, noise; };
  const bool composite = FUSED && a.S == 8 && a.rgbd;
  auto fetch = [&](int b, RawIn(&R)[NCB]) {
    static_for<NCB>([&](auto cbc) {
      constexpr int cb = decltype(cbc)::value;
      const int64_t r0 = (int64_t)b * (NW * COLS) + wave * COLS + cb * 32 + col;
      const int64_t rr = r0 < nrows ? r0 : nrows - 1;
      const float* pp = a.pts + rr * 3;
      const float* ry = a.rays + (a.S == 8 ? (rr >> 3) : rr / a.S) * 11;
#pragma unroll
      for (int c = 0; c < 3; ++c) { R[cb].x[c] = pp[c]; R[cb].v[c] = ry[8 + c]; }
      if (composite) {
#pragma unroll
        for (int c = 0; c < 3; ++c) R[cb].d[c] = ry[3 + c];
        R[cb].z = a.z[rr]; R[cb].add = a.add ? a.add[rr] : 0.f; R[cb].mul = a.mul ? a.mul[rr] : 1.f;
        R[cb].noise = a.noise ? a.noise[rr] : 0.f;
      }
    });
  };
  RawIn raw[NCB];

  for (int batch = blockIdx.x; batch < a.nbatch; batch += gridDim.x) {
    int64_t row[NCB];
    bool valid[NCB];
    bf16x8 Bo[NCB][KS_HID], Bn[NCB][KS_HID], Bx[NCB][N_KSX];
    float e_dn[NCB], e_z[NCB], e_add[NCB], e_mul[NCB], e_noise[NCB];
    if constexpr (FUSED) fetch(batch, raw);
    static_for<NCB>([&](auto cbc) {
      constexpr int cb = decltype(cbc)::value;
      row[cb] = (int64_t)batch * (NW * COLS) + wave * COLS + cb * 32 + col;
      valid[cb] = row[cb] < nrows;
      const int64_t rr = valid[cb] ? row[cb] : nrows - 1;
      if (FUSED && composite) {
        const float* r = raw[cb].d;
        e_dn[cb] = ieee_sqrt(ieee_add(ieee_add(ieee_mul(r[0], r[0]), ieee_mul(r[1], r[1])), ieee_mul(r[2], r[2])));
        e_z[cb] = raw[cb].z; e_add[cb] = raw[cb].add; e_mul[cb] = raw[cb].mul; e_noise[cb] = raw[cb].noise;
      }
      if (FUSED) {
        // positional encoding in B-fragment order (nerf_in0 / nerf_inx): half 0 = sin, half 1 = cos.
        // sin/cos(2^k x): sin/cos at k=0, then the exact double-angle recurrence, whose error doubles per octave: with an
        // accurate sincosf (<= 2^k * 1e-7) or, by default, the hardware v_sin/v_cos (abs error ~1e-6 for |x| of a few units ->
        // <= 5e-4 at k=9) it stays below the bf16 rounding (2^-9 relative) applied to the MLP input; -DPNRF_EXACT_SINCOS
        // selects the accurate one (6 calls of ~40 VALU each were 2/3 of the batch prologue).
        const float x3[3] = {raw[cb].x[0], raw[cb].x[1], raw[cb].x[2]};
        const float v3[3] = {raw[cb].v[0], raw[cb].v[1], raw[cb].v[2]};
        float f0[32], fx[16];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float s, co;
          pe_sincos(x3[c], s, co);                                  // helpers:669-670: sin/cos(x * 2^k)
#pragma unroll
          for (int k = 0; k < 10; ++k) {
            f0[3 * k + c] = h ? co : s;
            const float s2 = 2.f * s * co, c2 = (co - s) * (co + s);
            s = s2; co = c2;
          }
          pe_sincos(v3[c], s, co);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            fx[3 * k + c] = h ? co : s;
            const float s2 = 2.f * s * co, c2 = (co - s) * (co + s);
            s = s2; co = c2;
          }
        }
        f0[30] = h ? x3[2] : x3[0];
        f0[31] = h ? 0.f : x3[1];
        fx[12] = h ? v3[2] : v3[0];
        fx[13] = h ? 0.f : v3[1];
        fx[14] = 0.f; fx[15] = 0.f;
#pragma unroll
        for (int ks = 0; ks < N_KS0; ++ks) {
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = f0[ks * 8 + j];
          Bo[cb][ks] = pack_bf16(v);
        }
#pragma unroll
        for (int e = 0; e < N_KSX; ++e) {
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = fx[e * 8 + j];
          Bx[cb][e] = pack_bf16(v);
        }
      } else {
        const float* xr = a.x + rr * N_IN;
        const float* xv = a.xv + rr * N_INV;
#pragma unroll
        for (int ks = 0; ks < N_KS0; ++ks) {
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int i = a.in0[(ks * 2 + h) * 8 + j];
            v[j] = i >= 0 ? xr[i] : 0.f;
          }
          Bo[cb][ks] = pack_bf16(v);
        }
#pragma unroll
        for (int e = 0; e < N_KSX; ++e) {
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int i = a.inx[(e * 2 + h) * 8 + j];
            v[j] = i >= 0 ? xv[i] : 0.f;
          }
          Bx[cb][e] = pack_bf16(v);
        }
      }
    });
    f32x16 fin[NCB];
    f32x16 pend[NCB];      // raw accumulators of the previous layer's last tile (its epilogue is deferred into the next layer)
    if constexpr (!CLS) {
      // ping-pong: layer 0 Bo -> Bn, then Bn -> Bo, Bo -> Bn, ... (6 hidden layers end in Bn)
      auto hidden = [&](bf16x8(&in)[NCB][KS_HID], bf16x8(&out)[NCB][KS_HID], int l) {
        f32x16 np[NCB];
        layer_bf16<NCB, KS_HID, NT_HID, N_POS_H>(st, ringlane, biaslane + (1 + l) * W_HID, [&](int cb, int ks) { return in[cb][ks]; },
                                                 HiddenEpi<NCB, ACT_RELU>{out}, [&](int pc) { HiddenEpi<NCB, ACT_RELU>{in}(NT_HID - 1, pc, pend); }, np);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) pend[cb] = np[cb];
      };
      layer_bf16<NCB, N_KS0, NT_HID, 0>(st, ringlane, biaslane, [&](int cb, int ks) { return Bo[cb][ks]; }, HiddenEpi<NCB, ACT_RELU>{Bn}, [](int) {}, pend);
      // This 32x32x16 form (PNRF_VARIANT_BF16_32X32, the module-level forward) is built for the Fern depth only — its launchers check: a run-time
      // layer count made it spill (48 .. 144 bytes, three formulations); the default 16x16x32 engine (nerf16_kernel) takes any netdepth.
      for (int l = 0; l < N_NHID; l += 2) {
        hidden(Bn, Bo, l);
        hidden(Bo, Bn, l + 1);
      }
      layer_bf16<NCB, N_KS_LAST, 1, N_POS_LAST>(
          st, ringlane, biaslane + (1 + N_NHID) * W_HID,
          [&](int cb, int ks) { return ks < KS_HID ? Bn[cb][ks < KS_HID ? ks : 0] : Bx[cb][ks >= KS_HID ? ks - KS_HID : 0]; },
          [&](int, int, f32x16(&)[NCB]) {}, [&](int pc) { HiddenEpi<NCB, ACT_RELU>{Bn}(NT_HID - 1, pc, pend); }, fin);
#pragma unroll
      for (int i = 0; i < N_SLOTS_PAD; ++i) st.begin();
    } else {
      // NeRF class (helpers:824-847).  E0 Bo->Bn | E1..E4 ping-pong (ends in Bn) | E5 [Bn, P]->Bo | E6 Bo->Bn | E7 Bn->Bo |
      // E89 [Bo, Bx]->Bn (views layer with feature_linear folded in, 128 wide) + alpha | E10 Bn -> rgb
      bf16x8 P[NCB][N_KS0];                 // positional B fragments, needed again by the skip connection at layer 5
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int k = 0; k < N_KS0; ++k) P[cb][k] = Bo[cb][k];
      auto hidden = [&](bf16x8(&in)[NCB][KS_HID], bf16x8(&out)[NCB][KS_HID], int l, auto posc) {
        constexpr int POS = decltype(posc)::value;
        f32x16 np[NCB];
        layer_bf16<NCB, KS_HID, NT_HID, POS>(st, ringlane, biaslane + l * W_HID, [&](int cb, int ks) { return in[cb][ks]; },
                                             HiddenEpi<NCB, ACT_RELU>{out}, [&](int pc) { HiddenEpi<NCB, ACT_RELU>{in}(NT_HID - 1, pc, pend); }, np);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) pend[cb] = np[cb];
      };
      layer_bf16<NCB, N_KS0, NT_HID, 0>(st, ringlane, biaslane, [&](int cb, int ks) { return Bo[cb][ks]; }, HiddenEpi<NCB, ACT_RELU>{Bn}, [](int) {}, pend);
      for (int l = 1; l < 5; l += 2) {        // E1..E4
        hidden(Bn, Bo, l, std::integral_constant<int, C_POS_E1>{});
        hidden(Bo, Bn, l + 1, std::integral_constant<int, C_POS_E1>{});
      }
      {                                       // E5: cat[pts, h] -> 256 (skip connection after layer 4)
        f32x16 np[NCB];
        layer_bf16<NCB, C_KS5, NT_HID, C_POS_E5>(
            st, ringlane, biaslane + 5 * W_HID,
            [&](int cb, int ks) { return ks < KS_HID ? Bn[cb][ks < KS_HID ? ks : 0] : P[cb][ks >= KS_HID ? ks - KS_HID : 0]; },
            HiddenEpi<NCB, ACT_RELU>{Bo}, [&](int pc) { HiddenEpi<NCB, ACT_RELU>{Bn}(NT_HID - 1, pc, pend); }, np);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) pend[cb] = np[cb];
      }
      hidden(Bo, Bn, 6, std::integral_constant<int, C_POS_E6>{});
      hidden(Bn, Bo, 7, std::integral_constant<int, C_POS_E6>{});
      float alpha[NCB];
      {                                       // E89: views layer (feature_linear folded in) on cat[h, view encoding] -> 128 ReLU (tiles 0-3 -> Bn)
        f32x16 np[NCB];                       //      + alpha (tile 4, row 0, no activation)
        layer_bf16<NCB, C_KS9, C_NT89, C_POS_E89>(
            st, ringlane, biaslane + C_BIAS_E89,
            [&](int cb, int ks) { return ks < KS_HID ? Bo[cb][ks < KS_HID ? ks : 0] : Bx[cb][ks >= KS_HID ? ks - KS_HID : 0]; },
            HiddenEpi<NCB, ACT_RELU>{Bn}, [&](int pc) { HiddenEpi<NCB, ACT_RELU>{Bo}(NT_HID - 1, pc, pend); }, np);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) alpha[cb] = np[cb][0];
      }
      layer_bf16<NCB, C_KS10, 1, C_POS_E10>(st, ringlane, biaslane + C_BIAS_E10, [&](int cb, int ks) { return Bn[cb][ks]; },
                                            [&](int, int, f32x16(&)[NCB]) {}, [](int) {}, fin);
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) fin[cb][3] = alpha[cb];          // raw = [rgb, alpha] (helpers:851)
#pragma unroll
      for (int i = 0; i < C_SLOTS_PAD; ++i) st.begin();
    }

    if (!FUSED) {
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
        if (valid[cb] && h == 0) *(float4*)(a.y + row[cb] * 4) = make_float4(fin[cb][0], fin[cb][1], fin[cb][2], fin[cb][3]);
      continue;
    }
    // ---- fused epilogue: raw rgb-sigma of column `col` sits in regs 0-3 of half 0.  S != 8 (stage-1 exploration): only the
    // raw output is written and pnrf_composite_fwd does the compositing.  S == 8: the 8 samples of a ray are 8 adjacent
    // lanes; compositing in the reference's sequential order (cumprod, sum).
    static_for<NCB>([&](auto cbc) {
      constexpr int cb = decltype(cbc)::value;
      const int64_t rr = valid[cb] ? row[cb] : nrows - 1;
      float r0 = fin[cb][0], r1 = fin[cb][1], r2 = fin[cb][2], r3 = fin[cb][3];
      if (a.raw && valid[cb] && h == 0) *(float4*)(a.raw + row[cb] * 4) = make_float4(r0, r1, r2, r3);
      if (!composite) return;
      const int64_t ray = rr >> 3;
      const int s = (int)(rr & 7);
      const float dn = e_dn[cb], zc = e_z[cb], ad = e_add[cb], mu = e_mul[cb];
      if (a.clampv > 0.f) {                                                         // base.py:523
        r0 = fminf(fmaxf(r0, -a.clampv), a.clampv); r1 = fminf(fmaxf(r1, -a.clampv), a.clampv);
        r2 = fminf(fmaxf(r2, -a.clampv), a.clampv); r3 = fminf(fmaxf(r3, -a.clampv), a.clampv);
      }
      const float znext = __shfl_down(zc, 1);
      float dist = (s < 7) ? ieee_sub(znext, zc) : 1e10f;                          // trt.py:579-581
      dist = ieee_mul(dist, dn);                                                   // :583
      const float cr = sigmoid_f(r0), cg = sigmoid_f(r1), cbv = sigmoid_f(r2);      // :585
      const float sg = fmaxf(ieee_add(a.noise ? ieee_add(r3, e_noise[cb]) : r3, ad), 0.f);     // refine2.py:508
      float alpha = ieee_sub(1.f, expf(ieee_mul(-sg, dist)));                     // :577,587
      if (a.mul) alpha = ieee_mul(alpha, fmaxf(mu, 0.f));                          // :588
      const float xk = ieee_add(ieee_sub(1.f, alpha), 1e-10f);                    // :590
      const int base = lane & 0x38;
      float T = 1.f;
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        const float xj = __shfl(xk, base + j);
        T = (j < s) ? ieee_mul(T, xj) : T;
      }
      const float wgt = ieee_mul(alpha, T);
      const float c0 = ieee_mul(wgt, cr), c1 = ieee_mul(wgt, cg), c2 = ieee_mul(wgt, cbv), c3 = ieee_mul(wgt, zc);
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, sa = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        s0 = ieee_add(s0, __shfl(c0, base + j));                                   // :591 sum over samples
        s1 = ieee_add(s1, __shfl(c1, base + j));
        s2 = ieee_add(s2, __shfl(c2, base + j));
        s3 = ieee_add(s3, __shfl(c3, base + j));                                   // :593 depth_map
        sa = ieee_add(sa, __shfl(wgt, base + j));                                  // acc_map
      }
      if (a.white_bkgd) {                                                           // refine2.py:519-520
        const float bg = ieee_sub(1.f, sa);
        s0 = ieee_add(s0, bg); s1 = ieee_add(s1, bg); s2 = ieee_add(s2, bg);
      }
      if (valid[cb] && h == 0 && s == 0) *(float4*)(a.rgbd + ray * 4) = make_float4(s0, s1, s2, s3);
    });
  }
  st.drain();
}

// ------------------------------------------------------------------------------------------ DoNeRFTRT on the 16x16x32 engine
// nerf16_kernel: the fused NeRF stage (positional encoding -> 8-layer MLP -> compositing) on layer_b16.  Per wave two blocks of
// 16 columns (ray samples), 8 waves = 256 columns per workgroup batch, like nerf_kernel<1, 8>.  Lane l: column l&15 of each block,
// group g = l>>4.  The four groups of a column share the positional encoding: group g evaluates four chains of two consecutive
// octaves of one component each and octave g of the view direction (nerf16_in0 / nerf16_inx define which stream feature each register slot is).  The network output (rows
// 0..3 of the last tile) lands in group 0: lanes 0..15 hold [r, g, b, sigma] of their column, 8 adjacent lanes = one ray.
// ReLU after the conversion, on the packed pair: as 16-bit integers the bf16 patterns of negative values (and -0) are negative, so one
// v_pk_max_i16 against 0 clears them — the same bits as converting max(x, 0), with one instruction per two values instead of two.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int relu_pack_bf16(float a, float b) {
#ifdef PNRF_RELU_F32
  return __builtin_bit_cast(int, bf16x2_t{(__bf16)act_fast(a, ACT_RELU), (__bf16)act_fast(b, ACT_RELU)});
#else
  int pk;                                    // one conversion for the pair (the vector-of-two form compiles to two conversions and a v_perm)
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(a), "v"(b));
  return __builtin_bit_cast(int, __builtin_elementwise_max(__builtin_bit_cast(i16x2_t, pk), i16x2_t{0, 0}));
#endif
}
template <class P>
__device__ __forceinline__ int relu_pack(float a, float b) {
  if constexpr (std::is_same<P, PrecBf16>::value) return relu_pack_bf16(a, b);
  // the same ReLU on the packed pair (the conversion saturates: MODE.FP16_OVFL)
  return __builtin_bit_cast(int, __builtin_elementwise_max(__builtin_bit_cast(i16x2_t, P::cvt_pk(a, b)), i16x2_t{0, 0}));
}
template <int NCB, class P = PrecBf16>
struct HiddenEpi16 {
  typename P::v8 (&Bn)[NCB][NB_KS_H];
  // column blocks cb0, cb0 + 1 of tile pc
  __device__ __forceinline__ void operator()(int tp, int pc, f32x4 (&acc)[2][NCB], int cb0) const {
#pragma unroll
    for (int cb = cb0; cb < cb0 + 2; ++cb) {
      i32x4_t w = __builtin_bit_cast(i32x4_t, Bn[cb][tp]);
      w[2 * pc] = relu_pack<P>(acc[pc][cb][0], acc[pc][cb][1]);
      w[2 * pc + 1] = relu_pack<P>(acc[pc][cb][2], acc[pc][cb][3]);
      Bn[cb][tp] = __builtin_bit_cast(typename P::v8, w);
    }
  }
};

// CLS = false: DoNeRFTRT; CLS = true: the NeRF class (layer sequence as nerf_kernel<.., CLS>, feature_linear folded, alpha as a 9th tile)
// NCB = 2: 8 waves of 32 columns, two waves per SIMD (the default).  NCB = 4: 4 waves of 64 columns, one wave per SIMD: every weight fragment read
// from LDS feeds four MFMAs instead of two (tools/lds_mfma_probe.hip: 1.57 -> 1.81 PFLOP/s for the bare hidden-layer loop).
// NW: waves per workgroup.  16 / NCB (the default); NCB = 2, NW = 4: 128 rows per batch, a SIMD per wave (as sampler_h16_kernel<4>).
template <bool CLS, int NCB = 2, class P = PrecBf16, int NW = 16 / NCB>
__global__ __launch_bounds__(64 * NW, NCB == 2 ? 2 : 1) void nerf16_kernel(NerfArgs a) {
  constexpr int TPB = 64 * NW;
  using Epi = HiddenEpi16<NCB, P>;
  using v8 = typename P::v8;
  P::enter();
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* bias_lds = (float*)(smem + RING_BYTES);
  WStream<NW> st;
  st.init(a.blob, a.nslots, smem);
  st.prologue();
  for (int i = threadIdx.x; i < a.nbias; i += TPB) bias_lds[i] = a.bias[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c16 = lane & 15, g = lane >> 4;
  int pe_c[4];                 // chain j4 of this lane group: component and 2^(first octave) (nerf16_in0)
  float pe_sc[4];
#pragma unroll
  for (int j4 = 0; j4 < 4; ++j4) {
    const int q = 4 * g + j4;
    pe_c[j4] = q % 3;
    pe_sc[j4] = (float)(1 << (2 * (q / 3)));
  }
  const float pe_vs = (float)(1 << g);
  const int64_t nrows = a.n * a.S;
  own_the_simd<NW>();
  young_half_priority<NW>();
  const char* ringlane = smem + lane * 16;
  const float* biaslane = bias_lds + 4 * g;
  const bool composite = a.S == 8 && a.rgbd;

  // Batch hand-out.  Without a queue (operator-level calls): the static stride.  With one (a context's calls): a workgroup's first batch is its
  // static one, every further one comes from an atomic counter — a workgroup that starts late because another stream's kernel held its CU (a
  // collective beside the renderer at N > 1: tools/cu_steal_probe.py) then finds the work done instead of running its whole share of ~12-93
  // batches after everybody else has finished.  The counter is fetched at the top of a batch and consumed at its end (its latency is hidden);
  // the last workgroup to leave puts the two words back to zero for the next launch on this context.
  __shared__ __attribute__((aligned(16))) int s_next[4];
  for (int batch = blockIdx.x; batch < a.nbatch;) {
    int q_pend = 0;
    if (a.queue && threadIdx.x == 0) q_pend = atomicAdd(a.queue, 1);
    int64_t row[NCB];
    bool valid[NCB];
    v8 Bo[NCB][NB_KS_H], Bn[NCB][NB_KS_H], Bx[NCB];
    float e_dn[NCB], e_z[NCB], e_add[NCB], e_mul[NCB], e_noise[NCB];
    static_for<NCB>([&](auto cbc) {
      constexpr int cb = decltype(cbc)::value;
      row[cb] = (int64_t)batch * (NW * 16 * NCB) + wave * (16 * NCB) + cb * 16 + c16;
      valid[cb] = row[cb] < nrows;
      const int64_t rr = valid[cb] ? row[cb] : nrows - 1;
      const float* pp = a.pts + rr * 3;
      const float* ry = a.rays + (a.S == 8 ? (rr >> 3) : rr / a.S) * 11;
      const float x3[3] = {pp[0], pp[1], pp[2]};
      const float v3[3] = {ry[8], ry[9], ry[10]};
      if (composite) {
        e_dn[cb] = ieee_sqrt(ieee_add(ieee_add(ieee_mul(ry[3], ry[3]), ieee_mul(ry[4], ry[4])), ieee_mul(ry[5], ry[5])));
        e_z[cb] = a.z[rr]; e_add[cb] = a.add ? a.add[rr] : 0.f; e_mul[cb] = a.mul ? a.mul[rr] : 1.f;
        e_noise[cb] = a.noise ? a.noise[rr] : 0.f;
      }
      // positional encoding straight into B-fragment order (see pe_sincos / the recurrence note in nerf_kernel)
      float f0[16], fx[8];
      // positional encoding in the slot order of nerf16_in0 / nerf16_inx: four chains of two octaves per lane (hardware sin / cos at the
      // chain's first octave, one double-angle step for the second), the view octave of the lane group directly
#pragma unroll
      for (int j4 = 0; j4 < 4; ++j4) {
        const float xv = pe_c[j4] == 0 ? x3[0] : (pe_c[j4] == 1 ? x3[1] : x3[2]);
        float s, co;
        pe_sincos_scaled(xv, pe_sc[j4], s, co);
        f0[4 * j4 + 0] = s; f0[4 * j4 + 1] = co;
        f0[4 * j4 + 2] = 2.f * s * co; f0[4 * j4 + 3] = (co - s) * (co + s);
      }
      if (g == 3) { f0[12] = x3[0]; f0[13] = x3[1]; f0[14] = x3[2]; f0[15] = 0.f; }
#pragma unroll
      for (int c = 0; c < 3; ++c) pe_sincos_scaled(v3[c], pe_vs, fx[2 * c], fx[2 * c + 1]);
      fx[6] = g == 0 ? v3[0] : (g == 1 ? v3[2] : 0.f);
      fx[7] = g == 0 ? v3[1] : 0.f;
#pragma unroll
      for (int ks = 0; ks < NB_KS0; ++ks) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = f0[ks * 8 + j];
        Bo[cb][ks] = P::pack(v);
      }
      Bx[cb] = P::pack(fx);
    });
    f32x4 fin[2][NCB], pend[2][NCB];
    if constexpr (!CLS) {
      auto hidden = [&](v8(&in)[NCB][NB_KS_H], v8(&out)[NCB][NB_KS_H], int l) {
        f32x4 np[2][NCB];
        layer_b16<NB_KS_H, NB_NTP_H, NB_POS_H, NCB, v8>(st, ringlane, biaslane + (1 + l) * W_HID, [&](int cb, int ks) { return in[cb][ks]; }, Epi{out},
                                               [&](int pc, int cb0) { Epi{in}(NB_NTP_H - 1, pc, pend, cb0); }, np);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) pend[t][cb] = np[t][cb];
      };
      layer_b16<NB_KS0, NB_NTP_H, 0, NCB, v8>(st, ringlane, biaslane, [&](int cb, int ks) { return Bo[cb][ks < NB_KS0 ? ks : 0]; }, Epi{Bn}, [](int, int) {}, pend);
      // Run-time layer count without a register move on any path: two layer bodies (Bn -> Bo, Bo -> Bn) in a loop that leaves after the first one when the
      // count is odd, and the (small) output layer once per place the activations can end in.  (A single output layer behind an "odd: move Bo over to Bn"
      // branch inside the loop made the compiler keep the activations in ONE register set and copy 64 + 112 registers per pair of layers on the main
      // path — every layer read the low set; found in the disassembly in round 6, -DPNRF_FIXED_NHID had no such moves.)
      const int nhid = PNRF_NHID(a.nhid, N_NHID);
      auto last_layer = [&](v8(&in)[NCB][NB_KS_H]) {
        layer_b16<NB_KS_LAST, 1, NB_POS_LAST, NCB, v8, PNRF_LAST_TMAX>(
            st, ringlane, biaslane + (1 + nhid) * W_HID, [&](int cb, int ks) { return ks < NB_KS_H ? in[cb][ks < NB_KS_H ? ks : 0] : Bx[cb]; },
            [&](int, int, f32x4(&)[2][NCB], int) {}, [&](int pc, int cb0) { Epi{in}(NB_NTP_H - 1, pc, pend, cb0); }, fin);
      };
      bool in_bo = false;
      for (int l = 0; l < nhid; l += 2) {
        hidden(Bn, Bo, l);
        if (l + 1 >= nhid) { in_bo = true; break; }
        hidden(Bo, Bn, l + 1);
      }
      if (in_bo) last_layer(Bo);
      else last_layer(Bn);
#pragma unroll
      for (int i = 0; i < NB_SLOTS_PAD; ++i) st.begin();
    } else {
      // E0 Bo->Bn | E1..E4 ping-pong (ends in Bn) | E5 [Bn, P]->Bo | E6 Bo->Bn | E7 Bn->Bo | E89 [Bo, Bx]->Bn (128 wide) + alpha | E10 Bn -> rgb
      v8 Pz[NCB][NB_KS0];                  // positional fragments, needed again by the skip connection at layer 5
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int k = 0; k < NB_KS0; ++k) Pz[cb][k] = Bo[cb][k];
      auto hidden = [&](v8(&in)[NCB][NB_KS_H], v8(&out)[NCB][NB_KS_H], int l, auto posc) {
        constexpr int POS = decltype(posc)::value;
        f32x4 np[2][NCB];
        layer_b16<NB_KS_H, NB_NTP_H, POS, NCB, v8>(st, ringlane, biaslane + l * W_HID, [&](int cb, int ks) { return in[cb][ks]; }, Epi{out},
                                          [&](int pc, int cb0) { Epi{in}(NB_NTP_H - 1, pc, pend, cb0); }, np);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) pend[t][cb] = np[t][cb];
      };
      layer_b16<NB_KS0, NB_NTP_H, 0, NCB, v8>(st, ringlane, biaslane, [&](int cb, int ks) { return Bo[cb][ks < NB_KS0 ? ks : 0]; }, Epi{Bn}, [](int, int) {}, pend);
      for (int l = 1; l < 5; l += 2) {        // E1..E4
        hidden(Bn, Bo, l, std::integral_constant<int, CB_POS_E1>{});
        hidden(Bo, Bn, l + 1, std::integral_constant<int, CB_POS_E1>{});
      }
      {                                       // E5: cat[pts, h] -> 256 (skip connection after layer 4)
        f32x4 np[2][NCB];
        layer_b16<CB_KS5, NB_NTP_H, CB_POS_E5, NCB, v8>(
            st, ringlane, biaslane + 5 * W_HID,
            [&](int cb, int ks) { return ks < NB_KS_H ? Bn[cb][ks < NB_KS_H ? ks : 0] : Pz[cb][ks >= NB_KS_H ? ks - NB_KS_H : 0]; }, Epi{Bo},
            [&](int pc, int cb0) { Epi{Bn}(NB_NTP_H - 1, pc, pend, cb0); }, np);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) pend[t][cb] = np[t][cb];
      }
      hidden(Bo, Bn, 6, std::integral_constant<int, CB_POS_E6>{});
      hidden(Bn, Bo, 7, std::integral_constant<int, CB_POS_E6>{});
      f32x4 al[2][NCB];                         // E89: view tiles (pairs 0..3, ReLU -> Bn k-steps 0..3) + alpha (pair 4, tile 0, row 0)
      layer_b16<NB_KS_LAST, CB_NTP89, CB_POS_E89, NCB, v8>(
          st, ringlane, biaslane + CB_BIAS_E89, [&](int cb, int ks) { return ks < NB_KS_H ? Bo[cb][ks < NB_KS_H ? ks : 0] : Bx[cb]; }, Epi{Bn},
          [&](int pc, int cb0) { Epi{Bo}(NB_NTP_H - 1, pc, pend, cb0); }, al);
      layer_b16<CB_KS10, 1, CB_POS_E10, NCB, v8, PNRF_LAST_TMAX>(st, ringlane, biaslane + CB_BIAS_E10, [&](int cb, int ks) { return Bn[cb][ks < CB_KS10 ? ks : 0]; },
                                        [&](int, int, f32x4(&)[2][NCB], int) {}, [](int, int) {}, fin);
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) fin[0][cb][3] = al[0][cb][0];          // raw = [rgb, alpha] (helpers:851)
#pragma unroll
      for (int i = 0; i < CB_SLOTS_PAD; ++i) st.begin();
    }

    // ---- fused epilogue (as nerf_kernel): group 0 holds raw rgb-sigma of its column in regs 0-3 of tile 0
    static_for<NCB>([&](auto cbc) {
      constexpr int cb = decltype(cbc)::value;
      const int64_t rr = valid[cb] ? row[cb] : nrows - 1;
      float r0 = fin[0][cb][0], r1 = fin[0][cb][1], r2 = fin[0][cb][2], r3 = fin[0][cb][3];
      if (a.raw && valid[cb] && g == 0) *(float4*)(a.raw + row[cb] * 4) = make_float4(r0, r1, r2, r3);
      if (!composite) return;
      const int64_t ray = rr >> 3;
      const int s = (int)(rr & 7);
      const float dn = e_dn[cb], zc = e_z[cb], ad = e_add[cb], mu = e_mul[cb];
      if (a.clampv > 0.f) {                                                         // base.py:523
        r0 = fminf(fmaxf(r0, -a.clampv), a.clampv); r1 = fminf(fmaxf(r1, -a.clampv), a.clampv);
        r2 = fminf(fmaxf(r2, -a.clampv), a.clampv); r3 = fminf(fmaxf(r3, -a.clampv), a.clampv);
      }
      // the 8 samples of a ray sit in 8 adjacent lanes: neighbour, exclusive product and the sums go through DPP lane moves
      // (row_shl / row_shr / quad_perm / row_half_mirror) instead of LDS permutes; tree order instead of torch's left-to-right order,
      // a difference of fp32 round-off under a bf16-grade network output
      const float znext = dpp_mov<0x101>(zc);                                       // row_shl:1
      float dist = (s < 7) ? ieee_sub(znext, zc) : 1e10f;                          // trt.py:579-581
      dist = ieee_mul(dist, dn);                                                   // :583
      const float cr = sigmoid_fast(r0), cg = sigmoid_fast(r1), cbv = sigmoid_fast(r2);   // :585
      const float sg = fmaxf(ieee_add(a.noise ? ieee_add(r3, e_noise[cb]) : r3, ad), 0.f);     // refine2.py:508
      float alpha = ieee_sub(1.f, __expf(ieee_mul(-sg, dist)));                   // :577,587
      if (a.mul) alpha = ieee_mul(alpha, fmaxf(mu, 0.f));                          // :588
      const float xk = ieee_add(ieee_sub(1.f, alpha), 1e-10f);                    // :590
      // exclusive cumprod: shift by one (row_shr:1), then scan.  Every lane move is executed by all lanes and selected afterwards: under
      // a branch the lanes switched off would read as zero in their neighbours' moves
      const float xprev = dpp_mov<0x111>(xk);
      float T = s >= 1 ? xprev : 1.f;
      { const float t = dpp_mov<0x111>(T); T = ieee_mul(T, s >= 1 ? t : 1.f); }
      { const float t = dpp_mov<0x112>(T); T = ieee_mul(T, s >= 2 ? t : 1.f); }
      { const float t = dpp_mov<0x114>(T); T = ieee_mul(T, s >= 4 ? t : 1.f); }
      const float wgt = ieee_mul(alpha, T);
      float s0 = sum8_dpp(ieee_mul(wgt, cr)), s1 = sum8_dpp(ieee_mul(wgt, cg)), s2 = sum8_dpp(ieee_mul(wgt, cbv));    // :591 sum over samples
      const float s3 = sum8_dpp(ieee_mul(wgt, zc));                                // :593 depth_map
      const float sa = a.white_bkgd ? sum8_dpp(wgt) : 0.f;                          // acc_map
      if (a.white_bkgd) {                                                           // refine2.py:519-520
        const float bg = ieee_sub(1.f, sa);
        s0 = ieee_add(s0, bg); s1 = ieee_add(s1, bg); s2 = ieee_add(s2, bg);
      }
      if (valid[cb] && g == 0 && s == 0) *(float4*)(a.rgbd + ray * 4) = make_float4(s0, s1, s2, s3);
    });
    if (a.queue) {
      if (threadIdx.x == 0) s_next[0] = (int)gridDim.x + q_pend;
      __syncthreads();
      batch = s_next[0];          // (rewritten at the end of the next batch, many slot barriers from here)
    } else {
      batch += gridDim.x;
    }
  }
  st.drain();
  if (a.queue && threadIdx.x == 0) {
    if (atomicAdd(a.queue + 1, 1) == (int)gridDim.x - 1) { a.queue[0] = 0; a.queue[1] = 0; }
  }
}

// ------------------------------------------------------------------------------------------ launch helpers
int g_num_cu = 0;
int num_cu() {
  if (!g_num_cu) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) g_num_cu = p.multiProcessorCount;
    if (g_num_cu <= 0) g_num_cu = 256;
  }
  return g_num_cu;
}

// Kernel variants are a property of the packed handle (pnrf_mlp_set_variant, include/pronerf_hip.h); nothing on the launch path reads the
// process environment.  Default: split-fp16 sampler with the folded first layer, refine and NeRF stages on v_mfma_f32_16x16x32_bf16.
template <class K, class A>
int launch_mlp(K kern, const A& a, int tpb, size_t lds, int nbatch, hipStream_t stream) {
  // the ring + bias region exceeds the 64 KiB default dynamic-LDS limit.  The attribute is per device, and the call is a table update in
  // the runtime, so it is simply made on every launch (a per-process "done" cache skipped it on a second GPU).
  {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((lds + 8191) & ~(size_t)4095));
    if (e != hipSuccess) {
      set_error("hipFuncSetAttribute(max dynamic LDS) failed: %s", hipGetErrorString(e));
      return (int)e;
    }
  }
  const int ncu = num_cu();
  const int grid = nbatch < ncu ? nbatch : ncu;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(tpb), lds, stream, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("kernel launch failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

// Workgroup shape of a fused stage for one launch (include/pronerf_hip.h, pnrf_mlp_set_shape):
//   WIDE    8 waves, one workgroup per CU, two waves per SIMD, the widest batch (every weight fragment the workgroup streams feeds 8 waves);
//   NARROW  4 waves, half-width batches, ONE workgroup per CU (a SIMD per wave): for calls of at most one narrow batch per CU (ray chunks),
//           whose time is the latency of one batch through the layers — two waves sharing a SIMD double it.
// Same results bit for bit (a wave's instruction stream per batch does not depend on the shape).
// A CU never holds two fused-MLP workgroups: wide ones exclude each other by registers (2 x 240 per SIMD lane), narrow ones by LDS (their
// request is padded to NARROW_LDS_BYTES > 80 KiB).  This is a PERFORMANCE choice: two 4-wave workgroups per CU ("paired" narrow launches) are
// 5-8 % slower than WIDE on whole frames in every stage (twice the weight stream per CU; the ELU kernels are bound by their VALU issue cycles,
// not by a phase lock of their waves: tools/elu_chain_probe.hip) and only 5 % faster on a 1/8-frame shard.
// (Round 4 also saw paired workgroups of DIFFERENT kernels return wrong rows — 16 consecutive rays of a refine wave — a few times per thousand
// calls and excluded the configuration without finding the cause.  Round 5 found it (NOTEBOOK §19): not the batch-head loads but compiler-generated
// PACKED FP32 arithmetic in the refine epilogue — loads, every layer's B operand and the last accumulators were bit-identical in the failing batches,
// a v_pk_mul_f32 of the query-point arithmetic returned 0 in the wave's last 16 lanes; tools/pkf32_coexec_probe.hip reproduces it in a micro-kernel and
// names the ingredients: the victim in MODE.FP16_OVFL, 240-register windows on both waves of the SIMD, a v_mfma_f32_16x16x32_bf16 partner (the NeRF
// stage).  Without packed-fp32 instructions the paired configuration is clean over 283 100 concurrent chunks (121-209 bad chunks per 29 800 with them).
// The library is built without them (pronerf_amd/build.py NO_PACKED_FP32; tests/test_abi_cpu.py checks the disassembly), so the shape rule below no
// longer carries correctness.)
enum { SHAPE_WIDE = PNRF_SHAPE_WIDE, SHAPE_NARROW = PNRF_SHAPE_NARROW };
#ifndef PNRF_NARROW_LDS_BYTES
#define PNRF_NARROW_LDS_BYTES (84 * 1024)
#endif
constexpr size_t NARROW_LDS_BYTES = PNRF_NARROW_LDS_BYTES;      // two of them do not fit a CU's 160 KiB (0: the paired shape of round 4, reproducer builds only — tools/coresidency_repro.py)
// cols: columns of the launch (rays or ray samples); cpw: columns per wave of the stage's engine
int stage_shape(const pnrf_mlp_t* h, int64_t cols, int cpw) {
  if (h->shape == SHAPE_WIDE || h->shape == SHAPE_NARROW) return h->shape;
  const int64_t nb4 = (cols + 4 * cpw - 1) / (4 * cpw);
  return nb4 <= num_cu() ? SHAPE_NARROW : SHAPE_WIDE;
}
size_t narrow_lds(size_t lds) { return lds > NARROW_LDS_BYTES ? lds : NARROW_LDS_BYTES; }

}  // namespace

// ------------------------------------------------------------------------------------------ C ABI
static int sampler_launch(const pnrf_mlp_t* h, const float* rays, int64_t n, float* depth_sorted, float* add_sorted, float* mul_sorted,
                          int64_t* sort_idx, float* mm_rgb, float* depth_raw, void* workspace, bool ws_clean, float kappa, void* stream) {
  SamplerArgs a = {};
  a.bias = h->d_bias; a.nbias = h->nbias; a.nhid = h->nhid;
  a.n = n; a.nbatch = (int)((n + 127) / 128);
  a.rays = rays; a.tvals = h->d_tvals;
  a.depth_sorted = depth_sorted; a.add_sorted = add_sorted; a.mul_sorted = mul_sorted;
  a.sort_idx = sort_idx; a.mm_rgb = mm_rgb; a.depth_raw = depth_raw;
  const size_t lds = RING_BYTES + (size_t)h->nbias * 4;
  hipStream_t st = (hipStream_t)stream;
  // split-fp16 kernel (16 columns per wave): 128 / 64 rays per workgroup batch.  Pass 2 renders a list whose length is known on the device
  // only (a tenth of the rays on most frames, a quarter in places): its shape is chosen as if every ray were on it — a narrow launch that meets
  // a long list walks two or three batches per workgroup at one-wave-per-SIMD latency (round 4: the last 1/8-frame shard took 0.707 ms, the
  // others 0.63), a wide launch that meets a short one loses 9 us
  auto launch_h16 = [&](SamplerArgs& x, int64_t rows, int64_t expect) {
    if (stage_shape(h, expect, 16) == SHAPE_NARROW) {
      x.nbatch = (int)((rows + 63) / 64);
      return launch_mlp(sampler_h16_kernel<4>, x, 256, narrow_lds(lds), x.nbatch, st);
    }
    x.nbatch = (int)((rows + 127) / 128);
    return launch_mlp(sampler_h16_kernel<8>, x, 512, lds, x.nbatch, st);
  };
  // pass 3: a few workgroups (the saturated list is empty on every net whose activations stay inside the fp16 range: they leave at once)
  auto launch_f32_list = [&](SamplerArgs& f, int* counters, int* sat_list) {
    f.blob = h->d_blob_fold; f.nslots = h->nslots_fold;
    f.list = sat_list; f.list_count = counters + 3; f.counters = counters; f.sat_list = nullptr;
    const int nb = (int)((n + 127) / 128);
    return launch_mlp(sampler_kernel<2>, f, 512, lds, nb < 16 ? nb : 16, st);
  };
  void* workspace_split = nullptr;
  if (workspace && h->variant == PNRF_VARIANT_SAMPLER_SPLIT) { workspace_split = workspace; workspace = nullptr; }
  if (workspace) {                      // two passes: plain fp16 for every ray, split fp16 for the rays pass 1 cannot decide
    int* counters = (int*)workspace;
    // counters[0] rays on the list, [1] the same for pnrf_ctx_sampler_stats, [2] finished workgroups of pass 2, [3] rays on the saturated list,
    // [4] finished workgroups of pass 3, [5] = [3] for the stats.  The last workgroups of passes 2 / 3 leave [0], [2] / [3], [4] at zero, so a
    // workspace that has been through a call (or was cleared once, as a context's is) needs no memset.
    if (!ws_clean) PNRF_HIP(hipMemsetAsync(counters, 0, 16 * sizeof(int), st));      // (the whole header: words 8, 9 are the context's NeRF-stage queue)
    SamplerArgs p = a;
    p.blob = h->d_blob_p1; p.nslots = h->nslots_p1; p.bias = h->d_bias_p1; p.nbias = h->nbias_p1;
    p.list = counters + 16; p.counters = counters; p.p1c = h->d_p1c; p.kappa = kappa;
    const size_t lds1 = RING_BYTES + (size_t)h->nbias_p1 * 4;
    int rc;
    if (stage_shape(h, n, 32) == SHAPE_NARROW) {
      p.nbatch = (int)((n + 127) / 128);
      rc = launch_mlp(sampler_p1_kernel<4>, p, 256, narrow_lds(lds1), p.nbatch, st);
    } else {
      p.nbatch = (int)((n + 255) / 256);
      rc = launch_mlp(sampler_p1_kernel<8>, p, 512, lds1, p.nbatch, st);
    }
    if (rc) return rc;
    SamplerArgs f = a;                                    // pass 3 (exact fp32, folded first layer) on the rays pass 2 reports as saturated
    a.blob = h->d_blob_h16; a.nslots = h->nslots_h16;
    a.list = counters + 16; a.counters = counters; a.list_count = counters; a.sat_list = counters + 16 + n;
    // the grid is sized for a list of every ray (workgroups beyond the list leave at once); the shape for a sixth of them on it (8-27 % on the
    // weight sets of tests/test_fullframe_gpu.py; a list of up to 40 % still costs a narrow launch no more than a wide one)
    if ((rc = launch_h16(a, n, (n + 5) / 6))) return rc;
    return launch_f32_list(f, counters, counters + 16 + n);
  }
  if (workspace_split) {                // PNRF_VARIANT_SAMPLER_SPLIT with a workspace: the split kernel for every ray + the exact-fp32 pass for saturated ones
    int* counters = (int*)workspace_split;
    if (!ws_clean) PNRF_HIP(hipMemsetAsync(counters, 0, 16 * sizeof(int), st));      // (the whole header: words 8, 9 are the context's NeRF-stage queue)
    SamplerArgs f = a;
    a.blob = h->d_blob_h16; a.nslots = h->nslots_h16;
    a.counters = counters; a.sat_list = counters + 16 + n;
    int rc;
    if ((rc = launch_h16(a, n, n))) return rc;
    return launch_f32_list(f, counters, counters + 16 + n);
  }
  if (h->variant == PNRF_VARIANT_SAMPLER_F32_FULL) {
    PNRF_REQUIRE(h->npts == S_NPTS && h->d_blob, PNRF_E_SHAPE, "PNRF_VARIANT_SAMPLER_F32_FULL (the unfolded first layer) is built for N_point_ray_enc = %d; this sampler has %d "
                 "(every other variant runs the folded first layer and takes any N_point_ray_enc)", S_NPTS, h->npts);
    a.blob = h->d_blob; a.nslots = h->nslots;
    return launch_mlp(sampler_kernel<1>, a, 512, lds, a.nbatch, st);
  }
  if (h->variant == PNRF_VARIANT_SAMPLER_F32) {
    a.blob = h->d_blob_fold; a.nslots = h->nslots_fold;
    return launch_mlp(sampler_kernel<2>, a, 512, lds, a.nbatch, st);
  }
  a.blob = h->d_blob_h16; a.nslots = h->nslots_h16;
  return launch_h16(a, n, n);
}

extern "C" int pnrf_sampler_fwd(const pnrf_mlp_t* h, const float* rays, int64_t n, float* depth_sorted,
                                float* add_sorted, float* mul_sorted, int64_t* sort_idx, float* mm_rgb,
                                float* depth_raw, void* stream) {
  PNRF_REQUIRE(h && h->net == PNRF_NET_SAMPLER, PNRF_E_ARG, "pnrf_sampler_fwd: handle is not a sampler net");
  PNRF_REQUIRE(n >= 0 && (n == 0 || (rays && depth_sorted && add_sorted && mul_sorted)), PNRF_E_ARG, "pnrf_sampler_fwd: null pointer / negative n");
  if (n == 0) return 0;
  return sampler_launch(h, rays, n, depth_sorted, add_sorted, mul_sorted, sort_idx, mm_rgb, depth_raw, nullptr, false, 0.f, stream);
}

// 16 counters, the list of pass 2 (n entries), the list of pass 3 (n entries)
extern "C" int64_t pnrf_sampler_workspace_bytes(int64_t n) { return n < 0 ? 0 : (int64_t)(16 + 2 * n) * (int64_t)sizeof(int); }

// ws_clean: the workspace's counters are known to be zero (a context's workspace: cleared at creation, left clean by every call)
int pnrf_sampler_fwd_ws_impl(const pnrf_mlp_t* h, const float* rays, int64_t n, float* depth_sorted, float* add_sorted, float* mul_sorted,
                             int64_t* sort_idx, float* mm_rgb, float* depth_raw, void* workspace, int64_t workspace_bytes, float kappa,
                             bool ws_clean, void* stream) {
  PNRF_REQUIRE(h && h->net == PNRF_NET_SAMPLER, PNRF_E_ARG, "pnrf_sampler_fwd_ws: handle is not a sampler net");
  PNRF_REQUIRE(n >= 0 && n < ((int64_t)1 << 31) && (n == 0 || (rays && depth_sorted && add_sorted && mul_sorted)), PNRF_E_ARG,
               "pnrf_sampler_fwd_ws: null pointer / negative n / more than 2^31 rays");
  if (n == 0) return 0;
  PNRF_REQUIRE(workspace && workspace_bytes >= pnrf_sampler_workspace_bytes(n) && ((uintptr_t)workspace & 15) == 0, PNRF_E_ARG,
               "pnrf_sampler_fwd_ws: workspace of %lld bytes (16-byte aligned) needed for %lld rays, got %lld", (long long)pnrf_sampler_workspace_bytes(n),
               (long long)n, (long long)workspace_bytes);
  PNRF_REQUIRE(kappa == kappa && kappa < 1e30f, PNRF_E_ARG, "pnrf_sampler_fwd_ws: kappa must be a number below 1e30 (negative = default), got %g", (double)kappa);
  if (h->variant != PNRF_VARIANT_DEFAULT && h->variant != PNRF_VARIANT_SAMPLER_SPLIT)      // the exact-fp32 variants of a handle ignore the workspace
    return sampler_launch(h, rays, n, depth_sorted, add_sorted, mul_sorted, sort_idx, mm_rgb, depth_raw, nullptr, false, 0.f, stream);
  return sampler_launch(h, rays, n, depth_sorted, add_sorted, mul_sorted, sort_idx, mm_rgb, depth_raw, workspace, ws_clean,
                        kappa < 0.f ? PNRF_SAMPLER_KAPPA : kappa, stream);
}

extern "C" int pnrf_sampler_fwd_ws(const pnrf_mlp_t* h, const float* rays, int64_t n, float* depth_sorted, float* add_sorted,
                                   float* mul_sorted, int64_t* sort_idx, float* mm_rgb, float* depth_raw, void* workspace,
                                   int64_t workspace_bytes, float kappa, void* stream) {
  return pnrf_sampler_fwd_ws_impl(h, rays, n, depth_sorted, add_sorted, mul_sorted, sort_idx, mm_rgb, depth_raw, workspace, workspace_bytes, kappa,
                                  false, stream);
}

// The fused refine stage in its two workgroup shapes (256 / 128 rays per batch; stage_shape) and two operand types
template <int MODE, int HEAD, int NV>
static int refine_launch_nv(const pnrf_mlp_t* h, RefineArgs& a, int64_t n, hipStream_t st) {
  const size_t lds = RING_BYTES + (size_t)h->nbias * 4;
  const bool bf16 = h->variant == PNRF_VARIANT_BF16;
  if constexpr (NV > 2 && HEAD == 1) {          // (its projecting head with 3 / 4 views per lane does not fit the register file beside bf16 packing: it would spill)
    PNRF_REQUIRE(!bf16, PNRF_E_SHAPE, "PNRF_VARIANT_BF16 of the projecting refine stage exists for num_neighbor <= 4 (this net: %d); the default fp16 stage takes 1 .. 8", h->nb);
    a.blob = h->d_blob_f16; a.nslots = h->nslots_f16;
    if (stage_shape(h, n, 32) == SHAPE_NARROW) {
      a.nbatch = (int)((n + 127) / 128);
      return launch_mlp(refine_kernel<1, 4, MODE, HEAD, PrecF16, NV>, a, 256, narrow_lds(lds), a.nbatch, st);
    }
    a.nbatch = (int)((n + 255) / 256);
    return launch_mlp(refine_kernel<1, 8, MODE, HEAD, PrecF16, NV>, a, 512, lds, a.nbatch, st);
  } else {
  if (!bf16) { a.blob = h->d_blob_f16; a.nslots = h->nslots_f16; }
  if (stage_shape(h, n, 32) == SHAPE_NARROW) {
    a.nbatch = (int)((n + 127) / 128);
    if (bf16) return launch_mlp(refine_kernel<1, 4, MODE, HEAD, PrecBf16, NV>, a, 256, narrow_lds(lds), a.nbatch, st);
    return launch_mlp(refine_kernel<1, 4, MODE, HEAD, PrecF16, NV>, a, 256, narrow_lds(lds), a.nbatch, st);
  }
  a.nbatch = (int)((n + 255) / 256);
  if (bf16) return launch_mlp(refine_kernel<1, 8, MODE, HEAD, PrecBf16, NV>, a, 512, lds, a.nbatch, st);
  return launch_mlp(refine_kernel<1, 8, MODE, HEAD, PrecF16, NV>, a, 512, lds, a.nbatch, st);
  }
}
// The inference stage on the 16x16x32 engine (refine16_kernel; PNRF_VARIANT_REFINE_16X16): its own stream and bias table (d_blob_b16 / d_bias_b16 of a refine handle)
template <int HEAD, int NV4>
static int refine16_launch_nv(const pnrf_mlp_t* h, RefineArgs& a, int64_t n, hipStream_t st) {
  const size_t lds = RING_BYTES + (size_t)h->nbias_b16 * 4;
  if (stage_shape(h, n, 32) == SHAPE_NARROW) {
    a.nbatch = (int)((n + 127) / 128);
    return launch_mlp(refine16_kernel<4, HEAD, NV4>, a, 256, narrow_lds(lds), a.nbatch, st);
  }
  a.nbatch = (int)((n + 255) / 256);
  return launch_mlp(refine16_kernel<8, HEAD, NV4>, a, 512, lds, a.nbatch, st);
}
template <int HEAD>
static int refine16_launch(const pnrf_mlp_t* h, RefineArgs& a, int64_t n, hipStream_t st) {
  constexpr bool FOLD = HEAD == 1;          // the projecting head runs the folded first layer (d_blob_fold), rows from memory the full one (d_blob_b16)
  const void* blob = FOLD ? h->d_blob_fold : h->d_blob_b16;
  const uint32_t ns = FOLD ? h->nslots_fold : h->nslots_b16;
  PNRF_REQUIRE(blob && h->d_bias_b16 && ns == (uint32_t)refine16_slots(h->nhid, refine16_nv(h->nb), FOLD), PNRF_E_ARG,
               "refine handle without the stream of the 16x16x32 engine");
  a.blob = blob; a.nslots = ns; a.bias = h->d_bias_b16; a.nbias = h->nbias_b16;
  return refine16_nv(h->nb) == 1 ? refine16_launch_nv<HEAD, 1>(h, a, n, st) : refine16_launch_nv<HEAD, 2>(h, a, n, st);
}
// ... for the handle's number of neighbour views (NV = views per lane half, pnrf_layout.h) and hidden layers
template <int MODE, int HEAD>
static int refine_launch(const pnrf_mlp_t* h, RefineArgs& a, int64_t n, hipStream_t st) {
  a.nhid = h->nhid; a.nb = h->nb;
  if constexpr (MODE == 1) {
    if (h->variant == PNRF_VARIANT_REFINE_16X16) return refine16_launch<HEAD>(h, a, n, st);
  }
  if constexpr (MODE == 2) {                    // the training-time epilogue belongs to the trainer's shapes
    PNRF_REQUIRE(h->nb == 4, PNRF_E_SHAPE, "pnrf_refine_train_fwd is built for num_neighbor = 4, this refine net has %d", h->nb);
    return refine_launch_nv<MODE, HEAD, 2>(h, a, n, st);
  } else {
    switch (refine_nv(h->nb)) {
      case 1: return refine_launch_nv<MODE, HEAD, 1>(h, a, n, st);
      case 2: return refine_launch_nv<MODE, HEAD, 2>(h, a, n, st);
      case 3: return refine_launch_nv<MODE, HEAD, 3>(h, a, n, st);
      default: return refine_launch_nv<MODE, HEAD, 4>(h, a, n, st);
    }
  }
}

extern "C" int pnrf_refine_train_fwd(const pnrf_mlp_t* h, const float* refine_in, const float* rays, const float* depth_sorted,
                                     const float* jitter, int jitter_dir, float* z, float* pts, float* rgb0, int64_t n, void* stream) {
  PNRF_REQUIRE(h && h->net == PNRF_NET_REFINE, PNRF_E_ARG, "pnrf_refine_train_fwd: handle is not a refine net");
  PNRF_REQUIRE(n >= 0 && (n == 0 || (refine_in && rays && depth_sorted && z && pts)), PNRF_E_ARG, "pnrf_refine_train_fwd: null pointer / negative n");
  PNRF_REQUIRE(!jitter || jitter_dir == 1 || jitter_dir == -1, PNRF_E_ARG, "pnrf_refine_train_fwd: jitter_dir must be +1 or -1");
  if (n == 0) return 0;
  RefineArgs a = {};
  a.blob = h->d_blob; a.bias = h->d_bias; a.nslots = h->nslots; a.nbias = h->nbias;
  a.n = n;
  a.x = refine_in; a.rays = rays; a.depth_sorted = depth_sorted; a.z = z; a.pts = pts;
  a.jitter = jitter; a.jitter_dir = jitter_dir; a.rgb0 = rgb0;
  return refine_launch<2, 0>(h, a, n, (hipStream_t)stream);
}

extern "C" int pnrf_refine_fwd(const pnrf_mlp_t* h, const float* refine_in, const float* rays,
                               const float* depth_sorted, float* z, float* pts, int64_t n, void* stream) {
  PNRF_REQUIRE(h && h->net == PNRF_NET_REFINE, PNRF_E_ARG, "pnrf_refine_fwd: handle is not a refine net");
  PNRF_REQUIRE(n >= 0 && (n == 0 || (refine_in && rays && depth_sorted && z && pts)), PNRF_E_ARG, "pnrf_refine_fwd: null pointer / negative n");
  if (n == 0) return 0;
  RefineArgs a = {};
  a.blob = h->d_blob; a.bias = h->d_bias; a.nslots = h->nslots; a.nbias = h->nbias;
  a.n = n;
  a.x = refine_in; a.rays = rays; a.depth_sorted = depth_sorted; a.z = z; a.pts = pts;
  return refine_launch<1, 0>(h, a, n, (hipStream_t)stream);
}

extern "C" int pnrf_refine_project_fwd(const pnrf_mlp_t* h, const float* rays, const float* or_rays, const float* depth_sorted, const float* img4,
                                       const float* proj, int nb, int Hf, int Wf, float eps, float* z, float* pts, int64_t n, void* stream) {
  PNRF_REQUIRE(h && h->net == PNRF_NET_REFINE, PNRF_E_ARG, "pnrf_refine_project_fwd: handle is not a refine net");
  PNRF_REQUIRE(n >= 0 && nb == h->nb && Hf >= 2 && Wf >= 2 && (int64_t)Hf * Wf * nb < (int64_t)1 << 27, PNRF_E_ARG,
               "pnrf_refine_project_fwd: bad sizes (nb must be the refine net's num_neighbor = %d, got %d; nb * Hf * Wf must stay below 2^27 texels)", h->nb, nb);
  if (n == 0) return 0;
  PNRF_REQUIRE(rays && or_rays && depth_sorted && img4 && proj && z && pts, PNRF_E_ARG, "pnrf_refine_project_fwd: null pointer");

  RefineArgs a = {};
  a.blob = h->d_blob; a.bias = h->d_bias; a.nslots = h->nslots; a.nbias = h->nbias;
  a.n = n;
  a.or_rays = or_rays; a.img4 = (const float4*)img4; a.proj = proj; a.Hf = Hf; a.Wf = Wf; a.eps = eps;
  a.rays = rays; a.depth_sorted = depth_sorted; a.z = z; a.pts = pts;
  return refine_launch<1, 1>(h, a, n, (hipStream_t)stream);
}

extern "C" int pnrf_nerf_fwd(const pnrf_mlp_t* h, const float* pts, const float* rays, const float* z,
                             const float* add_sorted, const float* mul_sorted, float* rgbd, float* raw,
                             int64_t n, void* stream) {
  PNRF_REQUIRE(n == 0 || (z && add_sorted && mul_sorted && rgbd), PNRF_E_ARG, "pnrf_nerf_fwd: null pointer");
  return pnrf_nerf_train_fwd(h, pts, rays, z, add_sorted, mul_sorted, nullptr, 0.f, 0, 8, rgbd, raw, n, stream);
}

extern "C" int pnrf_nerf_train_fwd(const pnrf_mlp_t* h, const float* pts, const float* rays, const float* z,
                                   const float* add_sorted, const float* mul_sorted, const float* noise, float clampv, int white_bkgd,
                                   int S, float* rgbd, float* raw, int64_t n, void* stream) {
  return pnrf_nerf_fwd_queue_impl(h, pts, rays, z, add_sorted, mul_sorted, noise, clampv, white_bkgd, S, rgbd, raw, n, nullptr, stream);
}

// queue: two ints of the caller's (a context's) device memory, zero before the first launch and left at zero by every launch (nerf16_kernel);
// NULL = static batch stride
int pnrf_nerf_fwd_queue_impl(const pnrf_mlp_t* h, const float* pts, const float* rays, const float* z,
                             const float* add_sorted, const float* mul_sorted, const float* noise, float clampv, int white_bkgd,
                             int S, float* rgbd, float* raw, int64_t n, int* queue, void* stream) {
  PNRF_REQUIRE(h && (h->net == PNRF_NET_NERF || h->net == PNRF_NET_NERFCLS), PNRF_E_ARG, "pnrf_nerf_fwd: handle is not a nerf net");
  PNRF_REQUIRE(n >= 0 && S >= 1 && (n == 0 || (pts && rays)), PNRF_E_ARG, "pnrf_nerf_fwd: null pointer / negative n / bad S");
  PNRF_REQUIRE(n == 0 || (S == 8 ? ((rgbd && z) || raw) : (raw && !rgbd)), PNRF_E_ARG,
               "pnrf_nerf_train_fwd: S == 8 needs rgbd (+z) or raw; S != 8 writes raw only (composite with pnrf_composite_fwd)");
  PNRF_REQUIRE((add_sorted == nullptr) == (mul_sorted == nullptr), PNRF_E_ARG, "pnrf_nerf_train_fwd: add and mul go together");
  if (n == 0) return 0;
  NerfArgs a = {};
  a.blob = h->d_blob; a.bias = h->d_bias; a.nslots = h->nslots; a.nbias = h->nbias;
  a.n = n;
  a.pts = pts; a.rays = rays; a.z = z; a.add = add_sorted; a.mul = mul_sorted; a.rgbd = rgbd; a.raw = raw;
  a.noise = noise; a.white_bkgd = white_bkgd; a.clampv = clampv; a.S = S; a.nhid = h->nhid;
  a.queue = h->variant == PNRF_VARIANT_BF16_32X32 ? nullptr : queue;          // (nerf_kernel, the 32x32x16 variant, keeps the static stride)
  const size_t lds = RING_BYTES + (size_t)h->nbias * 4;
  const int rows = 256;
  a.nbatch = (int)((n * S + rows - 1) / rows);
  // default: the 16x16x32 engine on bf16 operands; PNRF_VARIANT_F16: the same engine on fp16 operands; NERF_4X64 / BF16_32X32: bf16 variants
  const bool b16 = h->variant != PNRF_VARIANT_BF16_32X32;
  const bool f16 = h->variant == PNRF_VARIANT_F16;
  if (b16) {
    a.blob = f16 ? h->d_blob_f16 : h->d_blob_b16; a.nslots = f16 ? h->nslots_f16 : h->nslots_b16; a.bias = h->d_bias_b16; a.nbias = h->nbias_b16;
  }
  const size_t lds16 = RING_BYTES + (size_t)h->nbias_b16 * 4;
  hipStream_t st = (hipStream_t)stream;
  // the 16x16x32 engine (two 16-column blocks per wave) in its two workgroup shapes: 256 rows (8 waves) or 128 rows (4 waves) per batch
  const bool narrow = b16 && h->variant != PNRF_VARIANT_NERF_4X64 && stage_shape(h, n * S, 32) == SHAPE_NARROW;
  if (narrow) a.nbatch = (int)((n * S + 127) / 128);
  if (h->net == PNRF_NET_NERFCLS) {
    if (narrow) return f16 ? launch_mlp(nerf16_kernel<true, 2, PrecF16, 4>, a, 256, narrow_lds(lds16), a.nbatch, st)
                           : launch_mlp(nerf16_kernel<true, 2, PrecBf16, 4>, a, 256, narrow_lds(lds16), a.nbatch, st);
    if (f16) return launch_mlp(nerf16_kernel<true, 2, PrecF16>, a, 512, lds16, a.nbatch, st);
    if (h->variant == PNRF_VARIANT_NERF_4X64) return launch_mlp(nerf16_kernel<true, 4>, a, 256, lds16, a.nbatch, st);
    if (b16) return launch_mlp(nerf16_kernel<true, 2>, a, 512, lds16, a.nbatch, st);
    return launch_mlp(nerf_kernel<1, 8, true, true>, a, 512, lds, a.nbatch, st);
  }
  if (narrow) return f16 ? launch_mlp(nerf16_kernel<false, 2, PrecF16, 4>, a, 256, narrow_lds(lds16), a.nbatch, st)
                         : launch_mlp(nerf16_kernel<false, 2, PrecBf16, 4>, a, 256, narrow_lds(lds16), a.nbatch, st);
  if (f16) return launch_mlp(nerf16_kernel<false, 2, PrecF16>, a, 512, lds16, a.nbatch, st);
  if (h->variant == PNRF_VARIANT_NERF_4X64) return launch_mlp(nerf16_kernel<false, 4>, a, 256, lds16, a.nbatch, st);
  if (b16) return launch_mlp(nerf16_kernel<false, 2>, a, 512, lds16, a.nbatch, st);
  PNRF_REQUIRE(h->nhid == N_NHID, PNRF_E_SHAPE, "PNRF_VARIANT_BF16_32X32 is built for netdepth %d; this net has netdepth %d — the default engine takes any", N_NHID + 2, h->nhid + 2);
  return launch_mlp(nerf_kernel<1, 8, true, false>, a, 512, lds, a.nbatch, st);
}

extern "C" int pnrf_mlp_fwd(const pnrf_mlp_t* h, const float* x, const float* x_views, float* y, int64_t m, int head_act, void* stream) {
  PNRF_REQUIRE(h, PNRF_E_ARG, "pnrf_mlp_fwd: null handle");
  PNRF_REQUIRE(m >= 0 && (m == 0 || (x && y)), PNRF_E_ARG, "pnrf_mlp_fwd: null pointer / negative m");
  PNRF_REQUIRE((h->net != PNRF_NET_NERF && h->net != PNRF_NET_NERFCLS) || m == 0 || x_views, PNRF_E_ARG, "pnrf_mlp_fwd: the nerf nets need x_views [m,27]");
  if (m == 0) return 0;
  const size_t lds = RING_BYTES + (size_t)h->nbias * 4;
  if (h->net == PNRF_NET_SAMPLER) {
    SamplerArgs a = {};
    PNRF_REQUIRE(h->npts == S_NPTS && h->d_blob, PNRF_E_SHAPE, "pnrf_mlp_fwd: the module-level sampler forward (unfolded first layer) is built for N_point_ray_enc = %d, "
                 "this sampler has %d; the fused operators (pnrf_sampler_fwd / pnrf_render_rays_fwd) take any", S_NPTS, h->npts);
    a.blob = h->d_blob; a.bias = h->d_bias; a.nslots = h->nslots; a.nbias = h->nbias; a.nhid = h->nhid;
    a.n = m; a.nbatch = (int)((m + 127) / 128);
    a.x = x; a.in0 = h->d_in0; a.y = y; a.outmap = h->d_out; a.head_act = head_act;
    return launch_mlp(sampler_kernel<0>, a, 512, lds, a.nbatch, (hipStream_t)stream);
  }
  if (h->net == PNRF_NET_REFINE) {
    RefineArgs a = {};
    a.blob = h->d_blob; a.bias = h->d_bias; a.nslots = h->nslots; a.nbias = h->nbias;
    a.n = m; a.nbatch = (int)((m + 127) / 128);
    a.x = x; a.y = y; a.outmap = h->d_out; a.head_act = head_act; a.nhid = h->nhid; a.nb = h->nb;
    // operand type as the fused stage's (pnrf_refine_fwd / pnrf_refine_project_fwd): fp16 by default, bf16 for PNRF_VARIANT_BF16, so that a
    // module-level parity check exercises the arithmetic the render path runs
    const bool bf16 = h->variant == PNRF_VARIANT_BF16;
    if (!bf16) { a.blob = h->d_blob_f16; a.nslots = h->nslots_f16; }
    auto go = [&](auto nvc) {
      constexpr int NV = decltype(nvc)::value;
      return bf16 ? launch_mlp(refine_kernel<1, 4, 0, 0, PrecBf16, NV>, a, 256, narrow_lds(lds), a.nbatch, (hipStream_t)stream)
                  : launch_mlp(refine_kernel<1, 4, 0, 0, PrecF16, NV>, a, 256, narrow_lds(lds), a.nbatch, (hipStream_t)stream);
    };
    switch (refine_nv(h->nb)) {
      case 1: return go(std::integral_constant<int, 1>{});
      case 2: return go(std::integral_constant<int, 2>{});
      case 3: return go(std::integral_constant<int, 3>{});
      default: return go(std::integral_constant<int, 4>{});
    }
  }
  NerfArgs a = {};
  a.blob = h->d_blob; a.bias = h->d_bias; a.nslots = h->nslots; a.nbias = h->nbias;
  a.n = m; a.nbatch = (int)((m + 127) / 128);
  a.x = x; a.xv = x_views; a.in0 = h->d_in0; a.inx = h->d_inx; a.y = y; a.outmap = h->d_out; a.S = 8; a.nhid = h->nhid;
  // (4-wave workgroups: LDS-exclusive like the narrow shapes of the fused stages — one fused-MLP workgroup per CU)
  if (h->net == PNRF_NET_NERFCLS) return launch_mlp(nerf_kernel<1, 4, false, true>, a, 256, narrow_lds(lds), a.nbatch, (hipStream_t)stream);
  PNRF_REQUIRE(h->nhid == N_NHID, PNRF_E_SHAPE, "pnrf_mlp_fwd: the module-level DoNeRFTRT forward is built for netdepth %d; this net has netdepth %d — "
               "the fused operators (pnrf_nerf_fwd / pnrf_render_rays_fwd) take any", N_NHID + 2, h->nhid + 2);
  return launch_mlp(nerf_kernel<1, 4, false, false>, a, 256, narrow_lds(lds), a.nbatch, (hipStream_t)stream);
}

