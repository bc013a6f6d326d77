// Diagnostic: what the ELU epilogue costs beside 32x32x16 MFMAs as a DEPENDENT chain per MFMA gap (exp -> fma -> med3 -> fma(|x|^2) -> cvt: the
// refine / sampler pass-1 kernels' one-activation pieces) against the same instructions SOFTWARE-PIPELINED across the gaps (each gap issues the
// exp of activation k+2, the fma of k+1, the med3 of k, the |x|^2 / cvt of k-1: no instruction waits for one issued in the same gap), at one and
// at two waves per SIMD, A fragments from an LDS queue, one barrier per 16 MFMAs (the engine's slot barrier).
//   hipcc --offload-arch=gfx950 -O3 tools/elu_chain_probe.hip -o /tmp/elu_chain_probe && /tmp/elu_chain_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__device__ __forceinline__ u32x4 rnd(unsigned seed) {
  u32x4 u;
  for (int i = 0; i < 4; ++i) u[i] = hash(seed * 4 + i) & 0x3bff3bffu;      // |x| < 1 as fp16
  return u;
}
constexpr float LOG2E = 1.4426950408889634f;
// MODE 5 (round 5): as 1 with two accumulators in turn.  MODE 0: MFMAs only; 1: dependent chain per gap; 2: pipelined over four gaps; 3: pipelined over two gaps (exp | rest);
// 4 (round 5): the pair finished in fp16 — even gap: v_exp_f32 clamp + v_fma_mixlo_f16 (L e - L, rounded once, straight into the low half of the packed dword);
//    odd gap: v_exp_f32 clamp, v_cvt_pk_f16_f32 of the two pre-activations, v_fma_mixhi_f16, v_pk_max_f16 — 7 instructions per pair instead of 8.5 (exp fma med3 x2 + cvt),
//    same bits (rounding is monotone: round(max(y, f)) = max(round y, round f)); with SSQ the |x|^2 of the pair is ONE v_dot2_f32_f16 on the packed dword instead of two v_fma_f32
template <int MODE, int WAVES, bool SSQ>
__global__ __launch_bounds__(WAVES * 64, WAVES == 8 ? 2 : 1) void k(float* out, int iters, unsigned long long* cyc) {
  __shared__ __attribute__((aligned(16))) u32x4 lds[4096];
  for (int i = threadIdx.x; i < 4096; i += WAVES * 64) lds[i] = rnd(i * 7 + 1);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  typedef __attribute__((ext_vector_type(4))) float f32x4;
  f32x4 c4[4];
  for (int i = 0; i < 4; ++i) c4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x16 acc, acc2, pend;
  for (int i = 0; i < 16; ++i) { acc[i] = 0.f; acc2[i] = 0.f; pend[i] = 0.01f * (float)((lane + i) % 7) - 0.03f; }
  f16x8 b[16];
  for (int i = 0; i < 16; ++i) b[i] = __builtin_bit_cast(f16x8, rnd(blockIdx.x * 4096 + threadIdx.x * 16 + i));
  f16x8 q[8];
  for (int f = 0; f < 8; ++f) q[f] = __builtin_bit_cast(f16x8, lds[f * 64 + lane]);
  float ssq = 0.f;
  int packed[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float e1 = 0.f, e2 = 0.f, m3 = 0.f, m3prev = 0.f;      // pipeline registers
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int f = 0; f < 16; ++f) {
      const f16x8 a = q[f & 7];
      q[f & 7] = __builtin_bit_cast(f16x8, lds[(((it * 16 + f + 8) & 63) * 64) + lane]);
      if (MODE == 6) {           // the same 32 pipe cycles as two v_mfma_f32_16x16x32_f16 on two of four accumulators (tile f & 1 of a pair x two 16-column blocks), one A fragment
        c4[(f & 1) * 2 + 0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b[f], c4[(f & 1) * 2 + 0], 0, 0, 0);
        c4[(f & 1) * 2 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b[(f + 8) & 15], c4[(f & 1) * 2 + 1], 0, 0, 0);
      } else if (MODE == 5 && (f & 1)) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[f], acc2, 0, 0, 0);
      else acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[f], acc, 0, 0, 0);
      if (MODE == 1 || MODE == 5 || MODE == 6) {                    // the whole activation of element f in this gap
        const float y = pend[f];
        const float t = fmaf(__builtin_amdgcn_exp2f(y), LOG2E, -LOG2E);
        const float v = __builtin_amdgcn_fmed3f(y, t, 0.f);
        pend[f] = v;
        if (SSQ) ssq = fmaf(v, v, ssq);
        if (f & 1) asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(packed[f >> 1]) : "v"(pend[f - 1]), "v"(v));
      } else if (MODE == 2) {             // gap f: exp of f+2, fma of f+1, med3 of f, |x|^2 and cvt of f-1 (wrapping inside the tile: same instruction mix)
        const float ex = __builtin_amdgcn_exp2f(pend[(f + 2) & 15]);
        const float fm = fmaf(e1, LOG2E, -LOG2E);
        const float md = __builtin_amdgcn_fmed3f(pend[f], e2, 0.f);
        if (SSQ) ssq = fmaf(m3, m3, ssq);
        if (f & 1) asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(packed[f >> 1]) : "v"(m3prev), "v"(m3));
        m3prev = m3; m3 = md; e2 = fm; e1 = ex;
        pend[f] = md;
      } else if (MODE == 4) {
        const float y = pend[f];
        if (!(f & 1)) {
          asm volatile("v_exp_f32 %0, %2 clamp\n\ts_nop 0\n\tv_fma_mixlo_f16 %1, %0, %3, %4 op_sel_hi:[0,0,0]" : "=&v"(e1), "+v"(packed[f >> 1]) : "v"(y), "v"(LOG2E), "v"(-LOG2E));
        } else {
          int ypk;
          asm volatile("v_exp_f32 %0, %3 clamp\n\tv_cvt_pk_f16_f32 %2, %4, %3\n\tv_fma_mixhi_f16 %1, %0, %5, %6 op_sel_hi:[0,0,0]\n\tv_pk_max_f16 %1, %1, %2"
                       : "=&v"(e1), "+v"(packed[f >> 1]), "=&v"(ypk) : "v"(y), "v"(pend[f - 1]), "v"(LOG2E), "v"(-LOG2E));
          if (SSQ) asm volatile("v_dot2_f32_f16 %0, %1, %1, %0" : "+v"(ssq) : "v"(packed[f >> 1]));
        }
      } else if (MODE == 3) {             // gap f: exp of f+1; fma, med3, |x|^2, cvt of f
        const float ex = __builtin_amdgcn_exp2f(pend[(f + 1) & 15]);
        const float t = fmaf(e1, LOG2E, -LOG2E);
        const float v = __builtin_amdgcn_fmed3f(pend[f], t, 0.f);
        if (SSQ) ssq = fmaf(v, v, ssq);
        if (f & 1) asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(packed[f >> 1]) : "v"(m3), "v"(v));
        m3 = v; e1 = ex;
        pend[f] = v;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 16; ++i) { pend[i] = (MODE == 6 ? c4[i >> 2][i & 3] : MODE == 5 ? acc[i] + acc2[i] : acc[i]) * 1e-3f; acc[i] = 0.f; acc2[i] = 0.f; }
    if (MODE == 6) for (int i = 0; i < 4; ++i) c4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) b[i] = __builtin_bit_cast(f16x8, __builtin_bit_cast(u32x4, b[i]) ^ u32x4{(unsigned)packed[i] & 0x03ff03ffu, 0, 0, 0});
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = ssq + e1 + e2 + m3 + m3prev;
  for (int i = 0; i < 16; ++i) s += pend[i];
  out[blockIdx.x * WAVES * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  float* out; (void)hipMalloc(&out, 256 * 512 * 4);
  unsigned long long* cyc; (void)hipMalloc(&cyc, 256 * 8);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 20000;
  auto run = [&](const char* name, auto launch, int waves) {
    launch(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double c = 0; for (int i = 0; i < 256; ++i) c += (double)h[i];
    c /= 256.0;
    const double per_mfma_wave = c / (iters * 16.0), per_mfma_pipe = per_mfma_wave / (waves / 4);
    printf("%-72s %8.2f ms  %6.1f cycles per MFMA per wave, %5.1f per MFMA on the SIMD's pipe (32 = busy), %7.1f TFLOP/s\n", name, ms, per_mfma_wave, per_mfma_pipe,
           256.0 * waves * iters * 16.0 * 32768.0 / (ms * 1e-3) / 1e12);
  };
#define RUN(M, W, S, txt) run(txt, [&] { hipLaunchKernelGGL((k<M, W, S>), dim3(256), dim3(W * 64), 0, 0, out, iters, cyc); }, W)
  for (int rep = 0; rep < 2; ++rep) {
    RUN(0, 4, false, "1 wave / SIMD, MFMAs only");
    RUN(1, 4, false, "1 wave / SIMD, ELU as a dependent chain per gap (refine)");
    RUN(2, 4, false, "1 wave / SIMD, ELU pipelined over four gaps");
    RUN(3, 4, false, "1 wave / SIMD, ELU pipelined over two gaps");
    RUN(4, 4, false, "1 wave / SIMD, pair finished in fp16: exp.clamp, fma_mix, cvt_pk, pk_max (round 5)");
    RUN(4, 4, true, "1 wave / SIMD, the same + |x|^2 as one v_dot2_f32_f16 per pair");
    RUN(5, 4, false, "1 wave / SIMD, ELU chain per gap, TWO accumulators alternating (no MFMA waits for its predecessor)");
    RUN(6, 4, false, "1 wave / SIMD, ELU chain per gap, 2 x 16x16x32 on four accumulators (the NeRF stage's engine shape)");
    RUN(6, 4, true, "1 wave / SIMD, the same + |x|^2");
    RUN(1, 4, true, "1 wave / SIMD, dependent chain + |x|^2 (sampler pass 1)");
    RUN(2, 4, true, "1 wave / SIMD, pipelined over four gaps + |x|^2");
    RUN(0, 8, false, "2 waves / SIMD, MFMAs only");
    RUN(1, 8, false, "2 waves / SIMD, ELU as a dependent chain per gap (refine)");
    RUN(2, 8, false, "2 waves / SIMD, ELU pipelined over four gaps");
    RUN(3, 8, false, "2 waves / SIMD, ELU pipelined over two gaps");
    RUN(4, 8, false, "2 waves / SIMD, pair finished in fp16: exp.clamp, fma_mix, cvt_pk, pk_max (round 5)");
    RUN(4, 8, true, "2 waves / SIMD, the same + |x|^2 as one v_dot2_f32_f16 per pair");
    RUN(5, 8, false, "2 waves / SIMD, ELU chain per gap, TWO accumulators alternating (no MFMA waits for its predecessor)");
    RUN(6, 8, false, "2 waves / SIMD, ELU chain per gap, 2 x 16x16x32 on four accumulators (the NeRF stage's engine shape)");
    RUN(6, 8, true, "2 waves / SIMD, the same + |x|^2");
    RUN(1, 8, true, "2 waves / SIMD, dependent chain + |x|^2 (sampler pass 1)");
    RUN(2, 8, true, "2 waves / SIMD, pipelined over four gaps + |x|^2");
  }
  return 0;
}
