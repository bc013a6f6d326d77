// Diagnostic: does a wave's VALU work hide under its OWN MFMA?  One or two waves per SIMD; per loop step one v_mfma_f32_32x32x16_f16 on a dependent accumulator
// (operands in registers, no LDS) followed by N independent v_fma_f32 (or N v_exp_f32) on registers the MFMA does not touch; cycles per step against N.
// If the VALU work hid under the 32-cycle MFMA the curve would stay flat until N x 5 cycles ~ 27; if it serialises it rises from the first instruction.
// (Round 5: behind the ELU stages' MFMA-pipe occupancy of ~0.5 — tools/elu_chain_probe.hip measures 35 -> 65 cycles per MFMA with 3.5 VALU, one wave per SIMD.)
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_overlap_probe.hip -o pronerf_amd/lib/mfma_valu_overlap_probe && pronerf_amd/lib/mfma_valu_overlap_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

// KIND 0: N x v_fma_f32; 1: N x v_exp_f32; SHAPE 0: 32x32x16 f16, one dependent accumulator; 1: 16x16x32 bf16, four accumulators in turn (two MFMAs per step = the same 32 pipe cycles)
template <int N, int KIND, int SHAPE, int WAVES>
__global__ __launch_bounds__(WAVES * 64, WAVES == 8 ? 2 : 1) void k(float* out, int iters, unsigned long long* cyc) {
  f16x8 a, b;
  bf16x8 ab, bb;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * (float)((threadIdx.x + i) % 13)); b[i] = (_Float16)(0.02f * (float)((threadIdx.x * 3 + i) % 11)); ab[i] = (__bf16)(float)a[i]; bb[i] = (__bf16)(float)b[i]; }
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  f32x4 c4[4];
  for (int i = 0; i < 4; ++i) c4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = 0.001f * (float)(threadIdx.x + i) - 0.5f;
  const float c0 = 0.999f, c1 = 0.0005f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      if (SHAPE == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
      else {
        c4[(2 * s) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, c4[(2 * s) & 3], 0, 0, 0);
        c4[(2 * s + 1) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, c4[(2 * s + 1) & 3], 0, 0, 0);
      }
#pragma unroll
      for (int n = 0; n < N; ++n) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[n & 7]) : "v"(c0), "v"(c1));
        else asm volatile("v_exp_f32 %0, %0" : "+v"(v[n & 7]));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += v[i];
  for (int i = 0; i < 16; ++i) s += acc[i];
  for (int i = 0; i < 4; ++i) s += c4[i][0] + c4[i][1] + c4[i][2] + c4[i][3];
  out[blockIdx.x * WAVES * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  float* out; (void)hipMalloc(&out, 256 * 512 * 4);
  unsigned long long* cyc; (void)hipMalloc(&cyc, 256 * 8);
  const int iters = 4000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto run = [&](const char* name, int n, auto launch, int waves) {
    launch(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double c = 0; for (int i = 0; i < 256; ++i) c += (double)h[i];
    c /= 256.0;
    const double per_step = c / (iters * 16.0);
    // wall clock: TFLOP/s of the MFMAs and the frequency of the s_memtime counter (ticks of the longest workgroup / kernel time)
    const double tflops = 256.0 * waves * iters * 16.0 * 32768.0 / (ms * 1e-3) / 1e12;
    printf("%-58s N = %d: %6.1f ticks per step per wave, %5.1f per step on the SIMD; %7.1f TFLOP/s; s_memtime ~ %.2f GHz\n", name, n, per_step, per_step / (waves / 4), tflops,
           c / (ms * 1e-3) / 1e9);
  };
#define RUN(N, KIND, SHAPE, W, txt) run(txt, N, [&] { hipLaunchKernelGGL((k<N, KIND, SHAPE, W>), dim3(256), dim3(W * 64), 0, 0, out, iters, cyc); }, W)
#define SWEEP(KIND, SHAPE, W, txt) RUN(0, KIND, SHAPE, W, txt); RUN(1, KIND, SHAPE, W, txt); RUN(2, KIND, SHAPE, W, txt); RUN(3, KIND, SHAPE, W, txt); RUN(4, KIND, SHAPE, W, txt); RUN(6, KIND, SHAPE, W, txt); RUN(8, KIND, SHAPE, W, txt)
  SWEEP(0, 0, 4, "1 wave/SIMD, 32x32x16 dependent chain, v_fma_f32");
  SWEEP(1, 0, 4, "1 wave/SIMD, 32x32x16 dependent chain, v_exp_f32");
  SWEEP(0, 1, 4, "1 wave/SIMD, 2 x 16x16x32, four accumulators, v_fma_f32");
  SWEEP(0, 0, 8, "2 waves/SIMD, 32x32x16 dependent chain, v_fma_f32");
  SWEEP(0, 1, 8, "2 waves/SIMD, 2 x 16x16x32, four accumulators, v_fma_f32");
  return 0;
}
