// MFMA pipe probe (diagnostic only): what a SIMD sustains for the tile shape the MLP kernels use —
// NACC dependent 32x32x16 bf16 chains of 16 k-steps, followed by NVALU epilogue VALU ops, optional
// s_barrier per tile — at 1 or 2 waves per SIMD.  Prints the fraction of the pipe's issue rate.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
template <int NACC, int NVALU, int BAR>
__global__ __launch_bounds__(512, 1) void k(float* out, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 1e-3f + j); b[j] = (__bf16)(blockIdx.x * 1e-3f + 1.f); }
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 0.5f + j;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < NVALU; ++u) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[u & 7]) : "v"(v[(u + 1) & 7]));
    __builtin_amdgcn_sched_barrier(0);
    if (BAR) __builtin_amdgcn_s_barrier();
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  for (int j = 0; j < 8; ++j) s += v[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// interleaved: the VALU ops sit between the MFMAs (what a deferred epilogue does)
template <int NACC, int NVALU, int BAR>
__global__ __launch_bounds__(512, 1) void ki(float* out, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 1e-3f + j); b[j] = (__bf16)(blockIdx.x * 1e-3f + 1.f); }
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 0.5f + j;
  constexpr int PER = NVALU / 16;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int w = 0; w < PER; ++w) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[w & 7]) : "v"(v[(w + 1) & 7]));
      __builtin_amdgcn_sched_barrier(0);
    }
    if (BAR) __builtin_amdgcn_s_barrier();
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  for (int j = 0; j < 8; ++j) s += v[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// same work per tile with v_mfma_f32_16x16x32_bf16: 32 MFMAs of half the flops, two accumulator chains (two 16-column blocks)
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int NVALU, int BAR>
__global__ __launch_bounds__(512, 1) void k16(float* out, int iters) {
  f32x4 acc[2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  bf16x8 a, b0, b1;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 1e-3f + j); b0[j] = (__bf16)(blockIdx.x * 1e-3f + 1.f + 0.1f * j); b1[j] = (__bf16)(blockIdx.x * 2e-3f - 1.f + 0.2f * j); }
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 0.5f + j;
  constexpr int PER = NVALU / 16;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b1, acc[1], 0, 0, 0);
#pragma unroll
      for (int w = 0; w < PER; ++w) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[w & 7]) : "v"(v[(w + 1) & 7]));
      __builtin_amdgcn_sched_barrier(0);
    }
    if (BAR) __builtin_amdgcn_s_barrier();
  }
  float s = 0;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j];
  for (int j = 0; j < 8; ++j) s += v[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;
  auto run = [&](const char* name, auto launch, int nacc, int threads) {
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double mf = 16.0 * nacc * iters * (threads / 64);            // MFMAs per CU
    double cyc_per_mfma_simd = (ms * 1e-3) / (mf / 4);             // seconds per MFMA per SIMD
    double tf = mf * 256 * 32768.0 / (ms * 1e-3) / 1e12;
    printf("%-46s %8.3f ms  %7.1f TF  %6.2f ns/MFMA/SIMD\n", name, ms, tf, cyc_per_mfma_simd * 1e9);
  };
#define RUN(K, NACC, NV, BAR, THR) run(#K " acc=" #NACC " valu=" #NV " bar=" #BAR " thr=" #THR, [&] { hipLaunchKernelGGL((K<NACC, NV, BAR>), dim3(256), dim3(THR), 0, 0, out, iters); }, NACC, THR)
  RUN(k, 1, 0, 0, 256);
  RUN(k, 2, 0, 0, 256);
  RUN(k, 1, 0, 0, 512);
  RUN(k, 2, 0, 0, 512);
  RUN(k, 1, 32, 0, 512);
  RUN(k, 1, 64, 0, 512);
  RUN(k, 1, 32, 1, 512);
  RUN(k, 1, 64, 1, 512);
  RUN(ki, 1, 32, 0, 512);
  RUN(ki, 1, 64, 0, 512);
  RUN(ki, 1, 32, 1, 512);
  RUN(ki, 1, 64, 1, 512);
  RUN(ki, 1, 64, 0, 256);
  RUN(ki, 2, 64, 0, 256);
  RUN(ki, 1, 128, 0, 512);
  auto run16 = [&](const char* name, auto launch, int threads) {
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double mf = 32.0 * iters * (threads / 64);
    printf("%-46s %8.3f ms  %7.1f TF\n", name, ms, mf * 256 * 16384.0 / (ms * 1e-3) / 1e12);
  };
  run16("k16 16x16x32 valu=64 bar=1 thr=512", [&] { hipLaunchKernelGGL((k16<64, 1>), dim3(256), dim3(512), 0, 0, out, iters); }, 512);
  run16("k16 16x16x32 valu=64 bar=0 thr=512", [&] { hipLaunchKernelGGL((k16<64, 0>), dim3(256), dim3(512), 0, 0, out, iters); }, 512);
  run16("k16 16x16x32 valu=0  bar=0 thr=512", [&] { hipLaunchKernelGGL((k16<0, 0>), dim3(256), dim3(512), 0, 0, out, iters); }, 512);
  run16("k16 16x16x32 valu=64 bar=1 thr=256", [&] { hipLaunchKernelGGL((k16<64, 1>), dim3(256), dim3(256), 0, 0, out, iters); }, 256);
  // repeat the 32x32x16 reference points right after, same thermal state
  RUN(ki, 1, 64, 1, 512);
  RUN(ki, 1, 64, 0, 512);
  RUN(k, 1, 0, 0, 512);
  return 0;
}
