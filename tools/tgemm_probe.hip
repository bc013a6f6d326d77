// fp32-MFMA issue probe (diagnostic only): the MFMA sequence of tgemm_kernel (pnrf_train.hip) without its loads — 16 accumulators,
// 4 fragment sets of 4 + 4 float4 operands in registers — at 1 and 2 workgroups (4 waves each) per CU, against variants: every MFMA
// with the same operand registers; 4 accumulators; operands taken from 8 distinct float4s only.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int VAR>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters) {
  f32x4 acc[4][4];
  typedef __attribute__((ext_vector_type(16))) float f32x16;
  f32x16 acc32[4][2];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc32[i][j][r] = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 fa[4][4], fb[4][4];
  for (int p = 0; p < 4; ++p) for (int i = 0; i < 4; ++i) {
    fa[p][i] = *(const f32x4*)(in + ((p * 4 + i) * 256 + threadIdx.x) * 4);
    fb[p][i] = *(const f32x4*)(in + ((16 + p * 4 + i) * 256 + threadIdx.x) * 4);
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (VAR == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[p][j][e], fa[p][i][e], acc[i][j], 0, 0, 0);
            if (VAR == 1) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[0][0][0], fa[0][0][1], acc[i][j], 0, 0, 0);
            if (VAR == 2) acc[i & 1][j & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[p][j][e], fa[p][i][e], acc[i & 1][j & 1], 0, 0, 0);
            if (VAR == 3) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[p][j][e], fa[p][i][(e + 1) & 3], acc[i][j], 0, 0, 0);
            if (VAR == 4 && (j & 1) == 0) {       // same FLOPs on v_mfma_f32_32x32x2_f32: 8 accumulators of 16 registers, half as many MFMAs of twice the work
              acc32[i][j >> 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[p][j][e], fa[p][i][e], acc32[i][j >> 1], 0, 0, 0);
              acc32[i][j >> 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[p][j + 1][e], fa[p][i][(e + 1) & 3], acc32[i][j >> 1], 0, 0, 0);
            }
          }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j];
  float s32 = 0.f;
  if (VAR == 4) for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s32 += acc32[i][j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + s32;
}
int main() {
  float *out, *in;
  hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&in, 32 * 256 * 16); hipMemset(in, 0, 32 * 256 * 16);
  float* hin = (float*)malloc(32 * 256 * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  auto run = [&](const char* name, auto launch, int wgs) {
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma = 256.0 * iters * 4 * wgs;                        // per launch: 256 MFMAs per iteration per wave
    printf("%-44s wgs %4d  %8.3f ms  %7.1f TFLOP/s\n", name, wgs, ms, mfma * 2048.0 / (ms * 1e-3) / 1e12);
  };
#define RUN(V, WGS) run("variant " #V, [&] { hipLaunchKernelGGL((k<V>), dim3(WGS), dim3(256), 0, 0, out, in, iters); }, WGS)
  printf("-- operands all zero\n");
  RUN(0, 256); RUN(0, 512); RUN(0, 768);
  RUN(1, 256); RUN(1, 512);
  RUN(2, 256); RUN(2, 512);
  RUN(3, 256); RUN(3, 512);
  printf("-- operands uniform random in [-1, 1)\n");
  srand(1);
  for (int i = 0; i < 32 * 256 * 4; ++i) hin[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
  hipMemcpy(in, hin, 32 * 256 * 16, hipMemcpyHostToDevice);
  RUN(0, 256); RUN(0, 512); RUN(0, 768); RUN(1, 512); RUN(2, 512);
  RUN(4, 256); RUN(4, 512);
  return 0;
}
