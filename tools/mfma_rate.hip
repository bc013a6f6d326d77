// Raw MFMA issue-rate probe (diagnostic only): back-to-back MFMAs on register operands.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
template <int NACC>
__global__ __launch_bounds__(512, 2) void k_f32_16(float* out, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 1.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256, 1) void k_f32_32(float* out, int iters) {
  f32x16 acc = {};
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 1.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  auto run = [&](const char* name, auto launch, double flop_per_wave_iter, int waves_per_block) {
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double tf = flop_per_wave_iter * iters * waves_per_block * 256 / (ms * 1e-3) / 1e12;
    printf("%-34s %8.3f ms  %7.1f TF\n", name, ms, tf);
  };
  run("f32 16x16x4, 8 waves, 1 acc", [&] { hipLaunchKernelGGL(k_f32_16<1>, dim3(256), dim3(512), 0, 0, out, iters); }, 16.0 * 2048, 8);
  run("f32 16x16x4, 8 waves, 2 acc", [&] { hipLaunchKernelGGL(k_f32_16<2>, dim3(256), dim3(512), 0, 0, out, iters); }, 32.0 * 2048, 8);
  run("f32 16x16x4, 4 waves(256thr), 2 acc", [&] { hipLaunchKernelGGL(k_f32_16<2>, dim3(256), dim3(256), 0, 0, out, iters); }, 32.0 * 2048, 4);
  run("f32 32x32x2, 4 waves, 1 acc", [&] { hipLaunchKernelGGL(k_f32_32, dim3(256), dim3(256), 0, 0, out, iters); }, 16.0 * 4096, 4);
  return 0;
}
