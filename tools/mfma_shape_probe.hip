// MFMA shape vs sustained clock under load (diagnostic only): the same FLOPs per tile as 16 k-steps of the MLP kernels, random
// operand bits cycling through 8 register fragments (a power-limited chip clocks by data toggling: constant operands hide it),
// 64 VALU + one s_barrier per tile, 2 waves per SIMD.  32x32x16 (one 32-column block) against 16x16x32 (two 16-column blocks).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__device__ __forceinline__ bf16x8 rnd_frag(unsigned seed) {
  u32x4 u;
  for (int i = 0; i < 4; ++i) {
    unsigned h = hash(seed * 4 + i);
    // two bf16 values in [-2, 2) with random mantissas: sign | exponent 126..128 | 7 random mantissa bits
    unsigned lo = ((h & 0x8000u)) | ((126u + ((h >> 7) & 1u)) << 7) | (h & 0x7fu);
    unsigned hi = (((h >> 16) & 0x8000u)) | ((126u + ((h >> 23) & 1u)) << 7) | ((h >> 16) & 0x7fu);
    u[i] = lo | (hi << 16);
  }
  return __builtin_bit_cast(bf16x8, u);
}
template <int SHAPE, int NVALU, int RANDOM>
__global__ __launch_bounds__(512, 1) void k(float* out, int iters) {
  bf16x8 a[8], b[8];
  const unsigned id = blockIdx.x * 512 + threadIdx.x;
  for (int i = 0; i < 8; ++i) { a[i] = rnd_frag(RANDOM ? id * 16 + i : 7); b[i] = rnd_frag(RANDOM ? id * 16 + 8 + i : 9); }
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 0.5f + j;
  f32x16 acc32 = {};
  f32x4 acc16[2] = {};
  constexpr int PER = NVALU / 16;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (SHAPE == 32) {
        acc32 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u & 7], b[(u * 3) & 7], acc32, 0, 0, 0);
      } else {
        acc16[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u & 7], b[(u * 3) & 7], acc16[0], 0, 0, 0);
        acc16[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u & 7], b[(u * 3 + 1) & 7], acc16[1], 0, 0, 0);
      }
#pragma unroll
      for (int w = 0; w < PER; ++w) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[w & 7]) : "v"(v[(w + 1) & 7]));
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_barrier();
    // keep the accumulators bounded without touching the MFMA stream much
    if ((it & 63) == 63) { for (int j = 0; j < 16; ++j) acc32[j] *= 1e-3f; for (int j = 0; j < 4; ++j) { acc16[0][j] *= 1e-3f; acc16[1][j] *= 1e-3f; } }
  }
  float s = 0;
  for (int j = 0; j < 16; ++j) s += acc32[j];
  for (int j = 0; j < 4; ++j) s += acc16[0][j] + acc16[1][j];
  for (int j = 0; j < 8; ++j) s += v[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  float* out; (void)hipMalloc(&out, 256 * 512 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 40000;
  auto run = [&](const char* name, auto launch) {
    launch(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = 16.0 * 32768.0 * iters * 8 * 256;
    printf("%-44s %8.2f ms  %7.1f TFLOP/s\n", name, ms, flop / (ms * 1e-3) / 1e12);
  };
  for (int rep = 0; rep < 2; ++rep) {
    run("32x32x16 random operands, 64 VALU + barrier", [&] { hipLaunchKernelGGL((k<32, 64, 1>), dim3(256), dim3(512), 0, 0, out, iters); });
    run("16x16x32 random operands, 64 VALU + barrier", [&] { hipLaunchKernelGGL((k<16, 64, 1>), dim3(256), dim3(512), 0, 0, out, iters); });
    run("32x32x16 constant operands", [&] { hipLaunchKernelGGL((k<32, 64, 0>), dim3(256), dim3(512), 0, 0, out, iters); });
    run("16x16x32 constant operands", [&] { hipLaunchKernelGGL((k<16, 64, 0>), dim3(256), dim3(512), 0, 0, out, iters); });
  }
  return 0;
}
