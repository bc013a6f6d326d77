// Diagnostic: the hidden-layer steady state of the fused inference kernels as a bare loop — A fragments (weights) read from LDS through an
// 8-deep register queue, U activation column tiles per wave fed by every fragment, random operand bits, V VALU instructions per MFMA, one barrier
// per 16 fragments — for 8 waves x U = 2 (the kernels' shape: 32 columns per wave, two waves per SIMD) against 4 waves x U = 4 (64 columns per
// wave, one wave per SIMD: half the LDS reads per MFMA).   hipcc --offload-arch=gfx950 -O3 tools/lds_mfma_probe.hip -o /tmp/lds_mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__device__ __forceinline__ u32x4 rnd(unsigned seed) {
  u32x4 u;
  for (int i = 0; i < 4; ++i) {
    const unsigned h = hash(seed * 4 + i);
    const unsigned lo = (h & 0x8000u) | ((126u + ((h >> 7) & 1u)) << 7) | (h & 0x7fu);
    const unsigned hi = ((h >> 16) & 0x8000u) | ((126u + ((h >> 23) & 1u)) << 7) | ((h >> 16) & 0x7fu);
    u[i] = lo | (hi << 16);
  }
  return u;
}
template <int U, int WAVES, bool FROM_LDS>
__global__ __launch_bounds__(WAVES * 64, 1) void k(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) u32x4 lds[4096];                 // 64 KiB of "weights": 64 fragments of 1 KiB
  for (int i = threadIdx.x; i < 4096; i += WAVES * 64) lds[i] = rnd(i * 7 + 1);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f32x4 acc[U];
  bf16x8 b[U][2];
  for (int u = 0; u < U; ++u) { acc[u] = f32x4{0, 0, 0, 0}; b[u][0] = __builtin_bit_cast(bf16x8, rnd(blockIdx.x * 4096 + threadIdx.x * 8 + u)); b[u][1] = __builtin_bit_cast(bf16x8, rnd(blockIdx.x * 4096 + threadIdx.x * 8 + u + 4)); }
  const bf16x8 areg = __builtin_bit_cast(bf16x8, rnd(threadIdx.x + 99));
  float v[4] = {1.f, 2.f, 3.f, 4.f};
  bf16x8 q[8];
  for (int f = 0; f < 8; ++f) q[f] = __builtin_bit_cast(bf16x8, lds[f * 64 + lane]);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int f = 0; f < 16; ++f) {
      const bf16x8 a = FROM_LDS ? q[f & 7] : areg;
      if (FROM_LDS) q[f & 7] = __builtin_bit_cast(bf16x8, lds[(((it * 16 + f + 8) & 63) * 64) + lane]);
#pragma unroll
      for (int u = 0; u < U; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[u][f & 1], acc[u], 0, 0, 0);
      // ~0.6 VALU per MFMA, as the NeRF stage's hidden layers
      if ((f & 1) == 0 || U == 4) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[f & 3]) : "v"(v[(f + 1) & 3]));
      if (U == 4 && (f & 3) == 0) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[(f + 2) & 3]) : "v"(v[(f + 3) & 3]));
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_barrier();
    if ((it & 63) == 63) for (int u = 0; u < U; ++u) acc[u] *= 1e-3f;
  }
  f32x4 s = {0, 0, 0, 0};
  for (int u = 0; u < U; ++u) s += acc[u];
  out[blockIdx.x * WAVES * 64 + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + v[0] + v[1] + v[2] + v[3];
}
// the split-fp16 sampler's steady state: a (hi, lo) fragment pair from LDS per three f16 MFMAs (main += hi x_hi; cross += hi x_lo + lo x_hi),
// CT activation column tiles per wave (1 in the kernel: 16 columns per wave, 8 waves), ~0.63 VALU per MFMA, a barrier per 16 fragments
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
template <int CT, int WAVES, bool FROM_LDS>
__global__ __launch_bounds__(WAVES * 64, 1) void ks(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) u32x4 lds[4096];
  for (int i = threadIdx.x; i < 4096; i += WAVES * 64) lds[i] = rnd(i * 7 + 1) & u32x4{0x3fff3fffu, 0x3fff3fffu, 0x3fff3fffu, 0x3fff3fffu};   // |x| < 2 as fp16
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f32x4 am[CT], ax[CT];
  f16x8 xh[CT], xl[CT];
  for (int u = 0; u < CT; ++u) {
    am[u] = f32x4{0, 0, 0, 0}; ax[u] = am[u];
    xh[u] = __builtin_bit_cast(f16x8, rnd(threadIdx.x * 8 + u) & u32x4{0x3fff3fffu, 0x3fff3fffu, 0x3fff3fffu, 0x3fff3fffu});
    xl[u] = __builtin_bit_cast(f16x8, rnd(threadIdx.x * 8 + u + 4) & u32x4{0x3fff3fffu, 0x3fff3fffu, 0x3fff3fffu, 0x3fff3fffu});
  }
  const f16x8 areg = __builtin_bit_cast(f16x8, rnd(threadIdx.x + 99) & u32x4{0x3fff3fffu, 0x3fff3fffu, 0x3fff3fffu, 0x3fff3fffu});
  float v[4] = {1.f, 2.f, 3.f, 4.f};
  f16x8 q[8];
  for (int f = 0; f < 8; ++f) q[f] = __builtin_bit_cast(f16x8, lds[f * 64 + lane]);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int f = 0; f < 16; f += 2) {                              // fragment pair (hi, lo)
      const f16x8 wh = FROM_LDS ? q[f & 7] : areg, wl = FROM_LDS ? q[(f + 1) & 7] : areg;
      if (FROM_LDS) { q[f & 7] = __builtin_bit_cast(f16x8, lds[(((it * 16 + f + 8) & 63) * 64) + lane]); q[(f + 1) & 7] = __builtin_bit_cast(f16x8, lds[(((it * 16 + f + 9) & 63) * 64) + lane]); }
#pragma unroll
      for (int u = 0; u < CT; ++u) {
        am[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh[u], am[u], 0, 0, 0);
        ax[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl[u], ax[u], 0, 0, 0);
        ax[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh[u], ax[u], 0, 0, 0);
        asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[u & 3]) : "v"(v[(u + 1) & 3]));
        asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[(u + 2) & 3]) : "v"(v[(u + 3) & 3]));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_barrier();
    if ((it & 63) == 63) for (int u = 0; u < CT; ++u) { am[u] *= 1e-3f; ax[u] *= 1e-3f; }
  }
  f32x4 s = {0, 0, 0, 0};
  for (int u = 0; u < CT; ++u) s += am[u] + ax[u];
  out[blockIdx.x * WAVES * 64 + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + v[0] + v[1] + v[2] + v[3];
}
int main() {
  float* out; (void)hipMalloc(&out, 256 * 512 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 30000;
  auto run = [&](const char* name, auto launch, int U, int waves) {
    launch(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-64s %8.2f ms  %7.1f TFLOP/s\n", name, ms, 256.0 * waves * iters * 16.0 * U * 16384.0 / (ms * 1e-3) / 1e12);
  };
#define RUN(U, W, L) run(#W " waves x " #U " column tiles per fragment, fragments from LDS: " #L, [&] { hipLaunchKernelGGL((k<U, W, L>), dim3(256), dim3(W * 64), 0, 0, out, iters); }, U, W)
  for (int rep = 0; rep < 2; ++rep) { RUN(2, 8, true); RUN(4, 4, true); RUN(2, 8, false); RUN(4, 4, false); RUN(4, 8, true); }
  auto runs = [&](const char* name, auto launch, int CT, int waves) {
    launch(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-64s %8.2f ms  %7.1f TFLOP/s of f16 MFMA = %5.1f TFLOP/s of fp32-grade product\n", name, ms, 256.0 * waves * iters * 8.0 * 3 * CT * 16384.0 / (ms * 1e-3) / 1e12,
           256.0 * waves * iters * 8.0 * CT * 16384.0 / (ms * 1e-3) / 1e12);
  };
#define RUNS(CT, W, L) runs("split fp16: " #W " waves x " #CT " column tiles, fragment pairs from LDS: " #L, [&] { hipLaunchKernelGGL((ks<CT, W, L>), dim3(256), dim3(W * 64), 0, 0, out, iters); }, CT, W)
  RUNS(1, 8, true); RUNS(1, 8, false); RUNS(2, 4, true); RUNS(2, 8, true);
  return 0;
}
