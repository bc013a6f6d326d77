// Diagnostic (round 4): what the LDS-fed hidden-layer loop of the NeRF stage (8 waves, two 16-column tiles per wave, weight fragments through an
// 8-deep register queue, a barrier per 16 fragments) loses against the same MFMAs fed from registers — split into (a) the reads' issue slots
// (reads issued, results unused), (b) waiting for the reads (results used), (c) the slot barrier, (d) accumulator dependence (T = 1: every second
// MFMA extends the same accumulator; T = 2: tile pairs as layer_b16, four independent accumulators).
//   hipcc --offload-arch=gfx950 -O3 tools/lds_mfma_probe2.hip -o /tmp/lds_mfma_probe2
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
// MODE 3: no reads, but the A operand rotates through Q resident random registers (the operand bits change from MFMA to MFMA as they do with
// fragments from LDS: the chip is power-limited, and a constant operand draws less).
// MODE 0: fragments from LDS, used.  1: reads issued, MFMAs on a register fragment (xor-ed with the read result of 8 fragments ago, so the read is
// live but old).  2: no reads.   BAR: barrier per 16 fragments.   T: tiles per k-step (1 or 2), U column tiles, Q queue depth.
template <int U, int T, int WAVES, int MODE, bool BAR, int Q = 8, int OCC = 1>
__global__ __launch_bounds__(WAVES * 64, OCC) void k(float* out, int iters, unsigned long long* clk) {
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  __shared__ __attribute__((aligned(16))) u32x4 lds[4096];                 // 64 KiB of "weights": 64 fragments of 1 KiB
  for (int i = threadIdx.x; i < 4096; i += WAVES * 64) lds[i] = rnd(i * 7 + 1);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f32x4 acc[T][U];
  bf16x8 b[U][2];
  for (int u = 0; u < U; ++u) {
    for (int t = 0; t < T; ++t) acc[t][u] = f32x4{0, 0, 0, 0};
    b[u][0] = __builtin_bit_cast(bf16x8, rnd(blockIdx.x * 4096 + threadIdx.x * 8 + u)); b[u][1] = __builtin_bit_cast(bf16x8, rnd(blockIdx.x * 4096 + threadIdx.x * 8 + u + 4));
  }
  const bf16x8 areg = __builtin_bit_cast(bf16x8, rnd(threadIdx.x + 99));
  float v[4] = {1.f, 2.f, 3.f, 4.f};
  bf16x8 q[Q];
  u32x4 sink = {0, 0, 0, 0};
  for (int f = 0; f < Q; ++f) q[f] = __builtin_bit_cast(bf16x8, lds[f * 64 + lane]);
  if (MODE == 3) for (int f = 0; f < Q; ++f) q[f] = __builtin_bit_cast(bf16x8, rnd(threadIdx.x * 16 + f + 777));
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int f = 0; f < 16; ++f) {
      bf16x8 a = areg;
      if (MODE == 0 || MODE == 3) a = q[f % Q];
      if (MODE == 1 && f == 15) sink ^= __builtin_bit_cast(u32x4, q[f % Q]);
      if (MODE < 2) q[f % Q] = __builtin_bit_cast(bf16x8, lds[(((it * 16 + f + Q) & 63) * 64) + lane]);
#pragma unroll
      for (int u = 0; u < U; ++u) acc[f % T][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[u][(f / T) & 1], acc[f % T][u], 0, 0, 0);
      if ((f & 1) == 0 || U == 4) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[f & 3]) : "v"(v[(f + 1) & 3]));
      if (U == 4 && (f & 3) == 0) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[(f + 2) & 3]) : "v"(v[(f + 3) & 3]));
      __builtin_amdgcn_sched_barrier(0);
    }
    if (BAR) __builtin_amdgcn_s_barrier();
    if ((it & 63) == 63) for (int t = 0; t < T; ++t) for (int u = 0; u < U; ++u) acc[t][u] *= 1e-3f;
  }
  f32x4 s = {0, 0, 0, 0};
  for (int t = 0; t < T; ++t) for (int u = 0; u < U; ++u) s += acc[t][u];
  out[blockIdx.x * WAVES * 64 + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + v[0] + v[1] + v[2] + v[3] + __builtin_bit_cast(float, sink[0] & 1u);
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
}
int main() {
  float* out; (void)hipMalloc(&out, 512 * 512 * 4);
  unsigned long long* clk; (void)hipHostMalloc(&clk, 16);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 20000;
  auto run = [&](const char* name, auto launch, int U, int waves, int grid) {
    launch(); (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 2; ++r) {
      (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    const double tf = (double)grid * waves * iters * 16.0 * U * 16384.0 / (best * 1e-3) / 1e12, ghz = (double)clk[0] / (double)clk[1] * 0.1;
    printf("%-72s %8.2f ms  %7.1f TFLOP/s  clock %.3f GHz  MFMA pipe occupancy %.3f\n", name, best, tf, ghz, tf * 1e12 / (4096.0 * 256 * ghz * 1e9));
    fflush(stdout);
  };
#define RUN(U, T, W, M, B, Q, G, OCC) run("U=" #U " T=" #T " waves=" #W " mode=" #M " barrier=" #B " queue=" #Q " grid=" #G, [&] { hipLaunchKernelGGL((k<U, T, W, M, B, Q, OCC>), dim3(G), dim3(W * 64), 0, 0, out, iters, clk); }, U, W, G)
  RUN(2, 1, 8, 0, true, 8, 256, 1);
  RUN(2, 2, 8, 0, true, 8, 256, 1);
  RUN(2, 2, 8, 0, false, 8, 256, 1);
  RUN(2, 2, 8, 1, true, 8, 256, 1);
  RUN(2, 2, 8, 1, false, 8, 256, 1);
  RUN(2, 2, 8, 2, true, 8, 256, 1);
  RUN(2, 2, 8, 2, false, 8, 256, 1);
  RUN(2, 2, 8, 3, true, 8, 256, 1);
  RUN(2, 2, 8, 3, false, 8, 256, 1);
  RUN(2, 2, 8, 0, true, 4, 256, 1);
  RUN(2, 2, 8, 0, true, 12, 256, 1);
  RUN(2, 2, 4, 0, true, 8, 256, 1);      // one wave per SIMD, 2 tiles
  RUN(4, 2, 4, 0, true, 8, 256, 1);      // one wave per SIMD, 4 tiles
  RUN(4, 2, 4, 0, false, 8, 256, 1);
  RUN(4, 2, 4, 2, false, 8, 256, 1);
  RUN(2, 2, 4, 0, true, 8, 512, 2);      // two 4-wave workgroups per CU (64 KiB LDS each), barriers independent
  RUN(2, 2, 4, 0, false, 8, 512, 2);
  RUN(3, 2, 8, 0, true, 8, 256, 1);      // 48 columns per wave
  return 0;
}
