// What the bf16 MFMA pipes of THIS chip sustain at its power limit (diagnostic; bench.py reports it next to the spec peak as roofline.sustained):
// a pure loop of independent v_mfma_f32_16x16x32_bf16 — no LDS, no VALU, no barrier, 2 waves per SIMD on every CU, 16 accumulators per wave —
// with random operand bits (what a real network feeds the pipes; a power-limited chip clocks by data toggling) and with constant operands.
// Prints one JSON line.   hipcc --offload-arch=gfx950 -O3 tools/mfma_ceiling.hip -o pronerf_amd/lib/mfma_ceiling   (python -m pronerf_amd.build does it)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__device__ __forceinline__ bf16x8 frag(unsigned seed, bool random) {
  u32x4 u;
  for (int i = 0; i < 4; ++i) {
    const unsigned h = random ? hash(seed * 4 + i) : 0x12345678u;
    // two bf16 values in [-2, 2): sign | exponent 126..127 | 7 mantissa bits
    const unsigned lo = (h & 0x8000u) | ((126u + ((h >> 7) & 1u)) << 7) | (h & 0x7fu);
    const unsigned hi = ((h >> 16) & 0x8000u) | ((126u + ((h >> 23) & 1u)) << 7) | ((h >> 16) & 0x7fu);
    u[i] = lo | (hi << 16);
  }
  return __builtin_bit_cast(bf16x8, u);
}
template <bool RANDOM>
__global__ __launch_bounds__(512, 1) void k(float* out, int iters) {
  bf16x8 a[4], b[4];
  const unsigned id = blockIdx.x * 512 + threadIdx.x;
  for (int i = 0; i < 4; ++i) { a[i] = frag(id * 8 + i, RANDOM); b[i] = frag(id * 8 + 4 + i, RANDOM); }
  f32x4 acc[16];
  for (int j = 0; j < 16; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j & 3], b[j >> 2], acc[j], 0, 0, 0);
    if ((it & 255) == 255)
      for (int j = 0; j < 16; ++j) acc[j] *= 1e-6f;                     // keep the sums bounded
  }
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int j = 0; j < 16; ++j) s += acc[j];
  out[blockIdx.x * 512 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
int main() {
  int cus = 256;
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, 0) == hipSuccess) cus = p.multiProcessorCount;
  float* out;
  if (hipMalloc(&out, (size_t)cus * 512 * 4) != hipSuccess) return 1;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 60000;                                               // ~20 ms per launch
  double tf[2];
  for (int r = 0; r < 2; ++r) {
    auto launch = [&] { if (r == 0) hipLaunchKernelGGL((k<true>), dim3(cus), dim3(512), 0, 0, out, iters);
                        else hipLaunchKernelGGL((k<false>), dim3(cus), dim3(512), 0, 0, out, iters); };
    launch(); launch(); (void)hipDeviceSynchronize();                   // let the clocks settle under this load
    (void)hipEventRecord(e0); launch(); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    tf[r] = 2.0 * 16.0 * 16384.0 * iters * 8.0 * cus / (ms * 1e-3) / 1e12;
  }
  printf("{\"mfma\": \"v_mfma_f32_16x16x32_bf16\", \"waves_per_simd\": 2, \"cus\": %d, \"random_operands_tflops\": %.1f, \"constant_operands_tflops\": %.1f}\n", cus, tf[0], tf[1]);
  return 0;
}
