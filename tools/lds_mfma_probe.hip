// Diagnostic: does feeding the MFMA's A operand from LDS (one ds_read_b128 per fragment, as the fused inference kernels do) limit the MFMA rate?
// Per wave and iteration: F fragment reads of 16 bytes per lane, each used by U MFMAs (U = activation column tiles per wave).  8 waves per
// workgroup (2 per SIMD), one workgroup per CU.   hipcc --offload-arch=gfx950 -O3 tools/lds_mfma_probe.hip -o /tmp/lds_mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
template <int U, bool FROM_LDS, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  for (int i = threadIdx.x; i < 65536 / 4; i += WAVES * 64) ((unsigned*)lds)[i] = 0x3f803f80u;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f32x4 acc[U];
  bf16x8 b[U];
  for (int u = 0; u < U; ++u) { acc[u] = f32x4{0, 0, 0, 0}; for (int e = 0; e < 8; ++e) b[u][e] = (__bf16)(1.0f + u); }
  bf16x8 areg;
  for (int e = 0; e < 8; ++e) areg[e] = (__bf16)0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int f = 0; f < 16; ++f) {
      bf16x8 a;
      if (FROM_LDS) a = *(const bf16x8*)(lds + ((it * 16 + f) & 63) * 1024 + lane * 16);
      else a = areg;
#pragma unroll
      for (int u = 0; u < U; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[u], acc[u], 0, 0, 0);
    }
  }
  f32x4 s = {0, 0, 0, 0};
  for (int u = 0; u < U; ++u) s += acc[u];
  out[blockIdx.x * WAVES * 64 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
int main() {
  float* out; (void)hipMalloc(&out, 256 * 512 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 4000;
  auto run = [&](const char* name, auto launch, int U, int waves) {
    launch(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfma = 256.0 * waves * iters * 16.0 * U;
    printf("%-52s %8.3f ms  %7.1f TFLOP/s   LDS fragment bytes per CU and clock at 2.1 GHz: %5.1f\n", name, ms, mfma * 16384.0 / (ms * 1e-3) / 1e12,
           256.0 * waves * iters * 16.0 * 1024.0 / 256.0 / (ms * 1e-3 * 2.1e9));
  };
#define RUN(U, L, W) run("U=" #U " columns tiles per fragment, from LDS=" #L ", waves=" #W, [&] { hipLaunchKernelGGL((k<U, L, W>), dim3(256), dim3(W * 64), 0, 0, out, iters); }, U, W)
  RUN(2, false, 8); RUN(2, true, 8); RUN(4, false, 8); RUN(4, true, 8); RUN(4, false, 4); RUN(4, true, 4); RUN(1, true, 8); RUN(8, true, 4); RUN(8, false, 4);
  return 0;
}
