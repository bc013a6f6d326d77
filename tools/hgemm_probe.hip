// Diagnostic: where hgemm_kernel's time goes — the product kernel against builds without its MFMAs (1), its epilogue's global traffic (2),
// its A fetches (4), its weight fetches (8).   hipcc --offload-arch=gfx950 -O3 -I pronerf_amd/csrc tools/hgemm_probe.hip -o /tmp/hgemm_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
enum { T_ACT_NONE = 0, T_ACT_RELU = 1, T_ACT_ELU = 2 };
#define PNRF_HG_PROBE 1
#include "pnrf_hgemm.h"
int main(int argc, char** argv) {
  const int64_t M = argc > 1 ? atoll(argv[1]) : 32768;
  const int N = 256, K = 256;
  float *A, *C, *H, *bias; _Float16* planes;
  hipMalloc(&A, M * K * 4); hipMalloc(&C, M * N * 4); hipMalloc(&H, M * N * 4); hipMalloc(&bias, N * 4); hipMalloc(&planes, 2 * N * K * 2);
  std::vector<float> h(M * K); for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.f - 0.5f;
  hipMemcpy(A, h.data(), M * K * 4, hipMemcpyHostToDevice); hipMemcpy(H, h.data(), M * N * 4, hipMemcpyHostToDevice);
  std::vector<_Float16> w(2 * N * K); for (size_t i = 0; i < w.size(); ++i) w[i] = (_Float16)(h[i] * 0.1f);
  hipMemcpy(planes, w.data(), w.size() * 2, hipMemcpyHostToDevice); hipMemset(bias, 0, N * 4);
  HGemmArgs a = {};
  a.A = A; a.lda = K; a.Bh = planes; a.Bl = planes + N * K; a.ldb = K; a.n_pad = N; a.C = C; a.ldc = N; a.M = M; a.N = N; a.K = K;
  a.bias = bias; a.act = T_ACT_RELU;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < 20; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    printf("%-40s %8.2f us  %7.1f TFLOP/s-equivalent  %6.2f TB/s (A + C)\n", name, ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12, (double)M * (N + K) * 4 / (ms * 1e-3) / 1e12);
  };
  const dim3 grid((unsigned)((M + 63) / 64 < 256 ? (M + 63) / 64 : 256));
#define RUN(P) run("forward, probe " #P, [&] { hipLaunchKernelGGL((hgemm_kernel<4, P>), grid, dim3(512), 0, 0, a); })
  RUN(0); RUN(2); RUN(14); RUN(8); RUN(4); RUN(1);
  HGemmArgs b = a; b.bwd = 1; b.H = H; b.ldh = N; b.bias = nullptr;
#define RUNB(P) run("backward (act'(H)), probe " #P, [&] { hipLaunchKernelGGL((hgemm_kernel<4, P>), grid, dim3(512), 0, 0, b); })
  RUNB(0); RUNB(2);
  if (M <= 8192) {                                             // the 4096-row nets: 16-row tiles, one per workgroup
    const dim3 g1((unsigned)((M + 15) / 16 < 256 ? (M + 15) / 16 : 256));
#define RUN1(P) run("16-row tiles, forward, probe " #P, [&] { hipLaunchKernelGGL((hgemm_kernel<1, P>), g1, dim3(512), 0, 0, a); })
    RUN1(0); RUN1(1); RUN1(2); RUN1(4); RUN1(8); RUN1(12); RUN1(14); RUN1(15);
  }
  return 0;
}
