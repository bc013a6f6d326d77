// Probe: where does global_load_lds_dwordx3 put a lane's three dwords, and does the instruction offset move the LDS address too?
//   hipcc --offload-arch=gfx950 -O2 tools/ldsdma_probe.hip -o gpurun_out/ldsdma_probe && gpurun_out/ldsdma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((address_space(3))) void* lptr_t;
__global__ void probe(const float* src, float* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* l = (float*)smem;
  for (int i = threadIdx.x; i < 1024; i += 64) l[i] = -1.f;
  __syncthreads();
  const uint32_t base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lptr_t)smem);
  const uint32_t voff = threadIdx.x * 12u;
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 2\n\tglobal_load_lds_dwordx3 %1, %2\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(src), "s"(base) : "memory");
  const uint32_t base2 = base + 2048u;
  const uint32_t voff1 = threadIdx.x * 12u;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 2\n\tglobal_load_lds_dword %1, %2 offset:8\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff1), "s"(src), "s"(base2) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 64) out[i] = l[i];
}
int main() {
  std::vector<float> h(64 * 3);
  for (int l = 0; l < 64; ++l) for (int k = 0; k < 3; ++k) h[l * 3 + k] = l * 10 + k;
  float *d, *o;
  hipMalloc(&d, h.size() * 4 + 64); hipMalloc(&o, 4096);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 8192, 0, d, o);
  std::vector<float> r(1024);
  hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
  printf("x3 region, first 24 dwords:"); for (int i = 0; i < 24; ++i) printf(" %g", r[i]); printf("\n");
  printf("x3 region, dwords 60..75:"); for (int i = 60; i < 76; ++i) printf(" %g", r[i]); printf("\n");
  printf("x3 region, dwords 186..200:"); for (int i = 186; i < 200; ++i) printf(" %g", r[i]); printf("\n");
  printf("dword + offset:8 region (dst + 2048 B = dword 512), dwords 508..524:"); for (int i = 508; i < 524; ++i) printf(" %g", r[i]); printf("\n");
  return 0;
}
