// Diagnostic: a kernel that occupies `grid` CUs exclusively for `us` microseconds (100 KiB of LDS per workgroup: no fused-MLP workgroup fits beside it) —
// a stand-in for a collective's kernel running beside the renderer at N > 1.   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/cu_hog.hip -o pronerf_amd/lib/libcu_hog.so
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(256) void hog_kernel(unsigned long long ticks, unsigned* sink) {
  __shared__ unsigned lds[25600];                 // 100 KiB
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const unsigned long long t0 = wall_clock64();   // 100 MHz
  unsigned acc = 0;
  while (wall_clock64() - t0 < ticks) acc += lds[(acc + threadIdx.x) % 25600];
  if (acc == 0xdeadbeefu) sink[0] = acc;
}
extern "C" int cu_hog_launch(void* stream, int grid, int us, unsigned* sink) {
  hipLaunchKernelGGL(hog_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (unsigned long long)us * 100ull, sink);
  return (int)hipGetLastError();
}
