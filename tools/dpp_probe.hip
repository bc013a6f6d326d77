// What the DPP controls used by the compositing epilogue deliver on gfx950: prints the source lane every lane reads.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CTRL>
__device__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(-1, v, CTRL, 0xF, 0xF, true); }
__global__ void probe(int* out) {
  const int l = threadIdx.x;
  out[0 * 64 + l] = dpp_i<0xB1>(l);
  out[1 * 64 + l] = dpp_i<0x4E>(l);
  out[2 * 64 + l] = dpp_i<0x141>(l);
  out[3 * 64 + l] = dpp_i<0x101>(l);
  out[4 * 64 + l] = dpp_i<0x111>(l);
  out[5 * 64 + l] = dpp_i<0x112>(l);
  out[6 * 64 + l] = dpp_i<0x114>(l);
}
int main() {
  int* d; int h[7 * 64];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[7] = {"quad_perm[1,0,3,2]", "quad_perm[2,3,0,1]", "row_half_mirror", "row_shl:1", "row_shr:1", "row_shr:2", "row_shr:4"};
  for (int k = 0; k < 7; ++k) { printf("%-20s", names[k]); for (int l = 0; l < 20; ++l) printf(" %2d", h[k * 64 + l]); printf("\n"); }
  return 0;
}
