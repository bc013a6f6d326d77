// Does MODE.FP16_OVFL (hwreg MODE bit 23) make v_cvt_pk_f16_f32 saturate at 65504 instead of producing +inf on gfx950?
//   hipcc --offload-arch=gfx950 tools/fp16_ovfl_probe.hip -o /tmp/fp16_ovfl_probe && /tmp/fp16_ovfl_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* in, unsigned* out, int ovfl) {
  if (ovfl) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);       // hwreg(HW_REG_MODE, 23, 1) = 1
  const float a = in[2 * threadIdx.x], b = in[2 * threadIdx.x + 1];
  int pk;
  asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(a), "v"(b));
  out[threadIdx.x] = (unsigned)pk;
}
int main() {
  float h[8] = {1.0f, 65504.f, 65520.f, 1e6f, -1e6f, 3e38f, 70000.f, -70000.f};
  float* d; unsigned* o; unsigned r[4];
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, 16);
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  for (int ovfl = 0; ovfl < 2; ++ovfl) {
    hipLaunchKernelGGL(k, dim3(1), dim3(4), 0, 0, d, o, ovfl);
    hipMemcpy(r, o, 16, hipMemcpyDeviceToHost);
    printf("FP16_OVFL=%d:", ovfl);
    for (int i = 0; i < 4; ++i) printf(" %04x %04x", r[i] & 0xffff, r[i] >> 16);
    printf("\n");
  }
  return 0;
}
