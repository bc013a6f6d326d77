// Diagnostic: is the scalar offset of a raw buffer load part of the range check on this GPU?  (LLVM documents soffset as "excluded from bounds checking".)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* p, float* out) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 1024, 0x00020000);   // 1 KiB window
  out[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, 0, 0, 0));        // in range
  out[1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, 2048, 0, 0));     // voffset past the window
  out[2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, 0, 2048, 0));     // soffset past the window
  out[3] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, 512, 768, 0));    // sum past the window
}
int main() {
  float *p, *o; hipMalloc(&p, 1 << 20); hipMalloc(&o, 64);
  float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = 100.f + i;
  hipMemcpy(p, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, p, o);
  float r[4]; hipMemcpy(r, o, 16, hipMemcpyDeviceToHost);
  printf("in range %.0f | voffset past %.0f | soffset past %.0f | voffset + soffset past %.0f   (0 = range-checked; data would be 100 + index)\n", r[0], r[1], r[2], r[3]);
  return 0;
}
