// Foreign kernels for the co-residency stress test (tests/test_coresidency_gpu.py, tools/coresidency_stress.py): stand-ins for whatever
// else may share a CU with a fused-MLP workgroup of the renderer when other streams are busy (a collective's kernel, an index_select, a
// neighbour upload's pack kernel).  Three footprints, chosen so that they CAN become resident beside the shipped launches:
//   kind 0  "small":  1 wave, <= 32 VGPRs, 1 KiB of LDS      — fits beside a WIDE workgroup (2 x 240 of a SIMD lane's 512 registers are taken)
//   kind 1  "gather": 4 waves, <= 128 VGPRs, 16 KiB of LDS   — fits beside a NARROW workgroup (240 registers, 84 KiB of LDS taken): 16-byte
//                     gathers at random addresses (the refine head's texel traffic through the same TA / L1)
//   kind 2  "dma":    4 waves, 64 KiB of LDS filled by LDS-DMA (global_load_lds_dwordx4 through M0, the weight stream's instruction) and
//                     read back with ds_read_b128 — the other fused kernels' footprint without their arithmetic
//   kind 3  "mfma":   4 waves, 240 VGPRs, no LDS: a loop of v_mfma_f32_16x16x32_bf16 — what a bf16 GEMM of another stream looks like to the SIMD; it fits beside a
//                     NARROW fused workgroup (240 + 240 of 512 registers) and is the aggressor of tools/pkf32_coexec_probe.hip's reproducer
// Every index is masked into the buffer, every loop is bounded by `iters`: nothing here can fault or spin.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/foreign_kernels.hip -o pronerf_amd/lib/libforeign_kernels.so   (pronerf_amd.build.build_foreign_kernels)
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// where[4 * blockIdx.x ..]: HW_ID, XCC_ID, LDS_ALLOC, GPR_ALLOC of the workgroup's first wave (placement evidence for the stress report)
__device__ __forceinline__ void record_placement(uint32_t* where) {
  if (where && threadIdx.x == 0) {
    uint32_t* w = where + 4 * blockIdx.x;
    w[0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    w[1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    w[2] = __builtin_amdgcn_s_getreg((31 << 11) | 6);
    w[3] = __builtin_amdgcn_s_getreg((31 << 11) | 5);
  }
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(32))) void foreign_small(const uint4* buf, uint32_t mask, int iters, uint32_t* sink, uint32_t* where) {
  __shared__ uint32_t lds[256];
  record_placement(where);
  uint32_t s = mix(blockIdx.x * 64u + threadIdx.x + 1u);
  uint32_t acc = 0;
  lds[threadIdx.x] = s;
  for (int i = 0; i < iters; ++i) {
    const uint4 v = buf[s & mask];
    acc += v.x ^ v.y ^ v.z ^ v.w;
    lds[(threadIdx.x + i) & 255] = acc;
    s = mix(s + lds[(threadIdx.x * 7 + i) & 255]);
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(128))) void foreign_gather(const uint4* buf, uint32_t mask, int iters, uint32_t* sink, uint32_t* where) {
  __shared__ uint4 lds[1024];               // 16 KiB
  record_placement(where);
  uint32_t s = mix(blockIdx.x * 256u + threadIdx.x + 1u);
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (int i = 0; i < iters; ++i) {
    uint4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = buf[mix(s + k) & mask];     // eight 16-byte gathers in flight per lane
#pragma unroll
    for (int k = 0; k < 8; ++k) { acc.x += v[k].x; acc.y ^= v[k].y; acc.z += v[k].z; acc.w ^= v[k].w; }
    lds[(threadIdx.x + 37 * i) & 1023] = acc;
    __syncthreads();
    const uint4 o = lds[(threadIdx.x * 5 + i) & 1023];
    s = mix(s ^ o.x ^ acc.w);
    __syncthreads();
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = acc.x;
}

__global__ __launch_bounds__(256) void foreign_dma(const uint4* buf, uint32_t mask, int iters, uint32_t* sink, uint32_t* where) {
  extern __shared__ __attribute__((aligned(16))) char smem[];       // 64 KiB: four 16 KiB slots
  record_placement(where);
  const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint4 acc = make_uint4(0, 0, 0, 0);
  const uint32_t nslot = (mask + 1u) / 1024u;                        // 16 KiB slots in the buffer (mask + 1 = uint4 elements, a power of two >= 4096)
  uint32_t slot = mix(blockIdx.x + 1u) % nslot;
  for (int i = 0; i < iters; ++i) {
    const char* src = (const char*)buf + (size_t)slot * 16384u + wave * 4096u;      // wave-uniform
    const uint32_t dst = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem + (uint32_t)(i & 3) * 16384u + wave * 4096u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      uint32_t keep;
      asm volatile(
          "s_mov_b32 %0, m0\n\t"
          "s_mov_b32 m0, %3\n\t"
          "s_nop 2\n\t"
          "global_load_lds_dwordx4 %1, %2\n\t"
          "s_mov_b32 m0, %0"
          : "=&s"(keep)
          : "v"(lane * 16u), "s"(src + k * 1024), "s"(dst + (uint32_t)(k * 1024))
          : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const uint4 o = *(const uint4*)(smem + ((i & 3) * 16384u + ((threadIdx.x * 16u + 64u * i) & 16383u)));
    acc.x += o.x; acc.y ^= o.y; acc.z += o.z; acc.w ^= o.w;
    slot = (slot + 1u + (acc.x & 1u)) % nslot;
    slot = __builtin_amdgcn_readfirstlane(slot);
    __syncthreads();
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = acc.x;
}

__global__ __launch_bounds__(256) void foreign_mfma(const uint4* buf, uint32_t mask, int iters, uint32_t* sink, uint32_t* where) {
  typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
  typedef __attribute__((ext_vector_type(4))) float f32x4;
  record_placement(where);
  asm volatile("v_mov_b32 v239, 0" ::: "v239");                  // 240 registers per wave
  const uint4 w = buf[(blockIdx.x * 256u + threadIdx.x) & mask];
  const bf16x8 a = __builtin_bit_cast(bf16x8, make_uint4(w.x & 0x3f803f80u, w.y & 0x3f803f80u, w.z & 0x3f803f80u, w.w & 0x3f803f80u));
  const bf16x8 b = __builtin_bit_cast(bf16x8, make_uint4(w.y & 0x3f803f80u, w.z & 0x3f803f80u, w.w & 0x3f803f80u, w.x & 0x3f803f80u));
  f32x4 c[4];
  for (int k = 0; k < 4; ++k) c[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) c[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[k & 3], 0, 0, 0);
  }
  float s = 0.f;
  for (int k = 0; k < 4; ++k) s += c[k][0] + c[k][1] + c[k][2] + c[k][3];
  if (s == 12345.678f) sink[0] = 1u;
}

}  // namespace

// buf: device buffer of `bytes` bytes (a power of two >= 64 KiB), read only; sink: 1 word; where: NULL or 4 words per workgroup.
extern "C" int foreign_launch(int kind, void* stream, int grid, int iters, const void* buf, uint64_t bytes, void* sink, void* where) {
  if (grid <= 0 || iters < 0 || !buf || bytes < 65536 || (bytes & (bytes - 1)) || bytes > (1ull << 35)) return -1;
  const uint32_t mask = (uint32_t)(bytes / 16 - 1);
  hipStream_t st = (hipStream_t)stream;
  if (kind == 0) hipLaunchKernelGGL(foreign_small, dim3(grid), dim3(64), 0, st, (const uint4*)buf, mask, iters, (uint32_t*)sink, (uint32_t*)where);
  else if (kind == 1) hipLaunchKernelGGL(foreign_gather, dim3(grid), dim3(256), 0, st, (const uint4*)buf, mask, iters, (uint32_t*)sink, (uint32_t*)where);
  else if (kind == 2) {
    static bool attr = false;
    if (!attr) { if (hipFuncSetAttribute((const void*)foreign_dma, hipFuncAttributeMaxDynamicSharedMemorySize, 65536) != hipSuccess) return -2; attr = true; }
    hipLaunchKernelGGL(foreign_dma, dim3(grid), dim3(256), 65536, st, (const uint4*)buf, mask, iters, (uint32_t*)sink, (uint32_t*)where);
  } else if (kind == 3) hipLaunchKernelGGL(foreign_mfma, dim3(grid), dim3(256), 0, st, (const uint4*)buf, mask, iters, (uint32_t*)sink, (uint32_t*)where);
  else return -1;
  return (int)hipGetLastError();
}
