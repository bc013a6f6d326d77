// Micro-reproducer for round 4's "co-residency hazard" (NOTEBOOK §12, §19): does compiler-generated PACKED FP32 arithmetic (v_pk_mul_f32 / v_pk_add_f32 /
// v_pk_fma_f32 / v_pk_mov_b32, formed by the SLP vectorizer from scalar fp32 code) give the same bits when a wave of a DIFFERENT kernel shares its SIMD?
//
// Victim kernel: per iteration a short MFMA burst, then the fused refine stage's scalar epilogue (sigmoid / tanh through v_exp_f32 + v_rcp_f32, interval
// refinement, query points o + d z + 0.01 tanh — pnrf_mlp_kernels.hip, refine_kernel) on values derived from (lane, iteration); a running xor of the
// results is dumped every 64 iterations.  4 waves and 72 KiB of LDS per workgroup: two workgroups fit a CU, one wave of each per SIMD.
// Aggressor kernels on another stream, same footprint: 0 = plain VALU, 1 = MFMA 16x16x32 bf16 from registers, 2 = MFMA 32x32x16 f16, 3 = transcendentals,
// 4 = LDS-DMA ring + ds_read, 5 = MFMA 16x16x32 + ds_read (the fused engines' inner loop), 6 = random 16-byte gathers (the refine head's texel traffic).
// The victim's dump alone is the reference; every differing dump beside an aggressor is a corruption event.  Built twice: the packed form (default flags)
// and with -Xclang -target-feature -Xclang -packed-fp32-ops (no packed fp32 instructions), which must stay clean.
//   hipcc --offload-arch=gfx950 -O3 tools/pkf32_coexec_probe.hip -o pronerf_amd/lib/pkf32_coexec_probe        [-DNO_PK marks the output]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__device__ __forceinline__ float ieee_mul(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float ieee_add(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ float ieee_sub(float a, float b) {
#pragma clang fp contract(off)
  return a - b;
}
__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanh_fast(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * x) + 1.f); }

constexpr int DUMP_EVERY = 64;

__global__ __launch_bounds__(256) void victim(unsigned* dump, int iters, const float* rays, const float* depth, float* zout, float* pout) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, h = lane >> 5;
  const int gid = blockIdx.x * 256 + threadIdx.x;
  // the epilogue's per-ray inputs (held in registers across the batch in the real kernel)
  float r[8];
  for (int i = 0; i < 8; ++i) r[i] = rays[(size_t)gid * 8 + i];
  const f32x4 d0 = *(const f32x4*)(depth + (size_t)gid * 8), d1 = *(const f32x4*)(depth + (size_t)gid * 8 + 4);
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * (float)((hash(gid * 8 + i) & 255) - 128)); b[i] = (_Float16)(0.01f * (float)((hash(gid * 8 + i + 77777) & 255) - 128)); }
  unsigned x = 0;
  ((volatile unsigned*)smem)[threadIdx.x] = gid;
#ifdef VICTIM_OVFL          // the fused fp16 kernels run with MODE.FP16_OVFL = 1 (PrecF16::enter); the NeRF stage's bf16 kernel beside them does not
  __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);
#endif
#if defined(BIG_VGPR) || defined(VICTIM_BIG)             // 240 registers per wave like the fused kernels: victim + aggressor fill 480 of the SIMD's 512
  asm volatile("v_mov_b32 v239, 0" ::: "v239");
#endif
  for (int it = 0; it < iters; ++it) {
    f32x16 fin;
    for (int i = 0; i < 16; ++i) fin[i] = 0.001f * (float)((int)(hash(it * 16 + i) & 1023) - 512);
#pragma unroll
    for (int k = 0; k < 4; ++k) fin = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, fin, 0, 0, 0);
    // ---- the fused refine epilogue (pnrf_mlp_kernels.hip), verbatim arithmetic
    const float ox = r[0], oy = r[1], oz = r[2], dx = r[3], dy = r[4], dz = r[5], near = r[6], far = r[7];
    float w[6];
    w[0] = h ? d0.w : near; w[1] = h ? d1.x : d0.x; w[2] = h ? d1.y : d0.y;
    w[3] = h ? d1.z : d0.z; w[4] = h ? d1.w : d0.w; w[5] = h ? far : d1.x;
    float zz[4], pp[12];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const float lower = ieee_mul(0.5f, ieee_add(w[s4 + 1], w[s4]));
      const float upper = ieee_mul(0.5f, ieee_add(w[s4 + 2], w[s4 + 1]));
      const float rf = sigmoid_fast(fin[4 * s4]);
      zz[s4] = ieee_add(lower, ieee_mul(ieee_sub(upper, lower), rf));
    }
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const float zv = zz[s4];
      const float fx = tanh_fast(fin[4 * s4 + 1]), fy = tanh_fast(fin[4 * s4 + 2]), fz = tanh_fast(fin[4 * s4 + 3]);
      pp[3 * s4 + 0] = ieee_add(ieee_add(ox, ieee_mul(dx, zv)), ieee_mul(1e-2f, fx));
      pp[3 * s4 + 1] = ieee_add(ieee_add(oy, ieee_mul(dy, zv)), ieee_mul(1e-2f, fy));
      pp[3 * s4 + 2] = ieee_add(ieee_add(oz, ieee_mul(dz, zv)), ieee_mul(1e-2f, fz));
    }
    {           // the epilogue's four 16-byte stores (z row half, three quarters of the pts row), interleaved with the arithmetic by the scheduler as in the kernel
      *(f32x4*)(zout + (size_t)gid * 4) = f32x4{zz[0], zz[1], zz[2], zz[3]};
      f32x4* pq = (f32x4*)(pout + (size_t)gid * 12);
      pq[0] = f32x4{pp[0], pp[1], pp[2], pp[3]};
      pq[1] = f32x4{pp[4], pp[5], pp[6], pp[7]};
      pq[2] = f32x4{pp[8], pp[9], pp[10], pp[11]};
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) x = (x * 31u) ^ __float_as_uint(zz[i]);
#pragma unroll
    for (int i = 0; i < 12; ++i) x = (x * 31u) ^ __float_as_uint(pp[i]);
    if ((it & (DUMP_EVERY - 1)) == DUMP_EVERY - 1) dump[((size_t)(it / DUMP_EVERY) * gridDim.x + blockIdx.x) * 256 + threadIdx.x] = x;
    __builtin_amdgcn_s_barrier();
  }
}

template <int KIND>
__global__ __launch_bounds__(256) void aggressor(float* sink, int iters, const u32x4* buf) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gid = blockIdx.x * 256 + threadIdx.x;
  for (int i = threadIdx.x; i < 4096; i += 256) ((u32x4*)smem)[i] = buf[(blockIdx.x * 4096 + i) & 65535];
  __syncthreads();
  f16x8 a, b; bf16x8 ab, bb;
  for (int i = 0; i < 8; ++i) {
    a[i] = (_Float16)(0.01f * (float)((hash(gid * 8 + i) & 255) - 128)); b[i] = (_Float16)(0.02f * (float)((hash(gid * 8 + i + 555) & 255) - 128));
    ab[i] = (__bf16)(float)a[i]; bb[i] = (__bf16)(float)b[i];
  }
  f32x16 c32; f32x4 c4[4];
  for (int i = 0; i < 16; ++i) c32[i] = 0.f;
  for (int k = 0; k < 4; ++k) c4[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = 0.001f * (float)(gid + i);
#if defined(BIG_VGPR) || defined(AGGR_BIG)
  asm volatile("v_mov_b32 v239, 0" ::: "v239");
#endif
#ifdef AGGR_OVFL
  __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);
#endif
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) {
#pragma unroll
      for (int k = 0; k < 64; ++k) v[k & 7] = fmaf(v[k & 7], 0.999f, 0.001f);
    } else if (KIND == 1) {
#pragma unroll
      for (int k = 0; k < 16; ++k) c4[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, c4[k & 3], 0, 0, 0);
    } else if (KIND == 2) {
#pragma unroll
      for (int k = 0; k < 8; ++k) c32 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c32, 0, 0, 0);
    } else if (KIND == 3) {
#pragma unroll
      for (int k = 0; k < 16; ++k) { v[k & 7] = __builtin_amdgcn_exp2f(v[k & 7] * -0.5f); v[(k + 3) & 7] = __builtin_amdgcn_sinf(v[(k + 3) & 7]) + __builtin_amdgcn_rcpf(1.5f + v[k & 7]); }
    } else if (KIND == 4) {
      const char* src = (const char*)buf + (size_t)((blockIdx.x * 7 + it) & 63) * 16384u + wave * 4096u;
      const unsigned dst = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem + (unsigned)(it & 3) * 16384u + wave * 4096u;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 2\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(lane * 16u), "s"(src + k * 1024), "s"(dst + (unsigned)(k * 1024)) : "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const u32x4 o = *(const u32x4*)(smem + ((it & 3) * 16384u + ((threadIdx.x * 16u + 64u * it) & 16383u)));
      v[0] += (float)(o.x & 255);
      __syncthreads();
    } else if (KIND == 6) {          // random 16-byte gathers, eight in flight per lane (the refine head's texel traffic)
      u32x4 t[8];
      const unsigned s0 = hash(gid * 131u + it);
#pragma unroll
      for (int k = 0; k < 8; ++k) t[k] = buf[hash(s0 + k) & 65535];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] += (float)(t[k].x & 255);
    } else {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const bf16x8 fa = __builtin_bit_cast(bf16x8, *(const u32x4*)(smem + (((it * 16 + k) & 63) * 1024 + lane * 16)));
        c4[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, bb, c4[k & 3], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += v[i];
  for (int i = 0; i < 16; ++i) s += c32[i];
  for (int k = 0; k < 4; ++k) s += c4[k][0] + c4[k][1] + c4[k][2] + c4[k][3];
  if (s == 12345.678f) sink[gid] = s;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 32768;
  const int G = 256, LDS = 72 * 1024;
  const size_t ndump = (size_t)(iters / DUMP_EVERY) * G * 256;
  unsigned *dump; float *sink, *rays, *depth; u32x4* buf;
  CK(hipMalloc(&dump, ndump * 4)); CK(hipMalloc(&sink, G * 256 * 4)); CK(hipMalloc(&rays, (size_t)G * 256 * 8 * 4)); CK(hipMalloc(&depth, (size_t)G * 256 * 8 * 4));
  CK(hipMalloc(&buf, 65536 * 16));
  {
    std::vector<float> hr((size_t)G * 256 * 8), hd((size_t)G * 256 * 8);
    std::vector<unsigned> hb(65536 * 4);
    unsigned s = 12345;
    auto rnd = [&] { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.f; };
    for (size_t i = 0; i < hr.size(); i += 8) {
      hr[i] = rnd() * 2 - 1; hr[i + 1] = rnd() * 2 - 1; hr[i + 2] = -1.f; hr[i + 3] = rnd() * 0.4f - 0.2f; hr[i + 4] = rnd() * 0.4f - 0.2f; hr[i + 5] = 2.f; hr[i + 6] = 0.f; hr[i + 7] = 1.f;
      float acc = 0.f;
      for (int k = 0; k < 8; ++k) { acc += 0.02f + rnd() * 0.1f; hd[i + k] = acc; }
    }
    for (auto& w : hb) { s = s * 1664525u + 1013904223u; w = s & 0x3bff3bffu; }
    CK(hipMemcpy(rays, hr.data(), hr.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(depth, hd.data(), hd.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(buf, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  }
  CK(hipFuncSetAttribute((const void*)victim, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  hipStream_t sa, sb; CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
  std::vector<unsigned> ref(ndump), got(ndump);
  CK(hipMemset(dump, 0, ndump * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, sa));
  float *zout, *pout;
  CK(hipMalloc(&zout, (size_t)G * 256 * 4 * 4)); CK(hipMalloc(&pout, (size_t)G * 256 * 12 * 4));
  hipLaunchKernelGGL(victim, dim3(G), dim3(256), LDS, sa, dump, iters, rays, depth, zout, pout);
  CK(hipEventRecord(e1, sa));
  CK(hipStreamSynchronize(sa));
  float ms_alone; CK(hipEventElapsedTime(&ms_alone, e0, e1));
  CK(hipMemcpy(ref.data(), dump, ndump * 4, hipMemcpyDeviceToHost));
  // determinism alone
  hipLaunchKernelGGL(victim, dim3(G), dim3(256), LDS, sa, dump, iters, rays, depth, zout, pout);
  CK(hipStreamSynchronize(sa));
  CK(hipMemcpy(got.data(), dump, ndump * 4, hipMemcpyDeviceToHost));
  size_t self = 0;
  for (size_t i = 0; i < ndump; ++i) self += got[i] != ref[i];
#ifdef BUILD_NAME
  const char* build = BUILD_NAME;
#elif defined(NO_PK)
  const char* build = "no packed fp32 instructions (-packed-fp32-ops)";
#elif defined(VICTIM_OVFL) && defined(BIG_VGPR)
  const char* build = "packed fp32; victim with MODE.FP16_OVFL; 240 VGPRs per wave on both sides";
#elif defined(VICTIM_OVFL)
  const char* build = "packed fp32; victim with MODE.FP16_OVFL";
#elif defined(BIG_VGPR)
  const char* build = "packed fp32; 240 VGPRs per wave on both sides";
#else
  const char* build = "packed fp32 (default flags)";
#endif
  printf("{\"build\": \"%s\", \"iters\": %d, \"victim_alone_ms\": %.2f, \"alone_vs_alone_dumps_differ\": %zu", build, iters, ms_alone, self);
  const char* names[7] = {"valu", "mfma_16x16x32_bf16", "mfma_32x32x16_f16", "transcendental", "lds_dma", "mfma_16x16x32_bf16+ds_read", "gather16"};
  auto launch_b = [&](int kind, int n) {
    switch (kind) {
      case 0: hipLaunchKernelGGL((aggressor<0>), dim3(G), dim3(256), LDS, sb, sink, n, buf); break;
      case 1: hipLaunchKernelGGL((aggressor<1>), dim3(G), dim3(256), LDS, sb, sink, n, buf); break;
      case 2: hipLaunchKernelGGL((aggressor<2>), dim3(G), dim3(256), LDS, sb, sink, n, buf); break;
      case 3: hipLaunchKernelGGL((aggressor<3>), dim3(G), dim3(256), LDS, sb, sink, n, buf); break;
      case 4: hipLaunchKernelGGL((aggressor<4>), dim3(G), dim3(256), LDS, sb, sink, n, buf); break;
      case 5: hipLaunchKernelGGL((aggressor<5>), dim3(G), dim3(256), LDS, sb, sink, n, buf); break;
      default: hipLaunchKernelGGL((aggressor<6>), dim3(G), dim3(256), LDS, sb, sink, n, buf); break;
    }
  };
  CK(hipFuncSetAttribute((const void*)aggressor<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)); CK(hipFuncSetAttribute((const void*)aggressor<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  CK(hipFuncSetAttribute((const void*)aggressor<2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)); CK(hipFuncSetAttribute((const void*)aggressor<3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  CK(hipFuncSetAttribute((const void*)aggressor<4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)); CK(hipFuncSetAttribute((const void*)aggressor<5>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  CK(hipFuncSetAttribute((const void*)aggressor<6>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  for (int kind = 0; kind < 7; ++kind) {
    // size the aggressor to ~1/8 of the victim's time, launched back to back on its own stream while the victim runs: the two keep changing phase
    int n = 64;
    float ms = 0.f;
    for (int t = 0; t < 8; ++t) {
      CK(hipEventRecord(e0, sb)); launch_b(kind, n); CK(hipEventRecord(e1, sb)); CK(hipStreamSynchronize(sb)); CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms > ms_alone / 8) break;
      n *= 2;
    }
    size_t events = 0, lanes_quad[4] = {0, 0, 0, 0};
    float ms_with = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(dump, 0, ndump * 4));
      CK(hipDeviceSynchronize());
      for (int t = 0; t < 24; ++t) launch_b(kind, n);         // ~3x the victim's time of aggressor launches queued on stream b
      CK(hipEventRecord(e0, sa));
      hipLaunchKernelGGL(victim, dim3(G), dim3(256), LDS, sa, dump, iters, rays, depth, zout, pout);
      CK(hipEventRecord(e1, sa));
      CK(hipDeviceSynchronize());
      CK(hipEventElapsedTime(&ms_with, e0, e1));
      CK(hipMemcpy(got.data(), dump, ndump * 4, hipMemcpyDeviceToHost));
      // a corruption event changes a thread's running xor from that dump on: count the FIRST differing dump per thread
      for (int wg = 0; wg < G; ++wg)
        for (int t = 0; t < 256; ++t)
          for (int dmp = 0; dmp < iters / DUMP_EVERY; ++dmp) {
            const size_t i = ((size_t)dmp * G + wg) * 256 + t;
            if (got[i] != ref[i]) { events += 1; lanes_quad[(t & 63) >> 4] += 1; break; }
          }
    }
    printf(", \"%s\": {\"aggressor_iters\": %d, \"aggressor_ms\": %.2f, \"victim_ms_beside_it\": %.2f, \"threads_with_a_wrong_result\": %zu, \"by_lane_quad\": [%zu, %zu, %zu, %zu]}", names[kind], n, ms,
           ms_with, events, lanes_quad[0], lanes_quad[1], lanes_quad[2], lanes_quad[3]);
    fflush(stdout);
  }
  printf("}\n");
  return 0;
}
