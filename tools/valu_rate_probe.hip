// Diagnostic: issue cycles per VALU instruction on gfx950, one wave per SIMD, independent operands (no dependency stalls): the price list behind
// the ELU epilogue's options ("cut VALU issue cycles per activation").  For each instruction a loop body of 32
// copies over 8 rotating destination registers; cycles from s_memtime around 4000 bodies.  Also three mixes: does a transcendental overlap with the
// plain VALU instructions behind it?
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate_probe.hip -o /tmp/valu_rate_probe && /tmp/valu_rate_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

#define R8(INS)                                                                                                                                   \
  INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)
#define BODY32(NAME, TXT)                                                                                                                          \
  template <> struct Body<NAME> {                                                                                                                 \
    static __device__ __forceinline__ void run(float (&d)[8], float x, float y, float z) {                                                        \
      asm volatile(TXT TXT TXT TXT : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]) : "v"(x), "v"(y), "v"(z)); \
    }                                                                                                                                             \
  };
template <int K> struct Body;
// one line of 8 instructions: destinations %0..%7, sources %8 %9 %10
#define L8_1(OP) OP " %0, %8\n" OP " %1, %9\n" OP " %2, %10\n" OP " %3, %8\n" OP " %4, %9\n" OP " %5, %10\n" OP " %6, %8\n" OP " %7, %9\n"
#define L8_2(OP) OP " %0, %8, %9\n" OP " %1, %9, %10\n" OP " %2, %10, %8\n" OP " %3, %8, %9\n" OP " %4, %9, %10\n" OP " %5, %10, %8\n" OP " %6, %8, %9\n" OP " %7, %9, %10\n"
#define L8_3(OP) OP " %0, %8, %9, %10\n" OP " %1, %9, %10, %8\n" OP " %2, %10, %8, %9\n" OP " %3, %8, %9, %10\n" OP " %4, %9, %10, %8\n" OP " %5, %10, %8, %9\n" OP " %6, %8, %9, %10\n" OP " %7, %9, %10, %8\n"
enum { EXP32, EXP16, RCP32, FMA32, MED3, MAX32, CVTPK, PKFMA16, PKMAX16, PKADD16, PKMADU16, PKFMA32, PKMUL32, DOT2, MIX_EXP_3FMA, MIX_EXP_7FMA, MIX_ELU, MIX_ELU_PK, NKIND };
BODY32(EXP32, L8_1("v_exp_f32"))
BODY32(EXP16, L8_1("v_exp_f16"))
BODY32(RCP32, L8_1("v_rcp_f32"))
BODY32(FMA32, L8_3("v_fma_f32"))
BODY32(MED3, L8_3("v_med3_f32"))
BODY32(MAX32, L8_2("v_max_f32"))
BODY32(CVTPK, L8_2("v_cvt_pk_f16_f32"))
BODY32(PKFMA16, L8_3("v_pk_fma_f16"))
BODY32(PKMAX16, L8_2("v_pk_max_f16"))
BODY32(PKADD16, L8_2("v_pk_add_f16"))
BODY32(PKMADU16, L8_3("v_pk_mad_u16"))
BODY32(DOT2, L8_3("v_dot2_f32_f16"))
// packed fp32: 64-bit operands — register pairs
template <> struct Body<PKFMA32> {
  static __device__ __forceinline__ void run(float (&d)[8], float x, float y, float z) {
    typedef __attribute__((ext_vector_type(2))) float f2;
    f2 a = {d[0], d[1]}, b = {d[2], d[3]}, c = {d[4], d[5]}, e = {d[6], d[7]}, s0 = {x, y}, s1 = {y, z}, s2 = {z, x};
#define P4 "v_pk_fma_f32 %0, %4, %5, %6\nv_pk_fma_f32 %1, %5, %6, %4\nv_pk_fma_f32 %2, %6, %4, %5\nv_pk_fma_f32 %3, %4, %5, %6\n"
    asm volatile(P4 P4 P4 P4 P4 P4 P4 P4 : "+v"(a), "+v"(b), "+v"(c), "+v"(e) : "v"(s0), "v"(s1), "v"(s2));
#undef P4
    d[0] = a[0]; d[1] = a[1]; d[2] = b[0]; d[3] = b[1]; d[4] = c[0]; d[5] = c[1]; d[6] = e[0]; d[7] = e[1];
  }
};
template <> struct Body<PKMUL32> {
  static __device__ __forceinline__ void run(float (&d)[8], float x, float y, float z) {
    typedef __attribute__((ext_vector_type(2))) float f2;
    f2 a = {d[0], d[1]}, b = {d[2], d[3]}, c = {d[4], d[5]}, e = {d[6], d[7]}, s0 = {x, y}, s1 = {y, z};
#define P4 "v_pk_mul_f32 %0, %4, %5\nv_pk_mul_f32 %1, %5, %4\nv_pk_mul_f32 %2, %4, %5\nv_pk_mul_f32 %3, %5, %4\n"
    asm volatile(P4 P4 P4 P4 P4 P4 P4 P4 : "+v"(a), "+v"(b), "+v"(c), "+v"(e) : "v"(s0), "v"(s1));
#undef P4
    d[0] = a[0]; d[1] = a[1]; d[2] = b[0]; d[3] = b[1]; d[4] = c[0]; d[5] = c[1]; d[6] = e[0]; d[7] = e[1];
  }
};
// mixes (32 instructions each): 1 exp + 3 fma (x8); 1 exp + 7 fma (x4); the ELU of today per activation: exp, fma, med3 (+ cvt_pk every second) ;
// the packed-fp32 form: exp (clamp), max, then per PAIR one v_pk_fma_f32 and one cvt_pk
#define M_E3F "v_exp_f32 %0, %8\nv_fma_f32 %1, %8, %9, %10\nv_fma_f32 %2, %9, %10, %8\nv_fma_f32 %3, %10, %8, %9\n" \
              "v_exp_f32 %4, %9\nv_fma_f32 %5, %8, %9, %10\nv_fma_f32 %6, %9, %10, %8\nv_fma_f32 %7, %10, %8, %9\n"
BODY32(MIX_EXP_3FMA, M_E3F)
#define M_E7F "v_exp_f32 %0, %8\nv_fma_f32 %1, %8, %9, %10\nv_fma_f32 %2, %9, %10, %8\nv_fma_f32 %3, %10, %8, %9\n" \
              "v_fma_f32 %4, %9, %8, %10\nv_fma_f32 %5, %8, %9, %10\nv_fma_f32 %6, %9, %10, %8\nv_fma_f32 %7, %10, %8, %9\n"
BODY32(MIX_EXP_7FMA, M_E7F)
// 2 activations = 7 instructions (+1 filler fma to keep 8 per line): exp, fma, med3, exp, fma, med3, cvt_pk
#define M_ELU "v_exp_f32 %0, %8\nv_fma_f32 %1, %0, %9, %10\nv_med3_f32 %2, %8, %1, %10\nv_exp_f32 %3, %9\nv_fma_f32 %4, %3, %9, %10\nv_med3_f32 %5, %9, %4, %10\n" \
              "v_cvt_pk_f16_f32 %6, %2, %5\nv_nop\n"
BODY32(MIX_ELU, M_ELU)
template <> struct Body<MIX_ELU_PK> {        // 2 activations = 6 instructions: exp clamp, exp clamp, max, max, pk_fma_f32, cvt_pk  (+ 2 nops to keep the body at 32 lines)
  static __device__ __forceinline__ void run(float (&d)[8], float x, float y, float z) {
    typedef __attribute__((ext_vector_type(2))) float f2;
    f2 e = {d[0], d[1]}, r = {d[2], d[3]}, L = {y, y};
    float pk = d[4];
#define P "v_exp_f32 %0, %4 clamp\nv_exp_f32 %1, %5 clamp\nv_max_f32 %2, 0, %4\nv_max_f32 %3, 0, %5\n"
#define Q "v_pk_fma_f32 %1, %0, %3, %1\nv_cvt_pk_f16_f32 %2, %1, %1\nv_nop\nv_nop\n"
    float e0 = e[0], e1 = e[1], r0 = r[0], r1 = r[1];
    for (int k = 0; k < 4; ++k) {
      asm volatile(P : "+v"(e0), "+v"(e1), "+v"(r0), "+v"(r1) : "v"(x), "v"(z));
      f2 ee = {e0, e1}, rr = {r0, r1};
      asm volatile("v_pk_fma_f32 %1, %0, %2, %1" : "+v"(ee), "+v"(rr) : "v"(L));
      e0 = ee[0]; e1 = ee[1]; r0 = rr[0]; r1 = rr[1];
      asm volatile("v_cvt_pk_f16_f32 %0, %1, %2\nv_nop\nv_nop" : "+v"(pk) : "v"(r0), "v"(r1));
    }
#undef P
#undef Q
    d[0] = e0; d[1] = e1; d[2] = r0; d[3] = r1; d[4] = pk;
  }
};

template <int K>
__global__ __launch_bounds__(256) void probe(float* out, int iters, unsigned long long* cyc) {
  float d[8];
  for (int i = 0; i < 8; ++i) d[i] = 0.001f * (float)(threadIdx.x + i);
  const float x = -0.25f - 0.001f * (float)(threadIdx.x & 31), y = 1.4426950408889634f, z = 0.5f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) Body<K>::run(d, x, y, z);
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += d[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  float* out; (void)hipMalloc(&out, 256 * 256 * 4);
  unsigned long long* cyc; (void)hipMalloc(&cyc, 256 * 8);
  const int iters = 4000;
  const char* names[NKIND] = {"v_exp_f32", "v_exp_f16", "v_rcp_f32", "v_fma_f32", "v_med3_f32", "v_max_f32", "v_cvt_pk_f16_f32", "v_pk_fma_f16", "v_pk_max_f16", "v_pk_add_f16",
                              "v_pk_mad_u16", "v_pk_fma_f32", "v_pk_mul_f32", "v_dot2_f32_f16", "mix: 1 exp + 3 fma", "mix: 1 exp + 7 fma",
                              "mix: ELU today (exp fma med3 | exp fma med3 | cvt_pk | nop)", "mix: ELU packed-f32 (exp.clamp x2, max x2, pk_fma_f32, cvt_pk, 2 nop)"};
  auto run = [&](int k, auto launch) {
    launch(); (void)hipDeviceSynchronize();
    launch(); (void)hipDeviceSynchronize();
    unsigned long long h[256]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double c = 0; for (int i = 0; i < 256; ++i) c += (double)h[i];
    c /= 256.0;
    printf("%-90s %7.2f cycles per instruction (32 per body)\n", names[k], c / (iters * 32.0));
  };
#define RUN(K) run(K, [&] { hipLaunchKernelGGL((probe<K>), dim3(256), dim3(256), 0, 0, out, iters, cyc); })
  RUN(EXP32); RUN(EXP16); RUN(RCP32); RUN(FMA32); RUN(MED3); RUN(MAX32); RUN(CVTPK); RUN(PKFMA16); RUN(PKMAX16); RUN(PKADD16); RUN(PKMADU16);
  RUN(PKFMA32); RUN(PKMUL32); RUN(DOT2); RUN(MIX_EXP_3FMA); RUN(MIX_EXP_7FMA); RUN(MIX_ELU); RUN(MIX_ELU_PK);
  return 0;
}
