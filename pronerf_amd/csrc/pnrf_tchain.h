// pnrf_tchain.h — the fine net's forward pass of a training iteration (pts0 .. pts7, feature_linear: run_nerf_helpers.py:824-841) as ONE launch on
// the fused-MLP engine of the inference path (pnrf_engine.h, layer_h16x2).  Included by pnrf_train.hip inside its anonymous namespace.
//
// Why: as products of their own (hgemm_kernel) or as 64-row layer chains (hgemm_wchain_kernel) every 64 rows stream a layer's 256 KiB of fp16
// weight planes from L2 — 4 KiB per row and layer, 41 TB/s at the MFMA rate — and read their input rows back from HBM; the MFMA pipes were 20-31 %
// busy.  Here a workgroup holds 128 rows (8 waves x 16 columns) in REGISTERS as the MFMA B operand through all nine layers, the weights come
// through the LDS ring once per 128 rows (2 KiB per row and layer, shared by the eight waves), and HBM sees one read of the 64 input columns and
// one fp32 write per layer — the saved activations the backward pass and the weight gradients need.
//
// Arithmetic: the split-fp16 product of pnrf_hgemm.h — x = x_hi + 2^-11 x_lo', main += W_hi x_hi, cross += W_hi x_lo' + W_lo' x_hi, fp32
// accumulation, bias in the accumulator, C = main + 2^-11 cross — on the same v_mfma_f32_16x16x32_f16, contraction in the engine's order.
// Results agree with the per-layer products to fp32 round-off (different summation order), not bit for bit.
//
// Stream: the trainer's parameters change every iteration, so the engine's fragment stream is rebuilt on the device whenever the fp16 planes
// are (tchain_pack_kernel): fragments (tile pair, k-step, tile of the pair, plane) of 16 output rows x 32 k, layer after layer, every layer a
// whole number of ring revolutions so that all of them start at ring position 0.
#pragma once

constexpr int TC_NL = 9;                                   // pts0 .. pts7, feature
__host__ __device__ constexpr int tc_ks(int l) { return l == 0 ? 2 : (l == 5 ? 10 : 8); }      // k-steps of 32: 63 (+1) | 256 | 64 + 256 inputs
__host__ __device__ constexpr int tc_frags(int l) { return 8 * tc_ks(l) * 4; }
__host__ __device__ constexpr int tc_frag0(int l) { int s = 0; for (int i = 0; i < l; ++i) s += tc_frags(i); return s; }
constexpr int TC_NFRAGS = tc_frag0(TC_NL);                 // 2176
constexpr int TC_NSLOTS = TC_NFRAGS / SLOT_FRAGS;          // 136 slots of 16 KiB
constexpr int TC_ROWS = 128;                               // rows per workgroup and batch
#ifndef PNRF_TC_QUEUE
#define PNRF_TC_QUEUE 8
#endif
constexpr int TC_QUEUE = PNRF_TC_QUEUE;                    // A fragments held in registers ahead of their MFMAs
constexpr int TC_RING = NSLOTS;                           // (a 7-of-8 ring, WStream<8, 7, 8>, measured no faster: the stores' cost is not their vmcnt)
constexpr int TC_RING_BYTES = TC_RING * SLOT_BYTES;
constexpr int TC_STG_ROW = 128 + 4;                        // floats per staged row: 128 features + padding (bank spread of the 16-byte accesses)
constexpr int TC_STG_BYTES = 8 * 16 * TC_STG_ROW * 4;      // eight waves x 16 rows
constexpr int TC_LDS_BYTES = TC_RING_BYTES + TC_NL * W_HID * 4 + TC_STG_BYTES;
__host__ __device__ constexpr int tc_pos(int l) { return (tc_frag0(l) / SLOT_FRAGS) % TC_RING; }   // ring position of layer l's first slot
static_assert(tc_frags(0) % SLOT_FRAGS == 0 && tc_frags(1) % SLOT_FRAGS == 0 && tc_frags(5) % SLOT_FRAGS == 0 && TC_NSLOTS % TC_RING == 0,
              "whole slots per layer, whole ring revolutions per batch");
static_assert(H16_PIECES == 8, "tchain_fwd_kernel's epilogue pieces are written for one accumulator register per piece");

typedef int tc_i32x4 __attribute__((ext_vector_type(4)));

struct TChainPackArgs {
  const float* P;                    // the trainer's flat parameters
  size_t w[TC_NL];                   // offset of layer l's weights [256][in_dim]
  int in_dim[TC_NL];                 // 63 | 256 | 319 (skip layer: [embedding 63 | h 256], run_nerf_helpers.py:829-831)
  _Float16* stream;                  // TC_NFRAGS KiB
};
// one thread per 16-byte piece of the stream (lane `lane` of fragment F)
__global__ void tchain_pack_kernel(TChainPackArgs a) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= TC_NFRAGS * 64) return;
  const int F = p >> 6, lane = p & 63;
  int l = 0;
  while (l + 1 < TC_NL && F >= tc_frag0(l + 1)) ++l;
  const int f = F - tc_frag0(l), pl = f & 1, t = (f >> 1) & 1, KS = tc_ks(l), ks = (f >> 2) % KS, tp = (f >> 2) / KS;
  const int out = 32 * tp + 16 * t + (lane & 15), g = lane >> 4;
  const float* W = a.P + a.w[l] + (size_t)out * a.in_dim[l];
  f16x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    int in;
    if (l == 0 || (l == 5 && ks < 2)) {                    // the 64 input columns in their natural order; column 63 is padding
      in = 32 * ks + 8 * g + j;
      if (in >= 63) in = -1;
    } else {
      in = hidden_feat_h16(l == 5 ? ks - 2 : ks, g, j) + (l == 5 ? 63 : 0);
    }
    const float x = in >= 0 ? W[in] : 0.f;
    const _Float16 h = (_Float16)x;
    v[j] = pl ? (_Float16)((x - (float)h) * H16_LO_SCALE) : h;
  }
  *(f16x8*)(a.stream + (size_t)p * 8) = v;
}

// Store of saved activations: 2.4 GB per launch at 262 144 rows that nobody reads before the backward pass.  Left in the L2 (plain stores) they
// turn it over every ~13 us and evict the 2.2 MB weight stream between two batches of a workgroup (FETCH_SIZE: 0.3 GB per launch for 67 MB of
// input).  Measured per launch at 262 144 rows: plain 954 us, sc1 945, sc0 sc1 939, nt 802 (no stores at all: 695).
#ifndef PNRF_TC_STORE_MODE
#define PNRF_TC_STORE_MODE 3
#endif
__device__ __forceinline__ void tc_store(float* p, const f32x4& v) {
#if PNRF_TC_STORE_MODE == 0
  *(f32x4*)p = v;
#elif PNRF_TC_STORE_MODE == 1
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v));
#elif PNRF_TC_STORE_MODE == 2
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v));
#else
  __builtin_nontemporal_store(v, (f32x4*)p);       // (the compiler's own store: as inline asm it ran, but its data-register hazards were nobody's business)
#endif
}

struct TChainArgs {
  const void* blob;                  // TC_NSLOTS slots
  const float* bias[TC_NL];
  const float* X0; int ldx0;         // [n][ldx0] fp32, 16-byte aligned rows: columns 0 .. 63 = position embedding (63) + one zero
  float* out[TC_NL]; int ldo[TC_NL]; // saved activations [n][ldo], 16-byte aligned rows
  int64_t n; int nbatch;
};

__global__ __launch_bounds__(512, 2) void tchain_fwd_kernel(TChainArgs a) {
  constexpr int NW = 8, NTP = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* bias_lds = (float*)(smem + TC_RING_BYTES);        // [layer][256]
  for (int i = threadIdx.x; i < TC_NL * W_HID; i += 512) bias_lds[i] = a.bias[i >> 8][i & 255];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 15, g = lane >> 4;
  WStream<NW> st;
  st.init(a.blob, TC_NSLOTS, smem);
  st.prologue();
  if (__builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(1);      // as young_half_priority (pnrf_mlp_kernels.hip)
  const char* ringlane = smem + lane * 16;
  const float* biaslane = bias_lds + 4 * g;
  constexpr float INV = 1.f / H16_LO_SCALE;

  for (int batch = blockIdx.x; batch < a.nbatch; batch += gridDim.x) {
    const int64_t row = (int64_t)batch * TC_ROWS + wave * 16 + col;
    const bool valid = row < a.n;
    const int64_t rr = valid ? row : a.n - 1;
    f16x8 Gh[2], Gl[2];                    // the 64 input columns: k-step ks, group g = columns 32 ks + 8 g .. + 7
    {
      const float4* x = (const float4*)(a.X0 + rr * a.ldx0 + 8 * g);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const float4 lo = x[8 * ks], hi = x[8 * ks + 1];
        const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const _Float16 h = (_Float16)v[j];
          Gh[ks][j] = h;
          Gl[ks][j] = (_Float16)((v[j] - (float)h) * H16_LO_SCALE);
        }
      }
    }
    f16x8 Xh[NTP], Xl[NTP], Yh[NTP], Yl[NTP];      // activations ping-pong between X and Y: per 32-feature k-step one hi and one lo plane
    f32x4 pm[2], pc[2];                            // pending (deferred) tile pair of the previous layer
    // Saved activations leave through a per-wave staging tile in LDS: four tile pairs (128 features) of the wave's 16 rows are collected, then
    // written out as whole 512-byte row segments (one store instruction = 2 rows x 512 B).  Stored straight from the accumulators — lane
    // (column, g) holds 4 features of ONE row, an instruction covers 16 rows x 64 B — the same bytes reached HBM in 128-byte pieces at a 1 KiB
    // stride and the launch ran at 2.3 TB/s of writes (1.06 ms at 262 144 rows; 0.65 ms without the stores; a plain fill writes 6.8 TB/s).
    float* const stage = (float*)(smem + TC_RING_BYTES + TC_NL * W_HID * 4) + wave * (16 * TC_STG_ROW);
    float* const stage_w = stage + col * TC_STG_ROW + 4 * g;                       // this lane's accumulator tiles go here (+ 32 (tp & 3) + 16 t)
    const float* const stage_r = stage + (lane >> 5) * TC_STG_ROW + 4 * (lane & 31);   // ... and it reads rows 2 i + (lane >> 5), 16 bytes at 4 (lane & 31)
    const int64_t row_f = (int64_t)batch * TC_ROWS + wave * 16 + (lane >> 5);      // first of the rows this lane writes out
    const int rows_left = (int)(a.n - row_f < 16 ? a.n - row_f : 16);              // rows 2 i with 2 i < rows_left exist
    float* o_prev = nullptr;                       // where the pending tile pair's rows go (flush base of its layer) ...
    int ld_prev = 0;

    // piece pcx = accumulator register pcx & 3 of tile pcx >> 2 of a tile pair: bias is in the accumulator; combine, activate (floor = 0: ReLU,
    // -inf: none), park the value in the accumulator register; the odd piece packs the pair into the next layer's planes (as sampler_h16_kernel),
    // the last piece of a tile stages its four features — registers 0 .. 3 of lane (column, g) are features 16 T + 4 g .. + 3 of the row
    auto piece = [&](f16x8(&dh)[NTP], f16x8(&dl)[NTP], int tp, int pcx, f32x4(&mn)[2], f32x4(&cr)[2], float floor_, float* optr, int ld) {
      const int t = pcx >> 2, r = pcx & 3, p = r >> 1;
      const float v = fmaxf(fmaf(cr[t][r], INV, mn[t][r]), floor_);
      mn[t][r] = v;
      if (!(r & 1)) return;
      const float v0 = mn[t][r - 1], v1 = v;
      int hi, lo;
      asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(v0), "v"(v1));
      const float s0 = v0 * H16_LO_SCALE, s1 = v1 * H16_LO_SCALE, sc = H16_LO_SCALE;
      asm("v_fma_mixlo_f16 %0, -%1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(sc), "v"(s0));
      asm("v_fma_mixhi_f16 %0, -%1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(sc), "v"(s1));
      tc_i32x4 wh = __builtin_bit_cast(tc_i32x4, dh[tp]), wl = __builtin_bit_cast(tc_i32x4, dl[tp]);
      wh[2 * t + p] = hi; wl[2 * t + p] = lo;
      dh[tp] = __builtin_bit_cast(f16x8, wh); dl[tp] = __builtin_bit_cast(f16x8, wl);
      if (r != 3) return;
      *(f32x4*)(stage_w + 32 * (tp & 3) + 16 * t) = mn[t];
      if (t == 1 && (tp & 3) == 3) {               // 128 features of 16 rows complete: out they go
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const f32x4 w = *(const f32x4*)(stage_r + 2 * i * TC_STG_ROW);
#ifdef PNRF_TC_PROBE_NOSTORE
          if (2 * i < rows_left && w[0] == 123.456f)
#else
          if (2 * i < rows_left)
#endif
            tc_store(optr + (int64_t)(2 * i) * ld + 32 * (tp & ~3), w);
        }
      }
    };
    // one layer: KSc = its k-steps (2: from the input columns; 10: input columns, then the hidden planes; 8: hidden planes)
    auto layer = [&](auto ksc, auto posc, f16x8(&ih)[NTP], f16x8(&il)[NTP], f16x8(&oh)[NTP], f16x8(&ol)[NTP], int l, float floor_prev, float floor_) {
      constexpr int KS = decltype(ksc)::value;
      f32x4 nm[2], nc[2];
      const int ld_cur = a.ldo[l], ld_pre = ld_prev;
      float* const o_cur = a.out[l] + row_f * ld_cur + 4 * (lane & 31);
      float* const o_pre = o_prev;
      layer_h16x2<KS, NTP, decltype(posc)::value, TC_QUEUE>(
          st, ringlane, biaslane + l * W_HID,
          [&](int ks, int pl) {
            if constexpr (KS == 2) return pl == 0 ? Gh[ks & 1] : Gl[ks & 1];
            else if constexpr (KS == 10) return ks < 2 ? (pl == 0 ? Gh[ks & 1] : Gl[ks & 1]) : (pl == 0 ? ih[(ks - 2) & 7] : il[(ks - 2) & 7]);
            else return pl == 0 ? ih[ks] : il[ks];
          },
          [&](int tp, int pcx, f32x4(&mn)[2], f32x4(&cr)[2]) { piece(oh, ol, tp, pcx, mn, cr, floor_, o_cur, ld_cur); },
          [&](int pcx) { if constexpr (KS != 2) piece(ih, il, NTP - 1, pcx, pm, pc, floor_prev, o_pre, ld_pre); }, nm, nc);
#pragma unroll
      for (int t = 0; t < 2; ++t) { pm[t] = nm[t]; pc[t] = nc[t]; }
      o_prev = o_cur; ld_prev = ld_cur;
    };
    const float NEG = -__builtin_inff();
#define POS(l) std::integral_constant<int, tc_pos(l)>{}
    layer(std::integral_constant<int, 2>{}, POS(0), Yh, Yl, Xh, Xl, 0, 0.f, 0.f);                       // pts0: input columns -> X
    using K8 = std::integral_constant<int, 8>;
    static_assert(tc_pos(1) == tc_pos(3) && tc_pos(2) == tc_pos(4), "pts1 .. pts4 as a loop of two layers");
#ifdef PNRF_TC_STRAIGHT
    layer(K8{}, POS(1), Xh, Xl, Yh, Yl, 1, 0.f, 0.f);
    layer(K8{}, POS(2), Yh, Yl, Xh, Xl, 2, 0.f, 0.f);
    layer(K8{}, POS(3), Xh, Xl, Yh, Yl, 3, 0.f, 0.f);
    layer(K8{}, POS(4), Yh, Yl, Xh, Xl, 4, 0.f, 0.f);
    layer(std::integral_constant<int, 10>{}, POS(5), Xh, Xl, Yh, Yl, 5, 0.f, 0.f);
    layer(K8{}, POS(6), Yh, Yl, Xh, Xl, 6, 0.f, 0.f);
    layer(K8{}, POS(7), Xh, Xl, Yh, Yl, 7, 0.f, 0.f);
    layer(K8{}, POS(8), Yh, Yl, Xh, Xl, 8, 0.f, NEG);
#else
#pragma nounroll
    for (int p = 0; p < 2; ++p) {                                                               // pts1 .. pts4
      layer(K8{}, POS(1), Xh, Xl, Yh, Yl, 2 * p + 1, 0.f, 0.f);
      layer(K8{}, POS(2), Yh, Yl, Xh, Xl, 2 * p + 2, 0.f, 0.f);
    }
    layer(std::integral_constant<int, 10>{}, POS(5), Xh, Xl, Yh, Yl, 5, 0.f, 0.f);                      // the skip layer: [input columns | pts4's planes]
    layer(K8{}, POS(6), Yh, Yl, Xh, Xl, 6, 0.f, 0.f);
    layer(K8{}, POS(7), Xh, Xl, Yh, Yl, 7, 0.f, 0.f);
    layer(K8{}, POS(8), Yh, Yl, Xh, Xl, 8, 0.f, NEG);                                                   // feature_linear has no activation
#endif
    // the last layer's last tile pair (its planes go to Y, which nobody reads)
#pragma unroll
    for (int pcx = 0; pcx < 8; ++pcx) piece(Yh, Yl, NTP - 1, pcx, pm, pc, NEG, o_prev, ld_prev);
  }
#undef POS
  st.drain();
}
