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
// are (tchain_pack_body, a block range of iter_prepare_kernel): fragments (tile pair, k-step, tile of the pair, plane) of 16 output rows x 32 k, layer after layer, every layer a
// whole number of ring revolutions so that all of them start at ring position 0.
#pragma once

constexpr int TC_NL = 10;                                  // layers with saved outputs: pts0 .. pts7, feature, views
constexpr int TC_NSL = 11;                                 // stream layers: those and rgb_linear; alpha_linear rides as a ninth tile pair of the feature layer
__host__ __device__ constexpr int tc_ks(int l) { return l == 0 ? 2 : (l == 5 ? 10 : (l == 9 ? 9 : (l == 10 ? 4 : 8))); }   // k-steps of 32: 63 (+1) | 256 | 64 + 256 | 256 + 27 (+5) | 128 inputs
__host__ __device__ constexpr int tc_ntp(int l) { return l == 8 ? 9 : (l == 9 ? 4 : (l == 10 ? 1 : 8)); }   // tile pairs of 32 outputs: 256; feature 256 + alpha; views 128; rgb 3
__host__ __device__ constexpr int tc_frags(int l) { return tc_ntp(l) * tc_ks(l) * 4; }
__host__ __device__ constexpr int tc_frag0(int l) { int s = 0; for (int i = 0; i < l; ++i) s += tc_frags(i); return s; }
constexpr int TC_NFRAGS = tc_frag0(TC_NSL);                // 2368
constexpr int TC_PAD_SLOTS = (NSLOTS - (TC_NFRAGS / SLOT_FRAGS) % NSLOTS) % NSLOTS;            // 0: a batch is a whole number of ring revolutions
constexpr int TC_NSLOTS = TC_NFRAGS / SLOT_FRAGS + TC_PAD_SLOTS;                               // 148 slots of 16 KiB
constexpr int TC_ROWS = 128;                               // rows per workgroup and batch
#ifndef PNRF_TC_QUEUE
#define PNRF_TC_QUEUE 8
#endif
constexpr int TC_QUEUE = PNRF_TC_QUEUE;                    // A fragments held in registers ahead of their MFMAs
constexpr int TC_RING = NSLOTS;                           // (a 7-of-8 ring, WStream<8, 7, 8>, measured no faster: the stores' cost is not their vmcnt)
constexpr int TC_RING_BYTES = TC_RING * SLOT_BYTES;
#ifndef PNRF_TC_GROUP_PAIRS
#define PNRF_TC_GROUP_PAIRS 4
#endif
constexpr int TC_GROUP_PAIRS = PNRF_TC_GROUP_PAIRS;        // forward chain: tile pairs per staged group (4: 512-byte row segments, 2: 256-byte)
constexpr int TC_STG_ROW = 128 + 4;                        // floats per staged row: 128 features + padding (bank spread of the 16-byte accesses)
constexpr int TC_STG_BYTES = 8 * 16 * TC_STG_ROW * 4;      // eight waves x 16 rows
constexpr int TC_BIAS_LD = W_HID + 32;                     // bias table [stream layer][288]: entry 256 of the feature layer is alpha_linear's bias
constexpr int TC_BIAS_BYTES = TC_NSL * TC_BIAS_LD * 4;
constexpr int TC_LDS_BYTES = TC_RING_BYTES + TC_BIAS_BYTES + TC_STG_BYTES;
__host__ __device__ constexpr int tc_pos(int l) { return (tc_frag0(l) / SLOT_FRAGS) % TC_RING; }   // ring position of layer l's first slot
static_assert(tc_frags(0) % SLOT_FRAGS == 0 && tc_frags(1) % SLOT_FRAGS == 0 && tc_frags(5) % SLOT_FRAGS == 0 && tc_frags(8) % SLOT_FRAGS == 0 && tc_frags(9) % SLOT_FRAGS == 0 &&
              tc_frags(10) % SLOT_FRAGS == 0 && TC_NSLOTS % TC_RING == 0,
              "whole slots per layer, whole ring revolutions per batch");
static_assert(H16_PIECES == 8, "tchain_fwd_kernel's epilogue pieces are written for one accumulator register per piece");

typedef int tc_i32x4 __attribute__((ext_vector_type(4)));

struct TChainPackArgs {
  const float* P;                    // the trainer's flat parameters
  size_t w[TC_NSL];                  // offset of layer l's weights [out][in_dim]
  size_t w_alpha;                    // alpha_linear [1][256]
  int in_dim[TC_NSL];                // 63 | 256 | 319 (skip layer: [embedding 63 | h 256], run_nerf_helpers.py:829-831) | 283 (views: [feature 256 | view embedding 27], :842-843)
  _Float16* stream;                  // TC_NFRAGS KiB
};
// one thread per 16-byte piece of the stream (lane `lane` of fragment F)
__device__ __forceinline__ void tchain_pack_body(const TChainPackArgs& a, unsigned bid) {
  const int p = bid * blockDim.x + threadIdx.x;
  if (p >= TC_NSLOTS * SLOT_FRAGS * 64) return;
  const int F = p >> 6, lane = p & 63;
  if (F >= TC_NFRAGS) {                                      // the padding slots
    f16x8 z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = (_Float16)0.f;
    *(f16x8*)(a.stream + (size_t)p * 8) = z;
    return;
  }
  int l = 0;
  while (l + 1 < TC_NSL && F >= tc_frag0(l + 1)) ++l;
  const int f = F - tc_frag0(l), pl = f & 1, t = (f >> 1) & 1, KS = tc_ks(l), ks = (f >> 2) % KS, tp = (f >> 2) / KS;
  const int out = 32 * tp + 16 * t + (lane & 15), g = lane >> 4;
  // rows past a layer's outputs are zero: the feature layer's ninth tile pair holds alpha_linear in its first row, rgb_linear has 3 rows
  const bool row_ok = l == 8 ? out <= 256 : (l == 10 ? out < 3 : true);
  const float* W = l == 8 && out == 256 ? a.P + a.w_alpha : a.P + a.w[l] + (size_t)out * a.in_dim[l];
  f16x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    int in;
    if (l == 0 || (l == 5 && ks < 2)) {                    // the 64 input columns in their natural order; column 63 is padding
      in = 32 * ks + 8 * g + j;
      if (in >= 63) in = -1;
    } else if (l == 9 && ks == 8) {                        // the view embedding: 27 columns + padding, natural order, behind the 256 feature columns
      in = 8 * g + j < 27 ? 256 + 8 * g + j : -1;
    } else {
      in = hidden_feat_h16(l == 5 ? ks - 2 : ks, g, j) + (l == 5 ? 63 : 0);
    }
    const float x = in >= 0 && row_ok ? W[in] : 0.f;
    const _Float16 h = (_Float16)x;
    v[j] = pl ? (_Float16)((x - (float)h) * H16_LO_SCALE) : h;
  }
  *(f16x8*)(a.stream + (size_t)p * 8) = v;
}

// Store of saved activations: 2.4 GB per launch at 262 144 rows that nobody reads before the backward pass.  Left in the L2 (plain stores) they
// turn it over every ~13 us and evict the 2.2 MB weight stream between two batches of a workgroup (FETCH_SIZE: 0.3 GB per launch for 67 MB of
// input).  Measured per launch at 262 144 rows: plain 954 us, sc1 945, sc0 sc1 939, nt 802 (no stores at all: 695); sc1 nt / sc0 nt / sc0 sc1 nt: as nt.
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
#elif PNRF_TC_STORE_MODE == 4        // (asm forms for timing the remaining policies: the s_nop covers the store-data hazard the compiler cannot see)
  asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p), "v"(v));
#elif PNRF_TC_STORE_MODE == 5
  asm volatile("global_store_dwordx4 %0, %1, off sc0 nt\n\ts_nop 1" ::"v"(p), "v"(v));
#elif PNRF_TC_STORE_MODE == 6
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_nop 1" ::"v"(p), "v"(v));
#else
  __builtin_nontemporal_store(v, (f32x4*)p);       // (the compiler's own store: as inline asm it ran, but its data-register hazards were nobody's business)
#endif
}

struct TChainArgs {
  const void* blob;                  // TC_NSLOTS slots
  const float* bias[TC_NSL];         // [10]: rgb_linear's
  const float* bias_alpha;
  float* raw;                        // [n rounded up to TC_ROWS][4]: rgb (pre-sigmoid) and alpha (pre-activation) of every row (run_nerf_helpers.py:838-846)
  const float* X0; int ldx0;         // [n][ldx0] fp32, 16-byte aligned rows: columns 0 .. 63 = position embedding (63) + one zero
  const float* XV; int ldxv;         // [n][ldxv]: columns 0 .. 31 = view embedding (27) + zeros
  float* out[TC_NSL]; int ldo[TC_NSL]; // saved activations ([10]: unused) [n rounded up to TC_ROWS][ldo], 16-byte aligned rows: whole batches are written, no row predicate —
                                     // a predicated store is a branch, and eight of them in a row serialise the flush of a staged group
  uint2* mask;                       // [batch][wave][layer][lane]: two words (tile pairs 0-3 | 4-7), bit 31 - (8 (tp & 3) + 4 t + r) = (activation of the lane's row, feature 32 tp + 16 t
                                     // + 4 g + r) > 0 — what
                                     // tchain_bwd_kernel needs of the saved activations (32 bytes per row and layer instead of 1 KiB)
  int64_t n; int nbatch;
};

__global__ __launch_bounds__(512, 2) void tchain_fwd_kernel(TChainArgs a) {
  constexpr int NW = 8, NTP = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* bias_lds = (float*)(smem + TC_RING_BYTES);        // [layer][256]
  for (int i = threadIdx.x; i < TC_NSL * TC_BIAS_LD; i += 512) {
    const int l = i / TC_BIAS_LD, c = i - l * TC_BIAS_LD;
    const int n_out = l == 9 ? 128 : (l == 10 ? 3 : 256);
    bias_lds[i] = c < n_out ? a.bias[l][c] : (l == 8 && c == 256 ? a.bias_alpha[0] : 0.f);
  }
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
    f16x8 Gh[2], Gl[2];                    // the 64 input columns: k-step ks, group g = columns 32 ks + 8 g .. + 7.  Fetched for pts0 and again
    auto load_inputs = [&]() {             // for the skip layer: 16 registers that pts1 .. pts4 do not have to carry
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
    };
    load_inputs();
    f16x8 Xh[NTP], Xl[NTP], Yh[NTP], Yl[NTP];      // activations ping-pong between X and Y: per 32-feature k-step one hi and one lo plane
    f32x4 pm[2], pc[2];                            // pending (deferred) tile pair of the previous layer
    // Saved activations leave through a per-wave staging tile in LDS: four tile pairs (128 features) of the wave's 16 rows are collected, then
    // written out as whole 512-byte row segments (one store instruction = 2 rows x 512 B).  Stored straight from the accumulators — lane
    // (column, g) holds 4 features of ONE row, an instruction covers 16 rows x 64 B — the same bytes reached HBM in 128-byte pieces at a 1 KiB
    // stride and the launch ran at 2.3 TB/s of writes (1.06 ms at 262 144 rows; 0.65 ms without the stores; a plain fill writes 6.8 TB/s).
    float* const stage = (float*)(smem + TC_RING_BYTES + TC_BIAS_BYTES) + wave * (16 * TC_STG_ROW);
    float* const stage_w = stage + col * TC_STG_ROW + 4 * g;                       // this lane's accumulator tiles go here (+ 32 (tp & 3) + 16 t)
    constexpr int GP = TC_GROUP_PAIRS, LPR = 8 * GP, RPI = 64 / LPR;                 // tile pairs per staged group; lanes per row, rows per store instruction
    const float* const stage_r = stage + (lane / LPR) * TC_STG_ROW + 4 * (lane % LPR);   // ... and it reads rows RPI i + lane / LPR, 16 bytes at 4 (lane % LPR)
    const int64_t row_f = (int64_t)batch * TC_ROWS + wave * 16 + lane / LPR;      // first of the rows this lane writes out
    float* o_prev = a.out[0];                      // where the pending tile pair's rows go (its layer's buffer; + this lane's 32-bit byte offset) ...
    int ld_prev = 0;
    uint32_t off_prev = 0;
    uint2* const m_lane = a.mask + ((int64_t)batch * 8 + wave) * TC_NL * 64 + lane;      // ReLU masks of this lane: + 64 l
    int l_prev = 0;                                // ... and its layer (for the mask)
    float alpha_v = 0.f;                           // alpha_linear's output of this lane's row (lanes of group 0)
    uint32_t mb0 = 0, mb1 = 0;                     // mask bits of the layer whose pieces are running (tile pairs 0-3 | 4-7)

    // piece pcx = accumulator register pcx & 3 of tile pcx >> 2 of a tile pair: bias is in the accumulator; combine, activate (floor = 0: ReLU,
    // -inf: none), park the value in the accumulator register; the odd piece packs the pair into the next layer's planes (as sampler_h16_kernel),
    // the last piece of a tile stages its four features — registers 0 .. 3 of lane (column, g) are features 16 T + 4 g .. + 3 of the row
    auto piece = [&](f16x8(&dh)[NTP], f16x8(&dl)[NTP], int tp, int pcx, f32x4(&mn)[2], f32x4(&cr)[2], float floor_, float* optr, uint32_t ooff, int ld, int lm, int ltp = NTP - 1) {
      const int t = pcx >> 2, r = pcx & 3, p = r >> 1;
      const float v = fmaxf(fmaf(cr[t][r], INV, mn[t][r]), floor_);
      mn[t][r] = v;
      // mask word <- 2 word + (v > 0): the compare's carry goes in through v_addc.  32 pieces fill a word: piece (tp, pcx) is bit 31 - (8 (tp & 3) + pcx)
      if (tp < 4) asm("v_cmp_lt_f32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mb0) : "v"(v) : "vcc");
      else asm("v_cmp_lt_f32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mb1) : "v"(v) : "vcc");
      if (tp == ltp && pcx == 7) {                   // the layer's last activation of this lane
        *(uint2*)(m_lane + 64 * lm) = make_uint2(mb0, mb1);
        mb0 = 0; mb1 = 0;
      }
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
      *(f32x4*)(stage_w + 32 * (tp % GP) + 16 * t) = mn[t];
      if (t == 1 && tp % GP == GP - 1) {           // 32 GP features of 16 rows complete: out they go
#pragma unroll
        for (int i = 0; i < 16 / RPI; ++i) {
          const f32x4 w = *(const f32x4*)(stage_r + RPI * i * TC_STG_ROW);
#ifdef PNRF_TC_PROBE_NOSTORE
          if (w[0] == 123.456f)
#endif
            tc_store((float*)((char*)(optr + ((RPI * i) * ld + 32 * (tp - tp % GP))) + ooff), w);       // uniform pointer + this lane's 32-bit offset
        }
      }
    };
    // one layer: KSc = its k-steps (2: from the input columns; 10: input columns, then the hidden planes; 8: hidden planes)
    // LNTP: the layer's tile pairs (9: feature + the alpha pair).  PRE: what its first pieces finish — 0 nothing, 1 the previous layer's last
    // pair (index PTP), 2 the alpha pair (one value per row: kept in alpha_v)
    auto layer = [&](auto ksc, auto posc, auto ntpc, auto prec, auto ptpc, f16x8(&ih)[NTP], f16x8(&il)[NTP], f16x8(&oh)[NTP], f16x8(&ol)[NTP], int l,
                     float floor_prev, float floor_) {
      constexpr int KS = decltype(ksc)::value, LNTP = decltype(ntpc)::value, PRE = decltype(prec)::value, PTP = decltype(ptpc)::value;
      constexpr int LTP = LNTP == 9 ? 7 : LNTP - 1;  // last pair of the layer that is an activation of the net
      f32x4 nm[2], nc[2];
      const int ld_cur = a.ldo[l], ld_pre = ld_prev, l_pre = l_prev;
      float* const o_cur = a.out[l];
      float* const o_pre = o_prev;
      const uint32_t off_cur = ((uint32_t)row_f * (uint32_t)ld_cur + 4u * (lane % LPR)) * 4u, off_pre = off_prev;   // < 4 GiB: checked by the launcher
      layer_h16x2<KS, LNTP, decltype(posc)::value, TC_QUEUE>(
          st, ringlane, biaslane + l * TC_BIAS_LD,
          [&](int ks, int pl) {
            if constexpr (KS == 2) return pl == 0 ? Gh[ks & 1] : Gl[ks & 1];
            else if constexpr (KS == 10) return ks < 2 ? (pl == 0 ? Gh[ks & 1] : Gl[ks & 1]) : (pl == 0 ? ih[(ks - 2) & 7] : il[(ks - 2) & 7]);
            else if constexpr (KS == 9) return ks < 8 ? (pl == 0 ? ih[ks & 7] : il[ks & 7]) : (pl == 0 ? Gh[0] : Gl[0]);      // views: G holds the view embedding
            else if constexpr (KS == 4) return pl == 0 ? ih[ks & 3] : il[ks & 3];
            else return pl == 0 ? ih[ks] : il[ks];
          },
          [&](int tp, int pcx, f32x4(&mn)[2], f32x4(&cr)[2]) { piece(oh, ol, tp, pcx, mn, cr, floor_, o_cur, off_cur, ld_cur, l, LTP); },
          [&](int pcx) {
            if constexpr (PRE == 1) piece(ih, il, PTP, pcx, pm, pc, floor_prev, o_pre, off_pre, ld_pre, l_pre, PTP);
            else if constexpr (PRE == 2) { if (pcx == 0) alpha_v = fmaf(pc[0][0], INV, pm[0][0]); }
          }, nm, nc);
#pragma unroll
      for (int t = 0; t < 2; ++t) { pm[t] = nm[t]; pc[t] = nc[t]; }
      o_prev = o_cur; ld_prev = ld_cur; off_prev = off_cur; l_prev = l;
    };
    const float NEG = -__builtin_inff();
#define POS(l) std::integral_constant<int, tc_pos(l)>{}
    using K8 = std::integral_constant<int, 8>;
    using N8 = std::integral_constant<int, 8>;
    using Pre0 = std::integral_constant<int, 0>;
    using Pre1 = std::integral_constant<int, 1>;
    using P7 = std::integral_constant<int, 7>;
    layer(std::integral_constant<int, 2>{}, POS(0), N8{}, Pre0{}, P7{}, Yh, Yl, Xh, Xl, 0, 0.f, 0.f);   // pts0: input columns -> X
    static_assert(tc_pos(1) == tc_pos(3) && tc_pos(2) == tc_pos(4), "pts1 .. pts4 as a loop of two layers");
#pragma nounroll
    for (int p = 0; p < 2; ++p) {                                                               // pts1 .. pts4
      layer(K8{}, POS(1), N8{}, Pre1{}, P7{}, Xh, Xl, Yh, Yl, 2 * p + 1, 0.f, 0.f);
      layer(K8{}, POS(2), N8{}, Pre1{}, P7{}, Yh, Yl, Xh, Xl, 2 * p + 2, 0.f, 0.f);
    }
    load_inputs();
    layer(std::integral_constant<int, 10>{}, POS(5), N8{}, Pre1{}, P7{}, Xh, Xl, Yh, Yl, 5, 0.f, 0.f);  // the skip layer: [input columns | pts4's planes]
    layer(K8{}, POS(6), N8{}, Pre1{}, P7{}, Yh, Yl, Xh, Xl, 6, 0.f, 0.f);
    layer(K8{}, POS(7), N8{}, Pre1{}, P7{}, Xh, Xl, Yh, Yl, 7, 0.f, 0.f);
    // feature_linear (no activation) with alpha_linear as the first row of a ninth tile pair: both read pts7's activations
    layer(K8{}, POS(8), std::integral_constant<int, 9>{}, Pre1{}, P7{}, Yh, Yl, Xh, Xl, 8, 0.f, NEG);
    {   // views_linear reads [feature | view embedding]: the embedding's k-step comes from the rows the rgb branch's weight gradient reads
      const float4* x = (const float4*)(a.XV + rr * a.ldxv + 8 * g);
      const float4 lo = x[0], hi = x[1];
      const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const _Float16 h = (_Float16)v[j];
        Gh[0][j] = h;
        Gl[0][j] = (_Float16)((v[j] - (float)h) * H16_LO_SCALE);
      }
    }
    layer(std::integral_constant<int, 9>{}, POS(9), std::integral_constant<int, 4>{}, std::integral_constant<int, 2>{}, P7{}, Xh, Xl, Yh, Yl, 9, NEG, 0.f);   // views_linear (ReLU)
    // rgb_linear: 4 k-steps of views' activations, one tile pair whose first three rows are r, g, b.  Its fourth k-step reads views' last tile
    // pair: finished here, not among rgb_linear's own MFMAs (a layer of fewer than 8 k-steps runs its deferred pieces behind its last k-step)
#pragma unroll
    for (int pcx = 0; pcx < 8; ++pcx) piece(Yh, Yl, 3, pcx, pm, pc, 0.f, o_prev, off_prev, ld_prev, l_prev, 3);
    layer(std::integral_constant<int, 4>{}, POS(10), std::integral_constant<int, 1>{}, Pre0{}, P7{}, Yh, Yl, Xh, Xl, 10, 0.f, NEG);
    if (g == 0 && valid)
      *(float4*)(a.raw + row * 4) = make_float4(fmaf(pc[0][0], INV, pm[0][0]), fmaf(pc[0][1], INV, pm[0][1]), fmaf(pc[0][2], INV, pm[0][2]), alpha_v);
#pragma unroll
    for (int i = 0; i < TC_PAD_SLOTS; ++i) st.begin();
  }
#undef POS
  st.drain();
}

// ------------------------------------------------------------------------------------------ backward: the input-gradient chain
// dZ_k = (dZ_{k+1} W_{k+1}) * relu'(a_k) from the rgb branch's hidden gradient down to pts0, one launch on the same engine: a workgroup's 128 rows
// of gradient stay in registers (MFMA B operand, split fp16), the TRANSPOSED weights stream through LDS, every gradient the weight-gradient
// products need is written once (fp32; those products follow as one grouped launch) and nothing of the saved activations is read but their
// sign bits (tchain_fwd_kernel's masks).  Stream layers (fragments as in the forward stream):
//   s0  views_linear^T, feature columns: 4 k-steps (128 hidden gradients of the rgb branch) -> d feature          s5  pts4^T -> dZ3
//   s1  [feature_linear^T ; alpha_linear^T]: 9 k-steps (d feature + the alpha gradient) -> dZ7                   s6  pts3^T -> dZ2
//   s2  pts7^T -> dZ6                                                                                           s7  pts2^T -> dZ1
//   s3  pts6^T -> dZ5                                                                                           s8  pts1^T -> dZ0
//   s4  pts5^T: 8 tile pairs of hidden columns -> dZ4, then 2 pairs of embedding columns -> d embedding          s9  pts0^T: 2 pairs -> d embedding
// Gradient range.  Gradients are 1e-9 .. 1e-3; the planes carry them times a power of two PER ROW.  The first comes from the row's own maximum
// (max |x| s in [2^13, 2^14)); from layer to layer the accumulators hold c' = (a s) W and |c'_j| <= |a s|_2 |W_:j|_2 <= |a s|_2 C with C^2 the
// layer's largest column sum of squares (tchain_norms_body), so t = the power of two with |a s|_2 C t < 2^14 is known — before the first
// output is split — from the sum of squares of the row's planes, which the previous layer's epilogue accumulated; the next planes are c' t,
// never above 2^14.  All factors are powers of two: undone exactly in the epilogue that stores the gradient.
constexpr int TB_NS = 10;
__host__ __device__ constexpr int tb_ks(int s) { return s == 0 ? 4 : (s == 1 ? 9 : 8); }
__host__ __device__ constexpr int tb_ntp(int s) { return s == 4 ? 10 : (s == 9 ? 2 : 8); }
__host__ __device__ constexpr int tb_frags(int s) { return tb_ntp(s) * tb_ks(s) * 4; }
__host__ __device__ constexpr int tb_frag0(int s) { int f = 0; for (int i = 0; i < s; ++i) f += tb_frags(i); return f; }
constexpr int TB_NFRAGS = tb_frag0(TB_NS);                                         // 2336
constexpr int TB_PAD_SLOTS = (NSLOTS - (TB_NFRAGS / SLOT_FRAGS) % NSLOTS) % NSLOTS;  // 2: a batch is a whole number of ring revolutions
constexpr int TB_NSLOTS = TB_NFRAGS / SLOT_FRAGS + TB_PAD_SLOTS;                   // 148
__host__ __device__ constexpr int tb_pos(int s) { return (tb_frag0(s) / SLOT_FRAGS) % NSLOTS; }
static_assert(TB_NFRAGS % SLOT_FRAGS == 0 && tb_frags(0) % SLOT_FRAGS == 0 && tb_frags(1) % SLOT_FRAGS == 0 && tb_frags(4) % SLOT_FRAGS == 0 &&
              tb_frags(9) % SLOT_FRAGS == 0, "whole slots per layer");
constexpr int TB_NSLOT_MAX = 10;                                                   // gradient buffers with a max-|.| slot: dZ0 .. dZ7, d feature, scratch
constexpr int TB_LDS_BYTES = TC_RING_BYTES + TC_STG_BYTES + 8 * 16 * 4;

struct TChainBwdPackArgs {
  const float* P;
  size_t w[TC_NL];                   // weights of pts0 .. pts7, feature, views (as TChainPackArgs)
  size_t w_alpha;                    // alpha_linear [1][256]
  _Float16* stream;                  // TB_NSLOTS slots
  float* cmax;                       // [TB_NS] largest column sum of squares per stream layer (of the columns whose outputs are split again)
};
// element (k, j) of stream layer s: weight that output gradient k contributes to input gradient j (0 where the layout pads)
__device__ __forceinline__ float tb_weight(const TChainBwdPackArgs& a, int s, int k, int j) {
  if (s == 0) return k < 128 ? a.P[a.w[9] + (size_t)k * 283 + j] : 0.f;           // views_linear [128][283]: feature columns 0 .. 255
  if (s == 1) return k < 256 ? a.P[a.w[8] + (size_t)k * 256 + j] : (k == 256 ? a.P[a.w_alpha + j] : 0.f);
  const int L = 9 - s;                                       // pts7 .. pts0
  if (L == 5) {                                              // [256][319]: embedding columns 0 .. 62, hidden columns 63 ..
    if (j < 256) return a.P[a.w[5] + (size_t)k * 319 + 63 + j];
    return j - 256 < 63 ? a.P[a.w[5] + (size_t)k * 319 + (j - 256)] : 0.f;
  }
  if (L == 0) return j < 63 ? a.P[a.w[0] + (size_t)k * 63 + j] : 0.f;
  return a.P[a.w[L] + (size_t)k * 256 + j];
}
__device__ __forceinline__ void tchain_pack_bwd_body(const TChainBwdPackArgs& a, unsigned bid) {
  const int p = bid * blockDim.x + threadIdx.x;
  if (p >= TB_NSLOTS * SLOT_FRAGS * 64) return;
  const int F = p >> 6, lane = p & 63;
  f16x8 v;
  if (F >= TB_NFRAGS) {                                      // the padding slots
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (_Float16)0.f;
    *(f16x8*)(a.stream + (size_t)p * 8) = v;
    return;
  }
  int s = 0;
  while (s + 1 < TB_NS && F >= tb_frag0(s + 1)) ++s;
  const int f = F - tb_frag0(s), pl = f & 1, t = (f >> 1) & 1, KS = tb_ks(s), ks = (f >> 2) % KS, tp = (f >> 2) / KS;
  const int out = 32 * tp + 16 * t + (lane & 15), g = lane >> 4;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    // s0 reads its gradients from HBM in their natural order; s1's ninth k-step is the alpha gradient (element 0 of group 0)
    const int k = s == 0 ? 32 * ks + 8 * g + j : (s == 1 && ks == 8 ? 256 + 8 * g + j : hidden_feat_h16(ks, g, j));
    const float x = tb_weight(a, s, k, out);
    const _Float16 h = (_Float16)x;
    v[j] = pl ? (_Float16)((x - (float)h) * H16_LO_SCALE) : h;
  }
  *(f16x8*)(a.stream + (size_t)p * 8) = v;
}
// cmax[s] = max over the re-split output columns j of sum_k w(k, j)^2.  Block (s, b) = index 16 s + b: columns 16 b .. 16 b + 15, 16 partial sums
// each, atomicMax of the block's maximum into cmax[s].  The blocks run beside the packing blocks in ONE launch (iter_prepare_kernel), so nobody
// can zero cmax ahead of them in that launch: the trainer alternates between two cmax arrays — this launch accumulates into one (zero since the
// launch before) and clears the other (`zero`, which the chains of the iterations in between have finished reading).  red: 256 floats of LDS.
__device__ __forceinline__ void tchain_norms_body(const TChainBwdPackArgs& a, int blk, float* red, float* zero) {
  const int s = blk >> 4, j = 16 * (blk & 15) + (threadIdx.x & 15), part = threadIdx.x >> 4;
  if (blk == 0 && threadIdx.x < 16) zero[threadIdx.x] = 0.f;
  float acc = 0.f;
  const int nk = s == 0 ? 128 : (s == 1 ? 257 : 256);
  if (s != 9)
    for (int k = part; k < nk; k += 16) { const float w = tb_weight(a, s, k, j); acc = fmaf(w, w, acc); }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o >= 16; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  for (int o = 8; o >= 1; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]); __syncthreads(); }
  if (threadIdx.x == 0) atomicMax((unsigned int*)a.cmax + s, __float_as_uint(red[0]));
}

struct TChainBwdArgs {
  const void* blob;                  // TB_NSLOTS slots
  const float* dH; int lddh;         // gradient of the rgb branch's hidden layer (after relu'): [n][lddh], 128 columns, 16-byte aligned rows
  const float* dA; int ldda;         // d alpha: dA[row * ldda]
  const uint2* mask;                 // tchain_fwd_kernel's ReLU masks (same batch / wave / lane geometry)
  const float* cmax;                 // [TB_NS]
  float* dz[8];                      // dZ_k [n][256], k = 0 .. 7
  float* dF; int lddf;               // d feature: columns 0 .. 255 of [n][lddf] (the views layer's input gradient; the rest of the row is not written)
  float* slot[TB_NSLOT_MAX];         // max-|.| slots (HG_SLOT floats each; pnrf_hgemm.h): [k] of dZ_k, [8] of d feature, [9] scratch
  float* dg; int lddg;               // d embedding from the skip layer: columns 0 .. 63 of [n][lddg]
  float* de0;                        // d embedding from pts0: [n][64]       (every output buffer: n rounded up to TC_ROWS rows, as in the forward pass)
  int64_t n; int nbatch;
};

__global__ __launch_bounds__(512, 2) void tchain_bwd_kernel(TChainBwdArgs a) {
  constexpr int NW = 8, NTP = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 15, g = lane >> 4;
  WStream<NW> st;
  st.init(a.blob, TB_NSLOTS, smem);
  st.prologue();
  if (__builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(1);
  const char* ringlane = smem + lane * 16;
  constexpr float INV = 1.f / H16_LO_SCALE;
  float* const stage = (float*)(smem + TC_RING_BYTES) + wave * (16 * TC_STG_ROW);
  float* const stage_w = stage + col * TC_STG_ROW + 4 * g;
  const float* const stage_r = stage + (lane >> 5) * TC_STG_ROW + 4 * (lane & 31);        // 128-feature groups: rows 2 i + (lane >> 5)
  float* const wave_max = (float*)(smem + TC_RING_BYTES + TC_STG_BYTES) + wave * 16;     // [wave][slot]: max |gradient| of the wave's rows
  if (lane < TB_NSLOT_MAX) wave_max[lane] = 0.f;
  auto pow2 = [](int k) { return __int_as_float((127 + (k < -120 ? -120 : (k > 120 ? 120 : k))) << 23); };

  for (int batch = blockIdx.x; batch < a.nbatch; batch += gridDim.x) {
    const int64_t row = (int64_t)batch * TC_ROWS + wave * 16 + col;
    const int64_t rr = row < a.n ? row : a.n - 1;
    const int64_t row_f = (int64_t)batch * TC_ROWS + wave * 16 + (lane >> 5);
    const uint2* const m_lane = a.mask + ((int64_t)batch * 8 + wave) * TC_NL * 64 + lane;
    const uint32_t off_z = ((uint32_t)row_f * 256u + 4u * (lane & 31)) * 4u;               // this lane's byte offset in a [n][256] buffer

    f16x8 Xh[NTP], Xl[NTP], Yh[NTP], Yl[NTP], Ah, Al;        // gradient planes; Ah / Al: s1's ninth k-step (the alpha gradient)
    f32x4 pm[2], pc[2];
    // per-row scale bookkeeping (all lanes of a column agree): planes = true value / inv
    float inv_in = 1.f, inv_cur, tscale = 1.f, sumsq, amax = 0.f;
    uint2 mk = make_uint2(0, 0), mk_next;
    int slot_prev = TB_NSLOT_MAX - 1;              // which buffer's maximum is being accumulated (the last: s0's inputs, nobody's business)
    const float da = a.dA[rr * a.ldda];            // the row's alpha gradient: joins s1's inputs, so s0's output scale has to hold it too
    float da_in;                                   // |da| on the scale of s0's input planes
    {   // s0's inputs: the 128 hidden gradients of the row, scaled by the row's own maximum -> Y[0 .. 3]
      const float4* x = (const float4*)(a.dH + rr * a.lddh + 8 * g);
      float v[4][8];
      float m = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const float4 lo = x[8 * ks], hi = x[8 * ks + 1];
        v[ks][0] = lo.x; v[ks][1] = lo.y; v[ks][2] = lo.z; v[ks][3] = lo.w; v[ks][4] = hi.x; v[ks][5] = hi.y; v[ks][6] = hi.z; v[ks][7] = hi.w;
#pragma unroll
        for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(v[ks][j]));
      }
      m = fmaxf(m, __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(m), 0x401F)));
      m = fmaxf(m, __int_as_float(__builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __float_as_int(m))));
      const int e = (__float_as_int(m) >> 23) - 127;             // floor(log2 m) for normal m
      const int k = m > 1e-37f && m < 1e37f ? 13 - e : 0;
      const float s0 = pow2(k);
      inv_cur = pow2(-k);
      da_in = fabsf(da) * s0;
      sumsq = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xs = v[ks][j] * s0;
          const _Float16 h = (_Float16)xs;
          Yh[ks][j] = h; Yl[ks][j] = (_Float16)((xs - (float)h) * H16_LO_SCALE);
          sumsq = fmaf(xs, xs, sumsq);
        }
      }
    }

    // a piece of a layer whose outputs are split again: scale, ReLU mask, statistics, planes, staged store of the true value ([n][ld] buffer)
    auto bpiece = [&](f16x8(&dh)[NTP], f16x8(&dl)[NTP], int tp, int pcx, f32x4(&mn)[2], f32x4(&cr)[2], float ts, float inv_o, uint2 mw, float* optr,
                      uint32_t ooff, int ld) {
      const int t = pcx >> 2, r = pcx & 3, p = r >> 1;
      float xv = fmaf(cr[t][r], INV, mn[t][r]) * ts;
      const int sel = __builtin_amdgcn_sbfe((int)(tp < 4 ? mw.x : mw.y), 31 - (8 * (tp & 3) + pcx), 1);       // 0 or -1
      xv = __int_as_float(__float_as_int(xv) & sel);
      sumsq = fmaf(xv, xv, sumsq);
      amax = fmaxf(amax, fabsf(xv));
      mn[t][r] = xv;
      if (!(r & 1)) return;
      const float v0 = mn[t][r - 1], v1 = xv;
      int hi, lo;
      asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(v0), "v"(v1));
      const float s0 = v0 * H16_LO_SCALE, s1 = v1 * H16_LO_SCALE, sc = H16_LO_SCALE;
      asm("v_fma_mixlo_f16 %0, -%1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(sc), "v"(s0));
      asm("v_fma_mixhi_f16 %0, -%1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(sc), "v"(s1));
      tc_i32x4 wh = __builtin_bit_cast(tc_i32x4, dh[tp]), wl = __builtin_bit_cast(tc_i32x4, dl[tp]);
      wh[2 * t + p] = hi; wl[2 * t + p] = lo;
      dh[tp] = __builtin_bit_cast(f16x8, wh); dl[tp] = __builtin_bit_cast(f16x8, wl);
      if (r != 3) return;
      *(f32x4*)(stage_w + 32 * (tp & 3) + 16 * t) = mn[t] * inv_o;
      if (t == 1 && (tp & 3) == 3) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const f32x4 w = *(const f32x4*)(stage_r + 2 * i * TC_STG_ROW);
          tc_store((float*)((char*)(optr + ((2 * i) * ld + 32 * (tp & ~3))) + ooff), w);
        }
      }
    };
    // a piece of a store-only tile pair (the embedding gradients): pair index q = 0, 1 of a 64-column group
    auto spiece = [&](int q, int pcx, f32x4(&mn)[2], f32x4(&cr)[2], float inv_src, float* optr, int ld) {
      const int t = pcx >> 2, r = pcx & 3;
      mn[t][r] = fmaf(cr[t][r], INV, mn[t][r]) * inv_src;
      if (r != 3) return;
      *(f32x4*)(stage_w + 32 * q + 16 * t) = mn[t];
      if (t == 1 && q == 1) {                      // 64 columns of 16 rows: a store instruction = 4 rows x 256 B (rows 4 i + (lane >> 4))
        const int64_t row_f4 = (int64_t)batch * TC_ROWS + wave * 16 + (lane >> 4);
        const float* const stage_r4 = stage + (lane >> 4) * TC_STG_ROW + 4 * (lane & 15);
        const uint32_t off4 = ((uint32_t)row_f4 * (uint32_t)ld + 4u * (lane & 15)) * 4u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const f32x4 w = *(const f32x4*)(stage_r4 + 4 * i * TC_STG_ROW);
          tc_store((float*)((char*)(optr + (int64_t)(4 * i) * ld) + off4), w);
        }
      }
    };
    // at a layer's first own piece: the previous outputs (this layer's inputs) are complete.  Their maximum goes to its slot, their sum of
    // squares fixes this layer's scale, the next mask takes over.
    auto switch_ctx = [&](int s, int slot_now, float other = 0.f) {     // other: |.| of a value (on the input planes' scale) that will join the outputs' planes
      float n2 = sumsq;
      n2 += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(n2), 0x401F));                       // lane ^ 16 (bit-mask mode: and 0x1f, xor 0x10)
      n2 += __int_as_float(__builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __float_as_int(n2)));           // lane ^ 32
      atomicMax((unsigned int*)wave_max + slot_prev, __float_as_uint(amax * inv_cur));                      // ds_max_u32: non-negative floats order like their bits
      slot_prev = slot_now;
      inv_in = inv_cur;
      const float b2 = n2 * a.cmax[s];
      const int e = (__float_as_int(b2) >> 23) - 127;
      int k = b2 > 1e-37f && b2 < 1e37f ? 14 - ((e + 2) >> 1) : 0;
      if (other > 1e-37f && other < 1e37f) {                 // other t < 2^14 as well
        const int k2 = 13 - ((__float_as_int(other) >> 23) - 127);
        k = k < k2 ? k : k2;
      }
      tscale = pow2(k);
      inv_cur = inv_in * pow2(-k);
      sumsq = 0.f; amax = 0.f;
      mk = mk_next;
    };
    // one layer whose outputs are split again: stream layer s -> gradient buffer k (8: d feature).  SP: its pending predecessor pair is a
    // store-only pair (after s4)
    auto layer = [&](auto ksc, auto posc, auto spc, f16x8(&ih)[NTP], f16x8(&il)[NTP], f16x8(&oh)[NTP], f16x8(&ol)[NTP], int s, int k) {
      constexpr int KS = decltype(ksc)::value;
      constexpr bool SP = decltype(spc)::value;
      f32x4 nm[2], nc[2];
      if (k < 8) mk_next = m_lane[64 * k]; else mk_next = make_uint2(0xffffffffu, 0xffffffffu);          // feature_linear has no activation
      float* const o_cur = k < 8 ? a.dz[k] : a.dF;
      const int ld_cur = k < 8 ? 256 : a.lddf;
      const uint32_t off_cur = k < 8 ? off_z : ((uint32_t)row_f * (uint32_t)ld_cur + 4u * (lane & 31)) * 4u;
      // the predecessor's pending pair: dZ_{k+1} (k = 7: d feature, stream layer s0's outputs)
      float* const o_pre = k == 7 ? a.dF : a.dz[k + 1 < 8 ? k + 1 : 7];
      const int ld_pre = k == 7 ? a.lddf : 256;
      const uint32_t off_pre = k == 7 ? ((uint32_t)row_f * (uint32_t)ld_pre + 4u * (lane & 31)) * 4u : off_z;
      const float ts_pre = tscale, inv_pre = inv_cur;
      const uint2 mk_pre = mk;
      layer_h16x2<KS, NTP, decltype(posc)::value, TC_QUEUE, false>(
          st, ringlane, (const float*)nullptr,
          [&](int ks, int pl) {
            if constexpr (KS == 9) return ks < 8 ? (pl == 0 ? ih[ks & 7] : il[ks & 7]) : (pl == 0 ? Ah : Al);
            else if constexpr (KS == 4) return pl == 0 ? ih[ks & 3] : il[ks & 3];
            else return pl == 0 ? ih[ks] : il[ks];
          },
          [&](int tp, int pcx, f32x4(&mn)[2], f32x4(&cr)[2]) {
            if (tp == 0 && pcx == 0) switch_ctx(s, k, KS == 4 ? da_in : 0.f);
            bpiece(oh, ol, tp, pcx, mn, cr, tscale, inv_cur, mk, o_cur, off_cur, ld_cur);
          },
          [&](int pcx) {
            if constexpr (KS == 4) return;
            else if constexpr (SP) spiece(1, pcx, pm, pc, inv_in, a.dg, a.lddg);
            else bpiece(ih, il, NTP - 1, pcx, pm, pc, ts_pre, inv_pre, mk_pre, o_pre, off_pre, ld_pre);
          }, nm, nc);
#pragma unroll
      for (int t = 0; t < 2; ++t) { pm[t] = nm[t]; pc[t] = nc[t]; }
    };
    using K8 = std::integral_constant<int, 8>;
    using NoSp = std::false_type;
#define TBPOS(s) std::integral_constant<int, tb_pos(s)>{}
    layer(std::integral_constant<int, 4>{}, TBPOS(0), NoSp{}, Yh, Yl, Xh, Xl, 0, 8);          // views^T : Y[0 .. 3] -> X = d feature
    {   // the alpha gradient joins as a ninth k-step, on the scale the feature gradient's planes got
      const float sc = __int_as_float(0x7f000000 - __float_as_int(inv_cur));                 // 1 / inv_cur (a power of two)
      const float xs = g == 0 ? da * sc : 0.f;
      const _Float16 h = (_Float16)xs;
      const _Float16 z = (_Float16)0.f;
      Ah = f16x8{h, z, z, z, z, z, z, z};
      Al = f16x8{(_Float16)((xs - (float)h) * H16_LO_SCALE), z, z, z, z, z, z, z};
      sumsq = fmaf(xs, xs, sumsq);
    }
    layer(std::integral_constant<int, 9>{}, TBPOS(1), NoSp{}, Xh, Xl, Yh, Yl, 1, 7);          // [feature ; alpha]^T : X (+ A) -> Y = dZ7
    layer(K8{}, TBPOS(2), NoSp{}, Yh, Yl, Xh, Xl, 2, 6);                                      // pts7^T -> X = dZ6
    layer(K8{}, TBPOS(3), NoSp{}, Xh, Xl, Yh, Yl, 3, 5);                                      // pts6^T -> Y = dZ5
    {   // s4 = pts5^T: tile pairs 0 .. 7 the hidden columns -> X = dZ4, pairs 8, 9 the embedding columns -> stored only
      f32x4 nm[2], nc[2];
      mk_next = m_lane[64 * 4];
      const float ts_pre = tscale, inv_pre = inv_cur;
      const uint2 mk_pre = mk;
      layer_h16x2<8, 10, tb_pos(4), TC_QUEUE, false>(
          st, ringlane, (const float*)nullptr, [&](int ks, int pl) { return pl == 0 ? Yh[ks] : Yl[ks]; },
          [&](int tp, int pcx, f32x4(&mn)[2], f32x4(&cr)[2]) {
            if (tp == 0 && pcx == 0) switch_ctx(4, 4);
            if (tp < 8) bpiece(Xh, Xl, tp & 7, pcx, mn, cr, tscale, inv_cur, mk, a.dz[4], off_z, 256);
            else spiece(0, pcx, mn, cr, inv_in, a.dg, a.lddg);                                // (tp = 8: the first embedding pair)
          },
          [&](int pcx) { bpiece(Yh, Yl, NTP - 1, pcx, pm, pc, ts_pre, inv_pre, mk_pre, a.dz[5], off_z, 256); }, nm, nc);
#pragma unroll
      for (int t = 0; t < 2; ++t) { pm[t] = nm[t]; pc[t] = nc[t]; }
    }
    layer(K8{}, TBPOS(5), std::true_type{}, Xh, Xl, Yh, Yl, 5, 3);                            // pts4^T -> Y = dZ3 (its first pieces finish the embedding pair)
    layer(K8{}, TBPOS(6), NoSp{}, Yh, Yl, Xh, Xl, 6, 2);                                      // pts3^T -> X = dZ2
    layer(K8{}, TBPOS(7), NoSp{}, Xh, Xl, Yh, Yl, 7, 1);                                      // pts2^T -> Y = dZ1
    layer(K8{}, TBPOS(8), NoSp{}, Yh, Yl, Xh, Xl, 8, 0);                                      // pts1^T -> X = dZ0
    {   // s9 = pts0^T: two tile pairs -> d embedding [n][64], stored only
      f32x4 nm[2], nc[2];
      const float ts_pre = tscale, inv_pre = inv_cur;
      const uint2 mk_pre = mk;
      layer_h16x2<8, 2, tb_pos(9), TC_QUEUE, false>(
          st, ringlane, (const float*)nullptr, [&](int ks, int pl) { return pl == 0 ? Xh[ks] : Xl[ks]; },
          [&](int, int pcx, f32x4(&mn)[2], f32x4(&cr)[2]) { spiece(0, pcx, mn, cr, inv_pre, a.de0, 64); },
          [&](int pcx) { bpiece(Xh, Xl, NTP - 1, pcx, pm, pc, ts_pre, inv_pre, mk_pre, a.dz[0], off_z, 256); }, nm, nc);
#pragma unroll
      for (int pcx = 0; pcx < 8; ++pcx) spiece(1, pcx, nm, nc, inv_pre, a.de0, 64);
    }
#undef TBPOS
#pragma unroll
    for (int i = 0; i < TB_PAD_SLOTS; ++i) st.begin();
    {   // dZ0's maximum; then the wave's maxima go to the slots (lane k: buffer k)
      atomicMax((unsigned int*)wave_max + slot_prev, __float_as_uint(amax * inv_cur));
      if (lane < TB_NSLOT_MAX - 1) {
        const float m = wave_max[lane];
        if (m > 0.f) atomicMax((unsigned int*)a.slot[lane] + ((unsigned)blockIdx.x & (HG_SLOT - 1)), __float_as_uint(m));
      }
      if (lane < TB_NSLOT_MAX) wave_max[lane] = 0.f;
    }
  }
  st.drain();
}
