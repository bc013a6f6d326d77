// pnrf_engine.h — device-side fused-MLP engine for gfx950 (MI355X, CDNA4).
//
// Design ("transposed MLP, register-resident activations, LDS weight stream"):
//   * Every MLP layer is computed as  H_out^T[256 x cols] = W[256 x K] * H_in^T[K x cols]
//     with the WEIGHT matrix as the MFMA A operand and the ACTIVATIONS as the B operand.
//     The 32x32 f32 result tile of v_mfma_f32_32x32x16_bf16 / v_mfma_f32_32x32x2_f32 has its
//     column (= ray / ray-sample) on the lane and its rows (= output features) in the 16
//     accumulator registers, so after bias+activation(+bf16 pack) it *is* the B operand of the
//     next layer: activations never leave the register file between layers.
//   * Weights are pre-packed on the host into 1 KiB "fragments" (64 lanes x 16 B) in exactly
//     the order the kernel consumes them, so the global->LDS copy is a linear LDS-DMA
//     (global_load_lds_dwordx4) and every ds_read_b128 is lane-linear (conflict free).
//   * The packed blob is streamed through a ring of NSLOTS x 16 KiB LDS slots shared by the
//     waves of the workgroup; PD slots are kept in flight behind a counted s_waitcnt vmcnt(N) + raw
//     s_barrier (one barrier per 16 KiB slot).
//
// This engine replaces the reference's chain of aten addmm + elu/relu launches
// (run_nerf_helpers.py:1490-1497, 1526-1533, 1331-1343).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace pnrf {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

constexpr int FRAG_BYTES = 1024;                  // 64 lanes x 16 B
#ifndef PNRF_SLOT_FRAGS
#define PNRF_SLOT_FRAGS 16
#endif
constexpr int SLOT_FRAGS = PNRF_SLOT_FRAGS;
constexpr int SLOT_BYTES = SLOT_FRAGS * FRAG_BYTES;   // 16 KiB
constexpr int NSLOTS = 4;                          // ring slots (64 KiB of LDS)
constexpr int PD = 3;                              // slots in flight ahead of the consumer
constexpr int RING_BYTES = NSLOTS * SLOT_BYTES;

enum { ACT_RELU = 0, ACT_ELU = 1, ACT_NONE = 2 };


typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// ------------------------------------------------------------------------------------------
// Weight stream: global blob -> LDS ring by LDS-DMA.  All members are wave-uniform except woff.
// NW = waves per workgroup (4 = one per SIMD, 8 = two per SIMD); every wave issues 16/NW of the
// sixteen 1 KiB LDS-DMA instructions of a slot.
// A-fragment queue policy.  XSLOT = 1: at the barrier that opens slot q the slots q AND q+1 have landed (wait_slot's
// vmcnt leaves one slot in flight instead of two), so the queue keeps refilling across the slot boundary and the first
// MFMAs behind a barrier find their fragments in registers.  Measured with the timeline stamps (tools/diag_stamps.py):
// filled from scratch behind every barrier, each tile opened with ~200 idle cycles on every SIMD (LDS round trip of the
// 8 waves' fills) on top of the ~150 cycles from last arrival to release.  The queue is filled from scratch only at a
// layer's first fragment.  XSLOT = 0: reads stay inside their slot.
#ifndef PNRF_XSLOT
#define PNRF_XSLOT 1
#endif
constexpr int XSLOT = PNRF_XSLOT;
constexpr int AHEAD_MAX = 8;               // deepest A-fragment queue of the layer engines
__host__ __device__ constexpr bool queue_fill(int f, int u, int nf) {        // at a slot head: load fragment f+u now?
  return f + u < nf && (XSLOT ? f == 0 : (f + u) / SLOT_FRAGS == f / SLOT_FRAGS);
}
__host__ __device__ constexpr bool queue_refill(int f, int ahead, int nf) {  // after consuming f: load fragment f+ahead?
  return f + ahead < nf && (XSLOT ? true : ((f + ahead) / SLOT_FRAGS == f / SLOT_FRAGS && (f % SLOT_FRAGS) + ahead < SLOT_FRAGS));
}
// PD_ slots in flight in a ring of NS_ = PD_ + 1 (the inference kernels: 3 of 4; the trainer's forward chain, whose per-layer stores share
// vmcnt with the LDS-DMA and have to have completed PD_ - 1 slots after their issue, runs 7 of 8; so do the inference kernels' one-batch
// shapes — a single 4-wave workgroup with a SIMD per wave consumes a 16 KiB slot in ~0.25 us, and three slots in flight over a ~1.1 us
// L2 round trip deliver 44 GB/s where the MFMAs ask for 64: with seven in flight the stream keeps up)
template <int NW, int PD_ = PD, int NS_ = NSLOTS>
struct WStream {
  static_assert(NS_ == PD_ + 1 && (NS_ & (NS_ - 1)) == 0, "ring protocol assumes one free slot and a power-of-two ring");
  static constexpr int RING_SLOTS = NS_;
  static constexpr int RING_BYTES = NS_ * SLOT_BYTES;
  static constexpr int LOADS_PER_WAVE = SLOT_FRAGS / NW;
  const char* g;       // packed blob
  char* ring;          // LDS ring base
  uint32_t nslots;     // slots in the blob (multiple of NSLOTS)
  uint32_t src_slot;   // next source slot
  uint32_t dst_pos;    // next ring position
  uint32_t woff;       // per-lane byte offset inside a slot (source side)
  uint32_t wbase;      // wave-uniform byte offset inside a slot (LDS side)

  __device__ __forceinline__ void init(const void* blob, uint32_t nslots_, char* ring_) {
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    g = (const char*)blob; ring = ring_; nslots = nslots_;
    src_slot = 0; dst_pos = 0;
    wbase = wave * (LOADS_PER_WAVE * FRAG_BYTES);
    woff = wbase + lane * 16;
  }
  // The LDS-DMA pieces go out through inline asm, not __builtin_amdgcn_global_load_lds: hipcc books the builtin as a FLAT
  // access that may touch LDS, and from then on waits lgkmcnt(0) before every MFMA that consumes a ds_read — behind each
  // barrier a wave then sits out its whole 8-9 fragment queue fill (all 8 waves': 72 KiB through the 256 B/clk LDS)
  // before its first MFMA.  Hidden from the compiler the reads keep their counted lgkmcnt(N).  Completion is counted by
  // wait_slot()'s own vmcnt; M0 (compiler-reserved) is saved and restored inside the statement.
  __device__ __forceinline__ void issue_loads() {
    const uint64_t src = (uint64_t)(uintptr_t)g + (uint64_t)src_slot * SLOT_BYTES;                 // wave-uniform
    const uint32_t dst = (uint32_t)(uintptr_t)(lptr_t)ring + dst_pos * SLOT_BYTES + wbase;       // wave-uniform LDS byte address
#pragma unroll
    for (int i = 0; i < LOADS_PER_WAVE; ++i) {
      uint32_t keep;
      // s_nop 2: with the two s_mov in front of it, five wait states between whatever wrote the address SGPRs and the VMEM instruction that reads
      // them.  The compiler cannot see a VMEM instruction in here, so it does not keep its "VALU writes SGPR -> VMEM reads it: 5 wait states"
      // rule for us — and in the 4-wave kernels (four pieces per wave and slot: the address pairs are spilled) it restores them with
      // v_readlane_b32 right in front of this statement (round 4: 36 such places in refine_kernel<1, 4, ..>, three wait states each).
      asm volatile(
          "s_mov_b32 %0, m0\n\t"
          "s_mov_b32 m0, %3\n\t"
          "s_nop 2\n\t"
          "global_load_lds_dwordx4 %1, %2\n\t"
          "s_mov_b32 m0, %0"
          : "=&s"(keep)
          : "v"(woff), "s"(src + (uint64_t)(i * FRAG_BYTES)), "s"(dst + (uint32_t)(i * FRAG_BYTES))
          : "memory");
    }
  }
  __device__ __forceinline__ void advance() {
    src_slot = (src_slot + 1 == nslots) ? 0u : src_slot + 1;
    dst_pos = (dst_pos + 1) & (NS_ - 1);
  }
  __device__ __forceinline__ void issue() { issue_loads(); advance(); }
  // Refill of the ring position freed by wait_slot(); the layers call it after the first MFMA of every slot, so that the
  // MFMA pipe is already busy while this wave pays the DMA issue.  (Moving the issue further into the slot, or staggering
  // it between the two waves of a SIMD, was measured and is slower: 3.92 -> 3.96..4.11 ms for the NeRF kernel.)
  __device__ __forceinline__ void slot_issue(int fpos) {
    if (fpos == 0) issue();
  }
  // Fill the pipeline: PD slots in flight.
  __device__ __forceinline__ void prologue() {
#pragma unroll
    for (int i = 0; i < PD_; ++i) issue();
  }
  // Called at the boundary between slot q-1 and slot q, by every wave, in the same order.
  //  vmcnt(N): my share of slot q has landed (N = LOADS_PER_WAVE * (PD-1) younger loads may still be in
  //            flight);
  //  barrier : every wave's share has landed AND every wave has finished reading slot q-1, whose
  //            ring position (== position of slot q+PD since NSLOTS == PD+1) is then refilled.
  //  (Measured with tools/diag_stamps.py: the vmcnt wait is ~0; letting reads run one slot ahead across
  //  the barrier with an 8-slot ring bought nothing, so reads stay inside their slot.)
  __device__ __forceinline__ void begin() { wait_slot(); issue(); }
  __device__ __forceinline__ void wait_slot() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS_PER_WAVE * (PD_ - 1 - XSLOT)) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  // LDS-DMA still in flight at kernel end would land in another workgroup's LDS: drain.
  __device__ __forceinline__ void drain() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
};
static_assert(NSLOTS == PD + 1, "ring protocol assumes one free slot");
static_assert(PD - 1 - XSLOT >= 1 && AHEAD_MAX <= SLOT_FRAGS, "queue reads at most one slot ahead");

// expm1(x) for x <= 0, branch-free (ocml's expm1f compiles to divergent branch blocks that cannot be
// interleaved with MFMAs).  x > -0.5: degree-8 Taylor polynomial in Horner form; else expf(x) - 1.
// <= 1 ulp against the exact value over the whole negative axis (same as torch/Sleef expm1).
__device__ __forceinline__ float expm1_neg(float x) {
  float p = 2.48015873e-5f;                 // 1/8!
  p = fmaf(p, x, 1.98412698e-4f);           // 1/7!
  p = fmaf(p, x, 1.38888889e-3f);           // 1/6!
  p = fmaf(p, x, 8.33333333e-3f);           // 1/5!
  p = fmaf(p, x, 4.16666667e-2f);           // 1/4!
  p = fmaf(p, x, 1.66666667e-1f);           // 1/3!
  p = fmaf(p, x, 0.5f);
  const float small = fmaf(p, x * x, x);
  const float large = expf(x) - 1.f;
  return x > -0.5f ? small : large;
}
__device__ __forceinline__ float act_f32(float v, int act) {
  // torch: relu = max(x,0); elu(alpha=1) = x > 0 ? x : expm1(x)   (F.elu, helpers:1494)
  if (act == ACT_RELU) return fmaxf(v, 0.f);
#ifdef PNRF_EXACT_ELU
  // <= 1 ulp expm1, ~30 VALU per activation; evaluated unconditionally: with an expensive negative branch a source-level
  // `v > 0 ? v : f(v)` compiles to a real branch per value, which the scheduler cannot interleave with MFMAs
  return fmaxf(v, 0.f) + expm1_neg(fminf(v, 0.f));
#else
  // exp(x) - 1 through v_exp_f32: absolute error <= 6e-8 (one fp32 rounding of an O(1) activation) instead of expm1's
  // relative 1e-7; measured on 61k rays x 5 weight sets: identical sort indices, max depth error 6.6e-7 vs 6.3e-7, and
  // 13 % less sampler time.  With this cheap negative side the compare + select form is branch-free (v_cmp, v_cndmask) and one
  // VALU shorter than max(v,0) + (exp(min(v,0)) - 1), with the same value on both sides of zero; exp of a large positive v is
  // inf and never selected.
  return v > 0.f ? v : __expf(v) - 1.f;
#endif
}
// ELU on log2(e)-scaled values (bf16 refine net, split-fp16 sampler).  The packer multiplies the first layer's weights and the bias of
// every ELU layer by log2(e) and divides the output layer's weights by it (pnrf_pack.hip: Layer::wscale / bscale), so a pre-activation
// arrives as y = log2(e) x and the activation is kept as h' = log2(e) ELU(x) = (y > 0 ? y : f(y)), f(y) = log2(e) (2^y - 1).
// f(y) >= y everywhere (their difference has its minimum 0 at y = 0) and f has the sign of y, so h' is the MEDIAN of (y, f(y), 0):
// v_exp_f32 straight on the accumulator, v_fma_f32, v_med3_f32 — 16 issue cycles instead of the 24 of v_mul (x log2 e), v_exp, v_add (-1),
// v_cmp, v_cndmask.  2^y = inf for large y gives f = inf and the median is y.  Same accuracy as exp(x) - 1 through v_exp_f32: an absolute
// error of one fp32 rounding of an O(1) value.
constexpr float LOG2E = 1.4426950408889634f;
constexpr double LOG2E_D = 1.4426950408889634074;
__device__ __forceinline__ float elu_scaled(float y) {
  const float f = fmaf(__builtin_amdgcn_exp2f(y), LOG2E, -LOG2E);
  return __builtin_amdgcn_fmed3f(y, f, 0.f);
}
__device__ __forceinline__ float act_fast(float v, int act) {
  // bf16 path: the result is rounded to 8 significant bits, v_exp_f32 is accurate enough.
  if (act == ACT_NONE) return v;
  if (act == ACT_RELU) {      // one v_max_f32; written in asm because fmaxf() on an MFMA result adds a canonicalising v_max v,v,v
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(v));
    return r;
  }
  return elu_scaled(v);
}

// ------------------------------------------------------------------------------------------
// Layer loops.  Common structure (both precisions):
//   * fragment f = to*KS + ks of the layer lives in slot f/16 (layer start is slot aligned);
//     POS0 = ring position of the layer's first slot (compile-time, see pnrf_layout.h);
//   * A fragments go through a rotating prefetch queue AHEAD deep, refilled from the head of a slot
//     right behind its barrier (reads never cross a slot barrier);
//   * the epilogue of a finished tile is DEFERRED and SPLIT: `epi1(to, piece, acc)` handles one
//     piece of tile `to`'s accumulators and the pieces are issued one by one between the MFMA groups
//     of tile to+1, each group fenced by sched_barrier(0).  With two waves per SIMD this keeps VALU
//     work and MFMAs interleaved at instruction granularity instead of phase-locked behind the slot
//     barrier.  The LAST tile's accumulators are returned in `last`; the caller passes their pieces
//     as `pre1(piece)` of whatever layer comes next.
//
// One bf16 layer:  NT output tiles of 32 rows, KS k-steps of 16, NCB column blocks of 32.
//   ringlane  = ring + lane*16            (LDS)
//   biaslane  = bias_layer + h*16 floats  (LDS; packed [tile][h][16])
//   Bi(cb,ks) = B operand of k-step ks for column block cb
//   epi1(to, piece, acc[NCB]) / pre1(piece): piece = 0,1 = accumulator registers 8*piece..8*piece+7
//               (= one packed bf16 B fragment of the next layer per column block).
#ifndef PNRF_BF16_AT0
#define PNRF_BF16_AT0 6
#endif
#ifndef PNRF_BF16_ATSTEP
#define PNRF_BF16_ATSTEP 6
#endif
constexpr int BF16_PIECES = 2;
// F16 = true: the same loop on v_mfma_f32_32x32x16_f16 (fragments and B operands are f16x8): pass 1 of the two-pass sampler.
__device__ __forceinline__ f32x16 mfma_32x32x16(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mfma_32x32x16(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
template <int NCB, int KS, int NT, int POS0, int PIECES = BF16_PIECES, bool F16 = false, class ST, class BFn, class Epi1, class Pre1>
__device__ __forceinline__ void layer_bf16(ST& st, const char* ringlane, const float* biaslane, BFn Bi, Epi1 epi1, Pre1 pre1,
                                           f32x16 (&last)[NCB]) {
  constexpr int NF = KS * NT;
  constexpr int AHEAD = KS < 8 ? KS : 8;
  using frag_t = typename std::conditional<F16, f16x8, bf16x8>::type;
  auto frag_ptr = [&](int g) {
    return (const frag_t*)(ringlane + ((POS0 + g / SLOT_FRAGS) % ST::RING_SLOTS) * SLOT_BYTES + (g % SLOT_FRAGS) * FRAG_BYTES);
  };
  f32x16 pend[NCB];
  frag_t aq[AHEAD];
  // the bias of tile to+1 is read from LDS at the head of tile `to` (software pipelined): read in place, the first MFMA of
  // every tile would wait a full LDS round trip for its accumulator with both waves of the SIMD phase-locked behind the barrier
  f32x4 nb0, nb1, nb2, nb3;
  { const f32x4* bp = (const f32x4*)biaslane; nb0 = bp[0]; nb1 = bp[1]; nb2 = bp[2]; nb3 = bp[3]; }
#pragma unroll
  for (int to = 0; to < NT; ++to) {
    f32x16 acc[NCB];
    {
      const f32x4 b0 = nb0, b1 = nb1, b2 = nb2, b3 = nb3;
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { acc[cb][i] = b0[i]; acc[cb][4 + i] = b1[i]; acc[cb][8 + i] = b2[i]; acc[cb][12 + i] = b3[i]; }
      }
      if (to + 1 < NT) { const f32x4* bp = (const f32x4*)(biaslane + (to + 1) * 32); nb0 = bp[0]; nb1 = bp[1]; nb2 = bp[2]; nb3 = bp[3]; }
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int f = to * KS + ks;
      if (f % SLOT_FRAGS == 0) {
        st.wait_slot();          // the slot is readable now: (re)fill the queue from its head
#pragma unroll
        for (int u = 0; u < AHEAD; ++u)
          if (queue_fill(f, u, NF)) aq[(f + u) % AHEAD] = *frag_ptr(f + u);
      }
      const frag_t a = aq[f % AHEAD];
      if (queue_refill(f, AHEAD, NF)) aq[f % AHEAD] = *frag_ptr(f + AHEAD);      // keep the queue full
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) acc[cb] = mfma_32x32x16(a, Bi(cb, ks), acc[cb]);
      // deferred epilogue pieces of the previous tile, spread over this tile's k-steps
#pragma unroll
      for (int pc = 0; pc < PIECES; ++pc) {
        // k-step after which piece pc is issued.  PIECES = 16 (one activation each; the ELU nets): one piece per k-step, so that every MFMA
        // gap carries the same three or four VALU instructions (v_exp, v_fma, v_med3, every second time v_cvt_pk) — 8 + 16..20 issue cycles
        // inside the MFMA's 32.  In a layer's first tile the pieces are the previous layer's last tile, whose two fragments are read by
        // k-steps 14 and 15: pieces 0..12 behind k-steps 0..12, the last three behind k-step 13; in the other tiles one k-step later (the
        // previous tile's last MFMA is still in flight at k-step 0).  PIECES = 8: one piece every second k-step.  PIECES = 2: two halves.
        // The packed dword a piece writes is produced by an inline-asm conversion, which hipcc's hazard recogniser does not see as a VALU
        // write: it is never the operand of the very next MFMA (one whole MFMA between the last conversion and its consumer, in every layer
        // engine here).
#ifndef PNRF_BF16_P8_AT0
#define PNRF_BF16_P8_AT0 1
#endif
#ifndef PNRF_BF16_P8_STEP
#define PNRF_BF16_P8_STEP 2
#endif
        const int at = PIECES == 16 ? (KS >= 16 ? (to == 0 ? (pc < 13 ? pc : 13) : (pc < 15 ? pc + 1 : 15)) : 1 + (pc * (KS - 1)) / PIECES)
                       : PIECES != 2 ? ((to == 0 || KS < 16) ? 1 + pc : PNRF_BF16_P8_AT0 + PNRF_BF16_P8_STEP * pc)
                                     : (KS >= 8 ? PNRF_BF16_AT0 + pc * PNRF_BF16_ATSTEP : (KS >= 4 ? 1 + pc * (KS / 4) : KS - 1));
        if (ks == (at < KS ? at : KS - 1)) {
          if (to == 0) pre1(pc);
          else epi1(to - 1, pc, pend);
        }
      }
      st.slot_issue(f % SLOT_FRAGS);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) pend[cb] = acc[cb];
  }
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) last[cb] = pend[cb];
}
template <int KS, int NT> constexpr int layer_slots_bf16() { return (KS * NT + SLOT_FRAGS - 1) / SLOT_FRAGS; }

// One f32 layer (exact fp32 FMA chain, v_mfma_f32_16x16x4_f32), one column block of 16 per wave.
// 16 columns/wave keep two layers of fp32 activations in 64 + 64 registers, so the sampler runs two
// waves per SIMD: one wave's VALU work overlaps the other wave's MFMAs, and the 40-cycle
// dependent-accumulator latency of this instruction (issue 32) is hidden the same way.
//   lane l: column l&15, quarter q = l>>4.  A[row l&15][k=q], B[k=q][col l&15]; D reg r = row 4q+r.
//   NT  = output tiles of 16 rows; KS4 = k-steps/4 = fragments per tile (a fragment carries, per
//         lane, the A values of 4 consecutive k-steps).
//   Bf(kk) = B operand (one float per lane) of k-step kk.
//   epi1(to, r, value) / pre1(r): piece r = accumulator register r (0..3) of the deferred tile.
template <int KS4, int NT, int POS0, class ST, class BFn, class Epi1, class Pre1>
__device__ __forceinline__ void layer_f32(ST& st, const char* ringlane, const float* biaslane, BFn Bf, Epi1 epi1, Pre1 pre1, f32x4& last) {
  constexpr int NF = KS4 * NT;                 // fragments in the layer
  constexpr int AHEAD = KS4 < 8 ? KS4 : 8;     // A fragments in flight ahead of the MFMAs
  auto frag_ptr = [&](int g) {
    return (const f32x4*)(ringlane + ((POS0 + g / SLOT_FRAGS) % ST::RING_SLOTS) * SLOT_BYTES + (g % SLOT_FRAGS) * FRAG_BYTES);
  };
  f32x4 pend;
  f32x4 aq[AHEAD];                             // rotating prefetch queue (static indices after unrolling)
  f32x4 nbias = *(const f32x4*)biaslane;        // bias of the next tile, read one tile ahead (see layer_bf16)
#pragma unroll
  for (int to = 0; to < NT; ++to) {
    f32x4 acc = nbias;
    if (to + 1 < NT) nbias = *(const f32x4*)(biaslane + (to + 1) * 16);
#pragma unroll
    for (int fr = 0; fr < KS4; ++fr) {
      const int f = to * KS4 + fr;
      if (f % SLOT_FRAGS == 0) {
        st.wait_slot();          // the slot is readable now: (re)fill the queue from its head
#pragma unroll
        for (int u = 0; u < AHEAD; ++u)
          if (queue_fill(f, u, NF)) aq[(f + u) % AHEAD] = *frag_ptr(f + u);
      }
      const f32x4 a = aq[f % AHEAD];
      if (queue_refill(f, AHEAD, NF)) aq[f % AHEAD] = *frag_ptr(f + AHEAD);      // keep the queue full
#pragma unroll
      for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], Bf(4 * fr + i), acc, 0, 0, 0);
      // deferred epilogue: register r of the previous tile after MFMA group 1 + r*(KS4/4)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int at = KS4 >= 4 ? 1 + r * (KS4 / 4) : KS4 - 1;
        if (fr == (at < KS4 ? at : KS4 - 1)) {
          if (to == 0) pre1(r);
          else epi1(to - 1, r, pend[r]);
        }
      }
      st.slot_issue(f % SLOT_FRAGS);
      __builtin_amdgcn_sched_barrier(0);
    }
    pend = acc;
  }
  last = pend;
}
template <int KS4, int NT> constexpr int layer_slots_f32() { return (KS4 * NT + SLOT_FRAGS - 1) / SLOT_FRAGS; }

// ------------------------------------------------------------------------------------------
// Split-fp16 layer ("f16x2"): fp32-grade products from half-precision MFMAs.
//   x = x_hi + x_lo, W = W_hi + W_lo with x_hi = fp16(x), x_lo = fp16((x - x_hi) * 2^11) (the low plane is stored
//   scaled by 2^11 so that it stays a normal fp16 number), and   W.x ~= W_hi.x_hi + 2^-11 (W_hi.x_lo + W_lo.x_hi):
//   22 significand bits per operand, the dropped W_lo.x_lo term is 2^-22 relative.  Three v_mfma_f32_16x16x32_f16
//   per 32-deep k-step instead of eight v_mfma_f32_16x16x4_f32 of twice the duration: 3/16 of the MFMA cycles.
//   Two fp32 accumulators per tile: `main` (hi.hi, initialised with the bias) and `cross` (the two mixed terms).
//   Measured (oracle emulation and on the GPU, tests/idxcheck_gpu.py): depth error and sort indices indistinguishable from
//   the exact-fp32 MFMA chain.
// Geometry: 16 columns per wave (lane l: column l&15, group g = l>>4), output tiles of 16 rows handled in PAIRS — the
//   pair (2tp, 2tp+1) is exactly the k-step tp (32 features) of the next layer: element j of group g = register j&3 of
//   tile 2tp + (j>>2) = feature 32tp + 16(j>>2) + 4g + (j&3).
//   Fragment order in the stream: (tp, ks, tile-in-pair, plane) -> 4 fragments per k-step; NTP tile pairs, KS k-steps.
//   Bf(ks, plane) -> f16x8 B operand; epi1(tp, pc, main[2], cross[2]) / pre1(pc): piece pc (0..3) = accumulator registers 2(pc&1), 2(pc&1)+1 of tile
//   pc>>1 of the deferred pair = one packed dword of the next layer's hi plane and one of its lo plane.  One piece every second k-step (a k-step
//   carries six MFMAs); in a layer's first tile pair one per k-step from k-step 1, so that the previous layer's activations are complete before
//   the k-step that reads them.
#ifndef PNRF_H16_AHEAD
#define PNRF_H16_AHEAD 8
#endif
#ifndef PNRF_H16_AT0
#define PNRF_H16_AT0 1
#endif
#ifndef PNRF_H16_ATSTEP
#define PNRF_H16_ATSTEP 2
#endif
#ifndef PNRF_H16_PIECES
#define PNRF_H16_PIECES 8
#endif
// deferred epilogue pieces per tile pair.  4: piece pc = registers 2(pc&1), 2(pc&1)+1 of tile pc>>1, one every second k-step;
// 8: piece pc = register pc&3 of tile pc>>2, one per k-step (the even one leaves its activation in the accumulator register for the odd one,
// which packs the pair)
constexpr int H16_PIECES = PNRF_H16_PIECES;
constexpr float H16_LO_SCALE = 2048.f;
template <int KS, int NTP, int POS0, int QUEUE = PNRF_H16_AHEAD, bool BIAS = true, class ST, class BFn, class Epi1, class Pre1>
__device__ __forceinline__ void layer_h16x2(ST& st, const char* ringlane, const float* biaslane, BFn Bf, Epi1 epi1, Pre1 pre1,
                                            f32x4 (&last_main)[2], f32x4 (&last_cross)[2]) {
  constexpr int NF = NTP * KS * 4;
  constexpr int AHEAD = NF < QUEUE ? NF : QUEUE;
  auto frag_ptr = [&](int g) {
    return (const f16x8*)(ringlane + ((POS0 + g / SLOT_FRAGS) % ST::RING_SLOTS) * SLOT_BYTES + (g % SLOT_FRAGS) * FRAG_BYTES);
  };
  f32x4 pm[2], pc_[2];
  f16x8 aq[AHEAD];
  f32x4 nbias[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};         // bias of the next tile pair, one pair ahead (BIAS = false: none —
  if constexpr (BIAS) { nbias[0] = *(const f32x4*)biaslane; nbias[1] = *(const f32x4*)(biaslane + 16); }   // the trainer's input-gradient products)
#pragma unroll
  for (int tp = 0; tp < NTP; ++tp) {
    f32x4 mn[2], cr[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) { mn[t] = nbias[t]; cr[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    if (BIAS && tp + 1 < NTP) {
#pragma unroll
      for (int t = 0; t < 2; ++t) nbias[t] = *(const f32x4*)(biaslane + (2 * (tp + 1) + t) * 16);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
          const int f = ((tp * KS + ks) * 2 + t) * 2 + pl;
          if (f % SLOT_FRAGS == 0) {
            st.wait_slot();
#pragma unroll
            for (int u = 0; u < AHEAD; ++u)
              if (queue_fill(f, u, NF)) aq[(f + u) % AHEAD] = *frag_ptr(f + u);
          }
          const f16x8 a = aq[f % AHEAD];
          if (queue_refill(f, AHEAD, NF)) aq[f % AHEAD] = *frag_ptr(f + AHEAD);
          if (pl == 0) {
            mn[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, Bf(ks, 0), mn[t], 0, 0, 0);      // W_hi . x_hi
            cr[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, Bf(ks, 1), cr[t], 0, 0, 0);      // W_hi . x_lo
          } else {
            cr[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, Bf(ks, 0), cr[t], 0, 0, 0);      // W_lo . x_hi
          }
          st.slot_issue(f % SLOT_FRAGS);
        }
      }
#pragma unroll
      for (int pc = 0; pc < H16_PIECES; ++pc) {
        const int at = KS < 8 ? KS - 1
                       : H16_PIECES == 8 ? (tp == 0 ? (pc < 5 ? pc : 5) : pc)       // first pair of a layer: done a whole k-step before k-step 7 reads the
                                                                                    // result (asm conversions: see layer_bf16's note on the consumer)
                                         : (tp == 0 ? 1 + pc : PNRF_H16_AT0 + pc * PNRF_H16_ATSTEP);
        if (ks == (at < KS ? at : KS - 1)) {
          if (tp == 0) pre1(pc);
          else epi1(tp - 1, pc, pm, pc_);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) { pm[t] = mn[t]; pc_[t] = cr[t]; }
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) { last_main[t] = pm[t]; last_cross[t] = pc_[t]; }
}
template <int KS, int NTP> constexpr int layer_slots_h16x2() { return (KS * NTP * 4 + SLOT_FRAGS - 1) / SLOT_FRAGS; }
// feature supplied by element j of lane group g in k-step ks when the B operand is the previous layer's tile pair
__host__ __device__ constexpr int hidden_feat_h16(int ks, int g, int j) { return 32 * ks + 16 * (j >> 2) + 4 * g + (j & 3); }

// ------------------------------------------------------------------------------------------
// bf16 layer on v_mfma_f32_16x16x32_bf16 ("b16").  Same FLOPs per fragment as layer_bf16 (one 1 KiB A fragment = 16 output
// rows x 32 k feeds two 16-cycle MFMAs, one per 16-column block) but the chip sustains a higher clock on this shape under
// real data: in the tile structure of these kernels (tools/mfma_shape_probe.hip, random operands, 64 VALU + barrier per
// tile) 1.77 PFLOP/s against 1.62 for 32x32x16.
// Geometry as layer_h16x2: lane l = column l&15 of a block, group g = l>>4; output tiles of 16 rows in PAIRS, the pair
// (2tp, 2tp+1) is k-step tp of the next layer (element j of group g = register j&3 of tile 2tp + (j>>2));
// fragment order in the stream: (tp, ks, tile-in-pair).  Bf(cb, ks) -> bf16x8; epi1(tp, pc, acc[2][2]) / pre1(pc): piece pc =
// tile pc of the deferred pair, acc[t][cb].
#ifndef PNRF_B16_AHEAD
#define PNRF_B16_AHEAD 8
#endif
#ifndef PNRF_B16_AT0
#define PNRF_B16_AT0 1
#endif
#ifndef PNRF_B16_ATSTEP
#define PNRF_B16_ATSTEP 2
#endif
// NCB: 16-column blocks per wave fed by every weight fragment (2 with 8 waves per workgroup, 4 with 4 waves: half the LDS fragment reads per MFMA)
// VT = bf16x8 (v_mfma_f32_16x16x32_bf16) or f16x8 (v_mfma_f32_16x16x32_f16: same cycles, 11 significand bits instead of 8)
__device__ __forceinline__ f32x4 mfma_16x16x32(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mfma_16x16x32(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
// TMAX = 1: only the first tile of every pair carries output rows (the 4-wide output layers): the second tile's fragments still pass through
// the ring (the stream layout is unchanged) but no MFMA is issued for them and last[1][*] stays the bias.
template <int KS, int NTP, int POS0, int NCB = 2, class VT = bf16x8, int TMAX = 2, class ST, class BFn, class Epi1, class Pre1>
__device__ __forceinline__ void layer_b16(ST& st, const char* ringlane, const float* biaslane, BFn Bf, Epi1 epi1, Pre1 pre1, f32x4 (&last)[2][NCB]) {
  constexpr int NF = NTP * KS * 2;
  constexpr int AHEAD = NF < PNRF_B16_AHEAD ? NF : PNRF_B16_AHEAD;
  auto frag_ptr = [&](int g) {
    return (const VT*)(ringlane + ((POS0 + g / SLOT_FRAGS) % ST::RING_SLOTS) * SLOT_BYTES + (g % SLOT_FRAGS) * FRAG_BYTES);
  };
  f32x4 pend[2][NCB];
  VT aq[AHEAD];
  f32x4 nbias[2] = {*(const f32x4*)biaslane, *(const f32x4*)(biaslane + 16)};
#pragma unroll
  for (int tp = 0; tp < NTP; ++tp) {
    f32x4 acc[2][NCB];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) acc[t][cb] = nbias[t];        // (no moves: the compiler passes the bias registers as the first MFMAs' C operand)
    if (tp + 1 < NTP) {
#pragma unroll
      for (int t = 0; t < 2; ++t) nbias[t] = *(const f32x4*)(biaslane + (2 * (tp + 1) + t) * 16);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int f = (tp * KS + ks) * 2 + t;
        if (f % SLOT_FRAGS == 0) {
          st.wait_slot();
#pragma unroll
          for (int u = 0; u < AHEAD; ++u)
            if (queue_fill(f, u, NF)) aq[(f + u) % AHEAD] = *frag_ptr(f + u);
        }
        const VT a = aq[f % AHEAD];
        if (queue_refill(f, AHEAD, NF)) aq[f % AHEAD] = *frag_ptr(f + AHEAD);
        if (t < TMAX) {
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) acc[t][cb] = mfma_16x16x32(a, Bf(cb, ks), acc[t][cb]);
        }
        st.slot_issue(f % SLOT_FRAGS);
      }
      // deferred epilogue of the previous pair in NCB pieces (tile pc, column blocks [cb0, cb0 + 2)): with one wave per SIMD (NCB = 4) nothing
      // else fills the MFMA pipe while a piece's VALU burst runs, so the pieces are half as large and sit in four different k-steps
#pragma unroll
      for (int pp = 0; pp < NCB; ++pp) {
        const int pc = NCB == 2 ? pp : pp >> 1, cb0 = NCB == 2 ? 0 : 2 * (pp & 1);
        const int at = KS >= 4 ? (NCB == 2 ? PNRF_B16_AT0 + pc * PNRF_B16_ATSTEP : 1 + pp) : KS - 1;
        if (ks == (at < KS ? at : KS - 1)) {
          if (tp == 0) pre1(pc, cb0);
          else epi1(tp - 1, pc, pend, cb0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) pend[t][cb] = acc[t][cb];
  }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) last[t][cb] = pend[t][cb];
}

// ------------------------------------------------------------------------------------------
// ELU layer on v_mfma_f32_16x16x32_f16 ("e16", round 6): layer_b16's geometry — two 16-column blocks per wave, 16-row output tiles in pairs, 32-deep
// k-steps, fragments (tp, ks, tile-in-pair) — with the deferred epilogue of the ELU nets: ONE activation per piece (v_exp, v_fma, v_med3, every second
// time the conversion of the pair), one piece behind every fragment's two MFMAs, so that each 32-cycle gap on the pipe carries the same four VALU
// instructions (layer_bf16 with PIECES = 16 is the same idea on 32-cycle MFMAs).
//   Bf(cb, ks) -> B operand; epi1(tp, p, pend) / pre1(p): piece p = 0..15 = register p & 3 of column block (p >> 2) & 1 of tile p >> 3 of the deferred pair.
//   Gap q = 2 ks + t of a pair.  Pairs tp > 0: piece p behind gap p G / 16 (G = 2 KS gaps).  First pair of a layer: the pieces finish the previous layer's
//   last pair = this layer's k-step KS - 1 (gaps G - 2, G - 1), and an asm conversion must not be the operand of the very next MFMA (layer_bf16's note):
//   all sixteen are out by gap G - 4.
template <int KS, int NTP, int POS0, class VT, class ST, class BFn, class Epi1, class Pre1>
__device__ __forceinline__ void layer_e16(ST& st, const char* ringlane, const float* biaslane, BFn Bf, Epi1 epi1, Pre1 pre1, f32x4 (&last)[2][2]) {
  constexpr int NF = NTP * KS * 2, G = 2 * KS;
  constexpr int AHEAD = NF < 8 ? NF : 8;
  static_assert(G >= 4, "at least two k-steps");
  auto frag_ptr = [&](int g) {
    return (const VT*)(ringlane + ((POS0 + g / SLOT_FRAGS) % ST::RING_SLOTS) * SLOT_BYTES + (g % SLOT_FRAGS) * FRAG_BYTES);
  };
  f32x4 pend[2][2];
  VT aq[AHEAD];
  f32x4 nbias[2] = {*(const f32x4*)biaslane, *(const f32x4*)(biaslane + 16)};
#pragma unroll
  for (int tp = 0; tp < NTP; ++tp) {
    f32x4 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) acc[t][cb] = nbias[t];
    if (tp + 1 < NTP) {
#pragma unroll
      for (int t = 0; t < 2; ++t) nbias[t] = *(const f32x4*)(biaslane + (2 * (tp + 1) + t) * 16);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int f = (tp * KS + ks) * 2 + t;
        if (f % SLOT_FRAGS == 0) {
          st.wait_slot();
#pragma unroll
          for (int u = 0; u < AHEAD; ++u)
            if (queue_fill(f, u, NF)) aq[(f + u) % AHEAD] = *frag_ptr(f + u);
        }
        const VT a = aq[f % AHEAD];
        if (queue_refill(f, AHEAD, NF)) aq[f % AHEAD] = *frag_ptr(f + AHEAD);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[t][cb] = mfma_16x16x32(a, Bf(cb, ks), acc[t][cb]);
        st.slot_issue(f % SLOT_FRAGS);
#pragma unroll
        for (int p = 0; p < 16; ++p) {
          const int at = tp == 0 ? (p * (G - 3)) / 16 : (p * G) / 16;
          if (2 * ks + t == at) {
            if (tp == 0) pre1(p);
            else epi1(tp - 1, p, pend);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) pend[t][cb] = acc[t][cb];
  }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) last[t][cb] = pend[t][cb];
}

// Row of a 32x32 accumulator tile held in register g of a lane in half h (cdna guide §3).
__host__ __device__ constexpr int acc_row(int g, int h) { return (g & 3) + 8 * (g >> 2) + 4 * h; }

// Feature (row of the 256-wide activation) that element j of half h supplies in bf16 k-step ks
// when the B operand is the previous layer's packed accumulator (hidden layers).
__host__ __device__ constexpr int hidden_feat_bf16(int ks, int h, int j) { return 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * h + (j & 3); }
// Same for the f32 engine (16x16x4): k-step kk = 4t + r, quarter q supplies register r of tile t.
__host__ __device__ constexpr int hidden_feat_f32(int kk, int q) { return 16 * (kk >> 2) + 4 * q + (kk & 3); }

}  // namespace pnrf
