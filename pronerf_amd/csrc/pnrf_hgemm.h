// Split-fp16 kernels of the trainer (included by pnrf_train.hip inside its anonymous namespace): the layer products (hgemm_kernel), the weight
// gradients (dwh_kernel), a layer's backward as one launch (layer_bwd_kernel), the layer chains of the 4096-row nets (hgemm_rchain_kernel), the
// 64-row chains of the wide layers (hgemm_wchain_kernel; an A/B alternative since pnrf_tchain.h), the iteration's grouped weight gradients
// (dwh_group_kernel), and the body that keeps the fp16 weight planes current (split_weights_body, a block range of pnrf_train.hip's iter_prepare_kernel).
//
// Y = act(X W^T + b) and dX = (dZ W [+ dX]) * act'(H) with fp32-grade results at 3/16 of the fp32 MFMA cycles: both operands are split
//   x = x_hi + 2^-11 x_lo',  x_hi = fp16(x),  x_lo' = fp16((x - x_hi) 2^11)          (22 significand bits per operand)
// and the product is accumulated in fp32 as   main += W_hi x_hi ;  cross += W_hi x_lo' + W_lo' x_hi ;  C = main + 2^-11 cross
// (the dropped W_lo x_lo term is 2^-22 relative) — the scheme of the inference sampler (pnrf_engine.h layer_h16x2), three
// v_mfma_f32_16x16x32_f16 per 32-deep step instead of eight v_mfma_f32_16x16x4_f32 of twice the duration.
//
// Operand ranges.  Weights and forward activations are O(1).  Gradients are not (1e-9 .. 1e-3): the A operand of a backward product is
// multiplied by a power of two s chosen from max |A| — which the kernel that produced A left in a device slot (hg_slot_write) —
// so that s max|A| lies in [2^11, 2^12); the result is multiplied by 1 / s.  Elements more than 2^36 below the tensor's maximum flush to
// zero: 2^-12 of an fp32 ulp of the largest term of their dot product.
//
// Layout.  Workgroup = 8 waves on tiles of 16 MI rows x 256 columns; wave w owns columns 32 w .. 32 w + 31 (two 16-column MFMA tiles) of all
// rows.  The kernel is persistent: workgroup b of G works on row tiles b, b + G, ... as ONE software pipeline over all their 64-k chunks, so
// the fetches of a tile's first chunks overlap the previous tile's last MFMAs and its epilogue (a product with K = 256 has only four chunks:
// started afresh per tile, every tile paid an HBM and an L2 round trip with the MFMA pipe idle — 36 us instead of 21 us per 32 768 x 256 x 256
// product in the probe build without weight fetches, tools/hgemm_probe.hip).
//   * A [M, K] fp32 streams from HBM once: 512 threads fetch a chunk (16 bytes each, whole 256-byte row segments) two chunks ahead into
//     registers, split it one chunk ahead and write the two fp16 planes [row][64 + 8] to LDS (double buffered; row stride 144 bytes: the
//     ds_read_b128 of the 16 rows of a fragment cover all 64 banks once).
//   * The weight planes are stored fragment-major (split_weights_body): the 16 columns x 32 k block that one MFMA consumes is one contiguous
//     KiB in lane order, so a fragment fetch is a single fully coalesced 16-byte-per-lane load from L2.  Fragments are fetched per 32-deep step,
//     three steps ahead, into four rotating register sets.  Two orientations are kept: [out][in] for the forward product, [in][out] for dX.
//   * Every fetch inside the loop is a buffer load — resource in scalar registers, a per-lane offset that never changes, a scalar offset per
//     chunk — so the loop writes no vector register for addressing.  (With 64-bit vector address arithmetic per iteration the register
//     allocator put address temporaries into destination registers of loads still in flight: a write-after-write dependency that made every
//     chunk wait for vmcnt(0) before it could issue its own fetches.)  The range check of the resource returns zeros for rows past M and for
//     column tiles past the padded planes.
//   * As in tgemm_kernel the weight fragment is the MFMA's A operand (C^T = W X^T), so D register e of lane (m, g) is C[row m][column 4 g + e],
//     and the epilogue goes through LDS to row-contiguous 16-byte accesses for bias / activation / act'(H) / beta / store.
#pragma once

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));

struct HGemmArgs {
  const float* A; int lda;                     // [M, K] fp32
  const _Float16 *Bh, *Bl; int ldb, n_pad;     // weight planes [n_pad][ldb], zero padded (n_pad, ldb multiples of 64, ldb >= K)
  float* C; int ldc;                           // [M, N]
  int64_t M; int N, K;
  int bwd;                                     // 0: Y = act(A B^T + bias); 1: C = (beta C + A B^T) * act'(H)
  const float* bias; int act;
  const float* H; int ldh; int act_col0; float beta;
  const float* a_amax;                         // max |A| slot (HG_SLOT floats, hg_slot_read) or NULL (A used as it is)
  float* c_amax;                               // slot that receives max |C| (hg_slot_write) or NULL
};
constexpr int HG_KC = 64, HG_LDS_ROW = HG_KC + 8, HG_LDC = 256 + 4;
constexpr float HG_LO_SCALE = 2048.f, HG_LO_INV = 1.f / 2048.f;
constexpr int HG_BUF_FLAGS = 0x00020000;       // raw buffer, 32-bit elements (gfx9 resource word 3)

// power of two s with s * mx in [2^11, 2^12) (mx = 0 or non-finite: 1)
__device__ __forceinline__ float hg_scale_for(float mx) {
  if (!(mx > 0.f) || !(mx < 3.0e38f)) return 1.f;
  int ex;
  (void)frexpf(mx, &ex);                       // mx = f 2^ex, f in [0.5, 1)
  int sh = 12 - ex;
  sh = sh > 100 ? 100 : (sh < -100 ? -100 : sh);
  return ldexpf(1.f, sh);
}
// A gradient's recorded maximum is HG_SLOT floats (one per workgroup index mod HG_SLOT: thousands of waves hammering one address with atomicMax
// cost 20 us per product).  Writer: one atomicMax per workgroup after a reduction through LDS; reader: the first HG_SLOT threads fetch one
// entry each.  Both need `red` = 16 floats of LDS and are called by every thread of the workgroup.
constexpr int HG_SLOT = 256;
// returns the workgroup's maximum to every thread; slot may be NULL (only the reduction)
__device__ __forceinline__ float hg_slot_write(float* slot, float m, float* red, unsigned wg) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  __syncthreads();
  if (lane == 0) red[wave] = m;
  __syncthreads();
  for (int w = 0; w < nw; ++w) m = fmaxf(m, red[w]);
  if (threadIdx.x == 0 && slot && m > 0.f)
    atomicMax((unsigned int*)slot + (wg & (HG_SLOT - 1)), __float_as_uint(m));                 // non-negative floats order like their bits
  __syncthreads();
  return m;
}
__device__ __forceinline__ float hg_slot_read(const float* slot, float* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float m = threadIdx.x < HG_SLOT ? slot[threadIdx.x] : 0.f;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if (lane == 0 && wave < HG_SLOT / 64) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  return m;
}
// offset (in halfs) of weight element (row n, k) in a fragment-major plane with k padded to kp (a multiple of 64): blocks of 16 rows x 32 k,
// inside a block lane (n & 15) + 16 ((k & 31) >> 3) holds 8 consecutive k
__host__ __device__ inline size_t hg_plane_index(int n, int k, int kp) {
  return ((size_t)(n >> 4) * (kp >> 5) + (k >> 5)) * 512 + (size_t)(((n & 15) + 16 * ((k & 31) >> 3)) * 8 + (k & 7));
}

// PNRF_HG_PROBE (defined by tools/hgemm_probe.hip only): timing builds without the MFMAs (1), the epilogue's stores (2), the A fetches (4), the
// weight fetches (8).  In the library HG_PROBE(x) is the constant false and the parameter has no effect.
#ifdef PNRF_HG_PROBE
#define HG_PROBE(bit) ((PROBE & (bit)) != 0)
#else
#define HG_PROBE(bit) false
#endif
// Shared-memory needs of the product body for 16 MI rows per tile
template <int MI> struct HgShape {
  static constexpr int ROWS = 16 * MI, PLANE = ROWS * HG_LDS_ROW;
  static constexpr int A_BYTES = 2 * 2 * PLANE * 2, C_BYTES = ROWS * HG_LDC * 4, BYTES = A_BYTES + C_BYTES;
};
// bx of gx workgroups share the row tiles of column block by (the launch's grid, or a slice of it when the launch also carries other work)
template <int MI, int PROBE = 0>
__device__ __forceinline__ void hgemm_body(const HGemmArgs& a, const int bx, const int by, const int gx, unsigned char* smem, float* s_red) {
  constexpr int ROWS = 16 * MI, NI = 2;
  constexpr int PLANE = ROWS * HG_LDS_ROW;                     // halfs per plane of one chunk
  unsigned char* const smem_a = smem;
  unsigned char* const smem_c = smem + HgShape<MI>::A_BYTES;
  _Float16* const sA = (_Float16*)smem_a;                      // [buffer][plane hi / lo][row][72]
  float* const sC = (float*)smem_c;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m16 = lane & 15, g = lane >> 4;
  const int tn = by;
  const int col0 = tn * 256 + wave * 32;
  const bool wave_on = col0 < a.N;
  const float a_scale = a.a_amax ? hg_scale_for(hg_slot_read(a.a_amax, s_red)) : 1.f;
  const float inv_scale = 1.f / a_scale;
  const int chunks = (a.K + HG_KC - 1) / HG_KC;
  const int64_t ntiles = (a.M + ROWS - 1) / ROWS;
  const int my_tiles = (int)((ntiles - bx + gx - 1) / gx);   // >= 1: the grid has at most ntiles workgroups
  const int total = my_tiles * chunks;                                                // chunks in this workgroup's pipeline

  f32x4_t accm[MI][NI], accx[MI][NI];
  auto zero_acc = [&] {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) { accm[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f}; accx[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
  };
  zero_acc();

  // ---- A chunk loader: thread t -> 16 bytes at k-offset 4 (t & 15) of rows (t >> 4) + 32 pass.  k past the end of a row reads the floats that
  // follow (zeros past the end of the buffer) and is zeroed when the chunk is written to LDS.  Rows need 4-byte alignment only.
  const int lc4 = 4 * (threadIdx.x & 15), lrow = threadIdx.x >> 4;
  constexpr int PASSES = (ROWS + 31) / 32;
  const bool loader_on = ROWS >= 32 || lrow < ROWS;
  const int64_t a_bytes = ((a.M - 1) * (int64_t)a.lda + a.K) * 4;                      // launch_hgemm refuses operands of 4 GiB and more
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.A, 0, (int)(unsigned)a_bytes, HG_BUF_FLAGS);
  unsigned a_off[PASSES];
#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps) a_off[ps] = (unsigned)(((lrow + 32 * ps) * a.lda + lc4) * 4);
  const unsigned tile_bytes = (unsigned)(ROWS * a.lda * 4), tile_step = tile_bytes * (unsigned)gx;
  // fetch cursor: (byte offset of the tile's first row, chunk in the tile) of the next chunk to request — runs two chunks ahead of the MFMAs,
  // past this workgroup's last tile it points behind the buffer (zeros)
  unsigned f_row = (unsigned)(bx * (size_t)tile_bytes);
  int f_kc = 0, f_left = total;
  auto fetch_a = [&](f32x4_t (&st)[PASSES]) {
    const unsigned soff = f_left > 0 ? f_row + (unsigned)f_kc * (HG_KC * 4) : 0xfffffff0u;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      if (!HG_PROBE(4)) st[ps] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, a_off[ps], soff, 0));
      else st[ps] = f32x4_t{1.f, 2.f, 3.f, (float)f_kc};
    }
    --f_left;
    if (++f_kc == chunks) { f_kc = 0; f_row += tile_step; }
  };
  auto store_a = [&](const f32x4_t (&st)[PASSES], int buf, int kc) {
    _Float16* hi = sA + buf * (2 * PLANE);
    _Float16* lo = hi + PLANE;
    const int k = kc * HG_KC + lc4;
    if (!loader_on) return;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      f16x4_t h, l;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x = k + e < a.K ? st[ps][e] * a_scale : 0.f;
        const _Float16 xh = (_Float16)x;
        h[e] = xh;
        l[e] = (_Float16)((x - (float)xh) * HG_LO_SCALE);
      }
      const int off = (lrow + 32 * ps) * HG_LDS_ROW + lc4;
      *(f16x4_t*)(hi + off) = h;
      *(f16x4_t*)(lo + off) = l;
    }
  };
  // ---- weight fragments of one 32-deep step (NI tiles x 2 planes), fragment-major planes: lane l reads bytes 16 l of its block
  const int ksteps = a.ldb >> 5;                                                        // 32-k blocks per 16-row block of the plane
  const int plane_bytes = a.n_pad * a.ldb * 2;
  const __amdgpu_buffer_rsrc_t h_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.Bh, 0, plane_bytes, HG_BUF_FLAGS);
  const __amdgpu_buffer_rsrc_t l_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.Bl, 0, plane_bytes, HG_BUF_FLAGS);
  unsigned b_off[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) b_off[j] = (unsigned)(((col0 >> 4) + j) * ksteps * 1024 + lane * 16);   // tiles past n_pad: out of range, zeros
  struct WFrag { f16x8_t h[NI], l[NI]; };
  int w_ks = 0;                                                                          // step of the next fragment request, cyclic over the tile's steps
  auto load_w = [&](WFrag& w) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      if (HG_PROBE(8)) { for (int e = 0; e < 8; ++e) { w.h[j][e] = (_Float16)(float)(w_ks + e); w.l[j][e] = (_Float16)(float)(w_ks - e); } continue; }
      w.h[j] = __builtin_bit_cast(f16x8_t, __builtin_amdgcn_raw_buffer_load_b128(h_rsrc, b_off[j], w_ks * 1024, 0));
      w.l[j] = __builtin_bit_cast(f16x8_t, __builtin_amdgcn_raw_buffer_load_b128(l_rsrc, b_off[j], w_ks * 1024, 0));
    }
    if (++w_ks == 2 * chunks) w_ks = 0;
  };
  // activation fragments of one 32-deep step: read from LDS one step before their MFMAs (two register sets)
  struct XFrag { f16x8_t h[MI], l[MI]; };
  auto read_x = [&](XFrag& x, int buf, int st) {
    const _Float16* hi = sA + buf * (2 * PLANE);
    const _Float16* lo = hi + PLANE;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int off = (16 * i + m16) * HG_LDS_ROW + 32 * st + 8 * g;
      x.h[i] = *(const f16x8_t*)(hi + off);
      x.l[i] = *(const f16x8_t*)(lo + off);
    }
  };
  auto mma_step = [&](const WFrag& w, const XFrag& x) {
    if (wave_on && !HG_PROBE(1)) {
      // three sweeps over the tiles, so that the two MFMAs that accumulate into the same cross registers are MI NI MFMAs apart
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) accx[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w.h[j], x.l[i], accx[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) accm[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w.h[j], x.h[i], accm[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) accx[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w.l[j], x.h[i], accx[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- epilogue of one tile, through LDS (its own region: the A buffers keep being filled for the next tile).  launch_hgemm only takes
  // products whose C / H / bias rows are 16-byte aligned and whose N is a multiple of 4.
  // gfx9 has ONE counter for vector-memory loads and stores, and loads and stores may complete out of order with respect to each other: with a
  // store in flight every wait for a load has to be vmcnt(0), i.e. wait for the store's round trip as well (one exposed round trip per stored
  // row was 3.3 us per tile).  So: (1) at the start all prefetches are drained — they are at least a step old — and the compiler's scoreboard
  // learns it from the explicit s_waitcnt; (2) everything the epilogue reads from global memory is requested and waited for before its first
  // store; (3) the stores go out back to back, and the first wait after them belongs to a fetch issued after them, whole steps later.
  const bool use_h = a.bwd && a.act != T_ACT_NONE, use_c = a.bwd && a.beta != 0.f;
  constexpr int QN = ROWS / 8 > 0 ? ROWS / 8 : 1;              // rows per thread: (t >> 6) + 8 q
  const int cl = 4 * (threadIdx.x & 63), c = tn * 256 + cl;
  const bool col_on = c < a.N;
  const int cc = col_on ? c : 0;
  const bool h_on = use_h && col_on && c >= a.act_col0;         // (columns past N must not form an address: the loads are unconditional)
  float amax = 0.f;
  f32x4_t bias4 = {0.f, 0.f, 0.f, 0.f};
  if (!a.bwd && a.bias) bias4 = *(const f32x4_t*)(a.bias + cc);
  auto epilogue = [&](int64_t row0) {
    __builtin_amdgcn_s_waitcnt(0x0F70);                        // vmcnt(0)
    if (wave_on) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const f32x4_t v = (accm[i][j] + accx[i][j] * HG_LO_INV) * inv_scale;
          *(f32x4_t*)(sC + (16 * i + m16) * HG_LDC + wave * 32 + 16 * j + 4 * g) = v;
        }
    }
    zero_acc();
    __syncthreads();
    f32x4_t hv[QN], cv[QN];
    const float* hp[QN]; float* cp[QN];
#pragma unroll
    for (int q = 0; q < QN; ++q) {
      const int rl = (threadIdx.x >> 6) + 8 * q;
      int64_t r = row0 + rl;
      r = r < a.M ? r : a.M - 1;                               // clamped: loads are unconditional, stores are masked
      cp[q] = a.C + r * a.ldc + cc;
      hp[q] = a.H + r * a.ldh + (h_on ? c - a.act_col0 : 0);
    }
    if (use_h) {
#pragma unroll
      for (int q = 0; q < QN; ++q) {
#ifdef PNRF_HG_PROBE_NOH
        hv[q] = f32x4_t{1.f, 1.f, 1.f, 1.f};                   // timing probe only: what the act'(H) rows cost (results are wrong)
#else
        hv[q] = *(const f32x4_t*)hp[q];
#endif
      }
    }
    if (use_c) {
#pragma unroll
      for (int q = 0; q < QN; ++q) cv[q] = *(const f32x4_t*)cp[q];
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                        // vmcnt(0): before the first store
#pragma unroll
    for (int q = 0; q < QN; ++q) {
      const int rl = (threadIdx.x >> 6) + 8 * q;
      f32x4_t v = *(const f32x4_t*)(sC + (rl < ROWS ? rl : 0) * HG_LDC + cl);
      if (!a.bwd) {
        v += bias4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (a.act == T_ACT_RELU) v[e] = fmaxf(v[e], 0.f);
          else if (a.act == T_ACT_ELU) v[e] = v[e] > 0.f ? v[e] : expm1f(v[e]);            // F.elu, alpha = 1
        }
      } else {
        if (use_c) v += cv[q];
        if (h_on) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (a.act == T_ACT_RELU) v[e] = hv[q][e] > 0.f ? v[e] : 0.f;
            else v[e] = hv[q][e] > 0.f ? v[e] : v[e] * (hv[q][e] + 1.f);                     // elu'(z) = elu(z) + 1 for z <= 0
          }
        }
      }
      const bool on = rl < ROWS && row0 + rl < a.M && col_on && !(HG_PROBE(2) && a.K != 12345);
      if (on) {
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        *(f32x4_t*)cp[q] = v;
      }
    }
  };

  // ---- the pipeline.  Chunk q of this workgroup (tile q / chunks, chunk q % chunks of it) has its A planes in LDS buffer q & 1 and the weight
  // fragments of its two steps in register sets 2 q & 3, (2 q + 1) & 3.  Distances, all in flight while MFMAs run:
  //   A rows: requested two chunks ahead (right after the register set is freed in the middle of chunk q - 2), split and written to LDS in the
  //           middle of chunk q - 1, behind that chunk's step-0 MFMAs; ONE barrier per chunk, after that write;
  //   activation fragments: read from LDS one step ahead (step 0 of chunk q during step 1 of chunk q - 1, i.e. right behind the barrier);
  //   weight fragments: requested three steps ahead into the set freed by the previous step.
  // Why one barrier is enough: buffer (q + 1) & 1 is written in the middle of chunk q; its last readers issued their reads in step 0 of chunk
  // q - 1, before they arrived at that chunk's barrier, which the writer has passed.  Unrolled by two chunks: every index is static.
  f32x4_t stg0[PASSES], stg1[PASSES];
  WFrag w0, w1, w2, w3;
  XFrag x0, x1;
  fetch_a(stg0);
  fetch_a(stg1);
  load_w(w0); load_w(w1); load_w(w2);
  store_a(stg0, 0, 0);
  fetch_a(stg0);                                               // chunk 2
  __syncthreads();
  read_x(x0, 0, 0);
  int64_t q_row0 = (int64_t)bx * ROWS;                 // first row of the tile chunk q belongs to
  int q_kc = 0;
  auto next_kc = [&](int kc) { return kc + 1 == chunks ? 0 : kc + 1; };
  auto finish_chunk = [&] {                                     // after the MFMAs of chunk q: the tile's epilogue if that was its last chunk
    if (q_kc + 1 == chunks) { epilogue(q_row0); q_row0 += (int64_t)gx * ROWS; q_kc = 0; }
    else ++q_kc;
  };
  for (int q = 0; q < total; q += 2) {
    // even chunk: A in buffer 0, weight sets 0, 1; chunk q + 1 is in stg1, chunk q + 2 on its way into stg0
    read_x(x1, 0, 1);
    load_w(w3);
    __builtin_amdgcn_sched_barrier(0);
    mma_step(w0, x0);
    store_a(stg1, 1, next_kc(q_kc));
    fetch_a(stg1);                                             // chunk q + 3
    __syncthreads();
    read_x(x0, 1, 0);
    load_w(w0);
    __builtin_amdgcn_sched_barrier(0);
    mma_step(w1, x1);
    finish_chunk();
    if (q + 1 >= total) break;
    // odd chunk: A in buffer 1, weight sets 2, 3
    read_x(x1, 1, 1);
    load_w(w1);
    __builtin_amdgcn_sched_barrier(0);
    mma_step(w2, x0);
    store_a(stg0, 0, next_kc(q_kc));
    fetch_a(stg0);                                             // chunk q + 4
    __syncthreads();
    read_x(x0, 0, 0);
    load_w(w2);
    __builtin_amdgcn_sched_barrier(0);
    mma_step(w3, x1);
    finish_chunk();
  }
  if (a.c_amax) (void)hg_slot_write(a.c_amax, amax, s_red, (unsigned)(bx + gx * by));
}
template <int MI, int PROBE = 0>
__global__ __launch_bounds__(512) void hgemm_kernel(HGemmArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[HgShape<MI>::BYTES];
  __shared__ float s_red[16];
  hgemm_body<MI, PROBE>(a, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x, smem, s_red);
}

// Splits the fp32 parameters of every layer into the fp16 planes hgemm_kernel reads (fragment-major, hg_plane_index): forward orientation
// [out padded][in padded] and backward orientation [in padded][out padded], hi and lo' each; the padding stays zero from the allocation.
// One thread per weight.
struct SplitLayer { size_t w; int in, out, gap; size_t fwd, bwd; int ld_fwd, ld_bwd; size_t plane_fwd, plane_bwd; };   // offsets in floats (w) / halfs
struct SplitArgs { SplitLayer l[26]; int n; const float* P; _Float16* planes; size_t total; float* gapped; };
// (bid, nblk): the block's index and the number of blocks that share the work — the kernel's own grid, or its share of iter_prepare_kernel's
__device__ __forceinline__ void split_weights_body(const SplitArgs& a, unsigned bid, unsigned nblk) {
  const size_t stride = (size_t)nblk * blockDim.x;
  for (size_t i = bid * (size_t)blockDim.x + threadIdx.x; i < a.total; i += stride) {
    int li = 0;
    while (li + 1 < a.n && i >= a.l[li + 1].w) ++li;
    const SplitLayer& L = a.l[li];
    const size_t e = i - L.w;
    if (e >= (size_t)L.in * L.out) continue;                   // bias / padding
    const int o = (int)(e / L.in);
    int k = (int)(e - (size_t)o * L.in);
    const float x = a.P[i];
    if (L.gap >= 0) {                                          // the layer's input rows carry a zero column at `gap`: input index k sits one further
      if (k >= L.gap) ++k;
      a.gapped[(size_t)o * (L.in + 1) + k] = x;                // fp32 copy in that layout for the exact-fp32 kernels
    }
    const _Float16 h = (_Float16)x, l = (_Float16)((x - (float)h) * HG_LO_SCALE);
    const size_t f = L.fwd + hg_plane_index(o, k, L.ld_fwd), b = L.bwd + hg_plane_index(k, o, L.ld_bwd);
    a.planes[f] = h; a.planes[f + L.plane_fwd] = l;
    a.planes[b] = h; a.planes[b + L.plane_bwd] = l;
  }
}

// ------------------------------------------------------------------------------------------ weight gradient, split fp16
// dW[out, in] = dZ^T X over R rows as split-K partials, on the same split-fp16 arithmetic: both operands are activations here (dZ scaled by
// the power of two from its recorded maximum), the contraction runs over rows, so both MFMA operands need 8 consecutive ROWS of one column per
// lane — a transposition of the row-major inputs.  It happens in registers on the way to LDS: a thread fetches the same four columns of four
// consecutive rows (each fetch instruction of a wave still reads whole 512-byte row segments), splits them, and writes one 8-byte column
// piece per column and plane; LDS holds [column][32 rows + 8] fp16 (column stride 80 bytes: the fragment reads of 16 columns cover all banks
// once).
// Workgroup = 8 waves on a 128 x 128 output tile (wave: 64 out x 32 in = 4 x 2 MFMA tiles, two accumulators each) over its split's rows in
// 32-row chunks (one MFMA step), fetched two chunks ahead, LDS double buffered, one barrier per chunk; blockIdx.y = split.  The loaders' column
// sums of dZ are the bias-gradient partials, as in dw_splitk_kernel.
struct DwhArgs {
  const float* dZ; int ldz; const float* X; int ldx;
  float* part; float* db_part;
  int out, in; int64_t R, rows_per;
  const float* dz_amax;                        // max |dZ| slot (HG_SLOT floats) or NULL
};
#ifndef PNRF_DW_PROBE
#define PNRF_DW_PROBE 0          // timing builds only (results are wrong): 1 no MFMAs, 2 no split on the way to LDS
#endif
constexpr int DH_KC = 32, DH_COL = DH_KC + 8;  // rows per chunk; halfs per column in LDS
constexpr int DH_BYTES = 2 * 2 * 2 * 128 * DH_COL * 2;      // LDS of the weight-gradient body
// workgroup (tile bx, split by) of the weight gradient
__device__ __forceinline__ void dwh_body(const DwhArgs& a, const int bx, const int by, unsigned char* smem, float* s_red) {
  constexpr int PLANE = 128 * DH_COL;          // halfs
  _Float16* const sm = (_Float16*)smem;        // [buffer][operand dZ / X][plane hi / lo][column][row]
  const int tiles_n = (a.in + 127) / 128;
  const int tm = bx / tiles_n, tn = bx - tm * tiles_n;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int c16 = lane & 15, g = lane >> 4;
  const int64_t r_begin = by * a.rows_per;
  const int64_t r_end = r_begin + a.rows_per < a.R ? r_begin + a.rows_per : a.R;
  const int chunks = (int)((r_end - r_begin + DH_KC - 1) / DH_KC);
  const float z_scale = a.dz_amax ? hg_scale_for(hg_slot_read(a.dz_amax, s_red)) : 1.f;

  f32x4_t accm[4][2], accx[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) { accm[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f}; accx[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }

  // ---- loader: thread -> operand op, rows 4 rg .. 4 rg + 3 of the chunk, columns 4 cq .. 4 cq + 3 of the tile
  const int op = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));     // uniform per wave: the buffer resource below must live in scalar registers
  const int rg = (threadIdx.x >> 5) & 7, cq = threadIdx.x & 31;
  const float* src = op == 0 ? a.dZ : a.X;
  const int ld = op == 0 ? a.ldz : a.ldx, width = op == 0 ? a.out : a.in;
  const int col = (op == 0 ? tm : tn) * 128 + 4 * cq;
  const float scale = op == 0 ? z_scale : 1.f;
  const int64_t span = ((r_end - r_begin - 1) * (int64_t)ld + width) * 4;                 // bytes of this split's rows: past them the fetches return zeros
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)(src + r_begin * ld), 0, (int)(unsigned)(span < 0xffffffffll ? span : 0xffffffffll), HG_BUF_FLAGS);
  const unsigned voff = (unsigned)((4 * rg * ld + col) * 4);                              // the thread's first row; rows + 1 .. + 3 through the scalar offset
  const unsigned chunk_bytes = (unsigned)(DH_KC * ld * 4), row_bytes = (unsigned)(ld * 4);
  bool col_on[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) col_on[e] = col + e < width;
  auto fetch = [&](f32x4_t (&st)[4], int kc) {
    const unsigned soff = kc < chunks ? (unsigned)kc * chunk_bytes : 0xffff0000u;
#pragma unroll
    for (int k = 0; k < 4; ++k) st[k] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff + k * row_bytes, 0));
  };
  f32x4_t colsum = {0.f, 0.f, 0.f, 0.f};
  // 4-row pieces of a column are stored at position rg ^ 2 ((column >> 4) & 3): the 32 lanes that write the same rows of 32 column quads then
  // spread over 16 bank pairs instead of 4 (an even swizzle: the two pieces of a fragment's 16 bytes stay adjacent and in order)
  const int sm_off = 4 * cq * DH_COL + 4 * (rg ^ (2 * ((cq >> 2) & 3)));
  auto store = [&](const f32x4_t (&st)[4], int buf) {
    _Float16* hi = sm + ((buf * 2 + op) * 2) * PLANE + sm_off;
    _Float16* lo = hi + PLANE;
#if PNRF_DW_PROBE & 2                                           // timing build: the fetched bits go to LDS as they are (no split, no column sums)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      *(f16x4_t*)(hi + e * DH_COL) = __builtin_bit_cast(f16x4_t, ((unsigned long long)__float_as_uint(st[0][e]) << 32 | __float_as_uint(st[1][e])));
      *(f16x4_t*)(lo + e * DH_COL) = __builtin_bit_cast(f16x4_t, ((unsigned long long)__float_as_uint(st[2][e]) << 32 | __float_as_uint(st[3][e])));
    }
    return;
#endif
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      f16x4_t h, l;
      float cs = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float raw = col_on[e] ? st[k][e] : 0.f;
        cs += raw;
        const float x = raw * scale;
        const _Float16 xh = (_Float16)x;
        h[k] = xh;
        l[k] = (_Float16)((x - (float)xh) * HG_LO_SCALE);
      }
      colsum[e] += cs;
      *(f16x4_t*)(hi + e * DH_COL) = h;
      *(f16x4_t*)(lo + e * DH_COL) = l;
    }
  };
  // fragment of 16 consecutive columns starting at a multiple of 16: all share the swizzle; rows 8 g .. 8 g + 7 = pieces 2 g, 2 g + 1
  auto frag_off = [&](int col0) { return (col0 + c16) * DH_COL + 8 * (g ^ ((col0 >> 4) & 3)); };
  auto mma_chunk = [&](int buf) {
    const _Float16* zh = sm + ((buf * 2 + 0) * 2) * PLANE;
    const _Float16* zl = zh + PLANE;
    const _Float16* xh = sm + ((buf * 2 + 1) * 2) * PLANE;
    const _Float16* xl = xh + PLANE;
    f16x8_t ah[4], al[4], bh[2], bl[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int o = frag_off(wm * 64 + 16 * i); ah[i] = *(const f16x8_t*)(zh + o); al[i] = *(const f16x8_t*)(zl + o); }
#pragma unroll
    for (int j = 0; j < 2; ++j) { const int o = frag_off(wn * 32 + 16 * j); bh[j] = *(const f16x8_t*)(xh + o); bl[j] = *(const f16x8_t*)(xl + o); }
#if PNRF_DW_PROBE & 1                                           // timing build: the fragments are read, one MFMA per chunk keeps them alive
    accx[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[0] + ah[1] + ah[2] + ah[3] + al[0] + al[1] + al[2] + al[3], bh[0] + bh[1] + bl[0] + bl[1], accx[0][0], 0, 0, 0);
    return;
#endif
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) accx[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], accx[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) accm[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[j], accm[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) accx[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], accx[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };

  // chunk q in LDS buffer q & 1; chunk q + 1 in one register set, chunk q + 2 on its way into the other (cf. hgemm_kernel)
  f32x4_t stg0[4], stg1[4];
  fetch(stg0, 0);
  fetch(stg1, 1);
  store(stg0, 0);
  fetch(stg0, 2);
  __syncthreads();
  for (int q = 0; q < chunks; q += 2) {
    mma_chunk(0);
    store(stg1, 1);                                            // past the last chunk: zeros
    fetch(stg1, q + 3);
    __syncthreads();
    if (q + 1 >= chunks) break;
    mma_chunk(1);
    store(stg0, 0);
    fetch(stg0, q + 4);
    __syncthreads();
  }

  // ---- partial tile: D register e of lane (c16, g) = dW[out 4 g + e][in c16] of its MFMA tile
  const float inv = HG_LO_INV, unscale = 1.f / z_scale;
  float* p = a.part + (size_t)by * a.out * a.in;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = tm * 128 + wm * 64 + 16 * i + 4 * g + e, n = tn * 128 + wn * 32 + 16 * j + c16;
        if (m < a.out && n < a.in) p[(size_t)m * a.in + n] = (accm[i][j][e] + accx[i][j][e] * inv) * unscale;
      }
  if (a.db_part && tn == 0) {
    __syncthreads();                                           // every wave is done reading the staging buffers
    float* red = (float*)sm;                                   // [8 row groups][128 columns]
    if (op == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) red[rg * 128 + 4 * cq + e] = colsum[e];
    }
    __syncthreads();
    if (threadIdx.x < 128 && tm * 128 + (int)threadIdx.x < a.out) {
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) sum += red[r * 128 + threadIdx.x];
      a.db_part[(size_t)by * a.out + tm * 128 + threadIdx.x] = sum;
    }
  }
}
// ---- the same product on a 256 x 128 output tile (round 5): the workgroup reads ALL 256 columns of dZ and 128 columns of X, so a 256 x 256 gradient is two
// tiles per split and dZ is the only operand fetched twice (1.5x the operand bytes at L2 instead of 2x with four 128 x 128 tiles; FETCH_SIZE of the grouped
// launch at 262 144 rows was 6.6 GB for 5.13 GB of operands).  Wave (wm, wn) = 64 dZ columns x 64 X columns = 4 x 4 MFMA tiles x two accumulators = 128
// registers; the X fragments come in two halves per chunk; ONE register set of fetched rows per operand, one chunk ahead of the chunk being stored.
// LDS per buffer: dZ [hi | lo][256][40] + X [hi | lo][128][40] halfs = 60 KiB, two buffers.  Loader: every thread a 4-row x 4-column piece of dZ (8 row
// groups x 64 column quads) and a 2-row x 4-column piece of X (16 row pairs x 32 column quads); same column-major layout and swizzle as dwh_body.
constexpr int DHW_ZP = 256 * DH_COL, DHW_XP = 128 * DH_COL, DHW_BUF = 2 * DHW_ZP + 2 * DHW_XP;      // halfs
constexpr int DHW_BYTES = 2 * DHW_BUF * 2;                                                             // 122 880
__device__ __forceinline__ void dwh_body_wide(const DwhArgs& a, const int bx, const int by, unsigned char* smem, float* s_red) {
  _Float16* const sm = (_Float16*)smem;
  const int tn = bx;                                                  // a.out == 256: one tile row
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int c16 = lane & 15, g = lane >> 4;
  const int64_t r_begin = by * a.rows_per;
  const int64_t r_end = r_begin + a.rows_per < a.R ? r_begin + a.rows_per : a.R;
  const int chunks = (int)((r_end - r_begin + DH_KC - 1) / DH_KC);
  const float z_scale = a.dz_amax ? hg_scale_for(hg_slot_read(a.dz_amax, s_red)) : 1.f;

  f32x4_t accm[4][4], accx[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) { accm[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f}; accx[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }

  // ---- loader
  const int zrg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), zcq = threadIdx.x & 63;      // dZ: rows 4 zrg .. + 3 of the chunk, columns 4 zcq .. + 3
  const int xrp = threadIdx.x >> 5, xcq = threadIdx.x & 31;                                              // X: rows 2 xrp, 2 xrp + 1, columns 4 xcq .. + 3 of the tile
  const int xcol = tn * 128 + 4 * xcq;
  const int64_t zspan = ((r_end - r_begin - 1) * (int64_t)a.ldz + a.out) * 4, xspan = ((r_end - r_begin - 1) * (int64_t)a.ldx + a.in) * 4;
  const __amdgpu_buffer_rsrc_t zrs =
      __builtin_amdgcn_make_buffer_rsrc((void*)(a.dZ + r_begin * a.ldz), 0, (int)(unsigned)(zspan < 0xffffffffll ? zspan : 0xffffffffll), HG_BUF_FLAGS);
  const __amdgpu_buffer_rsrc_t xrs =
      __builtin_amdgcn_make_buffer_rsrc((void*)(a.X + r_begin * a.ldx), 0, (int)(unsigned)(xspan < 0xffffffffll ? xspan : 0xffffffffll), HG_BUF_FLAGS);
  const unsigned zvoff = (unsigned)((4 * zrg * a.ldz + 4 * zcq) * 4), xvoff = (unsigned)((2 * xrp * a.ldx + xcol) * 4);
  const unsigned zchunk = (unsigned)(DH_KC * a.ldz * 4), zrow = (unsigned)(a.ldz * 4), xchunk = (unsigned)(DH_KC * a.ldx * 4), xrow = (unsigned)(a.ldx * 4);
  bool xon[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) xon[e] = xcol + e < a.in;
  f32x4_t stz[4], stx[2];
  auto fetch = [&](int kc) {
    const bool in = kc < chunks;
    const unsigned zs = in ? (unsigned)kc * zchunk : 0xffff0000u, xs = in ? (unsigned)kc * xchunk : 0xffff0000u;
#pragma unroll
    for (int k = 0; k < 4; ++k) stz[k] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(zrs, zvoff, zs + k * zrow, 0));
#pragma unroll
    for (int k = 0; k < 2; ++k) stx[k] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(xrs, xvoff, xs + k * xrow, 0));
  };
  f32x4_t colsum = {0.f, 0.f, 0.f, 0.f};
  const int z_off = 4 * zcq * DH_COL + 4 * (zrg ^ (2 * ((zcq >> 2) & 3)));
  const int x_off = 4 * xcq * DH_COL + 4 * ((xrp >> 1) ^ (2 * ((xcq >> 2) & 3))) + 2 * (xrp & 1);
  auto store = [&](int buf) {
    _Float16* zh = sm + buf * DHW_BUF + z_off;
    _Float16* zl = zh + DHW_ZP;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      f16x4_t h, l;
      float cs = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float raw = stz[k][e];
        cs += raw;
        const float x = raw * z_scale;
        const _Float16 xh = (_Float16)x;
        h[k] = xh;
        l[k] = (_Float16)((x - (float)xh) * HG_LO_SCALE);
      }
      colsum[e] += cs;
      *(f16x4_t*)(zh + e * DH_COL) = h;
      *(f16x4_t*)(zl + e * DH_COL) = l;
    }
    _Float16* xh = sm + buf * DHW_BUF + 2 * DHW_ZP + x_off;
    _Float16* xl = xh + DHW_XP;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      f16x2_t h, l;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float x = xon[e] ? stx[k][e] : 0.f;
        const _Float16 hh = (_Float16)x;
        h[k] = hh;
        l[k] = (_Float16)((x - (float)hh) * HG_LO_SCALE);
      }
      *(f16x2_t*)(xh + e * DH_COL) = h;
      *(f16x2_t*)(xl + e * DH_COL) = l;
    }
  };
  auto frag_off = [&](int col0) { return (col0 + c16) * DH_COL + 8 * (g ^ ((col0 >> 4) & 3)); };
  auto mma_chunk = [&](int buf) {
    const _Float16* zh = sm + buf * DHW_BUF;
    const _Float16* zl = zh + DHW_ZP;
    const _Float16* xh = zh + 2 * DHW_ZP;
    const _Float16* xl = xh + DHW_XP;
    f16x8_t ah[4], al[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int o = frag_off(wm * 64 + 16 * i); ah[i] = *(const f16x8_t*)(zh + o); al[i] = *(const f16x8_t*)(zl + o); }
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      f16x8_t bh[2], bl[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) { const int o = frag_off(wn * 64 + 32 * jj + 16 * j); bh[j] = *(const f16x8_t*)(xh + o); bl[j] = *(const f16x8_t*)(xl + o); }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) accx[i][2 * jj + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], accx[i][2 * jj + j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) accm[i][2 * jj + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[j], accm[i][2 * jj + j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) accx[i][2 * jj + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], accx[i][2 * jj + j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  // chunk q in LDS buffer q & 1, chunk q + 1 in the register set, fetched while chunk q - 1 was multiplied
  fetch(0);
  store(0);
  fetch(1);
  __syncthreads();
  for (int q = 0; q < chunks; ++q) {
    mma_chunk(q & 1);
    store((q + 1) & 1);                                          // past the last chunk: zeros
    fetch(q + 2);
    __syncthreads();
  }

  const float inv = HG_LO_INV, unscale = 1.f / z_scale;
  float* p = a.part + (size_t)by * a.out * a.in;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = wm * 64 + 16 * i + 4 * g + e, n = tn * 128 + wn * 64 + 16 * j + c16;
        if (n < a.in) p[(size_t)m * a.in + n] = (accm[i][j][e] + accx[i][j][e] * inv) * unscale;
      }
  if (a.db_part && tn == 0) {
    __syncthreads();
    float* red = (float*)sm;                                     // [8 row groups][256 columns]
#pragma unroll
    for (int e = 0; e < 4; ++e) red[zrg * 256 + 4 * zcq + e] = colsum[e];
    __syncthreads();
    if (threadIdx.x < 256) {
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) sum += red[r * 256 + threadIdx.x];
      a.db_part[(size_t)by * a.out + threadIdx.x] = sum;
    }
  }
}
__global__ __launch_bounds__(512) void dwh_kernel(DwhArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[DH_BYTES];
  __shared__ float s_red[16];
  dwh_body(a, (int)blockIdx.x, (int)blockIdx.y, smem, s_red);
}

// One launch for a layer's backward pass: the first n_dx workgroups form the persistent grid (gx x gy) of the input-gradient product, the
// others are the (tile, split) workgroups of the weight gradient.  The two read the same dZ and are independent: dispatched together, the
// weight gradient fills the CUs the product leaves idle (the 4096-row nets: 13 + 10 us as two launches).
template <int MI>
__global__ __launch_bounds__(512) void layer_bwd_kernel(HGemmArgs g, DwhArgs d, int gx, int gy, int dw_tiles) {
  constexpr int BYTES = HgShape<MI>::BYTES > DH_BYTES ? HgShape<MI>::BYTES : DH_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[BYTES];
  __shared__ float s_red[16];
  const int b = (int)blockIdx.x, n_dx = gx * gy;
  if (b < n_dx) {
    hgemm_body<MI, 0>(g, b % gx, b / gx, gx, smem, s_red);
  } else {
    const int w = b - n_dx;
    dwh_body(d, w % dw_tiles, w / dw_tiles, smem, s_red);
  }
}

// ------------------------------------------------------------------------------------------ grouped weight gradients
// The weight gradients of several layers in one launch (their dZ / X buffers must all still exist): blocks [first[j], first[j + 1]) belong to
// job j as (tile, split) = (b % tiles, b / tiles).
constexpr int DH_GROUP_MAX = 24;
struct DwhGroupArgs { DwhArgs j[DH_GROUP_MAX]; int first[DH_GROUP_MAX + 1]; int tiles[DH_GROUP_MAX]; int splits[DH_GROUP_MAX]; int n; unsigned wide; };   // wide: bit j = job j runs dwh_body_wide
// Workgroups go to the XCDs round robin (blockIdx % 8) and every tile of a split reads the split's rows — of X or of dZ — again.  Jobs start at
// multiples of 8 and, where the split count is one too, block b of a job is (tile (b / 8) % tiles, split b % 8 + 8 (b / (8 tiles))): the tiles of a
// split run on ONE XCD at about the same time and its L2 fetches the rows once (at 262 144 rows the grouped launch is bound by HBM; FETCH_SIZE
// said 10.1 GB per launch for 5.4 GB of operands).
#ifndef PNRF_DWG_PLAIN
#define PNRF_DWG_PLAIN 0
#endif
__global__ __launch_bounds__(512) void dwh_group_kernel(DwhGroupArgs g) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[DHW_BYTES > DH_BYTES ? DHW_BYTES : DH_BYTES];
  __shared__ float s_red[16];
  int j = 0;
  while (j + 1 < g.n && (int)blockIdx.x >= g.first[j + 1]) ++j;
  const int b = (int)blockIdx.x - g.first[j], tiles = g.tiles[j], splits = g.splits[j];
  if (b >= tiles * splits) return;                             // padding up to the next multiple of 8
  int tile = b % tiles, split = b / tiles;
  if (!PNRF_DWG_PLAIN && splits % 8 == 0) { const int q = b >> 3; tile = q % tiles; split = (b & 7) + 8 * (q / tiles); }
  if ((g.wide >> j) & 1u) dwh_body_wide(g.j[j], tile, split, smem, s_red);
  else dwh_body(g.j[j], tile, split, smem, s_red);
}

// ------------------------------------------------------------------------------------------ layer chains of the 4096-row nets, handed over in LDS
// The sampler / refine nets see one batch of rays (4096 rows): a layer product is 4 MB in, 4 MB out, and as a launch of its own it costs 7.6 us —
// 3.7 us before the first useful instruction plus the exposed latencies of a pipeline that is only four chunks long.  Rows are independent
// through the layers, so one launch walks a whole chain: workgroup b owns rows 16 b .. 16 b + 15 in every layer (grid = number of 16-row tiles).
// Walking the layers with hgemm_body alone (stores drained, the output fetched back from L2, the weight pipeline restarted per layer) still
// cost ~7 us per layer; for the 256 -> 256 layers the hand-over stays on chip instead: the epilogue that stores a layer's rows (they are
// needed for the backward pass / the weight gradients) also splits them and writes the next layer's fp16 operand planes into LDS —
// 16 rows x 256 k x 2 planes = 17 KB, two such regions — so a layer is: eight MFMA steps fed from LDS and from a weight stream that is
// prefetched three steps ahead ACROSS the layer boundary, plus one epilogue.  An optional first layer of another input width runs through
// hgemm_body and its output is fetched once.  Backward chains prefetch the rows of ELU'(h) at the start of each layer and take the scale of the
// next product from the maximum over the workgroup's own 16 rows.
constexpr int RC_LDX = 256 + 8;                                // halfs per row of a resident operand plane
constexpr int HG_CHAIN_MAX = 6;
// per resident layer: act = T_ACT_ELU / T_ACT_RELU / T_ACT_NONE (forward: the layer's activation; backward: the activation whose derivative
// is taken at H); ldc / ldh = row strides of C and H (floats, multiples of 4; 0 = 256)
struct RChainLayer { const _Float16 *Bh, *Bl; int ldb, n_pad; const float* bias; float* C; const float* H; float* c_amax; int act, ldc, ldh; };
struct RChainArgs {
  HGemmArgs first; int has_first;                              // optional leading layer through hgemm_body (forward: 288 / 144 / 63 / 320 -> 256)
  const float* X0; const float* x0_amax; int ldx0;             // [M, 256] input rows of the first resident layer (row stride ldx0, 0 = 256); its max-|.| slot or NULL
  RChainLayer l[HG_CHAIN_MAX]; int n;
  int64_t M; int bwd;                                          // forward: C = act(X W^T + bias); backward: C = (X W^T) * act'(H)
};
// MI: 16-row MFMA tiles per workgroup.  Every workgroup reads all 256 KB of a layer's weight planes from L2: with 16-row tiles (256 workgroups
// for 4096 rows) that is 64 MB per layer — 6 us of L2 bandwidth; 32-row tiles halve it and still give half of the CUs a workgroup.
// MI = 4 (64-row tiles, the NeRF layers' 32 768 .. 1 M rows: a quarter of the weight traffic of 16-row tiles) keeps ONE operand region and
// rewrites it in place — the next layer's planes are written behind the barrier that follows the last MFMA of the current one, so nobody
// reads the old planes any more — which brings the workgroup to 134 KB of LDS; MI <= 2 alternate between two regions as before.
// PNRF_RC_PROBE (timing builds only; results are wrong): 1 no MFMA steps, 2 no stores, 4 no weight fetches
#ifndef PNRF_RC_PROBE
#define PNRF_RC_PROBE 0
#endif
template <int MI>
__global__ __launch_bounds__(512) void hgemm_rchain_kernel(RChainArgs c) {
  constexpr int ROWS = 16 * MI, QN = 2 * MI;
  constexpr int XPLANE = ROWS * RC_LDX;                        // halfs per plane
  constexpr int REGIONS = MI >= 4 ? 1 : 2;
  constexpr int XBYTES = REGIONS * 2 * XPLANE * 2;
  __shared__ __attribute__((aligned(16))) unsigned char smem[HgShape<MI>::BYTES > XBYTES + ROWS * HG_LDC * 4 ? HgShape<MI>::BYTES : XBYTES + ROWS * HG_LDC * 4];
  __shared__ float s_red[16];
  _Float16* const sX = (_Float16*)smem;                        // [region][plane hi / lo][ROWS][RC_LDX]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m16 = lane & 15, g = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * ROWS;
  if (c.has_first) {
    hgemm_body<MI, 0>(c.first, (int)blockIdx.x, 0, (int)gridDim.x, smem, s_red);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
  }
  // thread t <-> rows (t >> 6) + 8 q (q < QN), columns 4 (t & 63) .. + 3: the mapping of the hand-over and of the epilogue
  const int cl = 4 * (threadIdx.x & 63), rl0 = threadIdx.x >> 6;
  int64_t rr[QN];
#pragma unroll
  for (int q = 0; q < QN; ++q) { const int64_t r = row0 + rl0 + 8 * q; rr[q] = r < c.M ? r : c.M - 1; }
  int64_t rrow[MI];                                            // rows of this lane's accumulator tiles
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) { const int64_t r = row0 + 16 * mi + m16; rrow[mi] = r < c.M ? r : c.M - 1; }
  float in_scale = 1.f;
  {
    if (c.x0_amax) in_scale = hg_scale_for(hg_slot_read(c.x0_amax, s_red));
    const int ldx0 = c.ldx0 ? c.ldx0 : 256;
#pragma unroll
    for (int q = 0; q < QN; ++q) {
      const f32x4_t x = *(const f32x4_t*)(c.X0 + rr[q] * ldx0 + cl);
      f16x4_t h, l;
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float v = x[e] * in_scale; const _Float16 vh = (_Float16)v; h[e] = vh; l[e] = (_Float16)((v - (float)vh) * HG_LO_SCALE); }
      const int off = (rl0 + 8 * q) * RC_LDX + cl;
      *(f16x4_t*)(sX + off) = h;
      *(f16x4_t*)(sX + XPLANE + off) = l;
    }
  }
  // weight stream: step s of layer i = global step 8 i + s
  struct WFrag { f16x8_t h[2], l[2]; };
  int w_layer = 0, w_ks = 0;
  auto load_w = [&](WFrag& w) {
    if (!(PNRF_RC_PROBE & 4) && w_layer < c.n) {
      const RChainLayer& L = c.l[w_layer];
      const int plane_bytes = L.n_pad * L.ldb * 2;
      const __amdgpu_buffer_rsrc_t hr = __builtin_amdgcn_make_buffer_rsrc((void*)L.Bh, 0, plane_bytes, HG_BUF_FLAGS);
      const __amdgpu_buffer_rsrc_t lr = __builtin_amdgcn_make_buffer_rsrc((void*)L.Bl, 0, plane_bytes, HG_BUF_FLAGS);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const unsigned off = (unsigned)(((wave * 2 + j) * (L.ldb >> 5)) * 1024 + lane * 16);
        w.h[j] = __builtin_bit_cast(f16x8_t, __builtin_amdgcn_raw_buffer_load_b128(hr, off, w_ks * 1024, 0));
        w.l[j] = __builtin_bit_cast(f16x8_t, __builtin_amdgcn_raw_buffer_load_b128(lr, off, w_ks * 1024, 0));
      }
    }
    if (++w_ks == 8) { w_ks = 0; ++w_layer; }
  };
#ifdef PNRF_RC_PREFETCH3
  WFrag w0, w1, w2, w3;
  load_w(w0); load_w(w1); load_w(w2);
#else
  // eight register sets = one layer of this wave's fragments: a set is refilled with the NEXT layer's step the moment its step has run, so a
  // layer's weights have the whole previous layer (steps + epilogue) to arrive.  (Three steps ahead — 0.2 us of MFMA steps — every step waited
  // out most of an L2 round trip: a layer took 6.3 us for 0.6 us of MFMAs.)
  WFrag w[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) load_w(w[s]);
#endif
  __syncthreads();                                             // the first resident operand planes are complete
  for (int i = 0; i < c.n; ++i) {
    const RChainLayer& L = c.l[i];
    const _Float16* xh = sX + (i & (REGIONS - 1)) * 2 * XPLANE;
    const _Float16* xl = xh + XPLANE;
    const int ldc = L.ldc ? L.ldc : 256, ldh = L.ldh ? L.ldh : 256;
    const bool use_h = c.bwd && L.act != T_ACT_NONE;
    f32x4_t hv[MI][2];                                         // act'(H) of this lane's accumulator elements (requested now, used in the epilogue)
    if (use_h) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int j = 0; j < 2; ++j) hv[mi][j] = *(const f32x4_t*)(L.H + rrow[mi] * ldh + wave * 32 + 16 * j + 4 * g);
    }
    f32x4_t accm[MI][2], accx[MI][2];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int j = 0; j < 2; ++j) { accm[mi][j] = f32x4_t{0.f, 0.f, 0.f, 0.f}; accx[mi][j] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
    auto step = [&](const WFrag& w, int s) {
      if (PNRF_RC_PROBE & 1) return;                             // timing probe: no MFMA steps
      f16x8_t ah[MI], al[MI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int off = (16 * mi + m16) * RC_LDX + 32 * s + 8 * g;
        ah[mi] = *(const f16x8_t*)(xh + off); al[mi] = *(const f16x8_t*)(xl + off);
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          accx[mi][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w.h[j], al[mi], accx[mi][j], 0, 0, 0);
          accm[mi][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w.h[j], ah[mi], accm[mi][j], 0, 0, 0);
        }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int j = 0; j < 2; ++j) accx[mi][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w.l[j], ah[mi], accx[mi][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    };
#ifdef PNRF_RC_PREFETCH3
    load_w(w3); step(w0, 0); load_w(w0); step(w1, 1); load_w(w1); step(w2, 2); load_w(w2); step(w3, 3);
    load_w(w3); step(w0, 4); load_w(w0); step(w1, 5); load_w(w1); step(w2, 6); load_w(w2); step(w3, 7);
#else
#pragma unroll
    for (int s = 0; s < 8; ++s) { step(w[s], s); load_w(w[s]); }
#endif
    // ---- epilogue from the accumulators: D register e of lane (m16, g) of tile (mi, j) = C[row 16 mi + m16][column 32 wave + 16 j + 4 g + e] —
    // 16 contiguous bytes of a row per lane: bias / activation / act'(H) / store / split into the next layer's planes without an LDS round
    // trip (through LDS, with the row-contiguous thread mapping, the epilogue was 3.4 of a layer's 6.2 us: three barriers more per layer).
    // All fetches in flight (the next layer's weight fragments, the rows of H) are drained before the first store (cf. hgemm_body: one counter
    // for loads and stores).
    __builtin_amdgcn_s_waitcnt(0x0F70);
    const float inv = 1.f / in_scale;
    f32x4_t v[MI][2];
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = wave * 32 + 16 * j + 4 * g;
      const f32x4_t b4 = !c.bwd && L.bias ? *(const f32x4_t*)(L.bias + col) : f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        f32x4_t x = (accm[mi][j] + accx[mi][j] * HG_LO_INV) * inv + b4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (!c.bwd) {
            if (L.act == T_ACT_ELU) x[e] = x[e] > 0.f ? x[e] : expm1f(x[e]);
            else if (L.act == T_ACT_RELU) x[e] = fmaxf(x[e], 0.f);
          } else if (use_h) {
            if (L.act == T_ACT_ELU) x[e] = hv[mi][j][e] > 0.f ? x[e] : x[e] * (hv[mi][j][e] + 1.f);
            else x[e] = hv[mi][j][e] > 0.f ? x[e] : 0.f;
          }
        }
        v[mi][j] = x;
        if (row0 + 16 * mi + m16 < c.M) {
          amax = fmaxf(fmaxf(amax, fmaxf(fabsf(x[0]), fabsf(x[1]))), fmaxf(fabsf(x[2]), fabsf(x[3])));
          if (!(PNRF_RC_PROBE & 2)) *(f32x4_t*)(L.C + rrow[mi] * ldc + col) = x;
        }
      }
    }
    in_scale = 1.f;
    if (c.bwd) in_scale = hg_scale_for(hg_slot_write(L.c_amax, amax, s_red, blockIdx.x));   // the workgroup's own maximum
    if (i + 1 < c.n) {
      if (REGIONS == 1) __syncthreads();                       // one region: every wave is done reading this layer's planes
      _Float16* nh = sX + ((i + 1) & (REGIONS - 1)) * 2 * XPLANE;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f16x4_t h, l;
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float x = v[mi][j][e] * in_scale; const _Float16 xh_ = (_Float16)x; h[e] = xh_; l[e] = (_Float16)((x - (float)xh_) * HG_LO_SCALE); }
          const int off = (16 * mi + m16) * RC_LDX + wave * 32 + 16 * j + 4 * g;
          *(f16x4_t*)(nh + off) = h;
          *(f16x4_t*)(nh + XPLANE + off) = l;
        }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------ layer chains of the wide (NeRF) layers, register epilogue
// hgemm_rchain_kernel at 64-row tiles holds ONE workgroup per CU (134 KB of LDS) and runs MFMA steps, the epilogue's LDS round trip, the store
// and the hand-over one after the other: measured 2 % slower than one persistent product per layer.  This kernel is shaped for TWO workgroups
// per CU, so that one's epilogue overlaps the other's MFMAs:
//   * workgroup = 4 waves on 32 rows x 256 columns; wave w owns columns 64 w .. 64 w + 63 (four 16-column MFMA tiles) of both 16-row tiles:
//     2 x 4 x 2 accumulators = 64 registers, four rotating sets of weight fragments (4 tiles x 2 planes) = 128, <= 256 in all: 2 waves per SIMD;
//   * LDS = the two fp16 operand planes of its 32 rows only (33 KB, rewritten in place behind the barrier that follows the layer's last
//     MFMA): the epilogue runs from the accumulator registers — D register e of lane (m, g) is C[row m][column 4 g + e], i.e. four consecutive
//     columns: one float4 of bias, one 16-byte store per tile (a wave instruction covers 16 rows x 64 bytes) and two 8-byte LDS writes
//     (hi, lo planes) for the hand-over;
//   * persistent over the row tiles (grid = 2 x CUs); the weight stream runs three steps ahead across layer and tile boundaries.
// Forward (bwd = 0): C = act(X W^T + bias).  Backward (bwd = 1): C = (X W^T) * act'(H), X scaled by a power of two from its recorded maximum
// (first layer) / from the workgroup's own maximum (later layers), max |C| left in the layer's slot.  Same products in the same order as
// hgemm_body: bit-identical results.
template <int MI = 4>
__global__ __launch_bounds__(256, 2) void hgemm_wchain_kernel(RChainArgs c) {
  constexpr int NI = 4, ROWS = 16 * MI, QL = ROWS / 4;
  constexpr int XPLANE = ROWS * RC_LDX;                        // halfs per plane
  __shared__ __attribute__((aligned(16))) _Float16 sX[2 * XPLANE];   // [plane hi / lo][ROWS][RC_LDX]
  __shared__ float s_red[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m16 = lane & 15, g = lane >> 4;
  const int64_t ntiles = (c.M + ROWS - 1) / ROWS;
  const int ldx0 = c.ldx0 ? c.ldx0 : 256;
  // weight stream: step s of layer i of tile t = global step (t n + i) 8 + s; two register sets, one step (48 MFMAs per wave, two waves
  // per SIMD: ~1500 cycles) ahead of the MFMAs that consume them
  struct WFrag { f16x8_t h[NI], l[NI]; };
  int w_layer = 0, w_ks = 0;
  auto load_w = [&](WFrag& w) {
    const RChainLayer& L = c.l[w_layer];
    const int plane_bytes = L.n_pad * L.ldb * 2;
    const __amdgpu_buffer_rsrc_t hr = __builtin_amdgcn_make_buffer_rsrc((void*)L.Bh, 0, plane_bytes, HG_BUF_FLAGS);
    const __amdgpu_buffer_rsrc_t lr = __builtin_amdgcn_make_buffer_rsrc((void*)L.Bl, 0, plane_bytes, HG_BUF_FLAGS);
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const unsigned off = (unsigned)(((wave * NI + j) * (L.ldb >> 5)) * 1024 + lane * 16);
      w.h[j] = __builtin_bit_cast(f16x8_t, __builtin_amdgcn_raw_buffer_load_b128(hr, off, w_ks * 1024, 0));
      w.l[j] = __builtin_bit_cast(f16x8_t, __builtin_amdgcn_raw_buffer_load_b128(lr, off, w_ks * 1024, 0));
    }
    if (++w_ks == 8) { w_ks = 0; if (++w_layer == c.n) w_layer = 0; }
  };
  WFrag w0, w1;
  load_w(w0);
  const float x0_scale = c.x0_amax ? hg_scale_for(hg_slot_read(c.x0_amax, s_red)) : 1.f;
  // loader of a tile's first operand planes: thread t -> rows (t >> 6) + 4 q (q < QL), columns 4 (t & 63) .. + 3
  const int cl = 4 * (threadIdx.x & 63), rl0 = threadIdx.x >> 6;
  float amax_l[HG_CHAIN_MAX];
#pragma unroll
  for (int i = 0; i < HG_CHAIN_MAX; ++i) amax_l[i] = 0.f;

  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t row0 = tile * ROWS;
    float in_scale = x0_scale;
    {
      f32x4_t xin[QL];
#pragma unroll
      for (int q = 0; q < QL; ++q) {
        int64_t r = row0 + rl0 + 4 * q;
        r = r < c.M ? r : c.M - 1;
        xin[q] = *(const f32x4_t*)(c.X0 + r * ldx0 + cl);
      }
      __syncthreads();                                         // the previous tile's last layer has read the planes
#pragma unroll
      for (int q = 0; q < QL; ++q) {
        f16x4_t h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float v = xin[q][e] * in_scale; const _Float16 vh = (_Float16)v; h[e] = vh; l[e] = (_Float16)((v - (float)vh) * HG_LO_SCALE); }
        const int off = (rl0 + 4 * q) * RC_LDX + cl;
        *(f16x4_t*)(sX + off) = h;
        *(f16x4_t*)(sX + XPLANE + off) = l;
      }
      __syncthreads();
    }
    for (int i = 0; i < c.n; ++i) {
      const RChainLayer& L = c.l[i];
      const int ldc = L.ldc ? L.ldc : 256, ldh = L.ldh ? L.ldh : 256;
      const bool use_h = c.bwd && L.act != T_ACT_NONE;
      f32x4_t accm[MI][NI], accx[MI][NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int j = 0; j < NI; ++j) { accm[mi][j] = f32x4_t{0.f, 0.f, 0.f, 0.f}; accx[mi][j] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
      auto step = [&](const WFrag& w, int s) {
        // the activation fragments of 16-row tile mi + 1 are read while the MFMAs of tile mi run (two register pairs, not MI)
        auto rd = [&](int mi, f16x8_t& h, f16x8_t& l) {
          const int off = (16 * mi + m16) * RC_LDX + 32 * s + 8 * g;
          h = *(const f16x8_t*)(sX + off); l = *(const f16x8_t*)(sX + XPLANE + off);
        };
        f16x8_t ah[2], al[2];
        rd(0, ah[0], al[0]);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          if (mi + 1 < MI) rd(mi + 1, ah[(mi + 1) & 1], al[(mi + 1) & 1]);
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            accx[mi][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w.h[j], al[mi & 1], accx[mi][j], 0, 0, 0);
            accm[mi][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w.h[j], ah[mi & 1], accm[mi][j], 0, 0, 0);
          }
#pragma unroll
          for (int j = 0; j < NI; ++j) accx[mi][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w.l[j], ah[mi & 1], accx[mi][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      // (same accumulation order per output element as hgemm_body / hgemm_rchain_kernel: steps 0 .. 7, cross before main inside a step)
      load_w(w1); step(w0, 0); load_w(w0); step(w1, 1); load_w(w1); step(w0, 2); load_w(w0); step(w1, 3);
      load_w(w1); step(w0, 4); load_w(w0); step(w1, 5); load_w(w1); step(w0, 6); load_w(w0); step(w1, 7);
      // now w0 holds step 0 of the next layer (of the next tile's first layer behind the last one)
      // ---- epilogue from the accumulators
      const float inv = 1.f / in_scale;
      f32x4_t v[MI][NI];
      float amax = 0.f;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int col = wave * 64 + 16 * j + 4 * g;
        const f32x4_t b4 = (!c.bwd && L.bias) ? *(const f32x4_t*)(L.bias + col) : f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          int64_t r = row0 + 16 * mi + m16;
          const bool on = r < c.M;
          r = on ? r : c.M - 1;
          f32x4_t x = (accm[mi][j] + accx[mi][j] * HG_LO_INV) * inv + b4;
          if (!c.bwd) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (L.act == T_ACT_ELU) x[e] = x[e] > 0.f ? x[e] : expm1f(x[e]);
              else if (L.act == T_ACT_RELU) x[e] = fmaxf(x[e], 0.f);
            }
          } else if (use_h) {
            const f32x4_t hv = *(const f32x4_t*)(L.H + r * ldh + col);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (L.act == T_ACT_ELU) x[e] = hv[e] > 0.f ? x[e] : x[e] * (hv[e] + 1.f);
              else x[e] = hv[e] > 0.f ? x[e] : 0.f;
            }
          }
          if (on) {
            amax = fmaxf(fmaxf(amax, fmaxf(fabsf(x[0]), fabsf(x[1]))), fmaxf(fabsf(x[2]), fabsf(x[3])));
#if !(defined(PNRF_WCHAIN_PROBE) && (PNRF_WCHAIN_PROBE & 1))
            *(f32x4_t*)(L.C + r * ldc + col) = x;
#endif
          }
          v[mi][j] = x;
        }
      }
      amax_l[i] = fmaxf(amax_l[i], amax);
      in_scale = 1.f;
      if (i + 1 < c.n) {
        if (c.bwd) {                                           // the next product's scale from the workgroup's own rows (undone in its epilogue)
          float m = amax;
#pragma unroll
          for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
          if (lane == 0) s_red[wave] = m;
        }
        __syncthreads();                                       // every wave is past its MFMAs: the planes may be rewritten
        if (c.bwd) in_scale = hg_scale_for(fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3])));
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            f16x4_t h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float x = v[mi][j][e] * in_scale; const _Float16 xh = (_Float16)x; h[e] = xh; l[e] = (_Float16)((x - (float)xh) * HG_LO_SCALE); }
            const int off = (16 * mi + m16) * RC_LDX + wave * 64 + 16 * j + 4 * g;
            *(f16x4_t*)(sX + off) = h;
            *(f16x4_t*)(sX + XPLANE + off) = l;
          }
        __syncthreads();
      }
    }
  }
  // max |C| of every layer into its slot (backward): one atomic per workgroup and layer
  if (c.bwd) {
    for (int i = 0; i < c.n; ++i)
      if (c.l[i].c_amax) (void)hg_slot_write(c.l[i].c_amax, amax_l[i], s_red, blockIdx.x);
  }
}
