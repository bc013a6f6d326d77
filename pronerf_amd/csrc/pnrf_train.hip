// pnrf_train.hip — stage-2 training step: forward with saved activations, backward, Adam.
//
// Reference: run_S_eS_eN_alter_base_refine2.py — render_rays :525-680 (forward), loss / backward / optimizer.step :858-869,
// create_nerf :337-395 (one Adam over fine net + sampler + refine net, betas (0.9, 0.999), eps 1e-8, L2 weight decay).
//
// Storage and accumulation are fp32, like the reference's training.  No library GEMM: the layer products (Y = X W^T, dX = dY W, dW = dY^T X)
// are kernels of this library — split-fp16 MFMA (pnrf_hgemm.h; the fine net from 8192 rows on as two launches on the inference path's fused-MLP
// engine, pnrf_tchain.h) or exact-fp32 MFMA (tgemm_kernel below) — and so is every other stage: bias + activation and its backward, the
// sampler head (sigmoid, depth affine, stable 8-sort, gathers) and its scatter backward, the refine head (interval refinement, depth jitter,
// query points) and its backward, positional-encoding forward / backward, alpha-compositing backward (suffix products, no division by the
// transmittance factors), the MSE losses and Adam.  The projection into the training views carries no gradient in the reference
// (`torch.no_grad`, :576) and the Pluecker moment does not depend on the depth along the ray, so no gradient reaches refine_in or mm_input and
// the first layers of the sampler / refine nets need no dX.  A trainer owns parameters, gradients, Adam moments and workspaces; nothing is
// allocated per step and every launch goes to the caller's stream.
#include <string.h>

#include <type_traits>

#include "pnrf_common.h"
#include "pnrf_ieee.h"

using namespace pnrf;

namespace {

constexpr int TPB = 256;
inline int grid_for(int64_t work, int per_block = TPB) {
  int64_t g = (work + per_block - 1) / per_block;
  const int64_t cap = 256 * 16;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

enum { T_ACT_NONE = 0, T_ACT_RELU = 1, T_ACT_ELU = 2 };
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
}  // namespace
namespace {
#include "pnrf_hgemm.h"
#include "pnrf_tchain.h"

// ------------------------------------------------------------------------------------------ layer products with fused epilogues
// The forward product Y = act(X W^T + b) and the input gradient dX = (dZ W [+ dX]) * act'(H_prev) of a Linear layer as ONE kernel each, on
// v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulation: the reference trains in fp32).  They replace a rocBLAS sgemm + a
// bias/activation pass (forward) and a sgemm + the activation-backward pass of the layer below (backward): the activation derivative of
// the PREVIOUS layer is applied in the epilogue of the product that creates its output gradient, so every gradient buffer is written once,
// already as dL/dZ, and no separate elementwise pass runs over [rows, 256] activations.
//
// Shape of the problem: M = rows (rays or ray samples, 4 096 .. 1 M), N, K <= 319.  A [M, K] streams from HBM once, C [M, N] goes back once,
// B (the weights, <= 320 KB) stays in L2: 64 FLOP per HBM byte, next to the crossover of the fp32 MFMA roofline — HBM efficiency decides.
//   * Workgroup = 4 waves on a tile of 16 MI rows x 256 columns (wave w: columns 64 w .. 64 w + 63), so a row of A is fetched by exactly one
//     workgroup.  A goes through LDS in chunks of 64 k: 256 threads fetch 16 MI rows x 256 contiguous bytes (a first, register-direct version
//     read 64-byte pieces of 16 rows per instruction and spent as long in HBM as in the MFMA pipe: 45 % pipe occupancy with 6 % wait
//     cycles), double buffered, one barrier per chunk (256 MFMAs per wave).  Row stride 68 floats: the ds_read_b128 of lane (m, g) — row m, k
//     = 16 s + 4 g — starts at bank 4 (m + g) mod 32, evenly spread.
//   * MFMA operands: lane l = (m = l & 15, g = l >> 4) holds 16 bytes along k — [row m][16 s + 4 g .. + 3] — of both operands, and the four
//     MFMAs of a 16-deep step use element e of every lane's vector, i.e. contract over k = {16 s + 4 g + e : g}: the same permutation of k
//     on both sides, the same sum.  The weight fragment is the MFMA's A operand (C^T = W X^T): D register e of lane (m, g) = C[row m][col 4 g + e].
//   * B fragments come straight from global memory (L2), one 16-deep step ahead.  MODE_NT: B = W [N, K], k contiguous (forward: N = out, K =
//     in); MODE_NN: B = W [K, N], n contiguous (backward: K = out, N = in).
//   * Epilogue through LDS (the A buffers, two halves of the tile): the accumulators are written [row][column], then every thread handles 16
//     contiguous bytes of a row — bias, activation, the saved activation's derivative, the running sum (beta) and the store are all full
//     256..1024-byte row segments.
enum { MODE_NT = 0, MODE_NN = 1 };
struct GemmArgs {
  const float* A; int lda;              // [M, K]
  const float* B; int ldb;              // MODE_NT: [N, K]; MODE_NN: [K, N]
  float* C; int ldc;                    // [M, N]
  int64_t M; int N, K;
  const float* bias;                    // forward: [N] or NULL
  int act;                              // forward: activation of this layer; backward: activation of the layer whose output gradient C is
  const float* H; int ldh;              // backward: saved output of that layer (act'(H)), applied to columns >= act_col0
  int act_col0;
  float beta;                           // backward: C = beta C + A B   (0 or 1)
  float* c_amax;                        // slot that receives max |C| (hg_slot_write; for a split-fp16 product that reads C next), or NULL
};
constexpr int TG_KC = 64;               // k per LDS chunk
constexpr int TG_LDA = TG_KC + 4;       // LDS row stride of the A chunk (floats)
constexpr int TG_LDC = 256 + 4;         // LDS row stride of the C staging tile (floats)
// MI: 16-row MFMA tiles per wave (4: 64-row workgroup tile; 2 / 1: 32 / 16 rows, for the 4 096-row layers of the sampler / refine nets).  VEC: K % 4 == 0 and 16-byte aligned rows
// of A (and of B in MODE_NT): 16-byte loads along k.
template <int MI, int MODE, bool VEC>
__global__ __launch_bounds__(256) void tgemm_kernel(GemmArgs a) {
  constexpr int ROWS = 16 * MI, NI = 4;
  // A chunks: 2 x ROWS x 68 floats; the epilogue reuses the space as [ROWS / 2][260]
  constexpr int HALVES = MI >= 2 ? 2 : 1;                      // the epilogue stages the tile through LDS in this many pieces
  __shared__ __attribute__((aligned(16))) float smem[(2 * ROWS * TG_LDA > (ROWS / HALVES) * TG_LDC) ? 2 * ROWS * TG_LDA : (ROWS / HALVES) * TG_LDC];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m16 = lane & 15, g = lane >> 4;
  const int tiles_n = (a.N + 255) / 256;
  const int64_t tm = blockIdx.x / tiles_n;
  const int tn = (int)(blockIdx.x - tm * tiles_n);
  const int64_t row0 = tm * ROWS;
  const int col0 = tn * 256 + wave * 64;
  const bool wave_on = col0 < a.N;                             // waves past the last column only help with the A chunks
  f32x4_t acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // ---- A chunk loader: thread t -> 16 bytes at k-offset 4 (t & 15) of rows (t >> 4) + 16 pass
  const int lc4 = 4 * (threadIdx.x & 15), lrow = threadIdx.x >> 4;
  constexpr int PASSES = ROWS / 16;
  const float* arow[PASSES];
#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps) {
    const int64_t r = row0 + lrow + 16 * ps;
    arow[ps] = a.A + (r < a.M ? r : a.M - 1) * a.lda;         // rows past the edge: clamped, computed, never stored
  }
  // Loads never feed a select: k past the edge is read from a clamped address and the A tile is zeroed when it is written to LDS (a zero in A
  // makes the matching B value irrelevant), so nothing depends on a fetched register before its planned use and the compiler's vmcnt waits
  // sit where the data is consumed.
  f32x4_t stage[PASSES];
  auto fetch_a = [&](int kc) {
    const int k = kc * TG_KC + lc4;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      if (VEC) {
        stage[ps] = *(const f32x4_t*)(arow[ps] + (k + 3 < a.K ? k : a.K - 4));
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) stage[ps][e] = arow[ps][k + e < a.K ? k + e : a.K - 1];
      }
    }
  };
  auto store_a = [&](int buf, int kc) {
    float* dst = smem + buf * (ROWS * TG_LDA);
    const int k = kc * TG_KC + lc4;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      f32x4_t v = stage[ps];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = k + e < a.K ? v[e] : 0.f;
      *(f32x4_t*)(dst + (lrow + 16 * ps) * TG_LDA + lc4) = v;
    }
  };
  // ---- B fragments of a 16-deep step, straight from global memory (L2)
  const float* bp[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int c = col0 + 16 * j + m16;
    const int cc = c < a.N ? c : a.N - 1;
    bp[j] = MODE == MODE_NT ? a.B + (size_t)cc * a.ldb : a.B + cc;
  }
  auto load_b = [&](int j, int k0) {
    const int k = k0 + 4 * g;
    if (MODE == MODE_NT && VEC) return *(const f32x4_t*)(bp[j] + (k + 3 < a.K ? k : a.K - 4));
    f32x4_t v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int kk = k + e < a.K ? k + e : a.K - 1;
      v[e] = MODE == MODE_NT ? bp[j][kk] : bp[j][(size_t)kk * a.ldb];
    }
    return v;
  };

  // Per chunk (4 steps of 64 MFMAs per wave) two fetch groups, each consumed two steps after it is issued:
  //   G1, before step 0: the next chunk's A rows (-> stage) and this chunk's B fragments of steps 2, 3 (into the registers steps 2, 3 of the
  //       previous chunk have just freed);   G2, before step 2: the next chunk's B fragments of steps 0, 1.
  // vmcnt counts in issue order, so a fetch issued right before a wait would be waited for with it; here every wait only covers fetches that
  // are at least 128 MFMAs old.  One B buffer (64 registers): two waves per SIMD.
  const int chunks = (a.K + TG_KC - 1) / TG_KC;
  f32x4_t fb[TG_KC / 16][NI];
  fetch_a(0);
#pragma unroll
  for (int j = 0; j < NI; ++j) { fb[0][j] = load_b(j, 0); fb[1][j] = load_b(j, 16); }
  store_a(0, 0);
  __syncthreads();
  for (int kc = 0; kc < chunks; ++kc) {
    const float* sA = smem + (kc & 1) * (ROWS * TG_LDA);
    const int kn = kc + 1 < chunks ? kc + 1 : kc;              // past the last chunk: harmless re-fetch of the last one
    fetch_a(kn);
#pragma unroll
    for (int j = 0; j < NI; ++j) { fb[2][j] = load_b(j, kc * TG_KC + 32); fb[3][j] = load_b(j, kc * TG_KC + 48); }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int st = 0; st < TG_KC / 16; ++st) {
      if (st == 2) {
#pragma unroll
        for (int j = 0; j < NI; ++j) { fb[0][j] = load_b(j, kn * TG_KC); fb[1][j] = load_b(j, kn * TG_KC + 16); }
        __builtin_amdgcn_sched_barrier(0);
      }
      f32x4_t fa[MI];
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[i] = *(const f32x4_t*)(sA + (16 * i + m16) * TG_LDA + 16 * st + 4 * g);
      if (wave_on) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[st][j][e], fa[i][e], acc[i][j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (kc + 1 < chunks) store_a((kc + 1) & 1, kc + 1);        // that buffer was last read in chunk kc - 1: every wave has passed its barrier
    __syncthreads();
  }

  // ---- epilogue, two halves of the tile through LDS
  float* sC = smem;
  const bool cvec = a.ldc % 4 == 0 && (((uintptr_t)a.C) & 15) == 0 && a.N % 4 == 0 && (((uintptr_t)a.bias) & 15) == 0 &&
                    (MODE == MODE_NT || a.act == T_ACT_NONE || (a.ldh % 4 == 0 && (((uintptr_t)a.H) & 15) == 0 && a.act_col0 % 4 == 0));
  const bool use_h = MODE == MODE_NN && a.act != T_ACT_NONE, use_c = MODE == MODE_NN && a.beta != 0.f;
  constexpr int HROWS = ROWS / HALVES, HI = MI / HALVES;
  float amax = 0.f;
#pragma unroll
  for (int half = 0; half < HALVES; ++half) {
    if (wave_on) {
#pragma unroll
      for (int ii = 0; ii < HI; ++ii)
#pragma unroll
        for (int j = 0; j < NI; ++j) *(f32x4_t*)(sC + (16 * ii + m16) * TG_LDC + wave * 64 + 16 * j + 4 * g) = acc[half * HI + ii][j];
    }
    __syncthreads();
    // thread t: 16 bytes at column 4 (t & 63) of rows (t >> 6) + 4 q
    const int cl = 4 * (threadIdx.x & 63), c = tn * 256 + cl;
#pragma unroll
    for (int q = 0; q < HROWS / 4; ++q) {
      const int rl = (threadIdx.x >> 6) + 4 * q;
      const int64_t r = row0 + half * HROWS + rl;
      if (r >= a.M || c >= a.N) continue;
      f32x4_t v = *(const f32x4_t*)(sC + rl * TG_LDC + cl);
      float* dst = a.C + r * a.ldc + c;
      if (cvec) {                                               // N % 4 == 0: the four columns are inside together
        if (MODE == MODE_NT) {
          if (a.bias) v += *(const f32x4_t*)(a.bias + c);
        } else {
          if (use_c) v += *(const f32x4_t*)dst;
        }
        f32x4_t h = {1.f, 1.f, 1.f, 1.f};
        if (use_h && c >= a.act_col0) h = *(const f32x4_t*)(a.H + r * a.ldh + (c - a.act_col0));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (MODE == MODE_NT) {
            if (a.act == T_ACT_RELU) v[e] = fmaxf(v[e], 0.f);
            else if (a.act == T_ACT_ELU) v[e] = v[e] > 0.f ? v[e] : expm1f(v[e]);          // F.elu, alpha = 1
          } else if (use_h && c >= a.act_col0) {
            if (a.act == T_ACT_RELU) v[e] = h[e] > 0.f ? v[e] : 0.f;
            else v[e] = h[e] > 0.f ? v[e] : v[e] * (h[e] + 1.f);                           // elu'(z) = exp(z) = elu(z) + 1 for z <= 0
          }
          amax = fmaxf(amax, fabsf(v[e]));
        }
        *(f32x4_t*)dst = v;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (c + e >= a.N) break;
          float x = v[e];
          if (MODE == MODE_NT) {
            if (a.bias) x += a.bias[c + e];
            if (a.act == T_ACT_RELU) x = fmaxf(x, 0.f);
            else if (a.act == T_ACT_ELU) x = x > 0.f ? x : expm1f(x);
          } else {
            if (use_c) x += dst[e];
            if (use_h && c + e >= a.act_col0) {
              const float h = a.H[r * a.ldh + (c + e - a.act_col0)];
              if (a.act == T_ACT_RELU) x = h > 0.f ? x : 0.f;
              else x = h > 0.f ? x : x * (h + 1.f);
            }
          }
          amax = fmaxf(amax, fabsf(x));
          dst[e] = x;
        }
      }
    }
    __syncthreads();
  }
  if (a.c_amax) hg_slot_write(a.c_amax, amax, smem, blockIdx.x);
}

// ------------------------------------------------------------------------------------------ the 1 .. 4-wide heads
// The rgb (128 -> 3) and alpha (256 -> 1) heads of the NeRF are 32 768 .. 1 M rows of a few dot products: streaming kernels, fp32 FMA.  On the
// MFMA tile kernels they cost as much as a 256-wide layer (a 16-column tile for 1 .. 3 columns, thousands of workgroups for the atomics).
constexpr int HEAD_MAX = 4, HEAD_MAX_SPLITS = 256;
struct HeadArgs {
  const float* X; int ldx;              // [M, K]   K = 4 x (32 | 64)
  const float* W; const float* bias;    // [n, K], [n]
  float* Y; int ldy;                    // forward: [M, n]
  const float* dZ; int ldz;             // backward: [M, n]
  float* dX; int lddx; const float* H; int ldh; int act; float beta; float* dx_amax;   // backward: dX = (beta dX + dZ W) * act'(H)
  float* part; float* db_part; int64_t rows_per;                                       // weight gradient: partials [split][n x K], [split][n]
  int64_t M; int n, K;
};
// Y[r, 0..n) = X[r] . W^T + b: K / 4 lanes per row, 16 bytes each; butterfly sums
__global__ __launch_bounds__(256) void head_fwd_kernel(HeadArgs a) {
  const int L = a.K >> 2, rows_w = 64 / L;                     // lanes per row, rows per wave and step
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane % L, rw = lane / L;
  f32x4_t w[HEAD_MAX];
#pragma unroll
  for (int j = 0; j < HEAD_MAX; ++j) w[j] = j < a.n ? *(const f32x4_t*)(a.W + (size_t)j * a.K + 4 * lr) : f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int64_t step = (int64_t)gridDim.x * 4 * rows_w;
  for (int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * rows_w; r0 < a.M; r0 += step) {
    const int64_t r = r0 + rw;
    const f32x4_t x = r < a.M ? *(const f32x4_t*)(a.X + r * a.ldx + 4 * lr) : f32x4_t{0.f, 0.f, 0.f, 0.f};
    float acc[HEAD_MAX];
#pragma unroll
    for (int j = 0; j < HEAD_MAX; ++j) acc[j] = (x[0] * w[j][0] + x[1] * w[j][1]) + (x[2] * w[j][2] + x[3] * w[j][3]);
    for (int o = L >> 1; o >= 1; o >>= 1) {
#pragma unroll
      for (int j = 0; j < HEAD_MAX; ++j) acc[j] += __shfl_xor(acc[j], o);
    }
    if (lr == 0 && r < a.M) {
#pragma unroll
      for (int j = 0; j < HEAD_MAX; ++j)
        if (j < a.n) a.Y[r * a.ldy + j] = acc[j] + a.bias[j];
    }
  }
}
// dX[r, c..c+3] = (beta dX + sum_j dZ[r, j] W[j, c..c+3]) * act'(H): a thread keeps its four columns of W in registers and walks down the rows
__global__ __launch_bounds__(256) void head_dx_kernel(HeadArgs a) {
  __shared__ float red[16];
  const int Q = a.K >> 2;                                      // column quads per row (32 | 64)
  const int cq = threadIdx.x % Q, rl = threadIdx.x / Q, rows_b = 256 / Q;
  f32x4_t w[HEAD_MAX];
#pragma unroll
  for (int j = 0; j < HEAD_MAX; ++j) w[j] = j < a.n ? *(const f32x4_t*)(a.W + (size_t)j * a.K + 4 * cq) : f32x4_t{0.f, 0.f, 0.f, 0.f};
  float amax = 0.f;
  auto finish = [&](int64_t r, f32x4_t v, const f32x4_t& h) {
    if (a.act != T_ACT_NONE) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (a.act == T_ACT_RELU) v[e] = h[e] > 0.f ? v[e] : 0.f;
        else v[e] = h[e] > 0.f ? v[e] : v[e] * (h[e] + 1.f);
      }
    }
    amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    *(f32x4_t*)(a.dX + r * a.lddx + 4 * cq) = v;
  };
  const int64_t stride = (int64_t)gridDim.x * rows_b;
  int64_t r = (int64_t)blockIdx.x * rows_b + rl;
  for (; r + 3 * stride < a.M; r += 4 * stride) {            // four rows' loads in flight per thread (as head_dw_kernel)
    f32x4_t v[4], h[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t ru = r + u * stride;
      v[u] = a.beta != 0.f ? *(const f32x4_t*)(a.dX + ru * a.lddx + 4 * cq) : f32x4_t{0.f, 0.f, 0.f, 0.f};
      h[u] = a.act != T_ACT_NONE ? *(const f32x4_t*)(a.H + ru * a.ldh + 4 * cq) : f32x4_t{0.f, 0.f, 0.f, 0.f};
      f32x4_t s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < HEAD_MAX; ++j)
        if (j < a.n) s += a.dZ[ru * a.ldz + j] * w[j];
      v[u] = s + v[u];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) finish(r + u * stride, v[u], h[u]);
  }
  for (; r < a.M; r += stride) {
    f32x4_t v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < HEAD_MAX; ++j)
      if (j < a.n) v += a.dZ[r * a.ldz + j] * w[j];
    if (a.beta != 0.f) v += *(const f32x4_t*)(a.dX + r * a.lddx + 4 * cq);
    const f32x4_t h = a.act != T_ACT_NONE ? *(const f32x4_t*)(a.H + r * a.ldh + 4 * cq) : f32x4_t{0.f, 0.f, 0.f, 0.f};
    finish(r, v, h);
  }
  if (a.dx_amax) hg_slot_write(a.dx_amax, amax, red, blockIdx.x);
}
// partial dW[j, c..c+3] and db[j] over the rows of split blockIdx.x: a thread accumulates its four columns for the n outputs over every
// (256 / quads)-th row, the row lanes are summed through LDS
__global__ __launch_bounds__(256) void head_dw_kernel(HeadArgs a) {
  __shared__ float red[8 * HEAD_MAX * 256];                   // [row lane][j][column]
  const int Q = a.K >> 2, rows_b = 256 / Q;
  const int cq = threadIdx.x % Q, rl = threadIdx.x / Q;
  const int64_t r_begin = blockIdx.x * a.rows_per, r_end = r_begin + a.rows_per < a.M ? r_begin + a.rows_per : a.M;
  f32x4_t acc[HEAD_MAX];
  float sz[HEAD_MAX];
#pragma unroll
  for (int j = 0; j < HEAD_MAX; ++j) { acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f}; sz[j] = 0.f; }
  // four rows' loads in flight per thread (one row at a time the loop waited out a memory latency per 16 bytes: 132 us for 268 MB at 262 144
  // rows); the sums keep their row order
  int64_t r = r_begin + rl;
  for (; r + 3 * rows_b < r_end; r += 4 * rows_b) {
    f32x4_t x[4];
    float z[4][HEAD_MAX];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      x[u] = *(const f32x4_t*)(a.X + (r + u * rows_b) * a.ldx + 4 * cq);
#pragma unroll
      for (int j = 0; j < HEAD_MAX; ++j) z[u][j] = j < a.n ? a.dZ[(r + u * rows_b) * a.ldz + j] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < HEAD_MAX; ++j)
        if (j < a.n) { acc[j] += z[u][j] * x[u]; sz[j] += z[u][j]; }
  }
  for (; r < r_end; r += rows_b) {
    const f32x4_t x = *(const f32x4_t*)(a.X + r * a.ldx + 4 * cq);
#pragma unroll
    for (int j = 0; j < HEAD_MAX; ++j)
      if (j < a.n) { const float z = a.dZ[r * a.ldz + j]; acc[j] += z * x; sz[j] += z; }
  }
#pragma unroll
  for (int j = 0; j < HEAD_MAX; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) red[(rl * HEAD_MAX + j) * 256 + 4 * cq + e] = acc[j][e];
  __syncthreads();
  float* p = a.part + (size_t)blockIdx.x * a.n * a.K;
  for (int i = threadIdx.x; i < a.n * a.K; i += 256) {
    const int j = i / a.K, c = i - j * a.K;
    float s = 0.f;
    for (int q = 0; q < rows_b; ++q) s += red[(q * HEAD_MAX + j) * 256 + c];
    p[i] = s;
  }
  __syncthreads();
  // bias partials: the threads of column quad 0 hold the row sums of dZ of their row lane
  if (cq == 0) {
#pragma unroll
    for (int j = 0; j < HEAD_MAX; ++j) red[rl * HEAD_MAX + j] = sz[j];
  }
  __syncthreads();
  if ((int)threadIdx.x < a.n) {
    float s = 0.f;
    for (int q = 0; q < rows_b; ++q) s += red[q * HEAD_MAX + threadIdx.x];
    a.db_part[(size_t)blockIdx.x * a.n + threadIdx.x] = s;
  }
}

// ------------------------------------------------------------------------------------------ weight gradient: split-K MFMA GEMM
// dW[out, in] = dZ^T[out, R] X[R, in] with R = rays or ray-samples (4096 .. 32768) and out, in <= 319: a tiny output with a
// very long contraction.  rocBLAS runs the 256x256 cases as four 128x128 macro-tiles without splitting K (4 workgroups on
// 256 CUs, 0.45 ms each, 4.0 of 6.8 ms per training iteration), hence this kernel: grid = (64x64 output tiles) x (K
// splits), each workgroup reduces its slice of rows with v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulate) and
// writes a partial tile; dw_reduce_kernel adds the partials in a fixed order (deterministic, no atomics).
//   workgroup = 4 waves, wave w owns the 32x32 quadrant (w>>1, w&1) of the tile = 2x2 MFMA tiles;
//   MFMA operands: A[m][k] = dZ[row k][out m], B[k][n] = X[row k][in n]; lane l supplies (m or n) = l&15, k = l>>4;
//   rows are staged 32 at a time through LDS with a row stride of 80 floats: the 4 k-rows a wave reads per MFMA then fall
//   into disjoint bank groups (ds_read_b32: 32 banks, conflicts within each 32-lane half).
constexpr int DW_TILE = 64, DW_ROWS = 32, DW_LDS_STRIDE = 80, DW_MAX_SPLITS = 64;
constexpr int DB_MAX_OUT = 512;          // widest layer output the bias-partial buffer is sized for
__global__ __launch_bounds__(256) void dw_splitk_kernel(const float* __restrict__ dZ, int ldz, const float* __restrict__ X, int ldx,
                                                        float* __restrict__ part, int out, int in, int64_t R, int64_t rows_per_split,
                                                        float* __restrict__ db_part) {
  __shared__ float sA[DW_ROWS * DW_LDS_STRIDE];
  __shared__ float sB[DW_ROWS * DW_LDS_STRIDE];
  const int tiles_n = (in + DW_TILE - 1) / DW_TILE;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int c16 = lane & 15, q = lane >> 4;
  const int64_t r_begin = blockIdx.y * rows_per_split;
  const int64_t r_end = (r_begin + rows_per_split < R) ? r_begin + rows_per_split : R;
  f32x4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int lc = threadIdx.x & 63, lr = threadIdx.x >> 6;          // loader: column 0..63, rows lr, lr+4, ...
  const int gm = tm * DW_TILE + lc, gn = tn * DW_TILE + lc;
  // the next 32-row block is fetched into registers while the MFMAs of the current one run (its global-load latency would
  // otherwise be exposed once per block: 32 times per workgroup)
  float pa[DW_ROWS / 4], pb[DW_ROWS / 4];
  auto fetch = [&](int64_t r0) {
#pragma unroll
    for (int k = 0; k < DW_ROWS / 4; ++k) {
      const int64_t r = r0 + lr + 4 * k;
      const bool ok = r < r_end;
      pa[k] = (ok && gm < out) ? dZ[r * ldz + gm] : 0.f;
      pb[k] = (ok && gn < in) ? X[r * ldx + gn] : 0.f;
    }
  };
  fetch(r_begin);
  float colsum = 0.f;            // bias gradient: column sums of dZ over this split's rows (the loader touches every element once)
  for (int64_t r0 = r_begin; r0 < r_end; r0 += DW_ROWS) {
#pragma unroll
    for (int k = 0; k < DW_ROWS / 4; ++k) {
      sA[(lr + 4 * k) * DW_LDS_STRIDE + lc] = pa[k];
      sB[(lr + 4 * k) * DW_LDS_STRIDE + lc] = pb[k];
      colsum += pa[k];
    }
    __syncthreads();
    if (r0 + DW_ROWS < r_end) fetch(r0 + DW_ROWS);
#pragma unroll
    for (int kk = 0; kk < DW_ROWS / 4; ++kk) {
      const int row = (4 * kk + q) * DW_LDS_STRIDE;
      const float a0 = sA[row + wm * 32 + c16], a1 = sA[row + wm * 32 + 16 + c16];
      const float b0 = sB[row + wn * 32 + c16], b1 = sB[row + wn * 32 + 16 + c16];
      acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();
  }
  float* p = part + (size_t)blockIdx.y * out * in;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = tm * DW_TILE + wm * 32 + 16 * i + 4 * q + e, n = tn * DW_TILE + wn * 32 + 16 * j + c16;      // D reg e = row 4q+e, col l&15
        if (m < out && n < in) p[(size_t)m * in + n] = acc[i][j][e];
      }
  if (db_part && tn == 0) {      // the tiles of the first column block carry the bias partials of their 64 output rows
    sA[lr * 64 + lc] = colsum;   // (the last row block's barrier has passed: sA is free)
    __syncthreads();
    if (lr == 0 && gm < out) db_part[(size_t)blockIdx.y * out + gm] = (sA[lc] + sA[64 + lc]) + (sA[128 + lc] + sA[192 + lc]);
  }
}
// 128 x 128 tiles for the square hidden layers: each workgroup reads its row range of dZ and X half as often as with 64 x 64 tiles
// (the kernel is bound by those reads: 16 FLOP per byte at 64 x 64), and an operand fetch is one ds_read_b128 per four MFMAs.
// Wave (wm, wn) owns 64 x 64 as 4 x 4 MFMA tiles; a lane's four columns 16 i + c16 of its 64-column span sit side by side in LDS
// (position 4 c16 + i), rows are 128 floats apart: the b128 reads of the four row groups of a wave hit disjoint banks.
// Requires out, in multiples of 128 and 16-byte aligned rows (ldz, ldx multiples of 4).
constexpr int DW128_ROWS = 32, DW128_MAX_SPLITS = 128;
__global__ __launch_bounds__(256) void dw_splitk128_kernel(const float* __restrict__ dZ, int ldz, const float* __restrict__ X, int ldx,
                                                           float* __restrict__ part, int out, int in, int64_t R, int64_t rows_per_split,
                                                           float* __restrict__ db_part) {
  __shared__ __attribute__((aligned(16))) float sAB[2][2][DW128_ROWS * 128];     // [buffer][A | B]: double buffered, one barrier per row block
  const int tiles_n = in / 128;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int c16 = lane & 15, q = lane >> 4;
  const int64_t r_begin = blockIdx.y * rows_per_split;
  const int64_t r_end = (r_begin + rows_per_split < R) ? r_begin + rows_per_split : R;
  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // loader: thread = 4 consecutive columns (float4) of rows lr, lr + 8, lr + 16, lr + 24
  const int lcol = (threadIdx.x & 31) * 4, lr = threadIdx.x >> 5;
  const int lpos = (lcol & 64) + 4 * (lcol & 15) + ((lcol & 63) >> 4);        // LDS position of column lcol; columns lcol + e sit 4 e further
  const float* gA = dZ + (size_t)tm * 128 + lcol;
  const float* gB = X + (size_t)tn * 128 + lcol;
  f32x4_t pa[DW128_ROWS / 8], pb[DW128_ROWS / 8];
  auto fetch = [&](int64_t r0) {
#pragma unroll
    for (int k = 0; k < DW128_ROWS / 8; ++k) {
      const int64_t r = r0 + lr + 8 * k;
      if (r < r_end) {
        pa[k] = *(const f32x4_t*)(gA + r * ldz);
        pb[k] = *(const f32x4_t*)(gB + r * ldx);
      } else {
        pa[k] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        pb[k] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  fetch(r_begin);
  f32x4_t colsum = {0.f, 0.f, 0.f, 0.f};      // bias gradient partials of this thread's four dZ columns
  int buf = 0;
  for (int64_t r0 = r_begin; r0 < r_end; r0 += DW128_ROWS, buf ^= 1) {
    // the block before last was read from this buffer; every wave has passed the barrier of the last block since, so it is free
    float* sA = sAB[buf][0];
    float* sB = sAB[buf][1];
#pragma unroll
    for (int k = 0; k < DW128_ROWS / 8; ++k) {
      colsum += pa[k];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sA[(lr + 8 * k) * 128 + lpos + 4 * e] = pa[k][e];
        sB[(lr + 8 * k) * 128 + lpos + 4 * e] = pb[k][e];
      }
    }
    __syncthreads();
    if (r0 + DW128_ROWS < r_end) fetch(r0 + DW128_ROWS);
#pragma unroll
    for (int kk = 0; kk < DW128_ROWS / 4; ++kk) {
      const f32x4_t a = *(const f32x4_t*)(sA + (4 * kk + q) * 128 + wm * 64 + 4 * c16);
      const f32x4_t b = *(const f32x4_t*)(sB + (4 * kk + q) * 128 + wn * 64 + 4 * c16);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  float* p = part + (size_t)blockIdx.y * out * in;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = tm * 128 + wm * 64 + 16 * i + 4 * q + e, n = tn * 128 + wn * 64 + 16 * j + c16;
        p[(size_t)m * in + n] = acc[i][j][e];
      }
  if (db_part && tn == 0) {
    __syncthreads();                                        // every wave is done reading the staging buffers
    float* red = sAB[0][0];                                 // [8 row lanes][128 columns]
#pragma unroll
    for (int e = 0; e < 4; ++e) red[lr * 128 + lcol + e] = colsum[e];
    __syncthreads();
    if (threadIdx.x < 128) {
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) sum += red[r * 128 + threadIdx.x];
      db_part[(size_t)blockIdx.y * out + tm * 128 + threadIdx.x] = sum;
    }
  }
}
// Adds the split-K partials of every weight gradient of the iteration in a fixed order, in ONE launch at the end of the backward pass (26
// layers: 26 launches of ~16 us each were 0.4 ms of a 3 ms iteration).  Each layer's weight-gradient kernel wrote its partials into its own
// slice of the trainer's pool; a job names that slice, and the workgroups [block0, block0 + blocks) belong to it: the first `blocks - db_blocks`
// of them sum dW, the rest the bias gradient's column-sum partials that the weight-gradient kernels' loaders produce on the way.
struct DwJob {
  const float* part; const float* db_part; float* dW; float* db;
  int numel, splits, out, block0, blocks, db_blocks;   // numel = out x in_src: elements of one partial
  int in_src, in_dst, gap;                             // partial rows have in_src columns; column `gap` of them (if >= 0) is not part of dW
};
constexpr int DW_MAX_JOBS = 32;
struct DwJobs { DwJob j[DW_MAX_JOBS]; int n; };
__global__ void dw_reduce_kernel(const DwJobs jobs) {
  int k = 0;
  while (k + 1 < jobs.n && (int)blockIdx.x >= jobs.j[k + 1].block0) ++k;
  const DwJob& J = jobs.j[k];
  const int b = (int)blockIdx.x - J.block0, dw_blocks = J.blocks - J.db_blocks;
  // Many splits (the 1 .. 4-wide heads write 256 partials of a few hundred values): eight lanes share an element — lane `sub` sums its eighth of
  // the splits in order, the eight sums are added as a fixed tree — instead of one thread walking 256 partials (that walk was 15 of this
  // launch's 38 us at 32 768 rows).  Fewer splits: one thread per element, in split order.  Either way the order is fixed: run-to-run identical.
  constexpr int SUB = 8, MANY = 64;
  if (b >= dw_blocks) {
    if (J.splits >= MANY) {
      const int t = (b - dw_blocks) * (int)blockDim.x + (int)threadIdx.x, j = t / SUB, sub = t % SUB, per = (J.splits + SUB - 1) / SUB;
      float s = 0.f;
      if (j < J.out)
        for (int q = sub * per; q < (sub + 1) * per && q < J.splits; ++q) s += J.db_part[(size_t)q * J.out + j];
#pragma unroll
      for (int o = 1; o < SUB; o <<= 1) s += __shfl_xor(s, o);
      if (j < J.out && sub == 0) J.db[j] = s;
      return;
    }
    const int j = (b - dw_blocks) * (int)blockDim.x + (int)threadIdx.x;
    if (j < J.out) {
      float s = 0.f;
      for (int q = 0; q < J.splits; ++q) s += J.db_part[(size_t)q * J.out + j];
      J.db[j] = s;
    }
    return;
  }
  if (J.gap < 0 && J.numel % 4 == 0 && J.splits >= MANY) {
    const int n4 = J.numel >> 2, per = (J.splits + SUB - 1) / SUB;
    const f32x4_t* part = (const f32x4_t*)J.part;
    const int total = (n4 * SUB + (int)blockDim.x - 1) / (int)blockDim.x * (int)blockDim.x;      // whole blocks: the lanes of a group stay together
    for (int t = b * (int)blockDim.x + (int)threadIdx.x; t < total; t += dw_blocks * (int)blockDim.x) {
      const int i = t / SUB, sub = t % SUB;
      f32x4_t s = {0.f, 0.f, 0.f, 0.f};
      if (i < n4) {
        const int q1 = (sub + 1) * per < J.splits ? (sub + 1) * per : J.splits;
        int q = sub * per;
        for (; q + 4 <= q1; q += 4) {
          const f32x4_t p0 = part[(size_t)q * n4 + i], p1 = part[(size_t)(q + 1) * n4 + i], p2 = part[(size_t)(q + 2) * n4 + i],
                        p3 = part[(size_t)(q + 3) * n4 + i];
          s = (((s + p0) + p1) + p2) + p3;
        }
        for (; q < q1; ++q) s += part[(size_t)q * n4 + i];
      }
#pragma unroll
      for (int o = 1; o < SUB; o <<= 1)
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] += __shfl_xor(s[e], o);
      if (i < n4 && sub == 0) ((f32x4_t*)J.dW)[i] = s;
    }
    return;
  }
  if (J.gap < 0 && J.numel % 4 == 0) {                         // 16 bytes per thread, four partials in flight: the sum stays in split order
    const int n4 = J.numel >> 2;
    const f32x4_t* part = (const f32x4_t*)J.part;
    for (int i = b * (int)blockDim.x + (int)threadIdx.x; i < n4; i += dw_blocks * (int)blockDim.x) {
      f32x4_t s = {0.f, 0.f, 0.f, 0.f};
      int q = 0;
      for (; q + 4 <= J.splits; q += 4) {
        const f32x4_t p0 = part[(size_t)q * n4 + i], p1 = part[(size_t)(q + 1) * n4 + i], p2 = part[(size_t)(q + 2) * n4 + i],
                      p3 = part[(size_t)(q + 3) * n4 + i];
        s = (((s + p0) + p1) + p2) + p3;
      }
      for (; q < J.splits; ++q) s += part[(size_t)q * n4 + i];
      ((f32x4_t*)J.dW)[i] = s;
    }
    return;
  }
  for (int i = b * (int)blockDim.x + (int)threadIdx.x; i < J.numel; i += dw_blocks * (int)blockDim.x) {
    float s = 0.f;
    for (int q = 0; q < J.splits; ++q) s += J.part[(size_t)q * J.numel + i];
    if (J.gap < 0) { J.dW[i] = s; continue; }
    const int o = i / J.in_src, c = i - o * J.in_src;
    if (c != J.gap) J.dW[o * J.in_dst + (c < J.gap ? c : c - 1)] = s;
  }
}

// ------------------------------------------------------------------------------------------ positional encoding
// The NeRF's inputs in one launch: gamma(x) (10 octaves, 63 values) into the head of the skip layer's concatenated rows — which are also the
// first layer's input rows (row stride ld_c5) —, gamma(v) (4 octaves, 27 values; the direction of ray row / rep) behind the feature columns of
// the views layer's rows.  One thread per (row, q of 32): q < 30 is the (octave q / 3, coordinate q % 3) pair of gamma(x) — one argument, its
// sine and its cosine —, q < 12 also that pair of gamma(v), q = 30 / 31 copy the raw x / v.  (The first version, one thread per (row,
// coordinate) with 51 strided stores and a second 90-wide copy of the embedding, took 85 us at 262 144 rows.)  Column order of an embedding =
// [x(3), sin 2^0 x, cos 2^0 x, sin 2^1 x, ...], the same expressions as Embedder.embed (run_nerf_helpers.py:666-671).
__global__ void nerf_inputs_kernel(const float* __restrict__ pts, const float* __restrict__ dirs, int dir_stride, int rep, float* __restrict__ c5, int ld_c5,
                                   float* __restrict__ cv, int ld_cv, int cv_col, int64_t rows) {
  const int64_t total = rows * 32;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i >> 5;
    const int q = (int)(i & 31);
    const float* x = pts + row * 3;
    const float* v = dirs + (row / rep) * dir_stride;
    float* o = c5 + row * ld_c5;
    float* ov = cv + row * ld_cv + cv_col;
    if (q < 30) {
      const int k = q / 3, c = q - 3 * k;
      const float arg = x[c] * (float)(1u << k);
      o[3 + 6 * k + c] = sinf(arg);
      o[3 + 6 * k + 3 + c] = cosf(arg);
      if (q < 12) {
        const float argv = v[c] * (float)(1u << k);
        ov[3 + 6 * k + c] = sinf(argv);
        ov[3 + 6 * k + 3 + c] = cosf(argv);
      }
    } else if (q == 30) {
      o[0] = x[0]; o[1] = x[1]; o[2] = x[2];
    } else {
      ov[0] = v[0]; ov[1] = v[1]; ov[2] = v[2];
    }
  }
}
// dx[row, c] = sum over the two gradient sources (either may be NULL) of  dE[c] + sum_k 2^k (cos(2^k x) dE[sin_k] - sin(2^k x) dE[cos_k])
__global__ void posenc_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dA, int lda, const float* __restrict__ dB, int ldb,
                                  float* __restrict__ dx, int64_t rows, int n_freq) {
  const int64_t total = rows * 3;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / 3;
    const int c = (int)(i - row * 3);
    const float v = x[i];
    float g = 0.f;
    for (int src = 0; src < 2; ++src) {
      const float* d = src == 0 ? dA : dB;
      if (!d) continue;
      const float* e = d + row * (src == 0 ? lda : ldb);
      g += e[c];
      for (int k = 0; k < n_freq; ++k) {
        const float f = (float)(1u << k), arg = v * f;
        g += f * (cosf(arg) * e[3 + 6 * k + c] - sinf(arg) * e[3 + 6 * k + 3 + c]);
      }
    }
    dx[i] = g;
  }
}

// ------------------------------------------------------------------------------------------ sampler head
// y[n,27] -> depth = sigmoid(y[0:8]) (far-near)+near, stable ascending sort, add/mul gathered by the sort indices,
// mm_rgb = sigmoid(y[24:27])  (refine2.py:551-568).  One thread per ray.
__global__ void sampler_head_fwd_kernel(const float* __restrict__ y, const float* __restrict__ rays, float* __restrict__ depth_sorted,
                                        int64_t* __restrict__ idx_out, float* __restrict__ add_s, float* __restrict__ mul_s, float* __restrict__ mm_rgb,
                                        int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float* yr = y + i * 27;
    const float near = rays[i * 11 + 6], far = rays[i * 11 + 7];
    const float span = ieee_sub(far, near);
    float dep[8];
    int idx[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) { dep[s] = ieee_add(ieee_mul(sigmoid_f(yr[s]), span), near); idx[s] = s; }
    // insertion sort on (value, index): stable, 8 elements
#pragma unroll
    for (int a = 1; a < 8; ++a) {
#pragma unroll
      for (int b = a; b > 0; --b) {
        const bool sw = dep[b - 1] > dep[b];
        const float tv = sw ? dep[b - 1] : dep[b]; dep[b - 1] = sw ? dep[b] : dep[b - 1]; dep[b] = tv;
        const int ti = sw ? idx[b - 1] : idx[b]; idx[b - 1] = sw ? idx[b] : idx[b - 1]; idx[b] = ti;
      }
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      depth_sorted[i * 8 + s] = dep[s];
      idx_out[i * 8 + s] = idx[s];
      float a = yr[8], m = yr[16];
#pragma unroll
      for (int k = 1; k < 8; ++k) { a = idx[s] == k ? yr[8 + k] : a; m = idx[s] == k ? yr[16 + k] : m; }
      add_s[i * 8 + s] = a; mul_s[i * 8 + s] = m;
    }
    if (mm_rgb) {
#pragma unroll
      for (int c = 0; c < 3; ++c) mm_rgb[i * 3 + c] = sigmoid_f(yr[24 + c]);
    }
  }
}
// dy[n,27] from d depth_sorted, d add_sorted, d mul_sorted (scatter through the sort indices) and d mm_rgb (may be NULL).
__global__ void sampler_head_bwd_kernel(const float* __restrict__ y, const float* __restrict__ rays, const int64_t* __restrict__ idx,
                                        const float* __restrict__ d_depth, const float* __restrict__ d_add, const float* __restrict__ d_mul,
                                        const float* __restrict__ d_rgb, float* __restrict__ dy, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float* yr = y + i * 27;
    const float span = rays[i * 11 + 7] - rays[i * 11 + 6];
    float g[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) g[k] = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int k = (int)idx[i * 8 + s];
      const float gd = d_depth[i * 8 + s], ga = d_add[i * 8 + s], gm = d_mul[i * 8 + s];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (q == k) { g[q] += gd; g[8 + q] += ga; g[16 + q] += gm; }
      }
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const float sg = sigmoid_f(yr[s]);
      g[s] = g[s] * span * sg * (1.f - sg);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float sg = sigmoid_f(yr[24 + c]);
      g[24 + c] = d_rgb ? d_rgb[i * 3 + c] * sg * (1.f - sg) : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 27; ++k) dy[i * 27 + k] = g[k];
  }
}

// ------------------------------------------------------------------------------------------ refine head
// y[n,35] -> refine = sigmoid(y[0:8]), offsets = tanh(y[8:32]), rgb0 = sigmoid(y[32:35]); interval refinement over the sorted
// depths, depth jitter toward the next (dir > 0) / previous (dir < 0) sample, query points o + d z + 0.01 offsets
// (refine2.py:635-668).  z_pre = refined depths before the jitter (saved for the backward).
__global__ void refine_head_fwd_kernel(const float* __restrict__ y, const float* __restrict__ rays, const float* __restrict__ depth_sorted,
                                       const float* __restrict__ jitter, int jitter_dir, float* __restrict__ z_pre, float* __restrict__ z,
                                       float* __restrict__ pts, float* __restrict__ rgb0, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float* yr = y + i * 35;
    const float* r = rays + i * 11;
    const float near = r[6], far = r[7];
    float D[8], zz[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) D[s] = depth_sorted[i * 8 + s];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const float lower = s == 0 ? ieee_mul(0.5f, ieee_add(near, D[0])) : ieee_mul(0.5f, ieee_add(D[s], D[s - 1]));
      const float upper = s == 7 ? ieee_mul(0.5f, ieee_add(far, D[7])) : ieee_mul(0.5f, ieee_add(D[s + 1], D[s]));
      zz[s] = ieee_add(lower, ieee_mul(ieee_sub(upper, lower), sigmoid_f(yr[s])));
      z_pre[i * 8 + s] = zz[s];
    }
    if (jitter) {
      float zn[8];
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const float j = jitter[i * 8 + s];
        if (jitter_dir > 0) zn[s] = ieee_add(zz[s], ieee_mul(j, fabsf(ieee_sub(zz[s], s < 7 ? zz[s < 7 ? s + 1 : 7] : far))));
        else zn[s] = ieee_sub(zz[s], ieee_mul(j, fabsf(ieee_sub(zz[s], s > 0 ? zz[s > 0 ? s - 1 : 0] : near))));
      }
#pragma unroll
      for (int s = 0; s < 8; ++s) zz[s] = zn[s];
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      z[i * 8 + s] = zz[s];
#pragma unroll
      for (int c = 0; c < 3; ++c)
        pts[(i * 8 + s) * 3 + c] = ieee_add(ieee_add(r[c], ieee_mul(r[3 + c], zz[s])), ieee_mul(1e-2f, tanhf(yr[8 + 3 * s + c])));
    }
    if (rgb0) {
#pragma unroll
      for (int c = 0; c < 3; ++c) rgb0[i * 3 + c] = sigmoid_f(yr[32 + c]);
    }
  }
}
// Backward of the refine head.  Inputs: d pts [n,8,3], d z [n,8] (from the compositing), d rgb0 [n,3] or NULL.
// Outputs: dy [n,35], d depth_sorted [n,8].
__global__ void refine_head_bwd_kernel(const float* __restrict__ y, const float* __restrict__ rays, const float* __restrict__ depth_sorted,
                                       const float* __restrict__ z_pre, const float* __restrict__ jitter, int jitter_dir,
                                       const float* __restrict__ d_pts, const float* __restrict__ d_z, const float* __restrict__ d_rgb0,
                                       float* __restrict__ dy, float* __restrict__ d_depth, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float* yr = y + i * 35;
    const float* r = rays + i * 11;
    const float near = r[6], far = r[7];
    float gz[8], zp[8], D[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      zp[s] = z_pre[i * 8 + s]; D[s] = depth_sorted[i * 8 + s];
      float g = d_z ? d_z[i * 8 + s] : 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float gp = d_pts[(i * 8 + s) * 3 + c];
        g += gp * r[3 + c];                                                   // pts = o + d z + 0.01 offs
        const float t = tanhf(yr[8 + 3 * s + c]);
        dy[i * 35 + 8 + 3 * s + c] = 1e-2f * gp * (1.f - t * t);
      }
      gz[s] = g;
    }
    if (jitter) {        // z' = z +- j |z - neighbour|
      float gp[8];
#pragma unroll
      for (int s = 0; s < 8; ++s) gp[s] = 0.f;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const float j = jitter[i * 8 + s];
        if (jitter_dir > 0) {
          const float nb = s < 7 ? zp[s < 7 ? s + 1 : 7] : far;
          const float u = zp[s] - nb;
          const float sg = u > 0.f ? 1.f : (u < 0.f ? -1.f : 0.f);
          gp[s] += gz[s] * (1.f + j * sg);
          if (s < 7) gp[s < 7 ? s + 1 : 7] += gz[s] * (-j * sg);
        } else {
          const float nb = s > 0 ? zp[s > 0 ? s - 1 : 0] : near;
          const float u = zp[s] - nb;
          const float sg = u > 0.f ? 1.f : (u < 0.f ? -1.f : 0.f);
          gp[s] += gz[s] * (1.f - j * sg);
          if (s > 0) gp[s > 0 ? s - 1 : 0] += gz[s] * (j * sg);
        }
      }
#pragma unroll
      for (int s = 0; s < 8; ++s) gz[s] = gp[s];
    }
    // z = lower + (upper - lower) refine
    float gD[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) gD[s] = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const float lower = s == 0 ? 0.5f * (near + D[0]) : 0.5f * (D[s] + D[s - 1]);
      const float upper = s == 7 ? 0.5f * (far + D[7]) : 0.5f * (D[s + 1] + D[s]);
      const float rf = sigmoid_f(yr[s]);
      dy[i * 35 + s] = gz[s] * (upper - lower) * rf * (1.f - rf);
      const float gl = gz[s] * (1.f - rf), gu = gz[s] * rf;
      if (s == 0) gD[0] += 0.5f * gl; else { gD[s] += 0.5f * gl; gD[s - 1] += 0.5f * gl; }
      if (s == 7) gD[7] += 0.5f * gu; else { gD[s + 1] += 0.5f * gu; gD[s] += 0.5f * gu; }
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) d_depth[i * 8 + s] = gD[s];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float sg = sigmoid_f(yr[32 + c]);
      dy[i * 35 + 32 + c] = d_rgb0 ? d_rgb0[i * 3 + c] * sg * (1.f - sg) : 0.f;
    }
  }
}

// ------------------------------------------------------------------------------------------ compositing backward
// raw2outputs (refine2.py:475-522) backward for d rgb_map [n,3]: d raw [n,S,4], d z [n,S], d add / d mul [n,S] (NULL to skip).
// Any S (a runtime value; the reference's exploration goes to 64, BASELINE.json's stress bound is 256).  Pass 1 (ascending) forms alpha_s and the
// exclusive transmittance T_s = prod_{k<s} x_k, x_k = 1 - alpha_k + 1e-10.  Pass 2 (descending) carries Q_s = sum_{j>s} dw_j alpha_j prod_{s<k<j} x_k through
// the recurrence Q_{s-1} = dw_s alpha_s + x_s Q_s, so that  d alpha_s = T_s (dw_s - Q_s)  — the derivative of the transmittance products
// without ever dividing by a factor x_k that can be 1e-10.
// One WAVE per ray, lane = sample (chunks of 64 for S > 64), as composite_kernel (pnrf_ops.hip): per-sample work in parallel, the two
// recurrences over the lanes in sample order through v_readlane.  (A thread per ray walked 2 S memory latencies: 51 us for 4096 rays x 64
// samples.)  d_raw is no scratch any more: alpha and T stay in registers (the chunks are walked from the last to the first for Q; for
// S > 64 the transmittance at a chunk's start is recomputed from the chunks before it).
__global__ __launch_bounds__(256) void composite_bwd_kernel(const float* __restrict__ raw, const float* __restrict__ z, const float* __restrict__ rays_d, int d_stride,
                                     const float* __restrict__ add, const float* __restrict__ mul, const float* __restrict__ noise, float clampv,
                                     int white_bkgd, const float* __restrict__ d_rgb, float* __restrict__ d_raw, float* __restrict__ d_z,
                                     float* __restrict__ d_add, float* __restrict__ d_mul, int64_t n, int S) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;                                                          // wave-uniform
  const float* d = rays_d + i * d_stride;
  const float dn = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  const float g0 = d_rgb[i * 3], g1 = d_rgb[i * 3 + 1], g2 = d_rgb[i * 3 + 2];
  const float gsum = white_bkgd ? (g0 + g1 + g2) : 0.f;                         // rgb_map += 1 - sum_s w_s
  const bool clamped = clampv > 0.f;
  auto lane_val = [](float v, int j) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j)); };
  const int nchunks = (S + 63) / 64;
  // per-sample quantities of one chunk (this lane's sample): everything both passes need
  struct Smp { float r0, r1, r2, raw3, sg, dist, al, a, ml; bool on; int64_t e; int s; };
  auto load = [&](int c0) {
    Smp q;
    const int m = S - c0 < 64 ? S - c0 : 64;
    q.s = c0 + lane; q.on = lane < m;
    q.e = i * S + (q.on ? q.s : c0);
    const float4 rw = *(const float4*)(raw + q.e * 4);
    q.r0 = rw.x; q.r1 = rw.y; q.r2 = rw.z; q.raw3 = rw.w;
    float r3 = rw.w;
    if (clamped) r3 = fminf(fmaxf(r3, -clampv), clampv);
    float sg = r3;
    if (noise) sg += noise[q.e];
    if (add) sg += add[q.e];
    q.sg = sg;
    const float zc = z[q.e], zn = (q.on && q.s + 1 < S) ? z[q.e + 1] : 0.f;
    q.dist = ((q.s + 1 < S) ? (zn - zc) : 1e10f) * dn;
    q.a = 1.f - expf(-fmaxf(sg, 0.f) * q.dist);
    q.ml = mul ? mul[q.e] : 1.f;
    q.al = mul ? q.a * fmaxf(q.ml, 0.f) : q.a;
    return q;
  };
  // T at the start of chunk c = the product over the chunks before it, in sample order (one chunk for S <= 64: nothing to do)
  auto t_start = [&](int c) {
    float T = 1.f;
    for (int k = 0; k < c; ++k) {
      const Smp q = load(64 * k);
      const float x = 1.f - q.al + 1e-10f;
      for (int j = 0; j < 64; ++j) T *= lane_val(x, j);
    }
    return T;
  };
  // pass 2 (descending)
  float Q = 0.f, dd_next = 0.f;
  for (int c = nchunks - 1; c >= 0; --c) {
    const int c0 = 64 * c, m = S - c0 < 64 ? S - c0 : 64;
    const Smp q = load(c0);
    const float x = 1.f - q.al + 1e-10f;
    float Tpre = 0.f, Tr = t_start(c);
    for (int j = 0; j < m; ++j) { Tpre = lane == j ? Tr : Tpre; Tr *= lane_val(x, j); }
    float r0 = q.r0, r1 = q.r1, r2 = q.r2;
    const bool in0 = !clamped || fabsf(r0) <= clampv, in1 = !clamped || fabsf(r1) <= clampv, in2 = !clamped || fabsf(r2) <= clampv;
    if (clamped) { r0 = fminf(fmaxf(r0, -clampv), clampv); r1 = fminf(fmaxf(r1, -clampv), clampv); r2 = fminf(fmaxf(r2, -clampv), clampv); }
    const float c0v = sigmoid_f(r0), c1v = sigmoid_f(r1), c2v = sigmoid_f(r2);
    const float dws = g0 * c0v + g1 * c1v + g2 * c2v - gsum;
    // Q_s (the value the thread-per-ray loop had when it reached sample s) for this lane, then Q <- dws al + x Q, sample by sample from the top
    float Ql = 0.f;
    for (int j = m - 1; j >= 0; --j) {
      Ql = lane == j ? Q : Ql;
      Q = lane_val(dws, j) * lane_val(q.al, j) + lane_val(x, j) * Q;
    }
    const float dal = Tpre * (dws - Ql);
    const float w = q.al * Tpre;
    const bool in3 = !clamped || fabsf(q.raw3) <= clampv;
    const float ee = fmaxf(q.sg, 0.f);
    const float ex = expf(-ee * q.dist);
    const float da = mul ? dal * fmaxf(q.ml, 0.f) : dal;
    const float dsg = q.sg > 0.f ? da * q.dist * ex : 0.f;
    const float ddist = (q.s + 1 < S) ? da * ee * ex * dn : 0.f;               // the last interval (1e10) does not depend on z
    // d z_{s+1} = ddist_s - ddist_{s+1}: the neighbour's value from the next lane (the next chunk's first sample across a chunk boundary)
    float dd_up = __int_as_float(__builtin_amdgcn_ds_bpermute(((lane + 1) & 63) << 2, __float_as_int(ddist)));
    if (lane == m - 1) dd_up = dd_next;
    dd_next = lane_val(ddist, 0);
    if (q.on) {
      float4 o;
      o.x = in0 ? g0 * w * c0v * (1.f - c0v) : 0.f;
      o.y = in1 ? g1 * w * c1v * (1.f - c1v) : 0.f;
      o.z = in2 ? g2 * w * c2v * (1.f - c2v) : 0.f;
      o.w = in3 ? dsg : 0.f;
      *(float4*)(d_raw + q.e * 4) = o;
      if (d_mul) d_mul[q.e] = (mul && q.ml > 0.f) ? dal * q.a : 0.f;
      if (d_add) d_add[q.e] = add ? dsg : 0.f;
      if (d_z) {
        if (q.s + 1 < S) d_z[q.e + 1] = ddist - dd_up;
        if (q.s == 0) d_z[q.e] = -ddist;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ losses, Adam
// The three image losses of an iteration in one launch: block k computes loss[1 + k] = mean((pred_k - target)^2) and, if asked, d_pred_k =
// scale_k * 2 (pred_k - target) / numel (one block per loss: a fixed summation order); the block that finishes last adds them up,
// loss[0] = loss[1] + a_total (loss[2] + loss[3]), and re-arms the counter.
struct LossArgs {
  const float* pred[3]; float* d_pred[3]; float scale[3];
  const float* target; int64_t numel;
  float* loss; unsigned* counter; float a_total;
};
__global__ __launch_bounds__(1024) void losses_kernel(LossArgs a) {
  __shared__ float red[1024];
  const int k = blockIdx.x;
  const float* pred = a.pred[k];
  float* d_pred = a.d_pred[k];
  const float scale = a.scale[k];
  float s = 0.f;
  const float inv = 1.f / (float)a.numel;
  for (int64_t i = threadIdx.x; i < a.numel; i += blockDim.x) {
    const float e = pred[i] - a.target[i];
    s += e * e;
    if (d_pred) d_pred[i] = scale * 2.f * e * inv;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int st = blockDim.x / 2; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    __atomic_store_n((unsigned*)a.loss + 1 + k, __float_as_uint(red[0] * inv), __ATOMIC_RELAXED);
    __threadfence();
    if (atomicAdd(a.counter, 1u) == gridDim.x - 1) {             // every block's loss is visible now
      __threadfence();
      const unsigned* lu = (const unsigned*)a.loss;             // (device-scope loads: the other blocks' stores went to L2)
      const float l1 = __uint_as_float(__atomic_load_n(lu + 1, __ATOMIC_RELAXED)), l2 = __uint_as_float(__atomic_load_n(lu + 2, __ATOMIC_RELAXED)),
                  l3 = __uint_as_float(__atomic_load_n(lu + 3, __ATOMIC_RELAXED));
      a.loss[0] = l1 + a.a_total * (l2 + l3);
      *a.counter = 0;
    }
  }
}

// torch.optim.Adam (no amsgrad): g += wd p; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= lr/(1-b1^t) m / (sqrt(v)/sqrt(1-b2^t) + eps)
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n, float lr,
                            float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float gi = g[i];
    const float pi = p[i];
    if (wd != 0.f) gi = fmaf(wd, pi, gi);
    const float mi = m[i] + (1.f - b1) * (gi - m[i]);                   // lerp, as torch's single-tensor Adam
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - (lr / bc1) * (mi / denom);
  }
}

// one launch copies a batch into the trainer's staging buffers (n rays; S = samples per ray of jitter / noise; either may be absent)
struct StageArgs {
  const float *rays, *or_rays, *target, *jitter, *noise; const int64_t* ref_nos;
  float *d_rays, *d_or_rays, *d_target, *d_jitter, *d_noise; int64_t* d_ref_nos;
  int64_t n; int S;
  float* amax; int n_amax;              // the iteration's max-|gradient| slots (hgemm_kernel operand scaling): cleared here
};
__device__ __forceinline__ void stage_batch_body(const StageArgs& a, unsigned bid, unsigned nblk) {
  const int64_t stride = (int64_t)nblk * blockDim.x, t0 = bid * (int64_t)blockDim.x + threadIdx.x;
  for (int64_t i = t0; i < a.n_amax; i += stride) a.amax[i] = 0.f;
  for (int64_t i = t0; i < a.n * 11; i += stride) { a.d_rays[i] = a.rays[i]; a.d_or_rays[i] = a.or_rays[i]; }
  for (int64_t i = t0; i < a.n * 3; i += stride) a.d_target[i] = a.target[i];
  for (int64_t i = t0; i < a.n * 4; i += stride) a.d_ref_nos[i] = a.ref_nos[i];
  if (a.jitter) for (int64_t i = t0; i < a.n * a.S; i += stride) a.d_jitter[i] = a.jitter[i];
  if (a.noise) for (int64_t i = t0; i < a.n * a.S; i += stride) a.d_noise[i] = a.noise[i];
}

// The start of an iteration as ONE launch: whatever derived form of the parameters is stale — the fp16 planes (split_weights), the two chains'
// fragment streams and the backward chain's column norms — and the copy of the batch into the staging buffers.  All of them read only the
// parameters / the caller's batch and write disjoint buffers, so they are block ranges of one grid instead of five dependent launches of
// 5-12 us each (the stage-2 iteration at 4096 rays is 35 launches without a gap between them; ~4.5 us of every small one is the launch itself).
struct PrepArgs {
  SplitArgs split; TChainPackArgs pack; TChainBwdPackArgs packb; StageArgs stage;
  float* cmax_zero;                                             // the cmax array of the next refresh (tchain_norms_body)
  unsigned n_split, n_pack, n_packb, n_norm, n_stage;          // blocks of each part (0: not this time)
};
__global__ __launch_bounds__(TPB) void iter_prepare_kernel(PrepArgs a) {
  __shared__ float red[TPB];
  unsigned b = blockIdx.x;
  if (b < a.n_stage) { stage_batch_body(a.stage, b, a.n_stage); return; }
  b -= a.n_stage;
  if (b < a.n_norm) { tchain_norms_body(a.packb, (int)b, red, a.cmax_zero); return; }
  b -= a.n_norm;
  if (b < a.n_pack) { tchain_pack_body(a.pack, b); return; }
  b -= a.n_pack;
  if (b < a.n_packb) { tchain_pack_bwd_body(a.packb, b); return; }
  b -= a.n_packb;
  split_weights_body(a.split, b, a.n_split);
}

}  // namespace

// ------------------------------------------------------------------------------------------ trainer object
constexpr int N_AMAX = 64;
struct TLin {
  int in, out; size_t w, b;                      // offsets into the flat parameter array
  int gap = -1;                                  // >= 0: the layer's input buffer has one extra (zero) column at this index (the skip layer reads
  int in_x() const { return gap >= 0 ? in + 1 : in; }   // cat[embedding 63, pad, h 256]: the hidden part then starts 16-byte aligned)
};
constexpr int LD_C5 = 320, C5_H = 64;            // skip-layer input / gradient rows: [embedding 63 | 0 | h 256]
constexpr int LD_CV = 288;                       // views-layer input / gradient rows: [feature 256 | view embedding 27 | 5 x 0]

struct pnrf_trainer {
  int device = 0;
  int64_t max_rays = 0;
  int max_samples = 8;                          // samples per ray the NeRF-side workspaces are sized for (stage-1 exploration: up to 64)
  std::vector<TLin> L;                           // 0..6 sampler, 7..13 refine, 14..25 fine net (pts0..7, feature, alpha, views, rgb)
  size_t nparam = 0;
  float *P = nullptr, *G = nullptr, *M = nullptr, *V = nullptr;
  float *M2 = nullptr, *V2 = nullptr;            // second Adam state over the NeRF layers only (stage 1: `optimizer` next to `s_optimizer`)
  int64_t step = 0, step2 = 0;
  int dw_tile = 0;                               // 0: by shape and row count; 64 / 128: force that weight-gradient kernel where it applies
  int64_t dw_wide_min_rows = 32768;              // grouped split-fp16 weight gradients: 256 x 128 tiles (dwh_body_wide) from this many rows on (pnrf_trainer_set_dw_kernel(.., tile 256 / 255))
  int64_t dw128_min_rows = 65536;
  // hipGraph replay of an iteration (pnrf_trainer_set_graph): the batch is copied into the trainer's own staging buffers by one kernel, so
  // every pointer and scalar argument inside the captured launch sequence is fixed; one instantiated graph per configuration key
  struct GraphKey {
    int kind, n_mult, dir1, jitter_dir, white_bkgd, layout, nv, Hf, Wf, has_jitter, has_noise;
    int64_t n;
    float eps, a_mmrgb, clamp;
    const void *img4, *poses, *K;
    const void* cmax;                            // which of the two column-norm arrays the captured backward chain reads (tchain_norms_body)
    hipStream_t stream;
  };
  struct GraphEntry { GraphKey key; hipGraphExec_t exec; };
  std::vector<GraphEntry> graphs;
  bool use_graph = false;                        // measured on one MI355X, 4096-ray stage-2 iteration: 2.98 ms kernel by kernel, 3.13 ms replayed
  hipStream_t own_stream = nullptr;              // the legacy default stream cannot be captured: iterations submitted on it run on this one,
  hipEvent_t ev_in = nullptr, ev_out = nullptr;  // ordered against the caller's stream by two events
  float *st_rays = nullptr, *st_or_rays = nullptr, *st_target = nullptr, *st_jitter = nullptr, *st_noise = nullptr;
  int64_t* st_ref_nos = nullptr;
  std::vector<void*> allocs;
  // workspaces
  float *mm_input, *s_h[6], *s_y, *depth_sorted, *add_s, *mul_s, *mm_rgb;
  int64_t* sort_idx;
  float *refine_in, *r_h[6], *r_y, *z_pre, *z, *pts, *rgb0;
  float *n_a[4], *n_c5, *n_a5, *n_a6, *n_a7, *n_cv, *n_hv, *raw, *rgb_map, *wts;
  float *d_rgb_map, *d_raw, *d_hv, *d_cv, *d_a, *d_b, *d_c5, *d_e0, *d_pts, *d_z, *d_add, *d_mul, *d_depth, *d_ry, *d_sy, *d_rgb0, *d_mmrgb,
      *d_h0, *d_h1, *d_hk[6], *dw_pool, *loss;   // d_hk: one gradient buffer per hidden layer of an ELU net (layer chains); d_h0 / d_h1 = d_hk[0 / 1]
  float* w_gapped = nullptr;                     // fp32 copy [out][in + 1] of the skip layer's weights with the zero column of its input layout
  _Float16* planes = nullptr;                    // fp16 hi / lo planes of every layer's weights, both orientations (pnrf_hgemm.h)
  SplitArgs split;
  bool planes_stale = true;                      // parameters changed since the planes were last written (set through params_changed())
  bool nerf_planes_stale = true;                 // ... the fine net's planes (not refreshed while both of its chains run on the engine: nobody reads them)
  bool streams_stale = true;                     // ... the chains' fragment streams
  bool use_f16 = true;                           // split-fp16 layer products (default) or the exact-fp32 MFMA kernels throughout
  _Float16* tc_stream = nullptr;                 // the fine net's pts0 .. feature weights as the fused-MLP engine's fragment stream (pnrf_tchain.h)
  TChainPackArgs tc_pack;
  _Float16* tb_stream = nullptr;                 // ... and the transposed layers' stream of the input-gradient chain (tchain_bwd_kernel)
  TChainBwdPackArgs tb_pack;
  float* cmax2 = nullptr;                        // the backward chain's column norms, two arrays used in turn (tb_pack.cmax = the current one)
  bool tc_ok = true;                             // workspaces small enough for the chains' 32-bit row offsets
  uint2* tc_mask = nullptr;                      // ReLU masks of pts0 .. pts7 (written by the forward chain, read by the backward chain)
  float* dz_x[6] = {};                           // dZ5 .. dZ0 of the backward chain (dZ7 = d_a, dZ6 = d_b)
  int nerf_fwd = 0;                              // fine net's forward from 8192 rows on: 0 one engine launch (tchain_fwd_kernel), 3 two 64-row layer chains
                                                 // (hgemm_wchain_kernel), 2 one product launch per layer (pnrf_trainer_set_products)
  float* amax = nullptr;                         // [N_AMAX] max |dL/dZ| per gradient buffer write, this iteration
  size_t pool_cap = 0, pool_used = 0;            // split-K partials of the iteration's weight gradients (floats); reset per iteration
  DwJobs jobs;                                   // ... and who sums them (dw_reduce_kernel, one launch per iteration)
  int grp_last_n = 0; unsigned grp_last_wide = 0;   // the most recent grouped launch (pnrf_trainer_dw_group_info)
  DwhGroupArgs grp;                              // split-fp16 weight gradients waiting for the iteration's one grouped launch (flush_dw_group): the 256-wide
  int grp_blocks = 0;                            // layers of all three nets; their gradient / activation buffers stay untouched until then
  float* d_hs[6] = {};                           // the sampler net's hidden gradients (d_hk holds the refine net's until the grouped launch)
};

namespace {

inline void params_changed(pnrf_trainer* t) { t->planes_stale = true; t->nerf_planes_stale = true; t->streams_stale = true; }
// Does the fine net's forward / backward pass of an iteration over R sample rows run on the fused-MLP engine (tchain_fwd_kernel /
// tchain_bwd_kernel + the grouped weight gradients)?  ONE definition: nerf_forward, nerf_backward and run_iteration's refresh of the
// derived weight forms (fp16 planes vs fragment streams) must agree, or an iteration reads stale weights.
inline bool engine_fwd(const pnrf_trainer* t, int64_t R) { return t->use_f16 && t->nerf_fwd == 0 && R >= 8192 && t->tc_ok; }
inline bool engine_bwd(const pnrf_trainer* t, int64_t R) { return engine_fwd(t, R) && t->dw_tile == 0; }

template <class T>
int dev_alloc(pnrf_trainer* t, T** p, size_t count) {
  void* q = nullptr;
  hipError_t e = hipMalloc(&q, count * sizeof(T));
  if (e != hipSuccess) { set_error("pnrf_trainer: hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e)); return (int)e; }
  t->allocs.push_back(q);
  *p = (T*)q;
  return 0;
}
#define T_ALLOC(ptr, count) do { int rc_ = dev_alloc(t, &(ptr), (size_t)(count)); if (rc_) return rc_; } while (0)
inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

template <int MODE>
int launch_tgemm(const GemmArgs& a, hipStream_t s) {
  const bool vec = a.lda % 4 == 0 && aligned16(a.A) && a.K % 4 == 0 && (MODE == MODE_NN || (a.ldb % 4 == 0 && aligned16(a.B)));
  const int tiles_n = (a.N + 255) / 256;
  // 64-row tiles when that still gives every CU a workgroup; below that 32-row and, for the 4 096-row layers of the sampler / refine nets,
  // 16-row tiles (256 workgroups: these products are latency-bound, more workgroups in flight is what helps)
  const int64_t t64 = ((a.M + 63) / 64) * tiles_n, t32 = ((a.M + 31) / 32) * tiles_n;
  const int mi = t64 >= 256 ? 4 : (t32 >= 256 ? 2 : 1);
  const dim3 grid((unsigned)(((a.M + 16 * mi - 1) / (16 * mi)) * tiles_n));
#define PNRF_TG(MI_) do { if (vec) hipLaunchKernelGGL((tgemm_kernel<MI_, MODE, true>), grid, dim3(256), 0, s, a); \
                          else hipLaunchKernelGGL((tgemm_kernel<MI_, MODE, false>), grid, dim3(256), 0, s, a); } while (0)
  if (mi == 4) PNRF_TG(4); else if (mi == 2) PNRF_TG(2); else PNRF_TG(1);
#undef PNRF_TG
  PNRF_LAUNCH_CHECK();
  return 0;
}

// dW[out,in] = dZ^T X and db[out] = column sums of dZ: dw_splitk*_kernel writes the split-K partials into a fresh slice of the pool and
// queues their summation; flush_dw_reduce() at the end of the backward pass does all of them in one launch.
// `in` counts the columns of X; with gap >= 0 column `gap` of X is padding and dW has in - 1 columns
// defer != NULL: if the split-fp16 kernel is the one to use, its arguments and grid go to *defer instead of a launch (layer_bwd dispatches it
// together with the layer's input-gradient product)
struct DwDefer { DwhArgs args; int tiles, splits; bool set; int wgs = 128; bool allow_wide = false, wide = false; };   // wgs (in): workgroups the deferred gradient should
                                                                                                                   // spread over; allow_wide (in): the grouped launch may use 256 x 128 tiles
int gemm_dw(pnrf_trainer* t, const float* X, int ldx, const float* dZ, int ldz, const float* dz_amax, float* dW, float* db, int in, int gap, int out,
            int64_t R, hipStream_t s, DwDefer* defer = nullptr) {
  PNRF_REQUIRE(out <= DB_MAX_OUT, PNRF_E_SHAPE, "pnrf_trainer: layer output %d wider than the bias-partial buffer (%d)", out, DB_MAX_OUT);
  PNRF_REQUIRE(t->jobs.n < DW_MAX_JOBS, PNRF_E_STATE, "pnrf_trainer: more than %d weight gradients in one iteration", DW_MAX_JOBS);
  const int numel = out * in;
  const bool can128 = out % 128 == 0 && in % 128 == 0 && ldx % 4 == 0 && ldz % 4 == 0 && aligned16(X) && aligned16(dZ);
  // split-fp16 kernel (128 x 128 tiles, 64-row chunks) wherever the gradient's magnitude is on record and the output is not a narrow head
  const bool use_h = t->use_f16 && dz_amax && out >= 64 && in >= 32 && t->dw_tile == 0;
  // the 1 .. 4-wide heads: streaming kernel, one workgroup per split
  const bool use_head = out <= HEAD_MAX && gap < 0 && (in == 128 || in == 256) && ldx % 4 == 0 && aligned16(X);
  const bool use128 = !use_h && !use_head && can128 && t->dw_tile != 64 && R >= t->dw128_min_rows;
  const int max_splits = out <= HEAD_MAX ? HEAD_MAX_SPLITS : (out % 128 == 0 && in % 128 == 0) ? DW128_MAX_SPLITS : DW_MAX_SPLITS;   // what the pool was sized for
  int tiles;
  bool wide_h = false;
  int64_t splits, rows_per;
  // the partials are written and then read again by the reduction: hold them to half of the operand bytes (26 layers x 64 .. 128 splits
  // were 1 GB per iteration, the reduction kernel alone 0.22 ms of 2.5)
  const int64_t by_traffic = R * (in + out) / (2 * (int64_t)numel) > 1 ? R * (in + out) / (2 * (int64_t)numel) : 1;
  if (use_head) {
    tiles = 1;
    splits = R / 128 < HEAD_MAX_SPLITS ? R / 128 : HEAD_MAX_SPLITS;      // partials are a few KB each: as many workgroups as there are CUs
  } else if (use_h) {
    // 256 x 128 tiles (dwh_body_wide: dZ is the only operand read twice) for the 256-wide gradients of the grouped launch from 32 768 rows on — below that
    // the launch wants the workgroups more than the bytes
    wide_h = defer && defer->allow_wide && out == 256 && in >= 128 && ldz % 4 == 0 && ldx % 4 == 0 && aligned16(X) && aligned16(dZ) && R >= t->dw_wide_min_rows;
    const int tiles_sq = ((out + 127) / 128) * ((in + 127) / 128);
    tiles = wide_h ? (in + 127) / 128 : tiles_sq;
    splits = ((defer ? defer->wgs : 256) + tiles_sq - 1) / tiles_sq;   // one workgroup per CU (half of the CUs when the launch is shared with the dX product) ...
                                                                       // (the wide form keeps the split count — and with it the partials' traffic — of the square tiles)
    if (splits > by_traffic) splits = by_traffic;               // ... unless the partials would outweigh the operands
    const int64_t by_rows = (R + DH_KC - 1) / DH_KC;
    if (splits > by_rows) splits = by_rows;
    if (splits > max_splits) splits = max_splits;
  } else if (use128) {
    tiles = (out / 128) * (in / 128);
    splits = (512 + tiles - 1) / tiles;                         // two workgroups per CU
    const int64_t by_rows = R / 256;
    if (splits > by_rows) splits = by_rows;
    if (splits > by_traffic) splits = by_traffic;
    if (splits > DW128_MAX_SPLITS) splits = DW128_MAX_SPLITS;
  } else {
    tiles = ((out + DW_TILE - 1) / DW_TILE) * ((in + DW_TILE - 1) / DW_TILE);
    // enough workgroups for 4 per CU (they hide each other's load latency), at least 128 rows per split, at most DW_MAX_SPLITS partials
    splits = (1024 + tiles - 1) / tiles;
    const int64_t by_rows = (R + 127) / 128;
    if (splits > by_rows) splits = by_rows;
    if (splits > 2 * by_traffic) splits = 2 * by_traffic;       // (64 x 64 tiles: four times the workgroups per split)
    if (splits > DW_MAX_SPLITS) splits = DW_MAX_SPLITS;
  }
  if (splits < 1) splits = 1;
  const int rows_q = use_head ? 8 : use_h ? DH_KC : (use128 ? DW128_ROWS : DW_ROWS);
  rows_per = (R + splits - 1) / splits;
  rows_per = (rows_per + rows_q - 1) / rows_q * rows_q;
  splits = (R + rows_per - 1) / rows_per;
  const size_t need = (size_t)splits * ((size_t)numel + out);
  PNRF_REQUIRE(t->pool_used + need <= t->pool_cap, PNRF_E_STATE, "pnrf_trainer: weight-gradient partials exceed their pool (%zu + %zu > %zu floats)",
               t->pool_used, need, t->pool_cap);
  float* part = t->dw_pool + t->pool_used;
  float* db_part = part + (size_t)splits * numel;
  t->pool_used += (need + 3) & ~(size_t)3;                      // slices stay 16-byte aligned
  if (defer) defer->set = false;
  if (use_head) {
    HeadArgs h = {};
    h.X = X; h.ldx = ldx; h.dZ = dZ; h.ldz = ldz; h.part = part; h.db_part = db_part; h.rows_per = rows_per; h.M = R; h.n = out; h.K = in;
    hipLaunchKernelGGL(head_dw_kernel, dim3((unsigned)splits), dim3(256), 0, s, h);
  } else if (use_h) {
    DwhArgs h = {dZ, ldz, X, ldx, part, db_part, out, in, R, rows_per, dz_amax};
    if (defer) { defer->args = h; defer->tiles = tiles; defer->splits = (int)splits; defer->set = true; defer->wide = wide_h; }
    else hipLaunchKernelGGL(dwh_kernel, dim3(tiles, (unsigned)splits), dim3(512), 0, s, h);
  } else if (use128)
    hipLaunchKernelGGL(dw_splitk128_kernel, dim3(tiles, (unsigned)splits), dim3(256), 0, s, dZ, ldz, X, ldx, part, out, in, R, rows_per, db_part);
  else
    hipLaunchKernelGGL(dw_splitk_kernel, dim3(tiles, (unsigned)splits), dim3(256), 0, s, dZ, ldz, X, ldx, part, out, in, R, rows_per, db_part);
  PNRF_LAUNCH_CHECK();
  DwJob& J = t->jobs.j[t->jobs.n];
  J.part = part; J.db_part = db_part; J.dW = dW; J.db = db; J.numel = numel; J.splits = (int)splits; J.out = out;
  J.in_src = in; J.in_dst = gap >= 0 ? in - 1 : in; J.gap = gap;
  J.db_blocks = (out * (splits >= 64 ? 8 : 1) + TPB - 1) / TPB;   // dw_reduce_kernel: eight lanes per element from 64 splits on
  J.blocks = grid_for(gap < 0 && numel % 4 == 0 ? numel / 4 : numel) + J.db_blocks;
  J.block0 = t->jobs.n ? t->jobs.j[t->jobs.n - 1].block0 + t->jobs.j[t->jobs.n - 1].blocks : 0;
  ++t->jobs.n;
  return 0;
}
void begin_dw(pnrf_trainer* t) { t->jobs.n = 0; t->pool_used = 0; t->grp.n = 0; t->grp_blocks = 0; }
// a deferred split-fp16 weight gradient joins the iteration's grouped launch (dwh_group_kernel)
int group_dw(pnrf_trainer* t, const DwDefer& d) {
  PNRF_REQUIRE(t->grp.n < DH_GROUP_MAX, PNRF_E_STATE, "pnrf_trainer: more than %d grouped weight gradients", DH_GROUP_MAX);
  DwhGroupArgs& g = t->grp;
  g.j[g.n] = d.args; g.first[g.n] = t->grp_blocks; g.tiles[g.n] = d.tiles; g.splits[g.n] = d.splits;
  if (g.n == 0) g.wide = 0;
  if (d.wide) g.wide |= 1u << g.n;
  t->grp_blocks += (d.tiles * d.splits + 7) & ~7;               // jobs start on XCD 0 (dwh_group_kernel)
  ++g.n;
  return 0;
}
int flush_dw_group(pnrf_trainer* t, hipStream_t s) {
  if (!t->grp.n) return 0;
  t->grp.first[t->grp.n] = t->grp_blocks;
  hipLaunchKernelGGL(dwh_group_kernel, dim3((unsigned)t->grp_blocks), dim3(512), 0, s, t->grp);
  PNRF_LAUNCH_CHECK();
  t->grp_last_n = t->grp.n; t->grp_last_wide = t->grp.wide;
  t->grp.n = 0; t->grp_blocks = 0;
  return 0;
}
int flush_dw_reduce(pnrf_trainer* t, hipStream_t s) {
  { int rc = flush_dw_group(t, s); if (rc) return rc; }
  if (!t->jobs.n) return 0;
  const DwJob& last = t->jobs.j[t->jobs.n - 1];
  hipLaunchKernelGGL(dw_reduce_kernel, dim3((unsigned)(last.block0 + last.blocks)), dim3(TPB), 0, s, t->jobs);
  PNRF_LAUNCH_CHECK();
  begin_dw(t);
  return 0;
}

// 64-row tiles while they give every CU a workgroup, else 32 / 16 rows (the 4 096-row layers of the sampler / refine nets); one persistent
// workgroup per CU, each taking every gx-th row tile of its column block
void hgemm_grid(const HGemmArgs& a, int* mi, int* gx, int* gy) {
  const int tiles_n = (a.N + 255) / 256;
  const int64_t t64 = (a.M + 63) / 64, t32 = (a.M + 31) / 32;
  *mi = t64 * tiles_n >= 256 ? 4 : (t32 * tiles_n >= 256 ? 2 : 1);   // (4096 rows: 16-row tiles 1.355 ms per iteration, 32: 1.368, 64: 1.423)
  const int64_t ntiles = (a.M + 16 * *mi - 1) / (16 * *mi);
  const int per_col = 256 / tiles_n > 0 ? 256 / tiles_n : 1;
  *gx = (int)(ntiles < per_col ? ntiles : per_col);
  *gy = tiles_n;
}
int launch_hgemm(const HGemmArgs& a, hipStream_t s) {
  int mi, gx, gy;
  hgemm_grid(a, &mi, &gx, &gy);
  const dim3 grid((unsigned)gx, (unsigned)gy);
  if (mi == 4) hipLaunchKernelGGL((hgemm_kernel<4>), grid, dim3(512), 0, s, a);
  else if (mi == 2) hipLaunchKernelGGL((hgemm_kernel<2>), grid, dim3(512), 0, s, a);
  else hipLaunchKernelGGL((hgemm_kernel<1>), grid, dim3(512), 0, s, a);
  PNRF_LAUNCH_CHECK();
  return 0;
}
// the layer's input-gradient product and its weight gradient as one launch
int launch_layer_bwd(const HGemmArgs& a, const DwDefer& d, hipStream_t s) {
  int mi, gx, gy;
  hgemm_grid(a, &mi, &gx, &gy);
  // both parts in ONE wave of workgroups (a CU holds one of either kind): the product's persistent grid shrinks to the CUs the weight gradient
  // leaves free, with larger row tiles if that keeps every workgroup at >= 1 tile
  const int dw_wgs = d.tiles * d.splits, room = 256 - dw_wgs;
  if (room >= 64 * gy && gx * gy > room) {
    gx = room / gy;
    while (mi < 4 && (a.M + 16 * mi - 1) / (16 * mi) > 2 * (int64_t)gx) mi *= 2;       // few workgroups: take more rows per tile
    const int64_t ntiles = (a.M + 16 * mi - 1) / (16 * mi);
    if (gx > ntiles) gx = (int)ntiles;
  }
  const dim3 grid((unsigned)(gx * gy + d.tiles * d.splits));
  if (mi == 4) hipLaunchKernelGGL((layer_bwd_kernel<4>), grid, dim3(512), 0, s, a, d.args, gx, gy, d.tiles);
  else if (mi == 2) hipLaunchKernelGGL((layer_bwd_kernel<2>), grid, dim3(512), 0, s, a, d.args, gx, gy, d.tiles);
  else hipLaunchKernelGGL((layer_bwd_kernel<1>), grid, dim3(512), 0, s, a, d.args, gx, gy, d.tiles);
  PNRF_LAUNCH_CHECK();
  return 0;
}
// which product kernel: the split-fp16 one wherever the tile shape fits (>= 64 output columns, a contraction of >= 32), the exact-fp32 one
// for the narrow heads and, in the backward pass, for gradients whose magnitude nobody recorded
inline bool hgemm_fits(const pnrf_trainer* t, int n, int k, int64_t rows, int ld, const float* C, int ldc, const float* H, int ldh, int act_col0) {
  // the kernel addresses its row operand with 32-bit byte offsets and reads / writes C, H, bias as aligned float4
  return t->use_f16 && n >= 64 && k >= 32 && rows * (int64_t)ld * 4 < ((int64_t)1 << 32) && n % 4 == 0 && ldc % 4 == 0 && aligned16(C) &&
         (!H || (ldh % 4 == 0 && aligned16(H) && act_col0 % 4 == 0));
}

// Y = act(X W^T + b): one kernel
// arguments of the split-fp16 forward product of layer li; false if the layer has to take another kernel
bool fwd_hgemm_args(pnrf_trainer* t, int li, const float* X, int ldx, float* Y, int ldy, int64_t R, int act, HGemmArgs* h) {
  const TLin& l = t->L[li];
  const int K = l.in_x();
  if (!(hgemm_fits(t, l.out, K, R, ldx, Y, ldy, nullptr, 0, 0) && aligned16(t->P + l.b))) return false;
  const SplitLayer& sl = t->split.l[li];
  *h = HGemmArgs{};
  h->A = X; h->lda = ldx; h->Bh = t->planes + sl.fwd; h->Bl = h->Bh + sl.plane_fwd; h->ldb = sl.ld_fwd; h->n_pad = (l.out + 63) / 64 * 64;
  h->C = Y; h->ldc = ldy; h->M = R; h->N = l.out; h->K = K; h->bwd = 0; h->bias = t->P + l.b; h->act = act;
  return true;
}
int layer_fwd(pnrf_trainer* t, int li, const float* X, int ldx, float* Y, int ldy, int64_t R, int act, hipStream_t s) {
  const TLin& l = t->L[li];
  const int K = l.in_x();
  if (l.out <= HEAD_MAX && l.gap < 0 && (K == 128 || K == 256) && ldx % 4 == 0 && aligned16(X)) {
    HeadArgs h = {};
    h.X = X; h.ldx = ldx; h.W = t->P + l.w; h.bias = t->P + l.b; h.Y = Y; h.ldy = ldy; h.M = R; h.n = l.out; h.K = K;
    PNRF_REQUIRE(act == T_ACT_NONE, PNRF_E_STATE, "pnrf_trainer: the narrow heads have no activation");
    const int64_t wgs = (R + 4 * (256 / K) - 1) / (4 * (256 / K));
    hipLaunchKernelGGL(head_fwd_kernel, dim3((unsigned)(wgs < 2048 ? wgs : 2048)), dim3(256), 0, s, h);
    PNRF_LAUNCH_CHECK();
    return 0;
  }
  HGemmArgs h;
  if (fwd_hgemm_args(t, li, X, ldx, Y, ldy, R, act, &h)) return launch_hgemm(h, s);
  GemmArgs a = {};
  a.A = X; a.lda = ldx; a.B = l.gap >= 0 ? t->w_gapped : t->P + l.w; a.ldb = K; a.C = Y; a.ldc = ldy; a.M = R; a.N = l.out; a.K = K;
  a.bias = t->P + l.b; a.act = act;
  return launch_tgemm<MODE_NT>(a, s);
}
// 16-row tiles while they fit one wave of workgroups (4096 rows: 1.227 ms per iteration; with 32-row tiles 1.254 — the chain is bound by its
// latencies, not by the weight planes every workgroup streams from L2), 32-row tiles beyond
void launch_rchain(const RChainArgs& c, int64_t N, hipStream_t s) {
  if (N > 4096) hipLaunchKernelGGL((hgemm_rchain_kernel<2>), dim3((unsigned)((N + 31) / 32)), dim3(512), 0, s, c);
  else hipLaunchKernelGGL((hgemm_rchain_kernel<1>), dim3((unsigned)((N + 15) / 16)), dim3(512), 0, s, c);
}
inline int trainer_num_cu() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}
// wide layers: two 4-wave workgroups per CU, persistent over 64-row tiles (hgemm_wchain_kernel)
void launch_wchain(const RChainArgs& c, int64_t N, hipStream_t s) {
  const int64_t ntiles = (N + 63) / 64;
  const int64_t cap = 2 * (int64_t)trainer_num_cu();
  hipLaunchKernelGGL((hgemm_wchain_kernel<4>), dim3((unsigned)(ntiles < cap ? ntiles : cap)), dim3(256), 0, s, c);
}
// rows few enough that a layer product is bound by launch and pipeline-fill latency: walk the layers in one launch (hgemm_rchain_kernel)
inline bool chain_rows(int64_t R) { return (R + 15) / 16 <= 512; }
// the six hidden layers of an ELU net: h[k] = ELU(h[k - 1] W^T + b), h[-1] = x0
int elu_net_forward(pnrf_trainer* t, int first, const float* x0, int in0, float* const* h, int64_t N, hipStream_t s) {
  if (chain_rows(N)) {
    // layer 0 (288 / 144 -> 256) through the general product body, layers 1 .. 5 (256 -> 256) handed over in LDS: one launch
    RChainArgs c = {};
    bool ok = fwd_hgemm_args(t, first, x0, in0, h[0], 256, N, T_ACT_ELU, &c.first);
    for (int k = 1; k < 6 && ok; ++k) {
      HGemmArgs a{};
      ok = fwd_hgemm_args(t, first + k, h[k - 1], 256, h[k], 256, N, T_ACT_ELU, &a) && a.K == 256 && a.N == 256;
      c.l[k - 1] = RChainLayer{a.Bh, a.Bl, a.ldb, a.n_pad, a.bias, h[k], nullptr, nullptr, T_ACT_ELU, 256, 256};
    }
    if (ok) {
      c.has_first = 1; c.X0 = h[0]; c.x0_amax = nullptr; c.n = 5; c.M = N; c.bwd = 0;
      launch_rchain(c, N, s);
      PNRF_LAUNCH_CHECK();
      return 0;
    }
  }
  const float* x = x0; int ldx = in0;
  for (int k = 0; k < 6; ++k) { int rc = layer_fwd(t, first + k, x, ldx, h[k], 256, N, T_ACT_ELU, s); if (rc) return rc; x = h[k]; ldx = 256; }
  return 0;
}
// dZ (row stride ldz) = dL/dZ of layer li (its activation derivative was applied by whoever produced it); X = the layer's saved input.
// Accumulates the weight / bias gradients and, unless dX == nullptr, writes dX = (beta dX + dZ W) * act'(Hprev) — Hprev = saved output of
// the layer (activation prev_act) that produced the columns >= act_col0 of X — i.e. dL/dZ of that layer, ready for its own layer_bwd.
// dz_amax: device scalar holding max |dZ| (left there by the product that wrote dZ) or NULL if unknown; dx_amax: where to leave max |dX|.
// arguments of the split-fp16 input-gradient product of layer li (the caller has checked hgemm_fits)
void bwd_hgemm_args(pnrf_trainer* t, int li, const float* dZ, int ldz, const float* dz_amax, float* dX, int lddx, float* dx_amax, float beta, int64_t R,
                    int prev_act, const float* Hprev, int ldh, int act_col0, int n_first, HGemmArgs* h) {
  const TLin& l = t->L[li];
  const SplitLayer& sl = t->split.l[li];
  const int N = l.in_x() - n_first;
  *h = HGemmArgs{};
  // fragment-major plane: input column n_first starts (n_first / 16) blocks of 16 rows x ld_bwd in
  h->A = dZ; h->lda = ldz; h->Bh = t->planes + sl.bwd + (size_t)(n_first >> 4) * 16 * sl.ld_bwd; h->Bl = h->Bh + sl.plane_bwd; h->ldb = sl.ld_bwd;
  h->n_pad = (N + 63) / 64 * 64;
  h->C = dX; h->ldc = lddx; h->M = R; h->N = (N + 3) & ~3; h->K = l.out; h->bwd = 1;
  h->act = prev_act; h->H = Hprev; h->ldh = ldh; h->act_col0 = act_col0; h->beta = beta; h->a_amax = dz_amax; h->c_amax = dx_amax;
}
// n_first (a multiple of 16): the input gradient is only wanted from that input column on — dX, Hprev and act_col0 then refer to column n_first
// (the skip layer without a position gradient: the 64 leading columns of its input are the embedding, whose gradient nobody reads)
int layer_bwd(pnrf_trainer* t, int li, const float* dZ, int ldz, const float* dz_amax, const float* X, int ldx, float* dX, int lddx, float* dx_amax,
              float beta, int64_t R, int prev_act, const float* Hprev, int ldh, int act_col0, hipStream_t s, int n_first = 0) {
  const TLin& l = t->L[li];
  const int N = l.in_x() - n_first;
  const int N4 = (N + 3) & ~3;                                  // the split-fp16 kernel stores whole float4s: the row padding of dX takes the rest (zeros)
  const bool dx_h = dX && dz_amax && N4 <= lddx &&
                    hgemm_fits(t, N4, l.out, R, ldz, dX, lddx, prev_act != T_ACT_NONE ? Hprev : nullptr, ldh, act_col0);
  DwDefer dw;
  dw.set = false;
  int rc = gemm_dw(t, X, ldx, dZ, ldz, dz_amax, t->G + l.w, t->G + l.b, l.in_x(), l.gap, l.out, R, s, dx_h ? &dw : nullptr);
  if (rc || !dX) return rc;
  if (l.out <= HEAD_MAX && l.gap < 0 && (N == 128 || N == 256) && lddx % 4 == 0 && aligned16(dX) &&
      (prev_act == T_ACT_NONE || (ldh % 4 == 0 && aligned16(Hprev) && act_col0 == 0))) {
    HeadArgs h = {};
    h.W = t->P + l.w; h.dZ = dZ; h.ldz = ldz; h.dX = dX; h.lddx = lddx; h.H = Hprev; h.ldh = ldh; h.act = prev_act; h.beta = beta; h.dx_amax = dx_amax;
    h.M = R; h.n = l.out; h.K = N;
    const int64_t wgs = (R + 256 / (N / 4) - 1) / (256 / (N / 4));
    hipLaunchKernelGGL(head_dx_kernel, dim3((unsigned)(wgs < 2048 ? wgs : 2048)), dim3(256), 0, s, h);
    PNRF_LAUNCH_CHECK();
    return 0;
  }
  if (dx_h) {
    HGemmArgs h;
    bwd_hgemm_args(t, li, dZ, ldz, dz_amax, dX, lddx, dx_amax, beta, R, prev_act, Hprev, ldh, act_col0, n_first, &h);
    return dw.set ? launch_layer_bwd(h, dw, s) : launch_hgemm(h, s);
  }
  GemmArgs a = {};
  a.A = dZ; a.lda = ldz; a.B = (l.gap >= 0 ? t->w_gapped : t->P + l.w) + n_first; a.ldb = l.in_x(); a.C = dX; a.ldc = lddx; a.M = R; a.N = N; a.K = l.out;
  a.act = prev_act; a.H = Hprev; a.ldh = ldh; a.act_col0 = act_col0; a.beta = beta; a.c_amax = dx_amax;
  return launch_tgemm<MODE_NN>(a, s);
}
constexpr int L_S = 0, L_R = 7, L_N = 14, L_FEAT = 22, L_ALPHA = 23, L_VIEWS = 24, L_RGB = 25, N_LAYERS = 26;

}  // namespace

// ------------------------------------------------------------------------------------------ C ABI: operators
extern "C" int pnrf_composite_bwd(const float* raw, const float* z, const float* rays_d, int d_stride, const float* add, const float* mul,
                                  const float* noise, float clampv, int white_bkgd, const float* d_rgb, float* d_raw, float* d_z, float* d_add,
                                  float* d_mul, int64_t n, int s, void* stream) {
  PNRF_REQUIRE(n >= 0 && s >= 1 && d_stride >= 3, PNRF_E_ARG, "pnrf_composite_bwd: bad sizes");
  if (n == 0) return 0;
  PNRF_REQUIRE(raw && z && rays_d && d_rgb && d_raw, PNRF_E_ARG, "pnrf_composite_bwd: null pointer");
  PNRF_REQUIRE((((uintptr_t)raw | (uintptr_t)d_raw) & 15) == 0, PNRF_E_ARG, "pnrf_composite_bwd: raw and d_raw must be 16-byte aligned (16-byte loads / stores per sample)");
  PNRF_REQUIRE((add == nullptr) == (mul == nullptr), PNRF_E_ARG, "pnrf_composite_bwd: add and mul go together");
  hipLaunchKernelGGL(composite_bwd_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, raw, z, rays_d, d_stride, add, mul, noise, clampv,
                     white_bkgd, d_rgb, d_raw, d_z, d_add, d_mul, n, s);
  PNRF_LAUNCH_CHECK();
  return 0;
}

extern "C" int pnrf_posenc_bwd(const float* x, const float* d_out, float* d_x, int64_t n, int n_freq, void* stream) {
  PNRF_REQUIRE(n >= 0 && n_freq >= 0 && n_freq <= 16 && (n == 0 || (x && d_out && d_x)), PNRF_E_ARG, "pnrf_posenc_bwd: bad arguments");
  if (n == 0) return 0;
  hipLaunchKernelGGL(posenc_bwd_kernel, dim3(grid_for(n * 3)), dim3(TPB), 0, (hipStream_t)stream, x, d_out, 3 + 6 * n_freq, (const float*)nullptr, 0,
                     d_x, n, n_freq);
  PNRF_LAUNCH_CHECK();
  return 0;
}

extern "C" int pnrf_sampler_head_fwd(const float* y, const float* rays, float* depth_sorted, int64_t* sort_idx, float* add_sorted, float* mul_sorted,
                                     float* mm_rgb, int64_t n, void* stream) {
  PNRF_REQUIRE(n >= 0 && (n == 0 || (y && rays && depth_sorted && sort_idx && add_sorted && mul_sorted)), PNRF_E_ARG, "pnrf_sampler_head_fwd: bad arguments");
  if (n == 0) return 0;
  hipLaunchKernelGGL(sampler_head_fwd_kernel, dim3(grid_for(n, 64)), dim3(64), 0, (hipStream_t)stream, y, rays, depth_sorted, sort_idx, add_sorted,
                     mul_sorted, mm_rgb, n);
  PNRF_LAUNCH_CHECK();
  return 0;
}
extern "C" int pnrf_sampler_head_bwd(const float* y, const float* rays, const int64_t* sort_idx, const float* d_depth_sorted, const float* d_add_sorted,
                                     const float* d_mul_sorted, const float* d_mm_rgb, float* d_y, int64_t n, void* stream) {
  PNRF_REQUIRE(n >= 0 && (n == 0 || (y && rays && sort_idx && d_depth_sorted && d_add_sorted && d_mul_sorted && d_y)), PNRF_E_ARG,
               "pnrf_sampler_head_bwd: bad arguments");
  if (n == 0) return 0;
  hipLaunchKernelGGL(sampler_head_bwd_kernel, dim3(grid_for(n, 64)), dim3(64), 0, (hipStream_t)stream, y, rays, sort_idx, d_depth_sorted, d_add_sorted,
                     d_mul_sorted, d_mm_rgb, d_y, n);
  PNRF_LAUNCH_CHECK();
  return 0;
}
extern "C" int pnrf_refine_head_fwd(const float* y, const float* rays, const float* depth_sorted, const float* jitter, int jitter_dir, float* z_pre,
                                    float* z, float* pts, float* rgb0, int64_t n, void* stream) {
  PNRF_REQUIRE(n >= 0 && (n == 0 || (y && rays && depth_sorted && z_pre && z && pts)), PNRF_E_ARG, "pnrf_refine_head_fwd: bad arguments");
  PNRF_REQUIRE(!jitter || jitter_dir == 1 || jitter_dir == -1, PNRF_E_ARG, "pnrf_refine_head_fwd: jitter_dir must be +1 or -1");
  if (n == 0) return 0;
  hipLaunchKernelGGL(refine_head_fwd_kernel, dim3(grid_for(n, 64)), dim3(64), 0, (hipStream_t)stream, y, rays, depth_sorted, jitter, jitter_dir, z_pre, z,
                     pts, rgb0, n);
  PNRF_LAUNCH_CHECK();
  return 0;
}
extern "C" int pnrf_refine_head_bwd(const float* y, const float* rays, const float* depth_sorted, const float* z_pre, const float* jitter,
                                    int jitter_dir, const float* d_pts, const float* d_z, const float* d_rgb0, float* d_y, float* d_depth_sorted,
                                    int64_t n, void* stream) {
  PNRF_REQUIRE(n >= 0 && (n == 0 || (y && rays && depth_sorted && z_pre && d_pts && d_y && d_depth_sorted)), PNRF_E_ARG, "pnrf_refine_head_bwd: bad arguments");
  PNRF_REQUIRE(!jitter || jitter_dir == 1 || jitter_dir == -1, PNRF_E_ARG, "pnrf_refine_head_bwd: jitter_dir must be +1 or -1");
  if (n == 0) return 0;
  hipLaunchKernelGGL(refine_head_bwd_kernel, dim3(grid_for(n, 64)), dim3(64), 0, (hipStream_t)stream, y, rays, depth_sorted, z_pre, jitter, jitter_dir,
                     d_pts, d_z, d_rgb0, d_y, d_depth_sorted, n);
  PNRF_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------ C ABI: trainer
static int trainer_init(pnrf_trainer* t, const float* const* W, const float* const* b, const int* in_dim, const int* out_dim, int64_t max_rays,
                        int max_samples);

extern "C" int pnrf_trainer_create(const float* const* W, const float* const* b, const int* in_dim, const int* out_dim, int n_layers,
                                   int64_t max_rays, int max_samples, pnrf_trainer_t** out) {
  PNRF_REQUIRE(W && b && in_dim && out_dim && out && max_rays >= 1, PNRF_E_ARG, "pnrf_trainer_create: null pointer / max_rays < 1");
  PNRF_REQUIRE(max_samples >= 8 && max_samples <= 256 && max_samples % 8 == 0, PNRF_E_ARG, "pnrf_trainer_create: max_samples must be a multiple of 8 in [8, 256]");
  PNRF_REQUIRE(n_layers == N_LAYERS, PNRF_E_ARG, "pnrf_trainer_create: expected %d Linear layers (7 sampler, 7 refine, 12 NeRF class), got %d", N_LAYERS, n_layers);
  static const int want_in[N_LAYERS] = {288, 256, 256, 256, 256, 256, 256, 144, 256, 256, 256, 256, 256, 256,
                                        63, 256, 256, 256, 256, 319, 256, 256, 256, 256, 283, 128};
  static const int want_out[N_LAYERS] = {256, 256, 256, 256, 256, 256, 27, 256, 256, 256, 256, 256, 256, 35,
                                         256, 256, 256, 256, 256, 256, 256, 256, 256, 1, 128, 3};
  for (int i = 0; i < N_LAYERS; ++i)
    PNRF_REQUIRE(in_dim[i] == want_in[i] && out_dim[i] == want_out[i], PNRF_E_ARG, "pnrf_trainer_create: layer %d is %dx%d, expected %dx%d", i, out_dim[i],
                 in_dim[i], want_out[i], want_in[i]);
  pnrf_trainer* t = new pnrf_trainer();
  const int rc = trainer_init(t, W, b, in_dim, out_dim, max_rays, max_samples);
  if (rc) {                          // an allocation failed part-way (e.g. out of memory at a large max_rays * max_samples): give everything back
    pnrf_trainer_free(t);
    return rc;
  }
  *out = t;
  return 0;
}

static int trainer_init(pnrf_trainer* t, const float* const* W, const float* const* b, const int* in_dim, const int* out_dim, int64_t max_rays,
                        int max_samples) {
  PNRF_HIP(hipGetDevice(&t->device));
  t->max_rays = max_rays;
  t->max_samples = max_samples;
  size_t off = 0;
  for (int i = 0; i < N_LAYERS; ++i) {
    TLin l; l.in = in_dim[i]; l.out = out_dim[i]; l.w = off; off += (size_t)l.in * l.out; l.b = off; off += l.out;
    off = (off + 3) & ~(size_t)3;
    t->L.push_back(l);
  }
  t->nparam = off;
  T_ALLOC(t->P, off); T_ALLOC(t->G, off); T_ALLOC(t->M, off); T_ALLOC(t->V, off); T_ALLOC(t->M2, off); T_ALLOC(t->V2, off);
  PNRF_HIP(hipMemset(t->P, 0, off * 4)); PNRF_HIP(hipMemset(t->G, 0, off * 4)); PNRF_HIP(hipMemset(t->M, 0, off * 4)); PNRF_HIP(hipMemset(t->V, 0, off * 4));
  PNRF_HIP(hipMemset(t->M2, 0, off * 4)); PNRF_HIP(hipMemset(t->V2, 0, off * 4));
  for (int i = 0; i < N_LAYERS; ++i) {
    PNRF_HIP(hipMemcpy(t->P + t->L[i].w, W[i], (size_t)t->L[i].in * t->L[i].out * 4, hipMemcpyDefault));
    PNRF_HIP(hipMemcpy(t->P + t->L[i].b, b[i], (size_t)t->L[i].out * 4, hipMemcpyDefault));
  }
  t->L[L_N + 5].gap = 63;                          // its input is cat[embedding 63 | 0 | h 256] (LD_C5, C5_H)
  {   // fp16 planes of the weights for the split-fp16 products: per layer [out64][in64] (forward) and [in64][out64] (backward), hi and lo each
    size_t halfs = 0;
    memset(&t->split, 0, sizeof(t->split));
    for (int i = 0; i < N_LAYERS; ++i) {
      const TLin& l = t->L[i];
      SplitLayer& sl = t->split.l[i];
      const int in64 = (l.in_x() + 63) / 64 * 64, out64 = (l.out + 63) / 64 * 64;
      sl.w = l.w; sl.in = l.in; sl.out = l.out; sl.gap = l.gap;
      sl.ld_fwd = in64; sl.ld_bwd = out64;
      sl.plane_fwd = sl.plane_bwd = (size_t)in64 * out64;
      sl.fwd = halfs; halfs += 2 * sl.plane_fwd;
      sl.bwd = halfs; halfs += 2 * sl.plane_bwd;
    }
    T_ALLOC(t->planes, halfs);
    PNRF_HIP(hipMemset(t->planes, 0, halfs * sizeof(_Float16)));
    const TLin& g = t->L[L_N + 5];
    T_ALLOC(t->w_gapped, (size_t)g.out * g.in_x());
    PNRF_HIP(hipMemset(t->w_gapped, 0, (size_t)g.out * g.in_x() * 4));
    t->split.n = N_LAYERS; t->split.P = t->P; t->split.planes = t->planes; t->split.total = t->nparam; t->split.gapped = t->w_gapped;
    params_changed(t);
    T_ALLOC(t->tc_stream, (size_t)TC_NSLOTS * SLOT_BYTES / sizeof(_Float16));
    memset(&t->tc_pack, 0, sizeof(t->tc_pack));
    t->tc_pack.P = t->P; t->tc_pack.stream = t->tc_stream;
    for (int l = 0; l < TC_NSL; ++l) {
      const TLin& tl = t->L[l < 8 ? L_N + l : (l == 8 ? L_FEAT : (l == 9 ? L_VIEWS : L_RGB))];
      PNRF_REQUIRE(tl.out == (l == 9 ? 128 : l == 10 ? 3 : 256) && tl.in == (l == 0 ? 63 : l == 5 ? 319 : l == 9 ? 283 : l == 10 ? 128 : 256), PNRF_E_SHAPE,
                   "pnrf_trainer: fine-net layer %d is %d -> %d", l, tl.in, tl.out);
      t->tc_pack.w[l] = tl.w; t->tc_pack.in_dim[l] = tl.in;
    }
    PNRF_REQUIRE(t->L[L_ALPHA].in == 256 && t->L[L_ALPHA].out == 1, PNRF_E_SHAPE, "pnrf_trainer: alpha_linear is %d -> %d", t->L[L_ALPHA].in, t->L[L_ALPHA].out);
    t->tc_pack.w_alpha = t->L[L_ALPHA].w;
    T_ALLOC(t->tb_stream, (size_t)TB_NSLOTS * SLOT_BYTES / sizeof(_Float16));
    memset(&t->tb_pack, 0, sizeof(t->tb_pack));
    t->tb_pack.P = t->P; t->tb_pack.stream = t->tb_stream;
    for (int l = 0; l < TC_NL; ++l) t->tb_pack.w[l] = t->tc_pack.w[l];
    t->tb_pack.w_alpha = t->L[L_ALPHA].w;
    T_ALLOC(t->cmax2, 32);                                     // two arrays of 16: tchain_norms_body
    PNRF_HIP(hipMemset(t->cmax2, 0, 32 * sizeof(float)));
    t->tb_pack.cmax = t->cmax2;
    T_ALLOC(t->amax, N_AMAX * HG_SLOT);
    PNRF_HIP(hipMemset(t->amax, 0, N_AMAX * HG_SLOT * 4));
  }
  const int64_t N = max_rays, R = (int64_t)max_samples * max_rays;
  T_ALLOC(t->mm_input, N * 288);
  for (int k = 0; k < 6; ++k) { T_ALLOC(t->s_h[k], N * 256); T_ALLOC(t->r_h[k], N * 256); }
  T_ALLOC(t->s_y, N * 27); T_ALLOC(t->depth_sorted, N * 8); T_ALLOC(t->add_s, N * 8); T_ALLOC(t->mul_s, N * 8); T_ALLOC(t->mm_rgb, N * 3);
  T_ALLOC(t->sort_idx, N * 8);
  T_ALLOC(t->refine_in, N * 144); T_ALLOC(t->r_y, N * 35); T_ALLOC(t->z_pre, N * 8); T_ALLOC(t->z, R); T_ALLOC(t->pts, R * 3); T_ALLOC(t->rgb0, N * 3);
  // the chain kernels (pnrf_tchain.h) write whole batches of TC_ROWS rows: their buffers hold the row count rounded up
  const int64_t Rp = (R + TC_ROWS - 1) / TC_ROWS * TC_ROWS;
  t->tc_ok = Rp * LD_C5 * 4 < ((int64_t)1 << 32);    // the chains address rows with 32-bit byte offsets; beyond that the per-layer products take over
  // the chain kernels' dynamic LDS exceeds the 64 KiB default limit: raised once per trainer (a per-device function attribute)
  {
    hipError_t ea = hipFuncSetAttribute((const void*)tchain_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TC_LDS_BYTES);
    if (ea == hipSuccess) ea = hipFuncSetAttribute((const void*)tchain_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TB_LDS_BYTES);
    if (ea != hipSuccess) { set_error("pnrf_trainer_create: hipFuncSetAttribute(max dynamic LDS) failed: %s", hipGetErrorString(ea)); pnrf_trainer_free(t); return (int)ea; }
  }
  for (int k = 0; k < 4; ++k) T_ALLOC(t->n_a[k], Rp * 256);
  T_ALLOC(t->n_c5, Rp * LD_C5); T_ALLOC(t->n_a5, Rp * 256); T_ALLOC(t->n_a6, Rp * 256); T_ALLOC(t->n_a7, Rp * 256);
  T_ALLOC(t->n_cv, Rp * LD_CV); T_ALLOC(t->n_hv, Rp * 128); T_ALLOC(t->raw, R * 4); T_ALLOC(t->rgb_map, N * 3); T_ALLOC(t->wts, R);
  T_ALLOC(t->d_rgb_map, N * 3); T_ALLOC(t->d_raw, R * 4); T_ALLOC(t->d_hv, R * 128); T_ALLOC(t->d_cv, Rp * LD_CV); T_ALLOC(t->d_a, Rp * 256);
  T_ALLOC(t->d_b, Rp * 256); T_ALLOC(t->d_c5, Rp * LD_C5); T_ALLOC(t->d_e0, Rp * 64); T_ALLOC(t->d_pts, N * 24); T_ALLOC(t->d_z, N * 8);
  for (int k = 0; k < 6; ++k) T_ALLOC(t->dz_x[k], Rp * 256);
  T_ALLOC(t->tc_mask, Rp / TC_ROWS * 8 * TC_NL * 64);
  T_ALLOC(t->d_add, N * 8); T_ALLOC(t->d_mul, N * 8); T_ALLOC(t->d_depth, N * 8); T_ALLOC(t->d_ry, N * 35); T_ALLOC(t->d_sy, N * 27);
  T_ALLOC(t->d_rgb0, N * 3); T_ALLOC(t->d_mmrgb, N * 3); for (int k = 0; k < 6; ++k) { T_ALLOC(t->d_hk[k], N * 256); T_ALLOC(t->d_hs[k], N * 256); }
  t->d_h0 = t->d_hk[0]; t->d_h1 = t->d_hk[1];
  // padding columns of the concatenated rows are zero and stay zero (the kernels write the payload columns only, or zeros)
  PNRF_HIP(hipMemset(t->n_c5, 0, (size_t)Rp * LD_C5 * 4)); PNRF_HIP(hipMemset(t->d_c5, 0, (size_t)Rp * LD_C5 * 4));
  PNRF_HIP(hipMemset(t->n_cv, 0, (size_t)Rp * LD_CV * 4)); PNRF_HIP(hipMemset(t->d_cv, 0, (size_t)Rp * LD_CV * 4));
  T_ALLOC(t->st_rays, N * 11); T_ALLOC(t->st_or_rays, N * 11); T_ALLOC(t->st_target, N * 3); T_ALLOC(t->st_ref_nos, N * 4); T_ALLOC(t->st_jitter, R); T_ALLOC(t->st_noise, R);
  T_ALLOC(t->loss, 8);                              // [total, mse x 3, completion counter of losses_kernel]
  PNRF_HIP(hipMemset(t->loss, 0, 32));
  {   // one slice of split-K partials per layer and iteration, each at its largest
    size_t cap = 0;
    for (int li = 0; li < N_LAYERS; ++li) {
      const TLin& l = t->L[li];
      const size_t sp = l.out <= HEAD_MAX ? HEAD_MAX_SPLITS : (l.out % 128 == 0 && l.in_x() % 128 == 0) ? DW128_MAX_SPLITS : DW_MAX_SPLITS;
      cap += sp * ((size_t)l.out * l.in_x() + l.out) + 4;
    }
    t->pool_cap = cap;
    T_ALLOC(t->dw_pool, cap);
  }
  PNRF_HIP(hipStreamCreateWithFlags(&t->own_stream, hipStreamNonBlocking));
  PNRF_HIP(hipEventCreateWithFlags(&t->ev_in, hipEventDisableTiming));
  PNRF_HIP(hipEventCreateWithFlags(&t->ev_out, hipEventDisableTiming));
  {   // pnrf_ray_encode_fwd allocates its table of ray points on first use: do that now, not inside a stream capture
    int rc = pnrf_ray_encode_fwd(t->st_rays, t->mm_input, 1, 48, nullptr);
    if (rc) return rc;
    PNRF_HIP(hipDeviceSynchronize());
  }
  return 0;
}

extern "C" int pnrf_trainer_free(pnrf_trainer_t* t) {
  if (!t) return 0;
  for (auto& g : t->graphs) (void)hipGraphExecDestroy(g.exec);
  if (t->own_stream) (void)hipStreamDestroy(t->own_stream);
  if (t->ev_in) (void)hipEventDestroy(t->ev_in);
  if (t->ev_out) (void)hipEventDestroy(t->ev_out);
  for (void* p : t->allocs) (void)hipFree(p);
  delete t;
  return 0;
}

// kind 0 parameters, 1 gradients, 2 / 3 Adam first / second moment (joint optimizer), 4 / 5 those of the NeRF-only optimizer;
// W / b: host or device destinations (either may be NULL)
extern "C" int pnrf_trainer_read(const pnrf_trainer_t* t, int kind, int layer, float* W, float* b, void* stream) {
  PNRF_REQUIRE(t && kind >= 0 && kind <= 5 && layer >= 0 && layer < N_LAYERS, PNRF_E_ARG, "pnrf_trainer_read: bad kind / layer");
  const float* base = kind == 0 ? t->P : kind == 1 ? t->G : kind == 2 ? t->M : kind == 3 ? t->V : kind == 4 ? t->M2 : t->V2;
  const TLin& l = t->L[layer];
  // on the caller's stream and complete on return (a plain hipMemcpy between device buffers runs on the null stream and may still be in
  // flight when it returns: a caller working on a non-blocking stream would race with it)
  if (W) PNRF_HIP(hipMemcpyAsync(W, base + l.w, (size_t)l.in * l.out * 4, hipMemcpyDefault, (hipStream_t)stream));
  if (b) PNRF_HIP(hipMemcpyAsync(b, base + l.b, (size_t)l.out * 4, hipMemcpyDefault, (hipStream_t)stream));
  PNRF_HIP(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}
extern "C" int pnrf_trainer_write(pnrf_trainer_t* t, int kind, int layer, const float* W, const float* b, void* stream) {
  PNRF_REQUIRE(t && kind >= 0 && kind <= 5 && layer >= 0 && layer < N_LAYERS, PNRF_E_ARG, "pnrf_trainer_write: bad kind / layer");
  float* base = kind == 0 ? t->P : kind == 1 ? t->G : kind == 2 ? t->M : kind == 3 ? t->V : kind == 4 ? t->M2 : t->V2;
  const TLin& l = t->L[layer];
  if (W) PNRF_HIP(hipMemcpyAsync(base + l.w, W, (size_t)l.in * l.out * 4, hipMemcpyDefault, (hipStream_t)stream));
  if (b) PNRF_HIP(hipMemcpyAsync(base + l.b, b, (size_t)l.out * 4, hipMemcpyDefault, (hipStream_t)stream));
  PNRF_HIP(hipStreamSynchronize((hipStream_t)stream));
  if (kind == 0) params_changed(t);
  return 0;
}
// Device address and element count of one of the flat arrays (kind as pnrf_trainer_read) — e.g. to all-reduce the gradients of
// data-parallel replicas in place with RCCL.  Layers are contiguous in trainer order, each [W (out x in), b (out)] padded to 4 floats.
extern "C" int pnrf_trainer_flat(pnrf_trainer_t* t, int kind, float** ptr, int64_t* count) {
  PNRF_REQUIRE(t && ptr && count && kind >= 0 && kind <= 5, PNRF_E_ARG, "pnrf_trainer_flat: bad arguments");
  *ptr = kind == 0 ? t->P : kind == 1 ? t->G : kind == 2 ? t->M : kind == 3 ? t->V : kind == 4 ? t->M2 : t->V2;
  *count = (int64_t)t->nparam;
  if (kind == 0) params_changed(t);        // the caller may write the parameters through this pointer (before the next iteration is submitted)
  return 0;
}

// captured iterations hold the kernel selection of the moment they were captured: every configuration change drops them
static void drop_graphs(pnrf_trainer* t) {
  for (auto& g : t->graphs) (void)hipGraphExecDestroy(g.exec);
  t->graphs.clear();
}

// Which weight-gradient kernel the square layers use: tile 0 = by shape and row count (128 x 128 tiles from min_rows_128 rows on, the default:
// 65 536), 64 / 128 = force that tile where the shape allows it.  A configuration step, like pnrf_mlp_set_variant: the library reads no
// environment.  (The two kernels differ in fp32 summation order only; tests/test_train_gpu.py runs the same batch through both.)
extern "C" int pnrf_trainer_set_dw_kernel(pnrf_trainer_t* t, int tile, int64_t min_rows_128) {
  PNRF_REQUIRE(t && (tile == 0 || tile == 64 || tile == 128 || tile == 255 || tile == 256) && min_rows_128 >= 0, PNRF_E_ARG,
               "pnrf_trainer_set_dw_kernel: tile must be 0, 64, 128 (or 256 / 255: the wide grouped tiles on / off)");
  drop_graphs(t);
  if (tile == 255 || tile == 256) {
    t->dw_wide_min_rows = tile == 255 ? INT64_MAX : (min_rows_128 > 0 ? min_rows_128 : 1);
    return 0;
  }
  t->dw_tile = tile;                        // (the wide-tile threshold is a separate choice: left as it is)
  t->dw128_min_rows = tile == 128 ? (min_rows_128 > 0 ? min_rows_128 : 256) : (min_rows_128 > 0 ? min_rows_128 : 65536);
  return 0;
}

// What the most recent iteration's grouped weight-gradient launch (dwh_group_kernel) held: number of gradients in it, and which of them ran on
// the 256 x 128 tiles (bit k = k-th job).  Host state only; tests assert with it that a forced tile shape was really the one that ran.
extern "C" int pnrf_trainer_dw_group_info(const pnrf_trainer_t* t, int* n_jobs, unsigned* wide_mask) {
  PNRF_REQUIRE(t && n_jobs && wide_mask, PNRF_E_ARG, "pnrf_trainer_dw_group_info: null argument");
  *n_jobs = t->grp_last_n; *wide_mask = t->grp_last_wide;
  return 0;
}

extern "C" int pnrf_trainer_set_step(pnrf_trainer_t* t, int64_t step, int64_t step_nerf) {
  PNRF_REQUIRE(t && step >= 0 && step_nerf >= 0, PNRF_E_ARG, "pnrf_trainer_set_step: bad arguments");
  t->step = step; t->step2 = step_nerf;
  return 0;
}

// which = 0: the joint optimizer over all 26 layers (stage 2 `optimizer`, stage 1 `s_optimizer`); which = 1: the NeRF-only
// optimizer of stage 1 (`optimizer`, base.py:398-421) with its own moments and step count.
extern "C" int pnrf_trainer_adam_step(pnrf_trainer_t* t, int which, float lr, float beta1, float beta2, float eps, float weight_decay, void* stream) {
  PNRF_REQUIRE(t && (which == 0 || which == 1) && lr >= 0.f && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f, PNRF_E_ARG,
               "pnrf_trainer_adam_step: bad arguments");
  int64_t& st = which == 0 ? t->step : t->step2;
  st += 1;
  const double bc1 = 1.0 - pow((double)beta1, (double)st), bc2 = 1.0 - pow((double)beta2, (double)st);
  const size_t first = which == 0 ? 0 : t->L[L_N].w;
  const int64_t count = (int64_t)(t->nparam - first);
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(count)), dim3(TPB), 0, (hipStream_t)stream, t->P + first, t->G + first, (which == 0 ? t->M : t->M2) + first,
                     (which == 0 ? t->V : t->V2) + first, count, lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2));
  PNRF_LAUNCH_CHECK();
  params_changed(t);
  return 0;
}

namespace {

#define T_RC(expr) do { int rc_ = (expr); if (rc_) return rc_; } while (0)

// rays -> sampler MLP -> head/sort -> projection -> refine MLP (activations saved in the trainer)
int sampler_refine_forward(pnrf_trainer* t, const pnrf_train_batch_t* bt, hipStream_t s) {
  void* stream = (void*)s;
  const int64_t N = bt->n;
  T_RC(pnrf_ray_encode_fwd(bt->rays, t->mm_input, N, 48, stream));                                                    // refine2.py:551-556
  {
    T_RC(elu_net_forward(t, L_S, t->mm_input, 288, t->s_h, N, s));
    T_RC(layer_fwd(t, L_S + 6, t->s_h[5], 256, t->s_y, 27, N, T_ACT_NONE, s));
  }
  T_RC(pnrf_sampler_head_fwd(t->s_y, bt->rays, t->depth_sorted, t->sort_idx, t->add_s, t->mul_s, t->mm_rgb, N, stream));     // :557-568
  T_RC(pnrf_refine_input_train_fwd(bt->rays, bt->or_rays, t->depth_sorted, bt->img4, bt->poses, bt->K, bt->ref_nos, bt->nv, 4, bt->Hf, bt->Wf, bt->eps,
                                   bt->layout, t->refine_in, N, stream));                                             // :570-634
  {
    T_RC(elu_net_forward(t, L_R, t->refine_in, 144, t->r_h, N, s));
    T_RC(layer_fwd(t, L_R + 6, t->r_h[5], 256, t->r_y, 35, N, T_ACT_NONE, s));
  }
  return 0;
}

// query points t->pts [N*S,3] + view directions -> raw [N*S,4]  (NeRF class, run_nerf_helpers.py:824-847)
int nerf_forward(pnrf_trainer* t, const pnrf_train_batch_t* bt, int S, hipStream_t s) {
  const int64_t R = bt->n * S;
  hipLaunchKernelGGL(nerf_inputs_kernel, dim3(grid_for(R * 32)), dim3(TPB), 0, s, t->pts, bt->rays + 8, 11, S, t->n_c5, LD_C5, t->n_cv, LD_CV, 256, R);
  PNRF_LAUNCH_CHECK();
  // Rows are independent through the layers, so the 256 -> 256 layers run as two layer chains (hgemm_wchain_kernel, pnrf_hgemm.h): a workgroup
  // keeps its 64 rows on chip from layer to layer — a chained layer writes its activation (the backward pass and the weight gradient need it)
  // and reads nothing but weights: 418 MB of activation traffic per forward pass at 32 768 rows instead of 670 MB.  pts0 and pts5 (63 / 320
  // inputs) stay products of their own; chain A = pts1 | pts2 | pts3 | pts4 (into the skip layer's input rows), chain B = pts6 | pts7 |
  // feature (no activation, into the view layer's input rows).  Same products in the same order as one launch per layer: bit-identical
  // results (tests/test_train_fullsize_gpu.py).  Measured (round 3, tools/train_iter.py): stage-2 iteration 1.254 vs 1.271 ms, exploration at
  // 64 samples 5.63 vs 5.68 ms — 1 % faster, not the 35 % the traffic suggests: the chain kernel keeps the MFMA pipes 31 % busy (the
  // product kernel 20 %), half of its wave time is spent in s_waitcnt / barriers (SQ_WAIT_ANY 49 %) — weight fragments one step ahead are
  // not far enough ahead of an L2 under this load, and the 16 x 64-byte stores of the register epilogue cost 23 % (no-store probe build:
  // 411 vs 535 us per launch).  A first version on hgemm_rchain_kernel<4> (LDS round trip in the epilogue, one workgroup per CU) was 2 % slower.
  bool chained = false, engine = false;
  if (engine_fwd(t, R)) {
    // pts0 .. pts7 and feature_linear in one launch on the fused-MLP engine: 128 rows per workgroup stay in registers through the nine layers
    TChainArgs c = {};
    c.blob = t->tc_stream;
    float* outs[TC_NSL] = {t->n_a[0], t->n_a[1], t->n_a[2], t->n_a[3], t->n_c5 + C5_H, t->n_a5, t->n_a6, t->n_a7, t->n_cv, t->n_hv, t->n_hv};
    for (int l = 0; l < TC_NSL; ++l) {
      c.bias[l] = t->P + t->L[l < 8 ? L_N + l : (l == 8 ? L_FEAT : (l == 9 ? L_VIEWS : L_RGB))].b;
      c.out[l] = outs[l]; c.ldo[l] = l == 4 ? LD_C5 : (l == 8 ? LD_CV : (l >= 9 ? 128 : 256));
    }
    c.bias_alpha = t->P + t->L[L_ALPHA].b; c.raw = t->raw;
    c.XV = t->n_cv + 256; c.ldxv = LD_CV;
    c.X0 = t->n_c5; c.ldx0 = LD_C5; c.mask = t->tc_mask; c.n = R; c.nbatch = (int)((R + TC_ROWS - 1) / TC_ROWS);
    const size_t lds = TC_LDS_BYTES;
    const int ncu = trainer_num_cu();
    hipLaunchKernelGGL(tchain_fwd_kernel, dim3((unsigned)(c.nbatch < ncu ? c.nbatch : ncu)), dim3(512), lds, s, c);
    PNRF_LAUNCH_CHECK();
    chained = true; engine = true;
  }
  if (!chained && t->use_f16 && t->nerf_fwd == 3 && R >= 8192) {
    // pts0 and pts5 (63 / 320 inputs) as products of their own; the 256 -> 256 layers behind each as a chain
    RChainArgs ca = {}, cb = {};
    HGemmArgs h0{}, h5{};
    bool ok = fwd_hgemm_args(t, L_N + 0, t->n_c5, LD_C5, t->n_a[0], 256, R, T_ACT_RELU, &h0) &&
              fwd_hgemm_args(t, L_N + 5, t->n_c5, LD_C5, t->n_a5, 256, R, T_ACT_RELU, &h5);
    float* outa[4] = {t->n_a[1], t->n_a[2], t->n_a[3], t->n_c5 + C5_H};
    const float* ina[4] = {t->n_a[0], t->n_a[1], t->n_a[2], t->n_a[3]};
    for (int k = 0; k < 4 && ok; ++k) {
      HGemmArgs a{};
      const int ldc = k == 3 ? LD_C5 : 256;
      ok = fwd_hgemm_args(t, L_N + 1 + k, ina[k], 256, outa[k], ldc, R, T_ACT_RELU, &a) && a.K == 256 && a.N == 256;
      ca.l[k] = RChainLayer{a.Bh, a.Bl, a.ldb, a.n_pad, a.bias, outa[k], nullptr, nullptr, T_ACT_RELU, ldc, 0};
    }
    float* outb[3] = {t->n_a6, t->n_a7, t->n_cv};
    const float* inb[3] = {t->n_a5, t->n_a6, t->n_a7};
    const int lib[3] = {L_N + 6, L_N + 7, L_FEAT};
    for (int k = 0; k < 3 && ok; ++k) {
      HGemmArgs a{};
      const int ldc = k == 2 ? LD_CV : 256, act = k == 2 ? T_ACT_NONE : T_ACT_RELU;
      ok = fwd_hgemm_args(t, lib[k], inb[k], 256, outb[k], ldc, R, act, &a) && a.K == 256 && a.N == 256;
      cb.l[k] = RChainLayer{a.Bh, a.Bl, a.ldb, a.n_pad, a.bias, outb[k], nullptr, nullptr, act, ldc, 0};
    }
    if (ok) {
      ca.has_first = 0; ca.X0 = t->n_a[0]; ca.n = 4; ca.M = R; ca.bwd = 0;
      cb.has_first = 0; cb.X0 = t->n_a5; cb.n = 3; cb.M = R; cb.bwd = 0;
      T_RC(launch_hgemm(h0, s));
      launch_wchain(ca, R, s);
      PNRF_LAUNCH_CHECK();
      T_RC(launch_hgemm(h5, s));
      launch_wchain(cb, R, s);
      PNRF_LAUNCH_CHECK();
      chained = true;
    }
  }
  if (!chained) {
    T_RC(layer_fwd(t, L_N + 0, t->n_c5, LD_C5, t->n_a[0], 256, R, T_ACT_RELU, s));
    for (int k = 1; k < 4; ++k) T_RC(layer_fwd(t, L_N + k, t->n_a[k - 1], 256, t->n_a[k], 256, R, T_ACT_RELU, s));
    T_RC(layer_fwd(t, L_N + 4, t->n_a[3], 256, t->n_c5 + C5_H, LD_C5, R, T_ACT_RELU, s));                                   // skip: cat[pts, h]
    T_RC(layer_fwd(t, L_N + 5, t->n_c5, LD_C5, t->n_a5, 256, R, T_ACT_RELU, s));
    T_RC(layer_fwd(t, L_N + 6, t->n_a5, 256, t->n_a6, 256, R, T_ACT_RELU, s));
    T_RC(layer_fwd(t, L_N + 7, t->n_a6, 256, t->n_a7, 256, R, T_ACT_RELU, s));
    T_RC(layer_fwd(t, L_FEAT, t->n_a7, 256, t->n_cv, LD_CV, R, T_ACT_NONE, s));
  }
  if (!engine) {                                     // (the engine launch ran alpha_linear, views_linear and rgb_linear too and wrote raw itself)
    T_RC(layer_fwd(t, L_ALPHA, t->n_a7, 256, t->raw + 3, 4, R, T_ACT_NONE, s));
    T_RC(layer_fwd(t, L_VIEWS, t->n_cv, LD_CV, t->n_hv, 128, R, T_ACT_RELU, s));
    T_RC(layer_fwd(t, L_RGB, t->n_hv, 128, t->raw, 4, R, T_ACT_NONE, s));
  }
  return 0;
}

// t->d_raw [R,4] -> gradients of the 12 NeRF layers; want_dpts: also d pts [R,3] into t->d_pts.  Every buffer handed to layer_bwd holds
// dL/dZ of its layer: the input-gradient product of the layer above applied the activation derivative in its epilogue.
int nerf_backward(pnrf_trainer* t, int64_t R, bool want_dpts, hipStream_t s) {
  const float* none = nullptr;
  constexpr int slot0 = 0;
  // rgb head: d_raw[:, 0:3] -> d_hv (x relu'(n_hv): the views layer) ; views layer -> d_cv = [d feature | d view embedding] (no activation) ;
  // feature -> d_a ; alpha: d_raw[:, 3] -> d_a += ..., then x relu'(n_a7)
  // one max-|gradient| slot per gradient buffer write (the two products that add up d_a share one)
  float* m = t->amax + slot0 * HG_SLOT;
  T_RC(layer_bwd(t, L_RGB, t->d_raw, 4, none, t->n_hv, 128, t->d_hv, 128, m + 0 * HG_SLOT, 0.f, R, T_ACT_RELU, t->n_hv, 128, 0, s));
  const bool engine = engine_bwd(t, R);
  if (!engine) T_RC(layer_bwd(t, L_VIEWS, t->d_hv, 128, m + 0 * HG_SLOT, t->n_cv, LD_CV, t->d_cv, LD_CV, m + 1 * HG_SLOT, 0.f, R, T_ACT_NONE, none, 0, 0, s));
  if (engine) {
    // The ten input-gradient products from the view layer down as ONE launch (tchain_bwd_kernel): the rows' gradients stay in registers from
    // the rgb branch's hidden gradient to pts0, every gradient a weight-gradient product needs is written once, the ReLU derivatives come from
    // the forward chain's masks.  The weight gradients follow as one grouped launch, each reading its gradient and the saved activation below
    // it once.
    float* dz[8] = {t->dz_x[5], t->dz_x[4], t->dz_x[3], t->dz_x[2], t->dz_x[1], t->dz_x[0], t->d_b, t->d_a};      // dz[k] = dZ_k
    const TLin& lf = t->L[L_FEAT];
    auto dw_job = [&](const float* X, int ldx, const float* dZ, int ldz, const float* amax_slot, const TLin& l) -> int {
      DwDefer d;
      d.allow_wide = true;
      // partials per gradient: at 32 768 rows 16 splits (the reduction of the partials is what shrinks: iteration 0.985 -> 0.960 ms), at 262 144 rows
      // 32 (16: 4.29 -> 4.50 ms — the grouped launch is bound by HBM there and wants the parallelism)
      d.wgs = R >= 65536 ? 128 : 64;
      int rc = gemm_dw(t, X, ldx, dZ, ldz, amax_slot, t->G + l.w, t->G + l.b, l.in_x(), l.gap, l.out, R, s, &d);
      if (rc || !d.set) return rc;                           // (not set: gemm_dw launched another kernel itself)
      return group_dw(t, d);
    };
    T_RC(dw_job(t->n_cv, LD_CV, t->d_hv, 128, m + 0 * HG_SLOT, t->L[L_VIEWS]));
    T_RC(dw_job(t->n_a7, 256, t->d_cv, LD_CV, m + 1 * HG_SLOT, lf));
    T_RC(layer_bwd(t, L_ALPHA, t->d_raw + 3, 4, none, t->n_a7, 256, nullptr, 0, nullptr, 0.f, R, T_ACT_NONE, none, 0, 0, s));          // weight gradient only
    TChainBwdArgs c = {};
    c.blob = t->tb_stream; c.dH = t->d_hv; c.lddh = 128; c.dF = t->d_cv; c.lddf = LD_CV; c.dA = t->d_raw + 3; c.ldda = 4; c.mask = t->tc_mask;
    c.cmax = t->tb_pack.cmax;
    for (int k = 0; k < 8; ++k) { c.dz[k] = dz[k]; c.slot[k] = m + (9 - k) * HG_SLOT; }
    c.slot[8] = m + 1 * HG_SLOT;                         // d feature (the rows of d_cv)
    c.slot[9] = m + 10 * HG_SLOT;                        // scratch
    c.dg = t->d_c5; c.lddg = LD_C5; c.de0 = t->d_e0; c.n = R; c.nbatch = (int)((R + TC_ROWS - 1) / TC_ROWS);
    const int ncu = trainer_num_cu();
    hipLaunchKernelGGL(tchain_bwd_kernel, dim3((unsigned)(c.nbatch < ncu ? c.nbatch : ncu)), dim3(512), TB_LDS_BYTES, s, c);
    PNRF_LAUNCH_CHECK();
    const float* xin[8] = {t->n_c5, t->n_a[0], t->n_a[1], t->n_a[2], t->n_a[3], t->n_c5, t->n_a5, t->n_a6};
    const int ldx[8] = {LD_C5, 256, 256, 256, 256, LD_C5, 256, 256};
    // the ten wide weight gradients join the iteration's grouped launch (flush_dw_group): at 32 768 rows each of them alone is a 30 us launch
    for (int k = 7; k >= 0; --k) T_RC(dw_job(xin[k], ldx[k], dz[k], 256, c.slot[k], t->L[L_N + k]));
    if (want_dpts) {
      hipLaunchKernelGGL(posenc_bwd_kernel, dim3(grid_for(R * 3)), dim3(TPB), 0, s, t->pts, t->d_e0, 64, t->d_c5, LD_C5, t->d_pts, R, 10);
      PNRF_LAUNCH_CHECK();
    }
    return 0;
  }
  T_RC(layer_bwd(t, L_FEAT, t->d_cv, LD_CV, m + 1 * HG_SLOT, t->n_a7, 256, t->d_a, 256, m + 2 * HG_SLOT, 0.f, R, T_ACT_NONE, none, 0, 0, s));
  T_RC(layer_bwd(t, L_ALPHA, t->d_raw + 3, 4, none, t->n_a7, 256, t->d_a, 256, m + 2 * HG_SLOT, 1.f, R, T_ACT_RELU, t->n_a7, 256, 0, s));
  T_RC(layer_bwd(t, L_N + 7, t->d_a, 256, m + 2 * HG_SLOT, t->n_a6, 256, t->d_b, 256, m + 3 * HG_SLOT, 0.f, R, T_ACT_RELU, t->n_a6, 256, 0, s));
  T_RC(layer_bwd(t, L_N + 6, t->d_b, 256, m + 3 * HG_SLOT, t->n_a5, 256, t->d_a, 256, m + 4 * HG_SLOT, 0.f, R, T_ACT_RELU, t->n_a5, 256, 0, s));
  // layer 5 reads cat[embedding(63), 0, h4(256)]: the activation derivative of layer 4 applies to the columns from 64 on
  if (want_dpts) {
    T_RC(layer_bwd(t, L_N + 5, t->d_a, 256, m + 4 * HG_SLOT, t->n_c5, LD_C5, t->d_c5, LD_C5, m + 5 * HG_SLOT, 0.f, R, T_ACT_RELU, t->n_c5 + C5_H, LD_C5, C5_H, s));
  } else {   // no position gradient: only the 256 hidden columns of the skip layer's input gradient are needed (one column block instead of two)
    T_RC(layer_bwd(t, L_N + 5, t->d_a, 256, m + 4 * HG_SLOT, t->n_c5, LD_C5, t->d_c5 + C5_H, LD_C5, m + 5 * HG_SLOT, 0.f, R, T_ACT_RELU, t->n_c5 + C5_H, LD_C5, 0, s,
                   C5_H));
  }
  T_RC(layer_bwd(t, L_N + 4, t->d_c5 + C5_H, LD_C5, m + 5 * HG_SLOT, t->n_a[3], 256, t->d_a, 256, m + 6 * HG_SLOT, 0.f, R, T_ACT_RELU, t->n_a[3], 256, 0, s));
  T_RC(layer_bwd(t, L_N + 3, t->d_a, 256, m + 6 * HG_SLOT, t->n_a[2], 256, t->d_b, 256, m + 7 * HG_SLOT, 0.f, R, T_ACT_RELU, t->n_a[2], 256, 0, s));
  T_RC(layer_bwd(t, L_N + 2, t->d_b, 256, m + 7 * HG_SLOT, t->n_a[1], 256, t->d_a, 256, m + 8 * HG_SLOT, 0.f, R, T_ACT_RELU, t->n_a[1], 256, 0, s));
  T_RC(layer_bwd(t, L_N + 1, t->d_a, 256, m + 8 * HG_SLOT, t->n_a[0], 256, t->d_b, 256, m + 9 * HG_SLOT, 0.f, R, T_ACT_RELU, t->n_a[0], 256, 0, s));
  T_RC(layer_bwd(t, L_N + 0, t->d_b, 256, m + 9 * HG_SLOT, t->n_c5, LD_C5, want_dpts ? t->d_e0 : nullptr, 64, nullptr, 0.f, R, T_ACT_NONE, none, 0, 0, s));   // 63 columns in rows of 64
  if (want_dpts) {
    hipLaunchKernelGGL(posenc_bwd_kernel, dim3(grid_for(R * 3)), dim3(TPB), 0, s, t->pts, t->d_e0, 64, t->d_c5, LD_C5, t->d_pts, R, 10);
    PNRF_LAUNCH_CHECK();
  }
  return 0;
}

// output-layer gradient dy [N, out_last] -> gradients of a 7-layer ELU net (sampler: first = L_S, refine: first = L_R); no gradient reaches the
// net's input (the Pluecker moment is depth-independent; the projection is under no_grad in the reference)
int elu_net_backward(pnrf_trainer* t, int first, const float* dy, int out_last, float* const* h, const float* x0, int in0, int64_t N, hipStream_t s) {
  float* m = t->amax + (first == L_S ? 16 : 24) * HG_SLOT;
  float* const* dh = first == L_S ? t->d_hs : t->d_hk;     // one set per net: the weight gradients read them in the iteration's grouped launch                // max-|gradient| slots of this net's six hidden gradients
  // dZ of hidden layer k lives in d_hk[k]; the output layer's product writes d_hk[5]
  T_RC(layer_bwd(t, first + 6, dy, out_last, nullptr, h[5], 256, dh[5], 256, m + 5 * HG_SLOT, 0.f, N, T_ACT_ELU, h[5], 256, 0, s));
  bool chain = chain_rows(N) && t->dw_tile == 0;
  for (int k = 5; k >= 1 && chain; --k)
    chain = hgemm_fits(t, 256, t->L[first + k].out, N, 256, dh[k - 1], 256, h[k - 1], 256, 0) && t->L[first + k].in == 256 && t->L[first + k].out == 256;
  if (chain) {
    // (1) the five input-gradient products in one launch, (2) the six weight gradients join the iteration's grouped launch
    RChainArgs c = {};
    for (int k = 5; k >= 1; --k) {
      HGemmArgs a{};
      bwd_hgemm_args(t, first + k, dh[k], 256, m + k * HG_SLOT, dh[k - 1], 256, m + (k - 1) * HG_SLOT, 0.f, N, T_ACT_ELU, h[k - 1], 256, 0, 0, &a);
      c.l[5 - k] = RChainLayer{a.Bh, a.Bl, a.ldb, a.n_pad, nullptr, dh[k - 1], h[k - 1], m + (k - 1) * HG_SLOT, T_ACT_ELU, 256, 256};
    }
    c.has_first = 0; c.X0 = dh[5]; c.x0_amax = m + 5 * HG_SLOT; c.n = 5; c.M = N; c.bwd = 1;
    launch_rchain(c, N, s);
    PNRF_LAUNCH_CHECK();
    for (int k = 5; k >= 0; --k) {
      const TLin& l = t->L[first + k];
      DwDefer d;
      d.wgs = 32;                                              // 4096 rows: 8 splits (stage-2 iteration 0.955 -> 0.942 ms against 16; 4 splits: 0.957)
      T_RC(gemm_dw(t, k ? h[k - 1] : x0, k ? 256 : in0, dh[k], 256, m + k * HG_SLOT, t->G + l.w, t->G + l.b, l.in_x(), l.gap, l.out, N, s, &d));
      PNRF_REQUIRE(d.set, PNRF_E_STATE, "pnrf_trainer: a hidden layer of an ELU net did not take the split-fp16 weight-gradient kernel");
      T_RC(group_dw(t, d));
    }
    return 0;
  }
  for (int k = 5; k >= 1; --k)
    T_RC(layer_bwd(t, first + k, dh[k], 256, m + k * HG_SLOT, h[k - 1], 256, dh[k - 1], 256, m + (k - 1) * HG_SLOT, 0.f, N, T_ACT_ELU, h[k - 1], 256, 0, s));
  return layer_bwd(t, first + 0, dh[0], 256, m, x0, in0, nullptr, 0, nullptr, 0.f, N, T_ACT_NONE, nullptr, 0, 0, s);
}

int check_batch(const pnrf_trainer* t, const pnrf_train_batch_t* bt, const float* loss, int S, const char* who) {
  PNRF_REQUIRE(t && bt && loss, PNRF_E_ARG, "%s: null pointer", who);
  PNRF_REQUIRE(bt->n >= 1 && bt->n <= t->max_rays, PNRF_E_ARG, "%s: n = %lld outside [1, max_rays = %lld]", who, (long long)bt->n, (long long)t->max_rays);
  PNRF_REQUIRE(S >= 8 && S <= t->max_samples && S % 8 == 0, PNRF_E_ARG, "%s: %d samples per ray outside [8, max_samples = %d] / not a multiple of 8", who, S, t->max_samples);
  PNRF_REQUIRE(bt->rays && bt->or_rays && bt->target && bt->img4 && bt->poses && bt->K && bt->ref_nos, PNRF_E_ARG, "%s: null batch pointer", who);
  PNRF_REQUIRE(!bt->jitter || bt->jitter_dir == 1 || bt->jitter_dir == -1, PNRF_E_ARG, "%s: jitter_dir must be +1 or -1", who);
  PNRF_REQUIRE(bt->layout == 0 || bt->layout == 1, PNRF_E_ARG, "%s: layout must be 0 (neighbour-major) or 1 (sample-major)", who);
  PNRF_REQUIRE(bt->clamp >= 0.f, PNRF_E_ARG, "%s: clamp must be >= 0", who);
  return 0;
}

}  // namespace

// Joint iteration: render_rays (training) + img2mse [+ a_mmrgb (img2mse(rgb_map0) + img2mse(mm_rgb))] + loss.backward() for all
// three networks.  Stage 2 (refine2.py:525-680, 858-868): layout 0, eps 1e-5, clamp 0, jitter + sigma noise.  Stage-1 even
// iterations (base.py:554-761 with train_sampler=True, :941-958): layout 1, eps 1e-6, clamp 10, no jitter / noise, a_mmrgb 1.
// Gradients of all 26 layers are left in the trainer (pnrf_trainer_read kind 1); loss dev [4] = {total, mse(rgb_map1),
// mse(rgb_map0), mse(mm_rgb)}; rgb_out dev [n,3] or NULL.
static int stage2_body(pnrf_trainer_t* t, const pnrf_train_batch_t* bt, hipStream_t s) {
  void* stream = (void*)s;
  const int64_t N = bt->n, R = 8 * bt->n;
  // ---------------- forward
  T_RC(sampler_refine_forward(t, bt, s));
  T_RC(pnrf_refine_head_fwd(t->r_y, bt->rays, t->depth_sorted, bt->jitter, bt->jitter_dir, t->z_pre, t->z, t->pts, t->rgb0, N, stream));   // :635-668
  T_RC(nerf_forward(t, bt, 8, s));
  T_RC(pnrf_composite_fwd(t->raw, t->z, bt->rays + 3, 11, t->add_s, t->mul_s, bt->raw_noise, bt->clamp, bt->white_bkgd, t->rgb_map, nullptr, nullptr,
                          t->wts, nullptr, N, 8, stream));                                                            // :674
  // ---------------- losses (:861-866)
  const bool aux = bt->a_mmrgb > 0.f;
  {
    LossArgs la = {{t->rgb_map, t->rgb0, t->mm_rgb}, {t->d_rgb_map, aux ? t->d_rgb0 : nullptr, aux ? t->d_mmrgb : nullptr}, {1.f, bt->a_mmrgb, bt->a_mmrgb},
                   bt->target, N * 3, t->loss, (unsigned*)(t->loss + 4), aux ? bt->a_mmrgb : 0.f};
    hipLaunchKernelGGL(losses_kernel, dim3(3), dim3(1024), 0, s, la);
  }
  PNRF_LAUNCH_CHECK();
  // ---------------- backward
  T_RC(pnrf_composite_bwd(t->raw, t->z, bt->rays + 3, 11, t->add_s, t->mul_s, bt->raw_noise, bt->clamp, bt->white_bkgd, t->d_rgb_map, t->d_raw, t->d_z,
                          t->d_add, t->d_mul, N, 8, stream));
  T_RC(nerf_backward(t, R, true, s));
  T_RC(pnrf_refine_head_bwd(t->r_y, bt->rays, t->depth_sorted, t->z_pre, bt->jitter, bt->jitter_dir, t->d_pts, t->d_z, aux ? t->d_rgb0 : nullptr, t->d_ry,
                            t->d_depth, N, stream));
  T_RC(elu_net_backward(t, L_R, t->d_ry, 35, t->r_h, t->refine_in, 144, N, s));
  T_RC(pnrf_sampler_head_bwd(t->s_y, bt->rays, t->sort_idx, t->d_depth, t->d_add, t->d_mul, aux ? t->d_mmrgb : nullptr, t->d_sy, N, stream));
  T_RC(elu_net_backward(t, L_S, t->d_sy, 27, t->s_h, t->mm_input, 288, N, s));
  return 0;
}

// Stage-1 odd iteration (base.py:554-761 with train_sampler=False, :929-940): sampler and refine nets run without gradient,
// the refined depths are explored into S = 8 n_mult samples (pnrf_explore_fwd: replicate toward dir1, sort, jitter toward
// jitter_dir with batch->jitter dev [n, S]), query points carry no learned offsets, compositing without add / mul and with
// batch->raw_noise dev [n, S]; loss = img2mse(rgb_map1); only the 12 NeRF layers get gradients (the others are left untouched).
static int explore_body(pnrf_trainer_t* t, const pnrf_train_batch_t* bt, int n_mult, int dir1, hipStream_t s) {
  void* stream = (void*)s;
  const int S = 8 * n_mult;
  const int64_t N = bt->n, R = (int64_t)S * bt->n;
  T_RC(sampler_refine_forward(t, bt, s));
  T_RC(pnrf_refine_head_fwd(t->r_y, bt->rays, t->depth_sorted, nullptr, 1, t->z_pre, t->z, t->pts, t->rgb0, N, stream));     // z_pre = refined depths
  T_RC(pnrf_explore_fwd(t->z_pre, bt->rays, bt->jitter, n_mult, dir1, bt->jitter_dir, t->z, t->pts, N, stream));            // base.py:689-729
  T_RC(nerf_forward(t, bt, S, s));
  T_RC(pnrf_composite_fwd(t->raw, t->z, bt->rays + 3, 11, nullptr, nullptr, bt->raw_noise, bt->clamp, bt->white_bkgd, t->rgb_map, nullptr, nullptr, t->wts,
                          nullptr, N, S, stream));
  {
    LossArgs la = {{t->rgb_map, t->rgb0, t->mm_rgb}, {t->d_rgb_map, nullptr, nullptr}, {1.f, 0.f, 0.f}, bt->target, N * 3, t->loss,
                   (unsigned*)(t->loss + 4), 0.f};
    hipLaunchKernelGGL(losses_kernel, dim3(3), dim3(1024), 0, s, la);
  }
  PNRF_LAUNCH_CHECK();
  T_RC(pnrf_composite_bwd(t->raw, t->z, bt->rays + 3, 11, nullptr, nullptr, bt->raw_noise, bt->clamp, bt->white_bkgd, t->d_rgb_map, t->d_raw, nullptr, nullptr,
                          nullptr, N, S, stream));
  T_RC(nerf_backward(t, R, false, s));
  return 0;
}
// Runs one iteration: the batch goes into the trainer's staging buffers (one launch), then the launch sequence of `body` — ~100 kernels whose
// pointer and scalar arguments are now all fixed for a given configuration — is replayed as a hipGraph (captured from this very stream the
// first time a configuration is seen), then loss / rgb are copied out.  Off by default (pnrf_trainer_set_graph): with ~100 kernels of 10-60 us
// each the iteration is bound by its kernels, not by the gaps between dependent launches — replayed it measured 5 % slower than launched
// kernel by kernel from a host that runs ahead of the GPU (2.98 vs 3.13 ms at 4096 rays, default stream, i.e. through the owned-stream hop).
template <class Body>
static int run_iteration(pnrf_trainer_t* t, const pnrf_train_batch_t* bt, int kind, int n_mult, int dir1, int S, float* loss, float* rgb_out,
                         hipStream_t caller, Body body_) {
  const int64_t N = bt->n;
  hipStream_t s = caller;
  const bool hop = t->use_graph && caller == nullptr;          // the legacy default stream cannot be captured
  if (hop) {
    s = t->own_stream;
    PNRF_HIP(hipEventRecord(t->ev_in, caller));
    PNRF_HIP(hipStreamWaitEvent(s, t->ev_in, 0));
  }
  {
    // which of the derived weight forms this iteration reads: the fp16 planes of the sampler / refine nets always; the fine net's planes
    // unless both of its chains run on the engine; the chains' fragment streams if the forward chain does.  One launch with the batch copy.
    const int64_t R = N * S;
    const bool eng_f = engine_fwd(t, R), eng = engine_bwd(t, R);
    static_assert(L_N + 12 == N_LAYERS && L_S < L_R && L_R < L_N, "the fine net's parameters are the tail of the flat parameter array (sp.total below)");
    static_assert(TPB == 256, "tchain_norms_body: 16 columns x 16 partial sums per block");
    PrepArgs pa = {};
    if (t->planes_stale || (!eng && t->nerf_planes_stale)) {
      pa.split = t->split;
      if (eng) pa.split.total = t->L[L_N].w;                   // the parameters before the fine net's
      pa.n_split = (unsigned)grid_for((int64_t)pa.split.total);
      t->planes_stale = false;
      if (!eng) t->nerf_planes_stale = false;
    }
    if (eng_f && t->streams_stale) {
      pa.cmax_zero = t->tb_pack.cmax;                          // the array in use until now: cleared for the refresh after this one
      t->tb_pack.cmax = t->cmax2 + (t->tb_pack.cmax == t->cmax2 ? 16 : 0);
      pa.pack = t->tc_pack; pa.packb = t->tb_pack;
      pa.n_pack = TC_NSLOTS * SLOT_FRAGS * 64 / TPB; pa.n_packb = TB_NSLOTS * SLOT_FRAGS * 64 / TPB; pa.n_norm = TB_NS * 16;
      t->streams_stale = false;
    }
    pa.stage = {bt->rays, bt->or_rays, bt->target, bt->jitter, bt->raw_noise, bt->ref_nos,
                t->st_rays, t->st_or_rays, t->st_target, t->st_jitter, t->st_noise, t->st_ref_nos, N, S, t->amax, N_AMAX * HG_SLOT};
    pa.n_stage = (unsigned)grid_for(N * 11);
    hipLaunchKernelGGL(iter_prepare_kernel, dim3(pa.n_stage + pa.n_norm + pa.n_pack + pa.n_packb + pa.n_split), dim3(TPB), 0, s, pa);
    PNRF_LAUNCH_CHECK();
  }
  pnrf_train_batch_t b = *bt;
  b.rays = t->st_rays; b.or_rays = t->st_or_rays; b.target = t->st_target; b.ref_nos = t->st_ref_nos;
  b.jitter = bt->jitter ? t->st_jitter : nullptr; b.raw_noise = bt->raw_noise ? t->st_noise : nullptr;
  auto body = [&](const pnrf_train_batch_t* bb, hipStream_t st) -> int {
    begin_dw(t);
    int rc = body_(bb, st);
    return rc ? rc : flush_dw_reduce(t, st);
  };
  if (!t->use_graph) {
    T_RC(body(&b, s));
  } else {
    pnrf_trainer::GraphKey key;
    memset(&key, 0, sizeof(key));
    key.kind = kind; key.n_mult = n_mult; key.dir1 = dir1; key.jitter_dir = b.jitter_dir; key.white_bkgd = b.white_bkgd; key.layout = b.layout;
    key.nv = b.nv; key.Hf = b.Hf; key.Wf = b.Wf; key.has_jitter = b.jitter != nullptr; key.has_noise = b.raw_noise != nullptr; key.n = N;
    key.eps = b.eps; key.a_mmrgb = b.a_mmrgb; key.clamp = b.clamp; key.img4 = b.img4; key.poses = b.poses; key.K = b.K; key.cmax = t->tb_pack.cmax; key.stream = s;
    hipGraphExec_t exec = nullptr;
    for (auto& g : t->graphs)
      if (memcmp(&g.key, &key, sizeof(key)) == 0) exec = g.exec;
    if (!exec) {
      hipGraph_t graph = nullptr;
      PNRF_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      const int rc = body(&b, s);
      const hipError_t e = hipStreamEndCapture(s, &graph);
      if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
      if (e != hipSuccess) { set_error("pnrf_trainer: stream capture of the iteration failed: %s", hipGetErrorString(e)); return (int)e; }
      const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
      (void)hipGraphDestroy(graph);
      if (ei != hipSuccess) { set_error("pnrf_trainer: hipGraphInstantiate failed: %s", hipGetErrorString(ei)); return (int)ei; }
      if (t->graphs.size() >= 8) { (void)hipGraphExecDestroy(t->graphs.front().exec); t->graphs.erase(t->graphs.begin()); }
      t->graphs.push_back({key, exec});
    }
    PNRF_HIP(hipGraphLaunch(exec, s));
  }
  PNRF_HIP(hipMemcpyAsync(loss, t->loss, 16, hipMemcpyDeviceToDevice, s));
  if (rgb_out) PNRF_HIP(hipMemcpyAsync(rgb_out, t->rgb_map, N * 3 * 4, hipMemcpyDeviceToDevice, s));
  if (hop) {
    PNRF_HIP(hipEventRecord(t->ev_out, s));
    PNRF_HIP(hipStreamWaitEvent(caller, t->ev_out, 0));
  }
  return 0;
}

extern "C" int pnrf_train_stage2_fwd_bwd(pnrf_trainer_t* t, const pnrf_train_batch_t* bt, float* loss, float* rgb_out, void* stream) {
  T_RC(check_batch(t, bt, loss, 8, "pnrf_train_stage2_fwd_bwd"));
  hipStream_t s = (hipStream_t)stream;
  return run_iteration(t, bt, 0, 1, 1, 8, loss, rgb_out, s, [&](const pnrf_train_batch_t* b, hipStream_t st) { return stage2_body(t, b, st); });
}

extern "C" int pnrf_train_explore_fwd_bwd(pnrf_trainer_t* t, const pnrf_train_batch_t* bt, int n_mult, int dir1, float* loss, float* rgb_out,
                                          void* stream) {
  PNRF_REQUIRE(n_mult >= 1 && n_mult <= 32 && (dir1 == 1 || dir1 == -1), PNRF_E_ARG, "pnrf_train_explore_fwd_bwd: n_mult 1..32, dir1 +-1");
  const int S = 8 * n_mult;
  T_RC(check_batch(t, bt, loss, S, "pnrf_train_explore_fwd_bwd"));
  PNRF_REQUIRE(bt->jitter, PNRF_E_ARG, "pnrf_train_explore_fwd_bwd: the exploration jitter [n, 8 n_mult] is required");
  hipStream_t s = (hipStream_t)stream;
  return run_iteration(t, bt, 1, n_mult, dir1, S, loss, rgb_out, s,
                       [&](const pnrf_train_batch_t* b, hipStream_t st) { return explore_body(t, b, n_mult, dir1, st); });
}

// How the layer products are computed.  0 (default): split-fp16 MFMA (pnrf_hgemm.h: fp32-grade, 22 significand bits per operand, fp32
// accumulation) wherever the shape fits; 1: exact-fp32 MFMA everywhere.
extern "C" int pnrf_trainer_set_products(pnrf_trainer_t* t, int kind) {
  PNRF_REQUIRE(t && kind >= 0 && kind <= 3, PNRF_E_ARG,
               "pnrf_trainer_set_products: kind 0 (split fp16), 1 (fp32), 2 (split fp16, one launch per layer) or 3 (split fp16, 64-row layer chains)");
  drop_graphs(t);
  t->use_f16 = kind != 1;
  t->nerf_fwd = kind == 1 ? 0 : kind;
  params_changed(t);                               // (also the fp32 kernels' gapped copy of the skip layer's weights is written with the planes)
  return 0;
}

// 1: replay every iteration as a hipGraph; 0 (default): launch its kernels one by one
extern "C" int pnrf_trainer_set_graph(pnrf_trainer_t* t, int enable) {
  PNRF_REQUIRE(t, PNRF_E_ARG, "pnrf_trainer_set_graph: null trainer");
  if (!enable) drop_graphs(t);
  t->use_graph = enable != 0;
  return 0;
}
#undef T_RC
