// pnrf_layout.h — compile-time layout of the three ProNeRF MLPs in the weight stream.
// Shared by the host packer (pnrf_pack.hip) and the kernels (pnrf_mlp_kernels.hip).
//
// Networks (fern_trt.txt:15,25-34; run_S_eS_eN_alter_trt.py:434-458):
//   SAMPLER  MinMaxRaySamplerTRT_Net     288 -> 6x256 (ELU) -> 27      fp32  (exact f32 MFMA)
//   REFINE   MinMaxRayEpiSamplerTRT_Net  144 -> 6x256 (ELU) -> 35      bf16
//   NERF     DoNeRFTRT(skip='auto')       63 -> 7x256 (ReLU) -> [256+27] -> 4   bf16
//   NERFCLS  NeRF(D=8,W=256,skips=[4],use_viewdirs)  63 -> 8x256 (skip at 5) -> {alpha, feature -> [256+27] -> 128 -> rgb}   bf16
#pragma once
#include "pnrf_engine.h"

namespace pnrf {

enum { NET_SAMPLER = 0, NET_REFINE = 1, NET_NERF = 2, NET_NERFCLS = 3 };
enum { PREC_F32 = 0, PREC_BF16 = 1, PREC_H16X2 = 2 };

constexpr int W_HID = 256;
constexpr int NT_HID = 8;                 // bf16 engine: 256 / 32 output tiles per hidden layer
constexpr int NT16_HID = 16;              // f32 engine: 256 / 16 output tiles per hidden layer

// ---- sampler (f32, v_mfma_f32_16x16x4_f32): k-step = 4 features (one per lane quarter)
constexpr int S_IN = 288, S_OUT = 27, S_NHID = 5;      // hidden 256->256 layers after layer 0
constexpr int S_KS0 = S_IN / 4;                          // 72 k-steps
constexpr int S_NPTS = S_IN / 6;                         // 48 ray points t = linspace(0, 1, 48), one Pluecker 6-vector each (trt.py:274-277, 556-557)
constexpr int S_KS4_0 = S_KS0 / 4;                       // 18 fragments per tile
constexpr int S_KS4_H = (W_HID / 4) / 4;                 // 16 fragments per tile (= one slot)
constexpr int S_NT_LAST = 2;                             // 27 outputs in two 16-row tiles
constexpr int S_SLOTS_L0 = layer_slots_f32<S_KS4_0, NT16_HID>();   // 18
constexpr int S_SLOTS_H = layer_slots_f32<S_KS4_H, NT16_HID>();    // 16
constexpr int S_SLOTS_LAST = layer_slots_f32<S_KS4_H, S_NT_LAST>();  // 2
constexpr int S_POS_H = S_SLOTS_L0 % NSLOTS;
constexpr int S_POS_LAST = (S_POS_H + S_NHID * S_SLOTS_H) % NSLOTS;
constexpr int S_SLOTS_USED = S_SLOTS_L0 + S_NHID * S_SLOTS_H + S_SLOTS_LAST;
constexpr int S_SLOTS_PAD = (NSLOTS - S_SLOTS_USED % NSLOTS) % NSLOTS;
constexpr int S_NSLOTS = S_SLOTS_USED + S_SLOTS_PAD;
constexpr int S_NBIAS = (1 + S_NHID) * W_HID + 16 * S_NT_LAST;      // packed bias floats
static_assert(S_SLOTS_H % NSLOTS == 0, "hidden layers must keep the ring position static");
// folded layer 0 (fused path only): the 48 Pluecker 6-vectors of a ray are the same vector in exact
// arithmetic (the moment (o+t d) x d^ does not depend on t), so W0[256x288] acts on them as
// Wf[256x6] = sum_p W0[:, 6p:6p+6].  K = 6 padded to 16 = 4 k-steps = 1 fragment per tile.
constexpr int SF_KS4_0 = 1;
constexpr int SF_SLOTS_L0 = layer_slots_f32<SF_KS4_0, NT16_HID>();   // 1
constexpr int SF_POS_H = SF_SLOTS_L0 % NSLOTS;
constexpr int SF_POS_LAST = (SF_POS_H + S_NHID * S_SLOTS_H) % NSLOTS;
constexpr int SF_SLOTS_USED = SF_SLOTS_L0 + S_NHID * S_SLOTS_H + S_SLOTS_LAST;
constexpr int SF_SLOTS_PAD = (NSLOTS - SF_SLOTS_USED % NSLOTS) % NSLOTS;
constexpr int SF_NSLOTS = SF_SLOTS_USED + SF_SLOTS_PAD;

// sampler in split-fp16 (layer_h16x2; fused path only, folded first layer): k-step = 32 features, tile pairs of 2x16 rows
constexpr int SH_KS_H = W_HID / 32;                       // 8
constexpr int SH_NTP_H = W_HID / 32;                      // 8 tile pairs
constexpr int SH_SLOTS_L0 = layer_slots_h16x2<1, SH_NTP_H>();         // 2
constexpr int SH_SLOTS_H = layer_slots_h16x2<SH_KS_H, SH_NTP_H>();    // 16
constexpr int SH_SLOTS_LAST = layer_slots_h16x2<SH_KS_H, 1>();        // 2
constexpr int SH_POS_H = SH_SLOTS_L0 % NSLOTS;
constexpr int SH_POS_LAST = (SH_POS_H + S_NHID * SH_SLOTS_H) % NSLOTS;
constexpr int SH_SLOTS_USED = SH_SLOTS_L0 + S_NHID * SH_SLOTS_H + SH_SLOTS_LAST;
constexpr int SH_SLOTS_PAD = (NSLOTS - SH_SLOTS_USED % NSLOTS) % NSLOTS;
constexpr int SH_NSLOTS = SH_SLOTS_USED + SH_SLOTS_PAD;
static_assert(SH_SLOTS_H % NSLOTS == 0, "hidden layers must keep the ring position static");

// sampler, pass 1 of the two-pass scheme (sampler_p1_kernel): plain fp16 on v_mfma_f32_32x32x16_f16 — the refine net's engine (32 columns
// per wave, 32-row tiles, 16-deep k-steps).  Layer 0 is the folded 6 -> 256 layer in split fp16 (fp32-grade: per tile one W_hi and one
// W_lo fragment, three MFMAs), hidden layers and the output layer are single fp16 products.
constexpr int P1_SLOTS_L0 = (NT_HID * 2 + SLOT_FRAGS - 1) / SLOT_FRAGS;               // 1
constexpr int P1_SLOTS_H = layer_slots_bf16<W_HID / 16, NT_HID>();                    // 8
constexpr int P1_SLOTS_LAST = layer_slots_bf16<W_HID / 16, 1>();                      // 1
constexpr int P1_POS_H = P1_SLOTS_L0 % NSLOTS;
constexpr int P1_POS_LAST = (P1_POS_H + S_NHID * P1_SLOTS_H) % NSLOTS;
constexpr int P1_SLOTS_USED = P1_SLOTS_L0 + S_NHID * P1_SLOTS_H + P1_SLOTS_LAST;
constexpr int P1_SLOTS_PAD = (NSLOTS - P1_SLOTS_USED % NSLOTS) % NSLOTS;
constexpr int P1_NSLOTS = P1_SLOTS_USED + P1_SLOTS_PAD;
constexpr int P1_NBIAS = (1 + S_NHID) * W_HID + 32;                                   // [tile][half][16], log2(e)-scaled for the ELU layers
static_assert(P1_SLOTS_H % NSLOTS == 0, "hidden layers must keep the ring position static");
// constants of the per-ray error model (two-pass sampler), computed from the weights at pack time
constexpr int P1_NCONST = 8;               // at least this many: [0] max_{k<8, j} W_out[k,j]^2 / log2(e)^2; [1 + l] C_l = max_j sum_i W_l[i,j]^2 of hidden layer l (p1_nconst)
// output tile of pass 1: half 0 holds the 8 depth logits (registers 0-7) and add (8-15), half 1 mul (0-7) and rgb (8-10)
__host__ __device__ constexpr int sampler_p1_out(int g, int h) {
  if (h == 0) return g;                    // depth 0..7, add 8..15
  return g < 8 ? 16 + g : (g < 11 ? 24 + (g - 8) : -1);
}

// ---- refine (bf16): k-step = 16 features
constexpr int R_IN = 144, R_OUT = 35, R_NHID = 5;
constexpr int R_KS0 = R_IN / 16;                         // 9
constexpr int KS_HID = W_HID / 16;                       // 16
constexpr int R_NT_LAST = 2;                             // tile 0: refine+offsets (32 rows), tile 1: rgb (3 rows)
constexpr int R_SLOTS_L0 = layer_slots_bf16<R_KS0, NT_HID>();    // 5
constexpr int SLOTS_HID = layer_slots_bf16<KS_HID, NT_HID>();    // 8
constexpr int R_SLOTS_LAST = layer_slots_bf16<KS_HID, R_NT_LAST>();  // 2
constexpr int R_POS_H = R_SLOTS_L0 % NSLOTS;
constexpr int R_POS_LAST = (R_POS_H + R_NHID * SLOTS_HID) % NSLOTS;
constexpr int R_SLOTS_USED = R_SLOTS_L0 + R_NHID * SLOTS_HID + R_SLOTS_LAST;
constexpr int R_SLOTS_PAD = (NSLOTS - R_SLOTS_USED % NSLOTS) % NSLOTS;
constexpr int R_NSLOTS = R_SLOTS_USED + R_SLOTS_PAD;
constexpr int R_NBIAS = (1 + R_NHID) * W_HID + 32 * R_NT_LAST;
static_assert(SLOTS_HID % NSLOTS == 0, "hidden layers must keep the ring position static");
// Free shape parameters of the reference's configs (run_S_eS_eN_alter_trt.py:62-82, 110-118, 427-457), round 6.  Width 256 and 8 samples per ray
// stay fixed; free are
//   mmnetdepth   sampler / refine: any number nhid = mmnetdepth - 1 >= 1 of hidden 256 -> 256 layers.  A hidden layer occupies a multiple of NSLOTS
//                ring slots in every stream, so the ring position of every later layer — a template constant of the kernels — does not depend on
//                nhid: the layer loop simply runs nhid times (netdepth of DoNeRFTRT likewise: nhid = netdepth - 2);
//   num_neighbor refine: 1 .. 8 neighbour views, refine input 48 + 24 nb.  A lane half holds NV = ceil(nb / 2) views: layer 0 has 3 NV + 3 k-steps
//                (RefineL0<NV>; NV = 2 is the Fern stream); views NV h + vv >= nb are padding with zero weights;
//   N_point_ray_enc  sampler: any number of ray points (input 6 P): the fused kernels run the FOLDED first layer Wf = sum_p W0[:, 6p:6p+6], K = 6
//                whatever P is; the unfolded stream (module-level forward, PNRF_VARIANT_SAMPLER_F32_FULL) exists for P = 48 only.
constexpr int MAX_NHID = 31;
constexpr int MAX_NB = 8;
template <int NV>
struct RefineL0 {
  static_assert(NV >= 1 && NV <= 4, "1 .. 8 neighbour views");
  static constexpr int KS0 = 3 * NV + 3;
  static constexpr int SLOTS_L0 = layer_slots_bf16<KS0, NT_HID>();
  static constexpr int POS_H = SLOTS_L0 % NSLOTS;
  static constexpr int POS_LAST = POS_H;                                          // nhid x SLOTS_HID is a multiple of NSLOTS
  static constexpr int SLOTS_PAD = (NSLOTS - (SLOTS_L0 + R_SLOTS_LAST) % NSLOTS) % NSLOTS;
};
static_assert(RefineL0<2>::KS0 == R_KS0 && RefineL0<2>::POS_H == R_POS_H && RefineL0<2>::POS_LAST == R_POS_LAST && RefineL0<2>::SLOTS_PAD == R_SLOTS_PAD, "NV = 2 is the Fern stream");
__host__ __device__ constexpr int refine_nv(int nb) { return (nb + 1) / 2; }
__host__ __device__ constexpr int refine_slots(int nhid, int nv) {
  const int used = (((3 * nv + 3) * NT_HID + SLOT_FRAGS - 1) / SLOT_FRAGS) + nhid * SLOTS_HID + R_SLOTS_LAST;
  return used + (NSLOTS - used % NSLOTS) % NSLOTS;
}
static_assert(refine_slots(R_NHID, 2) == R_NSLOTS, "refine_slots");
// stream sizes / bias counts as functions of the hidden-layer count (the constants above are the Fern values)
__host__ __device__ constexpr int pad_slots(int used) { return used + (NSLOTS - used % NSLOTS) % NSLOTS; }
__host__ __device__ constexpr int s_nslots(int nhid) { return pad_slots(S_SLOTS_L0 + nhid * S_SLOTS_H + S_SLOTS_LAST); }
__host__ __device__ constexpr int sf_nslots(int nhid) { return pad_slots(SF_SLOTS_L0 + nhid * S_SLOTS_H + S_SLOTS_LAST); }
__host__ __device__ constexpr int sh_nslots(int nhid) { return pad_slots(SH_SLOTS_L0 + nhid * SH_SLOTS_H + SH_SLOTS_LAST); }
__host__ __device__ constexpr int p1_slots_used(int nhid) { return P1_SLOTS_L0 + nhid * P1_SLOTS_H + P1_SLOTS_LAST; }
__host__ __device__ constexpr int p1_nslots(int nhid) { return pad_slots(p1_slots_used(nhid)); }
__host__ __device__ constexpr int p1_nconst(int nhid) { return 1 + nhid < P1_NCONST ? P1_NCONST : 1 + nhid; }      // [0] output layer, [1 + l] hidden layer l
__host__ __device__ constexpr int s_nbias(int nhid) { return (1 + nhid) * W_HID + 16 * S_NT_LAST; }
__host__ __device__ constexpr int p1_nbias(int nhid) { return (1 + nhid) * W_HID + 32; }
__host__ __device__ constexpr int r_nbias(int nhid) { return (1 + nhid) * W_HID + 32 * R_NT_LAST; }
static_assert(s_nslots(S_NHID) == S_NSLOTS && sf_nslots(S_NHID) == SF_NSLOTS && sh_nslots(S_NHID) == SH_NSLOTS && p1_nslots(S_NHID) == P1_NSLOTS &&
              s_nbias(S_NHID) == S_NBIAS && p1_nbias(S_NHID) == P1_NBIAS && r_nbias(R_NHID) == R_NBIAS, "Fern values");

// ---- refine on the 16x16x32 engine (layer_e16, fp16 operands; refine16_kernel, round 6): 32-deep k-steps, 16-row tiles in pairs, two 16-column blocks
// per wave.  Lane l: ray l & 15 of each block, group g = l >> 4.  A lane group holds NV4 = ceil(nb / 4) neighbour views x 8 samples x 3 colours, then
//   FOLD = false (refine_in rows from memory, any row): the Pluecker values of its two samples 2g, 2g + 1: 24 NV4 + 12 inputs in 3 NV4 + 2 k-steps of 8 slots
//                per group (the last 4 slots are padding);
//   FOLD = true  (the projecting head): the 8 Pluecker 6-vectors of a ray's samples are ONE vector in exact arithmetic — the moment (o + t d) x d^ = o x d^
//                does not depend on t (the sampler's folded first layer rests on the same identity, above) — so W0[:, 0:48] acts on them as
//                Wp[256 x 6] = sum_s W0[:, 6s:6s+6]: group 0 holds that vector in slots 0..5 of ONE more k-step: 3 NV4 + 1 k-steps (Fern: 4 instead of 5).
constexpr int R16_NTP_LAST = 2;                            // pair 0: refine + offsets (tile t, row 4g + r = sample 2g + t), pair 1: rgb (tile 2, rows 0..2)
constexpr int R16_SLOTS_LAST = (R16_NTP_LAST * (W_HID / 32) * 2 + SLOT_FRAGS - 1) / SLOT_FRAGS;      // 2
__host__ __device__ constexpr int refine16_nv(int nb) { return (nb + 3) / 4; }
__host__ __device__ constexpr int refine16_ks0(int nv4, bool fold) { return 3 * nv4 + (fold ? 1 : 2); }
template <int NV4, bool FOLD>
struct RefineE16 {
  static_assert(NV4 >= 1 && NV4 <= 2, "1 .. 8 neighbour views");
  static constexpr int KS0 = refine16_ks0(NV4, FOLD);
  static constexpr int SLOTS_L0 = (8 * KS0 * 2 + SLOT_FRAGS - 1) / SLOT_FRAGS;
  static constexpr int POS_H = SLOTS_L0 % NSLOTS;
  static constexpr int POS_LAST = POS_H;                                          // a hidden layer is 8 slots
  static constexpr int SLOTS_PAD = (NSLOTS - (SLOTS_L0 + R16_SLOTS_LAST) % NSLOTS) % NSLOTS;
};
__host__ __device__ constexpr int refine16_slots(int nhid, int nv4, bool fold) { return pad_slots(refine16_ks0(nv4, fold) + nhid * 8 + R16_SLOTS_LAST); }
__host__ __device__ constexpr int r16_nbias(int nhid) { return (1 + nhid) * W_HID + 16 * 2 * R16_NTP_LAST; }
// input that slot n = 8 ks + j of lane group g supplies; -1 = padding.  FOLD = false: index into refine_in = [pluecker(48: s*6+j), epi((k*8+s)*3+c)];
// FOLD = true: index into [pluecker(6), epi(24 nb)]
__host__ __device__ constexpr int refine16_in0(int nv4, int nb, int ks, int g, int j, bool fold) {
  const int n = 8 * ks + j;
  if (n < 24 * nv4) {
    const int view = nv4 * g + n / 24;
    return view < nb ? (fold ? 6 : 48) + ((view * 8 + (n % 24) / 3) * 3 + n % 3) : -1;
  }
  const int m = n - 24 * nv4;
  if (fold) return (g == 0 && m < 6) ? m : -1;
  return m < 12 ? (2 * g + m / 6) * 6 + m % 6 : -1;
}
// output row r16 = 4g + r of tile T -> network output [refine(8), offsets(24 = s*3+c), rgb(3)]
__host__ __device__ constexpr int refine16_out(int T, int r16) {
  const int g = r16 >> 2, r = r16 & 3;
  if (T < 2) { const int s = 2 * g + T; return r == 0 ? s : 8 + 3 * s + (r - 1); }
  return (T == 2 && r16 < 3) ? 32 + r16 : -1;
}

// ---- nerf (bf16)
constexpr int N_IN = 63, N_INV = 27, N_OUT = 4, N_NHID = 6;
constexpr int N_KS0 = 4;                                 // 63 padded to 64
constexpr int N_KSX = 2;                                 // 27 view features padded to 32
constexpr int N_KS_LAST = KS_HID + N_KSX;                // 18
constexpr int N_SLOTS_L0 = layer_slots_bf16<N_KS0, NT_HID>();    // 2
constexpr int N_SLOTS_LAST = layer_slots_bf16<N_KS_LAST, 1>();   // 2
constexpr int N_POS_H = N_SLOTS_L0 % NSLOTS;
constexpr int N_POS_LAST = (N_POS_H + N_NHID * SLOTS_HID) % NSLOTS;
constexpr int N_SLOTS_USED = N_SLOTS_L0 + N_NHID * SLOTS_HID + N_SLOTS_LAST;
constexpr int N_SLOTS_PAD = (NSLOTS - N_SLOTS_USED % NSLOTS) % NSLOTS;
constexpr int N_NSLOTS = N_SLOTS_USED + N_SLOTS_PAD;
constexpr int N_NBIAS = (1 + N_NHID) * W_HID + 32;
__host__ __device__ constexpr int n_nslots(int nhid) { return pad_slots(N_SLOTS_L0 + nhid * SLOTS_HID + N_SLOTS_LAST); }
__host__ __device__ constexpr int n_nbias(int nhid) { return (1 + nhid) * W_HID + 32; }
static_assert(n_nslots(N_NHID) == N_NSLOTS && n_nbias(N_NHID) == N_NBIAS, "Fern values");

// ---- nerf class (bf16): the `NeRF` module that stages 1/2 train (run_nerf_helpers.py:792-847).
// Engine layers: E0 pts0 (63->256) | E1-E4 pts1-4 | E5 pts5 ([63+256]->256) | E6,E7 pts6,7 |
// E89 views layer with feature_linear folded in + alpha as a 5th tile ([256+27] -> 128 ReLU + 1) | E10 rgb (128->3).
// feature_linear has no activation, so views(cat[feature(h), v]) = (Wv[:, :256] Wf) h + Wv[:, 256:] v + (bv + Wv[:, :256] bf): the packer
// multiplies the two weight matrices in fp64 once — one 256x256 layer (11 % of the fine net's MFMAs) less per sample, and one bf16
// rounding of an intermediate less.  alpha = Wa h + ba reads the same h and rides along as output row 128.
constexpr int C_NLIN = 12;                                // nn.Linear modules, pack order: pts0..7, feature, alpha, views, rgb
constexpr int C_KS5 = KS_HID + N_KS0;                     // 20: hidden k-steps first, then the 4 positional k-steps
constexpr int C_KS9 = KS_HID + N_KSX;                     // 18
constexpr int C_NT9 = 4;                                  // 128 view-layer outputs
constexpr int C_NT89 = C_NT9 + 1;                         // + the alpha tile
constexpr int C_KS10 = 8;                                 // 128 inputs
constexpr int C_SLOTS_E5 = layer_slots_bf16<C_KS5, NT_HID>();     // 10
constexpr int C_SLOTS_E89 = layer_slots_bf16<C_KS9, C_NT89>();    // 6
constexpr int C_SLOTS_E10 = layer_slots_bf16<C_KS10, 1>();        // 1
constexpr int C_POS_E1 = N_SLOTS_L0 % NSLOTS;
constexpr int C_POS_E5 = (C_POS_E1 + 4 * SLOTS_HID) % NSLOTS;
constexpr int C_POS_E6 = (C_POS_E5 + C_SLOTS_E5) % NSLOTS;
constexpr int C_POS_E89 = (C_POS_E6 + 2 * SLOTS_HID) % NSLOTS;
constexpr int C_POS_E10 = (C_POS_E89 + C_SLOTS_E89) % NSLOTS;
constexpr int C_SLOTS_USED = N_SLOTS_L0 + 4 * SLOTS_HID + C_SLOTS_E5 + 2 * SLOTS_HID + C_SLOTS_E89 + C_SLOTS_E10;
constexpr int C_SLOTS_PAD = (NSLOTS - C_SLOTS_USED % NSLOTS) % NSLOTS;
constexpr int C_NSLOTS = C_SLOTS_USED + C_SLOTS_PAD;
// packed bias offsets (floats): E0..E7 8x256, E89 5x32, E10 32
constexpr int C_BIAS_E89 = 8 * W_HID;
constexpr int C_BIAS_E10 = C_BIAS_E89 + 32 * C_NT89;
constexpr int C_NBIAS = C_BIAS_E10 + 32;

// ---- input-feature maps of layer 0 (and of the NeRF view k-steps); -1 = zero padding.
// sampler: k-step kk, quarter q  ->  mm_input feature (natural order)
__host__ __device__ constexpr int sampler_in0(int kk, int q) { return 4 * kk + q; }
// refine layer 0: the lane half h of a column (ray) supplies, in k-step ks, the values n = 8 ks + j of ITS 72 of the 144 inputs —
//   n <  48: colour channel n % 3 of sample (n % 24) / 3 in neighbour view 2h + n / 24   = refine_in[48 + ((2h + n/24) 8 + (n%24)/3) 3 + n%3]
//   n >= 48: Pluecker value (n - 48) % 6 of sample 4h + (n - 48) / 6                      = refine_in[(4h + (n-48)/6) 6 + (n-48)%6]
// (refine_in = [pluecker(48: s*6+j), epi(96: (k*8+s)*3+c)], run_S_eS_eN_alter_trt.py:653-661).  A lane that projects "its" two views and
// encodes "its" four samples therefore holds exactly its own B fragments: the projection runs in the head of the refine kernel with no
// exchange between lanes and no refine_in round trip through HBM (refine_kernel, HEAD = 1).
__host__ __device__ constexpr int refine_in0(int ks, int h, int j) {
  const int n = 8 * ks + j;
  return n < 48 ? 48 + (((2 * h + n / 24) * 8 + (n % 24) / 3) * 3 + n % 3) : (4 * h + (n - 48) / 6) * 6 + (n - 48) % 6;
}
// ... for nv views per lane half and nb views in all: value n = 8 ks + j of half h is colour (n % 24) of view nv h + n / 24 (padding, -1, when that
// view does not exist) for n < 24 nv, else Pluecker value n - 24 nv of the half's four samples.  refine_in0_nv(2, 4, ..) = refine_in0(..).
__host__ __device__ constexpr int refine_in0_nv(int nv, int nb, int ks, int h, int j) {
  const int n = 8 * ks + j;
  if (n < 24 * nv) {
    const int view = nv * h + n / 24;
    return view < nb ? 48 + ((view * 8 + (n % 24) / 3) * 3 + n % 3) : -1;
  }
  return (4 * h + (n - 24 * nv) / 6) * 6 + (n - 24 * nv) % 6;
}
// nerf layer 0: slot n = ks*8+j.  n<30: (freq k=n/3, coord c=n%3), half 0 = sin, half 1 = cos;
// n=30: x0|x2, n=31: x1|pad.  Feature order of the embedder: [x, sin f0 x, cos f0 x, ...]
// (run_nerf_helpers.py:666-671).
__host__ __device__ constexpr int nerf_in0(int ks, int h, int j) {
  const int n = ks * 8 + j;
  if (n < 30) return 3 + 6 * (n / 3) + 3 * h + (n % 3);
  if (n == 30) return h ? 2 : 0;
  return h ? -1 : 1;
}
// nerf view k-steps (e = 0,1): index into the 27-wide view embedding (4 frequencies).
__host__ __device__ constexpr int nerf_inx(int e, int h, int j) {
  const int n = e * 8 + j;
  if (n < 12) return 3 + 6 * (n / 3) + 3 * h + (n % 3);
  if (n == 12) return h ? 2 : 0;
  if (n == 13) return h ? -1 : 1;
  return -1;
}

// ---- output-row maps of the last layers: tile row -> network output index (-1 = unused).
// sampler (two 16-row tiles tt = 0,1; lane quarter q; reg r): quarter 0 holds depth[4tt+r], quarter 1
// add[4tt+r], quarter 2 mul[4tt+r], quarter 3 rgb[r] (tile 0 only) — each quarter ends up with all 8
// values of "its" quantity for one ray.  Output order of the net: [depth(8), add(8), mul(8), rgb(3)]
// (run_nerf_helpers.py:1502-1505).
__host__ __device__ constexpr int sampler_out(int tt, int q, int r) {
  if (q < 3) return 8 * q + 4 * tt + r;
  return (tt == 0 && r < 3) ? 24 + r : -1;
}
// refine tile 0: half h, reg g=4a+b -> sample s=4h+a; b=0 refine logit, b=1..3 offset xyz.
// Net output order: [refine(8), offsets(24 = s*3+c), rgb(3)] (run_nerf_helpers.py:1536-1538).
__host__ __device__ constexpr int refine_out0(int g, int h) {
  const int s = 4 * h + (g >> 2), b = g & 3;
  return b == 0 ? s : 8 + 3 * s + (b - 1);
}
__host__ __device__ constexpr int refine_out1(int g, int h) { return (h == 0 && g < 3) ? 32 + g : -1; }
__host__ __device__ constexpr int nerf_out(int g, int h) { return (h == 0 && g < 4) ? g : -1; }

// ---- NeRF (DoNeRFTRT) on the 16x16x32 engine (layer_b16): 32-deep k-steps, 16-row output tiles handled in pairs, two
// 16-column blocks per wave.  Lane l: column l&15, group g = l>>4.  Same slot structure as the 32x32x16 stream.
constexpr int NB_KS_H = W_HID / 32;                         // 8 k-steps per hidden layer
constexpr int NB_NTP_H = W_HID / 32;                        // 8 tile pairs per hidden layer
constexpr int NB_KS0 = 2;                                   // 63 positional features padded to 64
constexpr int NB_KS_LAST = NB_KS_H + 1;                     // hidden k-steps + one k-step of view features (27 padded to 32)
constexpr int NB_SLOTS_L0 = (NB_NTP_H * NB_KS0 * 2 + SLOT_FRAGS - 1) / SLOT_FRAGS;      // 2
constexpr int NB_SLOTS_H = (NB_NTP_H * NB_KS_H * 2 + SLOT_FRAGS - 1) / SLOT_FRAGS;      // 8
constexpr int NB_SLOTS_LAST = (1 * NB_KS_LAST * 2 + SLOT_FRAGS - 1) / SLOT_FRAGS;       // 2
constexpr int NB_POS_H = NB_SLOTS_L0 % NSLOTS;
constexpr int NB_POS_LAST = (NB_POS_H + N_NHID * NB_SLOTS_H) % NSLOTS;
constexpr int NB_SLOTS_USED = NB_SLOTS_L0 + N_NHID * NB_SLOTS_H + NB_SLOTS_LAST;
constexpr int NB_SLOTS_PAD = (NSLOTS - NB_SLOTS_USED % NSLOTS) % NSLOTS;
constexpr int NB_NSLOTS = NB_SLOTS_USED + NB_SLOTS_PAD;
static_assert(NB_SLOTS_H % NSLOTS == 0, "hidden layers must keep the ring position");
__host__ __device__ constexpr int nb_nslots(int nhid) { return pad_slots(NB_SLOTS_L0 + nhid * NB_SLOTS_H + NB_SLOTS_LAST); }
__host__ __device__ constexpr int nb_nbias(int nhid) { return ((1 + nhid) * (W_HID / 16) + 2) * 16; }
static_assert(nb_nslots(N_NHID) == NB_NSLOTS, "Fern values");
// NeRF class on the same engine: E0 | E1-E4 | E5 ([h, pos] -> 256: 8 + 2 k-steps) | E6, E7 | E89 ([h, views] -> 128 ReLU + alpha: 9
// k-steps, 8 view tiles + the alpha tile = 5 tile pairs) | E10 (128 -> 3: 4 k-steps, 1 pair)
constexpr int CB_KS5 = NB_KS_H + NB_KS0;                                                // 10
constexpr int CB_NTP89 = 5;
constexpr int CB_KS10 = (W_HID / 2) / 32;                                               // 4
constexpr int CB_SLOTS_E5 = (NB_NTP_H * CB_KS5 * 2 + SLOT_FRAGS - 1) / SLOT_FRAGS;      // 10
constexpr int CB_SLOTS_E89 = (CB_NTP89 * NB_KS_LAST * 2 + SLOT_FRAGS - 1) / SLOT_FRAGS; // 6
constexpr int CB_SLOTS_E10 = (1 * CB_KS10 * 2 + SLOT_FRAGS - 1) / SLOT_FRAGS;           // 1
constexpr int CB_POS_E1 = NB_SLOTS_L0 % NSLOTS;
constexpr int CB_POS_E5 = (CB_POS_E1 + 4 * NB_SLOTS_H) % NSLOTS;
constexpr int CB_POS_E6 = (CB_POS_E5 + CB_SLOTS_E5) % NSLOTS;
constexpr int CB_POS_E89 = (CB_POS_E6 + 2 * NB_SLOTS_H) % NSLOTS;
constexpr int CB_POS_E10 = (CB_POS_E89 + CB_SLOTS_E89) % NSLOTS;
constexpr int CB_SLOTS_USED = NB_SLOTS_L0 + 4 * NB_SLOTS_H + CB_SLOTS_E5 + 2 * NB_SLOTS_H + CB_SLOTS_E89 + CB_SLOTS_E10;
constexpr int CB_SLOTS_PAD = (NSLOTS - CB_SLOTS_USED % NSLOTS) % NSLOTS;
constexpr int CB_NSLOTS = CB_SLOTS_USED + CB_SLOTS_PAD;
constexpr int CB_BIAS_E89 = 8 * W_HID;                                                  // [tile of 16 rows][16]
constexpr int CB_BIAS_E10 = CB_BIAS_E89 + 16 * 2 * CB_NTP89;
constexpr int CB_NBIAS = CB_BIAS_E10 + 32;
// Input feature (index into the reference's 63-wide position embedding [x, sin 2^0 x, cos 2^0 x, ...]: 3 + 6 k + 3 fn + c) that
// element j of lane group g supplies in k-step ks of layer 0.  The 30 (component, octave) combinations are cut into 15 chains of two
// consecutive octaves; a lane group evaluates four chains q = 4g .. 4g+3 (component q % 3, octaves 2 (q/3), 2 (q/3) + 1) and chain
// j4 fills slots 4 j4 .. 4 j4 + 3 = [sin k0, cos k0, sin k0+1, cos k0+1].  The 16th chain position (g = 3, slots 12..15) carries
// the raw x, y, z and one padding slot instead.
__host__ __device__ constexpr int nerf16_in0(int ks, int g, int j) {
  const int idx = 8 * ks + j, q = 4 * g + idx / 4, e = idx % 4;
  if (q == 15) return e < 3 ? e : -1;
  return 3 + 6 * (2 * (q / 3) + (e >> 1)) + 3 * (e & 1) + q % 3;
}
// Same for the 27-wide view embedding in its single k-step: group g evaluates octave g of the three components, slots 2c, 2c+1 =
// sin, cos of component c; slots 6 / 7: raw vx / vy for g = 0, raw vz / padding for g = 1, padding for g = 2, 3.
__host__ __device__ constexpr int nerf16_inx(int g, int j) {
  if (j == 6) return g == 0 ? 0 : (g == 1 ? 2 : -1);
  if (j == 7) return g == 0 ? 1 : -1;
  return 3 + 6 * g + 3 * (j & 1) + j / 2;
}

}  // namespace pnrf
