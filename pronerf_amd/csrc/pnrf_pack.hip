// pnrf_pack.hip — host side: error state, weight packing into the MFMA weight stream.
#include <math.h>
#include <stdarg.h>
#include <string.h>

#include "pnrf_common.h"

namespace pnrf {

static thread_local std::string g_err = "";

void set_error(const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
}

// fp32 -> bf16, round to nearest even; NaN stays NaN.
static inline uint16_t f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
  return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

struct Layer {
  const float* W;
  const float* b;
  int in_dim, out_dim;
  int nt;                    // output tiles (bf16: 32 rows; f32: 16 rows)
  int nk;                    // k-steps (bf16: 16 features each; f32: 4 features each, multiple of 4)
  std::vector<int> in_map;   // bf16: [nk][2][8]; f32: [nk][4]   -> input feature or -1
  std::vector<int> out_map;  // bf16: [nt][32], f32: [nt][16]  tile row -> output index or -1
  double wscale = 1.0;       // weights / biases are stored multiplied by these (in double, before the rounding to the stream's type):
  double bscale = 1.0;       // the log2(e) scaling of the ELU nets' streams, see elu_scaled() in pnrf_engine.h
};
// ELU nets on log2(e)-scaled activations: first layer W, every ELU layer's bias x log2(e); output layer W / log2(e)
static void scale_for_elu(std::vector<Layer>& Ls) {
  for (size_t l = 0; l + 1 < Ls.size(); ++l) Ls[l].bscale = LOG2E_D;
  Ls.front().wscale = LOG2E_D;
  Ls.back().wscale = 1.0 / LOG2E_D;
}

static size_t layer_frags(const Layer& L, int prec) { return (size_t)L.nt * (prec == PREC_BF16 ? L.nk : L.nk / 4); }
static size_t layer_slots(const Layer& L, int prec) { return (layer_frags(L, prec) + SLOT_FRAGS - 1) / SLOT_FRAGS; }

static inline float wval(const Layer& L, int out, int in) { return (out >= 0 && in >= 0) ? (float)((double)L.W[(size_t)out * L.in_dim + in] * L.wscale) : 0.f; }

static void pack_layer(const Layer& L, int prec, char* dst) {
  if (prec == PREC_BF16) {
    for (int to = 0; to < L.nt; ++to)
      for (int ks = 0; ks < L.nk; ++ks) {
        uint16_t* frag = (uint16_t*)(dst + ((size_t)to * L.nk + ks) * FRAG_BYTES);
        for (int lane = 0; lane < 64; ++lane) {
          const int r = lane & 31, h = lane >> 5;
          const int out = L.out_map[to * 32 + r];
          for (int j = 0; j < 8; ++j) frag[lane * 8 + j] = f2bf(wval(L, out, L.in_map[(ks * 2 + h) * 8 + j]));
        }
      }
  } else {      // 16x16x4: lane = (row i = lane&15, quarter q = lane>>4); a fragment = 4 consecutive k-steps
    const int ks4 = L.nk / 4;
    for (int to = 0; to < L.nt; ++to)
      for (int fr = 0; fr < ks4; ++fr) {
        float* frag = (float*)(dst + ((size_t)to * ks4 + fr) * FRAG_BYTES);
        for (int lane = 0; lane < 64; ++lane) {
          const int r = lane & 15, q = lane >> 4;
          const int out = L.out_map[to * 16 + r];
          for (int i = 0; i < 4; ++i) frag[lane * 4 + i] = wval(L, out, L.in_map[(4 * fr + i) * 4 + q]);
        }
      }
  }
}

// split-fp16 packing (layer_h16x2): in_map [nk][4][8], out_map [nt][16], fragments (tp, ks, tile-in-pair, plane)
static inline uint16_t f2h(float f) { _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }
static inline float h2f(uint16_t u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }
static void pack_layer_h16x2(const Layer& L, char* dst) {
  const int ntp = L.nt / 2;
  for (int tp = 0; tp < ntp; ++tp)
    for (int ks = 0; ks < L.nk; ++ks)
      for (int t = 0; t < 2; ++t) {
        uint16_t* hi = (uint16_t*)(dst + ((((size_t)tp * L.nk + ks) * 2 + t) * 2 + 0) * FRAG_BYTES);
        uint16_t* lo = (uint16_t*)(dst + ((((size_t)tp * L.nk + ks) * 2 + t) * 2 + 1) * FRAG_BYTES);
        for (int lane = 0; lane < 64; ++lane) {
          const int r = lane & 15, g = lane >> 4;
          const int out = L.out_map[(2 * tp + t) * 16 + r];
          for (int j = 0; j < 8; ++j) {
            const int in = L.in_map[(ks * 4 + g) * 8 + j];
            const double w = (out >= 0 && in >= 0) ? (double)L.W[(size_t)out * L.in_dim + in] * L.wscale : 0.0;     // split the scaled value itself
            const uint16_t h = f2h((float)w);
            hi[lane * 8 + j] = h;
            lo[lane * 8 + j] = f2h((float)((w - (double)h2f(h)) * H16_LO_SCALE));
          }
        }
      }
}

// plain fp16 packing for the 32x32x16 engine (pass 1 of the two-pass sampler): as pack_layer's bf16 branch with fp16 rounding
static void pack_layer_f16(const Layer& L, char* dst) {
  for (int to = 0; to < L.nt; ++to)
    for (int ks = 0; ks < L.nk; ++ks) {
      uint16_t* frag = (uint16_t*)(dst + ((size_t)to * L.nk + ks) * FRAG_BYTES);
      for (int lane = 0; lane < 64; ++lane) {
        const int r = lane & 31, h = lane >> 5;
        const int out = L.out_map[to * 32 + r];
        for (int j = 0; j < 8; ++j) frag[lane * 8 + j] = f2h(wval(L, out, L.in_map[(ks * 2 + h) * 8 + j]));
      }
    }
}
// ... and its split first layer (one k-step): per tile a W_hi fragment and a W_lo * 2^11 fragment
static void pack_layer0_f16x2(const Layer& L, char* dst) {
  for (int to = 0; to < L.nt; ++to) {
    uint16_t* hi = (uint16_t*)(dst + ((size_t)to * 2 + 0) * FRAG_BYTES);
    uint16_t* lo = (uint16_t*)(dst + ((size_t)to * 2 + 1) * FRAG_BYTES);
    for (int lane = 0; lane < 64; ++lane) {
      const int r = lane & 31, h = lane >> 5;
      const int out = L.out_map[to * 32 + r];
      for (int j = 0; j < 8; ++j) {
        const int in = L.in_map[h * 8 + j];
        const double w = (out >= 0 && in >= 0) ? (double)L.W[(size_t)out * L.in_dim + in] * L.wscale : 0.0;
        const uint16_t hh = f2h((float)w);
        hi[lane * 8 + j] = hh;
        lo[lane * 8 + j] = f2h((float)((w - (double)h2f(hh)) * H16_LO_SCALE));
      }
    }
  }
}

// bf16 packing for layer_b16 (16x16x32): in_map [nk][4][8], out_map [nt][16], fragments (tp, ks, tile-in-pair)
static void pack_layer_b16(const Layer& L, char* dst, bool f16 = false) {
  const int ntp = L.nt / 2;
  for (int tp = 0; tp < ntp; ++tp)
    for (int ks = 0; ks < L.nk; ++ks)
      for (int t = 0; t < 2; ++t) {
        uint16_t* frag = (uint16_t*)(dst + (((size_t)tp * L.nk + ks) * 2 + t) * FRAG_BYTES);
        for (int lane = 0; lane < 64; ++lane) {
          const int r = lane & 15, g = lane >> 4;
          const int out = L.out_map[(2 * tp + t) * 16 + r];
          for (int j = 0; j < 8; ++j) {
            const float w = wval(L, out, L.in_map[(ks * 4 + g) * 8 + j]);
            frag[lane * 8 + j] = f16 ? f2h(w) : f2bf(w);
          }
        }
      }
}

static void pack_bias(const Layer& L, int prec, float* dst) {
  if (prec == PREC_F32 || prec == PREC_H16X2) {      // [tile][16 rows] in tile-row order: lane quarter q reads rows 4q..4q+3
    for (int i = 0; i < L.nt * 16; ++i) dst[i] = L.out_map[i] >= 0 ? (float)((double)L.b[L.out_map[i]] * L.bscale) : 0.f;
    return;
  }
  for (int to = 0; to < L.nt; ++to)
    for (int h = 0; h < 2; ++h)
      for (int g = 0; g < 16; ++g) {
        const int out = L.out_map[to * 32 + acc_row(g, h)];
        dst[(to * 2 + h) * 16 + g] = out >= 0 ? (float)((double)L.b[out] * L.bscale) : 0.f;
      }
}

static std::vector<int> identity_out(int rows) {
  std::vector<int> m(rows);
  for (int i = 0; i < rows; ++i) m[i] = i;
  return m;
}
static std::vector<int> hidden_in(int prec) {
  std::vector<int> m;
  if (prec == PREC_BF16) {
    m.resize(KS_HID * 16);
    for (int ks = 0; ks < KS_HID; ++ks)
      for (int h = 0; h < 2; ++h)
        for (int j = 0; j < 8; ++j) m[(ks * 2 + h) * 8 + j] = hidden_feat_bf16(ks, h, j);
  } else {
    m.resize((W_HID / 4) * 4);
    for (int kk = 0; kk < W_HID / 4; ++kk)
      for (int q = 0; q < 4; ++q) m[kk * 4 + q] = hidden_feat_f32(kk, q);
  }
  return m;
}

}  // namespace pnrf

using namespace pnrf;

extern "C" int pnrf_abi_version(void) { return PNRF_ABI_VERSION; }
extern "C" const char* pnrf_last_error(void) { return pnrf::g_err.c_str(); }

extern "C" int pnrf_linspace(float start, float end, int n, float* out) {
  PNRF_REQUIRE(out && n >= 1, PNRF_E_ARG, "pnrf_linspace: bad arguments");
  // torch.linspace (CPU, float): step = (end-start)/(n-1); i < n/2 ? start + step*i : end - step*(n-1-i)
  if (n == 1) { out[0] = start; return 0; }
  const float step = (end - start) / (float)(n - 1);
  const int half = n / 2;
  // torch's vectorised kernel evaluates start + step*i / end - step*(n-1-i) with one rounding (FMA)
  for (int i = 0; i < n; ++i) out[i] = i < half ? fmaf(step, (float)i, start) : fmaf(-step, (float)(n - 1 - i), end);
  return 0;
}

// NeRF-class fine net: 12 Linear modules -> 11 engine layers (feature + alpha share one layer).
static int upload(pnrf_mlp* h, const std::vector<char>& blob, const std::vector<float>& bias, const std::vector<int>& in0,
                  const std::vector<int>& inx, const std::vector<int>& outm);

static int pack_nerfcls(const float* const* W, const float* const* b, const int* in_dim, const int* out_dim, int n_layers, pnrf_mlp_t** out) {
  PNRF_REQUIRE(n_layers == C_NLIN, PNRF_E_SHAPE, "pnrf_mlp_pack: the NeRF class expects %d Linear layers (pts0..7, feature, alpha, views, rgb), got %d", C_NLIN, n_layers);
  const int ei[C_NLIN] = {N_IN, W_HID, W_HID, W_HID, W_HID, W_HID + N_IN, W_HID, W_HID, W_HID, W_HID, W_HID + N_INV, W_HID / 2};
  const int eo[C_NLIN] = {W_HID, W_HID, W_HID, W_HID, W_HID, W_HID, W_HID, W_HID, W_HID, 1, W_HID / 2, 3};
  for (int l = 0; l < C_NLIN; ++l) {
    PNRF_REQUIRE(W[l] && b[l], PNRF_E_ARG, "pnrf_mlp_pack: null weight/bias at layer %d", l);
    PNRF_REQUIRE(in_dim[l] == ei[l] && out_dim[l] == eo[l], PNRF_E_SHAPE, "pnrf_mlp_pack: NeRF-class layer %d is %dx%d, kernels are built for %dx%d",
                 l, out_dim[l], in_dim[l], eo[l], ei[l]);
  }
  const std::vector<int> hid = hidden_in(PREC_BF16);
  std::vector<Layer> Ls(10);
  auto base = [&](Layer& L, int lin) {
    L.W = W[lin]; L.b = b[lin]; L.in_dim = in_dim[lin]; L.out_dim = out_dim[lin];
    L.nt = NT_HID; L.nk = KS_HID; L.in_map = hid; L.out_map = identity_out(W_HID);
  };
  for (int e = 0; e < 8; ++e) base(Ls[e], e);
  // E0: positional k-steps (same slot order as the DoNeRFTRT first layer)
  Ls[0].nk = N_KS0; Ls[0].in_map.assign(N_KS0 * 16, -1);
  for (int ks = 0; ks < N_KS0; ++ks) for (int h = 0; h < 2; ++h) for (int j = 0; j < 8; ++j) Ls[0].in_map[(ks * 2 + h) * 8 + j] = nerf_in0(ks, h, j);
  // E5: input = cat[pts(63), h(256)] (helpers:838-839): 16 hidden k-steps on columns 63.., then the 4 positional k-steps on columns 0..62
  Ls[5].nk = C_KS5; Ls[5].in_map.assign(C_KS5 * 16, -1);
  for (int ks = 0; ks < KS_HID; ++ks) for (int h = 0; h < 2; ++h) for (int j = 0; j < 8; ++j) Ls[5].in_map[(ks * 2 + h) * 8 + j] = N_IN + hidden_feat_bf16(ks, h, j);
  for (int ks = 0; ks < N_KS0; ++ks) for (int h = 0; h < 2; ++h) for (int j = 0; j < 8; ++j) Ls[5].in_map[((KS_HID + ks) * 2 + h) * 8 + j] = nerf_in0(ks, h, j);
  // E89: views layer with feature_linear folded in (helpers:842-848; see pnrf_layout.h) + alpha as row 128.
  //   rows 0..127: Wc[r, :256] = sum_k Wv[r, k] Wf[k, :]  (fp64), Wc[r, 256:283] = Wv[r, 256:283], bc[r] = bv[r] + sum_k Wv[r, k] bf[k]
  //   row 128    : Wc[128, :256] = Wa, no view inputs, bc[128] = ba
  const int VI = W_HID + N_INV, VO = W_HID / 2;
  std::vector<float> Wc((size_t)(VO + 1) * VI, 0.f), bc(VO + 1, 0.f);
  {
    const float *Wf = W[8], *bf = b[8], *Wa = W[9], *Wv = W[10], *bv = b[10];
    std::vector<double> acc(W_HID);
    for (int r = 0; r < VO; ++r) {
      for (int c = 0; c < W_HID; ++c) acc[c] = 0.0;
      double ab = (double)bv[r];
      for (int k = 0; k < W_HID; ++k) {
        const double wv = (double)Wv[(size_t)r * VI + k];
        const float* row = Wf + (size_t)k * W_HID;
        for (int c = 0; c < W_HID; ++c) acc[c] += wv * (double)row[c];
        ab += wv * (double)bf[k];
      }
      for (int c = 0; c < W_HID; ++c) Wc[(size_t)r * VI + c] = (float)acc[c];
      for (int c = 0; c < N_INV; ++c) Wc[(size_t)r * VI + W_HID + c] = Wv[(size_t)r * VI + W_HID + c];
      bc[r] = (float)ab;
    }
    for (int c = 0; c < W_HID; ++c) Wc[(size_t)VO * VI + c] = Wa[c];
    bc[VO] = b[9][0];
  }
  Layer& E89 = Ls[8];
  E89.W = Wc.data(); E89.b = bc.data(); E89.in_dim = VI; E89.out_dim = VO + 1; E89.nt = C_NT89; E89.nk = C_KS9;
  E89.in_map.assign(C_KS9 * 16, -1);
  std::vector<int> inx_map(N_KSX * 16, -1);
  for (int ks = 0; ks < KS_HID; ++ks) for (int h = 0; h < 2; ++h) for (int j = 0; j < 8; ++j) E89.in_map[(ks * 2 + h) * 8 + j] = hidden_feat_bf16(ks, h, j);
  for (int e = 0; e < N_KSX; ++e) for (int h = 0; h < 2; ++h) for (int j = 0; j < 8; ++j) {
    const int v = nerf_inx(e, h, j);
    inx_map[(e * 2 + h) * 8 + j] = v;
    E89.in_map[((KS_HID + e) * 2 + h) * 8 + j] = v >= 0 ? W_HID + v : -1;
  }
  E89.out_map.assign(32 * C_NT89, -1);
  for (int i = 0; i <= VO; ++i) E89.out_map[i] = i;     // alpha = output 128 -> tile 4 row 0 = half 0, register 0
  // E10: rgb (helpers:850)
  Layer& E10 = Ls[9];
  E10.W = W[11]; E10.b = b[11]; E10.in_dim = W_HID / 2; E10.out_dim = 3; E10.nt = 1; E10.nk = C_KS10;
  E10.in_map.assign(C_KS10 * 16, -1);
  for (int ks = 0; ks < C_KS10; ++ks) for (int h = 0; h < 2; ++h) for (int j = 0; j < 8; ++j) E10.in_map[(ks * 2 + h) * 8 + j] = hidden_feat_bf16(ks, h, j);
  E10.out_map.assign(32, -1);
  for (int g = 0; g < 3; ++g) E10.out_map[acc_row(g, 0)] = g;

  size_t slots = 0, nbias = 0;
  for (auto& L : Ls) { slots += layer_slots(L, PREC_BF16); nbias += (size_t)L.nt * 32; }
  slots += (NSLOTS - slots % NSLOTS) % NSLOTS;
  PNRF_REQUIRE(slots == (size_t)C_NSLOTS && nbias == (size_t)C_NBIAS, PNRF_E_SHAPE, "pnrf_mlp_pack: internal NeRF-class layout mismatch (%zu slots, %zu bias floats)", slots, nbias);
  std::vector<char> blob(slots * SLOT_BYTES, 0);
  std::vector<float> bias(nbias, 0.f);
  size_t so = 0, bo = 0;
  for (auto& L : Ls) {
    pack_layer(L, PREC_BF16, blob.data() + so * SLOT_BYTES);
    pack_bias(L, PREC_BF16, bias.data() + bo);
    so += layer_slots(L, PREC_BF16); bo += (size_t)L.nt * 32;
  }
  // second stream for the 16x16x32 engine (nerf16_kernel<true>): same ten engine layers, 16-row tiles in pairs, 32-deep k-steps
  std::vector<Layer> Lb = Ls;
  std::vector<int> hid16(NB_KS_H * 32);
  for (int ks = 0; ks < NB_KS_H; ++ks) for (int g = 0; g < 4; ++g) for (int j = 0; j < 8; ++j) hid16[(ks * 4 + g) * 8 + j] = hidden_feat_h16(ks, g, j);
  for (int e = 0; e < 8; ++e) { Lb[e].nt = W_HID / 16; Lb[e].out_map = identity_out(W_HID); Lb[e].nk = NB_KS_H; Lb[e].in_map = hid16; }
  Lb[0].nk = NB_KS0; Lb[0].in_map.assign(NB_KS0 * 32, -1);
  for (int ks = 0; ks < NB_KS0; ++ks) for (int g = 0; g < 4; ++g) for (int j = 0; j < 8; ++j) Lb[0].in_map[(ks * 4 + g) * 8 + j] = nerf16_in0(ks, g, j);
  Lb[5].nk = CB_KS5; Lb[5].in_map.assign(CB_KS5 * 32, -1);
  for (int ks = 0; ks < NB_KS_H; ++ks) for (int g = 0; g < 4; ++g) for (int j = 0; j < 8; ++j) Lb[5].in_map[(ks * 4 + g) * 8 + j] = N_IN + hidden_feat_h16(ks, g, j);
  for (int ks = 0; ks < NB_KS0; ++ks) for (int g = 0; g < 4; ++g) for (int j = 0; j < 8; ++j) Lb[5].in_map[((NB_KS_H + ks) * 4 + g) * 8 + j] = nerf16_in0(ks, g, j);
  Lb[8].nt = 2 * CB_NTP89; Lb[8].nk = NB_KS_LAST; Lb[8].in_map.assign(NB_KS_LAST * 32, -1);
  for (int ks = 0; ks < NB_KS_H; ++ks) for (int g = 0; g < 4; ++g) for (int j = 0; j < 8; ++j) Lb[8].in_map[(ks * 4 + g) * 8 + j] = hidden_feat_h16(ks, g, j);
  for (int g = 0; g < 4; ++g) for (int j = 0; j < 8; ++j) {
    const int v = nerf16_inx(g, j);
    Lb[8].in_map[(NB_KS_H * 4 + g) * 8 + j] = v >= 0 ? W_HID + v : -1;
  }
  Lb[8].out_map.assign(16 * 2 * CB_NTP89, -1);
  for (int i = 0; i <= VO; ++i) Lb[8].out_map[i] = i;                      // rows 0..127 = view layer, row 128 (tile 8, row 0) = alpha
  Lb[9].nt = 2; Lb[9].nk = CB_KS10; Lb[9].in_map.assign(CB_KS10 * 32, -1);
  for (int ks = 0; ks < CB_KS10; ++ks) for (int g = 0; g < 4; ++g) for (int j = 0; j < 8; ++j) Lb[9].in_map[(ks * 4 + g) * 8 + j] = hidden_feat_h16(ks, g, j);
  Lb[9].out_map.assign(32, -1);
  for (int r = 0; r < 3; ++r) Lb[9].out_map[r] = r;
  auto bs = [&](const Layer& L) { return ((size_t)(L.nt / 2) * L.nk * 2 + SLOT_FRAGS - 1) / SLOT_FRAGS; };
  size_t slots_b16 = 0, nb16 = 0;
  for (auto& L : Lb) { slots_b16 += bs(L); nb16 += (size_t)L.nt * 16; }
  slots_b16 += (NSLOTS - slots_b16 % NSLOTS) % NSLOTS;
  PNRF_REQUIRE(slots_b16 == (size_t)CB_NSLOTS && nb16 == (size_t)CB_NBIAS, PNRF_E_SHAPE, "pnrf_mlp_pack: internal NeRF-class b16 layout mismatch (%zu slots, %zu bias floats)",
               slots_b16, nb16);
  std::vector<char> blob_b16(slots_b16 * SLOT_BYTES, 0), blob_f16(slots_b16 * SLOT_BYTES, 0);      // the same stream with bf16 / fp16 operands
  std::vector<float> bias_b16(nb16, 0.f);
  {
    size_t sb = 0, bb = 0;
    for (auto& L : Lb) {
      pack_layer_b16(L, blob_b16.data() + sb * SLOT_BYTES);
      pack_layer_b16(L, blob_f16.data() + sb * SLOT_BYTES, true);
      pack_bias(L, PREC_F32, bias_b16.data() + bb);
      sb += bs(L); bb += (size_t)L.nt * 16;
    }
  }

  std::vector<int> outm(64, -1);                  // module-level store map: (half, reg) -> output index, rgb only (alpha handled by the kernel)
  pnrf_mlp* h = new pnrf_mlp();
  memset(h, 0, sizeof(*h));
  h->net = PNRF_NET_NERFCLS; h->prec = PREC_BF16; h->in_dim = N_IN; h->in_dim_x = N_INV; h->out_dim = 4;
  h->nslots = (uint32_t)slots; h->nbias = (int)nbias;
  int rc = upload(h, blob, bias, Ls[0].in_map, inx_map, outm);
  if (rc == 0) {
    h->nslots_b16 = (uint32_t)slots_b16; h->nbias_b16 = (int)nb16;
    hipError_t e = hipMalloc(&h->d_blob_b16, blob_b16.size());
    if (e == hipSuccess) e = hipMemcpy(h->d_blob_b16, blob_b16.data(), blob_b16.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void**)&h->d_bias_b16, nb16 * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(h->d_bias_b16, bias_b16.data(), nb16 * sizeof(float), hipMemcpyHostToDevice);
    h->nslots_f16 = (uint32_t)slots_b16;
    if (e == hipSuccess) e = hipMalloc(&h->d_blob_f16, blob_f16.size());
    if (e == hipSuccess) e = hipMemcpy(h->d_blob_f16, blob_f16.data(), blob_f16.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) { set_error("pnrf_mlp_pack: device allocation/copy failed: %s", hipGetErrorString(e)); rc = (int)e; }
  }
  if (rc) { pnrf_mlp_free(h); return rc; }
  *out = h;
  return 0;
}

static int upload(pnrf_mlp* h, const std::vector<char>& blob, const std::vector<float>& bias, const std::vector<int>& in0,
                  const std::vector<int>& inx, const std::vector<int>& outm) {
  h->n_in0 = (int)in0.size(); h->n_inx = (int)inx.size(); h->n_out = (int)outm.size();
  hipError_t e = hipGetDevice(&h->device);
  if (e == hipSuccess) e = hipMalloc(&h->d_blob, blob.size());
  if (e == hipSuccess) e = hipMemcpy(h->d_blob, blob.data(), blob.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc((void**)&h->d_bias, bias.size() * sizeof(float));
  if (e == hipSuccess) e = hipMemcpy(h->d_bias, bias.data(), bias.size() * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc((void**)&h->d_in0, in0.size() * sizeof(int));
  if (e == hipSuccess) e = hipMemcpy(h->d_in0, in0.data(), in0.size() * sizeof(int), hipMemcpyHostToDevice);
  if (e == hipSuccess && !inx.empty()) {
    e = hipMalloc((void**)&h->d_inx, inx.size() * sizeof(int));
    if (e == hipSuccess) e = hipMemcpy(h->d_inx, inx.data(), inx.size() * sizeof(int), hipMemcpyHostToDevice);
  }
  if (e == hipSuccess) e = hipMalloc((void**)&h->d_out, outm.size() * sizeof(int));
  if (e == hipSuccess) e = hipMemcpy(h->d_out, outm.data(), outm.size() * sizeof(int), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    set_error("pnrf_mlp_pack: device allocation/copy failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

extern "C" int pnrf_mlp_pack(int net, const float* const* W, const float* const* b, const int* in_dim,
                             const int* out_dim, int n_layers, pnrf_mlp_t** out) {
  PNRF_REQUIRE(W && b && in_dim && out_dim && out, PNRF_E_ARG, "pnrf_mlp_pack: null argument");
  if (net == PNRF_NET_NERFCLS) return pack_nerfcls(W, b, in_dim, out_dim, n_layers, out);
  PNRF_REQUIRE(net == PNRF_NET_SAMPLER || net == PNRF_NET_REFINE || net == PNRF_NET_NERF, PNRF_E_ARG,
               "pnrf_mlp_pack: unknown net kind %d", net);
  const int prec = net == PNRF_NET_SAMPLER ? PREC_F32 : PREC_BF16;
  // Supported shapes (pnrf_layout.h, "free shape parameters"): width 256 and 8 samples per ray are fixed; free are the number of hidden layers
  // (mmnetdepth / netdepth), the sampler's ray points (N_point_ray_enc: input 6 P) and the refine net's neighbour views (num_neighbor: input 48 + 24 nb)
  static const char* SUPPORTED = "supported: hidden width 256, N_samples 8 (sampler 6*P -> D x 256 -> 27, any P >= 1; refine 48 + 24*nb -> D x 256 -> 35, nb = 1..8; "
                                 "DoNeRFTRT 63 -> (netdepth - 1) x 256 -> [256 + 27] -> 4 with 3 <= netdepth <= 8; sampler / refine depth 2 <= D <= 32; the NeRF class: D = 8, skips = [4])";
  const int nhid = n_layers - 2;
  PNRF_REQUIRE(n_layers >= 3 && nhid <= MAX_NHID, PNRF_E_SHAPE, "pnrf_mlp_pack: net %d with %d Linear layers; %s", net, n_layers, SUPPORTED);
  const int in0 = in_dim[0];
  int npts = 0, nbv = 0;
  if (net == PNRF_NET_SAMPLER) {
    npts = in0 / 6;
    PNRF_REQUIRE(in0 >= 6 && in0 % 6 == 0, PNRF_E_SHAPE, "pnrf_mlp_pack: sampler input width %d is not 6 * N_point_ray_enc; %s", in0, SUPPORTED);
  } else if (net == PNRF_NET_REFINE) {
    nbv = (in0 - 48) / 24;
    PNRF_REQUIRE(in0 >= 72 && (in0 - 48) % 24 == 0 && nbv <= MAX_NB, PNRF_E_SHAPE, "pnrf_mlp_pack: refine input width %d is not 48 + 24 * num_neighbor with 1 <= num_neighbor <= %d; %s",
                 in0, MAX_NB, SUPPORTED);
  } else {
    PNRF_REQUIRE(in0 == N_IN, PNRF_E_SHAPE, "pnrf_mlp_pack: NeRF input width %d, kernels encode 10 positional octaves (63); %s", in0, SUPPORTED);
    // the reference's skip='auto' puts the view encoding at layer 7 D / 8 (run_nerf_helpers.py:1190-1201): the last layer only up to D = 8
    PNRF_REQUIRE(n_layers <= 8, PNRF_E_SHAPE, "pnrf_mlp_pack: DoNeRFTRT with netdepth %d: from 9 layers on skip='auto' feeds the view encoding into a hidden layer; %s", n_layers, SUPPORTED);
  }
  const int outN = net == PNRF_NET_SAMPLER ? S_OUT : net == PNRF_NET_REFINE ? R_OUT : N_OUT;
  const int last_in = net == PNRF_NET_NERF ? W_HID + N_INV : W_HID;
  for (int l = 0; l < n_layers; ++l) {
    const int ei = l == 0 ? in0 : (l == n_layers - 1 ? last_in : W_HID);
    const int eo = l == n_layers - 1 ? outN : W_HID;
    PNRF_REQUIRE(W[l] && b[l], PNRF_E_ARG, "pnrf_mlp_pack: null weight/bias at layer %d", l);
    PNRF_REQUIRE(in_dim[l] == ei && out_dim[l] == eo, PNRF_E_SHAPE,
                 "pnrf_mlp_pack: net %d layer %d is %dx%d where %dx%d is needed; %s", net, l, out_dim[l], in_dim[l], eo, ei, SUPPORTED);
  }
  const bool full_stream = net != PNRF_NET_SAMPLER || npts == S_NPTS;      // the sampler's unfolded stream (module-level forward, SAMPLER_F32_FULL): P = 48 only
  const int nv = refine_nv(nbv);

  std::vector<Layer> Ls(n_layers);
  for (int l = 0; l < n_layers; ++l) {
    Layer& L = Ls[l];
    L.W = W[l]; L.b = b[l]; L.in_dim = in_dim[l]; L.out_dim = out_dim[l];
    L.nt = prec == PREC_BF16 ? NT_HID : NT16_HID; L.out_map = identity_out(W_HID);
    L.nk = prec == PREC_BF16 ? KS_HID : W_HID / 4;
    L.in_map = hidden_in(prec);
  }
  if (net == PNRF_NET_REFINE) scale_for_elu(Ls);          // every refine kernel computes its ELU on the log2(e) scale
  std::vector<int> in0_map, inx_map, out_map;
  Layer& F = Ls[0];
  Layer& Z = Ls[n_layers - 1];
  if (net == PNRF_NET_SAMPLER) {
    F.nk = S_KS0; F.in_map.assign(S_KS0 * 4, -1);
    for (int kk = 0; kk < S_KS0; ++kk) for (int q = 0; q < 4; ++q) F.in_map[kk * 4 + q] = full_stream ? sampler_in0(kk, q) : -1;
    Z.nt = S_NT_LAST; Z.out_map.assign(16 * S_NT_LAST, -1);
    for (int tt = 0; tt < S_NT_LAST; ++tt) for (int q = 0; q < 4; ++q) for (int r = 0; r < 4; ++r) Z.out_map[tt * 16 + 4 * q + r] = sampler_out(tt, q, r);
  } else if (net == PNRF_NET_REFINE) {
    const int ks0 = 3 * nv + 3;
    F.nk = ks0; F.in_map.assign(ks0 * 16, -1);
    for (int ks = 0; ks < ks0; ++ks) for (int h = 0; h < 2; ++h) for (int j = 0; j < 8; ++j) F.in_map[(ks * 2 + h) * 8 + j] = refine_in0_nv(nv, nbv, ks, h, j);
    Z.nt = R_NT_LAST; Z.out_map.assign(64, -1);
    for (int h = 0; h < 2; ++h) for (int g = 0; g < 16; ++g) {
      Z.out_map[acc_row(g, h)] = refine_out0(g, h);
      Z.out_map[32 + acc_row(g, h)] = refine_out1(g, h);
    }
  } else {
    F.nk = N_KS0; F.in_map.assign(N_KS0 * 16, -1);
    for (int ks = 0; ks < N_KS0; ++ks) for (int h = 0; h < 2; ++h) for (int j = 0; j < 8; ++j) F.in_map[(ks * 2 + h) * 8 + j] = nerf_in0(ks, h, j);
    Z.nt = 1; Z.nk = N_KS_LAST; Z.in_map.resize(N_KS_LAST * 16);
    inx_map.assign(N_KSX * 16, -1);
    for (int e = 0; e < N_KSX; ++e) for (int h = 0; h < 2; ++h) for (int j = 0; j < 8; ++j) {
      const int v = nerf_inx(e, h, j);
      inx_map[(e * 2 + h) * 8 + j] = v;
      Z.in_map[((KS_HID + e) * 2 + h) * 8 + j] = v >= 0 ? W_HID + v : -1;
    }
    Z.out_map.assign(32, -1);
    for (int h = 0; h < 2; ++h) for (int g = 0; g < 16; ++g) Z.out_map[acc_row(g, h)] = nerf_out(g, h);
  }
  in0_map = F.in_map;
  if (prec == PREC_F32) {
    out_map = Z.out_map;                       // [tile][16 rows]: lane quarter q, reg r -> [tile*16 + 4q + r]
  } else {
    out_map.assign(Z.nt * 32, -1);
    for (int to = 0; to < Z.nt; ++to) for (int h = 0; h < 2; ++h) for (int g = 0; g < 16; ++g)
      out_map[(to * 2 + h) * 16 + g] = Z.out_map[to * 32 + acc_row(g, h)];
  }

  size_t slots = 0;
  for (auto& L : Ls) slots += layer_slots(L, prec);
  slots += (NSLOTS - slots % NSLOTS) % NSLOTS;
  const uint32_t expect = net == PNRF_NET_SAMPLER ? s_nslots(nhid) : net == PNRF_NET_REFINE ? refine_slots(nhid, nv) : n_nslots(nhid);
  PNRF_REQUIRE(slots == expect, PNRF_E_SHAPE, "pnrf_mlp_pack: internal layout mismatch (%zu slots, kernels expect %u)", slots, expect);
  if (!full_stream) slots = 0;                   // no unfolded stream for this sampler

  std::vector<char> blob(slots * SLOT_BYTES, 0);
  const int tile_rows = prec == PREC_BF16 ? 32 : 16;
  size_t nbias = 0;
  for (auto& L : Ls) nbias += (size_t)L.nt * tile_rows;
  std::vector<float> bias(nbias, 0.f);
  size_t so = 0, bo = 0;
  for (auto& L : Ls) {
    if (full_stream) pack_layer(L, prec, blob.data() + so * SLOT_BYTES);
    pack_bias(L, prec, bias.data() + bo);
    so += layer_slots(L, prec);
    bo += (size_t)L.nt * tile_rows;
  }

  // refine: the same stream with fp16 operands (default of the refine stage; the bf16 stream above stays as PNRF_VARIANT_BF16)
  std::vector<char> blob_f16;
  size_t slots_f16 = 0;
  if (net == PNRF_NET_REFINE) {
    slots_f16 = slots;
    blob_f16.assign(slots_f16 * SLOT_BYTES, 0);
    size_t sf = 0;
    for (auto& L : Ls) { pack_layer_f16(L, blob_f16.data() + sf * SLOT_BYTES); sf += layer_slots(L, prec); }
  }

  // refine: the streams of the 16x16x32 engine (layer_e16 / refine16_kernel; fp16 operands, log2(e)-scaled like every refine stream) and their bias table:
  // the full first layer for rows from memory, and for the projecting head the first layer with the eight Pluecker 6-vectors folded into one (fp64 sum)
  std::vector<char> blob_r16, blob_r16f;
  std::vector<float> bias_r16, wfold_r;
  if (net == PNRF_NET_REFINE) {
    const int nv4 = refine16_nv(nbv), fin = 6 + 24 * nbv;
    wfold_r.assign((size_t)W_HID * fin, 0.f);
    for (int o = 0; o < W_HID; ++o) {
      for (int c = 0; c < 6; ++c) {
        double acc = 0.0;
        for (int sm = 0; sm < 8; ++sm) acc += (double)W[0][(size_t)o * in0 + 6 * sm + c];
        wfold_r[(size_t)o * fin + c] = (float)acc;
      }
      for (int k = 0; k < 24 * nbv; ++k) wfold_r[(size_t)o * fin + 6 + k] = W[0][(size_t)o * in0 + 48 + k];
    }
    for (int fold = 0; fold < 2; ++fold) {
      const int ks0 = refine16_ks0(nv4, fold != 0);
      std::vector<Layer> Lr = Ls;
      for (auto& L : Lr) {
        L.nt = W_HID / 16; L.out_map = identity_out(W_HID);
        L.nk = NB_KS_H; L.in_map.assign(NB_KS_H * 32, -1);
        for (int ks = 0; ks < NB_KS_H; ++ks) for (int g = 0; g < 4; ++g) for (int j = 0; j < 8; ++j) L.in_map[(ks * 4 + g) * 8 + j] = hidden_feat_h16(ks, g, j);
      }
      Layer& G = Lr[0];
      if (fold) { G.W = wfold_r.data(); G.in_dim = fin; }
      G.nk = ks0; G.in_map.assign(ks0 * 32, -1);
      for (int ks = 0; ks < ks0; ++ks) for (int g = 0; g < 4; ++g) for (int j = 0; j < 8; ++j) G.in_map[(ks * 4 + g) * 8 + j] = refine16_in0(nv4, nbv, ks, g, j, fold != 0);
      Layer& Y = Lr[n_layers - 1];
      Y.nt = 2 * R16_NTP_LAST; Y.out_map.assign(16 * Y.nt, -1);
      for (int T = 0; T < Y.nt; ++T) for (int r16 = 0; r16 < 16; ++r16) Y.out_map[T * 16 + r16] = refine16_out(T, r16);
      auto bs = [&](const Layer& L) { return ((size_t)(L.nt / 2) * L.nk * 2 + SLOT_FRAGS - 1) / SLOT_FRAGS; };
      size_t sl = 0, nb = 0;
      for (auto& L : Lr) { sl += bs(L); nb += (size_t)L.nt * 16; }
      sl += (NSLOTS - sl % NSLOTS) % NSLOTS;
      PNRF_REQUIRE(sl == (size_t)refine16_slots(nhid, nv4, fold != 0) && nb == (size_t)r16_nbias(nhid), PNRF_E_SHAPE,
                   "pnrf_mlp_pack: internal layout mismatch (refine 16x16 stream %zu slots, %zu bias floats)", sl, nb);
      std::vector<char>& blob_x = fold ? blob_r16f : blob_r16;
      blob_x.assign(sl * SLOT_BYTES, 0);
      if (!fold) bias_r16.assign(nb, 0.f);
      size_t sb = 0, bb = 0;
      for (auto& L : Lr) {
        pack_layer_b16(L, blob_x.data() + sb * SLOT_BYTES, true);
        if (!fold) pack_bias(L, PREC_F32, bias_r16.data() + bb);
        sb += bs(L); bb += (size_t)L.nt * 16;
      }
    }
  }

  // sampler: second stream with the folded first layer Wf[256x6] = sum_p W0[:, 6p:6p+6] (fp64 sum)
  std::vector<char> blob_fold;
  std::vector<float> wfold;
  size_t slots_fold = 0;
  if (net == PNRF_NET_SAMPLER) {
    wfold.assign((size_t)W_HID * 6, 0.f);
    for (int o = 0; o < W_HID; ++o)
      for (int j = 0; j < 6; ++j) {
        double acc = 0.0;
        for (int pnt = 0; pnt < npts; ++pnt) acc += (double)W[0][(size_t)o * in0 + 6 * pnt + j];
        wfold[(size_t)o * 6 + j] = (float)acc;
      }
    std::vector<Layer> Lf = Ls;
    Layer& G = Lf[0];
    G.W = wfold.data(); G.in_dim = 6; G.nk = 4 * SF_KS4_0; G.in_map.assign(G.nk * 4, -1);
    for (int kk = 0; kk < 2; ++kk) for (int q = 0; q < 4; ++q) G.in_map[kk * 4 + q] = (4 * kk + q < 6) ? 4 * kk + q : -1;
    for (auto& L : Lf) slots_fold += layer_slots(L, prec);
    slots_fold += (NSLOTS - slots_fold % NSLOTS) % NSLOTS;
    PNRF_REQUIRE(slots_fold == (size_t)sf_nslots(nhid), PNRF_E_SHAPE, "pnrf_mlp_pack: internal layout mismatch (folded stream %zu slots, expected %d)", slots_fold, sf_nslots(nhid));
    blob_fold.assign(slots_fold * SLOT_BYTES, 0);
    size_t sf = 0;
    for (auto& L : Lf) { pack_layer(L, prec, blob_fold.data() + sf * SLOT_BYTES); sf += layer_slots(L, prec); }
  }

  // sampler: third stream, folded first layer, split fp16 (hi / lo*2^11) for layer_h16x2
  std::vector<char> blob_h16;
  size_t slots_h16 = 0;
  if (net == PNRF_NET_SAMPLER) {
    std::vector<Layer> Lh = Ls;
    scale_for_elu(Lh);                                         // weights only: this stream's kernel scales the shared bias table itself
    for (auto& L : Lh) {                                       // hidden geometry of the 16x16x32 engine
      L.nk = SH_KS_H; L.in_map.assign(SH_KS_H * 32, -1);
      for (int ks = 0; ks < SH_KS_H; ++ks) for (int g = 0; g < 4; ++g) for (int j = 0; j < 8; ++j) L.in_map[(ks * 4 + g) * 8 + j] = hidden_feat_h16(ks, g, j);
    }
    Layer& G = Lh[0];
    G.W = wfold.data(); G.in_dim = 6; G.nk = 1; G.in_map.assign(32, -1);
    for (int j = 0; j < 6; ++j) G.in_map[j] = j;               // group 0 holds the 6 Pluecker features, the rest is padding
    auto hs = [&](const Layer& L) { return ((size_t)(L.nt / 2) * L.nk * 4 + SLOT_FRAGS - 1) / SLOT_FRAGS; };
    for (auto& L : Lh) slots_h16 += hs(L);
    slots_h16 += (NSLOTS - slots_h16 % NSLOTS) % NSLOTS;
    PNRF_REQUIRE(slots_h16 == (size_t)sh_nslots(nhid), PNRF_E_SHAPE, "pnrf_mlp_pack: internal layout mismatch (f16x2 stream %zu slots, expected %d)", slots_h16, sh_nslots(nhid));
    blob_h16.assign(slots_h16 * SLOT_BYTES, 0);
    size_t sh = 0;
    for (auto& L : Lh) { pack_layer_h16x2(L, blob_h16.data() + sh * SLOT_BYTES); sh += hs(L); }
  }

  // sampler: fourth stream = pass 1 of the two-pass scheme (sampler_p1_kernel): plain fp16 on the 32x32x16 engine, folded first layer in
  // split fp16; its own bias table in that engine's [tile][half][16] order, log2(e)-scaled for the ELU layers; and the constants of the
  // per-ray error model (pnrf_layout.h: P1_NCONST) taken from the fp32 weights
  std::vector<char> blob_p1;
  std::vector<float> bias_p1, p1c;
  if (net == PNRF_NET_SAMPLER) {
    std::vector<Layer> Lp = Ls;
    scale_for_elu(Lp);
    const std::vector<int> hid = hidden_in(PREC_BF16);
    for (auto& L : Lp) { L.nt = NT_HID; L.nk = KS_HID; L.in_map = hid; L.out_map = identity_out(W_HID); }
    Layer& G = Lp[0];
    G.W = wfold.data(); G.in_dim = 6; G.nk = 1; G.in_map.assign(16, -1);
    for (int j = 0; j < 6; ++j) G.in_map[j] = j;                   // half 0 supplies the 6 Pluecker features, half 1 padding
    Layer& Y = Lp[n_layers - 1];
    Y.nt = 1; Y.out_map.assign(32, -1);
    for (int hh = 0; hh < 2; ++hh) for (int g = 0; g < 16; ++g) Y.out_map[acc_row(g, hh)] = sampler_p1_out(g, hh);
    blob_p1.assign((size_t)p1_nslots(nhid) * SLOT_BYTES, 0);
    bias_p1.assign(p1_nbias(nhid), 0.f);
    size_t sp = 0, bp = 0;
    for (int l = 0; l < n_layers; ++l) {
      const Layer& L = Lp[l];
      if (l == 0) pack_layer0_f16x2(L, blob_p1.data());
      else pack_layer_f16(L, blob_p1.data() + sp * SLOT_BYTES);
      pack_bias(L, PREC_BF16, bias_p1.data() + bp);
      sp += l == 0 ? P1_SLOTS_L0 : layer_slots(L, PREC_BF16);
      bp += (size_t)L.nt * 32;
    }
    PNRF_REQUIRE(sp == (size_t)p1_slots_used(nhid) && bp == (size_t)p1_nbias(nhid), PNRF_E_SHAPE, "pnrf_mlp_pack: internal layout mismatch (pass-1 stream %zu slots, %zu bias floats)", sp, bp);
    p1c.assign(p1_nconst(nhid), 0.f);             // [0] output-layer constant, [1 + l] C of hidden layer l (sampler_p1_kernel)
    for (int l = 1; l <= nhid; ++l) {                             // C_l = max over input features j of the column norm sum_i W_l[i,j]^2
      double cmax = 0.0;
      for (int j = 0; j < W_HID; ++j) {
        double cn = 0.0;
        for (int i = 0; i < W_HID; ++i) { const double w = (double)W[l][(size_t)i * W_HID + j]; cn += w * w; }
        cmax = cn > cmax ? cn : cmax;
      }
      p1c[l] = (float)(cmax * (1.0 + 1e-6));
    }
    double mmax = 0.0;                                              // depth rows of the output layer
    for (int k = 0; k < 8; ++k) for (int j = 0; j < W_HID; ++j) { const double w = (double)W[n_layers - 1][(size_t)k * W_HID + j]; mmax = w * w > mmax ? w * w : mmax; }
    p1c[0] = (float)(mmax / (LOG2E_D * LOG2E_D) * (1.0 + 1e-6));
  }

  // DoNeRFTRT: second stream for the 16x16x32 engine (layer_b16): 16-row tiles in pairs, 32-deep k-steps
  std::vector<char> blob_b16;
  std::vector<float> bias_b16;
  size_t slots_b16 = 0;
  if (net == PNRF_NET_NERF) {
    std::vector<Layer> Lb = Ls;
    for (auto& L : Lb) {
      L.nt = W_HID / 16; L.out_map = identity_out(W_HID);
      L.nk = NB_KS_H; L.in_map.assign(NB_KS_H * 32, -1);
      for (int ks = 0; ks < NB_KS_H; ++ks) for (int g = 0; g < 4; ++g) for (int j = 0; j < 8; ++j) L.in_map[(ks * 4 + g) * 8 + j] = hidden_feat_h16(ks, g, j);
    }
    Layer& G = Lb[0];
    G.nk = NB_KS0; G.in_map.assign(NB_KS0 * 32, -1);
    for (int ks = 0; ks < NB_KS0; ++ks) for (int g = 0; g < 4; ++g) for (int j = 0; j < 8; ++j) G.in_map[(ks * 4 + g) * 8 + j] = nerf16_in0(ks, g, j);
    Layer& Y = Lb[n_layers - 1];
    Y.nt = 2; Y.out_map.assign(32, -1);
    for (int r = 0; r < N_OUT; ++r) Y.out_map[r] = r;                     // rows 0..3 of tile 0 = rgb, sigma; tile 1 is padding
    Y.nk = NB_KS_LAST; Y.in_map.resize(NB_KS_LAST * 32);
    for (int ks = 0; ks < NB_KS_H; ++ks) for (int g = 0; g < 4; ++g) for (int j = 0; j < 8; ++j) Y.in_map[(ks * 4 + g) * 8 + j] = hidden_feat_h16(ks, g, j);
    for (int g = 0; g < 4; ++g) for (int j = 0; j < 8; ++j) {
      const int v = nerf16_inx(g, j);
      Y.in_map[(NB_KS_H * 4 + g) * 8 + j] = v >= 0 ? W_HID + v : -1;
    }
    auto bs = [&](const Layer& L) { return ((size_t)(L.nt / 2) * L.nk * 2 + SLOT_FRAGS - 1) / SLOT_FRAGS; };
    size_t nb = 0;
    for (auto& L : Lb) { slots_b16 += bs(L); nb += (size_t)L.nt * 16; }
    slots_b16 += (NSLOTS - slots_b16 % NSLOTS) % NSLOTS;
    PNRF_REQUIRE(slots_b16 == (size_t)nb_nslots(nhid), PNRF_E_SHAPE, "pnrf_mlp_pack: internal layout mismatch (b16 stream %zu slots, expected %d)", slots_b16, nb_nslots(nhid));
    blob_b16.assign(slots_b16 * SLOT_BYTES, 0);
    slots_f16 = slots_b16;
    blob_f16.assign(slots_f16 * SLOT_BYTES, 0);                             // ... and with fp16 operands (the default of the NeRF stage)
    bias_b16.assign(nb, 0.f);
    size_t sb = 0, bb = 0;
    for (auto& L : Lb) {
      pack_layer_b16(L, blob_b16.data() + sb * SLOT_BYTES);
      pack_layer_b16(L, blob_f16.data() + sb * SLOT_BYTES, true);
      pack_bias(L, PREC_F32, bias_b16.data() + bb);                        // [tile][16 rows]
      sb += bs(L); bb += (size_t)L.nt * 16;
    }
  }

  pnrf_mlp* h = new pnrf_mlp();
  memset(h, 0, sizeof(*h));
  h->net = net; h->prec = prec; h->in_dim = in0; h->in_dim_x = net == PNRF_NET_NERF ? N_INV : 0; h->out_dim = outN;
  h->nhid = nhid; h->nb = nbv; h->npts = npts;
  h->nslots = (uint32_t)slots; h->nbias = (int)nbias;
  h->n_in0 = (int)in0_map.size(); h->n_inx = (int)inx_map.size(); h->n_out = (int)out_map.size();
  hipError_t e = hipGetDevice(&h->device);
  if (e == hipSuccess && !blob.empty()) {
    e = hipMalloc(&h->d_blob, blob.size());
    if (e == hipSuccess) e = hipMemcpy(h->d_blob, blob.data(), blob.size(), hipMemcpyHostToDevice);
  }
  if (e == hipSuccess) e = hipMalloc((void**)&h->d_bias, nbias * sizeof(float));
  if (e == hipSuccess) e = hipMemcpy(h->d_bias, bias.data(), nbias * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc((void**)&h->d_in0, in0_map.size() * sizeof(int));
  if (e == hipSuccess) e = hipMemcpy(h->d_in0, in0_map.data(), in0_map.size() * sizeof(int), hipMemcpyHostToDevice);
  if (e == hipSuccess && !inx_map.empty()) {
    e = hipMalloc((void**)&h->d_inx, inx_map.size() * sizeof(int));
    if (e == hipSuccess) e = hipMemcpy(h->d_inx, inx_map.data(), inx_map.size() * sizeof(int), hipMemcpyHostToDevice);
  }
  if (e == hipSuccess) e = hipMalloc((void**)&h->d_out, out_map.size() * sizeof(int));
  if (e == hipSuccess) e = hipMemcpy(h->d_out, out_map.data(), out_map.size() * sizeof(int), hipMemcpyHostToDevice);
  if (e == hipSuccess && net == PNRF_NET_SAMPLER) {
    h->nslots_fold = (uint32_t)slots_fold;
    e = hipMalloc(&h->d_blob_fold, blob_fold.size());
    if (e == hipSuccess) e = hipMemcpy(h->d_blob_fold, blob_fold.data(), blob_fold.size(), hipMemcpyHostToDevice);
  }
  if (e == hipSuccess && net == PNRF_NET_SAMPLER) {
    h->nslots_h16 = (uint32_t)slots_h16;
    e = hipMalloc(&h->d_blob_h16, blob_h16.size());
    if (e == hipSuccess) e = hipMemcpy(h->d_blob_h16, blob_h16.data(), blob_h16.size(), hipMemcpyHostToDevice);
  }
  if (e == hipSuccess && slots_f16) {
    h->nslots_f16 = (uint32_t)slots_f16;
    e = hipMalloc(&h->d_blob_f16, blob_f16.size());
    if (e == hipSuccess) e = hipMemcpy(h->d_blob_f16, blob_f16.data(), blob_f16.size(), hipMemcpyHostToDevice);
  }
  if (e == hipSuccess && net == PNRF_NET_SAMPLER) {
    h->nslots_p1 = p1_nslots(nhid); h->nbias_p1 = p1_nbias(nhid); h->n_p1c = p1_nconst(nhid);
    e = hipMalloc(&h->d_blob_p1, blob_p1.size());
    if (e == hipSuccess) e = hipMemcpy(h->d_blob_p1, blob_p1.data(), blob_p1.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void**)&h->d_bias_p1, bias_p1.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(h->d_bias_p1, bias_p1.data(), bias_p1.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void**)&h->d_p1c, p1c.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(h->d_p1c, p1c.data(), p1c.size() * sizeof(float), hipMemcpyHostToDevice);
  }
  if (e == hipSuccess && net == PNRF_NET_NERF) {
    h->nslots_b16 = (uint32_t)slots_b16; h->nbias_b16 = (int)bias_b16.size();
    e = hipMalloc(&h->d_blob_b16, blob_b16.size());
    if (e == hipSuccess) e = hipMemcpy(h->d_blob_b16, blob_b16.data(), blob_b16.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void**)&h->d_bias_b16, bias_b16.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(h->d_bias_b16, bias_b16.data(), bias_b16.size() * sizeof(float), hipMemcpyHostToDevice);
  }
  if (e == hipSuccess && net == PNRF_NET_REFINE) {
    h->nslots_b16 = (uint32_t)(blob_r16.size() / SLOT_BYTES); h->nbias_b16 = (int)bias_r16.size();
    e = hipMalloc(&h->d_blob_b16, blob_r16.size());
    if (e == hipSuccess) e = hipMemcpy(h->d_blob_b16, blob_r16.data(), blob_r16.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void**)&h->d_bias_b16, bias_r16.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(h->d_bias_b16, bias_r16.data(), bias_r16.size() * sizeof(float), hipMemcpyHostToDevice);
    h->nslots_fold = (uint32_t)(blob_r16f.size() / SLOT_BYTES);
    if (e == hipSuccess) e = hipMalloc(&h->d_blob_fold, blob_r16f.size());
    if (e == hipSuccess) e = hipMemcpy(h->d_blob_fold, blob_r16f.data(), blob_r16f.size(), hipMemcpyHostToDevice);
  }
  if (e == hipSuccess && net == PNRF_NET_SAMPLER && full_stream) {
    float tv[S_NPTS];
    pnrf_linspace(0.f, 1.f, S_NPTS, tv);
    e = hipMalloc((void**)&h->d_tvals, sizeof(tv));
    if (e == hipSuccess) e = hipMemcpy(h->d_tvals, tv, sizeof(tv), hipMemcpyHostToDevice);
  }
  if (e != hipSuccess) {
    set_error("pnrf_mlp_pack: device allocation/copy failed: %s", hipGetErrorString(e));
    pnrf_mlp_free(h);
    return (int)e;
  }
  *out = h;
  return 0;
}

extern "C" int pnrf_mlp_free(pnrf_mlp_t* h) {
  if (!h) return 0;
  if (h->d_blob) (void)hipFree(h->d_blob);
  if (h->d_blob_fold) (void)hipFree(h->d_blob_fold);
  if (h->d_blob_h16) (void)hipFree(h->d_blob_h16);
  if (h->d_blob_b16) (void)hipFree(h->d_blob_b16);
  if (h->d_bias_b16) (void)hipFree(h->d_bias_b16);
  if (h->d_blob_p1) (void)hipFree(h->d_blob_p1);
  if (h->d_bias_p1) (void)hipFree(h->d_bias_p1);
  if (h->d_p1c) (void)hipFree(h->d_p1c);
  if (h->d_blob_f16) (void)hipFree(h->d_blob_f16);
  if (h->d_bias) (void)hipFree(h->d_bias);
  if (h->d_in0) (void)hipFree(h->d_in0);
  if (h->d_inx) (void)hipFree(h->d_inx);
  if (h->d_out) (void)hipFree(h->d_out);
  if (h->d_tvals) (void)hipFree(h->d_tvals);
  delete h;
  return 0;
}

// ---- engine files -----------------------------------------------------------------------------------------------------
// One packed network as one flat byte image: a 128-byte header followed by the device buffers of the handle in a fixed order.
// This is what the reference keeps in its serialized TensorRT engines (pronerf/cli.py:105-157 builds them, trt_infer_v2.py
// deserializes them at start-up); here the "engine" is the pre-tiled weight stream, so the file is the stream.  Like a TensorRT
// plan it is only valid for the library build that wrote it: the header carries the ABI version and a tag of the stream layout.
#ifndef PNRF_LAYOUT_TAG
#define PNRF_LAYOUT_TAG 0u
#endif

namespace pnrf {

struct EngineHeader {
  char magic[8];                 // "PNRFENG\0"
  uint32_t format, abi, layout_tag, slot_bytes;
  int32_t net, prec, in_dim, in_dim_x, out_dim;
  uint32_t nslots, nslots_fold, nslots_h16, nslots_b16;
  int32_t nbias_b16, nbias, n_in0, n_inx, n_out, n_tvals;
  uint64_t payload_bytes, checksum;   // FNV-1a 64 of the payload
  uint32_t nslots_p1;                 // format 2: pass-1 stream of the two-pass sampler, its bias table and error-model constants
  int32_t nbias_p1, n_p1c;
  uint32_t nslots_f16;                // format 3: fp16-operand stream of the refine / NeRF stages
  uint16_t nhid, nb, npts;            // format 4: the net's free shape parameters (hidden layers behind layer 0, neighbour views, ray points)
  uint8_t reserved[2];
};
static constexpr uint32_t ENGINE_FORMAT = 5;      // 5: refine handles carry the stream of the 16x16x32 engine (nslots_b16 / nbias_b16)
static constexpr int ENGINE_SECTIONS = 14;
static_assert(sizeof(EngineHeader) == 128, "engine header is 128 bytes");
static const char ENGINE_MAGIC[8] = {'P', 'N', 'R', 'F', 'E', 'N', 'G', 0};

struct Section { void** dptr; size_t bytes; };

// the buffers of a handle, in file order; sizes come from the counts in `h` (which the header restores on load)
static int sections(pnrf_mlp* h, int n_tvals, Section* s) {
  int n = 0;
  s[n++] = {&h->d_blob, (size_t)h->nslots * SLOT_BYTES};
  s[n++] = {&h->d_blob_fold, (size_t)h->nslots_fold * SLOT_BYTES};
  s[n++] = {&h->d_blob_h16, (size_t)h->nslots_h16 * SLOT_BYTES};
  s[n++] = {&h->d_blob_b16, (size_t)h->nslots_b16 * SLOT_BYTES};
  s[n++] = {(void**)&h->d_bias_b16, (size_t)h->nbias_b16 * sizeof(float)};
  s[n++] = {(void**)&h->d_bias, (size_t)h->nbias * sizeof(float)};
  s[n++] = {(void**)&h->d_in0, (size_t)h->n_in0 * sizeof(int)};
  s[n++] = {(void**)&h->d_inx, (size_t)h->n_inx * sizeof(int)};
  s[n++] = {(void**)&h->d_out, (size_t)h->n_out * sizeof(int)};
  s[n++] = {(void**)&h->d_tvals, (size_t)n_tvals * sizeof(float)};
  s[n++] = {&h->d_blob_p1, (size_t)h->nslots_p1 * SLOT_BYTES};
  s[n++] = {(void**)&h->d_bias_p1, (size_t)h->nbias_p1 * sizeof(float)};
  s[n++] = {(void**)&h->d_p1c, (size_t)h->n_p1c * sizeof(float)};
  s[n++] = {&h->d_blob_f16, (size_t)h->nslots_f16 * SLOT_BYTES};
  return n;
}

// what pnrf_mlp_pack produces for a net kind (the counts the kernels rely on)
static void expected_counts(int net, int nhid, int nb, int npts, EngineHeader* w) {
  memset(w, 0, sizeof(*w));
  w->nhid = (uint16_t)nhid; w->nb = (uint16_t)nb; w->npts = (uint16_t)npts;
  switch (net) {
    case PNRF_NET_SAMPLER: {
      const bool full = npts == S_NPTS;
      w->prec = PREC_F32; w->in_dim = 6 * npts; w->out_dim = S_OUT; w->nslots = full ? s_nslots(nhid) : 0; w->nslots_fold = sf_nslots(nhid); w->nslots_h16 = sh_nslots(nhid);
      w->nbias = s_nbias(nhid); w->n_in0 = S_KS0 * 4; w->n_out = 16 * S_NT_LAST; w->n_tvals = full ? S_NPTS : 0;
      w->nslots_p1 = p1_nslots(nhid); w->nbias_p1 = p1_nbias(nhid); w->n_p1c = p1_nconst(nhid);
      break;
    }
    case PNRF_NET_REFINE:
      w->prec = PREC_BF16; w->in_dim = 48 + 24 * nb; w->out_dim = R_OUT; w->nslots = refine_slots(nhid, refine_nv(nb)); w->nbias = r_nbias(nhid);
      w->n_in0 = (3 * refine_nv(nb) + 3) * 16; w->n_out = R_NT_LAST * 32;
      w->nslots_b16 = refine16_slots(nhid, refine16_nv(nb), false); w->nslots_fold = refine16_slots(nhid, refine16_nv(nb), true); w->nbias_b16 = r16_nbias(nhid);
      w->nslots_f16 = w->nslots;
      break;
    case PNRF_NET_NERF:
      w->prec = PREC_BF16; w->in_dim = N_IN; w->in_dim_x = N_INV; w->out_dim = N_OUT; w->nslots = n_nslots(nhid); w->nslots_b16 = nb_nslots(nhid);
      w->nbias = n_nbias(nhid); w->nbias_b16 = nb_nbias(nhid); w->n_in0 = N_KS0 * 16; w->n_inx = N_KSX * 16; w->n_out = 32;
      w->nslots_f16 = w->nslots_b16;
      break;
    default:      // PNRF_NET_NERFCLS
      w->prec = PREC_BF16; w->in_dim = N_IN; w->in_dim_x = N_INV; w->out_dim = 4; w->nslots = C_NSLOTS; w->nslots_b16 = CB_NSLOTS;
      w->nbias = C_NBIAS; w->nbias_b16 = CB_NBIAS; w->n_in0 = N_KS0 * 16; w->n_inx = N_KSX * 16; w->n_out = 64;
      w->nslots_f16 = CB_NSLOTS;
      break;
  }
}

static uint64_t fnv1a(const uint8_t* p, size_t n) {
  uint64_t x = 1469598103934665603ull;
  for (size_t i = 0; i < n; ++i) { x ^= p[i]; x *= 1099511628211ull; }
  return x;
}

}  // namespace pnrf

extern "C" int pnrf_mlp_serialize(const pnrf_mlp_t* hc, void* buf, int64_t capacity, int64_t* size) {
  using namespace pnrf;
  PNRF_REQUIRE(hc && size, PNRF_E_ARG, "pnrf_mlp_serialize: null argument");
  pnrf_mlp* h = const_cast<pnrf_mlp*>(hc);
  const int n_tvals = h->d_tvals ? S_NPTS : 0;
  Section sec[ENGINE_SECTIONS];
  const int ns = sections(h, n_tvals, sec);
  size_t payload = 0;
  for (int i = 0; i < ns; ++i) {
    PNRF_REQUIRE((sec[i].bytes == 0) == (*sec[i].dptr == nullptr), PNRF_E_STATE, "pnrf_mlp_serialize: inconsistent handle (section %d)", i);
    payload += sec[i].bytes;
  }
  *size = (int64_t)(sizeof(EngineHeader) + payload);
  if (!buf) return 0;                                               // size query
  PNRF_REQUIRE(capacity >= *size, PNRF_E_ARG, "pnrf_mlp_serialize: buffer of %lld bytes, need %lld", (long long)capacity, (long long)*size);
  int prev = 0;
  PNRF_HIP(hipGetDevice(&prev));
  PNRF_HIP(hipSetDevice(h->device));
  uint8_t* p = (uint8_t*)buf + sizeof(EngineHeader);
  hipError_t e = hipSuccess;
  for (int i = 0; i < ns && e == hipSuccess; ++i) {
    if (sec[i].bytes) e = hipMemcpy(p, *sec[i].dptr, sec[i].bytes, hipMemcpyDeviceToHost);
    p += sec[i].bytes;
  }
  (void)hipSetDevice(prev);
  if (e != hipSuccess) { set_error("pnrf_mlp_serialize: device read failed: %s", hipGetErrorString(e)); return (int)e; }
  EngineHeader hd;
  memset(&hd, 0, sizeof(hd));
  memcpy(hd.magic, ENGINE_MAGIC, 8);
  hd.format = ENGINE_FORMAT; hd.abi = PNRF_ABI_VERSION; hd.layout_tag = PNRF_LAYOUT_TAG; hd.slot_bytes = SLOT_BYTES;
  hd.net = h->net; hd.prec = h->prec; hd.in_dim = h->in_dim; hd.in_dim_x = h->in_dim_x; hd.out_dim = h->out_dim;
  hd.nslots = h->nslots; hd.nslots_fold = h->nslots_fold; hd.nslots_h16 = h->nslots_h16; hd.nslots_b16 = h->nslots_b16;
  hd.nbias_b16 = h->nbias_b16; hd.nbias = h->nbias; hd.n_in0 = h->n_in0; hd.n_inx = h->n_inx; hd.n_out = h->n_out; hd.n_tvals = n_tvals;
  hd.nslots_p1 = h->nslots_p1; hd.nbias_p1 = h->nbias_p1; hd.n_p1c = h->n_p1c; hd.nslots_f16 = h->nslots_f16;
  hd.nhid = (uint16_t)h->nhid; hd.nb = (uint16_t)h->nb; hd.npts = (uint16_t)h->npts;
  hd.payload_bytes = payload;
  hd.checksum = fnv1a((const uint8_t*)buf + sizeof(EngineHeader), payload);
  memcpy(buf, &hd, sizeof(hd));
  return 0;
}

extern "C" int pnrf_mlp_deserialize(const void* buf, int64_t size, pnrf_mlp_t** out) {
  using namespace pnrf;
  PNRF_REQUIRE(buf && out, PNRF_E_ARG, "pnrf_mlp_deserialize: null argument");
  PNRF_REQUIRE(size >= (int64_t)sizeof(EngineHeader), PNRF_E_ARG, "pnrf_mlp_deserialize: %lld bytes is shorter than the header", (long long)size);
  EngineHeader hd;
  memcpy(&hd, buf, sizeof(hd));
  PNRF_REQUIRE(memcmp(hd.magic, ENGINE_MAGIC, 8) == 0, PNRF_E_ARG, "pnrf_mlp_deserialize: not an engine file (bad magic)");
  PNRF_REQUIRE(hd.format == ENGINE_FORMAT && hd.abi == PNRF_ABI_VERSION && hd.layout_tag == PNRF_LAYOUT_TAG && hd.slot_bytes == SLOT_BYTES, PNRF_E_STATE,
               "pnrf_mlp_deserialize: engine written by another build (format %u abi %u layout %08x, this library: %u %d %08x); export it again",
               hd.format, hd.abi, hd.layout_tag, ENGINE_FORMAT, PNRF_ABI_VERSION, (unsigned)PNRF_LAYOUT_TAG);
  PNRF_REQUIRE(hd.net == PNRF_NET_SAMPLER || hd.net == PNRF_NET_REFINE || hd.net == PNRF_NET_NERF || hd.net == PNRF_NET_NERFCLS, PNRF_E_ARG,
               "pnrf_mlp_deserialize: unknown net kind %d", hd.net);
  PNRF_REQUIRE(hd.nbias_b16 >= 0 && hd.nbias >= 0 && hd.n_in0 >= 0 && hd.n_inx >= 0 && hd.n_out >= 0 && (hd.n_tvals == 0 || hd.n_tvals == S_NPTS) &&
                   hd.nslots < (1u << 16) && hd.nslots_fold < (1u << 16) && hd.nslots_h16 < (1u << 16) && hd.nslots_b16 < (1u << 16) &&
                   hd.nbias < (1 << 24) && hd.nbias_b16 < (1 << 24) && hd.n_in0 < (1 << 20) && hd.n_inx < (1 << 20) && hd.n_out < (1 << 20) &&
                   hd.nslots_p1 < (1u << 16) && hd.nslots_f16 < (1u << 16) && hd.nbias_p1 >= 0 && hd.nbias_p1 < (1 << 24) && hd.n_p1c >= 0 && hd.n_p1c < (1 << 10),
               PNRF_E_ARG, "pnrf_mlp_deserialize: implausible section counts");
  {
    // section counts the kernels of this net kind index with compile-time constants: an image whose counts differ (a crafted file with a
    // matching non-cryptographic checksum, or a stale one under the same layout tag) would make them read past the buffers
    const bool shape_ok = hd.net == PNRF_NET_NERFCLS ? (hd.nhid == 0 && hd.nb == 0 && hd.npts == 0)
                          : (hd.nhid >= 1 && hd.nhid <= MAX_NHID && (hd.net == PNRF_NET_REFINE ? (hd.nb >= 1 && hd.nb <= MAX_NB) : hd.nb == 0) &&
                             (hd.net == PNRF_NET_SAMPLER ? (hd.npts >= 1 && hd.npts < 4096) : hd.npts == 0));
    PNRF_REQUIRE(shape_ok, PNRF_E_ARG, "pnrf_mlp_deserialize: shape parameters (hidden layers %u, views %u, ray points %u) outside what net kind %d supports",
                 hd.nhid, hd.nb, hd.npts, hd.net);
    EngineHeader want;
    expected_counts(hd.net, hd.nhid, hd.nb, hd.npts, &want);
    PNRF_REQUIRE(hd.prec == want.prec && hd.in_dim == want.in_dim && hd.in_dim_x == want.in_dim_x && hd.out_dim == want.out_dim &&
                     hd.nslots == want.nslots && hd.nslots_fold == want.nslots_fold && hd.nslots_h16 == want.nslots_h16 &&
                     hd.nslots_b16 == want.nslots_b16 && hd.nbias_b16 == want.nbias_b16 && hd.nbias == want.nbias && hd.n_in0 == want.n_in0 &&
                     hd.n_inx == want.n_inx && hd.n_out == want.n_out && hd.n_tvals == want.n_tvals && hd.nslots_p1 == want.nslots_p1 &&
                     hd.nbias_p1 == want.nbias_p1 && hd.n_p1c == want.n_p1c && hd.nslots_f16 == want.nslots_f16,
                 PNRF_E_ARG, "pnrf_mlp_deserialize: section counts do not match what net kind %d is packed as by this build", hd.net);
  }
  pnrf_mlp* h = new pnrf_mlp();
  memset(h, 0, sizeof(*h));
  h->net = hd.net; h->prec = hd.prec; h->in_dim = hd.in_dim; h->in_dim_x = hd.in_dim_x; h->out_dim = hd.out_dim;
  h->nslots = hd.nslots; h->nslots_fold = hd.nslots_fold; h->nslots_h16 = hd.nslots_h16; h->nslots_b16 = hd.nslots_b16;
  h->nbias_b16 = hd.nbias_b16; h->nbias = hd.nbias; h->n_in0 = hd.n_in0; h->n_inx = hd.n_inx; h->n_out = hd.n_out;
  h->nslots_p1 = hd.nslots_p1; h->nbias_p1 = hd.nbias_p1; h->n_p1c = hd.n_p1c; h->nslots_f16 = hd.nslots_f16;
  h->nhid = hd.nhid; h->nb = hd.nb; h->npts = hd.npts;
  Section sec[ENGINE_SECTIONS];
  const int ns = sections(h, hd.n_tvals, sec);
  size_t payload = 0;
  for (int i = 0; i < ns; ++i) payload += sec[i].bytes;
  const uint8_t* p = (const uint8_t*)buf + sizeof(EngineHeader);
  if (payload != hd.payload_bytes || (int64_t)(sizeof(EngineHeader) + payload) != size) {
    set_error("pnrf_mlp_deserialize: truncated or padded engine (%lld bytes, header describes %lld)", (long long)size,
              (long long)(sizeof(EngineHeader) + payload));
    delete h;
    return PNRF_E_ARG;
  }
  if (fnv1a(p, payload) != hd.checksum) {
    set_error("pnrf_mlp_deserialize: checksum mismatch (corrupt engine file)");
    delete h;
    return PNRF_E_ARG;
  }
  hipError_t e = hipGetDevice(&h->device);
  for (int i = 0; i < ns && e == hipSuccess; ++i) {
    if (sec[i].bytes) {
      e = hipMalloc(sec[i].dptr, sec[i].bytes);
      if (e == hipSuccess) e = hipMemcpy(*sec[i].dptr, p, sec[i].bytes, hipMemcpyHostToDevice);
    }
    p += sec[i].bytes;
  }
  if (e != hipSuccess) {
    set_error("pnrf_mlp_deserialize: device allocation/copy failed: %s", hipGetErrorString(e));
    pnrf_mlp_free(h);
    return (int)e;
  }
  *out = h;
  return 0;
}

extern "C" int pnrf_mlp_set_variant(pnrf_mlp_t* h, int variant) {
  PNRF_REQUIRE(h, PNRF_E_ARG, "pnrf_mlp_set_variant: null handle");
  const bool sampler = h->net == PNRF_NET_SAMPLER;
  const bool ok = variant == PNRF_VARIANT_DEFAULT || (sampler && (variant == PNRF_VARIANT_SAMPLER_F32 || variant == PNRF_VARIANT_SAMPLER_F32_FULL || variant == PNRF_VARIANT_SAMPLER_SPLIT)) ||
                  ((h->net == PNRF_NET_NERF || h->net == PNRF_NET_NERFCLS) && variant == PNRF_VARIANT_BF16_32X32) ||
                  (!sampler && variant == PNRF_VARIANT_BF16) || (h->net == PNRF_NET_REFINE && variant == PNRF_VARIANT_REFINE_16X16) || ((h->net == PNRF_NET_NERF || h->net == PNRF_NET_NERFCLS) && variant == PNRF_VARIANT_F16) ||
                  ((h->net == PNRF_NET_NERF || h->net == PNRF_NET_NERFCLS) && variant == PNRF_VARIANT_NERF_4X64);
  PNRF_REQUIRE(ok, PNRF_E_ARG, "pnrf_mlp_set_variant: variant %d does not exist for net kind %d", variant, h->net);
  h->variant = variant;
  return 0;
}

extern "C" int pnrf_mlp_set_shape(pnrf_mlp_t* h, int shape) {
  PNRF_REQUIRE(h, PNRF_E_ARG, "pnrf_mlp_set_shape: null handle");
  PNRF_REQUIRE(shape == PNRF_SHAPE_AUTO || shape == PNRF_SHAPE_WIDE || shape == PNRF_SHAPE_NARROW, PNRF_E_ARG,
               "pnrf_mlp_set_shape: PNRF_SHAPE_AUTO (0), _WIDE (8) or _NARROW (4), got %d", shape);
  h->shape = shape;
  return 0;
}

extern "C" int pnrf_mlp_kind(const pnrf_mlp_t* h, int* net, int* in_dim, int* in_dim_x, int* out_dim) {
  PNRF_REQUIRE(h, PNRF_E_ARG, "pnrf_mlp_kind: null handle");
  if (net) *net = h->net;
  if (in_dim) *in_dim = h->in_dim;
  if (in_dim_x) *in_dim_x = h->in_dim_x;
  if (out_dim) *out_dim = h->out_dim;
  return 0;
}
