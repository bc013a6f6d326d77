// pnrf_common.h — host-side helpers shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "../../include/pronerf_hip.h"
#include "pnrf_layout.h"

namespace pnrf {

void set_error(const char* fmt, ...);

#define PNRF_REQUIRE(cond, code, ...)      \
  do {                                     \
    if (!(cond)) {                         \
      pnrf::set_error(__VA_ARGS__);        \
      return (code);                       \
    }                                      \
  } while (0)

#define PNRF_HIP(expr)                                                                   \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) {                                                              \
      pnrf::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return (int)e_;                                                                    \
    }                                                                                    \
  } while (0)

#define PNRF_LAUNCH_CHECK()                                                              \
  do {                                                                                   \
    hipError_t e_ = hipGetLastError();                                                   \
    if (e_ != hipSuccess) {                                                              \
      pnrf::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); \
      return (int)e_;                                                                    \
    }                                                                                    \
  } while (0)

}  // namespace pnrf

// Packed, device-resident network (immutable after pnrf_mlp_pack).
struct pnrf_mlp {
  int net;               // PNRF_NET_*
  int prec;              // pnrf::PREC_*
  int in_dim, in_dim_x;  // input width, extra (view) input width
  int out_dim;
  void* d_blob;          // weight stream: nslots x 16 KiB
  uint32_t nslots;
  void* d_blob_fold;     // sampler: stream with the folded 6->256 first layer (fused path); refine: 16x16x32 stream with the folded Pluecker inputs (projecting head)
  uint32_t nslots_fold;
  void* d_blob_h16;      // sampler only: folded stream in split fp16 (hi / lo*2^11 planes) for layer_h16x2
  uint32_t nslots_h16;
  void* d_blob_b16;      // DoNeRFTRT / NeRF class: bf16 stream for the 16x16x32 engine (layer_b16); refine: fp16 stream of layer_e16 (rows from memory)
  uint32_t nslots_b16;
  float* d_bias_b16;     // ... and its biases, [tile of 16 rows][16]
  int nbias_b16;
  float* d_bias;         // packed biases
  int nbias;
  int* d_in0;            // layer-0 input map   (device copy, for the module-level forward)
  int* d_inx;            // extra-input map (nerf view k-steps) or NULL
  int* d_out;            // last-layer output map [tiles*64]: (tile, half, reg) -> output index
  int n_in0, n_inx, n_out;
  void* d_blob_p1;       // sampler only: pass-1 stream of the two-pass scheme (plain fp16, 32x32x16 engine; sampler_p1_kernel)
  uint32_t nslots_p1;
  float* d_bias_p1;      // ... its bias table ([tile][half][16], log2(e)-scaled for the ELU layers)
  int nbias_p1;
  float* d_p1c;          // ... and the constants of the per-ray error model (pnrf_layout.h: P1_NCONST)
  int n_p1c;
  void* d_blob_f16;      // refine / NeRF handles: the stream of the default engine with fp16 operands (refine: as d_blob; NeRF: as d_blob_b16)
  uint32_t nslots_f16;
  float* d_tvals;        // sampler only: t = torch.linspace(0,1,48) of the ray points (trt.py:556-557)
  int nhid;              // hidden 256 -> 256 layers behind layer 0: sampler / refine mmnetdepth - 1 (Fern: 5), DoNeRFTRT netdepth - 2 (Fern: 6); class net: 0 (fixed structure)
  int nb;                // refine: neighbour views (num_neighbor, 1 .. 8; Fern: 4); other nets: 0
  int npts;              // sampler: ray points of the encoding (N_point_ray_enc; Fern: 48); other nets: 0
  int device;
  int variant;           // PNRF_VARIANT_* (pnrf_mlp_set_variant); 0 = default kernels
  int shape;             // workgroup shape of the fused stages (pnrf_mlp_set_shape): PNRF_SHAPE_AUTO = per launch from the column count, or one forced
};

// pnrf_sampler_fwd_ws with the caller's word that the workspace's counters are zero (a context's workspace): no memset on the stream
int pnrf_sampler_fwd_ws_impl(const pnrf_mlp_t* h, const float* rays, int64_t n, float* depth_sorted, float* add_sorted, float* mul_sorted,
                             int64_t* sort_idx, float* mm_rgb, float* depth_raw, void* workspace, int64_t workspace_bytes, float kappa,
                             bool ws_clean, void* stream);
// pnrf_nerf_train_fwd with a batch queue (two ints, zero between launches): the context's NeRF stage hands its batches out dynamically
int pnrf_nerf_fwd_queue_impl(const pnrf_mlp_t* h, const float* pts, const float* rays, const float* z, const float* add_sorted,
                             const float* mul_sorted, const float* noise, float clampv, int white_bkgd, int S, float* rgbd, float* raw,
                             int64_t n, int* queue, void* stream);
