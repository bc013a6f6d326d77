/* pronerf_hip.h — C ABI of libpronerf_hip.so (MI355X / gfx950).
 *
 * The reference (KAIST-VICLab/pronerf) is pure Python over torch; it has no FFI layer.  Its
 * "operator boundary" is the set of Python callables the driver scripts import by name
 * (run_S_eS_eN_alter_trt.py:19,27).  Each entry point below replaces the torch-op sequence of
 * one of those callables (cited per function, paths relative to the reference root) and is
 * what a ctypes binding on the reference side would call.
 *
 * Conventions
 *   - every pointer marked "dev" is device memory owned by the caller (e.g. a torch tensor's
 *     data_ptr()); tensors are contiguous, fp32 unless noted; nothing is allocated per call;
 *   - `stream` is a hipStream_t (NULL = default stream); all launches are asynchronous on it,
 *     there is no hidden device synchronisation;
 *   - return 0 = ok, < 0 = argument error (PNRF_E_*), > 0 = hipError_t; the message for the
 *     last non-zero return of the calling thread is available from pnrf_last_error();
 *   - handles are immutable once configured (pack / deserialize [+ pnrf_mlp_set_variant]): entry points are re-entrant across
 *     streams, and no entry point reads the process environment.
 */
#ifndef PRONERF_HIP_H
#define PRONERF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PNRF_ABI_VERSION 1

#define PNRF_E_ARG (-1)      /* bad argument (null pointer, negative size, wrong dims) */
#define PNRF_E_SHAPE (-2)    /* network shape not supported by the packed kernels */
#define PNRF_E_STATE (-3)    /* handle/context misuse (e.g. more rays than the context was sized for) */

/* network kinds for pnrf_mlp_pack */
#define PNRF_NET_SAMPLER 0   /* MinMaxRaySamplerTRT_Net / MinMaxRay_Net, 288->6x256->27, ELU  (run_nerf_helpers.py:1440-1507) */
#define PNRF_NET_REFINE 1    /* MinMaxRayEpiSamplerTRT_Net, 144->6x256->35, ELU                 (run_nerf_helpers.py:1509-1540) */
#define PNRF_NET_NERF 2      /* DoNeRFTRT(skip='auto'), 63->7x256->[256+27]->4, ReLU            (run_nerf_helpers.py:1186-1343) */
#define PNRF_NET_NERFCLS 3   /* NeRF(D=8,W=256,skips=[4],use_viewdirs=True): the fine net stages 1/2 train and save
                                (run_nerf_helpers.py:792-847).  12 Linear layers in the order pts_linears[0..7],
                                feature_linear, alpha_linear, views_linears[0], rgb_linear; output [rgb(3), alpha]. */

typedef struct pnrf_mlp pnrf_mlp_t;
typedef struct pnrf_ctx pnrf_ctx_t;
typedef struct pnrf_trainer pnrf_trainer_t;

int pnrf_abi_version(void);
const char* pnrf_last_error(void);

/* ---- weights ------------------------------------------------------------------------------
 * Pack one network's nn.Linear parameters (HOST pointers, torch layout W[out][in], n_layers
 * matrices) into the device-resident, pre-tiled weight stream consumed by the MFMA kernels.
 * Replaces load_state_dict + .to(device) for the three inference nets
 * (run_S_eS_eN_alter_trt.py:427-458, 468-481).  Synchronous; not on the render path.
 *
 * Shapes (round 6).  The reference sizes its modules from --mmnetdepth, --N_point_ray_enc, --num_neighbor, --netdepth
 * (run_S_eS_eN_alter_trt.py:62-82, 110-118, 427-457); the Fern configs are 6 / 48 / 4 / 8.  Hidden width 256 and 8 samples per ray are fixed; taken are
 *   PNRF_NET_SAMPLER  6 P -> D x 256 (ELU) -> 27      any number of ray points P >= 1 (the fused kernels run the folded first layer, K = 6), 2 <= D <= 32
 *   PNRF_NET_REFINE   48 + 24 nb -> D x 256 -> 35     num_neighbor nb = 1 .. 8, 2 <= D <= 32
 *   PNRF_NET_NERF     63 -> (D - 1) x 256 (ReLU) -> [256 + 27] -> 4     DoNeRFTRT, 3 <= netdepth D <= 8 (from 9 on the reference's skip='auto' moves the
 *                                                                        view input into a hidden layer, run_nerf_helpers.py:1190-1201)
 *   PNRF_NET_NERFCLS  the NeRF class with D = 8, skips = [4], use_viewdirs (12 Linear layers: pts0..7, feature, alpha, views, rgb)
 * No skip connections inside the sampler / refine stacks (mmnetskips beyond the depth, as in the Fern configs).  Everything else returns PNRF_E_SHAPE and
 * the message lists this set.  Three things exist for the Fern values only and say so: the sampler's UNFOLDED first layer (pnrf_mlp_fwd on a sampler handle,
 * PNRF_VARIANT_SAMPLER_F32_FULL: P = 48), PNRF_VARIANT_BF16_32X32 and pnrf_mlp_fwd on a DoNeRFTRT handle (netdepth 8), the training entry points (the
 * trainer's shapes).  pnrf_mlp_kind reports in_dim, from which P = in_dim / 6 resp. nb = (in_dim - 48) / 24 follow. */
int pnrf_mlp_pack(int net, const float* const* W, const float* const* b, const int* in_dim,
                  const int* out_dim, int n_layers, pnrf_mlp_t** out);
int pnrf_mlp_free(pnrf_mlp_t* h);

/* ---- engine files -------------------------------------------------------------------------
 * A packed network as one flat byte image (128-byte header + the weight stream and its maps):
 * the counterpart of the serialized TensorRT engines the reference builds with `export-trt`
 * (pronerf/cli.py:105-157, onnx2trt.get_engine) and deserializes at start-up
 * (run_S_eS_eN_alter_trt.py:490-499, trt_infer_v2.py NeRFEngine/MMEngine/RefineEngine).
 * serialize: buf == NULL only reports the size in *size; otherwise capacity >= *size and the image
 * is written to HOST memory (synchronous device read).  deserialize builds a new handle on the
 * current device from a HOST image; images from another library build (ABI version / stream layout
 * tag), truncated or corrupt images are refused with PNRF_E_STATE / PNRF_E_ARG.
 * pnrf_mlp_kind reports what a handle holds (any out pointer may be NULL). */
int pnrf_mlp_serialize(const pnrf_mlp_t* h, void* buf, int64_t capacity, int64_t* size);
int pnrf_mlp_deserialize(const void* buf, int64_t size, pnrf_mlp_t** out);
int pnrf_mlp_kind(const pnrf_mlp_t* h, int* net, int* in_dim, int* in_dim_x, int* out_dim);

/* Kernel variant of a packed network.  PNRF_VARIANT_DEFAULT is what pack / deserialize produce and what every product path runs:
 * sampler with the folded first layer in split fp16 (fp32-grade, three v_mfma_f32_16x16x32_f16 per product) — through pnrf_sampler_fwd_ws /
 * pnrf_render_rays_fwd as the second of two passes, see there —, refine stage on v_mfma_f32_32x32x16_f16 (fp16 operands: its refined depths
 * feed 2^9 positional octaves, and with bf16 operands they are the largest error of the frame), NeRF stage on v_mfma_f32_16x16x32_bf16
 * (the "bf16 MLP" of BASELINE.json configs[1]); fp32 accumulation everywhere.  fp16 kernels run with MODE.FP16_OVFL: packed activations
 * saturate at 65 504 instead of overflowing.  The others exist for parity tests and A/B timing (tools/perf_ab.py, tools/variant_psnr.py):
 *   SAMPLER_F32       sampler on the exact fp32 FMA chain (v_mfma_f32_16x16x4_f32), folded first layer;
 *   SAMPLER_F32_FULL  ... with the full K = 288 first layer on the 48 Pluecker points (no fold);
 *   SAMPLER_SPLIT     the split-fp16 kernel for every ray, also where a workspace is given (single pass: the default of round 2);
 *   BF16              refine handles: bf16 operands (8 significand bits; the round-2 default); NeRF handles: same as DEFAULT;
 *   F16               NeRF handles: fp16 operands (11 significand bits, like the FP16 TensorRT engines of the reference's own fast path,
 *                     trt_infer_v2.py: raw output within the authors' 1e-3 tolerance) — same MFMA count, 5 % slower under the power limit;
 *   BF16_32X32        NeRF handles: the v_mfma_f32_32x32x16_bf16 engine;
 *   REFINE_16X16      refine handles: the fused inference stage (pnrf_refine_fwd, pnrf_refine_project_fwd, pnrf_render_rays_fwd) on
 *                     v_mfma_f32_16x16x32_f16 — the NeRF stage's engine shape, four lanes per ray instead of two; same fp16 operands;
 *   NERF_4X64         NeRF handles: bf16, 4 waves of 64 columns per workgroup (one wave per SIMD) instead of 8 waves of 32 — same arithmetic,
 *                     same packed stream, every weight fragment read from LDS feeds four MFMAs instead of two.
 * A variant is part of a handle's configuration, like its weights: set it right after pack / deserialize, before the handle is given to
 * a context or a stream (the call is not synchronised against launches that use the handle).  Nothing in the library reads the process
 * environment to pick kernels.  Returns PNRF_E_ARG for a variant the handle's net kind does not have. */
#define PNRF_VARIANT_DEFAULT 0
#define PNRF_VARIANT_SAMPLER_F32 1
#define PNRF_VARIANT_SAMPLER_F32_FULL 2
#define PNRF_VARIANT_BF16_32X32 3
#define PNRF_VARIANT_NERF_4X64 4
#define PNRF_VARIANT_SAMPLER_SPLIT 5
#define PNRF_VARIANT_BF16 6
#define PNRF_VARIANT_F16 7
#define PNRF_VARIANT_REFINE_16X16 8
int pnrf_mlp_set_variant(pnrf_mlp_t* h, int variant);
/* Workgroup shape of the fused stages (sampler passes, projection + refine, NeRF) launched from this handle — how a launch's columns (rays or
 * ray samples) are cut into batches and spread over the 256 CUs.  The reference renders any ray count through the same modules
 * (run_S_eS_eN_alter_trt.py:223: `chunk` is accepted and unused); here a whole frame and a 1024-ray chunk want different shapes:
 *   WIDE    one 8-wave workgroup per CU (two waves per SIMD) walking batches of 256 columns (128 rays in the split-fp16 sampler kernel):
 *           every weight fragment streamed into the CU feeds 8 waves.  Whole frames, ray shards.
 *   NARROW  one 4-wave workgroup per CU (a SIMD per wave) on batches of half the width: a call of at most one such batch per CU takes the
 *           latency of one batch through the layers, and a wave that has its SIMD to itself gets through them in 0.7 of the time
 *           (a 1024-ray call: 8 / 16 / 8 / 64 workgroups in the four kernels; 0.148 -> 0.114 ms).
 *   AUTO    (what pack / deserialize produce) NARROW when the launch has at most one narrow batch per CU, else WIDE.
 * Every ray's instruction stream is the same in both shapes: results are bit-identical (tests/test_render_gpu.py).  In both shapes a CU
 * holds at most ONE workgroup of these kernels (registers exclude a second wide one, the LDS request a second narrow one), whatever runs on
 * other streams.  Configuration like the variant: set before the handle is used; the single-kernel test variants (SAMPLER_F32*, BF16_32X32,
 * NERF_4X64) have one shape. */
#define PNRF_SHAPE_AUTO 0
#define PNRF_SHAPE_NARROW 4
#define PNRF_SHAPE_WIDE 8
int pnrf_mlp_set_shape(pnrf_mlp_t* h, int shape);

/* Module-level forward y = net(x), [m, out_dim].  head_act = 0: the raw output of the last Linear
 * (MinMaxRay_Net.forward, DoNeRFTRT.forward); head_act = 1: with the head activations of the TRT
 * wrapper classes applied in place of their slicing ops — sampler: sigmoid on y[0:8] and y[24:27];
 * refine: sigmoid on y[0:8] and y[32:35], tanh on y[8:32] (run_nerf_helpers.py:1502-1505, 1536-1538).
 * x: dev [m, in_dim]; x_views: dev [m, 27] (PNRF_NET_NERF / PNRF_NET_NERFCLS: the view embedding; x is then
 * the 63-wide position embedding), else NULL.
 * Replaces <module>.forward (run_nerf_helpers.py:1490-1507, 1526-1540, 1331-1343). */
int pnrf_mlp_fwd(const pnrf_mlp_t* h, const float* x, const float* x_views, float* y, int64_t m,
                 int head_act, void* stream);

/* ---- element-wise operators --------------------------------------------------------------- */
/* Embedder.embed: out[n, 3+6*n_freq] = [x, sin(2^k x), cos(2^k x)]_k  (run_nerf_helpers.py:666-671). */
int pnrf_posenc_fwd(const float* x, float* out, int64_t n, int n_freq, void* stream);
/* Pluecker.forward: out[n,6] = [d/|d|, o x d/|d|]  (run_nerf_helpers.py:629-632). */
int pnrf_plucker_fwd(const float* o, const float* d, float* out, int64_t n, void* stream);
/* Sampler input: mm_input[n, 6*n_pts] = Pluecker of n_pts points o + t d, t = linspace(0,1,n_pts);
 * rays: dev [n, 11] NDC ray batch (run_S_eS_eN_alter_trt.py:274-277, 546-562). */
int pnrf_ray_encode_fwd(const float* rays, float* mm_input, int64_t n, int n_pts, void* stream);
/* Per-frame ray set-up: get_rays + viewdirs + ndc_rays -> rays[H*W,11] = [o',d',near,far,viewdir],
 * or_rays[H*W,11] = [o,d,or_near,or_far,viewdir].  K (3x3) and c2w (3x4) are HOST pointers.
 * (run_S_eS_eN_alter_trt.py:245-271; run_nerf_helpers.py:2705-2714, 2776-2793). */
int pnrf_frame_rays_fwd(const float* K, const float* c2w, int H, int W, float near, float far,
                        float or_near, float or_far, int64_t first, int64_t count,
                        float* rays, float* or_rays, void* stream);
/* The same for the pixels of a block-cyclic ray partition (multi-GPU frames; pronerf_amd.render.RayPartition): output row q is
 * pixel first + (q / block) * stride + q % block — rank r of `world` ranks dealt blocks of `block` consecutive pixels round-robin passes
 * first = r * block, stride = world * block and count = the number of pixels it owns.  stride = 0 with count <= block is the contiguous
 * range of pnrf_frame_rays_fwd.  Blocks that leave the frame or overlap are refused. */
int pnrf_frame_rays_blocks_fwd(const float* K, const float* c2w, int H, int W, float near, float far,
                               float or_near, float or_far, int64_t first, int64_t block, int64_t stride,
                               int64_t count, float* rays, float* or_rays, void* stream);
/* ndc_rays on an arbitrary ray set: rays_o, rays_d dev [n,3] -> out_o, out_d dev [n,3]
 * (run_nerf_helpers.py:2776-2793). */
int pnrf_ndc_rays_fwd(const float* rays_o, const float* rays_d, int H, int W, float focal, float near,
                      float* out_o, float* out_d, int64_t n, void* stream);
/* inverse_warp_rod1_rt2_coords_trt: img dev [B,3,Hf,Wf]; depth dev [B,n]; ro1,rd1 dev [4,n] per
 * batch entry with batch stride ray_bstride floats (0 = shared by all B, the reference's
 * expand()); w2c dev [B,3,4]; out dev [B,3,n].
 * Bilinear, zero padding, align_corners=True (inverse_warp.py:584-619). */
int pnrf_warp_trt_fwd(const float* img, const float* depth, const float* ro1, const float* rd1,
                      int64_t ray_bstride, const float* w2c, float* out, int B, int Hf, int Wf,
                      int64_t n, void* stream);
/* inverse_warp_rod1_rt2_coords (training variant): c2 = R^T w - R^T t, c2 /= |c2.z|+1e-8, c2.z = 1, c2.y = -c2.y,
 * p = K c2; samples whose normalised X or Y leaves [-1,1] give 0.  img dev [B,3,Hf,Wf]; depth dev [B,n]; ro1, rd1 dev
 * [3,n] per batch entry (batch stride ray_bstride floats, 0 = shared); c2w2 dev [B,3,4]; K dev [B,3,3]; out dev
 * [B,3,n]  (inverse_warp.py:515-581). */
int pnrf_warp_train_fwd(const float* img, const float* depth, const float* ro1, const float* rd1,
                        int64_t ray_bstride, const float* c2w2, const float* K, float* out, int B, int Hf,
                        int Wf, int64_t n, void* stream);
/* Neighbour images [nv,3,Hf,Wf] -> texel-interleaved [nv,Hf,Wf,4] used by the fused projection
 * (replaces the x8 image replication of run_S_eS_eN_alter_trt.py:296-298). */
int pnrf_images_pack(const float* img_nchw, float* out_nhwc4, int nv, int Hf, int Wf, void* stream);
/* Projection + sample Pluecker: refine_in[n, 48 + 24 nb] (nb = 4: [n,144]) = [pluecker(8 samples)(48), epi(24 nb)], nb = 1 .. 8, with
 * epi index (k*8+s)*3+c  (run_S_eS_eN_alter_trt.py:637-661).  rays,or_rays dev [n,11];
 * depth_sorted dev [n,8]; img4 dev [nb,Hf,Wf,4]; proj dev [nb,3,4]; eps = 1e-5 (stage 1: 1e-6). */
int pnrf_refine_input_fwd(const float* rays, const float* or_rays, const float* depth_sorted,
                          const float* img4, const float* proj, int nb, int Hf, int Wf, float eps,
                          float* refine_in, int64_t n, void* stream);
/* Training-time refine_in[n,144]: like pnrf_refine_input_fwd but with per-ray source views ref_nos dev [n,4] (int64,
 * indices into the nv training views img4 dev [nv,Hf,Wf,4] / poses dev [nv,3,4] camera-to-world), the training
 * projection (pnrf_warp_train_fwd) and the valid-mask mean fill: valid = (sum_c rgb > 0), invalid (view, sample)
 * entries are replaced by the mean over the valid views of that sample.  K dev [3,3].  layout 0: epi index
 * (k*8+s)*3+c (stage 2); 1: s*12+k*3+c (stage 1).
 * (run_S_eS_eN_alter_base_refine2.py:570-634; run_S_eS_eN_alter_base.py:607-673) */
int pnrf_refine_input_train_fwd(const float* rays, const float* or_rays, const float* depth_sorted,
                                const float* img4, const float* poses, const float* K, const int64_t* ref_nos,
                                int nv, int nb, int Hf, int Wf, float eps, int layout, float* refine_in,
                                int64_t n, void* stream);
/* raw2outputs (infer variant; clamp/noise/white_bkgd select the training variants):
 * raw dev [n,s,4]; z dev [n,s]; rays_d dev [n,3] with row stride d_stride floats; add,mul dev
 * [n,s] or NULL; noise dev [n,s] or NULL.  Outputs (any may be NULL): rgb[n,3], disp[n],
 * acc[n], weights[n,s], depth[n]  (run_S_eS_eN_alter_trt.py:564-597). */
int pnrf_composite_fwd(const float* raw, const float* z, const float* rays_d, int d_stride,
                       const float* add, const float* mul, const float* noise, float clamp,
                       int white_bkgd, float* rgb, float* disp, float* acc, float* weights,
                       float* depth, int64_t n, int s, void* stream);

/* ---- fused stages ---------------------------------------------------------------------------
 * Sampler: Pluecker ray encoding -> MLP -> sigmoid, depth affine, stable ascending sort of the 8 depths,
 * permutation of add/mul.  Arithmetic of a DEFAULT handle: every layer product in SPLIT fp16 (hi + lo planes, three
 * v_mfma_f32_16x16x32_f16 per product, fp32 accumulation: fp32-grade, 22 significand bits per operand), every ray in one pass —
 * the exact-index form; SAMPLER_F32 / SAMPLER_F32_FULL handles run the exact fp32 FMA chain (v_mfma_f32_16x16x4_f32) instead.
 * (The frame path pnrf_render_rays_fwd runs pnrf_sampler_fwd_ws below: a plain-fp16 first pass + this kernel on the undecided
 * rays.)  rays dev [n,11].  Outputs dev:
 * depth_sorted[n,8], add_sorted[n,8], mul_sorted[n,8]; optional (NULL to skip) sort_idx[n,8]
 * (int64, the "sampler indices"), mm_rgb[n,3], depth_raw[n,8] (sigmoid output before the sort).
 * (run_S_eS_eN_alter_trt.py:628-635) */
int pnrf_sampler_fwd(const pnrf_mlp_t* h, const float* rays, int64_t n, float* depth_sorted,
                     float* add_sorted, float* mul_sorted, int64_t* sort_idx, float* mm_rgb,
                     float* depth_raw, void* stream);
/* The same operator in two passes (what pnrf_render_rays_fwd runs).  The eight depths of a ray are sorted, so the products must be
 * fp32-grade wherever two of them are close — and only there.  Pass 1 renders every ray in plain fp16 (one v_mfma_f32_32x32x16_f16 per
 * product, a third of the split kernel's MFMA work) and carries, per ray, a bound s_k on the standard deviation of its own rounding error
 * in depth k (variance propagation through the layers from |x_l|^2 and the weights' column norms); a ray is "undecided" when
 * some adjacent sorted gap is not larger than kappa (s_i + s_i+1) + 2e-6 (far - near), when a depth is not finite, or when one of its
 * activations reached the fp16 limit.  Pass 2 renders the undecided rays with the split-fp16 kernel of pnrf_sampler_fwd and overwrites
 * their rows.  Pass 3 (a few workgroups that leave at once when there is nothing to do) renders with the exact-fp32 kernel
 * (PNRF_VARIANT_SAMPLER_F32's) the rays of pass 2 in which a hidden activation saturated at 65 504 — fp16 range is DEFINED: every kernel
 * runs with MODE.FP16_OVFL (saturation instead of inf), a saturated ray is detected and re-rendered in fp32, nothing becomes NaN.
 * Guarantee: the depths / add / mul of the rays of passes 2 and 3 are identical to pnrf_sampler_fwd's (split fp16 resp. exact fp32:
 * fp32-grade); those of the other rays are fp16-grade (measured <= 6e-4 (far - near), tested <= 2e-3) and their SORT INDICES equal
 * pnrf_sampler_fwd's under the error model — a statistical bound (fp16 roundings treated as independent zero-mean errors; subnormals and
 * fp32 accumulation error are covered by the 2e-6 allowance), not a proof.  kappa = 2 (round 5; 4 before): the bound s is >= 2.7 sigma of the
 * measured error where it is tightest (largest |error| / s over 524 288 depths: 0.8 .. 1.9 on seven weight sets, tools/sampler_twopass_model.py),
 * so the threshold is >= 7.6 sigma of the difference of two errors; it is 2x the smallest kappa that was ever clean and >= 4x the first that
 * was not (0 index mismatches on 762 048 rays x synthetic, heavy-tailed, x4-scaled and optimizer-trained weights down to kappa = 1; one ray
 * of 3 M at 0.5: tools/kappa_scan.py, tests/test_fullframe_gpu.py; 0 of 45.7 M rays of 60 further weight sets at kappa = 2 and at 1:
 * tools/kappa_population.py).  It halves the second pass (7.5 % of the bench frame's rays instead of
 * 14.6 %).  Where exactness matters more than 0.5 ms per frame, PNRF_VARIANT_SAMPLER_SPLIT renders every ray fp32-grade;
 * pnrf_ctx_set_sampler_kappa restores a wider margin per context.
 * kappa < 0 selects PNRF_SAMPLER_KAPPA; kappa = 0 leaves only the fp32 round-off allowance (tests); NaN and values >= 1e30 are refused.
 * workspace: dev, 16-byte aligned, >= pnrf_sampler_workspace_bytes(n) bytes, contents irrelevant (its counters are reset on the stream
 * by every call); concurrent calls need separate workspaces.  Handles set to SAMPLER_SPLIT run the split kernel for every ray + pass 3;
 * SAMPLER_F32* handles run their single kernel. */
#define PNRF_SAMPLER_KAPPA 2.0f
int64_t pnrf_sampler_workspace_bytes(int64_t n);
int pnrf_sampler_fwd_ws(const pnrf_mlp_t* h, const float* rays, int64_t n, float* depth_sorted, float* add_sorted,
                        float* mul_sorted, int64_t* sort_idx, float* mm_rgb, float* depth_raw, void* workspace,
                        int64_t workspace_bytes, float kappa, void* stream);
/* Refine: MLP on refine_in[n, 48 + 24 nb] (the handle's num_neighbor; Fern: [n,144]) -> sigmoid/tanh -> interval refinement -> query points.  Arithmetic of a DEFAULT handle: fp16
 * operands (v_mfma_f32_32x32x16_f16, 11 significand bits, MODE.FP16_OVFL saturation), fp32 accumulation; a handle set to
 * PNRF_VARIANT_BF16 runs bf16 operands (8 bits; the round-2 default).
 * Outputs dev: z[n,8], pts[n,8,3]  (run_S_eS_eN_alter_trt.py:668-681). */
int pnrf_refine_fwd(const pnrf_mlp_t* h, const float* refine_in, const float* rays,
                    const float* depth_sorted, float* z, float* pts, int64_t n, void* stream);
/* Projection + refine in ONE kernel: what pnrf_refine_input_fwd followed by pnrf_refine_fwd compute, without the refine_in [n,144]
 * round trip through HBM.  NOT bit for bit: the colours go straight into the MFMA operands (fp16 on a DEFAULT handle, bf16 on a PNRF_VARIANT_BF16 one), so the head uses a linearised projection
 * p(z) = A + z B per (ray, view) with FMA contraction and v_rcp_f32 for 1 / (1 - d - eps) and 1 / p.z, and skips grid_sample's normalise /
 * un-normalise round trip: pixel coordinates move by a few fp32 ulps (~1e-4 px), colours by ~1e-4 of the local texel difference, well
 * below the operand rounding applied next (fp16: 5e-4 relative) (tests/test_ops_gpu.py bounds z within 2e-3 of the two-kernel path).  The operator that replays
 * the reference's fp32 projection sequence exactly is pnrf_refine_input_fwd.  Here every workgroup projects the samples of its 256 rays into the four neighbour views, fetches
 * the colours and encodes the sample Pluecker values in the head of its batch, straight into the MFMA operand registers.
 * rays, or_rays dev [n,11]; depth_sorted dev [n,8]; img4 dev [nb,Hf,Wf,4] (pnrf_images_pack); proj dev [nb,3,4]; eps as pnrf_refine_input_fwd;
 * nb must be the handle's num_neighbor (1 .. 8; 4 in the Fern configs: a lane half then projects two views, ceil(nb / 2) in general).
 * This is the refine stage of pnrf_render_rays_fwd.  (run_S_eS_eN_alter_trt.py:637-681; inverse_warp.py:584-619) */
int pnrf_refine_project_fwd(const pnrf_mlp_t* h, const float* rays, const float* or_rays, const float* depth_sorted,
                            const float* img4, const float* proj, int nb, int Hf, int Wf, float eps, float* z,
                            float* pts, int64_t n, void* stream);
/* Training-time refine stage (stage 2): as pnrf_refine_fwd, plus the depth jitter of refine2.py:646-662 —
 * jitter dev [n,8] = min(|N(0,1)|/5, 1-2e-6) or NULL, jitter_dir +1 (toward the next refined sample / far) or -1 (toward
 * the previous / near): z += dir * jitter * |z - neighbour| — and the refine rgb head rgb0 dev [n,3] =
 * sigmoid(y[32:35]) (rgb_map0) or NULL.  (run_S_eS_eN_alter_base_refine2.py:635-668) */
int pnrf_refine_train_fwd(const pnrf_mlp_t* h, const float* refine_in, const float* rays,
                          const float* depth_sorted, const float* jitter, int jitter_dir, float* z, float* pts,
                          float* rgb0, int64_t n, void* stream);
/* NeRF: positional encoding of pts/viewdirs -> MLP (DEFAULT handle: bf16 operands on v_mfma_f32_16x16x32_bf16, fp32 accumulation;
 * PNRF_VARIANT_F16: fp16 operands) -> alpha compositing with the sampler's
 * density modulation.  pts dev [n,8,3]; rays dev [n,11]; z, add_sorted, mul_sorted dev [n,8].
 * Outputs dev: rgbd[n,4] = (r,g,b,depth); raw[n,8,4] optional (NULL to skip).  h may be a PNRF_NET_NERF or a
 * PNRF_NET_NERFCLS handle (the fine-net class mismatch of the released scripts).
 * (run_S_eS_eN_alter_trt.py:691-694; run_network :195-208) */
int pnrf_nerf_fwd(const pnrf_mlp_t* h, const float* pts, const float* rays, const float* z,
                  const float* add_sorted, const float* mul_sorted, float* rgbd, float* raw,
                  int64_t n, void* stream);

/* Training-time NeRF stage: as pnrf_nerf_fwd, plus noise dev [n,S] (= randn * raw_noise_std, added to sigma before
 * the density modulation) or NULL; clamp > 0 clamps raw to +-clamp first (stage 1: 10); white_bkgd (rgb += 1 - acc);
 * add_sorted/mul_sorted may both be NULL (stage-1 odd steps composite without them); S samples per ray: pts dev
 * [n,S,3].  S == 8: fused compositing into rgbd (raw optional).  S != 8 (stage-1 exploration, 16..256): rgbd must be
 * NULL and raw dev [n,S,4] is written; composite it with pnrf_composite_fwd.
 * (run_S_eS_eN_alter_base_refine2.py:497-520, 669-676; run_S_eS_eN_alter_base.py:523, 731-751) */
int pnrf_nerf_train_fwd(const pnrf_mlp_t* h, const float* pts, const float* rays, const float* z,
                        const float* add_sorted, const float* mul_sorted, const float* noise, float clamp,
                        int white_bkgd, int S, float* rgbd, float* raw, int64_t n, void* stream);
/* Stage-1 exploration of the refined depths (run_S_eS_eN_alter_base.py:689-729): every one of the 8 depths z8 dev
 * [n,8] is replicated n_mult times toward the next (dir1 = +1) or previous (-1) sample with offsets
 * linspace(0, 1-1/n_mult, n_mult)*|gap|, the 8*n_mult values are sorted, then jittered: z += dir2 * jitter * |z -
 * neighbour| with jitter dev [n, 8*n_mult] = min(|N(0,1)|/5, 0.99).  Outputs dev: z_out [n, 8*n_mult] and the query
 * points pts_out [n, 8*n_mult, 3] = o + d*z (no learned offsets on these steps).  n_mult 1..32. */
int pnrf_explore_fwd(const float* z8, const float* rays, const float* jitter, int n_mult, int dir1, int dir2,
                     float* z_out, float* pts_out, int64_t n, void* stream);

/* ---- whole path: render_rays (inference) ----------------------------------------------------
 * A context owns the per-ray workspace for up to max_rays rays (allocated once). */
int pnrf_ctx_create(const pnrf_mlp_t* sampler, const pnrf_mlp_t* refine, const pnrf_mlp_t* nerf,
                    int64_t max_rays, pnrf_ctx_t** out);
int pnrf_ctx_free(pnrf_ctx_t* ctx);
/* render_rays(ray_batch, or_ray_batch, ...) -> rgbd[n,4] = (rgb_map, depth_map); sort_idx
 * optional.  img4 dev [nb,Hf,Wf,4] (pnrf_images_pack), proj dev [nb,3,4].
 * (run_S_eS_eN_alter_trt.py:599-696) */
int pnrf_render_rays_fwd(pnrf_ctx_t* ctx, const float* rays, const float* or_rays,
                         const float* img4, const float* proj, int nb, int Hf, int Wf, float eps,
                         float* rgbd, int64_t* sort_idx, int64_t n, void* stream);
/* Threshold of the two-pass sampler for this context's calls (pnrf_sampler_fwd_ws's kappa): negative = PNRF_SAMPLER_KAPPA (what a new
 * context has), 0 = only the fp32 round-off allowance; NaN and values >= 1e30 are refused.  A larger kappa sends more rays through the
 * fp32-grade second pass; PNRF_VARIANT_SAMPLER_SPLIT on the sampler handle sends all of them (the exact path). */
int pnrf_ctx_set_sampler_kappa(pnrf_ctx_t* ctx, float kappa);
/* The kappa this context's calls run with (PNRF_SAMPLER_KAPPA unless pnrf_ctx_set_sampler_kappa chose another): what bench.py reports. */
int pnrf_ctx_get_sampler_kappa(const pnrf_ctx_t* ctx, float* kappa);
/* Rays the sampler's second pass rendered in the context's most recent pnrf_render_rays_fwd (waits for the device; diagnostics). */
int pnrf_ctx_sampler_stats(pnrf_ctx_t* ctx, int64_t* rays_second_pass);
/* Rays whose hidden activations reached the fp16 limit in the split-fp16 kernel and were therefore rendered by the exact-fp32 kernel (the
 * third pass) in the context's most recent pnrf_render_rays_fwd: 0 for every net whose activations stay below 65 504 / log2(e) (waits for the
 * device; diagnostics). */
int pnrf_ctx_sampler_saturated(pnrf_ctx_t* ctx, int64_t* rays_third_pass);
/* Per-stage device time of pnrf_render_rays_fwd (what the reference gets from line_profiler / the cuda events around
 * render(), run_S_eS_eN_alter_trt.py:327-332, at frame granularity): after _begin, the next max_frames calls on this
 * context record an event before and after each of the three kernels on the caller's stream; _end waits for the last
 * recorded call and returns the mean milliseconds ms[3] = {sampler, projection + refine, NeRF + compositing}
 * over *frames calls.  Recording costs four event records per call and changes no result. */
int pnrf_ctx_profile_begin(pnrf_ctx_t* ctx, int max_frames);
int pnrf_ctx_profile_end(pnrf_ctx_t* ctx, float* ms, int* frames);

/* ---- stage-2 training step --------------------------------------------------
 * fp32 storage and accumulation throughout, like the reference's training.  Every stage is a kernel of this library: the layer
 * products carry bias + activation (forward) and the activation derivative of the layer below (backward) in their epilogues and
 * multiply either in split fp16 (default: both operands as hi + 2^-11 lo fp16 pairs, 22 significand bits, three fp16 MFMAs per
 * block, gradients scaled by a power of two from their recorded maximum) or in exact fp32 on v_mfma_f32_16x16x4_f32
 * (pnrf_trainer_set_products).  From 8192 sample rows on the fine net runs as two launches on the fused-MLP engine of the inference
 * path — forward pts0 .. feature, views with every activation saved once; backward the ten input-gradient products with per-row
 * scaled fp16 planes — and the 256-wide weight gradients of all three nets as one grouped launch per iteration; below that, and
 * for the 4096-row sampler / refine nets, a layer's products are launches or 16-row layer chains.  The 1-4 wide heads, sort,
 * interval refinement, jitter, encodings, compositing, losses and Adam are per-ray / streaming kernels. */

/* Operator-level backward passes (each mirrors what torch.autograd derives for the forward it names). */
/* raw2outputs backward for d rgb_map [n,3]: arguments as pnrf_composite_fwd; outputs d_raw dev [n,s,4], d_z dev [n,s] (NULL to
 * skip), d_add / d_mul dev [n,s] (NULL to skip).  Any s >= 1 (a wave per ray, a lane per sample, two recurrences in sample order).
 * (run_S_eS_eN_alter_base_refine2.py:475-522) */
int pnrf_composite_bwd(const float* raw, const float* z, const float* rays_d, int d_stride, const float* add,
                       const float* mul, const float* noise, float clamp, int white_bkgd, const float* d_rgb,
                       float* d_raw, float* d_z, float* d_add, float* d_mul, int64_t n, int s, void* stream);
/* Embedder.embed backward: d_x[n,3] from d_out[n, 3+6*n_freq] (run_nerf_helpers.py:666-671). */
int pnrf_posenc_bwd(const float* x, const float* d_out, float* d_x, int64_t n, int n_freq, void* stream);
/* Sampler head on the raw output y dev [n,27] of MinMaxRay_Net: depth = sigmoid(y[0:8]) (far-near)+near, stable ascending
 * sort, add / mul gathered by the sort indices, mm_rgb = sigmoid(y[24:27]) (NULL to skip)
 * (run_S_eS_eN_alter_base_refine2.py:557-568); the backward scatters through sort_idx. */
int pnrf_sampler_head_fwd(const float* y, const float* rays, float* depth_sorted, int64_t* sort_idx, float* add_sorted,
                          float* mul_sorted, float* mm_rgb, int64_t n, void* stream);
int pnrf_sampler_head_bwd(const float* y, const float* rays, const int64_t* sort_idx, const float* d_depth_sorted,
                          const float* d_add_sorted, const float* d_mul_sorted, const float* d_mm_rgb, float* d_y,
                          int64_t n, void* stream);
/* Refine head on the raw output y dev [n,35]: refine = sigmoid(y[0:8]), offsets = tanh(y[8:32]), rgb0 = sigmoid(y[32:35])
 * (NULL to skip); interval refinement, depth jitter (jitter dev [n,8] or NULL, jitter_dir +-1), query points
 * pts dev [n,8,3] = o + d z + 0.01 offsets; z_pre dev [n,8] = depths before the jitter (input of the backward)
 * (run_S_eS_eN_alter_base_refine2.py:635-668).  Backward: d_pts dev [n,8,3], d_z dev [n,8] or NULL, d_rgb0 dev [n,3] or
 * NULL -> d_y dev [n,35], d_depth_sorted dev [n,8]. */
int pnrf_refine_head_fwd(const float* y, const float* rays, const float* depth_sorted, const float* jitter,
                         int jitter_dir, float* z_pre, float* z, float* pts, float* rgb0, int64_t n, void* stream);
int pnrf_refine_head_bwd(const float* y, const float* rays, const float* depth_sorted, const float* z_pre,
                         const float* jitter, int jitter_dir, const float* d_pts, const float* d_z,
                         const float* d_rgb0, float* d_y, float* d_depth_sorted, int64_t n, void* stream);

/* Trainer: fp32 parameters, gradients, Adam moments and workspaces of the three networks for batches of up to max_rays
 * rays.  26 Linear layers in this order: sampler fc_backbone.0..5, fc_output (MinMaxRay_Net, 288 -> 27); refine net
 * likewise (144 -> 35); fine net of class NeRF: pts_linears.0..7, feature_linear, alpha_linear, views_linears.0, rgb_linear.
 * W[i]: [out_dim[i], in_dim[i]] row-major (torch layout), host or device.  max_samples: samples per ray the NeRF-side
 * workspaces are sized for (8; stage-1 exploration: up to 256).
 * Replaces create_nerf's modules + torch.optim.Adam (run_S_eS_eN_alter_base_refine2.py:337-395). */
int pnrf_trainer_create(const float* const* W, const float* const* b, const int* in_dim, const int* out_dim,
                        int n_layers, int64_t max_rays, int max_samples, pnrf_trainer_t** out);
int pnrf_trainer_free(pnrf_trainer_t* t);
/* Weight-gradient kernel of the square layers: tile 0 = chosen by shape and row count (default), 64 / 128 = forced where the shape allows;
 * min_rows_128 = row count from which the 128 x 128-tile kernel is used (0 = default).  tile 256 / 255 (round 5) leave that choice alone and switch
 * the 256 x 128-tile form of the grouped split-fp16 gradients on from min_rows_128 rows (0: from the first row) / off; default: from 32 768 rows.
 * The two choices are independent: tile 0 / 64 / 128 do not touch the wide-tile threshold either.  Configuration, not on the step path. */
int pnrf_trainer_set_dw_kernel(pnrf_trainer_t* t, int tile, int64_t min_rows_128);
/* Diagnostics of the most recent iteration's grouped weight-gradient launch: number of gradients it held and which of them (bit k = k-th) ran on
 * the 256 x 128 tiles.  Host state; lets a test assert that the tile shape it forced is the one that ran. */
int pnrf_trainer_dw_group_info(const pnrf_trainer_t* t, int* n_jobs, unsigned* wide_mask);

/* kind 0 parameters, 1 gradients, 2 / 3 Adam first / second moment of the joint optimizer, 4 / 5 those of the NeRF-only
 * optimizer; W, b: host or device (NULL to skip). */
int pnrf_trainer_read(const pnrf_trainer_t* t, int kind, int layer, float* W, float* b, void* stream);
int pnrf_trainer_write(pnrf_trainer_t* t, int kind, int layer, const float* W, const float* b, void* stream);
int pnrf_trainer_set_step(pnrf_trainer_t* t, int64_t step, int64_t step_nerf);
/* Device address / element count of a flat array (kind as above): layers contiguous in trainer order, each [W, b] padded to 4
 * floats.  For in-place collectives over data-parallel replicas (RCCL all-reduce of the gradients). */
int pnrf_trainer_flat(pnrf_trainer_t* t, int kind, float** ptr, int64_t* count);

typedef struct pnrf_train_batch {
  const float* rays;       /* dev [n,11] NDC ray batch (o, d, near, far, viewdir) */
  const float* or_rays;    /* dev [n,11] world-space rays */
  const float* target;     /* dev [n,3] ground-truth colours */
  const float* img4;       /* dev [nv,Hf,Wf,4] training views (pnrf_images_pack) */
  const float* poses;      /* dev [nv,3,4] camera-to-world */
  const float* K;          /* dev [3,3] */
  const int64_t* ref_nos;  /* dev [n,4] source views of each ray (the reference's random neighbour draw, made explicit) */
  const float* jitter;     /* dev [n,8] = min(|N(0,1)|/5, 1-2e-6) or NULL */
  const float* raw_noise;  /* dev [n,8] = randn * raw_noise_std or NULL */
  int64_t n;
  int nv, Hf, Wf;
  int jitter_dir;          /* +1 toward the next sample / far, -1 toward the previous / near */
  int white_bkgd;
  float eps;               /* NDC -> metric epsilon: 1e-5 */
  float a_mmrgb;           /* weight of mse(rgb_map0) + mse(mm_rgb) in the loss (0 in fern_refine.txt; 1 on stage-1 even iterations) */
  float clamp;             /* raw clamp before compositing: 0 = none (stage 2), 10 (stage 1, base.py:523) */
  int layout;              /* epi feature layout: 0 neighbour-major (stage 2), 1 sample-major (stage 1) */
} pnrf_train_batch_t;

/* Joint iteration: render_rays (training) + img2mse [+ a_mmrgb (...)] + loss.backward(): leaves the gradients of all 26
 * layers in the trainer.  loss dev [4] = {total, mse(rgb_map1), mse(rgb_map0), mse(mm_rgb)}; rgb_map1 dev [n,3] or NULL.
 * Stage 2 (run_S_eS_eN_alter_base_refine2.py:525-680, 858-868) and, with layout 1 / eps 1e-6 / clamp 10 / a_mmrgb 1 / no
 * jitter and noise, the even iterations of stage 1 (run_S_eS_eN_alter_base.py:554-761, 941-958). */
int pnrf_train_stage2_fwd_bwd(pnrf_trainer_t* t, const pnrf_train_batch_t* batch, float* loss, float* rgb_map1,
                              void* stream);
/* Odd iterations of stage 1 (run_S_eS_eN_alter_base.py:929-940, exploration :689-729): sampler / refine nets without
 * gradient, refined depths explored into 8 n_mult samples (replicated toward dir1, jittered toward batch->jitter_dir with
 * batch->jitter dev [n, 8 n_mult]), no offsets, compositing without add / mul with batch->raw_noise dev [n, 8 n_mult];
 * loss = img2mse(rgb_map1); gradients for the 12 NeRF layers only. */
int pnrf_train_explore_fwd_bwd(pnrf_trainer_t* t, const pnrf_train_batch_t* batch, int n_mult, int dir1, float* loss,
                               float* rgb_map1, void* stream);
/* Both iteration entry points copy the batch into buffers of the trainer with one launch; with pnrf_trainer_set_graph(t, 1) they then replay
 * their ~100-kernel launch sequence as a hipGraph (captured from the given stream — the legacy default stream is served through a stream of
 * the trainer's own — the first time a configuration (ray count, views, flags, stream) is seen; up to 8 configurations are kept).  Default
 * 0: the kernels are launched one by one (same results bit for bit; measured slightly faster). */
int pnrf_trainer_set_graph(pnrf_trainer_t* t, int enable);
/* Arithmetic of the layer products X W^T and dZ W.  kind 0 (default): split-fp16 MFMA — both operands as hi + 2^-11 lo fp16 pairs (22
   significand bits), three fp16 MFMAs per product block, fp32 accumulation; gradients are scaled by a power of two taken from their
   recorded maximum before the split.  kind 1: exact-fp32 MFMA products everywhere (the reference trains in fp32; torch's fp32 GEMMs are
   this).  From 8192 rows on kind 0 runs the fine net on the fused-MLP engine of the inference path: forward (pts0 .. pts7, feature_linear,
   views_linear) and backward (their input gradients) as ONE launch each — 128 rows per workgroup stay in registers through the layers, the
   weights stream through LDS once per 128 rows, each activation / gradient is written once, ReLU derivatives travel as bit masks — and the
   weight gradients as one grouped launch (same split-fp16 arithmetic, the engine's contraction order and per-row gradient scales: fp32
   round-off apart from the per-layer products).  kind 2: one product launch per layer; kind 3: pts1-4 and pts6, pts7, feature forward as two
   64-row layer chains (kinds 2 and 3: bit-identical to each other; kept for A/B timing and tests).  The narrow heads (fewer than 64 output
   columns) always use the fp32 kernel. */
int pnrf_trainer_set_products(pnrf_trainer_t* t, int kind);
/* optimizer.step() of torch.optim.Adam (L2 weight decay added to the gradient).  which 0: the joint optimizer over all
 * parameters (run_S_eS_eN_alter_base_refine2.py:394, 869; stage 1 s_optimizer); which 1: the NeRF-only optimizer of stage 1
 * (run_S_eS_eN_alter_base.py:398-421, 940) with its own moments and step count. */
int pnrf_trainer_adam_step(pnrf_trainer_t* t, int which, float lr, float beta1, float beta2, float eps,
                           float weight_decay, void* stream);

/* Host helper: torch.linspace(start,end,n) in fp32, as used for the 48 ray points
 * (run_S_eS_eN_alter_trt.py:556-557).  out: HOST [n]. */
int pnrf_linspace(float start, float end, int n, float* out);

#ifdef __cplusplus
}
#endif
#endif /* PRONERF_HIP_H */
