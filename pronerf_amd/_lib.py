"""ctypes binding of libpronerf_hip.so (include/pronerf_hip.h).

There is no CPU fallback: if the library is missing or a call fails, a ``PnrfError`` is
raised — the product path never silently degrades to eager PyTorch.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'libpronerf_hip.so')

NET_SAMPLER, NET_REFINE, NET_NERF, NET_NERFCLS = 0, 1, 2, 3
# kernel variants of a packed handle (pnrf_mlp_set_variant)
VARIANTS = {'default': 0, 'sampler_f32': 1, 'sampler_f32_full': 2, 'bf16_32x32': 3, 'nerf_4x64': 4, 'sampler_split': 5, 'bf16': 6, 'f16': 7, 'refine_16x16': 8}
ABI_VERSION = 1


class PnrfError(RuntimeError):
    pass


_p = C.c_void_p
_i64 = C.c_int64
_f = C.c_float
_i = C.c_int

# name -> (restype, argtypes): every symbol include/pronerf_hip.h declares
SIGNATURES = {
    'pnrf_abi_version': (_i, []),
    'pnrf_last_error': (C.c_char_p, []),
    'pnrf_mlp_pack': (_i, [_i, C.POINTER(_p), C.POINTER(_p), C.POINTER(_i), C.POINTER(_i), _i, C.POINTER(_p)]),
    'pnrf_mlp_free': (_i, [_p]),
    'pnrf_mlp_serialize': (_i, [_p, _p, _i64, C.POINTER(_i64)]),
    'pnrf_mlp_deserialize': (_i, [_p, _i64, C.POINTER(_p)]),
    'pnrf_mlp_kind': (_i, [_p, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    'pnrf_mlp_set_variant': (_i, [_p, _i]),
    'pnrf_mlp_set_shape': (_i, [_p, _i]),
    'pnrf_mlp_fwd': (_i, [_p, _p, _p, _p, _i64, _i, _p]),
    'pnrf_posenc_fwd': (_i, [_p, _p, _i64, _i, _p]),
    'pnrf_plucker_fwd': (_i, [_p, _p, _p, _i64, _p]),
    'pnrf_ray_encode_fwd': (_i, [_p, _p, _i64, _i, _p]),
    'pnrf_frame_rays_fwd': (_i, [C.POINTER(_f), C.POINTER(_f), _i, _i, _f, _f, _f, _f, _i64, _i64, _p, _p, _p]),
    'pnrf_frame_rays_blocks_fwd': (_i, [C.POINTER(_f), C.POINTER(_f), _i, _i, _f, _f, _f, _f, _i64, _i64, _i64, _i64, _p, _p, _p]),
    'pnrf_ndc_rays_fwd': (_i, [_p, _p, _i, _i, _f, _f, _p, _p, _i64, _p]),
    'pnrf_warp_trt_fwd': (_i, [_p, _p, _p, _p, _i64, _p, _p, _i, _i, _i, _i64, _p]),
    'pnrf_warp_train_fwd': (_i, [_p, _p, _p, _p, _i64, _p, _p, _p, _i, _i, _i, _i64, _p]),
    'pnrf_refine_input_train_fwd': (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _i, _p, _i64, _p]),
    'pnrf_refine_train_fwd': (_i, [_p, _p, _p, _p, _p, _i, _p, _p, _p, _i64, _p]),
    'pnrf_nerf_train_fwd': (_i, [_p, _p, _p, _p, _p, _p, _p, _f, _i, _i, _p, _p, _i64, _p]),
    'pnrf_explore_fwd': (_i, [_p, _p, _p, _i, _i, _i, _p, _p, _i64, _p]),
    'pnrf_images_pack': (_i, [_p, _p, _i, _i, _i, _p]),
    'pnrf_refine_input_fwd': (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _f, _p, _i64, _p]),
    'pnrf_composite_fwd': (_i, [_p, _p, _p, _i, _p, _p, _p, _f, _i, _p, _p, _p, _p, _p, _i64, _i, _p]),
    'pnrf_sampler_fwd': (_i, [_p, _p, _i64, _p, _p, _p, _p, _p, _p, _p]),
    'pnrf_sampler_workspace_bytes': (_i64, [_i64]),
    'pnrf_sampler_fwd_ws': (_i, [_p, _p, _i64, _p, _p, _p, _p, _p, _p, _p, _i64, _f, _p]),
    'pnrf_refine_fwd': (_i, [_p, _p, _p, _p, _p, _p, _i64, _p]),
    'pnrf_refine_project_fwd': (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _p, _p, _i64, _p]),
    'pnrf_nerf_fwd': (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _p]),
    'pnrf_ctx_create': (_i, [_p, _p, _p, _i64, C.POINTER(_p)]),
    'pnrf_ctx_free': (_i, [_p]),
    'pnrf_render_rays_fwd': (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _f, _p, _p, _i64, _p]),
    'pnrf_ctx_sampler_stats': (_i, [_p, C.POINTER(_i64)]),
    'pnrf_ctx_set_sampler_kappa': (_i, [_p, _f]),
    'pnrf_ctx_get_sampler_kappa': (_i, [_p, C.POINTER(_f)]),
    'pnrf_ctx_sampler_saturated': (_i, [_p, C.POINTER(_i64)]),
    'pnrf_ctx_profile_begin': (_i, [_p, _i]),
    'pnrf_ctx_profile_end': (_i, [_p, C.POINTER(C.c_float), C.POINTER(_i)]),
    'pnrf_linspace': (_i, [_f, _f, _i, C.POINTER(_f)]),
    # stage-2 training step
    'pnrf_composite_bwd': (_i, [_p, _p, _p, _i, _p, _p, _p, _f, _i, _p, _p, _p, _p, _p, _i64, _i, _p]),
    'pnrf_posenc_bwd': (_i, [_p, _p, _p, _i64, _i, _p]),
    'pnrf_sampler_head_fwd': (_i, [_p, _p, _p, _p, _p, _p, _p, _i64, _p]),
    'pnrf_sampler_head_bwd': (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _p]),
    'pnrf_refine_head_fwd': (_i, [_p, _p, _p, _p, _i, _p, _p, _p, _p, _i64, _p]),
    'pnrf_refine_head_bwd': (_i, [_p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _i64, _p]),
    'pnrf_trainer_create': (_i, [C.POINTER(_p), C.POINTER(_p), C.POINTER(_i), C.POINTER(_i), _i, _i64, _i, C.POINTER(_p)]),
    'pnrf_trainer_free': (_i, [_p]),
    'pnrf_trainer_read': (_i, [_p, _i, _i, _p, _p, _p]),
    'pnrf_trainer_write': (_i, [_p, _i, _i, _p, _p, _p]),
    'pnrf_trainer_set_step': (_i, [_p, _i64, _i64]),
    'pnrf_trainer_set_dw_kernel': (_i, [_p, _i, _i64]),
    'pnrf_trainer_dw_group_info': (_i, [_p, C.POINTER(_i), C.POINTER(C.c_uint)]),
    'pnrf_trainer_set_graph': (_i, [_p, _i]),
    'pnrf_trainer_set_products': (_i, [_p, _i]),
    'pnrf_trainer_flat': (_i, [_p, _i, C.POINTER(_p), C.POINTER(_i64)]),
    'pnrf_train_stage2_fwd_bwd': (_i, [_p, _p, _p, _p, _p]),
    'pnrf_train_explore_fwd_bwd': (_i, [_p, _p, _i, _i, _p, _p, _p]),
    'pnrf_trainer_adam_step': (_i, [_p, _i, _f, _f, _f, _f, _f, _p]),
}


class TrainBatch(C.Structure):
    """pnrf_train_batch_t (include/pronerf_hip.h)."""
    _fields_ = [('rays', _p), ('or_rays', _p), ('target', _p), ('img4', _p), ('poses', _p), ('K', _p), ('ref_nos', _p), ('jitter', _p),
                ('raw_noise', _p), ('n', _i64), ('nv', _i), ('Hf', _i), ('Wf', _i), ('jitter_dir', _i), ('white_bkgd', _i), ('eps', _f),
                ('a_mmrgb', _f), ('clamp', _f), ('layout', _i)]

_lib = None


def load():
    """Load the shared library (once) and bind every declared symbol."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: its wheel bundles its own libamdhip64.so.  Loaded after this library (which would then have pulled in /opt/rocm's copy),
    # the process ends up with two HIP runtimes and every hipMalloc of this library fails with "no ROCm-capable device is detected"
    # (seen with __graft_entry__.build() followed by smoke() in one process).  With torch's runtime already mapped, the library binds to it.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise PnrfError(f'{LIB_PATH} is missing: run `python -m pronerf_amd.build` (or __graft_entry__.build()). '
                        'pronerf_amd has no CPU fallback.')
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise PnrfError(f'cannot load {LIB_PATH}: {e}') from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise PnrfError(f'{LIB_PATH} does not export {name}; rebuild it') from e
        fn.restype = res
        fn.argtypes = args
    v = lib.pnrf_abi_version()
    if v != ABI_VERSION:
        raise PnrfError(f'ABI version mismatch: library {v}, binding {ABI_VERSION}')
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().pnrf_last_error()
        raise PnrfError(f'{what} failed (rc={rc}): {msg.decode() if msg else "?"}')
