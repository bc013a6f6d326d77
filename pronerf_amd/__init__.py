"""pronerf_amd — MI355X-native ProNeRF rendering hot path.

HIP kernels (pronerf_amd/csrc) behind a C ABI (include/pronerf_hip.h), with a host-side mirror
of the reference's operator interface:

    pronerf_amd.run_nerf_helpers     get_embedder, Pluecker, model classes, get_rays, ndc_rays ...
    pronerf_amd.inverse_warp         inverse_warp_rod1_rt2_coords_trt, inverse_warp_rod1_rt2_coords
    pronerf_amd.run_S_eS_eN_alter_trt  render_rays, raw2outputs, render, render_path
    pronerf_amd.run_S_eS_eN_alter_base, ..._base_refine2   the stage-1 / stage-2 training drivers
    pronerf_amd.cli (also reachable as `python -m pronerf.cli`, the reference's entry point name)
"""
__version__ = '0.1.0'
