"""Host-side mirror of the reference's ``inverse_warp`` module for the inference path.

``inverse_warp_rod1_rt2_coords_trt`` (inverse_warp.py:584-619) keeps the reference's signature
and return value; the projection + bilinear fetch run in one HIP kernel (pnrf_warp_trt_fwd).
The 13 legacy warps of the reference module are unused by any config and are not
provided; the training variant ``inverse_warp_rod1_rt2_coords`` is a later row of SURVEY.md §8.
"""
from __future__ import annotations

from . import ops
from .ops import PnrfError


def inverse_warp_rod1_rt2_coords_trt(img, depth, ro1, rd1, w2c, scale=1., padding_mode='zeros'):
    """Warp ``img`` [B,3,Hf,Wf] to the target rays: world point w = ro1 + rd1*depth (homogeneous,
    [B,4,H*W]), pixel = (w2c @ w)[:2] / (w2c @ w)[2], bilinear fetch with zero padding,
    ``align_corners=True``.  depth [B,H,W]; w2c [B,3,4] = K.diag(1,-1,-1).[R|t].
    Returns ``(projected_img [B,3,H,W], None)`` like the reference."""
    if padding_mode != 'zeros':
        raise PnrfError(f"inverse_warp_rod1_rt2_coords_trt: padding_mode={padding_mode!r}; only 'zeros' is implemented "
                        '(the only mode the reference passes, run_S_eS_eN_alter_trt.py:652)')
    B, H, W = depth.shape
    out = ops.warp_trt(img, depth.reshape(B, H * W), ro1, rd1, w2c)
    return out.reshape(B, 3, H, W), None
