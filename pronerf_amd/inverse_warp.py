"""Host-side mirror of the reference's ``inverse_warp`` module: the two warps its drivers call.

``inverse_warp_rod1_rt2_coords_trt`` (inverse_warp.py:584-619; inference, run_S_eS_eN_alter_trt.py:652) and
``inverse_warp_rod1_rt2_coords`` (inverse_warp.py:515-581; training, run_S_eS_eN_alter_base_refine2.py:617 and
run_S_eS_eN_alter_base.py:654) keep the reference's names, positional order, keyword defaults and ``(projected, None)``
return value; projection + bilinear fetch run in one HIP kernel each (``pnrf_warp_trt_fwd`` / ``pnrf_warp_train_fwd``).
The 13 legacy warps of the reference module are unused by any config and are not provided.
"""
from __future__ import annotations

from . import ops
from .ops import PnrfError


def _only_plain(fn, scale, padding_mode):
    if padding_mode != 'zeros':
        raise PnrfError(f"{fn}: padding_mode={padding_mode!r}; only 'zeros' is implemented (the only mode the reference's drivers pass)")
    if scale != 1:
        raise PnrfError(f'{fn}: scale={scale!r}; only scale=1 is implemented (no driver of the reference passes another value)')


def inverse_warp_rod1_rt2_coords_trt(img, depth, ro1, rd1, w2c, scale=1., padding_mode='zeros'):
    """Warp ``img`` [B,3,Hf,Wf] to the target rays: world point w = ro1 + rd1*depth (homogeneous,
    [B,4,H*W]), pixel = (w2c @ w)[:2] / (w2c @ w)[2], bilinear fetch with zero padding,
    ``align_corners=True``.  depth [B,H,W]; w2c [B,3,4] = K.diag(1,-1,-1).[R|t].
    Returns ``(projected_img [B,3,H,W], None)`` like the reference."""
    _only_plain('inverse_warp_rod1_rt2_coords_trt', scale, padding_mode)
    B, H, W = depth.shape
    out = ops.warp_trt(img, depth.reshape(B, H * W), ro1, rd1, w2c)
    return out.reshape(B, 3, H, W), None


def inverse_warp_rod1_rt2_coords(img, depth, ro1, rd1, c2w2, intrinsics, intrinsics_inv, scale=1., padding_mode='zeros'):
    """Training warp: w = ro1 + rd1*depth ([B,3,H*W], no homogeneous row); c2 = R2^T w - R2^T t2 with ``c2w2`` [B,3,4]
    camera-to-world; c2 /= |c2.z| + 1e-8, c2.z = 1, c2.y = -c2.y; p = ``intrinsics`` [B,3,3] @ c2; a sample whose
    normalised X or Y leaves [-1, 1] yields 0 (the reference moves it to 2, inverse_warp.py:557-561); otherwise a bilinear
    fetch with zero padding, ``align_corners=True``.  ``intrinsics_inv`` is accepted and, as in the reference, not used.
    depth [B,H,W].  Returns ``(projected_img [B,3,H,W], None)``."""
    _only_plain('inverse_warp_rod1_rt2_coords', scale, padding_mode)
    B, H, W = depth.shape
    out = ops.warp_train(img, depth.reshape(B, H * W), ro1, rd1, c2w2, intrinsics)
    return out.reshape(B, 3, H, W), None
