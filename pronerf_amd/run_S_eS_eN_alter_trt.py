"""Host-side mirror of the reference's inference driver functions
(run_S_eS_eN_alter_trt.py): ``render_rays``, ``raw2outputs``, ``run_network``, ``render``,
``compute_query_points_from_rays``, ``render_path`` and a ``create_nerf`` that builds the four
modules.  Signatures, kwargs and return values follow the reference so that a driver written
against it runs unchanged; the per-ray work is one fused launch sequence
(pnrf_render_rays_fwd) instead of ~150 aten launches.

Not provided here: the TensorRT/ONNX export path (``use_trt=True``, ``model2onnx``) — NVIDIA
only and excluded by BASELINE.json's north_star; ``render_rays(use_trt=True)`` raises.
"""
from __future__ import annotations

import os
import time
import struct
import zlib

import numpy as np
import torch

from . import inverse_warp, ops
from .ops import PnrfError
from .render import Renderer, projection_matrices, select_neighbors
from .run_nerf_helpers import (DoNeRFTRT, MinMaxRayEpiSamplerTRT_Net, MinMaxRaySamplerTRT_Net, NeRF, Pluecker, get_embedder,  # noqa: F401
                               get_rays, img2mse, mse2psnr, ndc_rays, to8b, weights_from_modules)


def batchify(fn, chunk):
    if chunk is None:
        return fn
    return lambda inputs: torch.cat([fn(inputs[i:i + chunk]) for i in range(0, inputs.shape[0], chunk)], 0)


def run_network(inputs, viewdirs, fn, embed_fn, embeddirs_fn, netchunk=1024 * 64):
    """Embed points and view directions, apply ``fn`` (run_S_eS_eN_alter_trt.py:195-208)."""
    inputs_flat = torch.reshape(inputs, [-1, inputs.shape[-1]]).contiguous()
    embedded = embed_fn(inputs_flat)
    embedded_dirs = None
    if viewdirs is not None:
        input_dirs = viewdirs[:, None].expand(inputs.shape)
        embedded_dirs = embeddirs_fn(torch.reshape(input_dirs, [-1, input_dirs.shape[-1]]).contiguous())
    outputs_flat = fn(embedded, embedded_dirs)
    return torch.reshape(outputs_flat, list(inputs.shape[:-1]) + [outputs_flat.shape[-1]])


def compute_query_points_from_rays(ray_origins, ray_directions, near_thresh, far_thresh, N_point_ray_enc, randomize=True):
    """(query_points [..., P, 3], depth_values [1, P])  (run_S_eS_eN_alter_trt.py:546-562).
    Kept for API parity: the sampler kernel derives these 48 points itself; prefer
    ``ops.ray_encode`` for the whole mm_input."""
    t = torch.from_numpy(ops.linspace(float(near_thresh), float(far_thresh), int(N_point_ray_enc))).to(ray_origins)[None]
    return ray_origins[..., None, :] + ray_directions[..., None, :] * t[..., :, None], t


def raw2outputs(raw, z_vals, rays_d, raw_noise_std=0, white_bkgd=False, pytest=False, mm_density_add=None,
                mm_density_mul=None, iter=1e6):
    """-> (rgb_map, disp_map, acc_map, weights, depth_map)  (run_S_eS_eN_alter_trt.py:564-597).
    Like the reference's inference variant this ignores raw_noise_std and white_bkgd."""
    if mm_density_add is None or mm_density_mul is None:
        raise PnrfError('raw2outputs: the inference variant needs mm_density_add and mm_density_mul')   # the reference would fail on None + tensor
    return ops.composite(raw, z_vals, rays_d, add=mm_density_add, mul=mm_density_mul)


# ------------------------------------------------------------------------------------ render_rays
_RENDERERS = {}


def _renderer(min_max_ray_net, refine_net, network_fine, n_rays, device):
    mods = (min_max_ray_net, refine_net, network_fine)
    for m, cls in zip(mods, (MinMaxRaySamplerTRT_Net, MinMaxRayEpiSamplerTRT_Net, DoNeRFTRT)):
        if not hasattr(m, 'weights'):
            raise PnrfError(f'render_rays: expected a pronerf_amd.run_nerf_helpers.{cls.__name__}, got {type(m).__name__}')
    packed = [m.packed() for m in mods]              # from the parameters, or the engine file loaded into the module
    # one cached Renderer (its workspace is 800 B per ray: 610 MB for a full frame).  The entry holds the modules and their packed
    # handles themselves and is matched by identity, so neither can be freed and have its id() reused while cached.
    ent = _RENDERERS.get('r')
    same = ent is not None and all(a is b for a, b in zip(ent[0], mods)) and all(a is b for a, b in zip(ent[1], packed))
    if not same or ent[2].ctx.max_rays < n_rays or ent[2].device != device:
        cap = max(n_rays, ent[2].ctx.max_rays if same else 0)
        _RENDERERS.clear()                               # free the old workspace before the new one is allocated
        ent = None
        _RENDERERS['r'] = ent = (mods, packed, Renderer(dict(zip(('sampler', 'refine', 'nerf'), packed)), max_rays=cap, device=device))
    return ent[2]


_VIEWS = []          # at most one entry: (ref_rgb, its _version, ref_pose, its _version, n_samples, (img4, mats))


def _packed_views(ref_rgb, ref_pose, n_samples, num_neighbor):
    """The reference hands over the neighbour images replicated x N_samples ([nb*S,3,Hf,Wf],
    trt.py:296-298) and the matrices likewise ([nb*S,3,4], :299-300): take one copy of each.
    The entry keeps the caller's tensors themselves and is matched by identity (+ in-place version): an address or id() can be
    handed out again by the caching allocator after the caller frees a frame's tensor, an object that is still referenced cannot."""
    if _VIEWS:
        r, rv, q, qv, ns, ent = _VIEWS[0]
        if r is ref_rgb and q is ref_pose and rv == ref_rgb._version and qv == ref_pose._version and ns == n_samples:
            return ent
    if ref_rgb.shape[0] == num_neighbor * n_samples:
        imgs, mats = ref_rgb[::n_samples], ref_pose[::n_samples]
    elif ref_rgb.shape[0] == num_neighbor:
        imgs, mats = ref_rgb, ref_pose
    else:
        raise PnrfError(f'render_rays: ref_rgb has {ref_rgb.shape[0]} images, expected num_neighbor*N_samples = {num_neighbor * n_samples}')
    ent = (ops.images_pack(imgs.contiguous()), mats.contiguous().clone())
    _VIEWS[:] = [(ref_rgb, ref_rgb._version, ref_pose, ref_pose._version, n_samples, ent)]
    return ent


_MM_CHECKED = []     # at most one entry: (weakref to the verified mm tensor, its _version, weakref to the ray tensor, its _version) — weak, so that a verified 878 MB encoding is not kept alive by this module


def _row_slice(t):
    """(base, first row) when ``t`` is a block of whole rows of a 2-D base tensor (``base[i:j]``, or the base itself), else (t, 0)."""
    base = t._base
    if base is None or base.dim() != 2 or t.dim() != 2 or t.stride() != base.stride() or t.shape[1] != base.shape[1] or base.stride(0) <= 0:
        return t, 0
    off = t.storage_offset() - base.storage_offset()
    if off < 0 or off % base.stride(0) or off // base.stride(0) + t.shape[0] > base.shape[0]:
        return t, 0
    return base, off // base.stride(0)


def _check_mm_input(mm_input, ray_batch, n_pts):
    """The reference's sampler consumes ``kwargs['mm_input']`` (run_S_eS_eN_alter_trt.py:625-628); the fused sampler recomputes the encoding
    from ``ray_batch`` in its batch head.  That is the same thing only while ``mm_input`` IS the Pluecker encoding of ``ray_batch`` — what
    ``render_path`` builds (trt.py:273-277).  So a caller's ``mm_input`` is verified once per tensor and refused (PnrfError) otherwise — a
    different ``mm_input`` would silently be ignored.

    Once per tensor: the verified pair is remembered by identity + in-place version (weak references: a freed tensor cannot match).  Row
    slices of one pair of frame tensors (``mm_input[i:j]`` with ``ray_batch[i:j]``, the reference's per-chunk calling pattern) resolve to
    their bases: the bases are verified whole on the first call and every later slice of them is a hit — no device work, no sync.
    Tolerance: 32 ulp of max(|value|, 1, (|o| + |d|) |d|) per ray — a moment o x d evaluated with another contraction on the caller's side
    differs by ulps of |o||d|, not of the (possibly cancelling) result.  Cost of a miss: one ``ops.ray_encode`` per 65 536 rays and ONE
    host sync at the end (the mismatch count is accumulated on the device)."""
    import weakref
    mm_b, mm_r0 = _row_slice(mm_input)
    ry_b, ry_r0 = _row_slice(ray_batch)
    if not (mm_r0 == ry_r0 and mm_b.shape[0] == ry_b.shape[0]):          # not the same rows of two frame tensors: verify exactly what was passed
        mm_b, ry_b = mm_input, ray_batch
    if _MM_CHECKED:
        m, mv, r, rv = _MM_CHECKED[0]
        if m() is mm_b and r() is ry_b and mv == mm_b._version and rv == ry_b._version:
            return
    if mm_b.shape != (ry_b.shape[0], 6 * n_pts):
        raise PnrfError(f'render_rays: mm_input has shape {tuple(mm_input.shape)}, expected {(ray_batch.shape[0], 6 * n_pts)} (6 * N_point_ray_enc per ray)')
    if mm_b.device != ry_b.device:
        raise PnrfError('render_rays: mm_input and ray_batch are on different devices')
    tol = 32 * torch.finfo(torch.float32).eps
    n_bad = torch.zeros((), dtype=torch.int64, device=ry_b.device)
    first = torch.full((), ry_b.shape[0], dtype=torch.int64, device=ry_b.device)
    worst = torch.zeros((), dtype=torch.float32, device=ry_b.device)
    for a in range(0, ry_b.shape[0], 65536):
        rb = ry_b[a:a + 65536].contiguous()
        want = ops.ray_encode(rb, n_pts)
        dn = rb[:, 3:6].norm(dim=1)
        scale = torch.maximum(want.abs().clamp_min(1.0), ((rb[:, 0:3].norm(dim=1) + dn) * dn)[:, None])
        diff = (mm_b[a:a + 65536].to(torch.float32) - want).abs()
        bad = ~(diff <= tol * scale)                                # NaN counts as different
        n_bad += bad.sum()
        rows = bad.any(1)
        first = torch.minimum(first, torch.where(rows.any(), a + rows.to(torch.int64).argmax(), first))
        worst = torch.maximum(worst, torch.where(bad, diff.nan_to_num(float('inf')), torch.zeros_like(diff)).max())
    n_bad, first, worst = int(n_bad), int(first), float(worst)     # the one sync
    if n_bad:
        raise PnrfError(f'render_rays: mm_input is not the Pluecker encoding of ray_batch ({n_bad} values differ, first in row {first}: max |diff| {worst:.3g}); '
                        'the fused sampler encodes ray_batch itself (trt.py:273-277) and cannot honour a different mm_input — pass mm_input=None, or use '
                        'ops.sampler_fwd on rays built from it')
    _MM_CHECKED[:] = [(weakref.ref(mm_b), mm_b._version, weakref.ref(ry_b), ry_b._version)]


def render_rays(ray_batch, or_ray_batch, network_fn, network_query_fn, N_samples, retraw=False, lindisp=False, perturb=0.,
                N_importance=0, network_fine=None, white_bkgd=False, raw_noise_std=0., min_max_ray_net=None, refine_net=None,
                N_point_ray_enc=0, embed_fn=None, embeddirs_fn=None, randomize=True, verbose=False, pytest=False, **kwargs):
    """Inference render of a ray batch (run_S_eS_eN_alter_trt.py:599-696).

    ray_batch [N,11] = [o'(3), d'(3), near, far, viewdir(3)] (NDC); or_ray_batch [N,11] the same in
    camera/world space.  kwargs consumed: ``use_trt, num_neighbor, ref_rgb, ref_pose, mm_input``; ``mm_input`` (the
    reference's sampler input, trt.py:625-628) is VERIFIED to be the Pluecker encoding of ``ray_batch`` once per tensor
    (``_check_mm_input``) and refused otherwise — the fused sampler recomputes that encoding in its batch head; ``ro1, rd1,
    embed_rays`` are accepted and unused: the kernels build the homogeneous rays from or_ray_batch, of which they are pure
    functions (trt.py:250-277).  Returns ``{'rgb_map0', 'rgb_map1', 'depth_map'}``.
    """
    if kwargs.get('use_trt'):                                # engines = engine files loaded into the modules (no TensorRT on ROCm)
        for m in (min_max_ray_net, refine_net, network_fine):
            if getattr(m, 'engine_path', None) is None:
                raise PnrfError(f'render_rays(use_trt=True): {type(m).__name__} has no engine file loaded (module.load_engine(path), '
                                'written by --export_only / save_engine); with use_trt=False the parameters are packed on first use')
    num_neighbor = kwargs['num_neighbor']
    n_pts = getattr(min_max_ray_net, 'input_ch', 288) // 6            # what the modules were built with (create_nerf: 6 * N_point_ray_enc, 48 + 24 * num_neighbor)
    nb_net = (getattr(refine_net, 'input_ch', 144) - 48) // 24
    if N_samples != 8 or num_neighbor != nb_net or N_point_ray_enc not in (0, n_pts):
        raise PnrfError(f'render_rays: N_samples must be 8 (the kernels\' sample count) and num_neighbor / N_point_ray_enc must be the ones the refine / sampler '
                        f'modules were built with ({nb_net} / {n_pts}); got N_samples={N_samples}, num_neighbor={num_neighbor}, N_point_ray_enc={N_point_ray_enc}')
    if ray_batch.shape[-1] != 11 or or_ray_batch.shape[-1] != 11:
        raise PnrfError('render_rays: ray batches must be [N,11] (use_viewdirs=True)')
    if kwargs.get('mm_input') is not None:
        _check_mm_input(kwargs['mm_input'], ray_batch, n_pts)
    rgbd = _render_rgbd(ray_batch, or_ray_batch, min_max_ray_net, refine_net, network_fine, kwargs['ref_rgb'], kwargs['ref_pose'], N_samples, num_neighbor,
                        out=kwargs.get('out_rgbd'))
    rgb_map, depth_map = rgbd[:, :3], rgbd[:, 3]
    return {'rgb_map0': rgb_map, 'rgb_map1': rgb_map, 'depth_map': depth_map}


def _render_rgbd(ray_batch, or_ray_batch, min_max_ray_net, refine_net, network_fine, ref_rgb, ref_pose, n_samples, num_neighbor, out=None):
    """The packed [n, 4] = (rgb_map, depth_map) result of the fused path (into ``out`` if given): what ``render_rays`` slices its dict from
    and what the sharded frame driver gathers (rgb_map0 and rgb_map1 are the same tensor at inference, trt.py:695)."""
    rend = _renderer(min_max_ray_net, refine_net, network_fine, ray_batch.shape[0], ray_batch.device)
    img4, proj = _packed_views(ref_rgb, ref_pose, n_samples, num_neighbor)
    with torch.cuda.device(rend.device):                     # the context's kernels go to ITS device's current stream
        rgbd, _ = rend.ctx.render_rays(ray_batch, or_ray_batch, img4, proj, out=out)
    return rgbd


def apply_preset(preset, min_max_ray_net, refine_net, network_fine, probe=None):
    """The renderer's operating point on the modules' packed handles (pronerf_amd.render.PRESETS; ``--pnrf_preset`` of the inference driver).
    'quality': every ray through the split-fp16 sampler (exact indices) + fp16 NeRF operands.  'auto': ``probe`` = (ray_batch, or_ray_batch, ref_rgb,
    ref_pose, n_samples, num_neighbor) is rendered once and the sampler switches to its exact single pass when the two-pass form re-rendered more
    than 55 % of the rays (Renderer.calibrate: a sampler that has learned surfaces) — once per checkpoint, deterministic.  Returns what is in force."""
    from .render import PRESETS
    if preset not in ('default', 'quality', 'auto'):
        raise PnrfError(f"pnrf_preset must be 'default', 'quality' or 'auto', got {preset!r}")
    if preset == 'quality':
        for net, variant in PRESETS['quality'].items():
            {'sampler': min_max_ray_net, 'refine': refine_net, 'nerf': network_fine}[net].packed().set_variant(variant)
        return 'quality'
    if preset == 'auto' and probe is not None and min_max_ray_net.packed().variant == 'default':
        rays, or_rays, ref_rgb, ref_pose, S, NB = probe
        _render_rgbd(rays, or_rays, min_max_ray_net, refine_net, network_fine, ref_rgb, ref_pose, S, NB)
        frac = _renderer(min_max_ray_net, refine_net, network_fine, rays.shape[0], rays.device).ctx.sampler_stats() / max(1, rays.shape[0])
        if frac > 0.55:
            min_max_ray_net.packed().set_variant('sampler_split')
        return f"auto: second pass {frac:.1%} -> sampler {min_max_ray_net.packed().variant}"
    return 'default'


def render(rays, or_rays, sh, **kwargs):
    """Render and reshape to the image (run_S_eS_eN_alter_trt.py:211-221)."""
    all_ret = render_rays(rays, or_rays, **kwargs)
    for k in all_ret:
        all_ret[k] = torch.reshape(all_ret[k], list(sh[:-1]) + list(all_ret[k].shape[1:]))
    k_extract = ['rgb_map0', 'rgb_map1', 'depth_map']
    return [all_ret[k] for k in k_extract] + [{k: all_ret[k] for k in all_ret if k not in k_extract}]


def render_sharded(rays, or_rays, sh, **kwargs):
    """``render`` with the frame's rays sharded over the ranks of the process group (not in the reference; the multi-GPU shape of
    SURVEY.md §8(e)): rank r renders the contiguous flat range ``shard_range(N, r, world)`` and one all-gather of the packed
    [n, 4] = (rgb_map, depth_map) tiles rebuilds the frame on every rank (rgb_map0 is rgb_map1 at inference, trt.py:695: one copy travels).
    Returns (rgb0, rgb1, depth) shaped like ``render``'s.  ``rays`` / ``or_rays`` are the whole frame's; ``render_path`` (which generates
    only its own shard's rays and pipelines the gathers over frames) is what a run under torchrun uses."""
    from .dist import render_frame_sharded

    def part(first, count):
        r = render_rays(rays[first:first + count].contiguous(), or_rays[first:first + count].contiguous(), **kwargs)
        return torch.cat([r['rgb_map1'], r['depth_map'][:, None]], 1)

    full = render_frame_sharded(part, rays.shape[0], out_channels=4, device=rays.device)
    hw = list(sh[:-1])
    rgb = full[:, 0:3].reshape(hw + [3])
    return rgb, rgb, full[:, 3].reshape(hw)


# ------------------------------------------------------------------------------------ model construction
def create_nerf(args, device='cuda'):
    """Build the inference modules and the test-time render kwargs (run_S_eS_eN_alter_trt.py:412-544),
    optionally loading ``args.ft_path`` (keys of :476-481).  No ONNX side effects."""
    embed_fn, input_ch = get_embedder(args.multires, getattr(args, 'i_embed', 0))
    embeddirs_fn, input_ch_views = get_embedder(args.multires_views, getattr(args, 'i_embed', 0))
    model_fine = DoNeRFTRT(D=args.netdepth, W=args.netwidth, n_in=input_ch + input_ch_views, n_out=4, skip='auto').to(device)
    model_mmray = MinMaxRaySamplerTRT_Net(D=args.mmnetdepth, W=args.mmnetwidth, input_ch=6 * args.N_point_ray_enc,
                                          output_ch=3 * args.N_samples + 3, skips=args.mmnetskips, N_samples=args.N_samples).to(device)
    model_refine = MinMaxRayEpiSamplerTRT_Net(D=args.mmnetdepth, W=args.mmnetwidth,
                                              input_ch=6 * args.N_samples + 3 * args.num_neighbor * args.N_samples,
                                              output_ch=4 * args.N_samples + 3, skips=args.mmnetskips, N_samples=args.N_samples).to(device)
    start = 0
    ft = getattr(args, 'ft_path', None)
    if ft is not None and ft != 'None':
        ckpt = torch.load(ft, map_location=device)
        start = ckpt.get('global_step', 0)
        model_mmray.load_state_dict(ckpt['mmr_network_fn_state_dict'])
        model_refine.load_state_dict(ckpt['refine_net_state_dict'])
        fine_sd = ckpt['network_fine_state_dict']
        if any(k.startswith('pts_linears.') for k in fine_sd):            # saved from the NeRF class
            model_fine = NeRF(D=args.netdepth, W=args.netwidth, input_ch=input_ch, input_ch_views=input_ch_views, output_ch=4,
                              skips=[4], use_viewdirs=True).to(device)
        model_fine.load_state_dict(fine_sd)
    network_query_fn = lambda inputs, viewdirs, network_fn: run_network(inputs, viewdirs, network_fn, embed_fn=embed_fn,
                                                                        embeddirs_fn=embeddirs_fn, netchunk=getattr(args, 'netchunk', 1024 * 64))
    kw = {'network_query_fn': network_query_fn, 'perturb': False, 'N_importance': 0, 'network_fine': model_fine,
          'N_samples': args.N_samples, 'network_fn': None, 'white_bkgd': getattr(args, 'white_bkgd', False), 'raw_noise_std': 0.,
          'min_max_ray_net': model_mmray, 'refine_net': model_refine, 'N_point_ray_enc': args.N_point_ray_enc,
          'embed_rays': Pluecker(), 'embed_fn': embed_fn, 'embeddirs_fn': embeddirs_fn, 'num_neighbor': args.num_neighbor,
          'use_trt': False, 'randomize': False, 'count_flops': False}
    return kw, start


def engine_paths(export_dir, args=None):
    """Engine files of one experiment (pronerf/tensorrt.py:8-14 ``expected_engine_paths``), with the per-net overrides of the
    script's ``--*_engine_path`` options."""
    out = {'nerf': os.path.join(export_dir, 'nerf.pnrf'), 'sampler': os.path.join(export_dir, 'minmaxrays_net.pnrf'),
           'refine': os.path.join(export_dir, 'refine_net.pnrf')}
    for key, opt in (('nerf', 'nerf_engine_path'), ('sampler', 'mm_engine_path'), ('refine', 'refine_engine_path')):
        v = getattr(args, opt, None) if args is not None else None
        if v not in (None, 'None'):
            out[key] = v
    return out


def load_fine_engine(model_fine, path, args, device):
    """Load the fine-network engine; it may hold either fine class (a checkpoint saved by the stage-2 trainer exports a ``NeRF``-class
    engine), so the module is swapped to the class the file was written from."""
    try:
        model_fine.load_engine(path)
        return model_fine
    except PnrfError as e:
        if 'expected' not in str(e):
            raise
    embed_ch = get_embedder(args.multires, getattr(args, 'i_embed', 0))[1]
    views_ch = get_embedder(args.multires_views, getattr(args, 'i_embed', 0))[1]
    if isinstance(model_fine, DoNeRFTRT):
        other = NeRF(D=args.netdepth, W=args.netwidth, input_ch=embed_ch, input_ch_views=views_ch, output_ch=4, skips=[4], use_viewdirs=True)
    else:
        other = DoNeRFTRT(D=args.netdepth, W=args.netwidth, n_in=embed_ch + views_ch, n_out=4, skip='auto')
    other = other.to(device)
    other.load_engine(path)
    return other


# ------------------------------------------------------------------------------------ render_path
def _write_png(path, img_u8):
    """Minimal 8-bit RGB/gray PNG writer (imageio is not a dependency of this package)."""
    img_u8 = np.ascontiguousarray(img_u8)
    h, w = img_u8.shape[:2]
    ch = 1 if img_u8.ndim == 2 else img_u8.shape[2]
    ctype = {1: 0, 3: 2, 4: 6}[ch]
    rows = img_u8.reshape(h, w * ch)
    raw = b''.join(b'\x00' + rows[i].tobytes() for i in range(h))

    def chunk(tag, data):
        c = struct.pack('>I', len(data)) + tag + data
        return c + struct.pack('>I', zlib.crc32(tag + data) & 0xffffffff)

    with open(path, 'wb') as f:
        f.write(b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 8, ctype, 0, 0, 0)) +
                chunk(b'IDAT', zlib.compress(raw, 6)) + chunk(b'IEND', b''))


def render_path(render_poses, hwf, K, chunk, render_kwargs, gt_imgs=None, savedir=None, render_factor=0, near=0., far=1.,
                or_near=1., or_far=10., n_timing_reps=20, verbose=True):
    """Render every pose (run_S_eS_eN_alter_trt.py:223-375): per-frame set-up, ``n_timing_reps``
    timed ``render()`` calls bracketed by device events (:327-332), PSNR and PNG output.
    ``render_kwargs`` needs ``poses`` [n,3,4], ``images`` [n,H,W,3], ``ref_K`` in addition to the
    ``create_nerf`` entries.  Returns (rgbs0, rgbs1, depths, depths) and stores the per-frame
    milliseconds in ``render_kwargs['render_ms']``."""
    H, W, focal = hwf
    if render_factor != 0:
        H, W, focal = H // render_factor, W // render_factor, focal / render_factor
    S, NB = render_kwargs['N_samples'], render_kwargs['num_neighbor']
    Kh = np.asarray(K.detach().cpu() if isinstance(K, torch.Tensor) else K, dtype=np.float32)
    poses = render_kwargs['poses']
    poses_h = np.asarray(poses.detach().cpu() if isinstance(poses, torch.Tensor) else poses, dtype=np.float32)
    images = render_kwargs['images']
    ref_K = render_kwargs.get('ref_K', K)
    ref_Kh = np.asarray(ref_K.detach().cpu() if isinstance(ref_K, torch.Tensor) else ref_K, dtype=np.float32)
    dev = next(render_kwargs['network_fine'].parameters()).device
    rgbs0, rgbs1, depths, psnrs, times, walls = [], [], [], [], [], []
    from concurrent.futures import ThreadPoolExecutor
    from .dist import FrameGather, world as _world
    from .render import RayPartition
    rank, world_size = _world()               # under torchrun every frame's rays are sharded over the ranks
    verbose = verbose and rank == 0
    t1, t2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fwd = {k: render_kwargs[k] for k in ('network_fn', 'network_query_fn', 'N_samples', 'network_fine', 'min_max_ray_net', 'refine_net',
                                          'N_point_ray_enc', 'embed_fn', 'embeddirs_fn', 'num_neighbor', 'use_trt', 'embed_rays')
           if k in render_kwargs}
    n_rays = H * W
    part = RayPartition(n_rays, world_size, os.environ.get('PNRF_RAY_PARTITION', 'cyclic'))      # blocks of 1024 rays round-robin: every rank gets the frame's
    count = part.count(rank)                                                                      # average share of second-pass rays (render.RayPartition)
    # N > 1 = what bench.py times: this rank's rays only, the [n, 4] tiles gathered by RCCL on its own stream while the next render runs
    # (FrameGather: two buffers in flight over the timing repetitions and over the poses), the next pose's neighbour images uploaded from
    # pinned memory on a copy stream meanwhile, PNG encoding on a worker thread.
    fg = FrameGather(n_rays, 4, device=dev, pipelined=True, partition=part) if world_size > 1 else None
    copy_stream = torch.cuda.Stream(device=dev)
    png_pool = ThreadPoolExecutor(max_workers=1) if (savedir is not None and rank == 0) else None
    png_jobs = []
    poses_list = [np.asarray(c.detach().cpu() if isinstance(c, torch.Tensor) else c, dtype=np.float32) for c in render_poses]

    def upload(i):
        """Neighbour images / matrices of pose i -> device, asynchronously on the copy stream (trt.py:281-296)."""
        ref_nos = select_neighbors(poses_list[i], poses_h, NB)                                                           # :281-284
        nb = images[ref_nos] if isinstance(images, np.ndarray) else images[torch.as_tensor(ref_nos)]
        host = torch.as_tensor(nb, dtype=torch.float32)
        if host.device.type == 'cpu':
            host = host.contiguous().pin_memory()
        with torch.cuda.stream(copy_stream):
            ref_rgb = host.to(dev, non_blocking=True).permute(0, 3, 1, 2).contiguous()                                  # :286,296
            ref_pose = torch.from_numpy(projection_matrices(ref_Kh, poses_h[ref_nos])).to(dev, non_blocking=True)     # :289-294
            ev = torch.cuda.Event(); ev.record(copy_stream)
        return ref_rgb, ref_pose, ev, host

    nxt = upload(0) if poses_list else None
    for i, c2w_h in enumerate(poses_list):
        tw = time.perf_counter()
        rays, or_rays = ops.frame_rays(Kh, c2w_h, H, W, near=near, far=far, or_near=or_near, or_far=or_far, device=dev, **part.frame_rays_args(rank))   # :245-271
        ref_rgb, ref_pose, ev, _host = nxt
        torch.cuda.current_stream(dev).wait_event(ev)
        ref_rgb.record_stream(torch.cuda.current_stream(dev))
        nxt = upload(i + 1) if i + 1 < len(poses_list) else None                 # overlaps this pose's renders
        sh = (H, W, 3)
        frame_ms = []
        b = 0
        if i == 0 and render_kwargs.get('pnrf_preset', 'default') != 'default' and count > 0:      # the operating point, before anything is timed
            render_kwargs['pnrf_preset_in_force'] = apply_preset(render_kwargs['pnrf_preset'], fwd['min_max_ray_net'], fwd['refine_net'], fwd['network_fine'],
                                                                 probe=(rays, or_rays, ref_rgb, ref_pose, S, NB))
            if verbose:
                print('pnrf_preset:', render_kwargs['pnrf_preset_in_force'])
        for _ in range(n_timing_reps):                                                                                     # :327-332
            t1.record()
            if fg is not None:
                b = fg.acquire()
                if count > 0:
                    _render_rgbd(rays, or_rays, fwd['min_max_ray_net'], fwd['refine_net'], fwd['network_fine'], ref_rgb, ref_pose, S, NB, out=fg.outs[b][:count])
                fg.submit(b)
            else:
                rgb0, rgb1, depth_map, _ = render(rays, or_rays, sh, ref_rgb=ref_rgb, ref_pose=ref_pose, **fwd)
            t2.record()
            torch.cuda.synchronize(device=dev)
            frame_ms.append(t1.elapsed_time(t2))
            if verbose:
                print('Render path time:', frame_ms[-1])
        if fg is not None:
            full = fg.frame(b)                                          # waits for the last repetition's gather
            rgb1 = full[:, 0:3].reshape(H, W, 3); rgb0 = rgb1; depth_map = full[:, 3].reshape(H, W)
        times.append(frame_ms)
        rgbs0.append(rgb0.cpu().numpy()); rgbs1.append(rgb1.cpu().numpy()); depths.append(depth_map.cpu().numpy())
        if gt_imgs is not None and render_factor == 0:
            psnrs.append(mse2psnr(img2mse(rgb1, torch.as_tensor(gt_imgs[i], dtype=torch.float32).to(dev))))
        if png_pool is not None:
            os.makedirs(savedir, exist_ok=True)
            png_jobs.append(png_pool.submit(_write_png, os.path.join(savedir, '{:03d}.png'.format(i)), to8b(rgbs1[-1])))
            png_jobs.append(png_pool.submit(_write_png, os.path.join(savedir, 'depth_{:03d}.png'.format(i)), to8b(depths[-1] / np.max(depths[-1]))))
        walls.append((time.perf_counter() - tw) * 1e3)
    for j in png_jobs:
        j.result()                                                      # re-raises a writer's exception
    if png_pool is not None:
        png_pool.shutdown()
    render_kwargs['pose_wall_ms'] = walls                             # host wall time per pose: set-up, n_timing_reps renders, gather, read-back
    render_kwargs['render_ms'] = times
    if len(psnrs) > 0 and verbose:
        print(psnrs)
        print(f'Mean Test PSNR {float(sum(psnrs) / len(psnrs))}')
    render_kwargs['psnrs'] = [float(p) for p in psnrs]
    return np.stack(rgbs0, 0), np.stack(rgbs1, 0), np.stack(depths, 0), np.stack(depths, 0)


# ------------------------------------------------------------------------------------ frame driver
def config_parser():
    """Options of the inference script (run_S_eS_eN_alter_trt.py:44-183); see ``pronerf_amd.config``."""
    from .config import config_parser as _cp
    return _cp('trt')


def train(argv=None, device='cuda'):
    """The inference driver (run_S_eS_eN_alter_trt.py:699-800 — the reference calls it ``train`` too): load the LLFF scene,
    pick the reference views, load the stage-2 checkpoint, render the hold-out views (``--render_test``) or the spiral path,
    write PNGs + PSNR.  Returns the ``render_kwargs`` dict (per-frame milliseconds in ``render_ms``, PSNRs in ``psnrs``).

    Deviations, all from SURVEY.md Appendix B: ``load_llff_data_infer`` gets ``num_neighbor=None`` like the reference's call
    (B-2), which here ranks all training views instead of raising; no ONNX export side effect (B-4).

    Engines: where the reference exports ONNX and runs serialized TensorRT engines, this build has engine files holding the packed
    weight stream (``pnrf_mlp_serialize``).  ``--export_only`` writes ``nerf.pnrf``, ``minmaxrays_net.pnrf`` and ``refine_net.pnrf``
    into ``<basedir>/<expname>`` (the directory of the reference's ``*_fp16.trt`` files, :497-499) and returns before rendering
    (:768-770); ``--use_trt`` loads the three networks from those files (or from ``--nerf_engine_path`` / ``--mm_engine_path`` /
    ``--refine_engine_path``) instead of packing the checkpoint — ``--ft_path`` is then not needed."""
    from .load_llff import load_llff_data_infer
    from .run_S_eS_eN_alter_base_refine2 import dist_setup
    args = config_parser().parse_args(argv)
    rank, _, device = dist_setup(device)        # torchrun: one process per GPU, each frame's rays sharded over them; rank 0 writes the output
    if args.dataset_type != 'llff':
        raise ValueError('only dataset_type=llff is supported (as in the reference release)')
    if args.no_ndc or args.lindisp:
        raise PnrfError('--no_ndc / --lindisp: the HIP path is built for forward-facing scenes in NDC with samples linear in depth (the LLFF configs)')
    out_root = os.path.join(args.basedir, args.expname or 'pronerf')
    if args.export_only:                                       # :768-770; needs the checkpoint only, so it runs before the scene is read
        os.makedirs(out_root, exist_ok=True)
        kw, _ = create_nerf(args, device=device)
        engines = engine_paths(out_root, args)
        kw['min_max_ray_net'].save_engine(engines['sampler'])
        kw['refine_net'].save_engine(engines['refine'])
        kw['network_fine'].save_engine(engines['nerf'])
        kw['engine_paths'] = engines
        print('Exported engine files; export_only requested, skipping render:', ', '.join(engines[k] for k in ('nerf', 'sampler', 'refine')))
        return kw
    images, poses, bds, render_poses, i_test, i_ref = load_llff_data_infer(args.datadir, args.factor, recenter=True, bd_factor=.75,
                                                                            spherify=args.spherify)
    hwf = poses[0, :3, -1]
    poses = poses[:, :3, :4]
    render_poses = render_poses[:, :3, :4]
    if args.llffhold > 0:
        i_test = np.arange(images.shape[0])[::args.llffhold]                                 # :722-724
    near, far = (float(bds.min()) * .9, float(bds.max())) if args.no_ndc else (0., 1.)    # :731-738
    H, W, focal = int(hwf[0]), int(hwf[1]), float(hwf[2])
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]], dtype=np.float32)  # :746-751
    if rank == 0:
        os.makedirs(out_root, exist_ok=True)
        with open(os.path.join(out_root, 'args.txt'), 'w') as f:                             # :757-761
            for k in sorted(vars(args)):
                f.write('{} = {}\n'.format(k, getattr(args, k)))
    kw, start = create_nerf(args, device=device)
    if args.use_trt:                                                                          # :490-499
        engines = engine_paths(out_root, args)
        kw['network_fine'] = load_fine_engine(kw['network_fine'], engines['nerf'], args, device)
        kw['min_max_ray_net'].load_engine(engines['sampler'])
        kw['refine_net'].load_engine(engines['refine'])
        kw['engine_paths'] = engines
        kw['use_trt'] = True
    kw.update({'near': near, 'far': far, 'images': images[i_ref], 'poses': poses[i_ref], 'ref_K': K})   # :773-787
    kw['pnrf_preset'] = getattr(args, 'pnrf_preset', 'default')
    if args.max_images is not None:
        i_test = i_test[:args.max_images]
    if args.render_test:
        targets, gt = poses[i_test], images[i_test]
    else:
        targets, gt = render_poses, None
    savedir = os.path.join(out_root, 'renderonly_{}_{:06d}'.format('test' if args.render_test else 'path', start))
    with torch.no_grad():
        render_path(targets, [H, W, focal], K, args.chunk, kw, gt_imgs=gt, savedir=savedir, render_factor=args.render_factor,
                    near=near, far=far)
    if rank == 0:
        print('Saved test set' if args.render_test else 'Saved render path', savedir)
    return kw


if __name__ == '__main__':
    train()
