"""Host-side mirror of the reference's stage-2 (refine) trainer functions for the FORWARD of the path
(run_S_eS_eN_alter_base_refine2.py): ``render_rays`` (:525-680) and ``raw2outputs`` (:475-522).

Same signature, kwargs and returned dict as the reference.  The per-ray work runs in HIP kernels:
sampler MLP + sort (pnrf_sampler_fwd), training projection with valid-mask mean fill
(pnrf_refine_input_train_fwd), refine MLP + depth jitter + query points (pnrf_refine_train_fwd), NeRF-class
MLP + compositing with sigma noise / white background (pnrf_nerf_train_fwd).  The random draws of the
reference (one ``random.sample`` of neighbour ranks and one coin flip per batch, |N(0,1)|/5 jitter, N(0,1)
sigma noise) are made here with the same generators (``random``, ``torch.normal``, ``torch.randn``) and handed to
the kernels as inputs.

Forward only: the outputs carry no autograd graph.  The fused backward of the three MLPs and the optimizer
step are the next row of SURVEY.md §8(f); until then this serves evaluation during training (i_testset renders,
refine2.py:981-994 style) and the parity of the training graph's forward.
"""
from __future__ import annotations

import random

import numpy as np
import torch

from . import ops
from .ops import PnrfError
from .run_nerf_helpers import NeRF, MinMaxRay_Net, Pluecker, get_embedder, img2mse, mse2psnr, to8b  # noqa: F401


def raw2outputs(raw, z_vals, rays_d, raw_noise_std=0, white_bkgd=False, pytest=False, mm_density_add=None, mm_density_mul=None, iter=1e6):
    """-> (rgb_map, disp_map, acc_map, weights, depth_map)  (refine2.py:475-522): sigma noise when
    raw_noise_std > 0 (``pytest`` replaces it by the reference's fixed numpy draw), optional density
    modulation, white background."""
    noise = None
    if raw_noise_std > 0.:
        noise = torch.randn(raw[..., 3].shape, device=raw.device) * raw_noise_std
        if pytest:
            np.random.seed(0)
            noise = torch.tensor(np.random.rand(*list(raw[..., 3].shape)) * raw_noise_std, dtype=torch.float32, device=raw.device)
    return ops.composite(raw, z_vals, rays_d, add=mm_density_add, mul=mm_density_mul if mm_density_add is not None else None, noise=noise,
                         white_bkgd=white_bkgd)


def neighbor_rank_table(poses):
    """[nv, nv]: row c = training cameras sorted by distance to camera c (refine2.py:587-588); host, O(nv^2)."""
    p = np.asarray(poses.detach().cpu() if isinstance(poses, torch.Tensor) else poses, dtype=np.float32)
    d = np.sqrt(((p[:, None, :3, 3] - p[None, :, :3, 3]) ** 2).sum(2, dtype=np.float32))
    return np.argsort(d, axis=1, kind='stable')


_PACK = {}


def _packed(mod, kind):
    if hasattr(mod, 'packed'):
        return mod.packed()
    raise PnrfError(f'render_rays: expected a pronerf_amd.run_nerf_helpers module for the {kind} net, got {type(mod).__name__}')


_VIEWS = {}


def _train_views(images, poses, ref_K, device):
    key = (id(images), getattr(images, '_version', 0), id(poses), getattr(poses, '_version', 0))
    ent = _VIEWS.get(key)
    if ent is None:
        img = torch.as_tensor(images, dtype=torch.float32).to(device)
        img4 = ops.images_pack(img.permute(0, 3, 1, 2).contiguous())                      # [nv,H,W,3] -> [nv,H,W,4]
        pz = torch.as_tensor(poses, dtype=torch.float32).to(device)[:, :3, :4].contiguous()
        K = torch.as_tensor(ref_K, dtype=torch.float32).to(device).reshape(3, 3).contiguous()
        _VIEWS.clear()
        _VIEWS[key] = ent = (img4, pz, K, torch.from_numpy(neighbor_rank_table(pz)).to(device))
    return ent


def render_rays(ray_batch, or_ray_batch, network_fn, network_query_fn, N_samples, retraw=False, lindisp=False, perturb=0.,
                N_importance=0, network_fine=None, white_bkgd=False, raw_noise_std=0., min_max_ray_net=None, refine_net=None,
                N_point_ray_enc=0, embed_fn=None, embeddirs_fn=None, randomize=True, verbose=False, pytest=False, **kwargs):
    """Stage-2 render of a ray batch (refine2.py:525-680).  kwargs consumed: ``images`` [nv,H,W,3], ``poses`` [nv,3,4],
    ``ref_K``, ``num_neighbor``, ``batch_rays_nearest_id`` [N,>=1] (randomize) or ``target_pose`` (evaluation),
    ``train_nerf``, ``iter``; ``embed_rays`` is accepted and ignored (the sampler kernel encodes the rays itself).
    Returns {'rgb_map0', 'rgb_map1', 'depth_map', 'mm_rgb', 'z_vals', 'z_vals0'}."""
    if N_samples != 8 or kwargs['num_neighbor'] != 4 or N_point_ray_enc not in (0, 48):
        raise PnrfError(f"render_rays: kernels are built for N_samples=8, num_neighbor=4, N_point_ray_enc=48 (got {N_samples}, "
                        f"{kwargs['num_neighbor']}, {N_point_ray_enc})")
    dev = ray_batch.device
    N = ray_batch.shape[0]
    sampler, refine, fine = _packed(min_max_ray_net, 'sampler'), _packed(refine_net, 'refine'), _packed(network_fine, 'fine')
    img4, poses, K, rank = _train_views(kwargs['images'], kwargs['poses'], kwargs['ref_K'], dev)
    nv = poses.shape[0]
    depth, _, add, mul, mm_rgb, _ = ops.sampler_fwd(sampler, ray_batch, want_idx=False, want_rgb=True)          # :551-568
    if randomize:                                                                                               # :590-597
        cur = kwargs['batch_rays_nearest_id'][:, 0].long().to(dev)
        order_idx = torch.as_tensor(sorted(random.sample(range(nv - 1), 4)), device=dev)
        ref_nos = rank[cur][:, 1:][:, order_idx]
    else:                                                                                                       # :598-600
        tp = np.asarray(kwargs['target_pose'].detach().cpu() if isinstance(kwargs['target_pose'], torch.Tensor) else kwargs['target_pose'], dtype=np.float32)
        pz = poses.cpu().numpy()
        d = np.sqrt(((tp[None, :3, 3] - pz[:, :3, 3]) ** 2).sum(1, dtype=np.float32))
        ref_nos = torch.from_numpy(np.argsort(d, kind='stable')[:4]).to(dev)[None].expand(N, -1)
    rin = ops.refine_input_train(ray_batch, or_ray_batch, depth, img4, poses, K, ref_nos.contiguous())          # :602-634
    train_nerf = kwargs.get('train_nerf', False)
    jitter, jdir = None, 1
    if train_nerf and randomize:                                                                                # :646-662
        jitter = torch.abs((1 / 5) * torch.normal(0.0, 1.0, size=[N, 8], device=dev)).clamp(max=1 - 2e-6)
        jdir = 1 if random.random() > 0.5 else -1
    z, pts, rgb0 = ops.refine_train_fwd(refine, rin, ray_batch, depth, jitter, jdir)                            # :635-668
    noise = None
    if train_nerf and raw_noise_std > 0.:                                                                       # :670-672, 497
        noise = torch.randn(N, 8, device=dev) * raw_noise_std
    rgbd, _ = ops.nerf_train_fwd(fine, pts, ray_batch, z, add, mul, noise=noise, white_bkgd=white_bkgd)         # :669-676
    return {'rgb_map0': rgb0, 'rgb_map1': rgbd[:, :3], 'depth_map': rgbd[:, 3], 'mm_rgb': mm_rgb,
            'z_vals': z.mean(dim=-1), 'z_vals0': depth.mean(dim=-1)}
