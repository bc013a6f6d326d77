"""Host-side mirror of the reference's stage-1 (sampler + NeRF, alternating) trainer functions for the FORWARD of
the path (run_S_eS_eN_alter_base.py): ``render_rays`` (:554-761) and ``raw2outputs`` (:501-551).

Stage-1 specifics relative to stage 2: NDC->metric uses 1e-6; ``epi_features`` are sample-major;
``raw`` is clamped to +-10; on even ("joint", ``train_sampler=True``) steps the learned offsets are added and the
compositing uses the sampler's add/mul; on odd steps (``train_sampler=False`` with ``randomize``) the *exploration
path* replicates each refined depth ``n_mult = randint(1, 64/8)`` times (8..64 samples per ray), jitters them, and
composites without add/mul but with sigma noise.  The random draws are made here with the reference's generators
and handed to the kernels.  Forward only (see run_S_eS_eN_alter_base_refine2.py in this package).
"""
from __future__ import annotations

import random

import numpy as np
import torch

from . import ops
from .ops import PnrfError
from .run_S_eS_eN_alter_base_refine2 import _packed, _train_views
from .run_nerf_helpers import NeRF, MinMaxRay_Net, Pluecker, get_embedder, img2mse, mse2psnr, to8b  # noqa: F401


def raw2outputs(raw, z_vals, rays_d, raw_noise_std=0, white_bkgd=False, pytest=False, mm_density_add=None, mm_density_mul=None, iter=1e6):
    """-> (rgb_map, disp_map, acc_map, weights, depth_map)  (base.py:501-551): raw clamped to +-10, optional sigma
    noise, optional density modulation, white background."""
    noise = None
    if raw_noise_std > 0.:
        noise = torch.randn(raw[..., 3].shape, device=raw.device) * raw_noise_std
        if pytest:
            np.random.seed(0)
            noise = torch.tensor(np.random.rand(*list(raw[..., 3].shape)) * raw_noise_std, dtype=torch.float32, device=raw.device)
    return ops.composite(raw, z_vals, rays_d, add=mm_density_add, mul=mm_density_mul if mm_density_add is not None else None, noise=noise,
                         clamp=10.0, white_bkgd=white_bkgd)


def render_rays(ray_batch, or_ray_batch, network_fn, network_query_fn, N_samples, retraw=False, lindisp=False, perturb=0.,
                N_importance=0, white_bkgd=False, raw_noise_std=0., min_max_ray_net=None, refine_net=None, N_point_ray_enc=0,
                embed_fn=None, embeddirs_fn=None, randomize=True, verbose=False, pytest=False, **kwargs):
    """Stage-1 render of a ray batch (base.py:554-761).  ``network_fn`` is the NeRF-class fine net.  kwargs consumed:
    ``images, poses, ref_K, num_neighbor, batch_rays_nearest_id | target_pose, train_sampler, train_nerf, epi_nerf, iter``.
    Returns {'rgb_map0', 'rgb_map1', 'depth_map', 'mm_rgb', 'depth_map0'} (+ 'sigma1' when train_sampler)."""
    if N_samples != 8 or kwargs['num_neighbor'] != 4 or N_point_ray_enc not in (0, 48):
        raise PnrfError(f"render_rays: kernels are built for N_samples=8, num_neighbor=4, N_point_ray_enc=48 (got {N_samples}, "
                        f"{kwargs['num_neighbor']}, {N_point_ray_enc})")
    if kwargs.get('epi_nerf', False):
        raise PnrfError('render_rays: epi_nerf=True is not supported (the reference itself references an undefined class there, SURVEY.md Appendix B-7)')
    train_sampler = kwargs.get('train_sampler', False)
    dev = ray_batch.device
    N = ray_batch.shape[0]
    sampler, refine, fine = _packed(min_max_ray_net, 'sampler'), _packed(refine_net, 'refine'), _packed(network_fn, 'fine')
    img4, poses, K, rank = _train_views(kwargs['images'], kwargs['poses'], kwargs['ref_K'], dev)
    nv = poses.shape[0]
    depth, _, add, mul, mm_rgb, _ = ops.sampler_fwd(sampler, ray_batch, want_idx=False, want_rgb=True)          # :586-604
    if randomize:                                                                                               # :627-633
        cur = kwargs['batch_rays_nearest_id'][:, 0].long().to(dev)
        order_idx = torch.as_tensor(sorted(random.sample(range(nv - 1), 4)), device=dev)
        ref_nos = rank[cur][:, 1:][:, order_idx]
    else:                                                                                                       # :634-636
        tp = np.asarray(kwargs['target_pose'].detach().cpu() if isinstance(kwargs['target_pose'], torch.Tensor) else kwargs['target_pose'], dtype=np.float32)
        d = np.sqrt(((tp[None, :3, 3] - poses.cpu().numpy()[:, :3, 3]) ** 2).sum(1, dtype=np.float32))
        ref_nos = torch.from_numpy(np.argsort(d, kind='stable')[:4]).to(dev)[None].expand(N, -1)
    rin = ops.refine_input_train(ray_batch, or_ray_batch, depth, img4, poses, K, ref_nos.contiguous(), eps=1e-6, layout=1)   # :607, 638-673
    z8, pts8, rgb0 = ops.refine_train_fwd(refine, rin, ray_batch, depth)                                        # :675-687
    if randomize and not train_sampler:                                                                         # :689-729 exploration
        n_mult = random.randint(1, int(64 / N_samples))
        dir1 = (1 if random.random() > 0.5 else -1) if n_mult > 1 else 1
        jitter = torch.abs((1 / 5) * torch.normal(0.0, 1.0, size=[N, 8 * n_mult], device=dev)).clamp(max=0.99)
        dir2 = 1 if random.random() > 0.5 else -1
        z, pts = ops.explore(z8, ray_batch, jitter, n_mult, dir1, dir2)
    elif train_sampler:
        z, pts = z8, pts8                                                                                       # :735-736 offsets added
    else:                                                                                                       # evaluation without offsets
        z, pts = ops.explore(z8, ray_batch, torch.zeros(N, 8, device=dev), 1, 1, 1)
    S = z.shape[1]
    if train_sampler:                                                                                           # :743-747
        rgbd, raw = ops.nerf_train_fwd(fine, pts, ray_batch, z, add, mul, clamp=10.0, white_bkgd=white_bkgd, want_raw=True)
        rgb_map, depth_map = rgbd[:, :3], rgbd[:, 3]
    else:                                                                                                       # :748-751
        noise = torch.randn(N, S, device=dev) * raw_noise_std if raw_noise_std > 0. else None
        if S == 8:
            rgbd, raw = ops.nerf_train_fwd(fine, pts, ray_batch, z, None, None, noise=noise, clamp=10.0, white_bkgd=white_bkgd, want_raw=True)
            rgb_map, depth_map = rgbd[:, :3], rgbd[:, 3]
        else:
            _, raw = ops.nerf_train_fwd(fine, pts, ray_batch)
            rgb_map, _, _, _, depth_map = ops.composite(raw, z, ray_batch[:, 3:6].contiguous(), noise=noise, clamp=10.0, white_bkgd=white_bkgd)
    ret = {'rgb_map0': rgb0, 'rgb_map1': rgb_map, 'depth_map': depth_map, 'mm_rgb': mm_rgb, 'depth_map0': z.mean(dim=-1)}
    if train_sampler:
        ret['sigma1'] = raw[..., 3]
    return ret
