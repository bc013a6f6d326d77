"""Host-side mirror of the reference's stage-1 (sampler + NeRF, alternating) trainer functions for the FORWARD of
the path (run_S_eS_eN_alter_base.py): ``render_rays`` (:554-761) and ``raw2outputs`` (:501-551).

Stage-1 specifics relative to stage 2: NDC->metric uses 1e-6; ``epi_features`` are sample-major;
``raw`` is clamped to +-10; on even ("joint", ``train_sampler=True``) steps the learned offsets are added and the
compositing uses the sampler's add/mul; on odd steps (``train_sampler=False`` with ``randomize``) the *exploration
path* replicates each refined depth ``n_mult = randint(1, 64/8)`` times (8..64 samples per ray), jitters them, and
composites without add/mul but with sigma noise.  The random draws are made here with the reference's generators
and handed to the kernels.  ``render_rays`` is forward only; ``train`` (:764-1000) runs the alternating optimisation on
``ops.Trainer`` (pnrf_train_explore_fwd_bwd on odd iterations with the NeRF-only Adam, pnrf_train_stage2_fwd_bwd on even
iterations with the joint Adam).
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


def _make_render():
    from .run_S_eS_eN_alter_base_refine2 import make_render
    return make_render(render_rays)


render = _make_render()          # base.py:215-288


# ------------------------------------------------------------------------------------ training loop
def config_parser():
    """Options of the stage-1 script (run_S_eS_eN_alter_base.py:31-164); see ``pronerf_amd.config``."""
    from .config import config_parser as _cp
    return _cp('base')


def train(argv=None, device='cuda'):
    """Stage-1 training driver (run_S_eS_eN_alter_base.py:764-1000): networks from their default initialisation (or --ft_path),
    odd iterations = NeRF alone on the explored samples (`optimizer`), even iterations = NeRF + sampler + refine net with
    img2mse(rgb1) + img2mse(rgb0) + img2mse(mm_rgb) (`s_optimizer`), learning rate decayed on global_step/2 (:961-967),
    checkpoints with the reference's keys.  Returns (trainer, list of (iteration, loss, psnr))."""
    import os
    from .load_llff import load_llff_data
    from .run_S_eS_eN_alter_base_refine2 import (dist_setup, evaluate_views, newest_checkpoint, restore_optimizer, save_checkpoint,
                                                 shared_permutation, trainer_layer_list)
    args = config_parser().parse_args(argv)
    if args.dataset_type != 'llff':
        raise ValueError('only dataset_type=llff is supported (as in the reference release)')
    if args.epi_nerf:
        raise PnrfError('--epi_nerf references a class the reference does not define (SURVEY.md Appendix B-7)')
    if args.no_ndc or args.lindisp:
        raise PnrfError('--no_ndc / --lindisp: the HIP path is built for forward-facing scenes in NDC with samples linear in depth (the LLFF configs)')
    if args.N_samples != 8 or args.num_neighbor != 4 or args.N_point_ray_enc != 48 or args.mmnetdepth != 6:
        raise PnrfError('the HIP trainer is built for N_samples=8, num_neighbor=4, N_point_ray_enc=48, mmnetdepth=6 (fern_epi.txt)')
    replica, world, dev = dist_setup(device)
    if args.N_rand % world:
        raise ValueError(f'N_rand = {args.N_rand} is not divisible by the {world} replicas')
    n_local = args.N_rand // world
    images, poses, bds, _, i_test = load_llff_data(args.datadir, args.factor, recenter=True, bd_factor=.75, spherify=args.spherify)
    hwf = poses[0, :3, -1]
    poses = poses[:, :3, :4]
    i_test = np.arange(images.shape[0])[::args.llffhold] if args.llffhold > 0 else np.atleast_1d(i_test)
    i_train = np.array([i for i in np.arange(int(images.shape[0])) if i not in i_test])
    H, W, focal = int(hwf[0]), int(hwf[1]), float(hwf[2])
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]], dtype=np.float32)
    out_root = os.path.join(args.basedir, args.expname or 'pronerf_stage1')
    os.makedirs(out_root, exist_ok=True)
    with open(os.path.join(out_root, 'args.txt'), 'w') as f:
        for k in sorted(vars(args)):
            f.write('{} = {}\n'.format(k, getattr(args, k)))
    start, ck = 0, None
    resume = None if args.no_reload else (args.ft_path if args.ft_path not in (None, 'None') else newest_checkpoint(out_root))   # base.py:429-446
    if resume is not None:
        ck = torch.load(resume, map_location='cpu')
        start = int(ck.get('global_step', 0))
        sds = (ck['mmr_network_fn_state_dict'], ck['refine_net_state_dict'], ck['network_fn_state_dict'])
        print('Reloading from', resume, 'at step', start)
    else:                                                # create_nerf (:337-380): torch's default nn.Linear initialisation
        sds = (MinMaxRay_Net(D=6, W=256, input_ch=288, output_ch=27, skips=[10000]).state_dict(),
               MinMaxRay_Net(D=6, W=256, input_ch=144, output_ch=35, skips=[10000]).state_dict(),
               NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4, skips=[4], use_viewdirs=True).state_dict())
    max_mult = 64 // 8                                                                                        # :690
    if world > 1:                                        # replicas start from the same weights: rank 0's initialisation
        import torch.distributed as dist
        box = [sds]
        dist.broadcast_object_list(box, src=0)
        sds = box[0]
    tr = ops.Trainer(*zip(*trainer_layer_list(*sds)), max_rays=n_local, device=dev, max_samples=8 * max_mult)
    if os.environ.get('PNRF_TRAIN_PRODUCTS'):          # 'f32': exact-fp32 layer products instead of the split-fp16 default (Trainer.set_products)
        tr.set_products(os.environ['PNRF_TRAIN_PRODUCTS'])
    if ck is not None:
        restore_optimizer(tr, ck, 1)
    adam_steps = [start - start // 2, start // 2] if ck is not None and 'pnrf_adam_steps' not in ck else ([0, 0] if ck is None else list(ck['pnrf_adam_steps']))
    with torch.cuda.device(dev):
        pr = [ops.frame_rays(K, poses[i], H, W, near=1e-6, far=1., device=dev) for i in i_train]               # near = 1e-6 (:798)
        rays_all = torch.cat([p[0] for p in pr], 0); or_rays_all = torch.cat([p[1] for p in pr], 0)
        del pr
        target_all = torch.as_tensor(images[i_train], dtype=torch.float32).reshape(-1, 3).to(dev)
        own_all = torch.arange(len(i_train), device=dev).repeat_interleave(H * W)
        img4, poses_t, K_t, rank = _train_views(images[i_train], poses[i_train], K, dev)
    n_total = rays_all.shape[0]
    epoch = 0
    perm = shared_permutation(n_total, epoch, dev) if world > 1 else torch.randperm(n_total, device=dev)
    i_batch, global_step, log = 0, start, []
    n_iters = 500000 + 1 if args.max_steps is None else start + args.max_steps + 1
    lr, nv = args.lrate * (0.1 ** ((start / 2) / (args.lrate_decay * 1000))), len(i_train)      # resumed runs: the decayed schedule
    for i in range(start + 1, n_iters):
        idx = perm[i_batch:i_batch + args.N_rand]
        i_batch += args.N_rand
        if i_batch >= n_total:
            epoch += 1
            perm = shared_permutation(n_total, epoch, dev) if world > 1 else torch.randperm(n_total, device=dev); i_batch = 0
        if idx.shape[0] < args.N_rand:
            continue
        idx = idx[replica * n_local:(replica + 1) * n_local]                                                          # this replica's share
        n = idx.shape[0]
        order = torch.as_tensor(sorted(random.sample(range(nv - 1), 4)), device=dev)                           # :629-634
        ref_nos = rank[own_all[idx]][:, 1:][:, order].contiguous()
        batch = (rays_all[idx], or_rays_all[idx], target_all[idx], img4, poses_t, K_t, ref_nos)
        if i % 2 != 0:                                                                                         # :929-940
            n_mult = random.randint(1, max_mult)                                                               # :690-691
            dir1 = (1 if random.random() > 0.5 else -1) if n_mult > 1 else 1
            jitter = torch.abs(torch.normal(0.0, 1.0, size=(n, 8 * n_mult), device=dev) / 5).clamp(max=0.99)   # :715-719
            dir2 = 1 if random.random() > 0.5 else -1
            noise = torch.randn(n, 8 * n_mult, device=dev) * args.raw_noise_std if args.raw_noise_std > 0 else None
            loss, _ = tr.explore_fwd_bwd(*batch, n_mult=n_mult, dir1=dir1, jitter=jitter, dir2=dir2, raw_noise=noise, white_bkgd=args.white_bkgd,
                                         want_rgb=False)
            if world > 1:
                from .dist import allreduce_gradients
                allreduce_gradients(tr)
            tr.adam_step(lr, betas=(0.9, 0.999), weight_decay=args.weight_decay, nerf_only=True)
            adam_steps[1] += 1
        else:                                                                                                  # :941-958
            loss, _ = tr.fwd_bwd(*batch, white_bkgd=args.white_bkgd, eps=1e-6, a_mmrgb=1.0, clamp=10.0, layout=1, want_rgb=False)
            if world > 1:
                from .dist import allreduce_gradients
                allreduce_gradients(tr)
            tr.adam_step(lr, betas=(0.9, 0.999), weight_decay=args.weight_decay)
            adam_steps[0] += 1
        lr = args.lrate * (0.1 ** ((global_step / 2) / (args.lrate_decay * 1000)))                            # :961-967
        if (i % args.i_weights == 0 or i == n_iters - 1) and replica == 0:
            path = os.path.join(out_root, '{:06d}.tar'.format(i))
            save_checkpoint(path, tr, global_step + 1, adam_steps)
            print('Saved checkpoints at', path)
        if i % args.i_testset == 0 and i > 0 and replica == 0:                                                  # :984-996
            ps = evaluate_views(tr, 1, poses[i_test], images[i_test], images[i_train], poses[i_train], K, H, W,
                                savedir=os.path.join(out_root, 'testset_{:06d}'.format(i)), near=1e-6, white_bkgd=args.white_bkgd)
            print(f'[TEST] Iter: {i} PSNR per view: {[round(p, 2) for p in ps]} mean {float(np.mean(ps)):.2f}')
            log.append((i, 'test_psnr', float(np.mean(ps))))
        if i % args.i_print == 0 or i == n_iters - 1:
            lh = loss.cpu().numpy()
            psnr = float(-10.0 * np.log10(max(float(lh[1]), 1e-12)))
            log.append((i, float(lh[0]), psnr))
            if replica == 0:
                print(f'[TRAIN] Iter: {i} Loss: {float(lh[0])}  PSNR: {psnr}')
        global_step += 1
    return tr, log
