"""Host-side mirror of the reference's stage-2 (refine) trainer (run_S_eS_eN_alter_base_refine2.py): ``render_rays`` (:525-680),
``raw2outputs`` (:475-522) and the training driver ``train`` (:683-1000).

``render_rays`` / ``raw2outputs``: same signature, kwargs and returned dict as the reference; inference-precision forward
through the fused kernels (pnrf_sampler_fwd, pnrf_refine_input_train_fwd, pnrf_refine_train_fwd, pnrf_nerf_train_fwd) without
an autograd graph — what the periodic test-set renders of a training run need.

``train``: the optimisation loop runs on ``ops.Trainer`` (pnrf_train_stage2_fwd_bwd + pnrf_trainer_adam_step: fp32 forward with
saved activations, full backward, Adam — pronerf_amd/csrc/pnrf_train.hip), not on torch.autograd.  The random draws of the
reference (one ``random.sample`` of neighbour ranks and one coin flip per batch, |N(0,1)|/5 jitter, N(0,1) sigma noise) are
made here with the same generators and handed to the kernels as inputs.
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


_VIEWS = []          # at most one entry: (images, its version, poses, its version, device, packed views)


def _train_views(images, poses, ref_K, device):
    """Packed training views of (images, poses).  The entry keeps the caller's objects themselves and is matched by identity (+ the
    in-place version of tensors): id() of a temporary such as ``images[i_train]`` is recycled once it is freed, a referenced object's is not."""
    ver = lambda x: getattr(x, '_version', 0)
    if _VIEWS:
        im, iv, po, pv, dv, ent = _VIEWS[0]
        if im is images and po is poses and iv == ver(images) and pv == ver(poses) and dv == device:
            return ent
    img = torch.as_tensor(images, dtype=torch.float32).to(device)
    img4 = ops.images_pack(img.permute(0, 3, 1, 2).contiguous())                      # [nv,H,W,3] -> [nv,H,W,4]
    pz = torch.as_tensor(poses, dtype=torch.float32).to(device)[:, :3, :4].contiguous()
    K = torch.as_tensor(ref_K, dtype=torch.float32).to(device).reshape(3, 3).contiguous()
    ent = (img4, pz, K, torch.from_numpy(neighbor_rank_table(pz)).to(device))
    _VIEWS[:] = [(images, ver(images), poses, ver(poses), device, ent)]
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


def make_render(render_rays_fn):
    """``render`` of the training scripts (refine2.py:206-279; base.py:215-288) around a given ``render_rays``: rays from ``c2w`` or the
    given (rays_o, rays_d) pair, view directions, the world-space ``or_rays`` batch, NDC, near / far columns, one ``render_rays`` call
    (no chunking: the kernels take the whole batch), outputs reshaped to the ray shape.  Returns [rgb_map0, rgb_map1, depth_map, extras]."""
    from .run_nerf_helpers import get_rays, ndc_rays

    def render(H, W, K, chunk=1024 * 32, rays=None, c2w=None, ndc=True, near=0., far=1., or_near=1., or_far=10., use_viewdirs=False,
               c2w_staticcam=None, **kwargs):
        if not ndc or not use_viewdirs:
            raise PnrfError('render: the HIP path is built for ndc=True, use_viewdirs=True (the LLFF configs)')
        Kt = torch.as_tensor(K, dtype=torch.float32)
        if c2w is not None:
            rays_o, rays_d = get_rays(H, W, Kt, c2w)
        else:
            rays_o, rays_d = rays
        viewdirs = rays_d
        if c2w_staticcam is not None:
            rays_o, rays_d = get_rays(H, W, Kt, c2w_staticcam)
        viewdirs = (viewdirs / torch.norm(viewdirs, dim=-1, keepdim=True)).reshape(-1, 3).float()
        sh = rays_d.shape
        or_o, or_d = rays_o.reshape(-1, 3).float().contiguous(), rays_d.reshape(-1, 3).float().contiguous()
        one = torch.ones_like(or_d[..., :1])
        or_rays = torch.cat([or_o, or_d, or_near * one, or_far * one, viewdirs], -1)
        o, d = ndc_rays(H, W, float(Kt[0][0]), 1., or_o, or_d)
        ray_batch = torch.cat([o.reshape(-1, 3), d.reshape(-1, 3), near * one, far * one, viewdirs], -1)
        ret = render_rays_fn(ray_batch.contiguous(), or_rays.contiguous(), **kwargs)
        ret = {k: v.reshape(list(sh[:-1]) + list(v.shape[1:])) for k, v in ret.items()}
        keys = ['rgb_map0', 'rgb_map1', 'depth_map']
        return [ret[k] for k in keys] + [{k: v for k, v in ret.items() if k not in keys}]
    return render


render = make_render(render_rays)


# ------------------------------------------------------------------------------------ training loop
_MM_KEYS = [f'fc_backbone.{i}' for i in range(6)] + ['fc_output']
_FINE_KEYS = [f'pts_linears.{i}' for i in range(8)] + ['feature_linear', 'alpha_linear', 'views_linears.0', 'rgb_linear']


def trainer_layer_list(sampler_sd, refine_sd, fine_sd):
    """The 26 (W, b) pairs in the order of ``ops.Trainer`` from the three reference-keyed state dicts (MinMaxRay_Net x2, NeRF)."""
    out = []
    for sd, keys in ((sampler_sd, _MM_KEYS), (refine_sd, _MM_KEYS), (fine_sd, _FINE_KEYS)):
        for k in keys:
            out.append((sd[k + '.weight'].detach().float().cpu().numpy(), sd[k + '.bias'].detach().float().cpu().numpy()))
    return out


def state_dicts_from_trainer(tr):
    """Inverse of ``trainer_layer_list``: (sampler_sd, refine_sd, fine_sd) with the reference's keys, CPU tensors."""
    sds = ({}, {}, {})
    li = 0
    for sd, keys in zip(sds, (_MM_KEYS, _MM_KEYS, _FINE_KEYS)):
        for k in keys:
            W, b = tr.read('param', li)
            sd[k + '.weight'], sd[k + '.bias'] = W.cpu(), b.cpu()
            li += 1
    return sds


# order of a reference module's .parameters() (definition order in run_nerf_helpers.py) expressed in trainer layer indices
_FINE_PARAM_ORDER = list(range(14, 22)) + [24, 22, 23, 25]          # pts_linears.0..7, views_linears.0, feature_linear, alpha_linear, rgb_linear
_SAMPLER_PARAM_ORDER = list(range(0, 7))
_REFINE_PARAM_ORDER = list(range(7, 14))


def _import_torch_adam(tr, opt_sd, layer_order, kinds):
    """Moments of a reference ``torch.optim.Adam`` state dict -> the trainer.  ``layer_order``: trainer layer of every (weight,
    bias) pair in the optimizer's parameter order; kinds: ('m', 'v') or ('m_nerf', 'v_nerf').  Returns the optimizer's step."""
    st = opt_sd.get('state', {})
    step = 0
    for k, li in enumerate(layer_order):
        sw, sb = st.get(2 * k), st.get(2 * k + 1)
        if sw is None or sb is None:
            continue
        tr.write(kinds[0], li, sw['exp_avg'].float(), sb['exp_avg'].float())
        tr.write(kinds[1], li, sw['exp_avg_sq'].float(), sb['exp_avg_sq'].float())
        step = max(step, int(sw['step']))
    return step


def restore_optimizer(tr, ck, stage):
    """Adam state of a checkpoint -> trainer: this package's own keys ('pnrf_adam_*', steps in 'pnrf_adam_steps') or the
    reference's torch.optim state dicts (stage 2: 'optimizer_state_dict' = [fine, sampler, refine], refine2.py:358-394; stage 1:
    's_optimizer_state_dict' = [NeRF, sampler, refine] and 'optimizer_state_dict' = [NeRF], base.py:398-422)."""
    if 'pnrf_adam_m' in ck:
        for kind, key in (('m', 'pnrf_adam_m'), ('v', 'pnrf_adam_v'), ('m_nerf', 'pnrf_adam_m_nerf'), ('v_nerf', 'pnrf_adam_v_nerf')):
            if key in ck:
                for li, (W, b) in enumerate(ck[key]):
                    tr.write(kind, li, W, b)
        steps = ck.get('pnrf_adam_steps', (int(ck.get('global_step', 0)), 0))
        tr.set_step(int(steps[0]), int(steps[1]))
        return
    joint_order = _FINE_PARAM_ORDER + _SAMPLER_PARAM_ORDER + _REFINE_PARAM_ORDER
    s_joint = s_nerf = 0
    if stage == 2 and 'optimizer_state_dict' in ck:
        s_joint = _import_torch_adam(tr, ck['optimizer_state_dict'], joint_order, ('m', 'v'))
    if stage == 1:
        if 's_optimizer_state_dict' in ck:
            s_joint = _import_torch_adam(tr, ck['s_optimizer_state_dict'], joint_order, ('m', 'v'))
        if 'optimizer_state_dict' in ck:
            s_nerf = _import_torch_adam(tr, ck['optimizer_state_dict'], _FINE_PARAM_ORDER, ('m_nerf', 'v_nerf'))
    tr.set_step(s_joint, s_nerf)


def newest_checkpoint(out_root):
    """The newest ``*.tar`` of an experiment directory (the reference's resume rule, refine2.py:402-412), or None."""
    import os
    c = sorted(f for f in os.listdir(out_root) if f.endswith('.tar')) if os.path.isdir(out_root) else []
    return os.path.join(out_root, c[-1]) if c else None


def save_checkpoint(path, tr, global_step, steps=None):
    """``torch.save`` of a dict with the reference's keys (refine2.py:884-893).  'network_fn_state_dict' carries the fine net as
    well: the next stage / a restart reads the NeRF from that key (refine2.py:365).  The optimizer state is this trainer's Adam
    moments ('pnrf_adam_m' / 'pnrf_adam_v', one [W, b] pair per layer) instead of torch.optim state dicts."""
    s_sd, r_sd, f_sd = state_dicts_from_trainer(tr)
    adam = {k: [[t.cpu() for t in tr.read(k, li)] for li in range(ops.TRAINER_LAYERS)] for k in ('m', 'v', 'm_nerf', 'v_nerf')}
    torch.save({'global_step': int(global_step), 'network_fn_state_dict': f_sd, 'network_fine_state_dict': f_sd, 'mmr_network_fn_state_dict': s_sd,
                'refine_net_state_dict': r_sd, 'pnrf_adam_m': adam['m'], 'pnrf_adam_v': adam['v'], 'pnrf_adam_m_nerf': adam['m_nerf'],
                'pnrf_adam_v_nerf': adam['v_nerf'], 'pnrf_adam_steps': tuple(int(x) for x in (steps or (global_step, 0)))}, path)


def dist_setup(device):
    """Data-parallel replicas (not in the reference): under ``torchrun`` (WORLD_SIZE > 1) every rank takes GPU LOCAL_RANK, joins the
    process group (nccl = RCCL; PNRF_DIST_BACKEND overrides, e.g. gloo to rehearse on one GPU) and seeds Python's ``random`` alike,
    so the per-batch draws (neighbour ranks, coin flips, n_mult) coincide.  Returns (rank, world, device)."""
    import os
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world == 1:
        return 0, 1, torch.device(device)
    import torch.distributed as dist
    backend = os.environ.get('PNRF_DIST_BACKEND', 'nccl')
    local = int(os.environ.get('LOCAL_RANK', '0'))
    dev = torch.device('cuda', local if backend == 'nccl' else 0)
    torch.cuda.set_device(dev)
    if not dist.is_initialized():
        dist.init_process_group(backend)
    random.seed(20240611)
    return dist.get_rank(), world, dev


def shared_permutation(n, epoch, dev):
    """The same ray permutation on every rank (CPU generator seeded by the epoch)."""
    gen = torch.Generator().manual_seed(1000003 * (epoch + 1))
    return torch.randperm(n, generator=gen).to(dev)


def evaluate_views(tr, stage, test_poses, test_images, train_images, train_poses, K, H, W, savedir=None, near=0., white_bkgd=False):
    """The periodic test-set render of the training scripts (refine2.py:905-917, base.py:984-996): every hold-out view through
    the evaluation branch of the stage's own ``render_rays`` (nearest ``num_neighbor`` training views of the target pose, no
    jitter / exploration / noise) with the trainer's current weights, on the training-time forward kernels.  stage 2: eps 1e-5,
    neighbour-major epi; stage 1: eps 1e-6, sample-major epi, raw clamp 10.  Returns the PSNR of each view; writes
    ``{i:03d}.png`` into ``savedir`` if given."""
    import os
    from .run_S_eS_eN_alter_trt import _write_png
    dev = tr.device
    s_sd, r_sd, f_sd = state_dicts_from_trainer(tr)
    sampler = ops.PackedMLP(ops.NET_SAMPLER, [s_sd[k + '.weight'] for k in _MM_KEYS], [s_sd[k + '.bias'] for k in _MM_KEYS])
    refine = ops.PackedMLP(ops.NET_REFINE, [r_sd[k + '.weight'] for k in _MM_KEYS], [r_sd[k + '.bias'] for k in _MM_KEYS])
    fine = ops.PackedMLP(ops.NET_NERFCLS, [f_sd[k + '.weight'] for k in _FINE_KEYS], [f_sd[k + '.bias'] for k in _FINE_KEYS])
    eps, layout, clamp = (1e-5, 0, 0.0) if stage == 2 else (1e-6, 1, 10.0)
    img4, poses_t, K_t, _ = _train_views(train_images, train_poses, K, dev)
    pt = np.asarray(train_poses, dtype=np.float32)[:, :3, :4]
    psnrs = []
    with torch.cuda.device(dev):
        for i, (c2w, gt) in enumerate(zip(np.asarray(test_poses, dtype=np.float32), np.asarray(test_images, dtype=np.float32))):
            rays, or_rays = ops.frame_rays(np.asarray(K, dtype=np.float32), c2w[:3, :4], H, W, near=near, far=1., device=dev)
            d = np.sqrt(((c2w[None, :3, 3] - pt[:, :3, 3]) ** 2).sum(1, dtype=np.float32))
            ref = np.argsort(d, kind='stable')[:4]                                       # evaluation: the nearest views, in rank order
            ref_nos = torch.as_tensor(ref, dtype=torch.int64, device=dev)[None].expand(rays.shape[0], -1).contiguous()
            depth, _, add, mul, _, _ = ops.sampler_fwd(sampler, rays, want_idx=False, want_rgb=False)
            rin = ops.refine_input_train(rays, or_rays, depth, img4, poses_t, K_t, ref_nos, eps=eps, layout=layout)
            z, pts, _ = ops.refine_train_fwd(refine, rin, rays, depth, want_rgb0=False)
            rgbd, _ = ops.nerf_train_fwd(fine, pts, rays, z, add, mul, clamp=clamp, white_bkgd=white_bkgd)
            rgb = rgbd[:, :3].reshape(H, W, 3)
            mse = float(torch.mean((rgb - torch.as_tensor(gt, device=dev)) ** 2))
            psnrs.append(-10.0 * np.log10(max(mse, 1e-12)))
            if savedir is not None:
                os.makedirs(savedir, exist_ok=True)
                _write_png(os.path.join(savedir, '{:03d}.png'.format(i)), to8b(rgb.cpu().numpy()))
    return psnrs


def config_parser():
    """Options of the stage-2 script (run_S_eS_eN_alter_base_refine2.py:27-160); see ``pronerf_amd.config``."""
    from .config import config_parser as _cp
    return _cp('refine2')


def train(argv=None, device='cuda'):
    """Stage-2 training driver (run_S_eS_eN_alter_base_refine2.py:683-1000): LLFF scene, stage-1 checkpoint (--pretrain_path),
    pre-shuffled ray batches of all training views, one ``Trainer.fwd_bwd`` + ``adam_step`` per iteration with the reference's
    per-batch random draws and learning-rate decay, checkpoints with the reference's keys every ``i_weights``.
    Returns (trainer, list of (iteration, loss, psnr))."""
    import os
    from .load_llff import load_llff_data
    from .render import N_SAMPLES
    args = config_parser().parse_args(argv)
    if args.dataset_type != 'llff':
        raise ValueError('only dataset_type=llff is supported (as in the reference release)')
    if not args.pretrain_path:
        raise ValueError('Stage 2 refinement requires --pretrain_path with a stage 1 checkpoint.')
    if args.no_ndc or args.lindisp:
        raise PnrfError('--no_ndc / --lindisp: the HIP path is built for forward-facing scenes in NDC with samples linear in depth (the LLFF configs)')
    if args.N_samples != N_SAMPLES or args.num_neighbor != 4 or args.N_point_ray_enc != 48 or args.mmnetdepth != 6:
        raise PnrfError('the HIP trainer is built for N_samples=8, num_neighbor=4, N_point_ray_enc=48, mmnetdepth=6 (fern_refine.txt)')
    replica, world, dev = dist_setup(device)
    if args.N_rand % world:
        raise ValueError(f'N_rand = {args.N_rand} is not divisible by the {world} replicas')
    n_local = args.N_rand // world
    images, poses, bds, _, i_test = load_llff_data(args.datadir, args.factor, recenter=True, bd_factor=.75, spherify=args.spherify)
    hwf = poses[0, :3, -1]
    poses = poses[:, :3, :4]
    i_test = np.arange(images.shape[0])[::args.llffhold] if args.llffhold > 0 else np.atleast_1d(i_test)          # :703-706
    i_train = np.array([i for i in np.arange(int(images.shape[0])) if i not in i_test])
    H, W, focal = int(hwf[0]), int(hwf[1]), float(hwf[2])
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]], dtype=np.float32)
    out_root = os.path.join(args.basedir, args.expname or 'pronerf_stage2')
    os.makedirs(out_root, exist_ok=True)
    with open(os.path.join(out_root, 'args.txt'), 'w') as f:
        for k in sorted(vars(args)):
            f.write('{} = {}\n'.format(k, getattr(args, k)))
    ck = torch.load(args.pretrain_path, map_location='cpu')
    start = 0
    fine_sd = ck['network_fn_state_dict']
    resume = None if args.no_reload else (args.ft_path if args.ft_path not in (None, 'None') else newest_checkpoint(out_root))   # :402-412
    if resume is not None:
        ck = torch.load(resume, map_location='cpu')
        start = int(ck.get('global_step', 0))
        fine_sd = ck['network_fine_state_dict']
        print('Reloading from', resume, 'at step', start)
    tr = ops.Trainer(*zip(*trainer_layer_list(ck['mmr_network_fn_state_dict'], ck['refine_net_state_dict'], fine_sd)), max_rays=n_local, device=dev)
    if os.environ.get('PNRF_TRAIN_PRODUCTS'):          # 'f32': exact-fp32 layer products instead of the split-fp16 default (Trainer.set_products)
        tr.set_products(os.environ['PNRF_TRAIN_PRODUCTS'])
    if resume is not None:
        restore_optimizer(tr, ck, 2)
    adam_steps = [start, 0]
    # rays of all training views, as render() prepares them (:206-279): NDC batch + world-space batch, [n_train*H*W, 11] each
    with torch.cuda.device(dev):
        pr = [ops.frame_rays(K, poses[i], H, W, near=0., far=1., device=dev) for i in i_train]
        rays_all = torch.cat([p[0] for p in pr], 0); or_rays_all = torch.cat([p[1] for p in pr], 0)
        del pr
        target_all = torch.as_tensor(images[i_train], dtype=torch.float32).reshape(-1, 3).to(dev)
        own_all = torch.arange(len(i_train), device=dev).repeat_interleave(H * W)
        img4, poses_t, K_t, rank = _train_views(images[i_train], poses[i_train], K, dev)
    n_total = rays_all.shape[0]
    epoch = 0
    perm = shared_permutation(n_total, epoch, dev) if world > 1 else torch.randperm(n_total, device=dev)       # :796-799
    i_batch, global_step, log = 0, start, []
    n_iters = 500000 + 1 if args.max_steps is None else start + args.max_steps + 1                              # :808-810
    # resumed runs continue on the decayed schedule (the reference restores it through optimizer.load_state_dict, :402-412)
    lr = args.lrate * (0.1 ** (start / (args.lrate_decay * 1000)))
    nv = len(i_train)
    for i in range(start + 1, n_iters):
        idx = perm[i_batch:i_batch + args.N_rand]
        i_batch += args.N_rand
        if i_batch >= n_total:                                                                                  # :840-844
            epoch += 1
            perm = shared_permutation(n_total, epoch, dev) if world > 1 else torch.randperm(n_total, device=dev); i_batch = 0
        if idx.shape[0] < args.N_rand:
            continue
        idx = idx[replica * n_local:(replica + 1) * n_local]                                                          # this replica's share
        # the per-batch draws of render_rays (:594-600, :649-661, raw2outputs :497)
        order = torch.as_tensor(sorted(random.sample(range(nv - 1), 4)), device=dev)
        ref_nos = rank[own_all[idx]][:, 1:][:, order].contiguous()
        jitter = torch.abs(torch.normal(0.0, 1.0, size=(idx.shape[0], 8), device=dev) / 5).clamp(max=1 - 2e-6)
        jdir = 1 if random.random() > 0.5 else -1
        noise = torch.randn(idx.shape[0], 8, device=dev) * args.raw_noise_std if args.raw_noise_std > 0 else None
        loss, _ = tr.fwd_bwd(rays_all[idx], or_rays_all[idx], target_all[idx], img4, poses_t, K_t, ref_nos, jitter=jitter, jitter_dir=jdir,
                             raw_noise=noise, white_bkgd=args.white_bkgd, a_mmrgb=args.a_mmrgb, want_rgb=False)
        if world > 1:
            from .dist import allreduce_gradients
            allreduce_gradients(tr)
        tr.adam_step(lr, betas=(0.9, 0.999), weight_decay=args.weight_decay)
        adam_steps[0] += 1
        lr = args.lrate * (0.1 ** (global_step / (args.lrate_decay * 1000)))                                   # :872-878
        if (i % args.i_weights == 0 or i == n_iters - 1) and replica == 0:
            path = os.path.join(out_root, '{:06d}.tar'.format(i))
            save_checkpoint(path, tr, global_step + 1, adam_steps)
            print('Saved checkpoints at', path)
        if i % args.i_testset == 0 and i > 0 and replica == 0:                                                  # :905-917
            ps = evaluate_views(tr, 2, poses[i_test], images[i_test], images[i_train], poses[i_train], K, H, W,
                                savedir=os.path.join(out_root, 'testset_{:06d}'.format(i)), white_bkgd=args.white_bkgd)
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
