#!/usr/bin/env python3
"""Golden of one stage-2 TRAINING ITERATION of the reference (run_S_eS_eN_alter_base_refine2.py): its own render_rays with
autograd enabled, img2mse loss (:861-866), loss.backward(), the Adam it builds in create_nerf (:358-395: three parameter
groups, betas (0.9, 0.999), weight decay) and optimizer.step().  The random draws of render_rays are captured as in
gen_golden.py.  Stored per parameter tensor (trainer order, oracle.trainer_layers): gradient L2 norm, a strided subsample
of the gradient and of the updated parameter — the full tensors would be 12 MB per case.

    python oracle/gen_golden_train.py          -> tests/golden/stage2_step_*.npz
Test infrastructure only; the reference never leaves this container."""
import os
import random as pyrandom
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import gen_golden as G          # noqa: E402
from oracle import synth        # noqa: E402

STRIDE = 97


def module_layers(sampler, refine, fine):
    """(weight, bias) parameters of the three reference modules in the trainer's order."""
    out = []
    for m in (sampler, refine):
        out += [(l.weight, l.bias) for l in m.fc_backbone] + [(m.fc_output.weight, m.fc_output.bias)]
    out += [(l.weight, l.bias) for l in fine.pts_linears]
    out += [(fine.feature_linear.weight, fine.feature_linear.bias), (fine.alpha_linear.weight, fine.alpha_linear.bias),
            (fine.views_linears[0].weight, fine.views_linears[0].bias), (fine.rgb_linear.weight, fine.rgb_linear.bias)]
    return out


def run(helpers, s2, name, seed, H, W, nv, white_bkgd, a_mmrgb, lr=5e-4, wd=5e-8, dtype=torch.float32):
    """dtype float64: the same iteration with every module and input in double precision (torch.set_default_dtype makes the
    reference's own torch.ones / torch.Tensor constants double too) — the accurate gradients the fp32 runs scatter around."""
    torch.set_default_dtype(dtype)
    try:
        _run(helpers, s2, name, seed, H, W, nv, white_bkgd, a_mmrgb, lr, wd, dtype)
    finally:
        torch.set_default_dtype(torch.float32)


def _run(helpers, s2, name, seed, H, W, nv, white_bkgd, a_mmrgb, lr, wd, dtype):
    torch.manual_seed(3407); pyrandom.seed(3407 + seed)
    S, NB, P = synth.N_SAMPLES, synth.NUM_NEIGHBOR, synth.N_POINT_RAY_ENC
    w = synth.make_weights(seed, 'trained')
    wc = synth.make_nerfcls_weights(seed, head_scale=0.3)
    sd = synth.state_dicts(w)
    sampler = helpers.MinMaxRay_Net(D=synth.MMNETDEPTH, W=synth.MMNETWIDTH, input_ch=6 * P, output_ch=3 * S + 3, skips=[10000])
    refine = helpers.MinMaxRay_Net(D=synth.MMNETDEPTH, W=synth.MMNETWIDTH, input_ch=6 * S + 3 * NB * S, output_ch=4 * S + 3, skips=[10000])
    fine = helpers.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4, skips=[4], use_viewdirs=True)
    sampler.load_state_dict(sd['sampler']); refine.load_state_dict(sd['refine']); fine.load_state_dict(synth.nerfcls_state_dict(wc))
    sampler, refine, fine = sampler.to(dtype), refine.to(dtype), fine.to(dtype)
    scene = synth.make_scene(seed, H=H, W=W, n_views=nv, sigma_t=0.2, rotate=True)
    own = 2
    K = torch.from_numpy(scene['K']).to(dtype); poses = torch.from_numpy(scene['poses']).to(dtype); c2w = poses[own]
    rays_o, rays_d = helpers.get_rays(H, W, K, c2w)
    viewdirs = (rays_d / torch.norm(rays_d, dim=-1, keepdim=True)).reshape(-1, 3).to(dtype)
    or_o, or_d = rays_o.reshape(-1, 3).to(dtype), rays_d.reshape(-1, 3).to(dtype)
    N = or_o.shape[0]
    or_rays = torch.cat([or_o, or_d, torch.ones(N, 1), 10 * torch.ones(N, 1), viewdirs], -1)
    o, d = helpers.ndc_rays(H, W, K[0][0], 1., rays_o, rays_d)
    o, d = o.reshape(-1, 3).to(dtype), d.reshape(-1, 3).to(dtype)
    rays = torch.cat([o, d, torch.zeros(N, 1), torch.ones(N, 1), viewdirs], -1)
    if dtype == torch.float64:          # same inputs as the fp32 case: the fp32-rounded rays, promoted
        rays, or_rays = rays.float().double(), or_rays.float().double()
    target = torch.from_numpy(scene['images'][own].reshape(-1, 3).astype(np.float32)).to(dtype)
    embed_fn, _ = helpers.get_embedder(synth.MULTIRES, 0)
    embeddirs_fn, _ = helpers.get_embedder(synth.MULTIRES_VIEWS, 0)
    query = lambda inputs, vd, fn: s2.run_network(inputs, vd, fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn)
    # the optimizer exactly as create_nerf builds it (:358-395)
    grad_vars = [{'params': fine.parameters(), 'weight_decay': wd, 'lr': lr}, {'params': sampler.parameters(), 'weight_decay': wd, 'lr': lr},
                 {'params': refine.parameters(), 'weight_decay': wd, 'lr': lr}]
    optimizer = torch.optim.Adam(params=grad_vars, lr=lr, betas=(0.9, 0.999))
    cap = {}
    o_sample, o_rand, o_normal, o_randn = pyrandom.sample, pyrandom.random, torch.normal, torch.randn

    def p_sample(pop, k):
        r = o_sample(pop, k); cap['order_idx'] = np.array(sorted(r), dtype=np.int64); return r

    def p_random():
        r = o_rand(); cap['coin'] = np.float64(r); return r

    def p_normal(*a, **k):
        r = o_normal(*a, **k); cap['normal'] = r.clone(); return r

    def p_randn(*a, **k):
        r = o_randn(*a, **k); cap['randn'] = r.clone(); return r

    pyrandom.sample, pyrandom.random, torch.normal, torch.randn = p_sample, p_random, p_normal, p_randn
    try:
        ret = s2.render_rays(rays, or_rays, network_fn=None, network_query_fn=query, N_samples=S, network_fine=fine, white_bkgd=white_bkgd,
                             raw_noise_std=1.0, min_max_ray_net=sampler, refine_net=refine, N_point_ray_enc=P, embed_fn=embed_fn,
                             embeddirs_fn=embeddirs_fn, randomize=True, embed_rays=helpers.Pluecker(), images=torch.from_numpy(scene['images']).to(dtype),
                             poses=poses, ref_K=K, num_neighbor=NB, batch_rays_nearest_id=torch.full((N, 1), own, dtype=torch.int64),
                             target_pose=c2w, train_nerf=True, iter=1000)
    finally:
        pyrandom.sample, pyrandom.random, torch.normal, torch.randn = o_sample, o_rand, o_normal, o_randn
    optimizer.zero_grad()
    img_loss = s2.img2mse(ret['rgb_map1'], target)                                   # :861-866
    loss = img_loss
    if a_mmrgb > 0:
        loss = loss + a_mmrgb * (s2.img2mse(ret['rgb_map0'], target) + s2.img2mse(ret['mm_rgb'], target))
    loss.backward()
    layers = module_layers(sampler, refine, fine)
    out = dict(seed=np.int64(seed), H=np.int64(H), W=np.int64(W), nv=np.int64(nv), own=np.int64(own), white_bkgd=np.bool_(white_bkgd),
               a_mmrgb=np.float32(a_mmrgb), lr=np.float32(lr), weight_decay=np.float32(wd), stride=np.int64(STRIDE),
               rays=rays.numpy(), or_rays=or_rays.numpy(), target=target.numpy(), loss=np.float64(loss.item()), img_loss=np.float64(img_loss.item()),
               rgb_map1=ret['rgb_map1'].detach().numpy(), order_idx=cap['order_idx'],
               jitter=torch.abs(cap['normal'] / 5).clamp(max=1 - 2e-6).numpy(), jitter_dir=np.int64(1 if cap['coin'] > 0.5 else -1),
               raw_noise=cap['randn'].numpy())
    for i, (Wt, bt) in enumerate(layers):
        out[f'gW_norm_{i}'] = np.float64(Wt.grad.double().norm().item()); out[f'gb_norm_{i}'] = np.float64(bt.grad.double().norm().item())
        out[f'gW_{i}'] = Wt.grad.reshape(-1)[::STRIDE].numpy().copy(); out[f'gb_{i}'] = bt.grad.numpy().copy()
    optimizer.step()                                                                  # :869
    for i, (Wt, bt) in enumerate(layers):
        out[f'pW_{i}'] = Wt.detach().reshape(-1)[::STRIDE].numpy().copy(); out[f'pb_{i}'] = bt.detach().numpy().copy()
    path = os.path.join(G.OUT, f'{name}.npz')
    np.savez_compressed(path, **out)
    print(f'{name}: N={N} loss={loss.item():.6f} -> {os.path.getsize(path) // 1024} KiB')


def run_stage1(helpers, s1, name, seed, H, W, nv, train_sampler, pyseed, lr=5e-4, wd=5e-8, dtype=torch.float32):
    """One stage-1 iteration of the reference (run_S_eS_eN_alter_base.py:908-958): odd iterations (train_sampler False) update the
    NeRF alone through `optimizer` on the explored samples, even iterations update everything through `s_optimizer` with
    loss = img2mse(rgb1) + img2mse(rgb0) + img2mse(mm_rgb).  Both optimizers are built as create_nerf does (:383-422)."""
    torch.set_default_dtype(dtype)
    try:
        _run_stage1(helpers, s1, name, seed, H, W, nv, train_sampler, pyseed, lr, wd, dtype)
    finally:
        torch.set_default_dtype(torch.float32)


def _run_stage1(helpers, s1, name, seed, H, W, nv, train_sampler, pyseed, lr, wd, dtype):
    torch.manual_seed(3407); pyrandom.seed(pyseed)
    S, NB, P = synth.N_SAMPLES, synth.NUM_NEIGHBOR, synth.N_POINT_RAY_ENC
    w = synth.make_weights(seed, 'trained')
    wc = synth.make_nerfcls_weights(seed, head_scale=0.3)
    sd = synth.state_dicts(w)
    sampler = helpers.MinMaxRay_Net(D=synth.MMNETDEPTH, W=synth.MMNETWIDTH, input_ch=6 * P, output_ch=3 * S + 3, skips=[10000])
    refine = helpers.MinMaxRay_Net(D=synth.MMNETDEPTH, W=synth.MMNETWIDTH, input_ch=6 * S + 3 * NB * S, output_ch=4 * S + 3, skips=[10000])
    fine = helpers.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4, skips=[4], use_viewdirs=True)
    sampler.load_state_dict(sd['sampler']); refine.load_state_dict(sd['refine']); fine.load_state_dict(synth.nerfcls_state_dict(wc))
    sampler, refine, fine = sampler.to(dtype), refine.to(dtype), fine.to(dtype)
    scene = synth.make_scene(seed, H=H, W=W, n_views=nv, sigma_t=0.2, rotate=True)
    own = 1
    K = torch.from_numpy(scene['K']).to(dtype); poses = torch.from_numpy(scene['poses']).to(dtype); c2w = poses[own]
    rays_o, rays_d = helpers.get_rays(H, W, K, c2w)
    viewdirs = (rays_d / torch.norm(rays_d, dim=-1, keepdim=True)).reshape(-1, 3).to(dtype)
    or_o, or_d = rays_o.reshape(-1, 3).to(dtype), rays_d.reshape(-1, 3).to(dtype)
    N = or_o.shape[0]
    or_rays = torch.cat([or_o, or_d, torch.ones(N, 1), 10 * torch.ones(N, 1), viewdirs], -1)
    o, d = helpers.ndc_rays(H, W, K[0][0], 1., rays_o, rays_d)
    o, d = o.reshape(-1, 3).to(dtype), d.reshape(-1, 3).to(dtype)
    rays = torch.cat([o, d, 1e-6 * torch.ones(N, 1), torch.ones(N, 1), viewdirs], -1)          # near = 1e-6 (base.py:798)
    if dtype == torch.float64:
        rays, or_rays = rays.float().double(), or_rays.float().double()
    target = torch.from_numpy(scene['images'][own].reshape(-1, 3).astype(np.float32)).to(dtype)
    embed_fn, _ = helpers.get_embedder(synth.MULTIRES, 0)
    embeddirs_fn, _ = helpers.get_embedder(synth.MULTIRES_VIEWS, 0)
    query = lambda inputs, vd, fn: s1.run_network(inputs, vd, fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn)
    optimizer = torch.optim.Adam(params=[{'params': fine.parameters(), 'weight_decay': wd, 'lr': lr}], lr=lr, betas=(0.9, 0.999))                # :398, 421
    s_optimizer = torch.optim.Adam(params=[{'params': fine.parameters(), 'weight_decay': wd, 'lr': lr}, {'params': sampler.parameters(), 'weight_decay': wd, 'lr': lr},
                                           {'params': refine.parameters(), 'weight_decay': wd, 'lr': lr}], lr=lr, betas=(0.9, 0.999))                  # :406-422
    cap = {'coins': [], 'normal': None, 'randn': None}
    o_sample, o_rand, o_randint, o_normal, o_randn = pyrandom.sample, pyrandom.random, pyrandom.randint, torch.normal, torch.randn

    def p_sample(pop, k):
        r = o_sample(pop, k); cap['order_idx'] = np.array(sorted(r), dtype=np.int64); return r

    def p_random():
        r = o_rand(); cap['coins'].append(r); return r

    def p_randint(a, b):
        r = o_randint(a, b); cap['n_mult'] = r; return r

    def p_normal(*a, **k):
        r = o_normal(*a, **k); cap['normal'] = r.clone(); return r

    def p_randn(*a, **k):
        r = o_randn(*a, **k); cap['randn'] = r.clone(); return r

    pyrandom.sample, pyrandom.random, pyrandom.randint, torch.normal, torch.randn = p_sample, p_random, p_randint, p_normal, p_randn
    try:
        ret = s1.render_rays(rays, or_rays, network_fn=fine, network_query_fn=query, N_samples=S, white_bkgd=False, raw_noise_std=1.0,
                             min_max_ray_net=sampler, refine_net=refine, N_point_ray_enc=P, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn,
                             randomize=True, embed_rays=helpers.Pluecker(), images=torch.from_numpy(scene['images']).to(dtype), poses=poses, ref_K=K,
                             num_neighbor=NB, batch_rays_nearest_id=torch.full((N, 1), own, dtype=torch.int64), target_pose=c2w,
                             train_nerf=True, train_sampler=train_sampler, epi_nerf=False, iter=1000)
    finally:
        pyrandom.sample, pyrandom.random, pyrandom.randint, torch.normal, torch.randn = o_sample, o_rand, o_randint, o_normal, o_randn
    opt = s_optimizer if train_sampler else optimizer
    opt.zero_grad()
    img_loss = s1.img2mse(ret['rgb_map1'], target)
    loss = img_loss + s1.img2mse(ret['rgb_map0'], target) + s1.img2mse(ret['mm_rgb'], target) if train_sampler else img_loss   # :936-955
    loss.backward()
    layers = module_layers(sampler, refine, fine)
    out = dict(seed=np.int64(seed), H=np.int64(H), W=np.int64(W), nv=np.int64(nv), own=np.int64(own), white_bkgd=np.bool_(False),
               train_sampler=np.bool_(train_sampler), lr=np.float32(lr), weight_decay=np.float32(wd), stride=np.int64(STRIDE),
               rays=rays.numpy(), or_rays=or_rays.numpy(), target=target.numpy(), loss=np.float64(loss.item()), img_loss=np.float64(img_loss.item()),
               rgb_map1=ret['rgb_map1'].detach().numpy(), order_idx=cap['order_idx'])
    if not train_sampler:
        n_mult, coins = cap['n_mult'], cap['coins']
        out.update(n_mult=np.int64(n_mult), dir1=np.int64((1 if coins[0] > 0.5 else -1) if n_mult > 1 else 1), dir2=np.int64(1 if coins[-1] > 0.5 else -1),
                   jitter=torch.abs(cap['normal'] / 5).clamp(max=0.99).numpy(), raw_noise=cap['randn'].numpy())
    for i, (Wt, bt) in enumerate(layers):
        gW = Wt.grad if Wt.grad is not None else torch.zeros_like(Wt)
        gb = bt.grad if bt.grad is not None else torch.zeros_like(bt)
        out[f'gW_norm_{i}'] = np.float64(gW.double().norm().item()); out[f'gb_norm_{i}'] = np.float64(gb.double().norm().item())
        out[f'gW_{i}'] = gW.reshape(-1)[::STRIDE].numpy().copy(); out[f'gb_{i}'] = gb.numpy().copy()
    opt.step()
    for i, (Wt, bt) in enumerate(layers):
        out[f'pW_{i}'] = Wt.detach().reshape(-1)[::STRIDE].numpy().copy(); out[f'pb_{i}'] = bt.detach().numpy().copy()
    path = os.path.join(G.OUT, f'{name}.npz')
    np.savez_compressed(path, **out)
    extra = '' if train_sampler else f" n_mult={int(out['n_mult'])} dir1={int(out['dir1'])} dir2={int(out['dir2'])}"
    print(f'{name}: N={N} loss={loss.item():.6f}{extra} -> {os.path.getsize(path) // 1024} KiB')


if __name__ == '__main__':
    helpers, iw, trt = G.load_reference()
    s2 = G.load_stage2()
    run(helpers, s2, 'stage2_step_12x16', 0, 12, 16, 7, False, 0.0)
    run(helpers, s2, 'stage2_step_white_mmrgb_10x14', 1, 10, 14, 6, True, 1.0)
    run(helpers, s2, 'stage2_step_12x16_f64', 0, 12, 16, 7, False, 0.0, dtype=torch.float64)
    run(helpers, s2, 'stage2_step_white_mmrgb_10x14_f64', 1, 10, 14, 6, True, 1.0, dtype=torch.float64)
    s1 = G.load_stage1()
    for dt, sfx in ((torch.float32, ''), (torch.float64, '_f64')):
        run_stage1(helpers, s1, 'stage1_step_joint_12x16' + sfx, 0, 12, 16, 6, True, 11, dtype=dt)
        run_stage1(helpers, s1, 'stage1_step_explore_10x14' + sfx, 2, 10, 14, 7, False, 8, dtype=dt)
