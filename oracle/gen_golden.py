"""TEST INFRASTRUCTURE ONLY — golden-vector generator.

Runs in the BUILD CONTAINER only: imports the reference Python from /root/reference
(CPU, fp32), feeds it the seeded synthetic inputs of ``oracle/synth.py`` and stores
inputs + outputs + captured intermediates as small ``tests/golden/*.npz`` fixtures.
The reference itself never travels; the GPU box only sees these data files.

Import recipe: ``torchvision``, ``imageio`` and ``cv2`` are imported
at module top level by the reference but never used on this path, so empty module
objects stand in for them; the driver script is loaded with importlib because its
``train()`` is guarded by ``__name__ == '__main__'``.

Usage:  python oracle/gen_golden.py            (writes tests/golden/)
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import synth  # noqa: E402

REF = '/root/reference'
OUT = os.path.join(ROOT, 'tests', 'golden')


def load_reference():
    sys.dont_write_bytecode = True
    for m in ('torchvision', 'torchvision.models', 'imageio', 'cv2'):
        sys.modules.setdefault(m, types.ModuleType(m))
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import run_nerf_helpers as helpers
    import inverse_warp
    spec = importlib.util.spec_from_file_location('ref_trt', os.path.join(REF, 'run_S_eS_eN_alter_trt.py'))
    trt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(trt)
    return helpers, inverse_warp, trt


def driver_K(scene):
    """The intrinsics as the reference's drivers hold them when they set up rays: a float64 numpy array built from the float32 focal of
    poses_bounds.npy (run_S_eS_eN_alter_trt.py:737-747; passed as it is to render_path, :798).  K[0][0] is then a numpy float64 scalar, so
    ndc_rays evaluates -1./(W/(2.*focal)) in double and rounds the factor to fp32 once; with a 0-dim fp32 TENSOR in its place torch would
    evaluate W / t as t.reciprocal() * W — another rounding, 1 ulp apart for most image sizes (equal at 756 x 1008)."""
    return np.asarray(scene['K'], dtype=np.float32).astype(np.float64)


FERN_SHAPE = dict(n_pts=synth.N_POINT_RAY_ENC, mmnetdepth=synth.MMNETDEPTH, num_neighbor=synth.NUM_NEIGHBOR, netdepth=synth.NETDEPTH)


def build_models(helpers, weights, shape=FERN_SHAPE):
    """The reference's modules as its create_nerf builds them from --mmnetdepth / --N_point_ray_enc / --num_neighbor / --netdepth
    (run_S_eS_eN_alter_trt.py:427-457)."""
    S, NB = synth.N_SAMPLES, shape['num_neighbor']
    sd = synth.state_dicts(weights)
    sampler = helpers.MinMaxRaySamplerTRT_Net(D=shape['mmnetdepth'], W=synth.MMNETWIDTH, input_ch=6 * shape['n_pts'],
                                              output_ch=3 * S + 3, skips=[10000], N_samples=S)
    refine = helpers.MinMaxRayEpiSamplerTRT_Net(D=shape['mmnetdepth'], W=synth.MMNETWIDTH, input_ch=6 * S + 3 * NB * S,
                                                output_ch=4 * S + 3, skips=[10000], N_samples=S)
    nerf = helpers.DoNeRFTRT(D=shape['netdepth'], W=synth.NETWIDTH, n_in=synth.POS_CH + synth.DIR_CH, n_out=4, skip='auto')
    sampler.load_state_dict(sd['sampler']); refine.load_state_dict(sd['refine']); nerf.load_state_dict(sd['nerf'])
    return sampler.eval(), refine.eval(), nerf.eval()


def reference_frame(helpers, trt, scene, shape=FERN_SHAPE):
    """The reference's per-frame setup, op for op, through the reference's own helpers
    (render_path, run_S_eS_eN_alter_trt.py:245-302; render_path itself needs a GPU)."""
    S, NB, P = synth.N_SAMPLES, shape['num_neighbor'], shape['n_pts']
    H, W = scene['H'], scene['W']
    K = torch.from_numpy(scene['K']); c2w = torch.from_numpy(scene['c2w']); Kd = driver_K(scene)
    poses = torch.from_numpy(scene['poses']); images = scene['images']
    rays_o, rays_d = helpers.get_rays(H, W, Kd, c2w)
    viewdirs = rays_d / torch.norm(rays_d, dim=-1, keepdim=True)
    viewdirs = torch.reshape(viewdirs, [-1, 3]).float()
    or_o = torch.reshape(rays_o, [-1, 3]).float(); or_d = torch.reshape(rays_d, [-1, 3]).float()
    or_rays = torch.cat([or_o, or_d, 1.0 * torch.ones_like(or_d[..., :1]), 10.0 * torch.ones_like(or_d[..., :1]), viewdirs], -1)
    ro1 = torch.cat([or_o.t()[None], torch.ones(1, 1, or_o.shape[0])], 1).expand(S * NB, -1, -1)
    rd1 = torch.cat([or_d.t()[None], torch.zeros(1, 1, or_d.shape[0])], 1).expand(S * NB, -1, -1)
    o, d = helpers.ndc_rays(H, W, Kd[0][0], 1., rays_o, rays_d)
    o = torch.reshape(o, [-1, 3]).float(); d = torch.reshape(d, [-1, 3]).float()
    rays = torch.cat([o, d, 0. * torch.ones_like(d[..., :1]), 1. * torch.ones_like(d[..., :1]), viewdirs], -1)
    embed_rays = helpers.Pluecker()
    pts, _ = trt.compute_query_points_from_rays(o, d, 0., 1., P, randomize=False)
    mm_input = embed_rays(pts, d[:, None, :].expand(-1, P, -1)).view(-1, P * 6)
    dist = torch.sum((c2w[None, :, 3] - poses[:, :, 3]) ** 2, 1) ** (1 / 2)
    ref_nos = torch.sort(dist, dim=0)[1][:NB]
    nb = torch.Tensor(images)[ref_nos]
    ref_pose = poses[ref_nos]
    flip = torch.eye(3); flip[1, 1] = -1; flip[2, 2] = -1
    pm = torch.bmm(flip[None].expand(NB, -1, -1), ref_pose)
    pm = torch.bmm(K[None].expand(NB, -1, -1), pm)
    ref_rgb = nb.permute(0, 3, 1, 2)
    sh = ref_rgb.shape
    ref_rgb_rep = ref_rgb.unsqueeze(1).expand(-1, S, -1, -1, -1).contiguous().view(sh[0] * S, sh[1], sh[2], sh[3])
    ref_pose_rep = pm.unsqueeze(1).expand(-1, S, -1, -1).contiguous().view(NB * S, 3, 4)
    return dict(rays=rays, or_rays=or_rays, ro1=ro1, rd1=rd1, mm_input=mm_input, ref_nos=ref_nos,
                ref_rgb=ref_rgb_rep, ref_pose=ref_pose_rep, images=ref_rgb.contiguous(), proj=pm,
                embed_rays=embed_rays, rays_o=rays_o, rays_d=rays_d)


def run_infer_case(helpers, iw, trt, name, seed, kind, H, W, Hf=None, Wf=None, rotate=False, sigma_t=0.05, take=None, sel=None, slim=False, shape=FERN_SHAPE,
                   n_views=None):
    """shape: the reference's free shape arguments (FERN_SHAPE = fern_trt.txt); n_views: size of the neighbour pool (default: num_neighbor)."""
    torch.manual_seed(3407)
    weights = synth.make_weights(seed, kind, **shape)
    scene = synth.make_scene(seed, H=H, W=W, Hf=Hf, Wf=Wf, rotate=rotate, sigma_t=sigma_t, n_views=n_views or shape['num_neighbor'])
    sampler, refine, nerf = build_models(helpers, weights, shape)
    fr = reference_frame(helpers, trt, scene, shape)
    N_full = fr['rays'].shape[0]
    if sel is None:
        sel = np.arange(N_full) if take is None else np.linspace(0, N_full - 1, take).astype(np.int64)
    else:
        take = len(sel)
    if take is not None:      # render only the selected rays of the frame (render_rays is per-ray independent)
        st = torch.from_numpy(sel)
        for k in ('rays', 'or_rays', 'mm_input'):
            fr[k] = fr[k][st].contiguous()
        fr['ro1'] = fr['ro1'][:, :, st]; fr['rd1'] = fr['rd1'][:, :, st]
    embed_fn, _ = helpers.get_embedder(synth.MULTIRES, 0)
    embeddirs_fn, _ = helpers.get_embedder(synth.MULTIRES_VIEWS, 0)
    query = lambda inputs, viewdirs, fn: trt.run_network(inputs, viewdirs, fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn)

    cap = {}
    h1 = sampler.register_forward_hook(lambda m, i, o: cap.update(mm_rgb=o[0], add=o[1], mul=o[2], depth_raw=o[3]))
    h2 = refine.register_forward_hook(lambda m, i, o: cap.update(refine_in=i[0], refine_depth=o[0], offsets=o[2]))
    h3 = nerf.register_forward_hook(lambda m, i, o: cap.update(emb_pts=i[0], emb_dirs=i[1], raw_flat=o))
    orig_warp = iw.inverse_warp_rod1_rt2_coords_trt
    orig_r2o = trt.raw2outputs
    orig_sort = torch.sort

    def warp(*a, **k):
        r = orig_warp(*a, **k); cap['warps'] = r[0]; return r

    def r2o(raw, z, rd, *a, **k):
        r = orig_r2o(raw, z, rd, *a, **k)
        cap.update(raw=raw, z=z, add_sorted=k['mm_density_add'], mul_sorted=k['mm_density_mul'],
                   disp=r[1], acc=r[2], weights=r[3])
        return r

    def sort(x, *a, **k):
        r = orig_sort(x, *a, **k); cap['depth_sorted'] = r[0]; cap['sort_idx'] = r[1]; return r

    iw.inverse_warp_rod1_rt2_coords_trt = warp; trt.raw2outputs = r2o; torch.sort = sort
    try:
        with torch.no_grad():
            ret = trt.render_rays(fr['rays'], fr['or_rays'], network_fn=None, network_query_fn=query,
                                  N_samples=synth.N_SAMPLES, network_fine=nerf, min_max_ray_net=sampler,
                                  refine_net=refine, N_point_ray_enc=shape['n_pts'], embed_fn=embed_fn,
                                  embeddirs_fn=embeddirs_fn, randomize=False, raw_noise_std=0., perturb=False,
                                  use_trt=False, mm_input=fr['mm_input'], num_neighbor=shape['num_neighbor'],
                                  ref_rgb=fr['ref_rgb'], ref_pose=fr['ref_pose'], ro1=fr['ro1'], rd1=fr['rd1'],
                                  embed_rays=fr['embed_rays'])
    finally:
        iw.inverse_warp_rod1_rt2_coords_trt = orig_warp; trt.raw2outputs = orig_r2o; torch.sort = orig_sort
        h1.remove(); h2.remove(); h3.remove()

    N = fr['rays'].shape[0]
    g = lambda t: t.detach().cpu().numpy()
    S = synth.N_SAMPLES
    epi = g(cap['refine_in'])[:, 6 * S:]
    out = dict(
        seed=np.int64(seed), kind=np.array(kind), H=np.int64(H), W=np.int64(W),
        Hf=np.int64(scene['images'].shape[1]), Wf=np.int64(scene['images'].shape[2]),
        rotate=np.bool_(rotate), sigma_t=np.float64(sigma_t), sel=sel, n_full=np.int64(N_full),
        n_pts=np.int64(shape['n_pts']), mmnetdepth=np.int64(shape['mmnetdepth']), num_neighbor=np.int64(shape['num_neighbor']), netdepth=np.int64(shape['netdepth']),
        n_views=np.int64(scene['poses'].shape[0]),
        rays=g(fr['rays']), or_rays=g(fr['or_rays']), ref_nos=g(fr['ref_nos']), proj=g(fr['proj']),
        mm_input_head=g(fr['mm_input'])[:, :12], mm_input_tail=g(fr['mm_input'])[:, -6:],
        mm_rgb=g(cap['mm_rgb']), depth_raw=g(cap['depth_raw']),
        depth_sorted=g(cap['depth_sorted']), sort_idx=g(cap['sort_idx']).astype(np.int64),
        add_sorted=g(cap['add_sorted']), mul_sorted=g(cap['mul_sorted']),
        epi=epi, plucker8=g(cap['refine_in'])[:, :6 * S],
        refine_depth=g(cap['refine_depth']), offsets=g(cap['offsets']),
        z=g(cap['z']), raw=g(cap['raw']),
        emb_pts_first=g(cap['emb_pts']).reshape(N, S, -1)[:, 0, :], emb_dirs_first=g(cap['emb_dirs']).reshape(N, S, -1)[:, 0, :],
        rgb=g(ret['rgb_map1']), depth=g(ret['depth_map']),
        disp=g(cap['disp']), acc=g(cap['acc']), weights=g(cap['weights']),
    )
    if slim:                  # large ray sets: inputs, the sampler's decision and the final outputs only
        oob = (epi.reshape(N, -1, 3) == 0).all(-1).sum(-1).astype(np.uint8)       # (view, sample) taps that fell outside their source image
        keep = ('seed', 'kind', 'H', 'W', 'Hf', 'Wf', 'rotate', 'sigma_t', 'sel', 'n_full', 'rays', 'ref_nos', 'proj', 'depth_sorted', 'z', 'rgb', 'depth')
        out = {k: out[k] for k in keep}
        out['sort_idx'] = g(cap['sort_idx']).astype(np.int8)
        out['oob_taps'] = oob
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, f'{name}.npz')
    np.savez_compressed(path, **out)
    print(f'{name}: N={N} kept={len(sel)} rgb mean={out["rgb"].mean():.4f} std={out["rgb"].std():.4f} '
          f'min sorted-depth gap={np.diff(out["depth_sorted"], axis=1).min():.3e} -> {os.path.getsize(path)/1024:.0f} KiB')


def run_operator_cases(helpers, iw, trt):
    """Operator-level goldens: embedder, Pluecker, rays/NDC, warp, raw2outputs, NeRF class."""
    rs = np.random.RandomState(7)
    out = {}
    x = torch.from_numpy(rs.uniform(-1.5, 1.5, (257, 3)).astype(np.float32))
    e10, _ = helpers.get_embedder(10, 0); e4, _ = helpers.get_embedder(4, 0)
    out['pe_x'] = x.numpy(); out['pe10'] = e10(x).numpy(); out['pe4'] = e4(x).numpy()
    o = torch.from_numpy(rs.randn(129, 3).astype(np.float32)); d = torch.from_numpy(rs.randn(129, 3).astype(np.float32))
    out['pl_o'] = o.numpy(); out['pl_d'] = d.numpy(); out['pl'] = helpers.Pluecker()(o, d).numpy()
    scene = synth.make_scene(5, H=9, W=13, rotate=True)
    Kd = driver_K(scene); c2w = torch.from_numpy(scene['c2w'])
    ro, rd = helpers.get_rays(9, 13, Kd, c2w)
    no, nd = helpers.ndc_rays(9, 13, Kd[0][0], 1., ro, rd)
    out['gr_K'] = scene['K']; out['gr_c2w'] = scene['c2w']
    out['gr_o'] = ro.contiguous().numpy(); out['gr_d'] = rd.numpy(); out['ndc_o'] = no.numpy(); out['ndc_d'] = nd.numpy()
    # warp with out-of-range samples: coordinates deliberately span beyond the image
    B, Hf, Wf, n = 3, 11, 17, 301
    img = torch.from_numpy(rs.rand(B, 3, Hf, Wf).astype(np.float32))
    ro1 = torch.cat([torch.from_numpy(rs.randn(1, 3, n).astype(np.float32)) * 0.1, torch.ones(1, 1, n)], 1).expand(B, -1, -1)
    rd1 = torch.cat([torch.from_numpy(np.concatenate([rs.randn(1, 2, n) * 0.6, -np.ones((1, 1, n))], 1).astype(np.float32)), torch.zeros(1, 1, n)], 1).expand(B, -1, -1)
    Kf = torch.tensor([[12.0, 0, Wf / 2], [0, 12.0, Hf / 2], [0, 0, 1]])
    flip = torch.diag(torch.tensor([1.0, -1.0, -1.0]))
    pose = torch.cat([torch.eye(3)[None].expand(B, -1, -1), torch.from_numpy(rs.randn(B, 3, 1).astype(np.float32)) * 0.1], 2)
    w2c = Kf[None] @ (flip[None] @ pose)
    depth = torch.from_numpy(rs.uniform(1.0, 30.0, (B, 1, n)).astype(np.float32))
    warped, _ = iw.inverse_warp_rod1_rt2_coords_trt(img, depth, ro1, rd1, w2c)
    out['wp_img'] = img.numpy(); out['wp_ro1'] = ro1[0].numpy(); out['wp_rd1'] = rd1[0].numpy()
    out['wp_w2c'] = w2c.numpy(); out['wp_depth'] = depth.numpy(); out['wp_out'] = warped.numpy()
    # raw2outputs (infer variant)
    N, S = 203, 8
    raw = torch.from_numpy((rs.randn(N, S, 4) * 2).astype(np.float32))
    z = torch.sort(torch.from_numpy(rs.rand(N, S).astype(np.float32)), -1)[0]
    rdn = torch.from_numpy(rs.randn(N, 3).astype(np.float32))
    add = torch.from_numpy(rs.randn(N, S).astype(np.float32) + 1); mul = torch.from_numpy(rs.randn(N, S).astype(np.float32) + 0.5)
    r = trt.raw2outputs(raw, z, rdn, 0., False, mm_density_add=add, mm_density_mul=mul)
    out.update(c_raw=raw.numpy(), c_z=z.numpy(), c_d=rdn.numpy(), c_add=add.numpy(), c_mul=mul.numpy(),
               c_rgb=r[0].numpy(), c_disp=r[1].numpy(), c_acc=r[2].numpy(), c_w=r[3].numpy(), c_depth=r[4].numpy())
    # NeRF class fine net (the class stage 1/2 train, run_nerf_helpers.py:792-847)
    wc = synth.make_nerfcls_weights(0)
    m = helpers.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4, skips=[4], use_viewdirs=True)
    m.load_state_dict(synth.nerfcls_state_dict(wc))
    xin = torch.from_numpy(rs.uniform(-1, 1, (64, 90)).astype(np.float32))
    with torch.no_grad():
        out['nc_x'] = xin.numpy(); out['nc_y'] = m(xin).numpy()
    path = os.path.join(OUT, 'operators.npz')
    np.savez_compressed(path, **out)
    print('operators ->', os.path.getsize(path) // 1024, 'KiB')


def load_stage2():
    spec = importlib.util.spec_from_file_location('ref_s2', os.path.join(REF, 'run_S_eS_eN_alter_base_refine2.py'))
    s2 = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(s2)
    return s2


def run_stage2_case(helpers, s2, name, seed, H, W, nv, randomize, white_bkgd, sigma_t=0.2):
    """Stage-2 training-time render_rays of the reference (run_S_eS_eN_alter_base_refine2.py:525-680) on seeded inputs;
    the random draws it makes (random.sample / random.random / torch.normal / torch.randn) are captured as inputs."""
    import random as pyrandom
    torch.manual_seed(3407); pyrandom.seed(3407 + seed)
    S, NB, P = synth.N_SAMPLES, synth.NUM_NEIGHBOR, synth.N_POINT_RAY_ENC
    w = synth.make_weights(seed, 'trained')
    wc = synth.make_nerfcls_weights(seed, head_scale=0.3)
    sd = synth.state_dicts(w)
    sampler = helpers.MinMaxRay_Net(D=synth.MMNETDEPTH, W=synth.MMNETWIDTH, input_ch=6 * P, output_ch=3 * S + 3, skips=[10000])
    refine = helpers.MinMaxRay_Net(D=synth.MMNETDEPTH, W=synth.MMNETWIDTH, input_ch=6 * S + 3 * NB * S, output_ch=4 * S + 3, skips=[10000])
    fine = helpers.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4, skips=[4], use_viewdirs=True)
    sampler.load_state_dict(sd['sampler']); refine.load_state_dict(sd['refine']); fine.load_state_dict(synth.nerfcls_state_dict(wc))
    scene = synth.make_scene(seed, H=H, W=W, n_views=nv, sigma_t=sigma_t, rotate=True)
    own = 2                                             # the rays come from training view `own`
    K = torch.from_numpy(scene['K']); poses = torch.from_numpy(scene['poses']); c2w = poses[own]; Kd = driver_K(scene)
    rays_o, rays_d = helpers.get_rays(H, W, Kd, c2w)
    viewdirs = (rays_d / torch.norm(rays_d, dim=-1, keepdim=True)).reshape(-1, 3).float()
    or_o, or_d = rays_o.reshape(-1, 3).float(), rays_d.reshape(-1, 3).float()
    N = or_o.shape[0]
    or_rays = torch.cat([or_o, or_d, torch.ones(N, 1), 10 * torch.ones(N, 1), viewdirs], -1)
    o, d = helpers.ndc_rays(H, W, Kd[0][0], 1., rays_o, rays_d)
    o, d = o.reshape(-1, 3).float(), d.reshape(-1, 3).float()
    rays = torch.cat([o, d, 1e-6 * torch.ones(N, 1) * 0, torch.ones(N, 1), viewdirs], -1)     # near 0, far 1 (refine2.py:794-795 / fern_refine.txt)
    embed_fn, _ = helpers.get_embedder(synth.MULTIRES, 0)
    embeddirs_fn, _ = helpers.get_embedder(synth.MULTIRES_VIEWS, 0)
    query = lambda inputs, vd, fn: s2.run_network(inputs, vd, fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn)
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
        with torch.no_grad():
            ret = s2.render_rays(rays, or_rays, network_fn=None, network_query_fn=query, N_samples=S, network_fine=fine,
                                 white_bkgd=white_bkgd, raw_noise_std=1.0, min_max_ray_net=sampler, refine_net=refine, N_point_ray_enc=P,
                                 embed_fn=embed_fn, embeddirs_fn=embeddirs_fn, randomize=randomize, embed_rays=helpers.Pluecker(),
                                 images=torch.from_numpy(scene['images']), poses=poses, ref_K=K, num_neighbor=NB,
                                 batch_rays_nearest_id=torch.full((N, 1), own, dtype=torch.int64), target_pose=c2w,
                                 train_nerf=True, iter=1000)
    finally:
        pyrandom.sample, pyrandom.random, torch.normal, torch.randn = o_sample, o_rand, o_normal, o_randn
    g = lambda t: t.detach().cpu().numpy()
    out = dict(seed=np.int64(seed), H=np.int64(H), W=np.int64(W), nv=np.int64(nv), own=np.int64(own), randomize=np.bool_(randomize),
               white_bkgd=np.bool_(white_bkgd), sigma_t=np.float64(sigma_t), rays=g(rays), or_rays=g(or_rays),
               rgb_map0=g(ret['rgb_map0']), rgb_map1=g(ret['rgb_map1']), depth_map=g(ret['depth_map']), mm_rgb=g(ret['mm_rgb']),
               z_vals=g(ret['z_vals']), z_vals0=g(ret['z_vals0']))
    if randomize:
        jit = torch.abs(cap['normal'] / 5).clamp(max=1 - 2e-6)
        out.update(order_idx=cap['order_idx'], jitter=g(jit), jitter_dir=np.int64(1 if cap['coin'] > 0.5 else -1))
    if 'randn' in cap:
        out.update(raw_noise=g(cap['randn']))
    np.savez_compressed(os.path.join(OUT, f'{name}.npz'), **out)
    print(f'{name}: N={N} rgb mean={out["rgb_map1"].mean():.4f} -> {os.path.getsize(os.path.join(OUT, name + ".npz")) // 1024} KiB')


def load_stage1():
    spec = importlib.util.spec_from_file_location('ref_s1', os.path.join(REF, 'run_S_eS_eN_alter_base.py'))
    s1 = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(s1)
    return s1


def run_stage1_case(helpers, s1, name, seed, H, W, nv, train_sampler, pyseed, sigma_t=0.2):
    """Stage-1 training-time render_rays of the reference (run_S_eS_eN_alter_base.py:554-761); odd steps
    (train_sampler=False) take the exploration path.  Random draws are captured as inputs."""
    import random as pyrandom
    torch.manual_seed(3407); pyrandom.seed(pyseed)
    S, NB, P = synth.N_SAMPLES, synth.NUM_NEIGHBOR, synth.N_POINT_RAY_ENC
    w = synth.make_weights(seed, 'trained')
    wc = synth.make_nerfcls_weights(seed, head_scale=0.3)
    sd = synth.state_dicts(w)
    sampler = helpers.MinMaxRay_Net(D=synth.MMNETDEPTH, W=synth.MMNETWIDTH, input_ch=6 * P, output_ch=3 * S + 3, skips=[10000])
    refine = helpers.MinMaxRay_Net(D=synth.MMNETDEPTH, W=synth.MMNETWIDTH, input_ch=6 * S + 3 * NB * S, output_ch=4 * S + 3, skips=[10000])
    fine = helpers.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4, skips=[4], use_viewdirs=True)
    sampler.load_state_dict(sd['sampler']); refine.load_state_dict(sd['refine']); fine.load_state_dict(synth.nerfcls_state_dict(wc))
    scene = synth.make_scene(seed, H=H, W=W, n_views=nv, sigma_t=sigma_t, rotate=True)
    own = 1
    K = torch.from_numpy(scene['K']); poses = torch.from_numpy(scene['poses']); c2w = poses[own]; Kd = driver_K(scene)
    rays_o, rays_d = helpers.get_rays(H, W, Kd, c2w)
    viewdirs = (rays_d / torch.norm(rays_d, dim=-1, keepdim=True)).reshape(-1, 3).float()
    or_o, or_d = rays_o.reshape(-1, 3).float(), rays_d.reshape(-1, 3).float()
    N = or_o.shape[0]
    or_rays = torch.cat([or_o, or_d, torch.ones(N, 1), 10 * torch.ones(N, 1), viewdirs], -1)
    o, d = helpers.ndc_rays(H, W, Kd[0][0], 1., rays_o, rays_d)
    o, d = o.reshape(-1, 3).float(), d.reshape(-1, 3).float()
    rays = torch.cat([o, d, 1e-6 * torch.ones(N, 1), torch.ones(N, 1), viewdirs], -1)          # near = 1e-6 (base.py:798)
    embed_fn, _ = helpers.get_embedder(synth.MULTIRES, 0)
    embeddirs_fn, _ = helpers.get_embedder(synth.MULTIRES_VIEWS, 0)
    query = lambda inputs, vd, fn: s1.run_network(inputs, vd, fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn)
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
        with torch.no_grad():
            ret = s1.render_rays(rays, or_rays, network_fn=fine, network_query_fn=query, N_samples=S, white_bkgd=False, raw_noise_std=1.0,
                                 min_max_ray_net=sampler, refine_net=refine, N_point_ray_enc=P, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn,
                                 randomize=True, embed_rays=helpers.Pluecker(), images=torch.from_numpy(scene['images']), poses=poses, ref_K=K,
                                 num_neighbor=NB, batch_rays_nearest_id=torch.full((N, 1), own, dtype=torch.int64), target_pose=c2w,
                                 train_nerf=True, train_sampler=train_sampler, epi_nerf=False, iter=1000)
    finally:
        pyrandom.sample, pyrandom.random, pyrandom.randint, torch.normal, torch.randn = o_sample, o_rand, o_randint, o_normal, o_randn
    g = lambda t: t.detach().cpu().numpy()
    out = dict(seed=np.int64(seed), H=np.int64(H), W=np.int64(W), nv=np.int64(nv), own=np.int64(own), train_sampler=np.bool_(train_sampler),
               sigma_t=np.float64(sigma_t), rays=g(rays), or_rays=g(or_rays), order_idx=cap['order_idx'],
               rgb_map0=g(ret['rgb_map0']), rgb_map1=g(ret['rgb_map1']), depth_map=g(ret['depth_map']), mm_rgb=g(ret['mm_rgb']),
               depth_map0=g(ret['depth_map0']))
    if train_sampler:
        out.update(sigma1=g(ret['sigma1']))
    else:
        n_mult = cap['n_mult']
        coins = cap['coins']
        dir1 = (1 if coins[0] > 0.5 else -1) if n_mult > 1 else 1
        dir2 = 1 if coins[-1] > 0.5 else -1
        jit = torch.abs(cap['normal'] / 5).clamp(max=0.99)
        out.update(n_mult=np.int64(n_mult), dir1=np.int64(dir1), dir2=np.int64(dir2), jitter=g(jit), raw_noise=g(cap['randn']))
    np.savez_compressed(os.path.join(OUT, f'{name}.npz'), **out)
    extra = '' if train_sampler else f' n_mult={out["n_mult"]} dir1={out["dir1"]} dir2={out["dir2"]}'
    print(f'{name}: N={N} rgb mean={out["rgb_map1"].mean():.4f}{extra} -> {os.path.getsize(os.path.join(OUT, name + ".npz")) // 1024} KiB')


def main():
    helpers, iw, trt = load_reference()
    s2 = load_stage2()
    os.makedirs(OUT, exist_ok=True)
    if '--stage2-only' not in sys.argv and '--stage1-only' not in sys.argv:
        main_infer(helpers, iw, trt)
    if '--stage1-only' not in sys.argv:
        run_stage2_case(helpers, s2, 'stage2_train_16x20', 0, 16, 20, 7, True, False)
        run_stage2_case(helpers, s2, 'stage2_eval_white_12x18', 1, 12, 18, 6, False, True)
    s1 = load_stage1()
    run_stage1_case(helpers, s1, 'stage1_joint_12x16', 0, 12, 16, 6, True, 11)
    run_stage1_case(helpers, s1, 'stage1_explore_a_12x16', 1, 12, 16, 6, False, 5)
    run_stage1_case(helpers, s1, 'stage1_explore_b_10x14', 2, 10, 14, 7, False, 8)
    run_stage1_case(helpers, s1, 'stage1_explore_c_8x12', 3, 8, 12, 6, False, 4)        # n_mult = 8 (64 samples/ray), dir1 = -1


def main_infer(helpers, iw, trt):
    run_operator_cases(helpers, iw, trt)
    # (name, seed, kind, H, W, Hf, Wf, rotate, sigma_t, take)
    run_infer_case(helpers, iw, trt, 'infer_trained_24x32', 0, 'trained', 24, 32)
    run_infer_case(helpers, iw, trt, 'infer_default_24x32', 1, 'default', 24, 32, rotate=True)
    run_infer_case(helpers, iw, trt, 'infer_spread_20x28_img48x64', 2, 'spread', 20, 28, Hf=48, Wf=64, rotate=True)
    run_infer_case(helpers, iw, trt, 'infer_trained_oob_16x24', 3, 'trained', 16, 24, rotate=True, sigma_t=0.6)
    # one full-geometry Fern frame (756x1008), every 1499th ray kept
    run_infer_case(helpers, iw, trt, 'infer_trained_fern_756x1008', 4, 'trained', 756, 1008, rotate=True, take=512)


# Off-Fern shapes (round 6): the reference's own render_rays with other --N_point_ray_enc / --mmnetdepth / --num_neighbor / --netdepth
# (its argparse defaults are not the Fern values: run_S_eS_eN_alter_trt.py:62-82, 110-118).  (name, seed, kind, H, W, shape, pool of views)
SHAPE_CASES = [
    ('infer_shape_p32_d8_nb3_24x32', 5, 'trained', 24, 32, dict(n_pts=32, mmnetdepth=8, num_neighbor=3, netdepth=8), 5),
    ('infer_shape_p64_d5_nb6_nd7_20x28', 6, 'trained', 20, 28, dict(n_pts=64, mmnetdepth=5, num_neighbor=6, netdepth=7), 7),
    ('infer_shape_p8_d2_nb1_nd3_16x20', 7, 'default', 16, 20, dict(n_pts=8, mmnetdepth=2, num_neighbor=1, netdepth=3), 3),
    ('infer_shape_p48_d9_nb8_nd4_16x20', 8, 'trained', 16, 20, dict(n_pts=48, mmnetdepth=9, num_neighbor=8, netdepth=4), 9),
]


def main_shapes():
    helpers, iw, trt = load_reference()
    for name, seed, kind, H, W, shape, nv in SHAPE_CASES:
        run_infer_case(helpers, iw, trt, name, seed, kind, H, W, rotate=True, sigma_t=0.1, shape=shape, n_views=nv)


def fern_8k_selection(H=756, W=1008, seed=12):
    """More than 8192 rays of the 756x1008 frame, stratified: the two outermost rows / columns on every side (whose samples project outside the
    neighbour images) and one seeded random pixel in each cell of an 84 x 84 grid over the frame (8 8xx rays after de-duplication)."""
    rs = np.random.RandomState(seed)
    cols = np.linspace(0, W - 1, 256).astype(np.int64)
    rows = np.linspace(0, H - 1, 192).astype(np.int64)
    border = [r * W + cols for r in (0, 1, H - 2, H - 1)] + [rows * W + c for c in (0, 1, W - 2, W - 1)]
    ys = np.linspace(0, H, 85).astype(np.int64); xs = np.linspace(0, W, 85).astype(np.int64)
    cells = []
    for i in range(84):
        for j in range(84):
            cells.append(rs.randint(ys[i], ys[i + 1]) * W + rs.randint(xs[j], xs[j + 1]))
    sel = np.unique(np.concatenate(border + [np.array(cells, dtype=np.int64)]))
    return sel


def main_fern_8k():
    helpers, iw, trt = load_reference()
    sel = fern_8k_selection()
    run_infer_case(helpers, iw, trt, 'infer_trained_fern_756x1008_8k', 4, 'trained', 756, 1008, rotate=True, sel=sel, slim=True)


if __name__ == '__main__':
    if '--fern-8k' in sys.argv:
        main_fern_8k()
    elif '--shapes' in sys.argv:
        main_shapes()
    else:
        main()
