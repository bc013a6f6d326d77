#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY — golden of the reference's TRAINING warp ``inverse_warp.inverse_warp_rod1_rt2_coords``
(inverse_warp.py:515-581), produced by calling the reference itself on CPU in the build container:

  * ``op_*``  : a direct call on seeded inputs whose projections deliberately leave the source images
                (the X_norm / Y_norm -> 2 branch of :557-561);
  * ``cap_*`` : the arguments and the result of the call the stage-2 driver makes at
                run_S_eS_eN_alter_base_refine2.py:617 (all k_ref training views x 8 samples, 1 x N_rays "image"),
                captured from the reference's own ``render_rays`` on a small seeded scene.

    python oracle/gen_golden_warp.py          -> tests/golden/warp_train.npz
The reference never leaves this container; the fixture holds arrays only."""
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


def operator_case(iw, out):
    rs = np.random.RandomState(11)
    B, Hf, Wf, n = 6, 21, 29, 517
    scene = synth.make_scene(6, H=Hf, W=Wf, n_views=B, sigma_t=0.4, rotate=True)
    img = torch.from_numpy(scene['images']).permute(0, 3, 1, 2).contiguous()
    c2w2 = torch.from_numpy(scene['poses'])
    K = torch.from_numpy(scene['K'])[None].repeat(B, 1, 1)
    ro1 = torch.from_numpy((rs.randn(1, 3, n) * 0.1).astype(np.float32)).repeat(B, 1, 1)
    rd1 = torch.from_numpy(np.concatenate([rs.randn(1, 2, n) * 0.45, -np.ones((1, 1, n))], 1).astype(np.float32)).repeat(B, 1, 1)
    depth = torch.from_numpy(rs.uniform(1.0, 25.0, (B, 1, n)).astype(np.float32))
    warped, none = iw.inverse_warp_rod1_rt2_coords(img, depth.clone(), ro1.clone(), rd1.clone(), c2w2, K, torch.inverse(K), padding_mode='zeros')
    assert none is None
    out.update(op_img=img.numpy(), op_depth=depth.numpy(), op_ro1=ro1[0].numpy(), op_rd1=rd1[0].numpy(), op_c2w2=c2w2.numpy(), op_K=K.numpy(),
               op_out=warped.numpy())
    nz = (warped.abs().sum(1) > 0).float().mean().item()
    print(f'operator case: B={B} n={n}, fraction of non-zero samples {nz:.2f}')


def captured_case(helpers, iw, s2, out):
    """The call at refine2.py:617 inside the reference's stage-2 render_rays (eval branch: ranks 0..3, no random draws needed)."""
    seed, H, W, nv = 2, 10, 14, 6
    torch.manual_seed(3407); pyrandom.seed(3407)
    S, NB, P = synth.N_SAMPLES, synth.NUM_NEIGHBOR, synth.N_POINT_RAY_ENC
    w = synth.make_weights(seed, 'trained'); wc = synth.make_nerfcls_weights(seed, head_scale=0.3); sd = synth.state_dicts(w)
    sampler = helpers.MinMaxRay_Net(D=synth.MMNETDEPTH, W=synth.MMNETWIDTH, input_ch=6 * P, output_ch=3 * S + 3, skips=[10000])
    refine = helpers.MinMaxRay_Net(D=synth.MMNETDEPTH, W=synth.MMNETWIDTH, input_ch=6 * S + 3 * NB * S, output_ch=4 * S + 3, skips=[10000])
    fine = helpers.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4, skips=[4], use_viewdirs=True)
    sampler.load_state_dict(sd['sampler']); refine.load_state_dict(sd['refine']); fine.load_state_dict(synth.nerfcls_state_dict(wc))
    scene = synth.make_scene(seed, H=H, W=W, n_views=nv, sigma_t=0.3, rotate=True)
    own = 1
    K = torch.from_numpy(scene['K']); poses = torch.from_numpy(scene['poses']); c2w = poses[own]
    rays_o, rays_d = helpers.get_rays(H, W, K, c2w)
    viewdirs = (rays_d / torch.norm(rays_d, dim=-1, keepdim=True)).reshape(-1, 3).float()
    or_o, or_d = rays_o.reshape(-1, 3).float(), rays_d.reshape(-1, 3).float()
    N = or_o.shape[0]
    or_rays = torch.cat([or_o, or_d, torch.ones(N, 1), 10 * torch.ones(N, 1), viewdirs], -1)
    o, d = helpers.ndc_rays(H, W, K[0][0], 1., rays_o, rays_d)
    o, d = o.reshape(-1, 3).float(), d.reshape(-1, 3).float()
    rays = torch.cat([o, d, torch.zeros(N, 1), torch.ones(N, 1), viewdirs], -1)
    embed_fn, _ = helpers.get_embedder(synth.MULTIRES, 0)
    embeddirs_fn, _ = helpers.get_embedder(synth.MULTIRES_VIEWS, 0)
    query = lambda inputs, vd, fn: s2.run_network(inputs, vd, fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn)
    cap = {}
    orig = iw.inverse_warp_rod1_rt2_coords

    def spy(img, depth, ro1, rd1, c2w2, Kb, Kinv, *a, **k):
        cap.update(img=img.clone(), depth=depth.clone(), ro1=ro1.clone(), rd1=rd1.clone(), c2w2=c2w2.clone(), K=Kb.clone(), kw=dict(k))
        r = orig(img, depth, ro1, rd1, c2w2, Kb, Kinv, *a, **k)
        cap['out'] = r[0].clone()
        return r

    iw.inverse_warp_rod1_rt2_coords = spy
    try:
        with torch.no_grad():
            s2.render_rays(rays, or_rays, network_fn=None, network_query_fn=query, N_samples=S, network_fine=fine, white_bkgd=False, raw_noise_std=0.0,
                           min_max_ray_net=sampler, refine_net=refine, N_point_ray_enc=P, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn,
                           randomize=False, embed_rays=helpers.Pluecker(), images=torch.from_numpy(scene['images']), poses=poses, ref_K=K,
                           num_neighbor=NB, batch_rays_nearest_id=torch.full((N, 1), own, dtype=torch.int64), target_pose=c2w, train_nerf=False, iter=1000)
    finally:
        iw.inverse_warp_rod1_rt2_coords = orig
    assert cap['kw'].get('padding_mode', 'zeros') == 'zeros'
    B = cap['img'].shape[0]
    assert B == nv * S and tuple(cap['depth'].shape) == (B, 1, N)
    # the replicated operands are stored once: images per view (b = view * S + sample), rays once (ro1/rd1 are repeats of one [3,N])
    assert torch.equal(cap['img'].view(nv, S, 3, H, W)[:, 0:1].expand(-1, S, -1, -1, -1).reshape(B, 3, H, W), cap['img'])
    assert torch.equal(cap['ro1'][0:1].expand(B, -1, -1), cap['ro1'])
    out.update(cap_seed=np.int64(seed), cap_H=np.int64(H), cap_W=np.int64(W), cap_nv=np.int64(nv), cap_S=np.int64(S),
               cap_img_views=cap['img'].view(nv, S, 3, H, W)[:, 0].numpy(), cap_depth=cap['depth'].numpy(), cap_ro1=cap['ro1'][0].numpy(),
               cap_rd1=cap['rd1'][0].numpy(), cap_c2w2=cap['c2w2'].numpy(), cap_K=cap['K'].numpy(), cap_out=cap['out'].numpy())
    print(f'captured case: B={B} (views {nv} x samples {S}), N={N}, non-zero fraction {(cap["out"].abs().sum(1) > 0).float().mean().item():.2f}')


def main():
    helpers, iw, _ = G.load_reference()
    s2 = G.load_stage2()
    out = {}
    operator_case(iw, out)
    captured_case(helpers, iw, s2, out)
    path = os.path.join(G.OUT, 'warp_train.npz')
    np.savez_compressed(path, **out)
    print('warp_train ->', os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    main()
