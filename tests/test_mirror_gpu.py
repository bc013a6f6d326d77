"""GPU: the host-side mirror of the reference's operator interface (pronerf_amd.run_nerf_helpers,
.inverse_warp, .run_S_eS_eN_alter_trt) driven the way the reference's driver drives it
(run_S_eS_eN_alter_trt.py:245-332), against the reference-generated goldens and the oracle."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import pronerf_oracle as orc
from oracle import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    from pronerf_amd import _lib
    _lib.load()
    return torch.device('cuda:0')


def _args():
    return SimpleNamespace(multires=10, multires_views=4, i_embed=0, netdepth=8, netwidth=256, mmnetdepth=6, mmnetwidth=256,
                           mmnetskips=[10000], N_point_ray_enc=48, N_samples=8, num_neighbor=4, ft_path=None)


def _models(dev, seed, kind):
    from pronerf_amd import run_S_eS_eN_alter_trt as trt
    kw, _ = trt.create_nerf(_args(), device=dev)
    sd = synth.state_dicts(synth.make_weights(seed, kind))
    kw['min_max_ray_net'].load_state_dict(sd['sampler']); kw['refine_net'].load_state_dict(sd['refine']); kw['network_fine'].load_state_dict(sd['nerf'])
    return trt, kw


def test_state_dict_keys_match_reference(dev):
    trt, kw = _models(dev, 0, 'trained')
    assert sorted(kw['min_max_ray_net'].state_dict()) == sorted([f'fc_backbone.{i}.{p}' for i in range(6) for p in ('weight', 'bias')] + ['fc_output.weight', 'fc_output.bias'])
    assert sorted(kw['network_fine'].state_dict()) == sorted(f'layers.{i}.{p}' for i in range(8) for p in ('weight', 'bias'))
    assert [tuple(l.weight.shape) for l in kw['network_fine'].layers] == [(256, 63)] + [(256, 256)] * 6 + [(4, 283)]


def test_module_forwards_and_operators(dev, golden_dir):
    from pronerf_amd import inverse_warp as iw
    from pronerf_amd import run_nerf_helpers as h
    trt, kw = _models(dev, 0, 'trained')
    w = synth.make_weights(0, 'trained')
    rs = np.random.RandomState(3)
    x = torch.from_numpy(rs.uniform(-1, 1, (500, 288)).astype(np.float32))
    mm_rgb, add, mul, depth = kw['min_max_ray_net'](x.to(dev))
    r = orc.sampler_forward(w['sampler'], x)
    for got, ref in zip((mm_rgb, add, mul, depth), r):
        np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=2e-5)
    x = torch.from_numpy(rs.uniform(-1, 1, (500, 144)).astype(np.float32))
    rd, rrgb, offs = kw['refine_net'](x.to(dev))
    r = orc.refine_forward(w['refine'], x)
    for got, ref in zip((rd, rrgb, offs), r):
        np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), rtol=0, atol=2e-2)        # bf16 MLP behind sigmoid/tanh
    # network_query_fn = run_network(embed -> DoNeRFTRT): raw rgb-sigma of points
    pts = torch.from_numpy(rs.uniform(-1, 1, (64, 8, 3)).astype(np.float32)); vd = torch.from_numpy(rs.randn(64, 3).astype(np.float32))
    vd = vd / vd.norm(dim=-1, keepdim=True)
    raw = kw['network_query_fn'](pts.to(dev), vd.to(dev), kw['network_fine']).cpu()
    ref = orc.nerf_forward(w['nerf'], orc.posenc(pts.reshape(-1, 3), 10), orc.posenc(vd[:, None].expand(-1, 8, -1).reshape(-1, 3), 4)).reshape(64, 8, 4)
    rel = float(((raw - ref).double() ** 2).mean().sqrt() / (ref.double() ** 2).mean().sqrt())
    assert rel < 2e-2, rel
    # operators from the operators fixture, through the mirror's names
    g = dict(np.load(os.path.join(golden_dir, 'operators.npz')))
    emb, out_dim = h.get_embedder(10, 0)
    assert out_dim == 63 and h.get_embedder(4, 0)[1] == 27 and h.get_embedder(10, -1)[1] == 3
    np.testing.assert_allclose(emb(torch.from_numpy(g['pe_x']).to(dev)).cpu().numpy(), g['pe10'], rtol=0, atol=2e-6)
    np.testing.assert_allclose(h.Pluecker()(torch.from_numpy(g['pl_o']).to(dev), torch.from_numpy(g['pl_d']).to(dev)).cpu().numpy(), g['pl'], rtol=0, atol=1e-6)
    ro, rd_ = h.get_rays(9, 13, torch.from_numpy(g['gr_K']), torch.from_numpy(g['gr_c2w']).to(dev))
    np.testing.assert_array_equal(rd_.cpu().numpy(), g['gr_d'])
    no, nd = h.ndc_rays(9, 13, float(g['gr_K'][0, 0]), 1., ro, rd_)
    np.testing.assert_array_equal(no.cpu().numpy(), g['ndc_o']); np.testing.assert_array_equal(nd.cpu().numpy(), g['ndc_d'])
    B, n = g['wp_depth'].shape[0], g['wp_depth'].shape[2]
    ro1 = torch.from_numpy(g['wp_ro1']).to(dev)[None].expand(B, -1, -1); rd1 = torch.from_numpy(g['wp_rd1']).to(dev)[None].expand(B, -1, -1)
    warped, none = iw.inverse_warp_rod1_rt2_coords_trt(torch.from_numpy(g['wp_img']).to(dev), torch.from_numpy(g['wp_depth']).to(dev),
                                                       ro1, rd1, torch.from_numpy(g['wp_w2c']).to(dev), padding_mode='zeros')
    assert none is None and warped.shape == (B, 3, 1, n)
    np.testing.assert_allclose(warped.cpu().numpy(), g['wp_out'], rtol=0, atol=1e-5)
    from pronerf_amd.ops import PnrfError
    with pytest.raises(PnrfError):
        iw.inverse_warp_rod1_rt2_coords_trt(torch.from_numpy(g['wp_img']).to(dev), torch.from_numpy(g['wp_depth']).to(dev), ro1, rd1,
                                            torch.from_numpy(g['wp_w2c']).to(dev), padding_mode='border')
    out = trt.raw2outputs(torch.from_numpy(g['c_raw']).to(dev), torch.from_numpy(g['c_z']).to(dev), torch.from_numpy(g['c_d']).to(dev), 0., False,
                          mm_density_add=torch.from_numpy(g['c_add']).to(dev), mm_density_mul=torch.from_numpy(g['c_mul']).to(dev))
    for got, key in zip(out, ('c_rgb', 'c_disp', 'c_acc', 'c_w', 'c_depth')):
        np.testing.assert_allclose(got.cpu().numpy(), g[key], rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize('case', ['op', 'cap'])
def test_training_warp_by_its_reference_name(dev, golden_dir, case):
    """inverse_warp.inverse_warp_rod1_rt2_coords (reference signature, (projected, None)) against the reference's own outputs:
    a direct call and the call of the stage-2 driver at refine2.py:617 with its repeat()-ed operands (oracle/gen_golden_warp.py)."""
    from pronerf_amd import inverse_warp as iw
    from pronerf_amd.ops import PnrfError
    g = dict(np.load(os.path.join(golden_dir, 'warp_train.npz')))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    if case == 'op':
        img, depth = t(g['op_img']), t(g['op_depth'])
    else:                                     # ref_rgb = repeat_interleave(images, 8): b = view * S + sample (refine2.py:602-604)
        img, depth = torch.repeat_interleave(t(g['cap_img_views']), int(g['cap_S']), dim=0), t(g['cap_depth'])
    B, n = depth.shape[0], depth.shape[2]
    ro1, rd1 = t(g[case + '_ro1'])[None].repeat(B, 1, 1), t(g[case + '_rd1'])[None].repeat(B, 1, 1)       # :606-607
    K = t(g[case + '_K'])
    warped, none = iw.inverse_warp_rod1_rt2_coords(img, depth, ro1, rd1, t(g[case + '_c2w2']), K, torch.inverse(K), padding_mode='zeros')
    assert none is None and warped.shape == (B, 3, 1, n)
    ref = g[case + '_out'][:, :, 0, :]
    _, margin = orc.warp_train(img.cpu(), depth[:, 0].cpu(), g[case + '_ro1'], g[case + '_rd1'], g[case + '_c2w2'], g[case + '_K'])
    safe = (margin > 1e-5).numpy()
    got = warped[:, :, 0, :].cpu().numpy()
    for b in range(B):
        np.testing.assert_allclose(got[b][:, safe[b]], ref[b][:, safe[b]], rtol=0, atol=2e-5)
    # samples on the in/out boundary itself (the own view's border pixels): either the reference's value or the other branch's zero
    edge = ~safe
    assert edge.mean() < 0.1
    for b in range(B):
        e = edge[b]
        assert np.all((np.abs(got[b][:, e] - ref[b][:, e]).max(0) < 2e-5) | (np.abs(got[b][:, e]).max(0) == 0) | (np.abs(ref[b][:, e]).max(0) == 0))
    # the expand()-ed form (stride-0 batch dimension) gives the same result without materialising the copies
    w2, _ = iw.inverse_warp_rod1_rt2_coords(img, depth, ro1[:1].expand(B, -1, -1), rd1[:1].expand(B, -1, -1), t(g[case + '_c2w2']), K, None)
    assert torch.equal(w2, warped)
    for bad in (dict(padding_mode='border'), dict(scale=0.5)):
        with pytest.raises(PnrfError):
            iw.inverse_warp_rod1_rt2_coords(img, depth, ro1, rd1, t(g[case + '_c2w2']), K, None, **bad)


@pytest.mark.parametrize('name', ['infer_trained_24x32', 'infer_spread_20x28_img48x64'])
def test_render_rays_like_the_reference_driver(dev, golden_dir, name):
    """Build the kwargs exactly as render_path does in the reference (x8 replicated ref_rgb / ref_pose,
    expanded ro1/rd1, mm_input) and call render() — compare with the reference's own outputs."""
    g = dict(np.load(os.path.join(golden_dir, name + '.npz')))
    seed, kind, Hh, Ww = int(g['seed']), str(g['kind']), int(g['H']), int(g['W'])
    trt, kw = _models(dev, seed, kind)
    scene = synth.make_scene(seed, H=Hh, W=Ww, Hf=int(g['Hf']), Wf=int(g['Wf']), rotate=bool(g['rotate']), sigma_t=float(g['sigma_t']))
    fr = orc.frame_setup(scene)          # checker-side frame set-up, pinned to the reference by test_oracle_golden
    S, NB = 8, 4
    rays, or_rays = fr['rays'].to(dev), fr['or_rays'].to(dev)
    ref_rgb = fr['images'].to(dev).unsqueeze(1).expand(-1, S, -1, -1, -1).contiguous().view(NB * S, 3, int(g['Hf']), int(g['Wf']))
    ref_pose = fr['proj'].to(dev).unsqueeze(1).expand(-1, S, -1, -1).contiguous().view(NB * S, 3, 4)
    ro1 = torch.cat([or_rays[:, 0:3].t()[None], torch.ones(1, 1, rays.shape[0], device=dev)], 1).expand(NB * S, -1, -1)
    rd1 = torch.cat([or_rays[:, 3:6].t()[None], torch.zeros(1, 1, rays.shape[0], device=dev)], 1).expand(NB * S, -1, -1)
    fwd = {k: kw[k] for k in ('network_fn', 'network_query_fn', 'N_samples', 'network_fine', 'min_max_ray_net', 'refine_net', 'N_point_ray_enc',
                              'embed_fn', 'embeddirs_fn', 'num_neighbor', 'use_trt', 'embed_rays')}
    rgb0, rgb1, depth_map, extras = trt.render(rays, or_rays, (Hh, Ww, 3), mm_input=fr['mm_input'].to(dev), ref_rgb=ref_rgb, ref_pose=ref_pose,
                                               ro1=ro1, rd1=rd1, **fwd)
    assert rgb1.shape == (Hh, Ww, 3) and depth_map.shape == (Hh, Ww) and extras == {}
    assert orc.psnr(rgb1.reshape(-1, 3).cpu(), torch.from_numpy(g['rgb'])) > 46.4
    np.testing.assert_allclose(depth_map.reshape(-1).cpu().numpy(), g['depth'], rtol=0, atol=2e-2)
    from pronerf_amd.ops import PnrfError
    # mm_input (trt.py:625-628) is verified, not ignored: the reference-built one passes (and is remembered by identity), one that is not the
    # encoding of ray_batch — a changed value, a NaN, another shape — is refused instead of silently rendering something else
    mm = fr['mm_input'].to(dev)
    trt.render(rays, or_rays, (Hh, Ww, 3), mm_input=mm, ref_rgb=ref_rgb, ref_pose=ref_pose, ro1=ro1, rd1=rd1, **fwd)
    assert trt._MM_CHECKED and trt._MM_CHECKED[0][0]() is mm
    # row slices of the verified pair (the reference's per-chunk calling pattern) resolve to their bases: no second verification, no sync
    real_encode = trt.ops.ray_encode
    trt.ops.ray_encode = lambda *a, **k: (_ for _ in ()).throw(AssertionError('a slice of a verified mm_input was verified again'))
    try:
        half = rays.shape[0] // 2
        out = trt.render_rays(rays[5:half], or_rays[5:half], mm_input=mm[5:half], ref_rgb=ref_rgb, ref_pose=ref_pose, ro1=ro1, rd1=rd1, **fwd)
        assert torch.equal(out['rgb_map1'], rgb1.reshape(-1, 3)[5:half]) and trt._MM_CHECKED[0][0]() is mm
    finally:
        trt.ops.ray_encode = real_encode
    with pytest.raises(PnrfError):           # a slice of mm_input that belongs to OTHER rows of the frame is not a hit, and is refused
        trt.render_rays(rays[5:half], or_rays[5:half], mm_input=mm[6:half + 1], ref_rgb=ref_rgb, ref_pose=ref_pose, ro1=ro1, rd1=rd1, **fwd)
    trt.render(rays, or_rays, (Hh, Ww, 3), mm_input=mm, ref_rgb=ref_rgb, ref_pose=ref_pose, ro1=ro1, rd1=rd1, **fwd)      # remembered again
    for tamper in ('value', 'nan', 'shape', 'inplace'):
        bad = mm.clone() if tamper != 'inplace' else mm
        if tamper == 'value':
            bad[rays.shape[0] // 2, 7] += 1e-3
        elif tamper == 'nan':
            bad[0, 0] = float('nan')
        elif tamper == 'shape':
            bad = bad[:, :282].contiguous()
        else:
            bad[3, 5] += 1e-3                   # the remembered tensor changed in place: its version moved, so it is checked again
        with pytest.raises(PnrfError):
            trt.render(rays, or_rays, (Hh, Ww, 3), mm_input=bad, ref_rgb=ref_rgb, ref_pose=ref_pose, ro1=ro1, rd1=rd1, **fwd)
    with pytest.raises(PnrfError):
        trt.render(rays, or_rays, (Hh, Ww, 3), ref_rgb=ref_rgb, ref_pose=ref_pose, **{**fwd, 'use_trt': True})
    with pytest.raises(PnrfError):
        trt.render(rays, or_rays, (Hh, Ww, 3), ref_rgb=ref_rgb, ref_pose=ref_pose, **{**fwd, 'N_samples': 16})


def test_render_path_frame_loop(dev, tmp_path):
    """render_path: per-frame set-up + timed loop + PSNR + PNG, on a tiny synthetic 'dataset'."""
    trt, kw = _models(dev, 0, 'trained')
    scene = synth.make_scene(0, H=24, W=32, n_views=6)
    kw.update(poses=scene['poses'], images=scene['images'], ref_K=scene['K'])
    targets = [scene['c2w'], scene['poses'][0]]
    fr = orc.frame_setup({**scene, 'c2w': scene['c2w']}, num_neighbor=4)
    ref = orc.render_rays_infer(synth.make_weights(0, 'trained'), fr['rays'], fr['or_rays'], fr['images'], fr['proj'])
    gt = [ref['rgb'].reshape(24, 32, 3).numpy(), np.zeros((24, 32, 3), np.float32)]
    rgbs0, rgbs1, depths, _ = trt.render_path(targets, (24, 32, scene['focal']), scene['K'], None, kw, gt_imgs=gt, savedir=str(tmp_path),
                                              n_timing_reps=2, verbose=False)
    assert rgbs1.shape == (2, 24, 32, 3) and depths.shape == (2, 24, 32)
    assert kw['psnrs'][0] > 46.4                      # frame 0 vs the oracle's image of the same pose
    assert len(kw['render_ms']) == 2 and all(len(t) == 2 and min(t) > 0 for t in kw['render_ms'])
    png = open(os.path.join(str(tmp_path), '000.png'), 'rb').read()
    assert png[:8] == b'\x89PNG\r\n\x1a\n' and os.path.exists(os.path.join(str(tmp_path), 'depth_001.png'))


def test_nerf_class_module_and_checkpoint_dispatch(dev, golden_dir, tmp_path):
    """NeRF module: reference state_dict keys, forward(x[M,90]); create_nerf picks the class from the checkpoint's keys
    (the released stage-2 trainer saves NeRF-class weights under 'network_fine_state_dict')."""
    from pronerf_amd import run_nerf_helpers as h
    from pronerf_amd import run_S_eS_eN_alter_trt as trt
    wc = synth.make_nerfcls_weights(0)
    sd = synth.nerfcls_state_dict(wc)
    m = h.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4, skips=[4], use_viewdirs=True).to(dev)
    assert sorted(m.state_dict()) == sorted(sd)
    m.load_state_dict(sd)
    g = dict(np.load(os.path.join(golden_dir, 'operators.npz')))
    y = m(torch.from_numpy(g['nc_x']).to(dev)).cpu()
    rel = float(((y - torch.from_numpy(g['nc_y'])).double() ** 2).mean().sqrt() / (torch.from_numpy(g['nc_y']).double() ** 2).mean().sqrt())
    assert rel < 2e-2, rel
    sds = synth.state_dicts(synth.make_weights(0, 'trained'))
    ck = os.path.join(str(tmp_path), '000001.tar')
    torch.save({'global_step': 1, 'mmr_network_fn_state_dict': sds['sampler'], 'refine_net_state_dict': sds['refine'], 'network_fine_state_dict': sd}, ck)
    a = _args(); a.ft_path = ck
    kw, start = trt.create_nerf(a, device=dev)
    assert start == 1 and isinstance(kw['network_fine'], h.NeRF)
    torch.save({'global_step': 2, 'mmr_network_fn_state_dict': sds['sampler'], 'refine_net_state_dict': sds['refine'], 'network_fine_state_dict': sds['nerf']}, ck)
    kw, start = trt.create_nerf(a, device=dev)
    assert start == 2 and isinstance(kw['network_fine'], h.DoNeRFTRT)


def test_frame_driver_on_an_llff_directory(dev, tmp_path):
    """train() of the inference mirror end to end: LLFF directory + COLMAP model + .tar checkpoint with the reference's keys
    -> hold-out renders, PNGs, PSNR; frame 0 is compared with the oracle's rendering of the same pose / reference views."""
    import llff_synth
    from pronerf_amd import load_llff as L
    from pronerf_amd import run_S_eS_eN_alter_trt as trt
    root = llff_synth.make_dataset(str(tmp_path / 'scene'), seed=1, n=10, H=24, W=32, factor=4)
    w = synth.make_weights(0, 'trained')
    sds = synth.state_dicts(w)
    ck = str(tmp_path / '000123.tar')
    torch.save({'global_step': 123, 'mmr_network_fn_state_dict': sds['sampler'], 'refine_net_state_dict': sds['refine'],
                'network_fine_state_dict': sds['nerf']}, ck)
    cfg = tmp_path / 'cfg.txt'
    cfg.write_text(f'expname = drv\nbasedir = {tmp_path}/logs\ndatadir = {root}\nft_path = {ck}\nfactor = 4\nllffhold = 8\nN_samples = 8\n'
                   'N_point_ray_enc = 48\nmmnetdepth = 6\nmmnetskips = [10000]\nnum_neighbor = 4\nuse_viewdirs = True\n')
    kw = trt.train(['--config', str(cfg), '--render_test'], device=dev)
    out = tmp_path / 'logs' / 'drv' / 'renderonly_test_000123'
    assert sorted(os.listdir(out)) == ['000.png', '001.png', 'depth_000.png', 'depth_001.png']
    assert (tmp_path / 'logs' / 'drv' / 'args.txt').read_text().count('\n') > 50
    assert len(kw['psnrs']) == 2 and len(kw['render_ms']) == 2
    # checker: same scene through the loader + oracle
    images, poses, bds, _, i_test, i_ref = L.load_llff_data_infer(root, factor=4)
    H, W, focal = int(poses[0, 0, 4]), int(poses[0, 1, 4]), float(poses[0, 2, 4])
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]], dtype=np.float32)
    scene = {'H': H, 'W': W, 'K': K, 'c2w': poses[i_test[0], :3, :4], 'poses': poses[i_ref][:, :3, :4], 'images': images[i_ref]}
    fr = orc.frame_setup(scene)
    ref = orc.render_rays_infer(w, fr['rays'], fr['or_rays'], fr['images'], fr['proj'])
    got = _read_png(str(out / '000.png')).astype(np.float32) / 255.
    want = np.clip(ref['rgb'].reshape(H, W, 3).numpy(), 0, 1)
    assert orc.psnr(torch.from_numpy(got), torch.from_numpy(want)) > 40.0        # 8-bit PNG (quantisation floor ~59 dB) vs the oracle's frame


def _read_png(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert('RGB'))


def test_frame_driver_on_the_scene_with_its_trained_checkpoint(dev, tmp_path):
    """The reference's own acceptance, end to end through the drop-in driver (run_S_eS_eN_alter_trt.py:699-800, 351-353): the LLFF directory of the
    geometrically consistent scene (tests/llff_synth.py Scene3D), a .tar checkpoint with the reference's keys holding the nets trained on it (the NeRF-class
    fine net, as the released trainers save it), ``--render_test`` -> PSNR of the hold-out views against their ground-truth pictures, PNGs.
    ``--pnrf_preset auto`` decides the sampler form from the first frame (a sampler that has learned surfaces: second pass 69 % -> the exact single pass);
    'default' and 'quality' render the same views within 0.05 dB of each other."""
    import llff_synth
    from pronerf_amd import run_S_eS_eN_alter_trt as trt
    root = llff_synth.make_dataset(str(tmp_path / 'scene'), seed=2, n=20, H=189, W=252, factor=4, consistent=True, n_points=3000)
    w = synth.load_trained_fixture('scene3d')
    stack = lambda d: {**{f'fc_backbone.{i}.{k}': torch.from_numpy(np.asarray(v[i])) for i in range(len(d['W']) - 1) for k, v in (('weight', d['W']), ('bias', d['b']))},
                       'fc_output.weight': torch.from_numpy(np.asarray(d['W'][-1])), 'fc_output.bias': torch.from_numpy(np.asarray(d['b'][-1]))}
    ck = str(tmp_path / '040000.tar')
    torch.save({'global_step': 40000, 'mmr_network_fn_state_dict': stack(w['sampler']), 'refine_net_state_dict': stack(w['refine']),
                'network_fine_state_dict': synth.nerfcls_state_dict(w['nerfcls'])}, ck)
    body = (f'basedir = {tmp_path}/logs\ndatadir = {root}\nft_path = {ck}\nfactor = 4\nllffhold = 8\nN_samples = 8\nN_point_ray_enc = 48\n'
            'mmnetdepth = 6\nmmnetskips = [10000]\nnum_neighbor = 4\nuse_viewdirs = True\n')
    psnr = {}
    for preset in ('auto', 'default', 'quality'):
        cfg = tmp_path / f'{preset}.txt'
        cfg.write_text(f'expname = {preset}\n' + body)
        kw = trt.train(['--config', str(cfg), '--render_test', '--pnrf_preset', preset], device=dev)
        psnr[preset] = kw['psnrs']
        assert len(kw['psnrs']) == 3 and min(kw['psnrs']) > 33.0, (preset, kw['psnrs'])          # hold-out views 0, 8, 16: 37.3 / 36.4 / 33.8 dB
        if preset == 'auto':
            assert 'sampler_split' in kw['pnrf_preset_in_force'], kw['pnrf_preset_in_force']
            assert sorted(os.listdir(tmp_path / 'logs' / 'auto' / 'renderonly_test_040000'))[:2] == ['000.png', '001.png']
        elif preset == 'quality':
            assert kw['pnrf_preset_in_force'] == 'quality'
    print('\n[frame driver on the scene] hold-out PSNR vs ground truth:', {k: [round(x, 3) for x in v] for k, v in psnr.items()})
    for a in ('auto', 'quality'):
        assert max(abs(x - y) for x, y in zip(psnr[a], psnr['default'])) <= 0.05
    from pronerf_amd.ops import PnrfError
    with pytest.raises(PnrfError):
        trt.apply_preset('fast', kw['min_max_ray_net'], kw['refine_net'], kw['network_fine'])


def test_frame_driver_under_torchrun_shards_every_frame(dev, tmp_path):
    """Two processes (one GPU here, gloo; one per GPU over RCCL in production) run the inference script: every frame's rays are split
    over the ranks and gathered; rank 0 writes the same PNG bytes as the single-process run."""
    import subprocess
    import sys
    import llff_synth
    from pronerf_amd import run_S_eS_eN_alter_trt as trt
    root = llff_synth.make_dataset(str(tmp_path / 'scene'), seed=1, n=10, H=24, W=32, factor=4)
    sds = synth.state_dicts(synth.make_weights(0, 'trained'))
    ck = str(tmp_path / '000123.tar')
    torch.save({'global_step': 123, 'mmr_network_fn_state_dict': sds['sampler'], 'refine_net_state_dict': sds['refine'],
                'network_fine_state_dict': sds['nerf']}, ck)
    body = (f'basedir = {tmp_path}/logs\ndatadir = {root}\nft_path = {ck}\nfactor = 4\nllffhold = 8\nN_samples = 8\nN_point_ray_enc = 48\n'
            'mmnetdepth = 6\nmmnetskips = [10000]\nnum_neighbor = 4\nuse_viewdirs = True\n')
    (tmp_path / 'one.txt').write_text('expname = one\n' + body)
    (tmp_path / 'two.txt').write_text('expname = two\n' + body)
    kw = trt.train(['--config', str(tmp_path / 'one.txt'), '--render_test'], device=dev)
    env = {**os.environ, 'PNRF_DIST_BACKEND': 'gloo', 'PYTHONPATH': os.path.dirname(os.path.dirname(os.path.abspath(__file__)))}
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', '29549', '-m', 'pronerf_amd.run_S_eS_eN_alter_trt', '--config', str(tmp_path / 'two.txt'), '--render_test'],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    d1, d2 = tmp_path / 'logs' / 'one' / 'renderonly_test_000123', tmp_path / 'logs' / 'two' / 'renderonly_test_000123'
    assert sorted(os.listdir(d1)) == sorted(os.listdir(d2)) == ['000.png', '001.png', 'depth_000.png', 'depth_001.png']
    for f in os.listdir(d1):
        assert (d1 / f).read_bytes() == (d2 / f).read_bytes(), f
    assert r.stdout.count('Mean Test PSNR') == 1 and f"{kw['psnrs'][0]:.4f}"[:6] in r.stdout        # only rank 0 reports
