"""GPU: the reference's free shape arguments (round 6, VERDICT r5 "missing 1").

``--N_point_ray_enc``, ``--mmnetdepth``, ``--num_neighbor`` and ``--netdepth`` are free in the reference (run_S_eS_eN_alter_trt.py:62-82, 110-118,
427-457: ``create_nerf`` sizes the three modules from them; its argparse DEFAULTS are not the Fern values).  The kernels keep width 256 and 8 samples per
ray and take: any number of ray points (the fused sampler runs the folded first layer), any depth of the sampler / refine stacks, 1 .. 8 neighbour views,
DoNeRFTRT depth 3 .. 8.  Parity on four off-Fern shapes — one of them ``N_point_ray_enc 32, mmnetdepth 8, num_neighbor 3`` —

  * against outputs of the REFERENCE itself on those shapes (tests/golden/infer_shape_*.npz, oracle/gen_golden.py --shapes; the CPU suite pins the
    oracle on the same files): sampler indices identical outside fp32 ties, rgb PSNR >= 46.4 dB, depth <= 2e-2, through ``Renderer`` and through the
    mirror of the reference's driver API (``create_nerf`` from args + ``render``);
  * at the BASELINE frame size (762 048 rays) against the oracle's eager fp32 graph on the device: indices + 46.4 dB;
  * operator by operator against the oracle: the three sampler kernels, the exact projection operator for every num_neighbor 1 .. 8, the
    projecting refine stage against projection + refine, engine-file round trips;
  * everything outside the supported set still raises PNRF_E_SHAPE (PnrfError) and the message names the supported set.
"""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import pronerf_oracle as orc
from oracle import synth

pytestmark = pytest.mark.gpu

SHAPE_CASES = ['infer_shape_p32_d8_nb3_24x32', 'infer_shape_p64_d5_nb6_nd7_20x28', 'infer_shape_p8_d2_nb1_nd3_16x20', 'infer_shape_p48_d9_nb8_nd4_16x20']
TIE = 1e-6


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    from pronerf_amd import _lib
    _lib.load()
    return torch.device('cuda:0')


def _case(golden_dir, name):
    g = dict(np.load(os.path.join(golden_dir, name + '.npz')))
    shape = dict(n_pts=int(g['n_pts']), mmnetdepth=int(g['mmnetdepth']), num_neighbor=int(g['num_neighbor']), netdepth=int(g['netdepth']))
    scene = synth.make_scene(int(g['seed']), H=int(g['H']), W=int(g['W']), Hf=int(g['Hf']), Wf=int(g['Wf']), rotate=bool(g['rotate']),
                             sigma_t=float(g['sigma_t']), n_views=int(g['n_views']))
    return g, shape, scene, synth.make_weights(int(g['seed']), str(g['kind']), **shape)


def _check_against_reference(g, rgbd, idx, tag):
    tie_free = np.diff(g['depth_sorted'], axis=1).min(axis=1) > TIE
    m = torch.from_numpy(tie_free)
    np.testing.assert_array_equal(idx.cpu().numpy()[tie_free], g['sort_idx'][tie_free])
    ps = orc.psnr(rgbd[:, :3].cpu()[m], torch.from_numpy(g['rgb'])[m])
    derr = float((rgbd[:, 3].cpu()[m] - torch.from_numpy(g['depth'])[m]).abs().max())
    print(f'\n[shapes] {tag}: {int(tie_free.sum())} of {len(tie_free)} rays outside the tie set, indices identical, rgb PSNR vs the reference {ps:.1f} dB, '
          f'max depth error {derr:.2e}')
    assert ps > 46.4 and derr < 2e-2 and int((~tie_free).sum()) <= 0.05 * len(tie_free)


@pytest.mark.parametrize('name', SHAPE_CASES)
def test_render_rays_on_off_fern_shapes_vs_the_reference(dev, golden_dir, name):
    from pronerf_amd.render import Renderer
    g, shape, scene, w = _case(golden_dir, name)
    H, W = int(g['H']), int(g['W'])
    for preset in ('default', 'quality'):
        rend = Renderer(w, max_rays=H * W, device=dev, preset=preset)
        assert rend.num_neighbor == shape['num_neighbor']
        ref_nos = rend.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
        np.testing.assert_array_equal(ref_nos, g['ref_nos'])
        np.testing.assert_allclose(rend.proj.cpu().numpy(), g['proj'], rtol=1e-6, atol=1e-5)
        rays, or_rays = rend.frame_rays(scene['K'], scene['c2w'], H, W)
        np.testing.assert_array_equal(rays.cpu().numpy(), g['rays'])                    # the device's rays are the reference's, bit for bit
        rgbd, idx = rend.render_rays(rays, or_rays, want_idx=True)
        _check_against_reference(g, rgbd, idx, f'{name} [{preset}] {shape}')


@pytest.mark.parametrize('name', SHAPE_CASES[:2])
def test_driver_api_on_off_fern_shapes(dev, golden_dir, name):
    """create_nerf(args) with the shape's arguments, the reference's kwargs (x8 replicated ref_rgb / ref_pose, mm_input [N, 6 P]) -> render()."""
    from pronerf_amd import run_S_eS_eN_alter_trt as trt
    g, shape, scene, w = _case(golden_dir, name)
    args = SimpleNamespace(multires=10, multires_views=4, i_embed=0, netdepth=shape['netdepth'], netwidth=256, mmnetdepth=shape['mmnetdepth'], mmnetwidth=256,
                           mmnetskips=[10000], N_point_ray_enc=shape['n_pts'], N_samples=8, num_neighbor=shape['num_neighbor'], ft_path=None)
    kw, _ = trt.create_nerf(args, device=dev)
    sd = synth.state_dicts(w)
    kw['min_max_ray_net'].load_state_dict(sd['sampler']); kw['refine_net'].load_state_dict(sd['refine']); kw['network_fine'].load_state_dict(sd['nerf'])
    assert len(kw['min_max_ray_net'].fc_backbone) == shape['mmnetdepth'] and kw['min_max_ray_net'].fc_backbone[0].in_features == 6 * shape['n_pts']
    assert kw['refine_net'].fc_backbone[0].in_features == 48 + 24 * shape['num_neighbor'] and len(kw['network_fine'].layers) == shape['netdepth']
    fr = orc.frame_setup(scene, num_neighbor=shape['num_neighbor'], n_pts=shape['n_pts'])
    S, NB, Hh, Ww = 8, shape['num_neighbor'], int(g['H']), int(g['W'])
    rays, or_rays = fr['rays'].to(dev), fr['or_rays'].to(dev)
    ref_rgb = fr['images'].to(dev).unsqueeze(1).expand(-1, S, -1, -1, -1).contiguous().view(NB * S, 3, int(g['Hf']), int(g['Wf']))
    ref_pose = fr['proj'].to(dev).unsqueeze(1).expand(-1, S, -1, -1).contiguous().view(NB * S, 3, 4)
    fwd = {k: kw[k] for k in ('network_fn', 'network_query_fn', 'N_samples', 'network_fine', 'min_max_ray_net', 'refine_net', 'N_point_ray_enc',
                              'embed_fn', 'embeddirs_fn', 'num_neighbor', 'use_trt', 'embed_rays')}
    rgb0, rgb1, depth_map, _ = trt.render(rays, or_rays, (Hh, Ww, 3), mm_input=fr['mm_input'].to(dev), ref_rgb=ref_rgb, ref_pose=ref_pose, **fwd)
    tie_free = torch.from_numpy(np.diff(g['depth_sorted'], axis=1).min(axis=1) > TIE)
    assert orc.psnr(rgb1.reshape(-1, 3).cpu()[tie_free], torch.from_numpy(g['rgb'])[tie_free]) > 46.4
    np.testing.assert_allclose(depth_map.reshape(-1).cpu().numpy()[tie_free.numpy()], g['depth'][tie_free.numpy()], rtol=0, atol=2e-2)
    from pronerf_amd.ops import PnrfError
    with pytest.raises(PnrfError):             # the arguments must be the ones the modules were built with
        trt.render(rays, or_rays, (Hh, Ww, 3), ref_rgb=ref_rgb, ref_pose=ref_pose, **{**fwd, 'num_neighbor': 4 if NB != 4 else 3})


@pytest.mark.parametrize('shape', [dict(n_pts=32, mmnetdepth=8, num_neighbor=3, netdepth=8), dict(n_pts=64, mmnetdepth=5, num_neighbor=6, netdepth=7)])
def test_full_frame_on_off_fern_shapes_vs_eager_oracle(dev, shape):
    """All 762 048 rays of the BASELINE frame with an off-Fern shape: the fused path against the oracle's eager fp32 graph on the device."""
    from pronerf_amd.render import Renderer
    torch.backends.cuda.matmul.allow_tf32 = False
    H, W, FOCAL = 756, 1008, 815.13
    N = H * W
    scene = synth.make_scene(11, H=H, W=W, focal=FOCAL, rotate=True, n_views=shape['num_neighbor'] + 2)
    w = synth.make_weights(11, 'trained', **shape)
    fr = orc.frame_setup(scene, num_neighbor=shape['num_neighbor'], n_pts=shape['n_pts'])
    wd = {k: {'W': [torch.as_tensor(x).to(dev) for x in v['W']], 'b': [torch.as_tensor(x).to(dev) for x in v['b']]} for k, v in w.items()}
    ref = {}
    rays, or_rays = fr['rays'].to(dev), fr['or_rays'].to(dev)
    with torch.no_grad():                       # in chunks: the eager graph holds [n, 6 P] and [n * 8, 256] intermediates
        outs = [orc.render_rays_infer(wd, rays[a:a + 131072], or_rays[a:a + 131072], fr['images'].to(dev), fr['proj'].to(dev), n_pts=shape['n_pts'])
                for a in range(0, N, 131072)]
        for k in ('depth_sorted', 'sort_idx', 'rgb', 'depth'):
            ref[k] = torch.cat([o[k] for o in outs])
        del outs
    free = (ref['depth_sorted'][:, 1:] - ref['depth_sorted'][:, :-1]).min(dim=1)[0] > 4e-6
    assert int((~free).sum()) <= 2e-3 * N
    # Tolerances.  rgb: error PSNR >= 46.4 dB.  Depth map, 'quality' preset (fp32-grade sampler depths): the derived bars of
    # tests/test_fullframe_gpu.py — error PSNR clears the rgb gate, 99 % of the pixels within half a grey level of the reference's 8-bit depth image,
    # 99.9 % within one, none beyond 2.5.  'default' preset: the PSNR-equivalent gate, 99 % within one level, 99.9 % within four — the fp16-grade
    # depths of its undecided-free rays go through deeper 'trained' stacks here than Fern's (mmnetdepth 8: 99.9 % <= 2.6 levels; Fern: <= 0.74).
    bars = {'default': (1.0, 4.0, 6e-2 * 255), 'quality': (0.5, 1.0, 2.5)}
    for preset in ('default', 'quality'):
        rend = Renderer(w, max_rays=N, device=dev, preset=preset)
        rend.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
        np.testing.assert_array_equal(rend.ref_nos, fr['ref_nos'].numpy())
        r2, o2 = rend.frame_rays(scene['K'], scene['c2w'], H, W)
        assert torch.equal(r2, rays) and torch.equal(o2, or_rays)
        rgbd, idx = rend.render_rays(r2, o2, want_idx=True)
        mism = int((idx[free] != ref['sort_idx'][free]).any(1).sum())
        ps = orc.psnr(rgbd[free, :3], ref['rgb'][free])
        e = (rgbd[free, 3].double() - ref['depth'][free].double()).abs()
        dpsnr = -10.0 * float(torch.log10((e ** 2).mean()))
        q99, q999 = (float(torch.quantile(e[::max(1, e.numel() // 1000000)], q)) * 255 for q in (0.99, 0.999))
        mx = float(e.max()) * 255
        print(f'\n[shapes, full frame] {shape} [{preset}]: {int(free.sum())} of {N} rays compared, index mismatches {mism}, rgb PSNR {ps:.1f} dB; depth error PSNR '
              f'{dpsnr:.1f} dB, 99 % <= {q99:.3f} grey levels, 99.9 % <= {q999:.3f}, max {mx:.2f}; second pass {rend.ctx.sampler_stats() / N:.1%}')
        assert mism == 0 and ps > 46.4 and bool(torch.isfinite(rgbd).all())
        assert dpsnr >= 46.4 and q99 <= bars[preset][0] and q999 <= bars[preset][1] and mx <= bars[preset][2]
        del rend


@pytest.mark.parametrize('n_pts,mmnetdepth', [(32, 8), (64, 5), (8, 2), (1, 3), (48, 7)])
def test_sampler_kernels_on_off_fern_shapes_vs_oracle(dev, n_pts, mmnetdepth):
    """The three fused sampler forms (split fp16, exact fp32 with the folded first layer, the two-pass default) on other N_point_ray_enc / mmnetdepth
    (odd and even numbers of hidden layers) against the oracle's sampler on the full 6 P-wide encoding."""
    from pronerf_amd import ops
    w = synth.make_weights(3, 'trained', n_pts=n_pts, mmnetdepth=mmnetdepth)['sampler']
    scene = synth.make_scene(3, H=60, W=80, rotate=True)
    rays, _ = ops.frame_rays(scene['K'], scene['c2w'], 60, 80, device=dev)
    o, d = rays[:, 0:3].cpu(), rays[:, 3:6].cpu()
    with torch.no_grad():
        _, add, mul, depth = orc.sampler_forward(w, orc.mm_input_from_rays(o, d, n_pts))
        ds, idx, add_s, mul_s = orc.sort_gather(depth, add, mul, rays[:, 6:7].cpu(), rays[:, 7:8].cpu())
    tie = (ds[:, 1:] - ds[:, :-1]).min(1)[0] <= TIE
    np.testing.assert_array_equal(ops.ray_encode(rays, n_pts).cpu().numpy(), orc.mm_input_from_rays(o, d, n_pts).numpy())     # the encoding operator takes any P
    for variant, two_pass in (('default', False), ('sampler_f32', False), ('default', True)):
        mlp = ops.PackedMLP(ops.NET_SAMPLER, w['W'], w['b'], variant=variant)
        out = ops.sampler_fwd(mlp, rays, want_idx=True, want_rgb=True, want_raw=True, two_pass=two_pass)
        assert int((out[1].cpu()[~tie] != idx[~tie]).any(1).sum()) == 0, (variant, two_pass)
        np.testing.assert_allclose(out[5].cpu().numpy(), depth.numpy(), rtol=0, atol=2e-3 if two_pass else 2e-6)
        np.testing.assert_allclose(out[2].cpu().numpy()[~tie.numpy()], add_s.numpy()[~tie.numpy()], rtol=2e-3 if two_pass else 1e-5, atol=2e-3 if two_pass else 1e-5)
    if n_pts != 48:                  # the unfolded first layer (module-level forward, SAMPLER_F32_FULL) exists for 48 ray points only — and says so
        with pytest.raises(ops.PnrfError, match='N_point_ray_enc'):
            ops.PackedMLP(ops.NET_SAMPLER, w['W'], w['b']).forward(torch.zeros(4, 6 * n_pts, device=dev))
        with pytest.raises(ops.PnrfError, match='N_point_ray_enc'):
            ops.sampler_fwd(ops.PackedMLP(ops.NET_SAMPLER, w['W'], w['b'], variant='sampler_f32_full'), rays)


@pytest.mark.parametrize('nb', [1, 2, 3, 5, 6, 7, 8])
def test_refine_stage_for_every_num_neighbor_vs_oracle(dev, nb):
    """num_neighbor 1 .. 8 (4 is everywhere else): the exact projection operator against the oracle's refine input (fp32, bit-level taps), the refine
    net on it against the oracle, and the projecting refine stage (the frame path) against projection + refine; module-level forward too."""
    from pronerf_amd import ops
    mmd = 6 if nb % 2 else 3
    w = synth.make_weights(nb, 'trained', num_neighbor=nb, mmnetdepth=mmd)
    scene = synth.make_scene(nb, H=40, W=52, Hf=44, Wf=60, rotate=True, sigma_t=0.15, n_views=nb + 1)
    fr = orc.frame_setup(scene, num_neighbor=nb)
    with torch.no_grad():
        ref = orc.render_rays_infer(w, fr['rays'], fr['or_rays'], fr['images'], fr['proj'])
    rays, or_rays = fr['rays'].to(dev), fr['or_rays'].to(dev)
    img4 = ops.images_pack(fr['images'].to(dev).contiguous())
    proj = fr['proj'].to(dev)
    ds = ref['depth_sorted'].to(dev)
    rin = ops.refine_input(rays, or_rays, ds, img4, proj)
    assert tuple(rin.shape) == (rays.shape[0], 48 + 24 * nb)
    np.testing.assert_allclose(rin.cpu().numpy(), ref['refine_in'].numpy(), rtol=0, atol=1e-4)
    mlp = ops.PackedMLP(ops.NET_REFINE, w['refine']['W'], w['refine']['b'])
    z, pts = ops.refine_fwd(mlp, rin, rays, ds)
    np.testing.assert_allclose(z.cpu().numpy(), ref['z'].numpy(), rtol=0, atol=2e-3)
    np.testing.assert_allclose(pts.cpu().numpy(), ref['pts'].numpy(), rtol=0, atol=4e-3)
    z2, pts2 = ops.refine_project_fwd(mlp, rays, or_rays, ds, img4, proj)
    np.testing.assert_allclose(z2.cpu().numpy(), ref['z'].numpy(), rtol=0, atol=2e-3)
    np.testing.assert_allclose(pts2.cpu().numpy(), ref['pts'].numpy(), rtol=0, atol=4e-3)
    y = mlp.forward(ref['refine_in'].to(dev).contiguous(), head_act=True)
    np.testing.assert_allclose(y[:, :8].cpu().numpy(), ref['refine_depth'].numpy(), rtol=0, atol=4e-3)
    with pytest.raises(ops.PnrfError):          # the stage refuses a view count that is not its net's
        more = torch.cat([fr['images'], fr['images'][:1]]).to(dev).contiguous()
        ops.refine_project_fwd(mlp, rays, or_rays, ds, ops.images_pack(more), torch.cat([proj, proj[:1]]).contiguous())


@pytest.mark.parametrize('netdepth', [3, 4, 5, 6, 7])
def test_nerf_stage_for_every_netdepth_vs_oracle(dev, netdepth):
    from pronerf_amd import ops
    w = synth.make_weights(netdepth, 'trained', netdepth=netdepth)
    scene = synth.make_scene(netdepth, H=40, W=52, rotate=True)
    fr = orc.frame_setup(scene)
    with torch.no_grad():
        ref = orc.render_rays_infer(w, fr['rays'], fr['or_rays'], fr['images'], fr['proj'])
    for variant in ('default', 'f16', 'nerf_4x64'):
        mlp = ops.PackedMLP(ops.NET_NERF, w['nerf']['W'], w['nerf']['b'], variant=variant)
        rgbd, raw = ops.nerf_fwd(mlp, ref['pts'].to(dev), fr['rays'].to(dev), ref['z'].to(dev), ref['add_sorted'].to(dev), ref['mul_sorted'].to(dev), want_raw=True)
        rel = float((raw.cpu().double() - ref['raw'].double()).norm() / ref['raw'].double().norm())
        assert rel < 2e-2 and orc.psnr(rgbd[:, :3].cpu(), ref['rgb']) > 46.4, (netdepth, variant, rel)
    with pytest.raises(ops.PnrfError, match='netdepth'):          # the 32x32x16 form (a variant, and the module-level forward) is the Fern depth's
        ops.nerf_fwd(ops.PackedMLP(ops.NET_NERF, w['nerf']['W'], w['nerf']['b'], variant='bf16_32x32'), ref['pts'].to(dev), fr['rays'].to(dev), ref['z'].to(dev),
                     ref['add_sorted'].to(dev), ref['mul_sorted'].to(dev))


def test_engine_files_keep_the_shape(dev, tmp_path):
    from pronerf_amd import ops
    from pronerf_amd.render import Renderer
    shape = dict(n_pts=32, mmnetdepth=8, num_neighbor=3, netdepth=6)
    w = synth.make_weights(2, 'trained', **shape)
    scene = synth.make_scene(2, H=24, W=32, rotate=True, n_views=5)
    a = Renderer(w, max_rays=24 * 32, device=dev)
    a.save_engines(str(tmp_path))
    b = Renderer.from_engines(str(tmp_path), 24 * 32, device=dev)
    assert b.num_neighbor == 3 and b.sampler.in_dim == 192
    outs = []
    for r in (a, b):
        r.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
        rays, or_rays = r.frame_rays(scene['K'], scene['c2w'], 24, 32)
        outs.append(r.render_rays(rays, or_rays, want_idx=True))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    blob = bytearray(open(os.path.join(str(tmp_path), 'refine_net.pnrf'), 'rb').read())
    with pytest.raises(ops.PnrfError):          # an engine whose header claims another shape than its sections hold is refused
        import struct
        off = 128 - 8                             # nhid, nb, npts: the three 16-bit fields in front of the last two reserved bytes
        nhid, nb, npts = struct.unpack_from('<HHH', blob, off)
        assert (nhid, nb, npts) == (7, 3, 0)
        struct.pack_into('<HHH', blob, off, nhid, 4, npts)
        ops.PackedMLP.deserialize(bytes(blob))


def test_unsupported_shapes_name_the_supported_set(dev):
    from pronerf_amd import ops
    rs = np.random.RandomState(0)
    mk = lambda dims: ([rs.randn(o, i).astype(np.float32) * 0.05 for i, o in zip(dims[:-1], dims[1:])], [np.zeros(o, np.float32) for o in dims[1:]])
    bad = [(ops.NET_SAMPLER, [288] + [128] * 6 + [27]),            # netwidth 128
           (ops.NET_SAMPLER, [288] + [256] * 6 + [30]),            # N_samples 9
           (ops.NET_SAMPLER, [100] + [256] * 6 + [27]),            # not 6 P
           (ops.NET_REFINE, [48 + 24 * 9] + [256] * 6 + [35]),     # 9 neighbour views
           (ops.NET_REFINE, [150] + [256] * 6 + [35]),
           (ops.NET_NERF, [63] + [256] * 8 + [4]),                 # DoNeRFTRT D = 9 (skip='auto' moves the view input into a hidden layer; also the wrong last-layer width)
           (ops.NET_NERF, [60] + [256] * 7 + [4])]                 # another multires
    for net, dims in bad:
        W, b = mk(dims)
        if net == ops.NET_NERF:
            W[-1] = rs.randn(4, 283).astype(np.float32)
        with pytest.raises(ops.PnrfError, match='supported'):
            ops.PackedMLP(net, W, b)
    from pronerf_amd import run_nerf_helpers as h
    with pytest.raises(ops.PnrfError):
        h.DoNeRFTRT(D=9, W=256, skip='auto', n_in=90, n_out=4).to(dev).packed()
    with pytest.raises(ops.PnrfError):
        h.MinMaxRaySamplerTRT_Net(D=6, W=128, input_ch=288, output_ch=27, skips=[10000]).to(dev).packed()
