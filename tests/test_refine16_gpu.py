"""GPU: the refine stage on the 16x16x32 engine (``refine16_kernel``, ``PNRF_VARIANT_REFINE_16X16``) against the oracle and against
the stage on the 32x32x16 engine it stands beside (``refine_kernel<.., PrecF16>``): the same fp16 operands and the same per-ray arithmetic, four lanes per
ray instead of two, fp32 accumulation in another order — so the two agree to fp16 noise and both meet the same bars against the oracle
(run_S_eS_eN_alter_trt.py:637-681; run_nerf_helpers.py:1526-1540)."""
import numpy as np
import pytest
import torch

from oracle import pronerf_oracle as orc
from oracle import synth

pytestmark = pytest.mark.gpu


def cu(t, dev):
    return (t if isinstance(t, torch.Tensor) else torch.from_numpy(np.asarray(t))).to(dev)


def relrms(a, b):
    return float(((a - b).double().pow(2).mean() / b.double().pow(2).mean()).sqrt())


@pytest.fixture(scope='module')
def dev():
    from pronerf_amd import _lib
    _lib.load()
    return torch.device('cuda:0')


def test_both_heads_against_the_oracle_and_the_32x32_stage(dev):
    from pronerf_amd import ops
    for seed, H, W, Hf, Wf, sig in ((0, 30, 41, 30, 41, 0.05), (2, 20, 28, 48, 64, 0.05), (3, 16, 24, 16, 24, 0.6)):      # the last: out-of-image taps
        w = synth.make_weights(seed, 'trained')
        old = ops.PackedMLP(ops.NET_REFINE, w['refine']['W'], w['refine']['b'])
        new = ops.PackedMLP(ops.NET_REFINE, w['refine']['W'], w['refine']['b'], variant='refine_16x16')
        scene = synth.make_scene(seed, H=H, W=W, Hf=Hf, Wf=Wf, rotate=True, sigma_t=sig)
        fr = orc.frame_setup(scene)
        o = orc.render_rays_infer(w, fr['rays'], fr['or_rays'], fr['images'], fr['proj'])
        rays, orr, ds = cu(fr['rays'], dev), cu(fr['or_rays'], dev), cu(o['depth_sorted'], dev)
        img4, proj = ops.images_pack(cu(fr['images'], dev)), cu(fr['proj'], dev)
        rin = ops.refine_input(rays, orr, ds, img4, proj)
        for name, run in (('rows', lambda m: ops.refine_fwd(m, rin, rays, ds)), ('projecting', lambda m: ops.refine_project_fwd(m, rays, orr, ds, img4, proj))):
            (z0, p0), (z1, p1) = run(old), run(new)
            assert bool(torch.isfinite(z1).all()) and bool(torch.isfinite(p1).all())
            # rows: the same operands, another summation order — fp32 round-off in front of a sigmoid of slope <= 1/4 on an interval <= 1.
            # projecting: the 16x16 head runs the first layer with the eight Pluecker 6-vectors folded into one (the moment (o + t d) x d^ does not depend
            # on t; pnrf_layout.h) — the folded weights are rounded to fp16 once instead of eight times: fp16 operand noise, the bar between the two paths of
            # tests/test_ops_gpu.py::test_refine_with_projection_in_the_kernel
            tight = name == 'rows'
            np.testing.assert_allclose(z1.cpu().numpy(), z0.cpu().numpy(), rtol=0, atol=2e-4 if tight else 2e-3, err_msg=name)
            np.testing.assert_allclose(p1.cpu().numpy(), p0.cpu().numpy(), rtol=0, atol=5e-4 if tight else 3e-3, err_msg=name)
            assert relrms(z1.cpu(), z0.cpu()) < (3e-5 if tight else 3e-4), (name, relrms(z1.cpu(), z0.cpu()))
            print(f'[refine 16x16] seed {seed} {name}: z vs the 32x32 stage rel. RMS {relrms(z1.cpu(), z0.cpu()):.2e}, vs the oracle {relrms(z1.cpu(), o["z"]):.2e} '
                  f'(32x32 stage: {relrms(z0.cpu(), o["z"]):.2e})')
            # ... and the oracle, at the bars of tests/test_ops_gpu.py
            np.testing.assert_allclose(z1.cpu().numpy(), o['z'].numpy(), rtol=0, atol=3e-3)
            np.testing.assert_allclose(p1.cpu().numpy(), o['pts'].numpy(), rtol=0, atol=5e-3)
            assert relrms(z1.cpu(), o['z']) < (3e-4 if name == 'rows' else 2e-3)
        # ragged prefix, narrow and wide workgroups: the same rows bit for bit
        z2, p2 = ops.refine_project_fwd(new, rays, orr, ds, img4, proj)
        z3, p3 = ops.refine_project_fwd(new, rays[:77].contiguous(), orr[:77].contiguous(), ds[:77].contiguous(), img4, proj)
        assert torch.equal(z3, z2[:77]) and torch.equal(p3, p2[:77])
        for shape in ('wide', 'narrow'):
            m = ops.PackedMLP(ops.NET_REFINE, w['refine']['W'], w['refine']['b'], variant='refine_16x16')
            m.set_shape(shape)
            z4, p4 = ops.refine_project_fwd(m, rays, orr, ds, img4, proj)
            assert torch.equal(z4, z2) and torch.equal(p4, p2), shape


@pytest.mark.parametrize('nb', [1, 2, 3, 4, 5, 6, 7, 8])
def test_every_num_neighbor_and_depth_parity(dev, nb):
    """num_neighbor 1 .. 8 (one or two views per lane group; a view beyond nb is padding) with an even and an odd hidden-layer count."""
    from pronerf_amd import ops
    for mmnetdepth in (6, 3):
        w = synth.make_weights(nb, 'trained', mmnetdepth=mmnetdepth, num_neighbor=nb)
        scene = synth.make_scene(nb, H=18, W=25, Hf=20, Wf=30, rotate=True, n_views=nb + 1)
        fr = orc.frame_setup(scene, num_neighbor=nb)
        o = orc.render_rays_infer(w, fr['rays'], fr['or_rays'], fr['images'], fr['proj'])
        rays, orr, ds = cu(fr['rays'], dev), cu(fr['or_rays'], dev), cu(o['depth_sorted'], dev)
        img4, proj = ops.images_pack(cu(fr['images'], dev)), cu(fr['proj'], dev)
        new = ops.PackedMLP(ops.NET_REFINE, w['refine']['W'], w['refine']['b'], variant='refine_16x16')
        z, p = ops.refine_project_fwd(new, rays, orr, ds, img4, proj)
        np.testing.assert_allclose(z.cpu().numpy(), o['z'].numpy(), rtol=0, atol=3e-3, err_msg=f'nb {nb} depth {mmnetdepth}')
        np.testing.assert_allclose(p.cpu().numpy(), o['pts'].numpy(), rtol=0, atol=5e-3)
        assert relrms(z.cpu(), o['z']) < 2e-3
        z1, p1 = ops.refine_fwd(new, cu(o['refine_in'], dev), rays, ds)
        np.testing.assert_allclose(z1.cpu().numpy(), o['z'].numpy(), rtol=0, atol=3e-3)
        assert relrms(z1.cpu(), o['z']) < 3e-4, (nb, mmnetdepth, relrms(z1.cpu(), o['z']))


def test_engine_file_round_trip_keeps_the_stream(dev, tmp_path):
    from pronerf_amd import ops
    w = synth.make_weights(0, 'trained')
    a = ops.PackedMLP(ops.NET_REFINE, w['refine']['W'], w['refine']['b'], variant='refine_16x16')
    b = ops.PackedMLP.deserialize(a.serialize())
    b.set_variant('refine_16x16')
    rs = np.random.RandomState(0)
    n = 300
    rin = cu(rs.uniform(0, 1, (n, 144)).astype(np.float32), dev)
    rays = cu(np.concatenate([rs.uniform(-1, 1, (n, 6)), np.zeros((n, 1)), np.ones((n, 1)), rs.uniform(-1, 1, (n, 3))], 1).astype(np.float32), dev)
    ds = cu(np.sort(rs.uniform(0.05, 0.95, (n, 8)), 1).astype(np.float32), dev)
    za, pa = ops.refine_fwd(a, rin, rays, ds)
    zb, pb = ops.refine_fwd(b, rin, rays, ds)
    assert torch.equal(za, zb) and torch.equal(pa, pb)
    with pytest.raises(ops.PnrfError):
        ops.PackedMLP(ops.NET_NERF, w['nerf']['W'], w['nerf']['b'], variant='refine_16x16')


def test_frame_with_the_16x16_refine_stage(dev):
    """The fused path end to end with the variant: the same sampler indices (the stage sits behind the sort), rgb / depth against the oracle at the
    frame bars, and against the default renderer at fp16 noise."""
    from pronerf_amd.render import Renderer
    w = synth.make_weights(0, 'trained')
    scene = synth.make_scene(0, H=40, W=52, rotate=True)
    fr = orc.frame_setup(scene)
    o = orc.render_rays_infer(w, fr['rays'], fr['or_rays'], fr['images'], fr['proj'])
    out = {}
    for name, variants in (('default', None), ('r16', {'refine': 'refine_16x16'})):
        rend = Renderer(w, max_rays=40 * 52, device=dev, variants=variants)
        rend.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
        rgbd, idx = rend.render_rays(cu(fr['rays'], dev), cu(fr['or_rays'], dev), want_idx=True)
        out[name] = rgbd.cpu()
        np.testing.assert_array_equal(idx.cpu().numpy(), o['sort_idx'].numpy())
        assert orc.psnr(rgbd[:, :3].cpu(), o['rgb']) > 46.4
    assert orc.psnr(out['r16'][:, :3], out['default'][:, :3]) > 60.0, orc.psnr(out['r16'][:, :3], out['default'][:, :3])
    print(f"\n[refine 16x16] frame vs the oracle: default {orc.psnr(out['default'][:, :3], o['rgb']):.1f} dB, 16x16 {orc.psnr(out['r16'][:, :3], o['rgb']):.1f} dB; "
          f"16x16 vs default {orc.psnr(out['r16'][:, :3], out['default'][:, :3]):.1f} dB")


def test_frame_size_rows_agree_with_the_32x32_stage(dev):
    """BASELINE configs[1] size (762 048 rays, 1008 x 756): the two refine stages over the same sampler output — every row finite, inside its interval, and
    within fp16 noise of the other stage; wide (the launch's own shape) and forced-narrow workgroups bit-identical."""
    from pronerf_amd import ops, synthetic
    H, W = 756, 1008
    w = synth.make_weights(0, 'trained')
    scene = synthetic.make_scene(0, H=H, W=W, focal=815.13, rotate=True)
    rays, orr = ops.frame_rays(scene['K'], scene['c2w'], H, W, device=dev)
    sampler = ops.PackedMLP(ops.NET_SAMPLER, w['sampler']['W'], w['sampler']['b'])
    ds = ops.sampler_fwd(sampler, rays, two_pass=True, want_idx=False, want_rgb=False)[0]
    from pronerf_amd.render import Renderer
    rend = Renderer(w, max_rays=H * W, device=dev)
    rend.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
    old = ops.PackedMLP(ops.NET_REFINE, w['refine']['W'], w['refine']['b'])
    new = ops.PackedMLP(ops.NET_REFINE, w['refine']['W'], w['refine']['b'], variant='refine_16x16')
    z0, p0 = ops.refine_project_fwd(old, rays, orr, ds, rend.img4, rend.proj)
    z1, p1 = ops.refine_project_fwd(new, rays, orr, ds, rend.img4, rend.proj)
    assert bool(torch.isfinite(z1).all()) and bool(torch.isfinite(p1).all())
    e = torch.cat([rays[:, 6:7], ds, rays[:, 7:8]], 1)                                  # [near, d0..d7, far]: sample s lies between the midpoints around d_s (trt.py:673-676)
    lo, hi = 0.5 * (e[:, :-2] + e[:, 1:-1]), 0.5 * (e[:, 1:-1] + e[:, 2:])
    assert bool(((z1 >= lo - 1e-6) & (z1 <= hi + 1e-6)).all())
    dz = (z1 - z0).abs()
    print(f'\n[refine 16x16] frame size: max |dz| {float(dz.max()):.2e}, rel. RMS {relrms(z1.cpu(), z0.cpu()):.2e}, max |dpts| {float((p1 - p0).abs().max()):.2e}')
    assert float(dz.max()) < 2e-3 and relrms(z1.cpu(), z0.cpu()) < 3e-4 and float((p1 - p0).abs().max()) < 3e-3
    m = ops.PackedMLP(ops.NET_REFINE, w['refine']['W'], w['refine']['b'], variant='refine_16x16')
    m.set_shape('narrow')
    z2, p2 = ops.refine_project_fwd(m, rays, orr, ds, rend.img4, rend.proj)
    assert torch.equal(z2, z1) and torch.equal(p2, p1)
