"""GPU: every C-ABI entry point against the CPU oracle on the same seeded inputs.

Tolerances: fp32 kernels (element-wise ops, sampler) are held to fp32 round-off; the bf16 MLP
stages (refine, nerf) to the tolerance BASELINE.md states for the path — RGB/depth PSNR >=
46.4 dB against the fp32 oracle (an uncorrelated error of that size moves a 27 dB image PSNR
by <= 0.05 dB) — and a relative-RMS bound on the raw network outputs.
"""
import os

import numpy as np
import pytest
import torch

from oracle import pronerf_oracle as orc
from oracle import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    from pronerf_amd import _lib
    _lib.load()          # fail loudly if the HIP library is missing
    return torch.device('cuda:0')


def cu(x, dev):
    return torch.as_tensor(x, dtype=torch.float32).to(dev).contiguous()


def relrms(a, b):
    a = a.double(); b = b.double()
    return float(((a - b) ** 2).mean().sqrt() / (b ** 2).mean().sqrt().clamp_min(1e-30))


def test_posenc_plucker(dev):
    from pronerf_amd import ops
    rs = np.random.RandomState(0)
    x = torch.from_numpy(rs.uniform(-1.5, 1.5, (1000, 3)).astype(np.float32))
    for nf in (10, 4, 0):
        got = ops.posenc(cu(x, dev), nf).cpu()
        np.testing.assert_allclose(got.numpy(), orc.posenc(x, nf).numpy(), rtol=0, atol=2e-6)
    o = torch.from_numpy(rs.randn(777, 3).astype(np.float32)); d = torch.from_numpy(rs.randn(777, 3).astype(np.float32))
    np.testing.assert_allclose(ops.plucker(cu(o, dev), cu(d, dev)).cpu().numpy(), orc.pluecker(o, d).numpy(), rtol=0, atol=1e-6)
    assert ops.posenc(torch.zeros(0, 3, device=dev), 10).shape == (0, 63)


def test_operator_goldens(dev, golden_dir):
    """The same fixtures that pin the oracle, straight against the HIP operators."""
    from pronerf_amd import ops
    g = dict(np.load(os.path.join(golden_dir, 'operators.npz')))
    np.testing.assert_allclose(ops.posenc(cu(g['pe_x'], dev), 10).cpu().numpy(), g['pe10'], rtol=0, atol=2e-6)
    np.testing.assert_array_equal(ops.plucker(cu(g['pl_o'], dev), cu(g['pl_d'], dev)).cpu().numpy(), g['pl'])
    # ray set-up: bit for bit what torch computes on the CPU (same operation order, IEEE divisions and square root, torch.norm's FMA chain)
    rays, orr = ops.frame_rays(g['gr_K'], g['gr_c2w'], 9, 13, device=dev)
    np.testing.assert_array_equal(orr[:, 3:6].cpu().numpy().reshape(9, 13, 3), g['gr_d'])
    np.testing.assert_array_equal(rays[:, 0:3].cpu().numpy().reshape(9, 13, 3), g['ndc_o'])
    np.testing.assert_array_equal(rays[:, 3:6].cpu().numpy().reshape(9, 13, 3), g['ndc_d'])
    out = ops.warp_trt(cu(g['wp_img'], dev), cu(g['wp_depth'][:, 0, :], dev), cu(g['wp_ro1'], dev), cu(g['wp_rd1'], dev), cu(g['wp_w2c'], dev))
    np.testing.assert_allclose(out.cpu().numpy(), g['wp_out'][:, :, 0, :], rtol=0, atol=1e-5)
    r = ops.composite(cu(g['c_raw'], dev), cu(g['c_z'], dev), cu(g['c_d'], dev), cu(g['c_add'], dev), cu(g['c_mul'], dev))
    for got, key in zip(r, ('c_rgb', 'c_disp', 'c_acc', 'c_w', 'c_depth')):
        np.testing.assert_allclose(got.cpu().numpy(), g[key], rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize('H,W,rot', [(24, 32, False), (17, 23, True)])
def test_frame_rays_and_ray_encode(dev, H, W, rot):
    from pronerf_amd import ops
    scene = synth.make_scene(3, H=H, W=W, rotate=rot)
    fr = orc.frame_setup(scene)
    rays, orr = ops.frame_rays(scene['K'], scene['c2w'], H, W, device=dev)
    np.testing.assert_array_equal(rays.cpu().numpy(), fr['rays'].numpy())                     # 0 ulp: origins, directions, NDC, view directions
    np.testing.assert_array_equal(orr.cpu().numpy(), fr['or_rays'].numpy())
    sub, _ = ops.frame_rays(scene['K'], scene['c2w'], H, W, first=37, count=101, device=dev)    # ray-range sharding
    np.testing.assert_array_equal(sub.cpu().numpy(), rays[37:138].cpu().numpy())
    mm = ops.ray_encode(cu(fr['rays'], dev), 48).cpu()
    np.testing.assert_array_equal(mm.numpy(), fr['mm_input'].numpy())


def test_composite_variants(dev):
    from pronerf_amd import ops
    rs = np.random.RandomState(5)
    for S in (8, 2, 64):       # S=1 is degenerate in the reference (empty dists[..., :1])
        N = 301
        raw = torch.from_numpy((rs.randn(N, S, 4) * 4).astype(np.float32))
        z = torch.sort(torch.from_numpy(rs.rand(N, S).astype(np.float32)), -1)[0]
        d = torch.from_numpy(rs.randn(N, 3).astype(np.float32))
        add = torch.from_numpy(rs.randn(N, S).astype(np.float32)); mul = torch.from_numpy(rs.randn(N, S).astype(np.float32) + 0.5)
        noise = torch.from_numpy(rs.randn(N, S).astype(np.float32))
        for kw in (dict(add=add, mul=mul), dict(), dict(add=add, mul=mul, noise=noise, clamp=10.0, white_bkgd=True)):
            ref = orc.raw2outputs(raw, z, d, **kw)
            gkw = {k: (cu(v, dev) if isinstance(v, torch.Tensor) else v) for k, v in kw.items()}
            got = ops.composite(cu(raw, dev), cu(z, dev), cu(d, dev), **gkw)
            for a, b, name in zip(got, ref, ('rgb', 'disp', 'acc', 'w', 'depth')):
                if name == 'disp':       # 1/max(1e-10, depth/acc) blows up where acc ~ 0: compare where it is conditioned
                    m = (ref[2] > 1e-3).numpy()
                    np.testing.assert_allclose(a.cpu().numpy()[m], b.numpy()[m], rtol=2e-4, atol=1e-5)
                else:
                    np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), rtol=2e-5, atol=2e-6, err_msg=f'{name} S={S} {list(kw)}')


def _packed(dev, seed, kind):
    from pronerf_amd import ops
    w = synth.make_weights(seed, kind)
    mlps = {
        'sampler': ops.PackedMLP(ops.NET_SAMPLER, w['sampler']['W'], w['sampler']['b']),
        'refine': ops.PackedMLP(ops.NET_REFINE, w['refine']['W'], w['refine']['b']),
        'nerf': ops.PackedMLP(ops.NET_NERF, w['nerf']['W'], w['nerf']['b']),
    }
    return w, mlps


@pytest.mark.parametrize('kind', ['trained', 'default'])
@pytest.mark.parametrize('m', [1, 128, 333])
def test_mlp_module_forward(dev, kind, m):
    """pnrf_mlp_fwd (x -> raw Linear output) for the three nets, ragged row counts."""
    w, mlps = _packed(dev, 0, kind)
    rs = np.random.RandomState(m)
    x = torch.from_numpy(rs.uniform(-1, 1, (m, 288)).astype(np.float32))
    ref = orc.mlp_elu_backbone(x, w['sampler']['W'], w['sampler']['b'])
    got = mlps['sampler'].forward(cu(x, dev)).cpu()
    assert relrms(got, ref) < 2e-6, relrms(got, ref)                      # fp32 MFMA: exact-fp32 FMA chain
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-4, atol=2e-5)
    x = torch.from_numpy(rs.uniform(-1, 1, (m, 144)).astype(np.float32))
    ref = orc.mlp_elu_backbone(x, w['refine']['W'], w['refine']['b'])
    got = mlps['refine'].forward(cu(x, dev)).cpu()
    assert relrms(got, ref) < 1.5e-2, relrms(got, ref)                    # bf16 inputs, fp32 accumulate
    x = torch.from_numpy(rs.uniform(-1, 1, (m, 63)).astype(np.float32)); xv = torch.from_numpy(rs.uniform(-1, 1, (m, 27)).astype(np.float32))
    ref = orc.nerf_forward(w['nerf'], x, xv)
    got = mlps['nerf'].forward(cu(x, dev), cu(xv, dev)).cpu()
    assert relrms(got, ref) < 1.5e-2, relrms(got, ref)


def test_mlp_linearity_exact_integers(dev):
    """MFMA operand-layout check with exactly representable data: a ReLU net with small-integer
    weights and inputs has an exact answer in bf16/fp32, so any lane/row/k permutation error in the
    packed weight stream shows up as a wrong integer (asymmetric weights, cf. cdna guide §3)."""
    from pronerf_amd import ops
    rs = np.random.RandomState(11)
    dims = synth.nerf_layer_dims()
    W = [rs.randint(-1, 2, (fo, fi)).astype(np.float32) * (rs.rand(fo, fi) < 0.03) for fi, fo in dims]   # sparse: activations stay small
    b = [rs.randint(-2, 3, (fo,)).astype(np.float32) for _, fo in dims]
    mlp = ops.PackedMLP(ops.NET_NERF, W, b)
    x = torch.from_numpy(rs.randint(-2, 3, (200, 63)).astype(np.float32)); xv = torch.from_numpy(rs.randint(-2, 3, (200, 27)).astype(np.float32))
    h = x
    acts_max = 0.0
    for l in range(len(W)):
        if l == len(W) - 1:
            h = torch.cat([h, xv], -1)
        h = h @ torch.from_numpy(W[l]).T + torch.from_numpy(b[l])
        if l + 1 < len(W):
            h = torch.relu(h)
            acts_max = max(acts_max, float(h.abs().max()))
    assert acts_max <= 256, acts_max        # every activation is an integer <= 2^8: exactly representable in bf16
    got = mlp.forward(cu(x, dev), cu(xv, dev)).cpu()
    np.testing.assert_array_equal(got.numpy(), h.numpy())


@pytest.mark.parametrize('prec', ['default', 'sampler_f32', 'sampler_f32_full'])
@pytest.mark.parametrize('kind,seed', [('trained', 0), ('spread', 2), ('default', 1)])
def test_sampler_stage(dev, kind, seed, prec):
    """The sampler's kernel variants (pnrf_mlp_set_variant): split fp16 (default; 22-bit operands, fp32 accumulate), the exact-fp32 MFMA
    chain with the folded first layer, and the exact-fp32 chain on all 288 inputs."""
    from pronerf_amd import ops
    w, mlps = _packed(dev, seed, kind)
    mlps['sampler'].set_variant(prec)
    scene = synth.make_scene(seed, H=40, W=52, rotate=True)
    fr = orc.frame_setup(scene)
    rays = fr['rays']
    mm_rgb, add, mul, depth = orc.sampler_forward(w['sampler'], fr['mm_input'])
    ds, idx, adds, muls = orc.sort_gather(depth, add, mul, rays[:, 6:7], rays[:, 7:8])
    g_ds, g_idx, g_add, g_mul, g_rgb, g_raw = ops.sampler_fwd(mlps['sampler'], cu(rays, dev), want_raw=True)
    np.testing.assert_allclose(g_raw.cpu().numpy(), depth.numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(g_rgb.cpu().numpy(), mm_rgb.numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(g_ds.cpu().numpy(), ds.numpy(), rtol=0, atol=2e-6)
    gap = (ds[:, 1:] - ds[:, :-1]).min(dim=1)[0]
    tie_free = (gap > 1e-6).numpy()
    # "sampler indices bit-exact": identical on every ray whose sorted depths are separated by more than
    # fp32 summation-order noise; the tie set is reported, and must be empty for spread/trained weights
    np.testing.assert_array_equal(g_idx.cpu().numpy()[tie_free], idx.numpy()[tie_free])
    if kind != 'default':
        assert tie_free.all()
    np.testing.assert_allclose(g_add.cpu().numpy()[tie_free], adds.numpy()[tie_free], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(g_mul.cpu().numpy()[tie_free], muls.numpy()[tie_free], rtol=1e-5, atol=1e-5)
    # property that holds regardless of ties: output is sorted and idx is a permutation that sorts depth_raw
    gd = g_ds.cpu(); gi = g_idx.cpu()
    assert bool((gd[:, 1:] >= gd[:, :-1]).all())
    assert bool((torch.sort(gi, dim=1)[0] == torch.arange(8)[None]).all())
    np.testing.assert_array_equal(torch.gather(g_raw.cpu(), 1, gi).numpy() * 1.0, (gd.numpy() - 0.0))


def test_refine_input_stage(dev):
    from pronerf_amd import ops
    for seed, Hf, Wf, sig in ((0, 24, 32, 0.05), (2, 48, 64, 0.05), (3, 16, 24, 0.6)):
        w = synth.make_weights(seed, 'trained')
        scene = synth.make_scene(seed, H=16, W=24, Hf=Hf, Wf=Wf, rotate=True, sigma_t=sig)
        fr = orc.frame_setup(scene)
        o = orc.render_rays_infer(w, fr['rays'], fr['or_rays'], fr['images'], fr['proj'])
        img4 = ops.images_pack(cu(fr['images'], dev))
        np.testing.assert_array_equal(img4[..., :3].cpu().numpy(), fr['images'].permute(0, 2, 3, 1).numpy())
        got = ops.refine_input(cu(fr['rays'], dev), cu(fr['or_rays'], dev), cu(o['depth_sorted'], dev), img4, cu(fr['proj'], dev)).cpu()
        np.testing.assert_allclose(got[:, :48].numpy(), o['refine_in'][:, :48].numpy(), rtol=0, atol=1e-6)
        np.testing.assert_allclose(got[:, 48:].numpy(), o['epi'].numpy(), rtol=0, atol=2e-4)
        if sig > 0.3:
            frac0 = float((o['epi'] == 0).float().mean())
            assert 0.02 < frac0 < 0.98         # the case exercises out-of-image taps


def test_refine_with_projection_in_the_kernel(dev):
    """pnrf_refine_project_fwd (the refine stage of the fused path: projection + colour fetch + sample Pluecker in the head of the refine
    kernel) against pnrf_refine_input_fwd + pnrf_refine_fwd (the [n,144] intermediate in memory) and against the oracle.  The in-kernel
    projection uses the linear form p(z) = A + z B and v_rcp_f32 (pnrf_geom.h): pixel coordinates differ by fp32 ulps, colours by ~1e-4 of the
    local texel difference, which is below the bf16 rounding of the MLP input — so the two paths agree to bf16 noise, not bit for bit."""
    from pronerf_amd import ops
    for seed, H, W, Hf, Wf, sig in ((0, 30, 41, 30, 41, 0.05), (2, 20, 28, 48, 64, 0.05), (3, 16, 24, 16, 24, 0.6)):      # the last: out-of-image taps
        w, mlps = _packed(dev, seed, 'trained')
        scene = synth.make_scene(seed, H=H, W=W, Hf=Hf, Wf=Wf, rotate=True, sigma_t=sig)
        fr = orc.frame_setup(scene)
        o = orc.render_rays_infer(w, fr['rays'], fr['or_rays'], fr['images'], fr['proj'])
        rays, orr, ds = cu(fr['rays'], dev), cu(fr['or_rays'], dev), cu(o['depth_sorted'], dev)
        img4, proj = ops.images_pack(cu(fr['images'], dev)), cu(fr['proj'], dev)
        z1, p1 = ops.refine_fwd(mlps['refine'], ops.refine_input(rays, orr, ds, img4, proj), rays, ds)
        z2, p2 = ops.refine_project_fwd(mlps['refine'], rays, orr, ds, img4, proj)
        # a bf16 input that flips by one unit (2^-8 relative) moves z by ~1e-3 at most; typically the two paths are identical
        np.testing.assert_allclose(z2.cpu().numpy(), z1.cpu().numpy(), rtol=0, atol=2e-3)
        np.testing.assert_allclose(p2.cpu().numpy(), p1.cpu().numpy(), rtol=0, atol=3e-3)
        assert relrms(z2.cpu(), z1.cpu()) < 3e-4
        np.testing.assert_allclose(z2.cpu().numpy(), o['z'].numpy(), rtol=0, atol=3e-3)
        np.testing.assert_allclose(p2.cpu().numpy(), o['pts'].numpy(), rtol=0, atol=5e-3)
        assert relrms(z2.cpu(), o['z']) < 2e-3
        # ragged prefix: same rays, smaller call
        z3, _ = ops.refine_project_fwd(mlps['refine'], rays[:77].contiguous(), orr[:77].contiguous(), ds[:77].contiguous(), img4, proj)
        assert torch.equal(z3, z2[:77])


def test_refine_and_nerf_stages(dev):
    from pronerf_amd import ops
    w, mlps = _packed(dev, 0, 'trained')
    scene = synth.make_scene(0, H=30, W=41, rotate=True)       # 1230 rays: ragged vs the 256/32-ray batches
    fr = orc.frame_setup(scene)
    o = orc.render_rays_infer(w, fr['rays'], fr['or_rays'], fr['images'], fr['proj'])
    rays = cu(fr['rays'], dev)
    z, pts = ops.refine_fwd(mlps['refine'], cu(o['refine_in'], dev), rays, cu(o['depth_sorted'], dev))
    np.testing.assert_allclose(z.cpu().numpy(), o['z'].numpy(), rtol=0, atol=3e-3)       # bf16 MLP in front of a sigmoid
    np.testing.assert_allclose(pts.cpu().numpy(), o['pts'].numpy(), rtol=0, atol=5e-3)
    assert relrms(z.cpu(), o['z']) < 2e-3
    # nerf stage fed with the oracle's refine outputs: isolates the bf16 NeRF MLP + compositing
    rgbd, raw = ops.nerf_fwd(mlps['nerf'], cu(o['pts'], dev), rays, cu(o['z'], dev), cu(o['add_sorted'], dev), cu(o['mul_sorted'], dev), want_raw=True)
    assert relrms(raw.cpu(), o['raw']) < 2e-2, relrms(raw.cpu(), o['raw'])
    ps = orc.psnr(rgbd[:, :3].cpu(), o['rgb'])
    assert ps > 46.4, ps
    # compositing given the kernel's own raw must agree with the oracle's compositing to fp32 round-off
    r = orc.raw2outputs(raw.cpu(), o['z'], fr['rays'][:, 3:6], o['add_sorted'], o['mul_sorted'])
    np.testing.assert_allclose(rgbd[:, :3].cpu().numpy(), r[0].numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(rgbd[:, 3].cpu().numpy(), r[4].numpy(), rtol=0, atol=2e-6)


@pytest.mark.parametrize('net', ['nerf', 'nerfcls'])
def test_nerf_stage_variants_agree(dev, net):
    """PNRF_VARIANT_NERF_4X64 (4 waves of 64 columns) runs the same MFMAs on the same packed bf16 stream in the same accumulation order as
    PNRF_VARIANT_BF16 (8 waves of 32 columns): bit-identical raw outputs and composited pixels, ragged row counts included.
    PNRF_VARIANT_BF16_32X32 (the 32x32x16 engine) contracts in another order, PNRF_VARIANT_F16 runs fp16 operands: bf16-grade agreement."""
    from pronerf_amd import ops
    w = synth.make_weights(0, 'trained')
    if net == 'nerf':
        Ws, bs, kind = w['nerf']['W'], w['nerf']['b'], ops.NET_NERF
    else:
        wc = synth.make_nerfcls_weights(0, head_scale=0.3)
        Ws = [w_ for w_, _ in wc['pts_linears']] + [wc['feature_linear'][0], wc['alpha_linear'][0], wc['views_linears'][0][0], wc['rgb_linear'][0]]
        bs = [b_ for _, b_ in wc['pts_linears']] + [wc['feature_linear'][1], wc['alpha_linear'][1], wc['views_linears'][0][1], wc['rgb_linear'][1]]
        kind = ops.NET_NERFCLS
    rs = np.random.RandomState(3)
    for n in (1, 37, 300):                                          # rays; 8 samples each: 8 .. 2400 columns, 256 per workgroup batch
        rays = torch.from_numpy(rs.uniform(-1, 1, (n, 11)).astype(np.float32)); rays[:, 6] = 0; rays[:, 7] = 1
        z = torch.sort(torch.from_numpy(rs.uniform(0.05, 0.95, (n, 8)).astype(np.float32)), -1)[0]
        pts = torch.from_numpy(rs.uniform(-1, 1, (n, 8, 3)).astype(np.float32))
        add = torch.from_numpy(rs.randn(n, 8).astype(np.float32)); mul = torch.from_numpy(rs.rand(n, 8).astype(np.float32))
        outs = {}
        for var in ('default', 'bf16', 'nerf_4x64', 'bf16_32x32', 'f16'):
            mlp = ops.PackedMLP(kind, Ws, bs, variant=var)
            outs[var] = ops.nerf_fwd(mlp, cu(pts, dev), cu(rays, dev), cu(z, dev), cu(add, dev), cu(mul, dev), want_raw=True)
        for a_, b_ in zip(outs['bf16'], outs['nerf_4x64']):
            assert torch.equal(a_, b_), (net, n)
        assert relrms(outs['bf16_32x32'][1].cpu(), outs['bf16'][1].cpu()) < 2e-2
        assert torch.equal(outs['default'][1], outs['bf16'][1])                                 # bf16 IS the NeRF stage's default
        assert relrms(outs['f16'][1].cpu(), outs['bf16'][1].cpu()) < 2e-2


def test_nerf_class_network(dev, golden_dir):
    """The NeRF class (skip-concat, feature/alpha heads, view branch) — module-level forward against the oracle and
    the reference-generated fixture, and the fused render path with it as the fine net."""
    from pronerf_amd import ops
    from pronerf_amd.render import Renderer
    wc = synth.make_nerfcls_weights(0)
    names = ['pts_linears'] * 8
    Ws = [w for w, _ in wc['pts_linears']] + [wc['feature_linear'][0], wc['alpha_linear'][0], wc['views_linears'][0][0], wc['rgb_linear'][0]]
    bs = [b for _, b in wc['pts_linears']] + [wc['feature_linear'][1], wc['alpha_linear'][1], wc['views_linears'][0][1], wc['rgb_linear'][1]]
    mlp = ops.PackedMLP(ops.NET_NERFCLS, Ws, bs)
    g = dict(np.load(os.path.join(golden_dir, 'operators.npz')))
    x = torch.from_numpy(g['nc_x'])
    got = mlp.forward(cu(x[:, :63], dev), cu(x[:, 63:], dev)).cpu()
    assert relrms(got, torch.from_numpy(g['nc_y'])) < 2e-2, relrms(got, torch.from_numpy(g['nc_y']))
    rs = np.random.RandomState(1)
    for m in (1, 130, 517):
        x = torch.from_numpy(rs.uniform(-1, 1, (m, 90)).astype(np.float32))
        ref = orc.nerfcls_forward(wc, x)
        got = mlp.forward(cu(x[:, :63], dev), cu(x[:, 63:], dev)).cpu()
        assert relrms(got, ref) < 2e-2, (m, relrms(got, ref))
    # fused render with the NeRF-class fine net (trained-like head magnitudes, as for the DoNeRFTRT 'trained' set)
    wc = synth.make_nerfcls_weights(0, head_scale=0.3)
    Ws = [w_ for w_, _ in wc['pts_linears']] + [wc['feature_linear'][0], wc['alpha_linear'][0], wc['views_linears'][0][0], wc['rgb_linear'][0]]
    bs = [b_ for _, b_ in wc['pts_linears']] + [wc['feature_linear'][1], wc['alpha_linear'][1], wc['views_linears'][0][1], wc['rgb_linear'][1]]
    w = synth.make_weights(0, 'trained')
    scene = synth.make_scene(0, H=20, W=27, rotate=True)
    fr = orc.frame_setup(scene)
    o = orc.render_rays_infer({**w, 'nerfcls': wc}, fr['rays'], fr['or_rays'], fr['images'], fr['proj'], nerf='cls')
    rend = Renderer({**w, 'nerf': {'W': Ws, 'b': bs}}, max_rays=20 * 27, device=dev)
    rend.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
    rgbd, idx = rend.render_rays(cu(fr['rays'], dev), cu(fr['or_rays'], dev), want_idx=True)
    np.testing.assert_array_equal(idx.cpu().numpy(), o['sort_idx'].numpy())
    assert orc.psnr(rgbd[:, :3].cpu(), o['rgb']) > 46.4
    np.testing.assert_allclose(rgbd[:, 3].cpu().numpy(), o['depth'].numpy(), rtol=0, atol=2e-2)


def test_fp16_operands_precision_and_saturation(dev):
    """Refine and NeRF stages with fp16 operands (refine: the default; NeRF: PNRF_VARIANT_F16) against the oracle, next to bf16: the raw NeRF output is held to the
    tolerance the reference's authors used for their FP16 TensorRT engines (rtol 1e-3 / atol 1e-5 as a norm-wise bound, trt_infer_v2.py:444),
    which bf16 misses by an order of magnitude.  Then the range: with the first NeRF layer scaled so that hidden activations pass 65 504 the packed
    fp16 activations saturate (v_pk_min_i16 on the packed pair) — every output stays finite, where an unguarded conversion gives inf and NaN."""
    from pronerf_amd import ops
    w, mlps = _packed(dev, 0, 'trained')
    scene = synth.make_scene(0, H=30, W=41, rotate=True)
    fr = orc.frame_setup(scene)
    o = orc.render_rays_infer(w, fr['rays'], fr['or_rays'], fr['images'], fr['proj'])
    rays = cu(fr['rays'], dev)
    args = (cu(o['pts'], dev), rays, cu(o['z'], dev), cu(o['add_sorted'], dev), cu(o['mul_sorted'], dev))
    err = {}
    for variant in ('default', 'bf16'):
        nerf = ops.PackedMLP(ops.NET_NERF, w['nerf']['W'], w['nerf']['b'], variant='f16' if variant == 'default' else 'bf16')
        refine = ops.PackedMLP(ops.NET_REFINE, w['refine']['W'], w['refine']['b'], variant=variant)
        _, raw = ops.nerf_fwd(nerf, *args, want_raw=True)
        z, _ = ops.refine_fwd(refine, cu(o['refine_in'], dev), rays, cu(o['depth_sorted'], dev))
        err[variant] = (relrms(raw.cpu(), o['raw']), relrms(z.cpu(), o['z']))
    print(f'\n[fp16] raw NeRF output rel. RMS vs the oracle: fp16 {err["default"][0]:.2e}, bf16 {err["bf16"][0]:.2e}; refined depths: fp16 {err["default"][1]:.2e}, '
          f'bf16 {err["bf16"][1]:.2e}')
    assert err['default'][0] < 1.5e-3 and err['default'][1] < 3e-4
    assert err['bf16'][0] < 2e-2 and err['default'][0] < 0.25 * err['bf16'][0]
    # saturation: hidden activations of ~1e6
    big = [x.copy() for x in w['nerf']['W']]
    big[0] = big[0] * 3e5
    nerf = ops.PackedMLP(ops.NET_NERF, big, w['nerf']['b'], variant='f16')
    rgbd, raw = ops.nerf_fwd(nerf, *args, want_raw=True)
    h = torch.relu(orc.posenc(o['pts'].reshape(-1, 3), 10) @ torch.from_numpy(big[0]).T + torch.from_numpy(w['nerf']['b'][0]))
    assert float(h.max()) > 65504 * 4                                        # the case does leave the fp16 range
    assert bool(torch.isfinite(raw).all()) and bool(torch.isfinite(rgbd).all())


def test_two_pass_sampler_arguments(dev):
    """pnrf_sampler_fwd_ws: workspace size / alignment and kappa are checked; the single-kernel variants ignore the workspace; ragged ray counts
    (1, 255, 257) agree with the split kernel on every row pass 2 re-rendered and to fp16 grade elsewhere."""
    import ctypes as C
    from pronerf_amd import _lib, ops
    w, mlps = _packed(dev, 0, 'trained')
    lib = _lib.load()
    scene = synth.make_scene(0, H=17, W=23, rotate=True)
    rays = cu(orc.frame_setup(scene)['rays'], dev)
    n = rays.shape[0]
    need = int(lib.pnrf_sampler_workspace_bytes(n))
    assert need == (16 + 2 * n) * 4 and int(lib.pnrf_sampler_workspace_bytes(-5)) == 0            # counters + the lists of passes 2 and 3
    d = torch.empty(n, 8, device=dev); a = torch.empty_like(d); m = torch.empty_like(d)
    ws = torch.empty(need // 4 + 8, device=dev, dtype=torch.int32)
    p = lambda t: C.c_void_p(t.data_ptr())
    call = lambda wsp, nbytes, kappa: lib.pnrf_sampler_fwd_ws(mlps['sampler'].handle, p(rays), n, p(d), p(a), p(m), None, None, None, wsp, nbytes, kappa, None)
    assert call(p(ws), need - 4, -1.0) == -1 and b'workspace' in lib.pnrf_last_error()
    assert call(C.c_void_p(ws.data_ptr() + 4), need, -1.0) == -1                                   # not 16-byte aligned
    assert call(None, need, -1.0) == -1
    assert call(p(ws), need, float('inf')) == -1 and b'kappa' in lib.pnrf_last_error()
    assert call(p(ws), need, float('nan')) == -1 and b'kappa' in lib.pnrf_last_error()              # a NaN threshold would flag every ray
    assert call(p(ws), need, -1.0) == 0
    for k in (1, 255, 257):
        r = rays[:k].contiguous()
        two = ops.sampler_fwd(mlps['sampler'], r, two_pass=True, want_raw=True)
        one = ops.sampler_fwd(mlps['sampler'], r, want_raw=True)
        assert torch.equal(two[1], one[1])                                                         # indices (no ties on these weights)
        np.testing.assert_allclose(two[0].cpu().numpy(), one[0].cpu().numpy(), rtol=0, atol=2e-3)
        assert 0 <= int(two[6]) <= k and int(two[7]) == 0                                          # nothing saturates on these weights
    split = ops.PackedMLP(ops.NET_SAMPLER, w['sampler']['W'], w['sampler']['b'], variant='sampler_split')
    s2 = ops.sampler_fwd(split, rays, two_pass=True, want_raw=True)                                # variant handles run their one kernel
    s1 = ops.sampler_fwd(split, rays, want_raw=True)
    assert all(torch.equal(x, y) for x, y in zip(s2[:6], s1[:6]) if x is not None)


def test_sampler_fp16_range_is_defined(dev):
    """Activations beyond the fp16 range: with two of the sampler's hidden layers scaled x300 (activations up to ~4e5) the
    plain-fp16 pass flags every such ray (|x|^2 at the saturation limit), the split kernel saturates (MODE.FP16_OVFL: no inf / NaN), reports
    them, and the exact-fp32 third pass renders them: every output finite, indices and depths equal to the exact-fp32 variant's on every ray
    that saturated; on the untouched weights the third pass renders nothing."""
    from pronerf_amd import ops
    w = synth.make_weights(0, 'trained')['sampler']
    scene = synth.make_scene(0, H=40, W=52, rotate=True)
    rays = cu(orc.frame_setup(scene)['rays'], dev)
    n = rays.shape[0]
    for gain, expect_sat in ((1.0, False), (300.0, True)):
        W = [x.copy() for x in w['W']]; b = [x.copy() for x in w['b']]
        W[2] = W[2] * gain; W[3] = W[3] * gain; W[-1] = W[-1] / (gain * gain)            # hidden activations x gain^2, logits of the usual size
        f32 = ops.PackedMLP(ops.NET_SAMPLER, W, b, variant='sampler_f32')
        ref = ops.sampler_fwd(f32, rays, want_raw=True)
        for variant in ('default', 'sampler_split'):
            mlp = ops.PackedMLP(ops.NET_SAMPLER, W, b, variant=variant)
            out = ops.sampler_fwd(mlp, rays, two_pass=True, want_raw=True)
            n2, n3 = int(out[6]), int(out[7])
            for t in out[:6]:
                assert t is None or bool(torch.isfinite(t.float()).all())
            assert (n3 > 0) == expect_sat, (gain, variant, n2, n3)
            tie = (ref[0][:, 1:] - ref[0][:, :-1]).min(1)[0] <= 1e-6
            assert bool((out[1][~tie] == ref[1][~tie]).all()), (gain, variant)
            if expect_sat:
                # the split kernel alone (no workspace: saturating arithmetic, no third pass) is finite but differs from fp32 on the saturated rays ...
                lone = ops.sampler_fwd(ops.PackedMLP(ops.NET_SAMPLER, W, b, variant='sampler_split'), rays, want_raw=True)
                assert bool(torch.isfinite(lone[0]).all())
                moved = (lone[5] - ref[5]).abs().max(1)[0] > 1e-4
                assert int(moved.sum()) > 0
                # ... and exactly those rows carry the exact-fp32 kernel's values after the third pass
                assert n3 >= int(moved.sum())
                assert torch.equal(out[5][moved], ref[5][moved]) and torch.equal(out[2][moved], ref[2][moved]) and torch.equal(out[3][moved], ref[3][moved])
            else:
                assert n3 == 0 and float((out[5] - ref[5]).abs().max()) <= 2e-3


def test_explore_with_a_non_finite_depth_is_deterministic(dev):
    """pnrf_explore_fwd's rank sort with NaN depths (a diverged training step): NaNs rank behind every number, each output slot is written
    exactly once — the same NaN / finite pattern on every run (the old rank left slots unwritten: uninitialised LDS), finite rays untouched."""
    from pronerf_amd import ops
    scene = synth.make_scene(0, H=9, W=11, rotate=True)
    rays = cu(orc.frame_setup(scene)['rays'], dev)
    n = rays.shape[0]
    g = torch.Generator().manual_seed(0)
    z8 = torch.sort(torch.rand(n, 8, generator=g), 1)[0]
    clean = z8.clone()
    z8[3, 2] = float('nan'); z8[5, :] = float('nan'); z8[7, 7] = float('inf')
    for n_mult in (4, 8):
        jit = (torch.rand(n, 8 * n_mult, generator=g) * 0.2)
        runs = [ops.explore(cu(z8, dev), rays, cu(jit, dev), n_mult, 1, -1) for _ in range(3)]
        ref = ops.explore(cu(clean, dev), rays, cu(jit, dev), n_mult, 1, -1)
        for zo, po in runs[1:]:
            assert torch.equal(torch.isnan(zo), torch.isnan(runs[0][0])) and torch.equal(torch.nan_to_num(zo), torch.nan_to_num(runs[0][0]))
            assert torch.equal(torch.nan_to_num(po), torch.nan_to_num(runs[0][1]))
        zo = runs[0][0]
        ok = torch.ones(n, dtype=torch.bool); ok[[3, 5, 7]] = False
        assert torch.equal(zo[ok.to(dev)], ref[0][ok.to(dev)])                    # rays without a non-finite depth: unchanged
        assert bool(torch.isnan(zo[5]).all()) and int(torch.isnan(zo[3]).sum()) >= n_mult
