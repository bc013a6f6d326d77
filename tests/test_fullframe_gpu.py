"""GPU: parity evidence at the size of the BASELINE workload (one 756 x 1008 Fern-geometry frame = 762 048 rays).

* sampler sort indices of EVERY ray of the frame against the CPU oracle's sampler (the oracle's fp32 torch graph, pinned to the
  reference by tests/test_oracle_golden.py), for several weight sets and for both sampler kernels (split fp16, exact fp32);
  the tie set — rays whose sorted oracle depths have a gap <= 1e-6, where fp32 summation order decides — is printed and bounded,
  and inside it the permutation may differ only between depths that close (run_S_eS_eN_alter_trt.py:631-635);
* 8 829 rays of the frame rendered by the REFERENCE itself (borders with out-of-image taps + stratified interior):
  indices, depths, rgb, depth map;
* rgb / depth of the whole frame against the oracle's eager fp32 graph run on the device, three weight seeds.

Tolerances: BASELINE.json north_star / BASELINE.md §4 — indices identical; RGB error PSNR >= 46.4 dB (moves a 27 dB image PSNR by
<= 0.05 dB); depth map: `depth_bars` below — the same 46.4 dB error-PSNR gate as rgb (NDC depths live in [0, 1]) and bars in grey levels of the
reference's own 8-bit depth PNG (99 % within half a level, 99.9 % within one, none beyond 2.5) — plus 2e-3 relative RMS; sampler depths within 2e-6.
"""
import os

import numpy as np
import pytest
import torch

from oracle import pronerf_oracle as orc
from oracle import synth

pytestmark = pytest.mark.gpu

H, W, FOCAL = 756, 1008, 815.13
N = H * W
TIE = 1e-6
# 'heavy' / 'x4': adversarial sets for the two-pass sampler (Student-t(3) hidden weights; hidden layers x4 each = activations up to the fp16 range);
# 'optimizer': the nets this package's own trainers produced on the synthetic LLFF pictures (tests/golden/trained_synth_scene.npz);
# 'scene' (round 6): the nets trained on the geometrically consistent 3-D scene (tests/golden/trained_scene3d.npz), on THAT scene's hold-out pose at the
# Fern frame size — a sampler that has learned surfaces: the 8 depths of a ray bunch there, 60 .. 80 % of the rays go through the second pass
WEIGHT_SETS = [(0, 'trained'), (2, 'spread'), (3, 'trained'), (1, 'default'), (0, 'heavy'), (0, 'x4'), (0, 'optimizer'), (0, 'scene')]
NEW_KINDS = ('heavy', 'x4', 'optimizer', 'scene')


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    from pronerf_amd import _lib
    _lib.load()
    return torch.device('cuda:0')


_ORACLE = {}


GREY = 1.0 / 255.0       # one level of the reference's own depth output: to8b(depth / max(depth)), an 8-bit PNG (run_S_eS_eN_alter_trt.py:360-361)


def edge_allowance(ref_map, k=5):
    """[H, W] local range (max - min over k x k pixels) of a depth map.  On a scene with occlusion edges the composited depth of an edge pixel is a mixture of
    the near and the far surface whose weight a rounding error can move: there — and only there — the depth MAP is discontinuous and a pixel's error is bounded
    by the depth range of its neighbourhood, not by grey levels.  ``depth_bars(.., allow=edge_allowance(ref))`` subtracts it (seeded nets have no surfaces: 0)."""
    import torch.nn.functional as F
    d = ref_map[None, None].float()
    rng = (F.max_pool2d(d, k, 1, k // 2) + F.max_pool2d(-d, k, 1, k // 2))[0, 0]
    return torch.where(rng > 8 * GREY, rng, torch.zeros_like(rng))          # a jump of more than 8 grey levels inside 5 x 5 pixels is an edge; smooth slopes get nothing


def depth_bars(got, ref, tag='', allow=None):
    """The depth channel's tolerance, derived instead of fitted.  BASELINE.json asks for RGB / depth "within a stated fp tolerance
    (PSNR-equivalent)": NDC depths live in [0, 1] like colours, so (i) the depth map's error PSNR (peak 1) must clear the SAME 46.4 dB gate as rgb; and the
    reference's depth product is an 8-bit image, so (ii) 99 % of the pixels are within half a grey level (1/510: they round to the same or the adjacent
    level), (iii) 99.9 % within one level (1/255) and (iv) no pixel is off by more than 2.5 levels (the old 1e-2 absolute bar, now with a unit).
    Returns the measured figures for the log."""
    e = (got.double() - ref.double()).abs().flatten()
    if allow is not None:          # the part of the error that the local depth range does not explain (occlusion edges: see edge_allowance)
        raw99, rawmax = float(torch.quantile(e[::max(1, e.numel() // 1000000)], 0.999)), float(e.max())
        e = (e - allow.double().flatten()).clamp_min(0)
        print(f'[depth bars{tag}] before the edge allowance: 99.9 % <= {raw99 / GREY:.3f} grey levels, max {rawmax / GREY:.2f}; pixels whose error it covers in part: {int((allow.flatten() > GREY).sum())}')
    mse = float((e ** 2).mean())
    psnr = float('inf') if mse == 0 else -10.0 * float(np.log10(mse))
    q99, q999 = (float(torch.quantile(e[::max(1, e.numel() // 1000000)], q)) for q in (0.99, 0.999))
    mx = float(e.max())
    print(f'[depth bars{tag}] error PSNR {psnr:.1f} dB (gate 46.4), 99 % <= {q99 / GREY:.3f} grey levels (bar 0.5), 99.9 % <= {q999 / GREY:.3f} (bar 1), max {mx / GREY:.2f} (bar 2.5)')
    assert psnr >= 46.4 and q99 <= 0.5 * GREY and q999 <= GREY and mx <= 2.5 * GREY, (psnr, q99, q999, mx)
    return psnr, q99, q999, mx


def depth_relrms(a, b):
    """RMS error of a depth map relative to the RMS of the depths (NDC depths in [0, 1])."""
    a = a.double(); b = b.double()
    return float(((a - b) ** 2).mean().sqrt() / (b ** 2).mean().sqrt())



def oracle_sampler_full_frame(seed, kind):
    """The oracle's frame set-up and sampler for all 762 048 rays on the host (0.3 TFLOP of fp32 torch CPU GEMMs), in chunks of 65 536 rays.
    Cached per weight set: both kernel variants are compared with the same oracle run."""
    key = (seed, kind)
    if key not in _ORACLE:
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        scene = synth.scene_for(seed, kind, H=H, W=W, focal=FOCAL, rotate=True)
        assert (scene['H'], scene['W']) == (H, W)
        w = synth.weight_set(seed, kind)
        rays_o, rays_d = orc.get_rays(H, W, scene['K'], scene['c2w'])
        vd = (rays_d / rays_d.norm(dim=-1, keepdim=True)).reshape(-1, 3)
        o, d = orc.ndc_rays(H, W, float(scene['K'][0, 0]), 1.0, rays_o, rays_d)
        o, d = o.reshape(-1, 3), d.reshape(-1, 3)
        rays = torch.cat([o, d, torch.zeros(N, 1), torch.ones(N, 1), vd], -1).contiguous()
        ds, idx, raw = [], [], []
        with torch.no_grad():
            for a in range(0, N, 65536):
                b = min(N, a + 65536)
                mm = orc.mm_input_from_rays(o[a:b], d[a:b])
                _, add, mul, depth = orc.sampler_forward(w['sampler'], mm)
                s, i, _, _ = orc.sort_gather(depth, add, mul, rays[a:b, 6:7], rays[a:b, 7:8])
                ds.append(s); idx.append(i); raw.append(depth)
        _ORACLE.clear()                      # one frame's worth at a time (64 MB per entry)
        _ORACLE[key] = dict(scene=scene, w=w, rays=rays, depth_sorted=torch.cat(ds), sort_idx=torch.cat(idx), depth_raw=torch.cat(raw))
    return _ORACLE[key]


@pytest.mark.parametrize('variant', ['default', 'sampler_split', 'sampler_f32'])
@pytest.mark.parametrize('seed,kind', WEIGHT_SETS)
def test_full_frame_sampler_indices(dev, seed, kind, variant):
    """variant 'default' = the two-pass sampler of the fused path (pnrf_sampler_fwd_ws: plain fp16 for every ray, split fp16 for the rays
    pass 1 cannot decide); 'sampler_split' / 'sampler_f32' = one fp32-grade kernel for every ray."""
    from pronerf_amd import ops
    from pronerf_amd.render import Renderer
    oc = oracle_sampler_full_frame(seed, kind)
    scene, w = oc['scene'], oc['w']
    two_pass = variant == 'default'
    mlp = ops.PackedMLP(ops.NET_SAMPLER, w['sampler']['W'], w['sampler']['b'], variant=variant)
    # (1) the sampler on the oracle's own rays: identical inputs on both sides
    rays = oc['rays'].to(dev)
    out = ops.sampler_fwd(mlp, rays, want_idx=True, want_rgb=False, want_raw=True, two_pass=two_pass)
    g_ds, g_idx, g_add, g_mul, _, g_raw = out[:6]
    g_ds, g_idx, g_raw = g_ds.cpu(), g_idx.cpu(), g_raw.cpu()
    ds, idx = oc['depth_sorted'], oc['sort_idx']
    gap = (ds[:, 1:] - ds[:, :-1]).min(dim=1)[0]
    tie = gap <= TIE
    n_tie = int(tie.sum())
    derr_row = (g_raw - oc['depth_raw']).abs().max(dim=1)[0]
    derr = float(derr_row.max())
    mism_free = int((g_idx[~tie] != idx[~tie]).any(1).sum())
    mism_tie = int((g_idx[tie] != idx[tie]).any(1).sum())
    print(f'\n[full frame] weights ({seed}, {kind}), sampler variant {variant}: {N} rays, tie set (sorted-depth gap <= {TIE:g}) {n_tie} rays '
          f'({n_tie / N:.2e}); rays with different indices: {mism_free} outside the tie set, {mism_tie} inside; max |depth - oracle| {derr:.2e}')
    assert mism_free == 0, 'sampler sort indices differ from the oracle outside the tie set'
    if two_pass:
        # the rows pass 2 re-rendered are bit for bit those of the split kernel; the others carry plain-fp16 products (fp16-grade depths)
        n2 = int(out[6])
        s_ds, s_idx, s_add, s_mul, _, s_raw = ops.sampler_fwd(mlp, rays, want_idx=True, want_rgb=False, want_raw=True)[:6]     # no workspace: split kernel
        same = ((s_raw.cpu() == g_raw).all(1) & (s_add == g_add).all(1).cpu() & (s_mul == g_mul).all(1).cpu())
        coarse = ~same
        print(f'[full frame] two passes: {n2} rays ({n2 / N:.2%}) rendered by the split-fp16 pass; {int(same.sum())} rows identical to the split kernel\'s; '
              f'max |depth - oracle| on the other rows {float(derr_row[coarse].max()) if bool(coarse.any()) else 0.0:.2e}, on the identical rows '
              f'{float(derr_row[same].max()) if bool(same.any()) else 0.0:.2e}')
        assert 0 < n2 <= int(same.sum()) <= n2 + 64 or kind == 'spread'          # (rows of pass 1 that coincide with the split result in all 24 values: a handful at most)
        # the second pass stays a minority — except on the heavy-tailed set, whose largest column norms make the error bound flag every ray
        # (safe: everything is then rendered fp32-grade, at the split kernel's speed)
        assert n2 <= (1.0 if kind == 'heavy' else 0.9 if kind == 'scene' else 0.6 if kind in NEW_KINDS else 0.35) * N
        assert int(out[7]) == 0 or kind == 'x4'                                  # the exact-fp32 third pass has nothing to do unless activations reach the fp16 limit
        assert float(derr_row[same].max()) <= 2e-6 if bool(same.any()) else True
        assert float(derr_row[coarse].max()) <= 2e-3 if bool(coarse.any()) else True
        assert bool((gap[tie] <= TIE).all()) and bool(same[tie].all())          # every tie-set ray went through the fp32-grade pass
        # the check has teeth: with kappa = 0 (pass 1 decides everything but fp32 round-off ties) indices DO differ on these weights
        o0 = ops.sampler_fwd(mlp, rays, want_idx=True, want_rgb=False, want_raw=False, two_pass=True, kappa=0.0)
        mism0 = int((o0[1].cpu()[~tie] != idx[~tie]).any(1).sum())
        print(f'[full frame] kappa = 0 (no safety margin): {mism0} rays outside the tie set with different indices, second pass {int(o0[6])} rays')
        if kind == 'trained':
            assert mism0 > 0
    else:
        assert derr <= 2e-6
    # the tie set is small for weights with spread depths; with default-initialised weights all 8 depths of a ray sit within ~1e-2
    assert n_tie <= (2e-2 if kind == 'default' else 2e-3 if kind in NEW_KINDS else 2e-4) * N, n_tie
    # inside the tie set: still a sorting permutation of the kernel's own depths, and equal to the oracle's up to transpositions of depths
    # closer than the tie threshold (so the sorted depth vectors agree)
    assert bool((g_ds[:, 1:] >= g_ds[:, :-1]).all())
    assert bool((torch.sort(g_idx, dim=1)[0] == torch.arange(8)[None]).all())
    np.testing.assert_array_equal(torch.gather(g_raw, 1, g_idx).numpy(), g_ds.numpy())          # near = 0, far = 1: the affine map is exact
    np.testing.assert_allclose(g_ds.numpy(), ds.numpy(), rtol=0, atol=2e-3 if two_pass else 2e-6)
    if n_tie:
        od = torch.gather(oc['depth_raw'][tie], 1, g_idx[tie])       # oracle depths in the kernel's order: ascending up to the threshold
        assert float((od[:, :-1] - od[:, 1:]).max()) <= TIE
    if variant != 'default':
        return
    # (2) end to end from (K, c2w): rays generated on the device, whole path in one call; same indices wherever the depths are apart
    rend = Renderer({k: w[k] for k in ('sampler', 'refine', 'nerf')}, max_rays=N, device=dev)
    rend.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
    r2, o2 = rend.frame_rays(scene['K'], scene['c2w'], H, W)
    np.testing.assert_array_equal(r2.cpu().numpy(), oc['rays'].numpy())        # the device's rays ARE the oracle's: 0 ulp on all 762 048 x 11 values
    _, idx2 = rend.render_rays(r2, o2, want_idx=True)
    mism2 = int((idx2.cpu()[~tie] != idx[~tie]).any(1).sum())
    print(f'[full frame] end to end (device-generated rays, fused path): {mism2} rays outside the tie set with different indices')
    assert mism2 == 0                                                           # same 1e-6 tie rule as the operator-level check above


def test_fern_8k_rays_vs_the_reference(dev, golden_dir):
    """rgb / depth / indices of 8 829 rays of the full frame against the REFERENCE's own render (oracle/gen_golden.py --fern-8k)."""
    from pronerf_amd.render import Renderer
    g = dict(np.load(os.path.join(golden_dir, 'infer_trained_fern_756x1008_8k.npz')))
    seed = int(g['seed'])
    scene = synth.make_scene(seed, H=H, W=W, rotate=True)
    rend = Renderer(synth.make_weights(seed, 'trained'), max_rays=N, device=dev)
    ref_nos = rend.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
    np.testing.assert_array_equal(ref_nos, g['ref_nos'])
    rays, or_rays = rend.frame_rays(scene['K'], scene['c2w'], H, W)
    sel = torch.from_numpy(g['sel']).to(dev)
    assert len(g['sel']) >= 8192 and int((g['oob_taps'] > 0).sum()) > 1000
    np.testing.assert_array_equal(rays[sel].cpu().numpy(), g['rays'])
    full, idx = rend.render_rays(rays, or_rays, want_idx=True)
    got, gi = full[sel].cpu(), idx[sel].cpu().numpy()
    tie_free = np.diff(g['depth_sorted'], axis=1).min(axis=1) > TIE
    print(f'\n[fern 8k] {len(sel)} reference rays, {int((g["oob_taps"] > 0).sum())} with out-of-image taps, tie set {int((~tie_free).sum())}')
    assert (~tie_free).sum() <= 4
    np.testing.assert_array_equal(gi[tie_free], g['sort_idx'][tie_free].astype(np.int64))
    m = torch.from_numpy(tie_free)
    ps = orc.psnr(got[m, :3], torch.from_numpy(g['rgb'])[m])
    rel = float(((got[m, :3].double() - torch.from_numpy(g['rgb'])[m].double()) ** 2).mean().sqrt() / (torch.from_numpy(g['rgb'])[m].double() ** 2).mean().sqrt())
    print(f'[fern 8k] rgb PSNR vs the reference {ps:.1f} dB, rel. RMS {rel:.2e}, max depth error {float((got[m, 3] - torch.from_numpy(g["depth"])[m]).abs().max()):.2e}')
    assert ps > 46.4 and rel < 1e-2
    depth_bars(got[m, 3], torch.from_numpy(g['depth'])[m], ' vs the reference\'s own rays')
    assert depth_relrms(got[m, 3], torch.from_numpy(g['depth'])[m]) < 2e-3
    # the border rays alone (their epi features are partly zero-padded): same bar
    border = torch.from_numpy((g['oob_taps'] > 0) & tie_free)
    assert orc.psnr(got[border, :3], torch.from_numpy(g['rgb'])[border]) > 46.4


@pytest.mark.parametrize('fixture', ['pictures', 'scene3d'])
def test_full_frame_with_optimizer_trained_nets(dev, fixture):
    """The whole frame with the nets an optimizer produced (sampler, refine and the NeRF-CLASS fine net the trainers save) against the oracle's eager
    fp32 graph on the device: indices, rgb, depth — the acceptance the reference applies to a trained Fern checkpoint
    (run_S_eS_eN_alter_trt.py:351-373), on the weights that exist here: 'pictures' (tests/golden/trained_synth_scene.npz, on a seeded frame) and
    'scene3d' (tests/golden/trained_scene3d.npz on ITS scene: hold-out pose 0 at the Fern frame size, the 17 training pictures as neighbour pool;
    the frame is also held against the ray-cast ground truth there)."""
    from pronerf_amd.render import Renderer
    torch.backends.cuda.matmul.allow_tf32 = False
    scene = synth.scene_for(0, 'scene' if fixture == 'scene3d' else 'optimizer', H=H, W=W, focal=FOCAL, rotate=True)
    w = synth.load_trained_fixture(fixture)
    rend = Renderer({k: w[k] for k in ('sampler', 'refine', 'nerf')}, max_rays=N, device=dev)
    rend.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
    fr = orc.frame_setup(scene)
    rays, or_rays = fr['rays'].to(dev), fr['or_rays'].to(dev)
    rgbd, idx = rend.render_rays(rays, or_rays, want_idx=True)
    td = lambda x: torch.as_tensor(x).to(dev)
    wd = {k: {'W': [td(x) for x in w[k]['W']], 'b': [td(x) for x in w[k]['b']]} for k in ('sampler', 'refine')}
    c = w['nerfcls']
    pair = lambda p: (td(p[0]), td(p[1]))
    wd['nerfcls'] = {'pts_linears': [pair(p) for p in c['pts_linears']], 'feature_linear': pair(c['feature_linear']), 'alpha_linear': pair(c['alpha_linear']),
                     'views_linears': [pair(c['views_linears'][0])], 'rgb_linear': pair(c['rgb_linear'])}
    with torch.no_grad():
        ref = orc.render_rays_infer(wd, rays, or_rays, fr['images'].to(dev), fr['proj'].to(dev), mm_input=fr['mm_input'].to(dev), nerf='nerfcls')
    gap = (ref['depth_sorted'][:, 1:] - ref['depth_sorted'][:, :-1]).min(dim=1)[0]
    free = gap > 4e-6
    mism = int((idx[free] != ref['sort_idx'][free]).any(1).sum())
    ps = orc.psnr(rgbd[free, :3], ref['rgb'][free])
    rel = float(((rgbd[free, :3].double() - ref['rgb'][free].double()) ** 2).mean().sqrt() / (ref['rgb'][free].double() ** 2).mean().sqrt())
    derr = float((rgbd[free, 3] - ref['depth'][free]).abs().max())
    n2, n3 = rend.ctx.sampler_stats(), rend.ctx.sampler_saturated()
    if fixture == 'scene3d':              # pixel (4j, 4i) of the 756 x 1008 frame IS the ray of pixel (j, i) of the 189 x 252 ground-truth picture of the same pose
        small = rgbd[:, :3].reshape(H, W, 3)[::4, ::4]
        ps_gt = orc.psnr(small.reshape(-1, 3), torch.as_tensor(scene['gt_small']).reshape(-1, 3).to(dev))
        print(f'\n[full frame, scene3d nets] hold-out view 0 at 756 x 1008, every 4th pixel: PSNR vs the ray-cast ground truth {ps_gt:.2f} dB')
        assert ps_gt > 33.0               # 37.28 dB in tests/test_quality_gate_gpu.py (the same rays rendered at 189 x 252)
    print(f'\n[full frame, optimizer-trained nets: {fixture}] {int(free.sum())} of {N} rays compared, index mismatches {mism}, rgb PSNR {ps:.1f} dB, rel. RMS {rel:.2e}, '
          f'max depth error {derr:.2e}, depth rel. RMS {depth_relrms(rgbd[free, 3], ref["depth"][free]):.2e}; second pass {n2 / N:.2%}, third pass {n3} rays')
    assert int((~free).sum()) <= (2e-2 if fixture == 'scene3d' else 2e-3) * N and mism == 0          # bunched depths: more fp32 ties
    assert ps > 46.4 and rel < 2e-2
    if fixture == 'scene3d':          # a scene with occlusion edges: the bars hold for what the local depth range does not explain (edge_allowance)
        allow = edge_allowance(ref['depth'].reshape(H, W)).reshape(-1)
        depth_bars(rgbd[free, 3], ref['depth'][free], f' optimizer-trained nets: {fixture}', allow=allow[free])
    else:
        depth_bars(rgbd[free, 3], ref['depth'][free], f' optimizer-trained nets: {fixture}')
    assert bool(torch.isfinite(rgbd).all())


@pytest.mark.parametrize('seed', [0, 3, 5])
def test_full_frame_rgb_vs_eager_oracle_on_device(dev, seed):
    """All 762 048 rays: the fused HIP path against the oracle's eager fp32 torch graph executed on the same device (same graph as the CPU
    oracle, rocBLAS fp32 GEMMs instead of CPU ones) — rgb, depth and indices, three "trained" weight seeds."""
    from pronerf_amd.render import Renderer
    torch.backends.cuda.matmul.allow_tf32 = False
    scene = synth.make_scene(seed, H=H, W=W, focal=FOCAL, rotate=True)
    w = synth.make_weights(seed, 'trained')
    rend = Renderer(w, max_rays=N, device=dev)
    rend.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
    fr = orc.frame_setup(scene)
    rays, or_rays = fr['rays'].to(dev), fr['or_rays'].to(dev)
    rgbd, idx = rend.render_rays(rays, or_rays, want_idx=True)
    wd = {k: {'W': [torch.as_tensor(x).to(dev) for x in v['W']], 'b': [torch.as_tensor(x).to(dev) for x in v['b']]} for k, v in w.items()}
    with torch.no_grad():
        ref = orc.render_rays_infer(wd, rays, or_rays, fr['images'].to(dev), fr['proj'].to(dev), mm_input=fr['mm_input'].to(dev))
    gap = (ref['depth_sorted'][:, 1:] - ref['depth_sorted'][:, :-1]).min(dim=1)[0]
    free = gap > 4e-6                      # two fp32 GEMM chains with different summation orders on the two sides
    mism = int((idx[free] != ref['sort_idx'][free]).any(1).sum())
    ps = orc.psnr(rgbd[free, :3], ref['rgb'][free])
    rel = float(((rgbd[free, :3].double() - ref['rgb'][free].double()) ** 2).mean().sqrt() / (ref['rgb'][free].double() ** 2).mean().sqrt())
    derr = float((rgbd[free, 3] - ref['depth'][free]).abs().max())
    print(f'\n[full frame rgb] seed {seed}: {int(free.sum())} of {N} rays compared, index mismatches {mism}, rgb PSNR {ps:.1f} dB, rel. RMS {rel:.2e}, '
          f'max depth error {derr:.2e}')
    assert int((~free).sum()) <= 1e-3 * N
    assert mism == 0
    assert ps > 46.4 and rel < 1e-2 and depth_relrms(rgbd[free, 3], ref['depth'][free]) < 2e-3
    depth_bars(rgbd[free, 3], ref['depth'][free], f' seed {seed}')
    assert bool(torch.isfinite(rgbd).all())


def test_full_frame_rgb_vs_the_cpu_oracle(dev):
    """65 536 rays strided over the whole 756 x 1008 frame, rendered by the fused path inside the full-frame call, against the CPU oracle itself (fp32
    torch on the host — the graph the goldens pin to the reference — not its copy on the device): indices, rgb, depth map."""
    from pronerf_amd.render import Renderer
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    seed = 0
    scene = synth.make_scene(seed, H=H, W=W, focal=FOCAL, rotate=True)
    w = synth.make_weights(seed, 'trained')
    rend = Renderer(w, max_rays=N, device=dev)
    rend.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
    rays, or_rays = rend.frame_rays(scene['K'], scene['c2w'], H, W)
    rgbd, idx = rend.render_rays(rays, or_rays, want_idx=True)
    fr = orc.frame_setup(scene)
    sel = torch.linspace(0, N - 1, 65536).long()
    assert torch.equal(rays.cpu()[sel], fr['rays'][sel])                           # same rays on both sides, bit for bit
    with torch.no_grad():
        ref = orc.render_rays_infer(w, fr['rays'][sel].contiguous(), fr['or_rays'][sel].contiguous(), fr['images'], fr['proj'])
    free = (ref['depth_sorted'][:, 1:] - ref['depth_sorted'][:, :-1]).min(1)[0] > TIE
    got = rgbd.cpu()[sel]
    mism = int((idx.cpu()[sel][free] != ref['sort_idx'][free]).any(1).sum())
    ps = orc.psnr(got[free, :3], ref['rgb'][free])
    rel = float(((got[free, :3].double() - ref['rgb'][free].double()) ** 2).mean().sqrt() / (ref['rgb'][free].double() ** 2).mean().sqrt())
    derr = float((got[free, 3] - ref['depth'][free]).abs().max())
    print(f'\n[full frame vs CPU oracle] {int(free.sum())} of 65536 rays outside the tie set: index mismatches {mism}, rgb PSNR {ps:.1f} dB, rel. RMS {rel:.2e}, '
          f'max depth error {derr:.2e}')
    assert int((~free).sum()) <= 16 and mism == 0
    assert ps > 46.4 and rel < 1e-2 and depth_relrms(got[free, 3], ref['depth'][free]) < 2e-3
    depth_bars(got[free, 3], ref['depth'][free], ' vs the CPU oracle')


def test_two_pass_sampler_on_a_population_of_weight_sets(dev):
    """The statistical guarantee of the default sampler (include/pronerf_hip.h: sort indices equal the split-fp16 kernel's) sampled wider than the eight
    sets above, inside the suite: 4 scenes x 3 kinds of seeded weights + the scene-trained nets on three poses of their scene, every ray of the 756 x 1008
    frame, at the SHIPPED kappa and at half of it; rays whose split-kernel sorted depths are closer than 2e-6 are ties.  (tools/kappa_population.py
    is the long form: 20 scenes x 60 sets, 45.7 M rays.)  Also holds pnrf_ctx_get_sampler_kappa to the header."""
    from pronerf_amd import ops
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include', 'pronerf_hip.h')).read()
    import re
    kappa = float(re.search(r'#define PNRF_SAMPLER_KAPPA ([0-9.]+)f', hdr).group(1))
    cases = [(seed, kind, synth.make_scene(seed, H=H, W=W, focal=FOCAL, rotate=True)) for seed in (100, 101, 102, 103) for kind in ('trained', 'spread', 'default')]
    cases += [(v, 'scene', synth.scene3d_frame(v, 4)) for v in (0, 8, 16)]
    tot = {k: [0, 0, 0, 0] for k in (kappa, kappa / 2)}        # rays, ties, differ, second pass
    for seed, kind, scene in cases:
        w = synth.weight_set(seed, kind)
        rays, _ = ops.frame_rays(scene['K'], scene['c2w'], H, W, near=0., far=1., device=dev)
        mlp = ops.PackedMLP(ops.NET_SAMPLER, w['sampler']['W'], w['sampler']['b'])
        s_ds, s_idx = ops.sampler_fwd(mlp, rays, want_idx=True, want_rgb=False)[:2]
        tie = (s_ds[:, 1:] - s_ds[:, :-1]).min(1)[0] <= 2e-6
        for k in tot:
            o = ops.sampler_fwd(mlp, rays, want_idx=True, want_rgb=False, two_pass=True, kappa=k)
            d = int(((o[1] != s_idx).any(1) & ~tie).sum())
            t = tot[k]
            t[0] += N; t[1] += int(tie.sum()); t[2] += d; t[3] += int(o[6])
            assert d == 0, (seed, kind, k, d)
        del mlp
    for k, t in tot.items():
        print(f'\n[kappa population] kappa {k:g}: {t[0]} rays of {len(cases)} weight set x frame pairs, {t[1]} ties, {t[2]} differ from the split kernel, second pass {t[3] / t[0]:.1%}')
    w = synth.make_weights(0, 'trained')
    from pronerf_amd.render import Renderer
    assert Renderer(w, max_rays=64, device=dev).ctx.sampler_kappa() == kappa
