"""GPU: the whole inference path (pnrf_render_rays_fwd through Renderer) against the oracle and
the committed golden fixtures.

Tolerance (BASELINE.json north_star / BASELINE.md §4): sampler sort indices identical; RGB
PSNR >= 46.4 dB vs the fp32 oracle (moves a 27 dB image PSNR by <= 0.05 dB); depth within 2e-2.
"""
import os

import numpy as np
import pytest
import torch

from oracle import pronerf_oracle as orc
from oracle import synth

pytestmark = pytest.mark.gpu

CASES = ['infer_trained_24x32', 'infer_spread_20x28_img48x64', 'infer_trained_oob_16x24', 'infer_default_24x32']


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    from pronerf_amd import _lib
    _lib.load()
    return torch.device('cuda:0')


def relrms(a, b):
    a = a.double(); b = b.double()
    return float(((a - b) ** 2).mean().sqrt() / (b ** 2).mean().sqrt().clamp_min(1e-30))


@pytest.mark.parametrize('name', CASES)
def test_render_rays_vs_golden(dev, golden_dir, name):
    """Same seeded inputs as the reference-generated fixture; compare with the REFERENCE's outputs."""
    from pronerf_amd.render import Renderer
    g = dict(np.load(os.path.join(golden_dir, name + '.npz')))
    seed, kind = int(g['seed']), str(g['kind'])
    Hh, Ww = int(g['H']), int(g['W'])
    scene = synth.make_scene(seed, H=Hh, W=Ww, Hf=int(g['Hf']), Wf=int(g['Wf']), rotate=bool(g['rotate']), sigma_t=float(g['sigma_t']))
    rend = Renderer(synth.make_weights(seed, kind), max_rays=Hh * Ww, device=dev)
    ref_nos = rend.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
    np.testing.assert_array_equal(ref_nos, g['ref_nos'])
    np.testing.assert_allclose(rend.proj.cpu().numpy(), g['proj'], rtol=1e-6, atol=1e-5)
    rays, or_rays = rend.frame_rays(scene['K'], scene['c2w'], Hh, Ww)
    np.testing.assert_array_equal(rays.cpu().numpy(), g['rays'])                               # the reference's own rays, bit for bit
    rgbd, idx = rend.render_rays(rays, or_rays, want_idx=True)
    rgbd = rgbd.cpu(); idx = idx.cpu().numpy()
    tie_free = np.diff(g['depth_sorted'], axis=1).min(axis=1) > 1e-6
    np.testing.assert_array_equal(idx[tie_free], g['sort_idx'][tie_free])
    if kind != 'default':
        assert tie_free.all()
    m = torch.from_numpy(tie_free)
    ps = orc.psnr(rgbd[m, :3], torch.from_numpy(g['rgb'])[m])
    assert ps > 46.4, ps
    if kind == 'trained':          # signal-carrying outputs: also bound the error relative to the signal
        assert relrms(rgbd[m, :3], torch.from_numpy(g['rgb'])[m]) < 1e-2
        np.testing.assert_allclose(rgbd[m, 3].numpy(), g['depth'][tie_free], rtol=0, atol=2e-2)


def test_render_rays_fern_geometry_subset(dev, golden_dir):
    """Full 756x1008 frame geometry: render the 512 fixture rays of the reference's frame."""
    from pronerf_amd.render import Renderer
    g = dict(np.load(os.path.join(golden_dir, 'infer_trained_fern_756x1008.npz')))
    seed = int(g['seed'])
    scene = synth.make_scene(seed, H=756, W=1008, rotate=True)
    rend = Renderer(synth.make_weights(seed, 'trained'), max_rays=756 * 1008, device=dev)
    rend.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
    rays, or_rays = rend.frame_rays(scene['K'], scene['c2w'], 756, 1008)
    sel = torch.from_numpy(g['sel']).to(dev)
    np.testing.assert_array_equal(rays[sel].cpu().numpy(), g['rays'])
    rgbd, idx = rend.render_rays(rays[sel].contiguous(), or_rays[sel].contiguous(), want_idx=True)
    np.testing.assert_array_equal(idx.cpu().numpy(), g['sort_idx'])
    assert orc.psnr(rgbd[:, :3].cpu(), torch.from_numpy(g['rgb'])) > 46.4
    # full-size properties that need no oracle: the full frame equals the concatenation of ray shards
    # (ray independence -> the 8-GPU sharding is exact), and every output is finite and in range
    full, _ = rend.render_rays(rays, or_rays)
    from pronerf_amd.render import shard_range
    parts = []
    for r in range(8):
        f, c = shard_range(756 * 1008, r, 8)
        parts.append(rend.render_rays(rays[f:f + c].contiguous(), or_rays[f:f + c].contiguous())[0].clone())
    assert torch.equal(torch.cat(parts, 0), full)
    assert bool(torch.isfinite(full).all())
    assert float(full[:, :3].min()) >= 0.0 and float(full[:, :3].max()) <= 3.0
    np.testing.assert_allclose(full[sel, :3].cpu().numpy(), rgbd[:, :3].cpu().numpy(), rtol=0, atol=0)


def test_empty_and_ragged(dev):
    from pronerf_amd.render import Renderer
    scene = synth.make_scene(0, H=8, W=9)
    w = synth.make_weights(0, 'trained')
    rend = Renderer(w, max_rays=300, device=dev)
    rend.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
    rays, or_rays = rend.frame_rays(scene['K'], scene['c2w'], 8, 9)
    full, _ = rend.render_rays(rays, or_rays)
    full = full.clone()
    assert rend.render_rays(rays[:0], or_rays[:0])[0].shape == (0, 4)
    for n in (1, 31, 33, 71):
        part, _ = rend.render_rays(rays[:n].contiguous(), or_rays[:n].contiguous())
        assert torch.equal(part, full[:n])
    from pronerf_amd import _lib
    with pytest.raises(_lib.PnrfError):
        big = torch.zeros(301, 11, device=dev)
        rend.render_rays(big, big)


def test_stage_timing_of_the_fused_path(dev):
    """pnrf_ctx_profile_begin / _end: events around the three kernels of the next calls; results untouched, bounded by the wall clock."""
    import time
    from pronerf_amd.ops import PnrfError
    from pronerf_amd.render import Renderer
    scene = synth.make_scene(0, H=96, W=128, rotate=True)
    rend = Renderer(synth.make_weights(0, 'trained'), max_rays=96 * 128, device=dev)
    rend.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
    rays, orr = rend.frame_rays(scene['K'], scene['c2w'], 96, 128)
    ref = rend.render_rays(rays, orr)[0].clone()
    with pytest.raises(PnrfError, match='profile_begin'):
        rend.ctx.profile_end()
    rend.ctx.profile_begin(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):                                   # only the first 3 calls are recorded
        out = rend.render_rays(rays, orr)[0]
    ms, frames = rend.ctx.profile_end()
    torch.cuda.synchronize()
    wall_ms = (time.perf_counter() - t0) * 1e3
    assert frames == 3 and list(ms) == ['sampler_kernel', 'refine_kernel', 'nerf_kernel']
    assert all(v > 0 for v in ms.values()) and 3 * sum(ms.values()) < wall_ms
    assert torch.equal(out, ref)
    rend.ctx.profile_begin(2)                            # re-arm with a smaller window; nothing rendered -> zero frames
    assert rend.ctx.profile_end() == ({k: 0.0 for k in ms}, 0)


def test_ray_counts_past_two_gib_of_workspace(dev):
    """4.2 M rays in one call: the per-ray workspace is 0.94 GB and the int64 index output 0.27 GB; ray-indexed byte offsets pass 2^31 in the
    operator-level refine_in buffer checked below (2.4 GB).  The rays are a
    756 x 1008 frame repeated; each repetition must equal the single frame bit for bit, and a ragged tail must equal its prefix."""
    from pronerf_amd.render import Renderer
    H, W = 756, 1008
    scene = synth.make_scene(0, H=H, W=W, focal=815.13, rotate=True)
    rend = Renderer(synth.make_weights(0, 'trained'), max_rays=4_200_000, device=dev)
    rend.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
    rays, orr = rend.frame_rays(scene['K'], scene['c2w'], H, W)
    one = rend.render_rays(rays, orr, want_idx=True)
    one = (one[0].clone(), one[1].clone())
    n = H * W
    reps, tail = 5, 4_200_000 - 5 * n                    # 5 frames + 389 760 rays
    big_r = torch.cat([rays] * reps + [rays[:tail]]).contiguous()
    big_o = torch.cat([orr] * reps + [orr[:tail]]).contiguous()
    assert big_r.shape[0] == 4_200_000 and big_r.shape[0] * 144 * 4 > 2 ** 31
    out, idx = rend.render_rays(big_r, big_o, want_idx=True)
    # the operator-level projection writes refine_in [n,144] (2.4 GB): its last rows must equal the rows of the single frame
    from pronerf_amd import ops
    depth = ops.sampler_fwd(rend.sampler, big_r, want_idx=False, want_rgb=False)[0]
    rin = ops.refine_input(big_r, big_o, depth, rend.img4, rend.proj)
    d1 = ops.sampler_fwd(rend.sampler, rays, want_idx=False, want_rgb=False)[0]
    assert torch.equal(rin[reps * n:], ops.refine_input(rays, orr, d1, rend.img4, rend.proj)[:tail])
    del rin, depth
    for k in range(reps):
        assert torch.equal(out[k * n:(k + 1) * n], one[0]) and torch.equal(idx[k * n:(k + 1) * n], one[1]), k
    assert torch.equal(out[reps * n:], one[0][:tail]) and torch.equal(idx[reps * n:], one[1][:tail])


def test_workgroup_shapes_agree_bit_for_bit(dev):
    """pnrf_mlp_set_shape: wide (8 waves, one workgroup per CU) and narrow (4 waves, two per CU, half-width batches) launches of every fused
    stage run the same instruction stream per ray — rgb, depth and the sampler indices are identical, at ragged sizes, on both fine-net classes
    and operand types; the per-launch default equals both."""
    from pronerf_amd import synthetic
    from pronerf_amd.render import Renderer
    Hh, Ww = 150, 209                                # 31 350 rays: 245 narrow sampler batches + a ragged tail
    scene = synth.make_scene(3, H=Hh, W=Ww, rotate=True)
    w = synth.make_weights(3, 'trained')
    c = synthetic.make_nerfcls_weights(3, head_scale=0.3)      # NeRF-class fine net in pack order: pts 0..7, feature, alpha, views, rgb
    order = list(c['pts_linears']) + [c['feature_linear'], c['alpha_linear'], c['views_linears'][0], c['rgb_linear']]
    wc = dict(w, nerf={'W': [a for a, _ in order], 'b': [b for _, b in order]})
    for weights, variants in ((w, None), (w, {'nerf': 'f16', 'refine': 'bf16'}), (wc, None), (w, {'sampler': 'sampler_split'})):
        outs = []
        for shape in ('wide', 'narrow', None):
            rend = Renderer(weights, max_rays=Hh * Ww, device=dev, variants=variants, shape=shape)
            rend.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
            rays, or_rays = rend.frame_rays(scene['K'], scene['c2w'], Hh, Ww)
            rgbd, idx = rend.render_rays(rays, or_rays, want_idx=True)
            for n in (1000, 1, 129):
                part, _ = rend.render_rays(rays[:n].contiguous(), or_rays[:n].contiguous())
                assert torch.equal(part, rgbd[:n]), (shape, n)
            outs.append((rgbd.clone(), idx.clone(), rend.ctx.sampler_stats()))
            del rend
        for rgbd, idx, n2 in outs[1:]:
            assert torch.equal(rgbd, outs[0][0]) and torch.equal(idx, outs[0][1]) and n2 == outs[0][2]
        assert bool(torch.isfinite(outs[0][0]).all())
    with pytest.raises(Exception):
        Renderer(w, max_rays=8, device=dev, shape=3)
    with pytest.raises(Exception):
        Renderer(w, max_rays=8, device=dev, shape='tall')


def test_two_pass_workspace_needs_no_memset_between_calls(dev):
    """The context clears the two-pass sampler's counters once; every call leaves them at zero (last workgroup of pass 2).  Calls of
    different sizes back to back on one context give the same rows and the same second-pass counts as fresh contexts."""
    from pronerf_amd.render import Renderer
    Hh, Ww = 64, 90
    scene = synth.make_scene(1, H=Hh, W=Ww, rotate=True)
    w = synth.make_weights(1, 'trained')
    rend = Renderer(w, max_rays=Hh * Ww, device=dev)
    rend.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
    rays, or_rays = rend.frame_rays(scene['K'], scene['c2w'], Hh, Ww)
    full, _ = rend.render_rays(rays, or_rays)
    full = full.clone()
    n_full = rend.ctx.sampler_stats()
    assert 0 < n_full < Hh * Ww
    counts = []
    for n in (Hh * Ww, 1, 700, Hh * Ww, 129, 4097):
        part, _ = rend.render_rays(rays[:n].contiguous(), or_rays[:n].contiguous())
        assert torch.equal(part, full[:n]), n
        counts.append(rend.ctx.sampler_stats())
    assert counts[0] == n_full and counts[3] == n_full
    fresh = Renderer(w, max_rays=Hh * Ww, device=dev)
    fresh.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
    for n, c in zip((1, 700, 129, 4097), (counts[1], counts[2], counts[4], counts[5])):
        fresh.render_rays(rays[:n].contiguous(), or_rays[:n].contiguous())
        assert fresh.ctx.sampler_stats() == c


def test_context_kappa_setter(dev):
    """pnrf_ctx_set_sampler_kappa: a larger kappa sends more rays through the second pass, kappa = 0 fewer; NaN / inf are refused; the
    sort indices do not move between the default and a very large kappa (every ray through the split kernel)."""
    from pronerf_amd import _lib
    from pronerf_amd.render import Renderer
    Hh, Ww = 64, 90
    scene = synth.make_scene(1, H=Hh, W=Ww, rotate=True)
    rend = Renderer(synth.make_weights(1, 'trained'), max_rays=Hh * Ww, device=dev)
    rend.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
    rays, or_rays = rend.frame_rays(scene['K'], scene['c2w'], Hh, Ww)
    res = {}
    for k in (-1.0, 0.0, 16.0, 1e20):
        rend.ctx.set_sampler_kappa(k)
        _, idx = rend.render_rays(rays, or_rays, want_idx=True)
        res[k] = (idx.clone(), rend.ctx.sampler_stats())
    assert res[0.0][1] < res[-1.0][1] < res[16.0][1] <= res[1e20][1] == Hh * Ww
    assert torch.equal(res[-1.0][0], res[1e20][0]) and torch.equal(res[16.0][0], res[1e20][0])
    for bad in (float('nan'), float('inf'), 1e31):
        with pytest.raises(_lib.PnrfError):
            rend.ctx.set_sampler_kappa(bad)


def test_chunked_renderer_streams_equal_the_one_call_frame(dev):
    """ChunkedRenderer (configs[1]'s ray chunks): calls of <= chunk rays, serial on one stream or round-robin over several streams with a
    context each, eager and replayed as one hipGraph — the same rows as the one-call frame, bit for bit, with a ragged last chunk."""
    from pronerf_amd.render import ChunkedRenderer, Renderer
    Hh, Ww = 90, 131                                 # 11 790 rays = 11 chunks of 1024 + 526
    scene = synth.make_scene(2, H=Hh, W=Ww, rotate=True)
    rend = Renderer(synth.make_weights(2, 'trained'), max_rays=Hh * Ww, device=dev)
    rend.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
    rays, or_rays = rend.frame_rays(scene['K'], scene['c2w'], Hh, Ww)
    ref, _ = rend.render_rays(rays, or_rays)
    ref = ref.clone()
    for streams in (1, 3):
        ch = ChunkedRenderer(rend, 1024, streams)
        out = torch.zeros_like(ref)
        ch.render_rays(rays, or_rays, out)
        torch.cuda.synchronize()
        assert torch.equal(out, ref), streams
        out.zero_()
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ch.render_rays(rays, or_rays, out)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            ch.render_rays(rays, or_rays, out)
        out.zero_()
        g.replay(); g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, ref), ('graph', streams)
        del g, ch


def test_concurrent_streams_never_change_a_row(dev):
    """Regression test of round 4: calls on four streams at once (a context each) with the narrow shape forced and with the per-launch default —
    250 x 1024-ray calls per frame, three frames each: every row equals the one-call frame.  (Two 4-wave workgroups of different fused kernels
    on one CU returned wrong rows a few times per thousand calls; round 5 found the cause — compiler-generated packed-fp32 instructions in the refine
    epilogue, DESIGN.md §4.5 — and the library is built without them; a CU still holds at most one fused-MLP workgroup, for speed.)"""
    from pronerf_amd.render import ChunkedRenderer, Renderer
    Hh, Ww = 400, 640
    scene = synth.make_scene(4, H=Hh, W=Ww, rotate=True)
    for shape in ('narrow', None):
        rend = Renderer(synth.make_weights(4, 'trained'), max_rays=Hh * Ww, device=dev, shape=shape)
        rend.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
        rays, or_rays = rend.frame_rays(scene['K'], scene['c2w'], Hh, Ww)
        ref, _ = rend.render_rays(rays, or_rays)
        ref = ref.clone()
        ch = ChunkedRenderer(rend, 1024, 4)
        for rep in range(3):
            out = torch.zeros_like(ref)
            ch.render_rays(rays, or_rays, out)
            torch.cuda.synchronize()
            bad = int((out != ref).any(1).sum())
            assert bad == 0, (shape, rep, bad)
        del ch, rend


def test_cyclic_partition_shards_equal_the_frame(dev):
    """RayPartition('cyclic'): every rank's rays come out of pnrf_frame_rays_blocks_fwd as the rows partition.rows(rank) of the one-call frame's
    rays (bit for bit), the rendered tiles put through the gather index give the one-call frame, and bad block arguments are refused."""
    from pronerf_amd import _lib
    from pronerf_amd.render import RayPartition, Renderer
    Hh, Ww = 201, 333                                # 66 933 rays: 66 blocks of 1024 (the last one 373 rays) over 8 ranks
    scene = synth.make_scene(5, H=Hh, W=Ww, rotate=True)
    rend = Renderer(synth.make_weights(5, 'trained'), max_rays=Hh * Ww, device=dev)
    rend.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
    rays, or_rays = rend.frame_rays(scene['K'], scene['c2w'], Hh, Ww)
    ref, _ = rend.render_rays(rays, or_rays)
    ref = ref.clone()
    for world in (8, 3):
        part = RayPartition(Hh * Ww, world, 'cyclic')
        buf = torch.zeros(world * part.cmax, 4, device=dev)
        for rank in range(world):
            r, o = rend.frame_rays(scene['K'], scene['c2w'], Hh, Ww, **part.frame_rays_args(rank))
            rows = part.rows(rank).to(dev)
            assert torch.equal(r, rays.index_select(0, rows)) and torch.equal(o, or_rays.index_select(0, rows))
            tile, _ = rend.render_rays(r, o)
            buf[rank * part.cmax: rank * part.cmax + part.count(rank)] = tile
        assert torch.equal(buf.index_select(0, part.gather_index(dev)), ref)
    with pytest.raises(_lib.PnrfError):
        rend.frame_rays(scene['K'], scene['c2w'], Hh, Ww, first=0, count=Hh * Ww, block=1024, stride=512)        # overlapping blocks
    with pytest.raises(_lib.PnrfError):
        rend.frame_rays(scene['K'], scene['c2w'], Hh, Ww, first=7 * 1024, count=9 * 1024, block=1024, stride=8 * 1024)   # leaves the frame
