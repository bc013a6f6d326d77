"""GPU: the sampler-index sweep and the soak run as collected tests (they were stand-alone scripts in round 1).

* index sweep: 12 288 rays x 5 weight sets x both sampler kernels, identical sort indices outside the tie set;
* soak: object lifetimes (create / use / free) of the renderer and the trainer — repeated identical launches give identical bits, and
  device memory does not grow between the second and the last cycle (the first cycle pays one-time costs: code objects, the rocBLAS
  workspace)."""
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


@pytest.mark.parametrize('variant', ['default', 'sampler_f32'])
@pytest.mark.parametrize('seed,kind', [(0, 'trained'), (2, 'spread'), (3, 'trained'), (5, 'trained'), (1, 'default')])
def test_sampler_index_sweep(dev, seed, kind, variant):
    from pronerf_amd import ops
    w = synth.make_weights(seed, kind)
    scene = synth.make_scene(seed, H=96, W=128, rotate=True)
    fr = orc.frame_setup(scene)
    _, add, mul, depth = orc.sampler_forward(w['sampler'], fr['mm_input'])
    ds, idx, _, _ = orc.sort_gather(depth, add, mul, fr['rays'][:, 6:7], fr['rays'][:, 7:8])
    mlp = ops.PackedMLP(ops.NET_SAMPLER, w['sampler']['W'], w['sampler']['b'], variant=variant)
    g_ds, g_idx, *_ = ops.sampler_fwd(mlp, fr['rays'].to(dev))
    gap = (ds[:, 1:] - ds[:, :-1]).min(1)[0]
    free = gap > 1e-6
    bad = int((g_idx.cpu()[free] != idx[free]).any(1).sum())
    err = float((g_ds.cpu() - ds).abs().max())
    print(f'\n[index sweep] ({seed}, {kind}) {variant}: {len(gap)} rays, tie set {int((~free).sum())}, mismatching rays outside it {bad}, max depth err {err:.2e}')
    assert bad == 0 and err <= 2e-6
    if kind != 'default':
        assert int((~free).sum()) <= 2


def _free_bytes():
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    return torch.cuda.mem_get_info()[0]


def test_soak_renderer_lifetimes(dev):
    from pronerf_amd.render import Renderer
    H, W = 378, 504
    scene = synth.make_scene(0, H=H, W=W, rotate=True)
    w = synth.make_weights(0, 'trained')
    free, ref = [], None
    for rep in range(6):
        rend = Renderer(w, max_rays=H * W, device=dev)
        rend.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
        rays, orr = rend.frame_rays(scene['K'], scene['c2w'], H, W)
        for _ in range(100):
            out = rend.render_rays(rays, orr)[0]
        torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
        assert torch.equal(out, ref), 'repeated identical launches differ'
        del rend, rays, orr, out
        free.append(_free_bytes())
    growth = (free[1] - free[-1]) / 2 ** 20
    print(f'\n[soak] 600 frames over 6 renderer lifetimes; free MiB after each: {[round(x / 2 ** 20) for x in free]}')
    assert growth < 1.0, f'device memory grows across renderer lifetimes: {growth:.1f} MiB'


def test_soak_trainer_lifetimes(dev):
    from pronerf_amd import ops
    import test_train_gpu as T
    b = T._batch(0, 12, 16, 7)
    layers = orc.trainer_layers(b['w'])
    free, losses = [], []
    for rep in range(5):
        tr = ops.Trainer([W_ for W_, _ in layers], [x for _, x in layers], max_rays=b['N'], device=dev, max_samples=32)
        img4 = ops.images_pack(T.cu(b['images'], dev))
        args = (T.cu(b['rays'], dev), T.cu(b['or_rays'], dev), T.cu(b['target'], dev), img4, T.cu(b['poses'], dev), T.cu(b['K'], dev),
                b['ref_nos'].to(dev).contiguous())
        rs = np.random.RandomState(rep)
        for it in range(90):
            if it % 2:
                jit = torch.from_numpy(np.minimum(np.abs(rs.randn(b['N'], 32)) / 5, 0.99).astype(np.float32)).to(dev)
                L, _ = tr.explore_fwd_bwd(*args, n_mult=4, dir1=1, jitter=jit, dir2=-1, raw_noise=None, want_rgb=False)
                tr.adam_step(5e-4, weight_decay=5e-8, nerf_only=True)
            else:
                L, _ = tr.fwd_bwd(*args, jitter=T.cu(b['jitter'], dev), jitter_dir=1, raw_noise=T.cu(b['noise'], dev), want_rgb=False)
                tr.adam_step(5e-4, weight_decay=5e-8)
        losses.append(float(L[0]))
        del tr, img4, args, L
        free.append(_free_bytes())
    growth = (free[1] - free[-1]) / 2 ** 20
    print(f'\n[soak] 450 training iterations over 5 trainer lifetimes; final losses {[round(x, 5) for x in losses]}; free MiB {[round(x / 2 ** 20) for x in free]}')
    assert all(np.isfinite(losses)) and growth < 1.0, f'device memory grows across trainer lifetimes: {growth:.1f} MiB'
