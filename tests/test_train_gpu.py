"""GPU: stage-2 training step — operator-level backward passes against torch.autograd of the oracle's
forward functions, then the whole step (loss, the gradients of all 26 Linear layers, Adam updates) against the oracle's
``loss.backward()`` + ``torch.optim.Adam``, and against goldens from the reference's own training iteration.
Tolerances: everything is fp32; differences are summation order (rocBLAS vs CPU BLAS) -> relative L2 error per tensor."""
import os
import sys

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


def cu(x, dev):
    return torch.as_tensor(x, dtype=torch.float32).to(dev).contiguous()


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize('variant', ['addmul_noise', 'plain_white', 'addmul_white'])
def test_composite_backward(dev, variant):
    from pronerf_amd import ops
    rs = np.random.RandomState(3)
    n, S = 300, 8
    raw = torch.from_numpy(rs.randn(n, S, 4).astype(np.float32) * 1.5).requires_grad_()
    z = torch.from_numpy(np.sort(rs.uniform(0.05, 0.95, (n, S)), -1).astype(np.float32)).requires_grad_()
    d = torch.from_numpy(rs.randn(n, 3).astype(np.float32))
    use_am = variant.startswith('addmul')
    add = torch.from_numpy(rs.randn(n, S).astype(np.float32)).requires_grad_() if use_am else None
    mul = torch.from_numpy((rs.randn(n, S) + 0.7).astype(np.float32)).requires_grad_() if use_am else None
    noise = torch.from_numpy(rs.randn(n, S).astype(np.float32)) if 'noise' in variant else None
    white = 'white' in variant
    g = torch.from_numpy(rs.randn(n, 3).astype(np.float32))
    rgb = orc.raw2outputs(raw, z, d, add, mul, noise=noise, white_bkgd=white)[0]
    rgb.backward(g)
    d_raw, d_z, d_add, d_mul = ops.composite_bwd(cu(raw.detach(), dev), cu(z.detach(), dev), cu(d, dev), cu(g, dev), None if add is None else cu(add.detach(), dev),
                                                 None if mul is None else cu(mul.detach(), dev), None if noise is None else cu(noise, dev), white_bkgd=white)
    assert rel(d_raw, raw.grad) < 2e-5 and rel(d_z, z.grad) < 2e-5
    if use_am:
        assert rel(d_add, add.grad) < 2e-5 and rel(d_mul, mul.grad) < 2e-5
        assert float((mul.detach() <= 0).float().mean()) > 0.1            # the relu(mul) = 0 branch is exercised


def test_posenc_backward(dev):
    from pronerf_amd import ops
    rs = np.random.RandomState(5)
    x = torch.from_numpy(rs.uniform(-1.5, 1.5, (500, 3)).astype(np.float32)).requires_grad_()
    g = torch.from_numpy(rs.randn(500, 63).astype(np.float32))
    orc.posenc(x, 10).backward(g)
    assert rel(ops.posenc_bwd(cu(x.detach(), dev), cu(g, dev), 10), x.grad) < 1e-5


def test_sampler_head_forward_backward(dev):
    from pronerf_amd import ops
    rs = np.random.RandomState(7)
    n = 400
    y = torch.from_numpy(rs.randn(n, 27).astype(np.float32)).requires_grad_()
    rays = torch.from_numpy(rs.randn(n, 11).astype(np.float32)); rays[:, 6] = 0.0; rays[:, 7] = 1.0
    depth = torch.sigmoid(y[:, :8])
    ds, idx, adds, muls = orc.sort_gather(depth, y[:, 8:16], y[:, 16:24], rays[:, 6:7], rays[:, 7:8])
    rgb = torch.sigmoid(y[:, 24:])
    gd, ga, gm, gr = [torch.from_numpy(rs.randn(n, k).astype(np.float32)) for k in (8, 8, 8, 3)]
    (ds * gd).sum().add((adds * ga).sum()).add((muls * gm).sum()).add((rgb * gr).sum()).backward()
    D, I, A, M, RGB = ops.sampler_head_fwd(cu(y.detach(), dev), cu(rays, dev))
    np.testing.assert_array_equal(I.cpu().numpy(), idx.numpy())
    np.testing.assert_allclose(D.cpu().numpy(), ds.detach().numpy(), rtol=0, atol=1e-6)
    np.testing.assert_array_equal(A.cpu().numpy(), adds.detach().numpy())
    np.testing.assert_allclose(RGB.cpu().numpy(), rgb.detach().numpy(), rtol=0, atol=1e-6)
    dy = ops.sampler_head_bwd(cu(y.detach(), dev), cu(rays, dev), I, cu(gd, dev), cu(ga, dev), cu(gm, dev), cu(gr, dev))
    assert rel(dy, y.grad) < 1e-5


@pytest.mark.parametrize('jdir', [0, 1, -1])
def test_refine_head_forward_backward(dev, jdir):
    from pronerf_amd import ops
    rs = np.random.RandomState(11 + jdir)
    n = 300
    y = torch.from_numpy(rs.randn(n, 35).astype(np.float32)).requires_grad_()
    rays = torch.from_numpy(rs.randn(n, 11).astype(np.float32)); rays[:, 6] = 0.0; rays[:, 7] = 1.0
    D = torch.from_numpy(np.sort(rs.uniform(0.02, 0.98, (n, 8)), -1).astype(np.float32)).requires_grad_()
    jit = None if jdir == 0 else torch.from_numpy(np.minimum(np.abs(rs.randn(n, 8)) / 5, 1 - 2e-6).astype(np.float32))
    near, far = rays[:, 6:7], rays[:, 7:8]
    z = orc.interval_refine(D, torch.sigmoid(y[:, :8]), near, far)
    if jdir > 0:
        z = z + jit * (z - torch.cat([z[:, 1:], far], 1)).abs()
    elif jdir < 0:
        z = z - jit * (z - torch.cat([near, z[:, :-1]], 1)).abs()
    pts = rays[:, None, 0:3] + rays[:, None, 3:6] * z[..., None] + 1e-2 * torch.tanh(y[:, 8:32]).reshape(n, 8, 3)
    rgb0 = torch.sigmoid(y[:, 32:])
    gp, gz, gr = torch.from_numpy(rs.randn(n, 8, 3).astype(np.float32)), torch.from_numpy(rs.randn(n, 8).astype(np.float32)), torch.from_numpy(rs.randn(n, 3).astype(np.float32))
    ((pts * gp).sum() + (z * gz).sum() + (rgb0 * gr).sum()).backward()
    zp, zz, pp, r0 = ops.refine_head_fwd(cu(y.detach(), dev), cu(rays, dev), cu(D.detach(), dev), None if jit is None else cu(jit, dev), jdir or 1)
    np.testing.assert_allclose(zz.cpu().numpy(), z.detach().numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(pp.cpu().numpy(), pts.detach().numpy(), rtol=0, atol=5e-6)
    np.testing.assert_allclose(r0.cpu().numpy(), rgb0.detach().numpy(), rtol=0, atol=1e-6)
    dy, dD = ops.refine_head_bwd(cu(y.detach(), dev), cu(rays, dev), cu(D.detach(), dev), zp, cu(gp, dev), cu(gz, dev), cu(gr, dev),
                                 None if jit is None else cu(jit, dev), jdir or 1)
    assert rel(dy, y.grad) < 2e-5 and rel(dD, D.grad) < 2e-5


def _batch(seed, H, W, nv, own=2):
    scene = synth.make_scene(seed, H=H, W=W, n_views=nv, sigma_t=0.2, rotate=True)
    w = synth.make_weights(seed, 'trained'); w['nerfcls'] = synth.make_nerfcls_weights(seed, head_scale=0.3)
    poses = torch.from_numpy(scene['poses']); images = torch.from_numpy(scene['images']).permute(0, 3, 1, 2).contiguous()
    K = torch.from_numpy(scene['K'])
    fr = orc.frame_setup({**scene, 'c2w': scene['poses'][own]})
    rays, or_rays = fr['rays'], fr['or_rays']
    N = rays.shape[0]
    rs = np.random.RandomState(100 + seed)
    target = torch.from_numpy(scene['images'][own].reshape(-1, 3).astype(np.float32))
    order = np.sort(rs.choice(np.arange(0, nv - 1), 4, replace=False)).astype(np.int64)      # rank positions after dropping self
    ref_nos = orc.select_neighbors_train(poses[own][None].expand(N, -1, -1), poses, 4, order)
    jitter = torch.from_numpy(np.minimum(np.abs(rs.randn(N, 8)) / 5, 1 - 2e-6).astype(np.float32))
    noise = torch.from_numpy(rs.randn(N, 8).astype(np.float32))
    return dict(w=w, rays=rays, or_rays=or_rays, target=target, images=images, poses=poses, K=K, ref_nos=ref_nos, jitter=jitter, noise=noise, N=N)


def low_frequency_nerf(w, keep_octaves):
    """Zero the fine net's input weights of the positional octaves >= keep_octaves (pts_linears.0 and the skip columns of pts_linears.5).  The
    kernels are unchanged (they still encode ten octaves), but d(raw)/d(pts) loses its 2^9 factor: the fp32 round-off of the chain then stays at
    the 1e-5 level instead of 1e-3 .. 5e-2, and the gradient bounds of the HIP trainer can be as tight as a kernel regression needs."""
    wc = w['nerfcls']
    for li in (0, 5):
        W, b = wc['pts_linears'][li]
        W = W.copy()
        W[:, 3 + 6 * keep_octaves:63] = 0.0
        wc['pts_linears'][li] = (W, b)
    return w


def _oracle_step(layers, b, jdir, white, a_mmrgb):
    loss, img_loss, o = orc.stage2_loss(layers, b['rays'], b['or_rays'], b['target'], b['images'], b['poses'], b['K'], b['ref_nos'], jitter=b['jitter'],
                                        jitter_dir=jdir, raw_noise=b['noise'], white_bkgd=white, a_mmrgb=a_mmrgb)
    return loss, img_loss, o


def _oracle_grads(b, jdir, white, a_mmrgb, dtype):
    """loss and per-layer gradients of the oracle in `dtype` (fp64: the arbiter; fp32: what the reference computes)."""
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        bb = {k: (v.to(dtype) if isinstance(v, torch.Tensor) and v.is_floating_point() else v) for k, v in b.items()}
        layers = [(torch.tensor(W, dtype=dtype, requires_grad=True), torch.tensor(x, dtype=dtype, requires_grad=True)) for W, x in orc.trainer_layers(b['w'])]
        loss, img_loss, o = _oracle_step(layers, bb, jdir, white, a_mmrgb)
        loss.backward()
    finally:
        torch.set_default_dtype(old)
    return float(loss.detach()), float(img_loss.detach()), o, [(W.grad, x.grad) for W, x in layers]


@pytest.mark.parametrize('products', ['f16x2', 'f32'])
@pytest.mark.parametrize('jdir,white,a_mmrgb', [(1, False, 0.0), (-1, True, 1.0)])
def test_stage2_step_gradients_vs_oracle_autograd(dev, jdir, white, a_mmrgb, products):
    """Gradients of all 26 Linear layers on the ill-conditioned chain the configs really have (2^9 positional frequencies, a 1e10 last
    interval), with the split-fp16 layer products (the default) and with the exact-fp32 ones.  Arbiter = the oracle run in fp64.  Torch's own
    fp32 CPU run of this batch is 3e-3 .. 1.8e-2 away from it per tensor (tools/diag_grad.py: 3e-3 .. 5e-2 over six seeds), so what can be
    asserted HERE is that the HIP trainer's round-off is of that size: per tensor within 6x of the CPU run's distance (a ReLU mask within
    round-off of zero flips in one implementation and not the other: measured up to 4.2x), within 2.5x of the largest distance any tensor of
    the CPU run has (the noise amplitude of the batch; measured <= 1.6x), median ratio over the 52 tensors < 2 (measured 0.1 .. 1.4).  A
    regression of 1e-2 in a product kernel would hide in this noise: the tight bound (1e-4 per tensor) is asserted on a well-conditioned net
    in test_stage2_step_gradients_tight_on_a_well_conditioned_net below."""
    from pronerf_amd import ops
    b = _batch(0, 12, 16, 7)
    loss64, img64, o64, g64 = _oracle_grads(b, jdir, white, a_mmrgb, torch.float64)
    _, _, _, g32 = _oracle_grads(b, jdir, white, a_mmrgb, torch.float32)
    layers = orc.trainer_layers(b['w'])
    tr = ops.Trainer([W for W, _ in layers], [x for _, x in layers], max_rays=b['N'], device=dev)
    tr.set_products(products)
    img4 = ops.images_pack(cu(b['images'], dev))
    L, rgb = tr.fwd_bwd(cu(b['rays'], dev), cu(b['or_rays'], dev), cu(b['target'], dev), img4, cu(b['poses'], dev), cu(b['K'], dev),
                        b['ref_nos'].to(dev).contiguous(), jitter=cu(b['jitter'], dev), jitter_dir=jdir, raw_noise=cu(b['noise'], dev), white_bkgd=white,
                        a_mmrgb=a_mmrgb)
    Lh = L.cpu().numpy()
    assert abs(Lh[0] - loss64) < 2e-5 * max(1.0, loss64) and abs(Lh[1] - img64) < 2e-5
    assert bool((o64['edge_margin'] > 1e-5).all())
    assert orc.psnr(rgb.cpu(), o64['rgb_map1'].detach().float()) > 80.0                      # fp32 path
    cmax = max(max(rel(g32[li][0], g64[li][0]), rel(g32[li][1], g64[li][1])) for li in range(26))     # noise amplitude of this batch
    ratios = []
    for li in range(26):
        gW, gb = tr.read('grad', li)
        eW, eb = rel(gW, g64[li][0]), rel(gb, g64[li][1])
        cW, cb = rel(g32[li][0], g64[li][0]), rel(g32[li][1], g64[li][1])
        assert eW < 2.5 * cmax and eb < 2.5 * cmax, (li, eW, eb, cmax)
        ratios += [eW / (cW + 1e-5), eb / (cb + 1e-5)]
        assert eW < 6 * cW + 1e-5 and eb < 6 * cb + 1e-5, (li, eW, cW, eb, cb)
    assert float(np.median(ratios)) < 2.0, sorted(ratios)[-8:]


@pytest.mark.parametrize('products', ['f16x2', 'f32'])
@pytest.mark.parametrize('jdir,white,a_mmrgb', [(1, False, 0.0), (-1, True, 1.0)])
def test_stage2_step_gradients_tight_on_a_well_conditioned_net(dev, jdir, white, a_mmrgb, products):
    """The same iteration with the fine net's input weights of the positional octaves >= 2 set to zero (low_frequency_nerf): every kernel
    runs as before, but the chain no longer amplifies fp32 round-off by 2^9, so torch's fp32 CPU run is within 7e-6 .. 6e-5 of the fp64 run
    and the HIP trainer can be held to 1e-4 per gradient tensor with either product arithmetic (measured, tools/diag_grad.py 2: exact fp32
    <= 9.0e-6, split fp16 <= 2.1e-5) — 100x below the noise of the full-frequency test above: a relative error of 1e-3 in any layer product,
    epilogue or reduction fails here.  What remains are discrete events — a ReLU mask, a bilinear tap or a sort order decided by round-off
    flips in one fp32 implementation and not in fp64: worth 1e-4 .. 4e-3 of the gradient tensors it reaches, on every second batch or so, and
    WHICH batch changes with any change of rounding anywhere upstream (an event reaches every tensor upstream of it).  So eight batches
    (seeds) are run: the median error over all tensors and batches must be < 2e-5 (measured 3e-7 .. 2e-6), each of the 52 tensors must be within
    1e-4 on at least two of the eight batches (a broken kernel fails on every batch; events leave 2 .. 5 of 5 clean), none may exceed the size
    of an event (5e-3) on any; the loss is tight on all eight."""
    from pronerf_amd import ops
    worst = {}
    for seed in range(8):
        b = _batch(seed, 12, 16, 7)
        low_frequency_nerf(b['w'], 2)
        loss64, img64, o64, g64 = _oracle_grads(b, jdir, white, a_mmrgb, torch.float64)
        layers = orc.trainer_layers(b['w'])
        tr = ops.Trainer([W for W, _ in layers], [x for _, x in layers], max_rays=b['N'], device=dev)
        tr.set_products(products)
        img4 = ops.images_pack(cu(b['images'], dev))
        L, rgb = tr.fwd_bwd(cu(b['rays'], dev), cu(b['or_rays'], dev), cu(b['target'], dev), img4, cu(b['poses'], dev), cu(b['K'], dev),
                            b['ref_nos'].to(dev).contiguous(), jitter=cu(b['jitter'], dev), jitter_dir=jdir, raw_noise=cu(b['noise'], dev), white_bkgd=white,
                            a_mmrgb=a_mmrgb)
        Lh = L.cpu().numpy()
        assert abs(Lh[0] - loss64) < 1e-6 * max(1.0, loss64) and abs(Lh[1] - img64) < 1e-6, (seed, Lh, loss64)
        worst[seed] = [e for li, (gW, gb) in ((li, tr.read('grad', li)) for li in range(26)) for e in (rel(gW, g64[li][0]), rel(gb, g64[li][1]))]
    E = np.array([worst[s] for s in sorted(worst)])                      # [batch, tensor]
    print(f'\n[tight] case {(jdir, white, a_mmrgb)} products {products}: relative gradient error, max over the 52 tensors per batch: '
          + ', '.join(f'{e:.1e}' for e in E.max(1)) + f'; median over tensors and batches {np.median(E):.1e}; tensors within 1e-4 on all batches: '
          f'{int((E < 1e-4).all(0).sum())} / 52, worst tensor: {int((E < 1e-4).sum(0).min())} of {E.shape[0]} batches')
    assert bool(((E < 1e-4).sum(0) >= 2).all()) and float(E.max()) < 5e-3, E.max(1)
    assert float(np.median(E)) < 2e-5


def test_graph_replay_equals_kernel_by_kernel(dev):
    """The iteration replayed as a hipGraph (default) and launched kernel by kernel (set_graph(False)): same loss, same gradients, bit for
    bit, for the joint iteration and for an exploration iteration; a second batch through the cached graph; a side stream as well as the default."""
    from pronerf_amd import ops
    b = _batch(0, 12, 16, 7)
    layers = orc.trainer_layers(b['w'])
    img4 = ops.images_pack(cu(b['images'], dev))
    rs = np.random.RandomState(5)
    jit32 = torch.from_numpy(np.minimum(np.abs(rs.randn(b['N'], 32)) / 5, 0.99).astype(np.float32)).to(dev)

    def run(graph, stream=None):
        tr = ops.Trainer([W for W, _ in layers], [x for _, x in layers], max_rays=b['N'], device=dev, max_samples=32)
        tr.set_graph(graph)
        out = []
        with torch.cuda.stream(stream) if stream is not None else torch.cuda.device(dev):
            for rep in range(2):                                       # the second pass replays the cached graph on other data
                rays = cu(b['rays'], dev) * (1.0 + 1e-3 * rep)
                args = (rays, cu(b['or_rays'], dev), cu(b['target'], dev), img4, cu(b['poses'], dev), cu(b['K'], dev), b['ref_nos'].to(dev).contiguous())
                L, rgb = tr.fwd_bwd(*args, jitter=cu(b['jitter'], dev), jitter_dir=1, raw_noise=cu(b['noise'], dev))
                out += [L.clone(), rgb.clone()] + [g.clone() for i in range(26) for g in tr.read('grad', i)]
                L, rgb = tr.explore_fwd_bwd(*args, n_mult=4, dir1=1, jitter=jit32, dir2=-1)
                out += [L.clone(), rgb.clone()] + [g.clone() for i in range(14, 26) for g in tr.read('grad', i)]
        torch.cuda.synchronize()
        return out
    ref = run(False)
    for got in (run(True), run(True, torch.cuda.Stream(device=dev))):
        assert len(got) == len(ref)
        for k, (x, y) in enumerate(zip(got, ref)):
            assert torch.equal(x, y), (k, tuple(x.shape), float((x - y).abs().max()), float(y.abs().max()))


def test_split_fp16_products_keep_tiny_gradients(dev):
    """The split-fp16 products scale each gradient tensor by a power of two taken from its recorded maximum before splitting it into fp16
    pairs.  A batch whose target is the rendered image itself plus 1e-4 noise has layer gradients of 1e-9 .. 1e-6 — below the fp16 range: without
    the scaling the weight gradients would flush to zero.  Here they agree with the exact-fp32 products' as well as two fp32 runs agree."""
    from pronerf_amd import ops
    b = _batch(0, 12, 16, 7)
    layers = orc.trainer_layers(b['w'])
    img4 = ops.images_pack(cu(b['images'], dev))
    args = lambda target: (cu(b['rays'], dev), cu(b['or_rays'], dev), target, img4, cu(b['poses'], dev), cu(b['K'], dev), b['ref_nos'].to(dev).contiguous())
    kw = dict(jitter=cu(b['jitter'], dev), jitter_dir=1, raw_noise=cu(b['noise'], dev))
    grads = {}
    target = None
    for products in ('f32', 'f16x2'):
        tr = ops.Trainer([W for W, _ in layers], [x for _, x in layers], max_rays=b['N'], device=dev)
        tr.set_products(products)
        if target is None:
            _, rgb = tr.fwd_bwd(*args(cu(b['target'], dev)), **kw)
            g = torch.Generator(device='cpu').manual_seed(3)
            target = (rgb + 1e-4 * torch.randn(rgb.shape, generator=g).to(dev)).contiguous()
        L, _ = tr.fwd_bwd(*args(target), **kw)
        assert float(L[1]) < 1e-7                                             # mse of a 1e-4 residual
        grads[products] = [tr.read('grad', i) for i in range(26)]
    worst = 0.0
    for li in range(14, 26):                                                  # the NeRF layers carry all of the loss here (a_mmrgb = 0)
        for x, y in zip(grads['f16x2'][li], grads['f32'][li]):
            assert 0 < float(y.abs().max()) < 1e-3, (li, float(y.abs().max()))
            worst = max(worst, rel(x, y))
    assert worst < 3e-2, worst                                                # (flushed to zero: 1.0)


def test_larger_batch_takes_the_same_gradients_on_both_product_paths(dev):
    """4 200 rays (not a multiple of any tile size): the ELU nets' layer chains then run on 32-row tiles and the NeRF layers on 33 600 rows with
    ragged last tiles.  The exact-fp32 products use none of those kernels (no chains, no split-fp16 tiles), so the two runs are independent
    implementations of the same iteration: same loss, gradients as close as two fp32 summation orders are on this ill-conditioned chain."""
    from pronerf_amd import ops
    b = _batch(1, 60, 70, 7)
    assert b['N'] == 4200
    layers = orc.trainer_layers(b['w'])
    img4 = ops.images_pack(cu(b['images'], dev))
    args = (cu(b['rays'], dev), cu(b['or_rays'], dev), cu(b['target'], dev), img4, cu(b['poses'], dev), cu(b['K'], dev), b['ref_nos'].to(dev).contiguous())
    kw = dict(jitter=cu(b['jitter'], dev), jitter_dir=-1, raw_noise=cu(b['noise'], dev), a_mmrgb=1.0)
    res = {}
    for products in ('f32', 'f16x2'):
        tr = ops.Trainer([W for W, _ in layers], [x for _, x in layers], max_rays=b['N'], device=dev)
        tr.set_products(products)
        L, rgb = tr.fwd_bwd(*args, **kw)
        res[products] = (L.cpu().numpy(), rgb.cpu(), [tr.read('grad', i) for i in range(26)])
    La, Lb = res['f32'][0], res['f16x2'][0]
    assert np.all(np.abs(La - Lb) < 2e-6 * np.maximum(1.0, np.abs(La))), (La, Lb)
    assert orc.psnr(res['f16x2'][1], res['f32'][1]) > 90.0
    worst = max(rel(x, y) for li in range(26) for x, y in zip(res['f16x2'][2][li], res['f32'][2][li]))
    assert worst < 2e-2, worst


def test_adam_step_matches_torch_optim(dev):
    """optimizer.step(): identical gradients in, torch.optim.Adam's parameters out (three steps, weight decay, bias correction)."""
    from pronerf_amd import ops
    w = synth.make_weights(0, 'trained'); w['nerfcls'] = synth.make_nerfcls_weights(0, head_scale=0.3)
    layers = [(torch.tensor(W), torch.tensor(x)) for W, x in orc.trainer_layers(w)]
    params = [p.clone().requires_grad_() for pair in layers for p in pair]
    opt = torch.optim.Adam(params, lr=5e-4, betas=(0.9, 0.999), weight_decay=5e-8)
    tr = ops.Trainer([W for W, _ in layers], [x for _, x in layers], max_rays=8, device=dev)
    rs = np.random.RandomState(0)
    for step in range(3):
        for li in range(26):
            gW = torch.from_numpy((rs.randn(*layers[li][0].shape) * 10.0 ** rs.uniform(-6, 0)).astype(np.float32))
            gb = torch.from_numpy((rs.randn(*layers[li][1].shape) * 1e-3).astype(np.float32))
            params[2 * li].grad, params[2 * li + 1].grad = gW.clone(), gb.clone()
            tr.write('grad', li, gW.to(dev), gb.to(dev))
        opt.step()
        tr.adam_step(5e-4 * 0.1 ** (step / 250000), weight_decay=5e-8)
        for g in opt.param_groups:
            g['lr'] = 5e-4 * 0.1 ** ((step + 1) / 250000)                                    # refine2.py:872-878
        for li in range(26):
            pW, pb = tr.read('param', li)
            np.testing.assert_allclose(pW.cpu().numpy(), params[2 * li].detach().numpy(), rtol=0, atol=2e-7)
            np.testing.assert_allclose(pb.cpu().numpy(), params[2 * li + 1].detach().numpy(), rtol=0, atol=2e-7)
    mW, _ = tr.read('m', 3); vW, _ = tr.read('v', 3)
    st = opt.state[params[6]]
    np.testing.assert_allclose(mW.cpu().numpy(), st['exp_avg'].numpy(), rtol=1e-4, atol=1e-8)
    np.testing.assert_allclose(vW.cpu().numpy(), st['exp_avg_sq'].numpy(), rtol=1e-4, atol=1e-14)


@pytest.mark.parametrize('products', ['f16x2', 'f32'])
def test_training_loop_reduces_the_loss_like_the_oracle(dev, products):
    """Ten iterations on one fixed batch: the loss of the HIP trainer follows the oracle's (torch autograd + torch Adam).  On the
    well-conditioned net (low_frequency_nerf) the gradients of two fp32 implementations agree to ~1e-5, so the trajectories stay together:
    1e-6 before the first update, 1e-5 after it (measured 1.6e-7).  From then on they separate — the first Adam step moves every weight by
    +-lr whatever the size of its gradient, including the zeroed high-octave input weights, so the net is full-frequency again and m / sqrt(v)
    amplifies round-off-sized gradient differences (measured over the ten steps: 1.1e-2 with the split-fp16 products, 3.7e-2 with the
    exact-fp32 ones, 1.0e-1 after a one-ulp change of the ray encoding): from the third loss on only the common descent is asserted (both down
    by > 10 %, final losses within 25 % of each other)."""
    from pronerf_amd import ops
    b = _batch(0, 12, 16, 7)
    low_frequency_nerf(b['w'], 2)
    layers = [(torch.tensor(W, requires_grad=True), torch.tensor(x, requires_grad=True)) for W, x in orc.trainer_layers(b['w'])]
    opt = torch.optim.Adam([p for pair in layers for p in pair], lr=5e-4, betas=(0.9, 0.999), weight_decay=5e-8)
    tr = ops.Trainer([W.detach() for W, _ in layers], [x.detach() for _, x in layers], max_rays=b['N'], device=dev)
    tr.set_products(products)
    img4 = ops.images_pack(cu(b['images'], dev))
    args = (cu(b['rays'], dev), cu(b['or_rays'], dev), cu(b['target'], dev), img4, cu(b['poses'], dev), cu(b['K'], dev), b['ref_nos'].to(dev).contiguous())
    kw = dict(jitter=cu(b['jitter'], dev), jitter_dir=1, raw_noise=cu(b['noise'], dev))
    ref, got = [], []
    for _ in range(10):
        opt.zero_grad()
        loss, _, _ = _oracle_step(layers, b, 1, False, 0.0)
        loss.backward(); opt.step()
        ref.append(float(loss.detach()))
        L, _ = tr.fwd_bwd(*args, **kw, want_rgb=False)
        tr.adam_step(5e-4, weight_decay=5e-8)
        got.append(float(L[0]))
    print(f'\n[loop] products {products}: loss {got[0]:.6f} -> {got[-1]:.6f} (oracle {ref[0]:.6f} -> {ref[-1]:.6f}); max relative difference '
          f'{max(abs(a - c) / c for a, c in zip(got, ref)):.2e}, after the first update {abs(got[1] - ref[1]) / ref[1]:.2e}')
    assert got[-1] < 0.9 * got[0] and ref[-1] < 0.9 * ref[0]
    np.testing.assert_allclose(got[0], ref[0], rtol=1e-6)
    np.testing.assert_allclose(got[1], ref[1], rtol=1e-5)
    np.testing.assert_allclose(got[-1], ref[-1], rtol=0.25)


@pytest.mark.parametrize('name', ['stage2_step_12x16', 'stage2_step_white_mmrgb_10x14'])
def test_stage2_step_vs_reference_golden(dev, golden_dir, name):
    """The HIP trainer against the reference's own training iteration.  (i) vs the float64 run of the reference: loss, image,
    and every gradient tensor no further from it than 6x (median over tensors: 3x) the distance of a CPU fp32 run of the same iteration (the chain's
    fp32 noise differs per case: 2e-3 .. 2e-2); (ii) vs the fp32 run: loss, image, parameters after optimizer.step()."""
    import train_golden_util as U
    from pronerf_amd import ops
    g64, b64 = U.load_case(golden_dir, name + '_f64')
    _, _, _, l32 = U.oracle_grads(b64, torch.float32)
    st = int(g64['stride'])
    for gname, (g, b) in (('f64', (g64, b64)), ('f32', U.load_case(golden_dir, name))):
        layers = orc.trainer_layers(b['w'])
        tr = ops.Trainer([W for W, _ in layers], [x for _, x in layers], max_rays=b['N'], device=dev)
        img4 = ops.images_pack(cu(b['images'], dev))
        L, rgb = tr.fwd_bwd(cu(b['rays'], dev), cu(b['or_rays'], dev), cu(b['target'], dev), img4, cu(b['poses'], dev), cu(b['K'], dev),
                            b['ref_nos'].to(dev).contiguous(), jitter=cu(b['jitter'], dev), jitter_dir=b['jdir'], raw_noise=cu(b['noise'], dev),
                            white_bkgd=b['white'], a_mmrgb=b['a_mmrgb'])
        Lh = L.cpu().numpy()
        assert abs(Lh[0] - float(g['loss'])) < 2e-5 and abs(Lh[1] - float(g['img_loss'])) < 2e-5
        np.testing.assert_allclose(rgb.cpu().numpy(), g['rgb_map1'], rtol=0, atol=2e-4)
        grads = [tr.read('grad', i) for i in range(26)]
        if gname == 'f64':
            ratios = []
            for i in range(26):
                for k, (mine, cpu32) in enumerate(((grads[i][0].reshape(-1)[::st], l32[i][0].grad.reshape(-1)[::st]), (grads[i][1], l32[i][1].grad))):
                    want = g[('gW_%d' if k == 0 else 'gb_%d') % i]
                    e, n = U.rel(mine, want), U.rel(cpu32, want)
                    # single tensors: both are one draw of the round-off noise; the 1e-3 floor covers a single ReLU mask flip (one
                    # activation within 1e-7 of zero), a discrete event worth ~1e-4 of a 256x256 gradient where the smooth noise is 3e-6
                    assert e < 6 * n + 1e-3 and e < 0.1, (i, k, e, n)
                    ratios.append(e / (n + 1e-7))
            assert float(np.median(ratios)) < 3.0, float(np.median(ratios))    # as a whole: the same noise level as torch's fp32 CPU run
        else:
            tr.adam_step(b['lr'], weight_decay=b['wd'])
            # the fp32 golden is one more draw of the same round-off noise: per tensor its distance n from the fp64 golden is known, the HIP
            # trainer was held to 6 n + 1e-3 of the fp64 golden above, so it is within 7 n + 1e-3 of this one
            U.check_against_golden(g, grads, [tr.read('param', i) for i in range(26)], tol_grad=U.noise_tolerances(g, g64, 7.0, 1e-3), tol_norm=5e-2)


@pytest.mark.parametrize('n_mult,dir1,dir2', [(16, 1, -1), (32, -1, 1)])
def test_exploration_beyond_64_samples(dev, n_mult, dir1, dir2):
    """BASELINE.json configs[4] words the exploration path as "256 samples/ray"; the reference draws n_mult from 1..8 (base.py:689-729), so
    128 and 256 samples per ray (n_mult 16, 32) are the stress bound of a runtime S: forward (explored depths, rgb) and the NeRF gradients of
    one odd stage-1 iteration against the oracle's autograd in float64, next to the oracle's own fp32 run."""
    from pronerf_amd import ops
    b = _batch(1, 8, 10, 6)                                     # 80 rays x 256 samples = 20 480 rows in the NeRF layers
    N, S = b['N'], 8 * n_mult
    rs = np.random.RandomState(n_mult)
    jitter = torch.from_numpy(np.minimum(np.abs(rs.randn(N, S)) / 5, 0.99).astype(np.float32))
    noise = torch.from_numpy(rs.randn(N, S).astype(np.float32))

    def oracle(dtype):
        old = torch.get_default_dtype()
        torch.set_default_dtype(dtype)
        try:
            layers = [(torch.tensor(W, dtype=dtype, requires_grad=True), torch.tensor(x, dtype=dtype, requires_grad=True)) for W, x in orc.trainer_layers(b['w'])]
            f = lambda x: x.to(dtype) if x.is_floating_point() else x
            loss, _, o = orc.stage1_loss(layers, f(b['rays']), f(b['or_rays']), f(b['target']), f(b['images']), f(b['poses']), f(b['K']), b['ref_nos'], False,
                                         n_mult=n_mult, dir1=dir1, jitter=f(jitter), dir2=dir2, raw_noise=f(noise))
            loss.backward()
        finally:
            torch.set_default_dtype(old)
        return float(loss.detach()), o, [(W.grad, x.grad) for W, x in layers]
    l64, o64, g64 = oracle(torch.float64)
    _, _, g32 = oracle(torch.float32)
    layers = orc.trainer_layers(b['w'])
    tr = ops.Trainer([W for W, _ in layers], [x for _, x in layers], max_rays=N, device=dev, max_samples=S)
    img4 = ops.images_pack(cu(b['images'], dev))
    L, rgb = tr.explore_fwd_bwd(cu(b['rays'], dev), cu(b['or_rays'], dev), cu(b['target'], dev), img4, cu(b['poses'], dev), cu(b['K'], dev),
                                b['ref_nos'].to(dev).contiguous(), n_mult=n_mult, dir1=dir1, jitter=cu(jitter, dev), dir2=dir2, raw_noise=cu(noise, dev))
    assert abs(float(L[0]) - l64) < 2e-5 * max(1.0, l64)
    assert orc.psnr(rgb.cpu(), o64['rgb_map1'].detach().float()) > 70.0
    ratios = []
    for li in range(14, 26):                                    # NeRF layers only: the sampler / refine nets get no gradient on these iterations
        gW, gb = tr.read('grad', li)
        eW, eb = rel(gW, g64[li][0]), rel(gb, g64[li][1])
        cW, cb = rel(g32[li][0], g64[li][0]), rel(g32[li][1], g64[li][1])
        assert eW < 4e-2 and eb < 4e-2, (li, eW, eb)
        assert eW < 6 * cW + 1e-4 and eb < 6 * cb + 1e-4, (li, eW, cW, eb, cb)
        ratios += [eW / (cW + 1e-6), eb / (cb + 1e-6)]
    assert float(np.median(ratios)) < 3.0, sorted(ratios)[-6:]
    # the operator-level compositing backward at this S
    raw = torch.from_numpy(rs.randn(64, S, 4).astype(np.float32)); z = torch.sort(torch.from_numpy(rs.rand(64, S).astype(np.float32)), -1)[0]
    d = torch.from_numpy(rs.randn(64, 3).astype(np.float32)); g = torch.from_numpy(rs.randn(64, 3).astype(np.float32))
    rawd, zd = raw.double().requires_grad_(), z.double().requires_grad_()
    (orc.raw2outputs(rawd, zd, d.double(), clamp=10.0)[0] * g.double()).sum().backward()
    d_raw, d_z, _, _ = ops.composite_bwd(cu(raw, dev), cu(z, dev), cu(d, dev), cu(g, dev), clamp=10.0)
    assert rel(d_raw, rawd.grad) < 1e-4 and rel(d_z, zd.grad) < 1e-4


def test_stage2_train_driver_end_to_end(dev, tmp_path):
    """train() of the stage-2 mirror on an LLFF directory: stage-1 checkpoint in, a few iterations, checkpoint with the
    reference's keys out — which the inference driver then loads and renders."""
    import llff_synth
    from pronerf_amd import run_S_eS_eN_alter_base_refine2 as s2
    from pronerf_amd import run_S_eS_eN_alter_trt as trt
    root = llff_synth.make_dataset(str(tmp_path / 'scene'), seed=2, n=10, H=24, W=32, factor=4)
    w = synth.make_weights(0, 'trained'); wc = synth.make_nerfcls_weights(0, head_scale=0.3)
    sds = synth.state_dicts(w)
    pre = str(tmp_path / 'stage1.tar')
    torch.save({'global_step': 7, 'network_fn_state_dict': synth.nerfcls_state_dict(wc), 'mmr_network_fn_state_dict': sds['sampler'],
                'refine_net_state_dict': sds['refine']}, pre)
    cfg = tmp_path / 'refine.txt'
    cfg.write_text(f'expname = s2\nbasedir = {tmp_path}/logs\ndatadir = {root}\npretrain_path = {pre}\nfactor = 4\nllffhold = 8\nN_rand = 512\nN_samples = 8\n'
                   'N_point_ray_enc = 48\nmmnetdepth = 6\nmmnetskips = [10000]\nnum_neighbor = 4\nuse_viewdirs = True\nraw_noise_std = 1e0\nlrate = 5e-4\n'
                   'weight_decay = 5e-8\ni_print = 5\ni_weights = 1000\n')
    import random
    random.seed(11); np.random.seed(11); torch.manual_seed(11); torch.cuda.manual_seed_all(11)          # the driver's per-batch draws
    tr, log = s2.train(['--config', str(cfg), '--max_steps', '40'], device=dev)
    assert [e[0] for e in log] == [5, 10, 15, 20, 25, 30, 35, 40] and all(np.isfinite(e[1]) for e in log)
    # the logged loss is that of each iteration's own random 512-ray batch with sigma noise — a noisy series; descent is asserted on ONE fixed
    # evaluation batch (all 768 rays of training view 0, fixed draws, no sigma noise) with the parameters before and after the 40 iterations
    from pronerf_amd import ops
    from pronerf_amd.load_llff import load_llff_data
    images, poses, _, _, _ = load_llff_data(root, 4, recenter=True, bd_factor=.75, spherify=False)
    i_train = np.array([i for i in range(images.shape[0]) if i % 8 != 0])
    Hh, Ww, focal = int(poses[0, 0, -1]), int(poses[0, 1, -1]), float(poses[0, 2, -1])
    Kk = np.array([[focal, 0, 0.5 * Ww], [0, focal, 0.5 * Hh], [0, 0, 1]], dtype=np.float32)
    with torch.cuda.device(dev):
        er, eo = ops.frame_rays(Kk, poses[i_train[0], :3, :4], Hh, Ww, near=0., far=1., device=dev)
        img4, pz, Kt, rank = s2._train_views(images[i_train], poses[i_train, :3, :4], Kk, dev)
    et = torch.as_tensor(images[i_train[0]], dtype=torch.float32).reshape(-1, 3).to(dev)
    eref = rank[0][1:5][None].expand(er.shape[0], -1).contiguous()
    tr0 = ops.Trainer(*zip(*s2.trainer_layer_list(sds['sampler'], sds['refine'], synth.nerfcls_state_dict(wc))), max_rays=er.shape[0], device=dev)
    before = float(tr0.fwd_bwd(er, eo, et, img4, pz, Kt, eref, want_rgb=False)[0][1])
    tr_eval = ops.Trainer(*zip(*[tuple(t.cpu().numpy() for t in tr.read('param', i)) for i in range(26)]), max_rays=er.shape[0], device=dev)
    after = float(tr_eval.fwd_bwd(er, eo, et, img4, pz, Kt, eref, want_rgb=False)[0][1])
    print(f'\n[driver] image loss on the fixed evaluation batch: {before:.5f} before, {after:.5f} after 40 iterations; logged batch losses {[round(e[1], 4) for e in log]}')
    assert after < 0.97 * before and max(e[1] for e in log) < 1.0
    ck_path = tmp_path / 'logs' / 's2' / '000040.tar'
    ck = torch.load(str(ck_path), map_location='cpu')
    assert sorted(ck['network_fine_state_dict']) == sorted(synth.nerfcls_state_dict(wc))
    assert sorted(ck['mmr_network_fn_state_dict']) == sorted(sds['sampler']) and ck['global_step'] == 40
    moved = float((ck['refine_net_state_dict']['fc_output.weight'] - sds['refine']['fc_output.weight']).abs().max())
    assert 0 < moved <= 40 * 5e-4 * 1.01                              # Adam moves a weight by at most lr per step
    icfg = tmp_path / 'infer.txt'
    icfg.write_text(f'expname = inf\nbasedir = {tmp_path}/logs\ndatadir = {root}\nft_path = {ck_path}\nfactor = 4\nllffhold = 8\nN_samples = 8\n'
                    'N_point_ray_enc = 48\nmmnetdepth = 6\nmmnetskips = [10000]\nnum_neighbor = 4\nuse_viewdirs = True\n')
    kw = trt.train(['--config', str(icfg), '--render_test', '--max_images', '1'], device=dev)
    assert len(kw['psnrs']) == 1 and np.isfinite(kw['psnrs'][0])


@pytest.mark.parametrize('name', ['stage1_step_joint_12x16', 'stage1_step_explore_10x14'])
def test_stage1_iterations_vs_reference_golden(dev, golden_dir, name):
    """Stage-1 alternation on the HIP trainer against the reference's own iterations (run_S_eS_eN_alter_base.py:929-958): even
    = joint step (sample-major epi, eps 1e-6, clamp 10, three MSE terms, joint Adam); odd = exploration to 8*n_mult samples,
    NeRF-only gradients, the NeRF-only Adam.  Same criteria as the stage-2 golden test."""
    import train_golden_util as U
    from pronerf_amd import ops
    g64, b64 = U.load_case(golden_dir, name + '_f64')
    _, _, _, l32 = U.oracle_grads(b64, torch.float32)
    st = int(g64['stride'])
    joint = bool(b64['train_sampler'])
    active = range(26) if joint else range(14, 26)
    for gname, (g, b) in (('f64', (g64, b64)), ('f32', U.load_case(golden_dir, name))):
        layers = orc.trainer_layers(b['w'])
        tr = ops.Trainer([W for W, _ in layers], [x for _, x in layers], max_rays=b['N'], device=dev, max_samples=8 if joint else 8 * b['n_mult'])
        img4 = ops.images_pack(cu(b['images'], dev))
        args = (cu(b['rays'], dev), cu(b['or_rays'], dev), cu(b['target'], dev), img4, cu(b['poses'], dev), cu(b['K'], dev), b['ref_nos'].to(dev).contiguous())
        if joint:
            L, rgb = tr.fwd_bwd(*args, eps=1e-6, a_mmrgb=1.0, clamp=10.0, layout=1)
        else:
            for li in range(14):                      # must stay untouched by the NeRF-only backward
                tr.write('grad', li, torch.full(layers[li][0].shape, 7.0).to(dev), torch.full(layers[li][1].shape, 7.0).to(dev))
            L, rgb = tr.explore_fwd_bwd(*args, n_mult=b['n_mult'], dir1=b['dir1'], jitter=cu(b['jitter'], dev), dir2=b['dir2'], raw_noise=cu(b['noise'], dev))
        Lh = L.cpu().numpy()
        assert abs(Lh[0] - float(g['loss'])) < 3e-5 and abs(Lh[1] - float(g['img_loss'])) < 3e-5
        np.testing.assert_allclose(rgb.cpu().numpy(), g['rgb_map1'], rtol=0, atol=2e-4)
        grads = [tr.read('grad', i) for i in range(26)]
        if not joint:
            assert all(float(grads[i][0].min()) == 7.0 and float(grads[i][1].max()) == 7.0 for i in range(14))
        if gname == 'f64':
            ratios = []
            for i in active:
                for k, (mine, cpu32) in enumerate(((grads[i][0].reshape(-1)[::st], l32[i][0].grad.reshape(-1)[::st]), (grads[i][1], l32[i][1].grad))):
                    want = g[('gW_%d' if k == 0 else 'gb_%d') % i]
                    e, n = U.rel(mine, want), U.rel(cpu32, want)
                    assert e < 6 * n + 1e-3 and e < 0.1, (i, k, e, n)
                    ratios.append(e / (n + 1e-7))
            assert float(np.median(ratios)) < 3.0, float(np.median(ratios))
        else:
            before = [tr.read('param', i) for i in range(14)]
            tr.adam_step(b['lr'], weight_decay=b['wd'], nerf_only=not joint)
            after = [tr.read('param', i) for i in range(26)]
            U.check_against_golden(g, grads, after, tol_grad=U.noise_tolerances(g, g64, 7.0, 1e-3, layers=active), tol_norm=5e-2, layers=active)
            if not joint:                                 # the NeRF-only optimizer leaves the sampler / refine nets alone
                assert all(torch.equal(before[i][0], after[i][0]) and torch.equal(before[i][1], after[i][1]) for i in range(14))
                mW, _ = tr.read('m_nerf', 20); jW, _ = tr.read('m', 20)
                assert float(mW.abs().max()) > 0 and float(jW.abs().max()) == 0


def test_stage1_then_stage2_then_inference_chain(dev, tmp_path):
    """The reference's three-script workflow on an LLFF directory, all on the HIP path: stage-1 train() from scratch ->
    checkpoint -> stage-2 train() (--pretrain_path) -> checkpoint -> inference train() (--ft_path) renders a hold-out view."""
    import llff_synth
    from pronerf_amd import run_S_eS_eN_alter_base as s1
    from pronerf_amd import run_S_eS_eN_alter_base_refine2 as s2
    from pronerf_amd import run_S_eS_eN_alter_trt as trt
    root = llff_synth.make_dataset(str(tmp_path / 'scene'), seed=3, n=10, H=24, W=32, factor=4)
    common = (f'basedir = {tmp_path}/logs\ndatadir = {root}\nfactor = 4\nllffhold = 8\nN_rand = 512\nN_samples = 8\nN_point_ray_enc = 48\nmmnetdepth = 6\n'
              'mmnetskips = [10000]\nnum_neighbor = 4\nuse_viewdirs = True\nraw_noise_std = 1e0\nlrate = 5e-4\nweight_decay = 5e-8\ni_print = 4\ni_weights = 1000\n'
              'i_testset = 8\n')
    (tmp_path / 'epi.txt').write_text('expname = s1\n' + common)
    torch.manual_seed(0)
    tr1, log1 = s1.train(['--config', str(tmp_path / 'epi.txt'), '--max_steps', '24'], device=dev)
    tests1 = [e for e in log1 if e[1] == 'test_psnr']
    log1 = [e for e in log1 if e[1] != 'test_psnr']
    assert [e[0] for e in tests1] == [8, 16, 24] and all(np.isfinite(e[2]) for e in tests1)          # i_testset renders of the 2 hold-out views
    assert sorted(os.listdir(tmp_path / 'logs' / 's1' / 'testset_000016')) == ['000.png', '001.png']
    assert [e[0] for e in log1] == [4, 8, 12, 16, 20, 24] and all(np.isfinite(e[1]) for e in log1)
    assert log1[-1][1] < log1[0][1]                                   # even iterations' joint loss goes down from the random initialisation
    ck1 = tmp_path / 'logs' / 's1' / '000024.tar'
    c = torch.load(str(ck1), map_location='cpu')
    assert {'global_step', 'network_fn_state_dict', 'mmr_network_fn_state_dict', 'refine_net_state_dict'} <= set(c)
    assert float(c['pnrf_adam_m_nerf'][20][0].abs().max()) > 0 and float(c['pnrf_adam_m'][3][0].abs().max()) > 0     # both optimizers stepped
    (tmp_path / 'refine.txt').write_text(f'expname = s2\npretrain_path = {ck1}\n' + common)
    tr2, log2 = s2.train(['--config', str(tmp_path / 'refine.txt'), '--max_steps', '8'], device=dev)
    assert [e[0] for e in log2 if e[1] == 'test_psnr'] == [8] and os.path.exists(tmp_path / 'logs' / 's2' / 'testset_000008' / '001.png')
    ck2 = tmp_path / 'logs' / 's2' / '000008.tar'
    (tmp_path / 'infer.txt').write_text(f'expname = inf\nft_path = {ck2}\n' + common)
    kw = trt.train(['--config', str(tmp_path / 'infer.txt'), '--render_test', '--max_images', '1'], device=dev)
    assert len(kw['psnrs']) == 1 and np.isfinite(kw['psnrs'][0])


def test_data_parallel_gradients_match_single_process(dev, tmp_path):
    """Two replicas (processes), each on half of the batch, gradients averaged through pronerf_amd.dist.allreduce_gradients on the
    trainer's flat array: same gradient and same parameters after Adam as one process on the whole batch."""
    import subprocess
    import sys
    from pronerf_amd import ops
    out = str(tmp_path / 'ddp.npz')
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', '29547', os.path.join(os.path.dirname(__file__), 'ddp_worker.py'), out], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    d = np.load(out)
    assert d['same'].all()
    b = _batch(0, 12, 16, 7)
    layers = orc.trainer_layers(b['w'])
    tr = ops.Trainer([W for W, _ in layers], [x for _, x in layers], max_rays=b['N'], device=dev)
    img4 = ops.images_pack(cu(b['images'], dev))
    tr.fwd_bwd(cu(b['rays'], dev), cu(b['or_rays'], dev), cu(b['target'], dev), img4, cu(b['poses'], dev), cu(b['K'], dev), b['ref_nos'].to(dev).contiguous(),
               jitter=cu(b['jitter'], dev), jitter_dir=1, raw_noise=cu(b['noise'], dev), want_rgb=False)
    g1 = tr.flat('grad').cpu()
    assert g1.shape[0] == d['grad'].shape[0] and rel(d['grad'], g1) < 2e-4          # summation order differs (two halves vs one batch)
    tr.adam_step(5e-4, weight_decay=5e-8)
    np.testing.assert_allclose(d['param'], tr.flat('param').cpu().numpy(), rtol=0, atol=2.1 * 5e-4)
    assert float((np.abs(d['param'] - tr.flat('param').cpu().numpy()) > 1e-5).mean()) < 0.02
    # the flat view is live: it sees what read() sees
    W0, _ = tr.read('grad', 0)
    assert torch.equal(tr.flat('grad')[:W0.numel()].cpu(), W0.reshape(-1).cpu())


def test_data_parallel_training_driver(dev, tmp_path):
    """The stage-2 driver under torchrun with two replicas: identical parameters on both ranks after 6 iterations, one checkpoint."""
    import subprocess
    import sys
    import llff_synth
    root = llff_synth.make_dataset(str(tmp_path / 'scene'), seed=4, n=10, H=24, W=32, factor=4)
    w = synth.make_weights(0, 'trained'); wc = synth.make_nerfcls_weights(0, head_scale=0.3)
    sds = synth.state_dicts(w)
    pre = str(tmp_path / 'stage1.tar')
    torch.save({'global_step': 0, 'network_fn_state_dict': synth.nerfcls_state_dict(wc), 'mmr_network_fn_state_dict': sds['sampler'],
                'refine_net_state_dict': sds['refine']}, pre)
    cfg = tmp_path / 'refine.txt'
    cfg.write_text(f'expname = ddp\nbasedir = {tmp_path}/logs\ndatadir = {root}\npretrain_path = {pre}\nfactor = 4\nllffhold = 8\nN_rand = 512\nN_samples = 8\n'
                   'N_point_ray_enc = 48\nmmnetdepth = 6\nmmnetskips = [10000]\nnum_neighbor = 4\nuse_viewdirs = True\nraw_noise_std = 1e0\nlrate = 5e-4\n'
                   'weight_decay = 5e-8\ni_print = 3\ni_weights = 1000\n')
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', PNRF_DIST_BACKEND='gloo')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', '29549', os.path.join(os.path.dirname(__file__), 'ddp_train_worker.py'), str(cfg), str(tmp_path)],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, '\n'.join(l for l in r.stderr.splitlines() if 'Warning' not in l)[-6000:]
    a, b = np.load(tmp_path / 'rank0.npz'), np.load(tmp_path / 'rank1.npz')
    assert np.array_equal(a['param'], b['param']) and np.isfinite(a['loss']).all()
    assert sorted(os.listdir(tmp_path / 'logs' / 'ddp')) == ['000006.tar', 'args.txt']


def test_resume_and_reference_optimizer_state_import(dev, tmp_path):
    """(i) A reference-style checkpoint with torch.optim.Adam state dicts (stage-2 group order [fine, sampler, refine], the
    modules' own parameter order) is imported into the trainer's moments / step counts; (ii) the stage-2 driver resumes from the
    newest .tar of its experiment directory with moments and step counts restored."""
    import llff_synth
    from pronerf_amd import ops
    from pronerf_amd import run_nerf_helpers as h
    from pronerf_amd import run_S_eS_eN_alter_base_refine2 as s2
    # ---- (i)
    fine = h.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4, skips=[4], use_viewdirs=True)
    smp = h.MinMaxRay_Net(D=6, W=256, input_ch=288, output_ch=27, skips=[10000])
    rfn = h.MinMaxRay_Net(D=6, W=256, input_ch=144, output_ch=35, skips=[10000])
    opt = torch.optim.Adam([{'params': fine.parameters()}, {'params': smp.parameters()}, {'params': rfn.parameters()}], lr=5e-4)
    g = torch.Generator().manual_seed(0)
    for p in [q for grp in opt.param_groups for q in grp['params']]:
        p.grad = torch.randn(p.shape, generator=g) * 1e-3
    opt.step(); opt.step()
    ck = {'global_step': 2, 'network_fine_state_dict': fine.state_dict(), 'network_fn_state_dict': fine.state_dict(), 'mmr_network_fn_state_dict': smp.state_dict(),
          'refine_net_state_dict': rfn.state_dict(), 'optimizer_state_dict': opt.state_dict()}
    layers = s2.trainer_layer_list(ck['mmr_network_fn_state_dict'], ck['refine_net_state_dict'], ck['network_fine_state_dict'])
    tr = ops.Trainer(*zip(*layers), max_rays=8, device=dev)
    s2.restore_optimizer(tr, ck, 2)
    for mod, li in ((fine.rgb_linear, 25), (fine.views_linears[0], 24), (fine.alpha_linear, 23), (fine.pts_linears[5], 19), (smp.fc_output, 6), (rfn.fc_backbone[2], 9)):
        mW, mb = tr.read('m', li); vW, vb = tr.read('v', li)
        assert torch.equal(mW.cpu(), opt.state[mod.weight]['exp_avg']) and torch.equal(vb.cpu(), opt.state[mod.bias]['exp_avg_sq'])
        assert torch.equal(tr.read('param', li)[0].cpu(), mod.weight.detach())
    # the next Adam step continues torch's bias correction (step 3)
    for p in [q for grp in opt.param_groups for q in grp['params']]:
        p.grad = torch.full(p.shape, 2e-3)
    for li, (W, b) in enumerate(layers):
        tr.write('grad', li, torch.full(W.shape, 2e-3).to(dev), torch.full(b.shape, 2e-3).to(dev))
    opt.step(); tr.adam_step(5e-4)
    np.testing.assert_allclose(tr.read('param', 24)[0].cpu().numpy(), fine.views_linears[0].weight.detach().numpy(), rtol=0, atol=2e-7)
    # ---- (ii)
    root = llff_synth.make_dataset(str(tmp_path / 'scene'), seed=5, n=10, H=24, W=32, factor=4)
    pre = str(tmp_path / 'stage1.tar')
    torch.save(ck, pre)
    cfg = tmp_path / 'refine.txt'
    cfg.write_text(f'expname = rs\nbasedir = {tmp_path}/logs\ndatadir = {root}\npretrain_path = {pre}\nfactor = 4\nllffhold = 8\nN_rand = 256\nN_samples = 8\n'
                   'N_point_ray_enc = 48\nmmnetdepth = 6\nmmnetskips = [10000]\nnum_neighbor = 4\nuse_viewdirs = True\nraw_noise_std = 1e0\nlrate = 5e-4\n'
                   'weight_decay = 5e-8\ni_print = 100\ni_weights = 100\n')
    s2.train(['--config', str(cfg), '--max_steps', '6'], device=dev)
    c1 = torch.load(str(tmp_path / 'logs' / 'rs' / '000006.tar'), map_location='cpu')
    assert c1['global_step'] == 6 and tuple(c1['pnrf_adam_steps']) == (6, 0)
    tr2, _ = s2.train(['--config', str(cfg), '--max_steps', '3'], device=dev)            # resumes from 000006.tar
    assert sorted(f for f in os.listdir(tmp_path / 'logs' / 'rs') if f.endswith('.tar')) == ['000006.tar', '000009.tar']
    c2 = torch.load(str(tmp_path / 'logs' / 'rs' / '000009.tar'), map_location='cpu')
    assert c2['global_step'] == 9 and tuple(c2['pnrf_adam_steps']) == (9, 0)
    # the moments were carried over, not restarted: after 3 more steps v >= 0.999^3 * v_saved
    v6, v9 = c1['pnrf_adam_v'][20][0], c2['pnrf_adam_v'][20][0]
    assert float(v6.max()) > 0 and bool((v9 >= 0.996 * v6).all())
    tr3, _ = s2.train(['--config', str(cfg), '--max_steps', '1', '--no_reload'], device=dev)     # --no_reload starts again from the stage-1 weights
    assert torch.load(str(tmp_path / 'logs' / 'rs' / '000001.tar'), map_location='cpu')['global_step'] == 1


def test_weight_gradient_kernels_agree(dev, tmp_path):
    """dW = dY^T X has two split-K MFMA kernels (64 x 64 tiles; 128 x 128 tiles for the square hidden layers on many rows).  The
    library picks per call by shape and row count, so the same batch is run in two processes, once with each kernel forced for the
    256 x 256 layers: the gradients may differ by fp32 summation order only."""
    import subprocess
    worker = os.path.join(os.path.dirname(__file__), 'dw_tile_worker.py')
    outs = []
    for name, tile in (('t64', '64'), ('t128', '128')):
        out = str(tmp_path / f'{name}.npz')
        r = subprocess.run([sys.executable, worker, out, tile], capture_output=True, text=True, timeout=240)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(dict(np.load(out)))
    a, b = outs
    for o in outs:
        o.pop('group_info'); o.pop('rows')
    assert sorted(a) == sorted(b) and len(a) == 52
    differ = 0
    for k in a:
        scale = np.abs(a[k]).max() + 1e-30
        assert np.abs(a[k] - b[k]).max() <= 2e-5 * scale, (k, np.abs(a[k] - b[k]).max() / scale)
        differ += int(not np.array_equal(a[k], b[k]))
    assert differ >= 4          # (some of) the seven square NeRF layers, 3072 rows each, did go through the other kernel


def test_wide_weight_gradient_tiles_agree_with_the_square_ones_and_with_fp64(dev, tmp_path):
    """The grouped split-fp16 weight gradients on 256 x 128 tiles (dwh_body_wide — the default from 32 768 rows on: the stage-2 iteration and both
    exploration workloads) against the same launch on 128 x 128 tiles and against the fp64 oracle, on a batch that reaches the grouped launch
    (>= 8192 rows), with a row count that is not a multiple of 32 (10 296) and the skip layer's 319 input columns (not a multiple of 128).
    The wide form keeps the square form's split count, so every partial sum has the same order: BIT-identical gradients.  The worker records which
    jobs of the grouped launch ran wide: all ten 256-row gradients (pts1..7 incl. the skip layer, feature) with tile 256, none with 255."""
    import subprocess
    worker = os.path.join(os.path.dirname(__file__), 'dw_tile_worker.py')
    outs = {}
    for tile in ('256', '255'):
        out = str(tmp_path / f't{tile}.npz')
        r = subprocess.run([sys.executable, worker, out, tile], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[tile] = dict(np.load(out))
    wide, sq = outs['256'], outs['255']
    assert int(wide['rows']) == 10296 and int(wide['rows']) % 32 != 0
    n_w, m_w = (int(v) for v in wide['group_info'])
    n_s, m_s = (int(v) for v in sq['group_info'])
    assert n_w == n_s and n_w >= 9, (n_w, n_s)
    assert m_s == 0 and bin(m_w).count('1') >= 8, (bin(m_w), bin(m_s))       # the wide bit really was set (and really was not)
    keys = [k for k in wide if k[0] in 'Wb' and k[1:].isdigit()]
    assert len(keys) == 52
    for k in keys:
        assert np.array_equal(wide[k], sq[k]), (k, np.abs(wide[k] - sq[k]).max())
    # ... and the pair against the fp64 oracle, held to the fp32 CPU run's own distance from it (the bound of the 12 x 16 test above)
    b = _batch(0, 33, 39, 7)
    _, _, _, g64 = _oracle_grads(b, 1, False, 0.0, torch.float64)
    _, _, _, g32 = _oracle_grads(b, 1, False, 0.0, torch.float32)
    cmax = max(max(rel(g32[li][0], g64[li][0]), rel(g32[li][1], g64[li][1])) for li in range(26))
    ratios = []
    for li in range(26):
        eW, eb = rel(torch.from_numpy(wide[f'W{li}']), g64[li][0]), rel(torch.from_numpy(wide[f'b{li}']), g64[li][1])
        cW, cb = rel(g32[li][0], g64[li][0]), rel(g32[li][1], g64[li][1])
        assert eW < 2.5 * cmax and eb < 2.5 * cmax, (li, eW, eb, cmax)
        ratios += [eW / (cW + 1e-5), eb / (cb + 1e-5)]
    assert float(np.median(ratios)) < 2.0, sorted(ratios)[-8:]
