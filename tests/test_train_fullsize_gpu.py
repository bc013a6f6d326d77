"""GPU: the training iterations at the sizes BASELINE.json names — configs[3] (stage-2 iteration: 4096 rays of one view, 17 training views of
756 x 1008, NeRF-class fine net) and configs[4] (stage-1 exploration iteration: 4096 rays x 256 samples per ray = 1 048 576 rows through the
fine net) — one step each through the C ABI trainer (pronerf_amd.workloads.TrainWorkload, the batch bench.py's `train` block times).

Per-ray outputs do not depend on the rest of the batch, so rgb_map1 of a 256-ray subset is compared with the CPU oracle's training-time
render_rays on exactly those rays (same source views, jitter and noise); the loss is tied to the image it is the mean square of; the two
product arithmetics (exact fp32 / split fp16), the three forms of the fine net's forward, graph replay against kernel-by-kernel launches, finiteness of every gradient and the size of
the trainer's device allocation are checked on the full batch.  Small-size gradients against torch.autograd: tests/test_train_gpu.py.
"""
import numpy as np
import pytest
import torch

from oracle import pronerf_oracle as orc

pytestmark = pytest.mark.gpu

SUB = 256


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    from pronerf_amd import _lib
    _lib.load()
    return torch.device('cuda:0')


@pytest.fixture(scope='module')
def work(dev):
    from pronerf_amd import workloads as wl
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(dev)[0]
    wk = wl.TrainWorkload(dev, max_samples=256)
    torch.cuda.synchronize()
    used = free0 - torch.cuda.mem_get_info(dev)[0]
    print(f'\n[full size] TrainWorkload(4096 rays, 17 views 756x1008, max 256 samples per ray): {used / 2**30:.2f} GiB of device memory '
          f'(trainer parameters / gradients / Adam state / workspaces for 1 048 576 NeRF rows + 17 packed views + the batch)')
    assert wk.n == 4096 and wk.nv == 17 and tuple(wk.img4.shape) == (17, 756, 1008, 4)
    return wk


def _subset(wk):
    sel = torch.linspace(0, wk.n - 1, SUB).long()
    cpu = lambda t: t[sel.to(t.device)].cpu()
    layers = [(torch.from_numpy(W), torch.from_numpy(b)) for W, b in wk.layers]
    return sel, cpu, orc.weights_from_layers(layers)


def _grads(tr, layers=range(26)):
    return [g.clone() for i in layers for g in tr.read('grad', i)]


def _rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def _forward_variants(wk, tr, step, layers, loss, rgb, g16):
    """The fine net's forward pass three ways on the same split-fp16 arithmetic: one launch on the fused-MLP engine (tchain_fwd_kernel, the default:
    128 rows per workgroup in registers through pts0 .. feature), one product launch per layer, and the 64-row layer chains (hgemm_wchain_kernel).
    The last two run the same products in the same order: bit for bit.  The engine contracts in its own order: fp32 round-off apart (measured:
    image 1e-7, gradients 4e-7 .. 2e-4 relative — the ill-conditioned tensors of test_train_gpu.py)."""
    tr.set_products('f16x2_unchained')
    lossu, rgbu = step()
    gu = _grads(tr, layers)
    tr.set_products('f16x2_wchain')
    lossw, rgbw = step()
    gw = _grads(tr, layers)
    tr.set_products('f16x2')
    assert torch.equal(lossu, lossw) and torch.equal(rgbu, rgbw) and all(torch.equal(a, b) for a, b in zip(gu, gw))
    assert abs(float(lossu[1]) - float(loss[1])) < 1e-6 * max(1.0, float(loss[1])) and _rel(rgb, rgbu) < 2e-6
    worst = max(_rel(a, b) for a, b in zip(g16, gu))
    print(f'[full size] forward on the engine vs one launch per layer: image {_rel(rgb, rgbu):.1e}, worst gradient tensor {worst:.1e} relative')
    assert worst < 2e-3


def test_stage2_iteration_at_config3_size(work, dev):
    wk, tr = work, work.trainer
    tr.set_products('f16x2'); tr.set_graph(False)
    loss, rgb = wk.stage2_step(want_rgb=True, adam=False)
    g16 = _grads(tr)
    L = loss.cpu().numpy()
    assert np.isfinite(L).all() and bool(torch.isfinite(rgb).all()) and all(bool(torch.isfinite(g).all()) for g in g16)
    # the loss is the mean square of the image that came with it (img2mse, refine2.py:860)
    mse = float(((rgb.double() - wk.target.double()) ** 2).mean())
    assert abs(L[1] - mse) < 2e-6 * max(1.0, mse) and abs(L[0] - L[1]) < 1e-12                       # a_mmrgb = 0: total = image loss
    # 256 rays of the batch against the oracle's stage-2 render_rays on the host
    sel, cpu, w = _subset(wk)
    with torch.no_grad():
        o = orc.render_rays_stage2(w, cpu(wk.rays), cpu(wk.or_rays), wk.images_nchw.cpu(), wk.poses.cpu(), wk.K.cpu(), cpu(wk.ref_nos), jitter=cpu(wk.jitter),
                                   jitter_dir=1, raw_noise=cpu(wk.noise))
    tie_free = (o['depth_sorted'][:, 1:] - o['depth_sorted'][:, :-1]).min(1)[0] > 1e-6
    ps = orc.psnr(cpu(rgb)[tie_free], o['rgb_map1'][tie_free])
    print(f'\n[full size] stage-2 iteration: loss {L[1]:.6f}; rgb_map1 of {int(tie_free.sum())} subset rays vs the oracle {ps:.1f} dB')
    assert int(tie_free.sum()) >= SUB - 4 and ps > 80.0                                              # fp32-grade path
    # exact-fp32 products on the same batch: same loss and image, gradients as close as two fp32 summation orders are on this ill-conditioned
    # chain (test_train_gpu.py: torch's own fp32 run is 3e-3 .. 5e-2 from fp64 per tensor; measured here 1.5e-2 .. 2.1e-2)
    tr.set_products('f32')
    loss32, rgb32 = wk.stage2_step(want_rgb=True, adam=False)
    g32 = _grads(tr)
    tr.set_products('f16x2')
    L32 = loss32.cpu().numpy()
    assert abs(L32[1] - L[1]) < 2e-6 * max(1.0, L[1]) and orc.psnr(rgb.cpu(), rgb32.cpu()) > 90.0
    worst = max(_rel(a, b) for a, b in zip(g16, g32))
    print(f'[full size] split-fp16 vs exact-fp32 products: worst relative gradient difference over the 52 tensors {worst:.2e}')
    assert worst < 6e-2
    # the iteration replayed as a hipGraph: bit for bit
    tr.set_graph(True)
    for _ in range(2):                                     # capture, then replay
        lossg, rgbg = wk.stage2_step(want_rgb=True, adam=False)
    gg = _grads(tr)
    tr.set_graph(False)
    assert torch.equal(lossg, loss) and torch.equal(rgbg, rgb) and all(torch.equal(a, b) for a, b in zip(gg, g16))
    _forward_variants(wk, tr, lambda: wk.stage2_step(want_rgb=True, adam=False), range(26), loss, rgb, g16)


def test_optimizer_steps_under_graph_replay_equal_kernel_by_kernel(work, dev):
    """Iterations WITH their Adam steps, replayed as hipGraphs: every step refreshes the derived weight forms (one grouped launch in front of the captured
    body) and the backward chain's column norms alternate between two arrays, which is part of the graph key — the losses of four consecutive steps
    must equal, bit for bit, those of the same steps launched kernel by kernel from the same parameters and moments."""
    wk, tr = work, work.trainer
    tr.set_products('f16x2')
    keep = {k: tr.flat(k).clone() for k in ('param', 'm', 'v')}
    W0, b0 = tr.read('param', 0)

    def restore():
        for k, v in keep.items():
            tr.flat(k).copy_(v)
        tr.write('param', 0, W0, b0)                        # the write marks every derived form stale (the flat views do not)
        tr.set_step(0, 0)

    def run(graph):
        restore()
        tr.set_graph(graph)
        out = []
        for _ in range(4):
            loss, _ = wk.stage2_step(adam=True)
            out.append(loss.clone())
        params = tr.flat('param').clone()
        tr.set_graph(False)
        return out, params

    try:
        a, pa = run(False)
        b, pb = run(True)
    finally:
        restore()
    assert all(bool(torch.isfinite(x).all()) for x in a) and float(a[0][0]) != float(a[3][0])            # the steps do change the loss
    assert all(torch.equal(x, y) for x, y in zip(a, b)) and torch.equal(pa, pb)


def test_exploration_iteration_at_config4_size(work, dev):
    """4096 rays x 256 samples per ray (n_mult = 32): 1 048 576 rows through the 12 NeRF layers, forward and backward."""
    wk, tr = work, work.trainer
    n_mult = 32
    tr.set_products('f16x2'); tr.set_graph(False)
    loss, rgb = wk.explore_step(n_mult, want_rgb=True, adam=False)
    g16 = _grads(tr, range(14, 26))
    L = loss.cpu().numpy()
    assert np.isfinite(L).all() and bool(torch.isfinite(rgb).all()) and all(bool(torch.isfinite(g).all()) for g in g16)
    mse = float(((rgb.double() - wk.target.double()) ** 2).mean())
    assert abs(L[1] - mse) < 2e-6 * max(1.0, mse)
    sel, cpu, w = _subset(wk)
    with torch.no_grad():
        o = orc.render_rays_stage1(w, cpu(wk.rays), cpu(wk.or_rays), wk.images_nchw.cpu(), wk.poses.cpu(), wk.K.cpu(), cpu(wk.ref_nos), False, n_mult=n_mult,
                                   dir1=1, jitter=cpu(wk.explore_jitter(n_mult)), dir2=-1)
    tie_free = (o['depth_sorted'][:, 1:] - o['depth_sorted'][:, :-1]).min(1)[0] > 1e-6
    ps = orc.psnr(cpu(rgb)[tie_free], o['rgb_map1'][tie_free])
    print(f'\n[full size] exploration iteration at 256 samples per ray: loss {L[1]:.6f}; rgb_map1 of {int(tie_free.sum())} subset rays vs the oracle {ps:.1f} dB')
    assert int(tie_free.sum()) >= SUB - 4 and ps > 75.0                                              # 256-term compositing sums in fp32
    tr.set_products('f32')
    loss32, rgb32 = wk.explore_step(n_mult, want_rgb=True, adam=False)
    g32 = _grads(tr, range(14, 26))
    tr.set_products('f16x2')
    assert abs(float(loss32[1]) - L[1]) < 2e-6 * max(1.0, L[1]) and orc.psnr(rgb.cpu(), rgb32.cpu()) > 90.0
    worst = max(_rel(a, b) for a, b in zip(g16, g32))
    print(f'[full size] split-fp16 vs exact-fp32 products: worst relative gradient difference over the 24 NeRF tensors {worst:.2e}')
    assert worst < 6e-2
    tr.set_graph(True)
    for _ in range(2):
        lossg, rgbg = wk.explore_step(n_mult, want_rgb=True, adam=False)
    gg = _grads(tr, range(14, 26))
    tr.set_graph(False)
    assert torch.equal(lossg, loss) and torch.equal(rgbg, rgb) and all(torch.equal(a, b) for a, b in zip(gg, g16))
    _forward_variants(wk, tr, lambda: wk.explore_step(n_mult, want_rgb=True, adam=False), range(14, 26), loss, rgb, g16)
    # one optimizer step of each kind leaves finite parameters
    wk.stage2_step(); wk.explore_step(n_mult)
    assert all(bool(torch.isfinite(p).all()) for i in range(26) for p in tr.read('param', i))


@pytest.mark.parametrize('offset', [1e-3, 1.0, 300.0])
def test_backward_chain_gradient_range(dev, offset):
    """The backward chain carries gradients as per-row power-of-two-scaled fp16 planes (pnrf_tchain.h): image gradients of 1e-7 per pixel (a target 1e-3 from
    the rendered image: the products' own 1e-6 differences in the image are then 1e-3 of the gradient signal), of the usual size and of 0.1 (a target 300 away) have to come out fp32-grade — against the exact-fp32 products and
    against one launch per layer — with rows whose gradient is dominated by the alpha branch and rows without any.  8192 sample rows: the
    smallest batch that takes the engine path."""
    from pronerf_amd import workloads as wl
    wk = wl.TrainWorkload(dev, n_rays=1024, n_views=6, H=96, W=128, focal=110.0, max_samples=8, seed=3)
    tr = wk.trainer
    tr.set_products('f32')
    _, rgb = wk.stage2_step(want_rgb=True, adam=False)
    g = torch.Generator(device='cpu').manual_seed(7)
    delta = torch.randn(rgb.shape, generator=g).to(dev) * offset
    delta[::5] = 0                                                            # every fifth ray: no image gradient at all
    wk.target = (rgb + delta).contiguous()
    res = {}
    for kind in ('f32', 'f16x2', 'f16x2_unchained'):
        tr.set_products(kind)
        loss, _ = wk.stage2_step(want_rgb=True, adam=False)
        res[kind] = (float(loss[1]), _grads(tr))
    tr.set_products('f16x2')
    assert all(bool(torch.isfinite(x).all()) for x in res['f16x2'][1])
    assert abs(res['f16x2'][0] - res['f32'][0]) <= 5e-3 * res['f32'][0]
    scale = max(float(x.abs().max()) for x in res['f32'][1])
    worst32 = max(_rel(a, b) for a, b in zip(res['f16x2'][1], res['f32'][1]))
    worstu = max(_rel(a, b) for a, b in zip(res['f16x2'][1], res['f16x2_unchained'][1]))
    base = max(_rel(a, b) for a, b in zip(res['f16x2_unchained'][1], res['f32'][1]))
    print(f'\n[gradient range] target offset {offset:g}: largest gradient entry {scale:.1e}; engine chains vs fp32 products {worst32:.1e} (one launch per layer vs '
          f'fp32: {base:.1e}), vs one launch per layer {worstu:.1e}')
    assert scale > 0 and worst32 < max(2e-2, 3 * base) and worstu < 1e-2
