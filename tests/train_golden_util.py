"""Shared by the CPU (oracle) and GPU (HIP trainer) tests against the reference's training-iteration goldens."""
import os

import numpy as np
import torch

from oracle import pronerf_oracle as orc
from oracle import synth

CASES = ['stage2_step_12x16', 'stage2_step_white_mmrgb_10x14']          # each also as <name>_f64: the reference run in float64


def load_case(golden_dir, name):
    g = dict(np.load(os.path.join(golden_dir, name + '.npz')))
    if name.startswith('stage1'):
        return g, _load_stage1(g)
    seed = int(g['seed'])
    scene = synth.make_scene(seed, H=int(g['H']), W=int(g['W']), n_views=int(g['nv']), sigma_t=0.2, rotate=True)
    w = synth.make_weights(seed, 'trained'); w['nerfcls'] = synth.make_nerfcls_weights(seed, head_scale=0.3)
    poses = torch.from_numpy(scene['poses']); images = torch.from_numpy(scene['images']).permute(0, 3, 1, 2).contiguous()
    rays, or_rays = torch.from_numpy(g['rays']), torch.from_numpy(g['or_rays'])
    N = rays.shape[0]
    ref_nos = orc.select_neighbors_train(poses[int(g['own'])][None].expand(N, -1, -1), poses, 4, g['order_idx'])
    b = dict(w=w, rays=rays, or_rays=or_rays, target=torch.from_numpy(g['target']), images=images, poses=poses, K=torch.from_numpy(scene['K']),
             ref_nos=ref_nos, jitter=torch.from_numpy(g['jitter']), noise=torch.from_numpy(g['raw_noise']), N=N, jdir=int(g['jitter_dir']),
             white=bool(g['white_bkgd']), a_mmrgb=float(g['a_mmrgb']), lr=float(g['lr']), wd=float(g['weight_decay']))
    return g, b


def _load_stage1(g):
    seed = int(g['seed'])
    scene = synth.make_scene(seed, H=int(g['H']), W=int(g['W']), n_views=int(g['nv']), sigma_t=0.2, rotate=True)
    w = synth.make_weights(seed, 'trained'); w['nerfcls'] = synth.make_nerfcls_weights(seed, head_scale=0.3)
    poses = torch.from_numpy(scene['poses']); images = torch.from_numpy(scene['images']).permute(0, 3, 1, 2).contiguous()
    rays, or_rays = torch.from_numpy(g['rays']), torch.from_numpy(g['or_rays'])
    N = rays.shape[0]
    ref_nos = orc.select_neighbors_train(poses[int(g['own'])][None].expand(N, -1, -1), poses, 4, g['order_idx'])
    ts = bool(g['train_sampler'])
    b = dict(w=w, rays=rays, or_rays=or_rays, target=torch.from_numpy(g['target']), images=images, poses=poses, K=torch.from_numpy(scene['K']),
             ref_nos=ref_nos, N=N, white=False, lr=float(g['lr']), wd=float(g['weight_decay']), stage=1, train_sampler=ts, jitter=None, noise=None)
    if not ts:
        b.update(n_mult=int(g['n_mult']), dir1=int(g['dir1']), dir2=int(g['dir2']), jitter=torch.from_numpy(g['jitter']), noise=torch.from_numpy(g['raw_noise']))
    return b


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu().reshape(-1), torch.as_tensor(b).double().cpu().reshape(-1)
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def oracle_grads(b, dtype):
    """loss + gradients of the oracle on the case's inputs in `dtype`."""
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        c = lambda v: v.to(dtype) if isinstance(v, torch.Tensor) and v.is_floating_point() else v
        layers = [(torch.tensor(W, dtype=dtype, requires_grad=True), torch.tensor(x, dtype=dtype, requires_grad=True)) for W, x in orc.trainer_layers(b['w'])]
        if b.get('stage', 2) == 1:
            loss, img_loss, o = orc.stage1_loss(layers, c(b['rays']), c(b['or_rays']), c(b['target']), c(b['images']), c(b['poses']), c(b['K']), b['ref_nos'],
                                                b['train_sampler'], n_mult=b.get('n_mult', 1), dir1=b.get('dir1', 1), jitter=c(b['jitter']), dir2=b.get('dir2', 1),
                                                raw_noise=c(b['noise']))
        else:
            loss, img_loss, o = orc.stage2_loss(layers, c(b['rays']), c(b['or_rays']), c(b['target']), c(b['images']), c(b['poses']), c(b['K']), b['ref_nos'],
                                                jitter=c(b['jitter']), jitter_dir=b['jdir'], raw_noise=c(b['noise']), white_bkgd=b['white'], a_mmrgb=b['a_mmrgb'])
        loss.backward()
    finally:
        torch.set_default_dtype(old)
    return float(loss.detach()), float(img_loss.detach()), o, layers


def noise_tolerances(g32, g64, factor, floor, layers=range(26)):
    """{(layer, 'W' | 'b'): factor x (distance of the fp32 golden's gradient tensor from the fp64 golden's) + floor}, capped at 0.1."""
    tol = {}
    for i in layers:
        for k in ('W', 'b'):
            tol[(i, k)] = min(0.1, factor * rel(torch.as_tensor(g32[f'g{k}_{i}']), g64[f'g{k}_{i}']) + floor)
    return tol


def check_against_golden(g, grads, params_after, tol_grad, tol_norm, layers=range(26)):
    """grads / params_after: 26 (W, b) pairs.  Subsampled gradient entries, gradient norms and post-Adam parameters.  tol_grad: one bound
    for all tensors or the dict of noise_tolerances()."""
    st = int(g['stride'])
    for i in layers:
        gW, gb = grads[i]
        tW, tb = (tol_grad[(i, 'W')], tol_grad[(i, 'b')]) if isinstance(tol_grad, dict) else (tol_grad, tol_grad)
        assert rel(gW.reshape(-1)[::st], g[f'gW_{i}']) < tW and rel(gb, g[f'gb_{i}']) < tb, (i, rel(gW.reshape(-1)[::st], g[f'gW_{i}']), tW, rel(gb, g[f'gb_{i}']), tb)
        assert abs(float(torch.as_tensor(gW).double().norm()) - float(g[f'gW_norm_{i}'])) < tol_norm * float(g[f'gW_norm_{i}'])
        if params_after is not None:
            pW, pb = params_after[i]
            # one Adam step moves a parameter by at most lr; entries whose gradient sign is at round-off level may go either way
            np.testing.assert_allclose(torch.as_tensor(pW).cpu().reshape(-1)[::st].numpy(), g[f'pW_{i}'], rtol=0, atol=2.1 * float(g['lr']))
            dW = torch.as_tensor(pW).cpu().reshape(-1)[::st].numpy() - g[f'pW_{i}']
            assert float((np.abs(dW) > 1e-5).mean()) < 0.05, (i, float((np.abs(dW) > 1e-5).mean()))
