#!/usr/bin/env python3
"""Optimizer-produced weights for the parity tests (VERDICT r3, next #2a): the package's own stage-1 driver from a random initialisation
(run_S_eS_eN_alter_base.train: alternating joint / exploration iterations) followed by its stage-2 driver from that checkpoint
(run_S_eS_eN_alter_base_refine2.train), both on the synthetic LLFF scene of tests/llff_synth.py, on the HIP trainer (Adam, 5e-4).  The 26
layers of the three nets after training -> tests/golden/trained_synth_scene.npz (fp32; what a reference checkpoint's state dicts hold), with
the logged losses.  Neither dataset nor checkpoint of the reference ships, so these are the only weights here that an
optimizer has shaped: heavy-tailed rows, correlated columns, biases moved off their initial scale.

    python tools/make_trained_fixture.py [--stage1 4000] [--stage2 3000] [--out tests/golden/trained_synth_scene.npz]

Round 6: ``--scene consistent`` trains on the geometrically consistent scene instead (tests/llff_synth.py ``Scene3D``: one 3-D scene ray-cast from
the rig, real COLMAP visibility) -> tests/golden/trained_scene3d.npz, and every run ends with the reference's own quality figure — PSNR of the
rendered hold-out views against their ground-truth pictures (run_S_eS_eN_alter_trt.py:351-353, 368-373) — through the HIP renderer:

    python tools/make_trained_fixture.py --scene consistent --stage1 20000 --stage2 20000"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import llff_synth   # noqa: E402
from pronerf_amd import run_S_eS_eN_alter_base as s1   # noqa: E402
from pronerf_amd import run_S_eS_eN_alter_base_refine2 as s2   # noqa: E402


SCENE_KW = {'pictures': {}, 'consistent': {'consistent': True, 'n_points': 3000}}       # tests rebuild the directory with the same arguments


def evaluate(root, fixture, device='cuda:0'):
    """PSNR against the ground-truth pictures of the hold-out views (every 8th) and of four training views, HIP renderer, neighbours chosen as
    the inference driver does (greedy COLMAP ranking, then the 4 nearest of them per pose)."""
    from pronerf_amd import load_llff as L, synthetic
    from pronerf_amd.render import Renderer
    images, poses, bds, _, i_test, i_ref = L.load_llff_data_infer(root, factor=4, llffhold=8)
    H, W, focal = int(poses[0, 0, 4]), int(poses[0, 1, 4]), float(poses[0, 2, 4])
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]], dtype=np.float32)
    w = synthetic.load_trained_fixture(fixture)
    rend = Renderer({k: w[k] for k in ('sampler', 'refine', 'nerf')}, max_rays=H * W, device=device)
    i_train = [int(i) for i in i_ref[[0, len(i_ref) // 3, 2 * len(i_ref) // 3, len(i_ref) - 1]]]
    res = {}
    for kind, views in (('hold-out', [int(i) for i in i_test]), ('training', i_train)):
        ps = []
        for v in views:
            rend.set_views(poses[v, :3, :4], poses[i_ref][:, :3, :4], images[i_ref], K)
            rays, or_rays = rend.frame_rays(K, poses[v, :3, :4], H, W)
            rgbd, _ = rend.render_rays(rays, or_rays)
            gt = torch.as_tensor(images[v], dtype=torch.float32).reshape(-1, 3).to(device)
            ps.append(float(-10 * torch.log10(((rgbd[:, :3] - gt) ** 2).mean())))
        res[kind] = ps
        print(f'{kind} views {views}: PSNR vs ground truth ' + ', '.join(f'{p:.2f}' for p in ps) + f' dB (mean {np.mean(ps):.2f}), second pass {rend.ctx.sampler_stats() / (H * W):.3f} of the rays of the last view', flush=True)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--stage1', type=int, default=4000)
    ap.add_argument('--stage2', type=int, default=3000)
    ap.add_argument('--n-rand', type=int, default=4096)
    ap.add_argument('--scene', choices=('pictures', 'consistent'), default='pictures')
    ap.add_argument('--out', default=None)
    a = ap.parse_args()
    if a.out is None:
        a.out = os.path.join(ROOT, 'tests', 'golden', 'trained_synth_scene.npz' if a.scene == 'pictures' else 'trained_scene3d.npz')
    tmp = tempfile.mkdtemp()
    root = llff_synth.make_dataset(os.path.join(tmp, 'scene'), seed=2, n=20, H=189, W=252, factor=4, **SCENE_KW[a.scene])
    common = (f'basedir = {tmp}/logs\ndatadir = {root}\nfactor = 4\nllffhold = 8\nN_rand = {a.n_rand}\nN_samples = 8\nN_point_ray_enc = 48\n'
              'mmnetdepth = 6\nmmnetskips = [10000]\nnum_neighbor = 4\nuse_viewdirs = True\nraw_noise_std = 1e0\nlrate = 5e-4\nweight_decay = 5e-8\n'
              'i_print = 100\ni_testset = 10000000\n')
    cfg1, cfg2 = os.path.join(tmp, 'epi.txt'), os.path.join(tmp, 'refine.txt')
    open(cfg1, 'w').write('expname = s1\n' + common + f'i_weights = {a.stage1}\n')
    torch.manual_seed(0); np.random.seed(0)
    t0 = time.perf_counter()
    tr1, log1 = s1.train(['--config', cfg1, '--max_steps', str(a.stage1)], device='cuda:0')
    torch.cuda.synchronize()
    ck = s2.newest_checkpoint(os.path.join(tmp, 'logs', 's1'))
    if ck is None:
        ck = os.path.join(tmp, 'stage1.tar')
        s2.save_checkpoint(ck, tr1, a.stage1)
    v1 = [e[1] for e in log1 if e[1] != 'test_psnr']
    print(f'stage 1: {a.stage1} iterations in {time.perf_counter() - t0:.1f} s, logged losses {v1[0]:.4f} -> {v1[-1]:.5f}; checkpoint {ck}', flush=True)
    del tr1
    open(cfg2, 'w').write('expname = s2\n' + common + f'pretrain_path = {ck}\ni_weights = 10000000\n')
    t0 = time.perf_counter()
    tr2, log2 = s2.train(['--config', cfg2, '--max_steps', str(a.stage2), '--no_reload'], device='cuda:0')
    torch.cuda.synchronize()
    v2 = [e[1] for e in log2 if e[1] != 'test_psnr']
    print(f'stage 2: {a.stage2} iterations in {time.perf_counter() - t0:.1f} s, logged losses {v2[0]:.5f} -> {v2[-1]:.6f}', flush=True)
    out = {'scene': a.scene, 'stage1_iters': a.stage1, 'stage2_iters': a.stage2, 'stage1_loss': np.array([v1[0], v1[-1]], np.float32), 'stage2_loss': np.array([v2[0], v2[-1]], np.float32)}
    finite = True
    for i in range(26):
        W, b = tr2.read('param', i)
        out[f'W{i}'], out[f'b{i}'] = W.cpu().numpy().astype(np.float32), b.cpu().numpy().astype(np.float32)
        finite = finite and bool(np.isfinite(out[f'W{i}']).all() and np.isfinite(out[f'b{i}']).all())
    assert finite
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    np.savez_compressed(a.out, **out)
    print(f'wrote {a.out}: {os.path.getsize(a.out) / 1e6:.2f} MB', flush=True)
    del tr2
    res = evaluate(root, a.out)
    g = dict(np.load(a.out))
    g['psnr_holdout'], g['psnr_training'] = np.array(res['hold-out'], np.float32), np.array(res['training'], np.float32)
    np.savez_compressed(a.out, **g)


if __name__ == '__main__':
    main()
