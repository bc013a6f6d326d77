#!/usr/bin/env python3
"""Optimizer-produced weights for the parity tests (VERDICT r3, next #2a): the package's own stage-1 driver from a random initialisation
(run_S_eS_eN_alter_base.train: alternating joint / exploration iterations) followed by its stage-2 driver from that checkpoint
(run_S_eS_eN_alter_base_refine2.train), both on the synthetic LLFF scene of tests/llff_synth.py, on the HIP trainer (Adam, 5e-4).  The 26
layers of the three nets after training -> tests/golden/trained_synth_scene.npz (fp32; what a reference checkpoint's state dicts hold), with
the logged losses.  Neither dataset nor checkpoint of the reference ships, so these are the only weights here that an
optimizer has shaped: heavy-tailed rows, correlated columns, biases moved off their initial scale.

    python tools/make_trained_fixture.py [--stage1 4000] [--stage2 3000] [--out tests/golden/trained_synth_scene.npz]"""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--stage1', type=int, default=4000)
    ap.add_argument('--stage2', type=int, default=3000)
    ap.add_argument('--n-rand', type=int, default=4096)
    ap.add_argument('--out', default=os.path.join(ROOT, 'tests', 'golden', 'trained_synth_scene.npz'))
    a = ap.parse_args()
    tmp = tempfile.mkdtemp()
    root = llff_synth.make_dataset(os.path.join(tmp, 'scene'), seed=2, n=20, H=189, W=252, factor=4)
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
    out = {'stage1_iters': a.stage1, 'stage2_iters': a.stage2, 'stage1_loss': np.array([v1[0], v1[-1]], np.float32), 'stage2_loss': np.array([v2[0], v2[-1]], np.float32)}
    finite = True
    for i in range(26):
        W, b = tr2.read('param', i)
        out[f'W{i}'], out[f'b{i}'] = W.cpu().numpy().astype(np.float32), b.cpu().numpy().astype(np.float32)
        finite = finite and bool(np.isfinite(out[f'W{i}']).all() and np.isfinite(out[f'b{i}']).all())
    assert finite
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    np.savez_compressed(a.out, **out)
    print(f'wrote {a.out}: {os.path.getsize(a.out) / 1e6:.2f} MB', flush=True)


if __name__ == '__main__':
    main()
