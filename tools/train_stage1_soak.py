#!/usr/bin/env python3
"""Stage-1 DRIVER (pronerf_amd.run_S_eS_eN_alter_base.train: alternating joint / exploration iterations, n_mult drawn per iteration) on a synthetic
LLFF directory at N_rand 4096 from a random initialisation: iterations per second and the logged losses — the engine path of the trainer (up to
64 samples per ray = 262 144 rows) under the real loop.      python tools/train_stage1_soak.py [--steps 3000]"""
import argparse, os, sys, tempfile, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import llff_synth   # noqa: E402
from pronerf_amd import run_S_eS_eN_alter_base as s1   # noqa: E402
ap = argparse.ArgumentParser(); ap.add_argument('--steps', type=int, default=3000); ap.add_argument('--n-rand', type=int, default=4096)
a = ap.parse_args()
tmp = tempfile.mkdtemp()
root = llff_synth.make_dataset(os.path.join(tmp, 'scene'), seed=2, n=20, H=189, W=252, factor=4)
cfg = os.path.join(tmp, 'epi.txt')
open(cfg, 'w').write(f'expname = s1\nbasedir = {tmp}/logs\ndatadir = {root}\nfactor = 4\nllffhold = 8\nN_rand = {a.n_rand}\nN_samples = 8\nN_point_ray_enc = 48\n'
                     'mmnetdepth = 6\nmmnetskips = [10000]\nnum_neighbor = 4\nuse_viewdirs = True\nraw_noise_std = 1e0\nlrate = 5e-4\nweight_decay = 5e-8\n'
                     'i_print = 100\ni_weights = 10000000\ni_testset = 10000000\n')
torch.manual_seed(0)
t0 = time.perf_counter()
tr, log = s1.train(['--config', cfg, '--max_steps', str(a.steps)], device='cuda:0')
torch.cuda.synchronize()
dt = time.perf_counter() - t0
vals = [e[1] for e in log if e[1] != 'test_psnr']
print(f'{a.steps} stage-1 iterations of {a.n_rand} rays (incl. set-up): {dt:.2f} s -> {dt / a.steps * 1e3:.2f} ms per iteration; logged joint losses: first {vals[0]:.4f}, '
      f'last {vals[-1]:.5f}, min {min(vals):.5f}; all finite: {bool(np.isfinite(vals).all())}')
