#!/usr/bin/env python3
"""Iterations per second of the stage-2 training DRIVER (pronerf_amd.run_S_eS_eN_alter_base_refine2.train: ray gathering, the per-batch random
draws, Trainer.fwd_bwd, Adam) on a synthetic LLFF directory — how much of the kernels' speed survives the Python loop.
    python tools/train_driver_rate.py [--n-rand 4096] [--steps 300]"""
import argparse
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import llff_synth   # noqa: E402
from pronerf_amd import run_S_eS_eN_alter_base_refine2 as s2, synthetic as synth   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--n-rand', type=int, default=4096)
ap.add_argument('--steps', type=int, default=300)
ap.add_argument('--seed', type=int, default=7)
ap.add_argument('--products', default=None, help="'f32' or 'f16x2' (PNRF_TRAIN_PRODUCTS for the driver)")
a = ap.parse_args()
if a.products:
    os.environ['PNRF_TRAIN_PRODUCTS'] = a.products
tmp = tempfile.mkdtemp()
root = llff_synth.make_dataset(os.path.join(tmp, 'scene'), seed=2, n=20, H=189, W=252, factor=4)      # 17 training views of 189 x 252 after llffhold
w = synth.make_weights(0, 'trained'); wc = synth.make_nerfcls_weights(0, head_scale=0.3)
sds = synth.state_dicts(w)
pre = os.path.join(tmp, 'stage1.tar')
torch.save({'global_step': 7, 'network_fn_state_dict': synth.nerfcls_state_dict(wc), 'mmr_network_fn_state_dict': sds['sampler'],
            'refine_net_state_dict': sds['refine']}, pre)
cfg = os.path.join(tmp, 'refine.txt')
open(cfg, 'w').write(f'expname = s2\nbasedir = {tmp}/logs\ndatadir = {root}\npretrain_path = {pre}\nfactor = 4\nllffhold = 8\nN_rand = {a.n_rand}\nN_samples = 8\n'
                     'N_point_ray_enc = 48\nmmnetdepth = 6\nmmnetskips = [10000]\nnum_neighbor = 4\nuse_viewdirs = True\nraw_noise_std = 1e0\nlrate = 5e-4\n'
                     'weight_decay = 5e-8\ni_print = 50\ni_weights = 1000000\ni_testset = 1000000\n')
s2.train(['--config', cfg, '--max_steps', '30', '--no_reload'], device='cuda:0')          # warm-up run (allocations, first launches)
torch.cuda.synchronize()
import contextlib
import io
import random
import numpy as np
random.seed(a.seed); np.random.seed(a.seed); torch.manual_seed(a.seed); torch.cuda.manual_seed_all(a.seed)      # the same batches and draws for every run of this script
t0 = time.perf_counter()
with contextlib.redirect_stdout(io.StringIO()):
    _, log = s2.train(['--config', cfg, '--max_steps', str(a.steps), '--no_reload'], device='cuda:0')
torch.cuda.synchronize()
dt = time.perf_counter() - t0
tail = [e[1] for e in log[-40:]]
print(f'{a.steps} iterations of {a.n_rand} rays (incl. set-up of the run): {dt:.2f} s  ->  {dt / a.steps * 1e3:.2f} ms per iteration; '
      f'mean batch loss over the last {len(tail)} logged iterations {sum(tail) / len(tail):.6f} (first logged {log[0][1]:.6f})')
