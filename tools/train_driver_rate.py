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
a = ap.parse_args()
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
                     'weight_decay = 5e-8\ni_print = 100000\ni_weights = 1000000\ni_testset = 1000000\n')
s2.train(['--config', cfg, '--max_steps', '30', '--no_reload'], device='cuda:0')          # warm-up run (allocations, first launches)
torch.cuda.synchronize()
t0 = time.perf_counter()
s2.train(['--config', cfg, '--max_steps', str(a.steps), '--no_reload'], device='cuda:0')
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f'{a.steps} iterations of {a.n_rand} rays (incl. set-up of the run): {dt:.2f} s  ->  {dt / a.steps * 1e3:.2f} ms per iteration')
