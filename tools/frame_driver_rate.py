#!/usr/bin/env python3
"""The inference driver end to end at Fern size: a synthetic LLFF scene on disk (756 x 1008 after the loader's factor), a
checkpoint with the trainers' keys, `python -m pronerf_amd.run_S_eS_eN_alter_trt --render_test` in process — per hold-out pose the 20 timed
renders of the reference's loop (device events), and the host wall time of the whole pose (ray set-up, neighbour upload, 20 renders, read-back,
PSNR, PNG hand-off), so that what the driver adds around the kernels is on the record.
    python3 tools/frame_driver_rate.py [--views 17] [--factor 2]"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import llff_synth                                           # noqa: E402
from pronerf_amd import synthetic as synth                  # noqa: E402
from pronerf_amd import run_S_eS_eN_alter_trt as trt        # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--views', type=int, default=17)
ap.add_argument('--factor', type=int, default=2)
a = ap.parse_args()
with tempfile.TemporaryDirectory() as tmp:
    t0 = time.perf_counter()
    root = llff_synth.make_dataset(os.path.join(tmp, 'scene'), seed=1, n=a.views, H=756, W=1008, factor=a.factor)
    sds = synth.state_dicts(synth.make_weights(0, 'trained'))
    ck = os.path.join(tmp, '000123.tar')
    torch.save({'global_step': 123, 'mmr_network_fn_state_dict': sds['sampler'], 'refine_net_state_dict': sds['refine'], 'network_fine_state_dict': sds['nerf']}, ck)
    cfg = os.path.join(tmp, 'cfg.txt')
    open(cfg, 'w').write(f'expname = rate\nbasedir = {tmp}/logs\ndatadir = {root}\nft_path = {ck}\nfactor = {a.factor}\nllffhold = 8\nN_samples = 8\nN_point_ray_enc = 48\n'
                         'mmnetdepth = 6\nmmnetskips = [10000]\nnum_neighbor = 4\nuse_viewdirs = True\n')
    t1 = time.perf_counter()
    kw = trt.train(['--config', cfg, '--render_test'], device='cuda:0')
    t2 = time.perf_counter()
    rms = np.asarray(kw['render_ms'])                      # [poses][20]
    walls = np.asarray(kw['pose_wall_ms'])
    out = {'poses': int(rms.shape[0]), 'timing_reps_per_pose': int(rms.shape[1]), 'render_ms_mean': float(rms[:, 2:].mean()), 'render_ms_first_pose_first_rep': float(rms[0, 0]),
           'pose_wall_ms': [round(float(w), 1) for w in walls], 'host_ms_per_pose_beyond_the_renders': [round(float(w - r.sum()), 1) for w, r in zip(walls, rms)],
           'driver_total_s': round(t2 - t1, 2), 'dataset_build_s': round(t1 - t0, 2)}
    print(json.dumps(out))
