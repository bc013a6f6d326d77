#!/usr/bin/env python3
"""Two-pass sampler at the SHIPPED kappa on a population of weight sets and scenes: for `--sets` seeds x the kinds of pronerf_amd.synthetic.make_weights
('trained', 'spread', 'default') the full 1008 x 756 frame's sort indices of the two-pass sampler against the split-fp16 kernel's (fp32-grade, the exact
path), ties (split-kernel sorted depths closer than 2e-6) excluded.  The statistical guarantee of include/pronerf_hip.h, sampled wider than the seven sets
of tests/test_fullframe_gpu.py.      python3 tools/kappa_population.py [--sets 20] [--kappa 2 1]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pronerf_amd import ops, synthetic as synth          # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--sets', type=int, default=20)
ap.add_argument('--kappa', type=float, nargs='*', default=[-1.0, 1.0])        # -1: the library default
ap.add_argument('--scene3d', action='store_true', help='instead: the scene-trained nets (tests/golden/trained_scene3d.npz) on all 20 poses of their scene at 756 x 1008')
a = ap.parse_args()
dev = torch.device('cuda:0')
H, W, FOCAL = 756, 1008, 815.13
tot = {k: {'rays': 0, 'ties': 0, 'differ': 0, 'second_pass': 0} for k in a.kappa}
worst = {k: 0.0 for k in a.kappa}
if a.scene3d:
    w = synth.load_trained_fixture('scene3d')
    mlp = ops.PackedMLP(ops.NET_SAMPLER, w['sampler']['W'], w['sampler']['b'])
    import numpy as np
    base = synth.scene3d_frame(0, 4)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
    from pronerf_amd import load_llff as L
    images, poses, bds, _, i_test, i_ref = L.load_llff_data_infer(synth._SCENE3D['root'], factor=4, llffhold=8)
    for v in range(poses.shape[0]):
        rays, _ = ops.frame_rays(base['K'], poses[v, :3, :4].astype(np.float32), H, W, near=0., far=1., device=dev)
        s_ds, s_idx = ops.sampler_fwd(mlp, rays, want_idx=True, want_rgb=False)[:2]
        tie = (s_ds[:, 1:] - s_ds[:, :-1]).min(1)[0] <= 2e-6
        for k in a.kappa:
            o = ops.sampler_fwd(mlp, rays, want_idx=True, want_rgb=False, two_pass=True, kappa=None if k < 0 else k)
            t = tot[k]
            t['rays'] += rays.shape[0]; t['ties'] += int(tie.sum()); t['differ'] += int(((o[1] != s_idx).any(1) & ~tie).sum()); t['second_pass'] += int(o[6])
            worst[k] = max(worst[k], int(o[6]) / rays.shape[0])
        print(f'view {v}: ' + '; '.join(f"kappa {('default' if k < 0 else k)}: {tot[k]['differ']} differ of {tot[k]['rays']}" for k in a.kappa), file=sys.stderr, flush=True)
    a.sets = 0
for seed in range(100, 100 + a.sets):
    scene = synth.make_scene(seed, H=H, W=W, focal=FOCAL, rotate=True)
    rays, _ = ops.frame_rays(scene['K'], scene['c2w'], H, W, near=0., far=1., device=dev)
    for kind in ('trained', 'spread', 'default'):
        w = synth.make_weights(seed, kind)
        mlp = ops.PackedMLP(ops.NET_SAMPLER, w['sampler']['W'], w['sampler']['b'])
        s_ds, s_idx = ops.sampler_fwd(mlp, rays, want_idx=True, want_rgb=False)[:2]
        tie = (s_ds[:, 1:] - s_ds[:, :-1]).min(1)[0] <= 2e-6
        for k in a.kappa:
            o = ops.sampler_fwd(mlp, rays, want_idx=True, want_rgb=False, two_pass=True, kappa=None if k < 0 else k)
            t = tot[k]
            t['rays'] += rays.shape[0]; t['ties'] += int(tie.sum()); t['differ'] += int(((o[1] != s_idx).any(1) & ~tie).sum()); t['second_pass'] += int(o[6])
            worst[k] = max(worst[k], int(o[6]) / rays.shape[0])
        del mlp
    print(f'seed {seed}: ' + '; '.join(f"kappa {('default' if k < 0 else k)}: {tot[k]['differ']} differ of {tot[k]['rays']}" for k in a.kappa), file=sys.stderr, flush=True)
print(json.dumps({('default (PNRF_SAMPLER_KAPPA)' if k < 0 else f'kappa {k:g}'): dict(v, second_pass_fraction=v['second_pass'] / v['rays'], largest_second_pass_fraction=worst[k]) for k, v in tot.items()}))
