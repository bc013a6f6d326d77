#!/usr/bin/env python3
"""Two-pass sampler: rays sent to the second pass and rays whose sort indices differ from the split-fp16 kernel's, as a function of kappa, on the
full 1008 x 756 frame for the weight sets of tests/test_fullframe_gpu.py (synthetic, heavy-tailed, x4-scaled, optimizer-trained on pictures, and — 'scene' — trained on the consistent 3-D scene and run on its hold-out pose).  Rays whose split-kernel sorted depths are closer than 2e-6 are ties."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pronerf_amd import ops, synthetic as synth
from pronerf_amd.render import Renderer
dev = torch.device('cuda:0')
H, W, FOCAL = 756, 1008, 815.13
for seed, kind in ((0, 'trained'), (3, 'trained'), (2, 'spread'), (1, 'default'), (0, 'heavy'), (0, 'x4'), (0, 'optimizer'), (0, 'scene')):
    scene = synth.scene_for(seed, kind, H=H, W=W, focal=FOCAL, rotate=True)
    w = synth.weight_set(seed, kind)
    rays, _ = ops.frame_rays(scene['K'], scene['c2w'], H, W, near=0., far=1., device=dev)
    mlp = ops.PackedMLP(ops.NET_SAMPLER, w['sampler']['W'], w['sampler']['b'])
    s_ds, s_idx = ops.sampler_fwd(mlp, rays, want_idx=True, want_rgb=False)[:2]
    tie = (s_ds[:, 1:] - s_ds[:, :-1]).min(1)[0] <= 2e-6
    line = []
    for k in (4.0, 3.0, 2.5, 2.0, 1.5, 1.0, 0.5):
        o = ops.sampler_fwd(mlp, rays, want_idx=True, want_rgb=False, two_pass=True, kappa=k)
        mism = int(((o[1] != s_idx).any(1) & ~tie).sum())
        line.append(f'kappa {k:g}: {int(o[6]) / rays.shape[0]:.2%} second pass, {mism} differ' + (f', {int(o[7])} third pass' if int(o[7]) else ''))
    print(f'({seed},{kind}) ties {int(tie.sum())}: ' + '; '.join(line))
