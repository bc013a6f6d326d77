#!/usr/bin/env python3
"""Frame time with the NeRF-class fine net (what checkpoints of the released stage-2 trainer contain)
next to the DoNeRFTRT fine net the headline benchmark names.  Prints one JSON line."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pronerf_amd import synthetic   # noqa: E402
from pronerf_amd.render import Renderer   # noqa: E402

H, W = 756, 1008
dev = torch.device('cuda:0')
scene = synthetic.make_scene(0, H=H, W=W, focal=815.13, rotate=True)
out = {}
for name in ('donerf', 'nerf_class'):
    w = synthetic.make_weights(0, 'trained')
    if name == 'nerf_class':
        wc = synthetic.make_nerfcls_weights(0, head_scale=0.3)
        w['nerf'] = {'W': [a for a, _ in wc['pts_linears']] + [wc['feature_linear'][0], wc['alpha_linear'][0], wc['views_linears'][0][0], wc['rgb_linear'][0]],
                     'b': [b for _, b in wc['pts_linears']] + [wc['feature_linear'][1], wc['alpha_linear'][1], wc['views_linears'][0][1], wc['rgb_linear'][1]]}
    rend = Renderer(w, max_rays=H * W, device=dev)
    rend.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
    rays, or_rays = rend.frame_rays(scene['K'], scene['c2w'], H, W)
    o = torch.empty(H * W, 4, device=dev)
    for _ in range(3):
        rend.render_rays(rays, or_rays, out=o)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        rend.render_rays(rays, or_rays, out=o)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    out[name] = {'ms_per_frame': round(ms, 3), 'rays_per_s': round(H * W / (ms * 1e-3))}
print(json.dumps(out))
