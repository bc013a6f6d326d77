#!/usr/bin/env python3
"""What a kernel of ANOTHER stream that holds some CUs (a collective beside the renderer at N > 1) costs a shard's frame: a 1/8-frame call of
pnrf_render_rays_fwd back to back, alone and with tools/cu_hog.hip occupying `--hog-cus` CUs for `--hog-us` microseconds, started beside every frame.
    python3 tools/cu_steal_probe.py [--hog-cus 16] [--hog-us 200]"""
import argparse
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pronerf_amd import synthetic                # noqa: E402
from pronerf_amd.render import Renderer          # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--hog-cus', type=int, nargs='+', default=[8, 16, 32])
ap.add_argument('--hog-us', type=int, nargs='+', default=[100, 300])
ap.add_argument('--shard', type=int, default=8)
a = ap.parse_args()
H, W = 756, 1008
dev = torch.device('cuda:0')
hog = C.CDLL(os.path.join(ROOT, 'pronerf_amd', 'lib', 'libcu_hog.so'))
weights = synthetic.make_weights(0, 'trained')
scene = synthetic.make_scene(0, H=H, W=W, focal=815.13, rotate=True)
n = H * W // a.shard
rend = Renderer(weights, max_rays=n, device=dev)
rend.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
rays, or_rays = rend.frame_rays(scene['K'], scene['c2w'], H, W, first=0, count=n)
out = torch.empty(n, 4, device=dev)
sink = torch.zeros(4, dtype=torch.int32, device=dev)
side = torch.cuda.Stream()


def run(frames, cus, us):
    cur = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for k in range(frames + 5):
        if k == 5:
            e0.record()
        if cus:
            ev = torch.cuda.Event(); ev.record(cur)
            side.wait_event(ev)                                   # the hog starts when this frame starts
            rc = hog.cu_hog_launch(C.c_void_p(side.cuda_stream), cus, us, C.c_void_p(sink.data_ptr()))
            assert rc == 0, rc
        rend.render_rays(rays, or_rays, out=out)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / frames


base = run(200, 0, 0)
res = {'rays': n, 'alone_ms_per_frame': round(base, 4)}
for us in a.hog_us:
    for cus in a.hog_cus:
        ms = run(200, cus, us)
        res[f'hog_{cus}cus_{us}us_ms_per_frame'] = round(ms, 4)
print(json.dumps(res))
