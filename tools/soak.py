#!/usr/bin/env python3
"""Determinism soak of the product path: the same frame rendered again and again — one call, 1/8 shards, 1024-ray chunks over four streams (eager
and as one hipGraph) — every result compared bit for bit with the first one-call frame.  Prints how many renders differed (expected: 0).
    python3 tools/soak.py [--frames 1500] [--chunked 40]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pronerf_amd import synthetic                        # noqa: E402
from pronerf_amd.render import ChunkedRenderer, Renderer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--frames', type=int, default=1500)
ap.add_argument('--chunked', type=int, default=40)
a = ap.parse_args()
H, W = 756, 1008
dev = torch.device('cuda:0')
weights = synthetic.make_weights(0, 'trained')
scene = synthetic.make_scene(0, H=H, W=W, focal=815.13, rotate=True)
n = H * W
rend = Renderer(weights, max_rays=n, device=dev)
rend.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
rays, or_rays = rend.frame_rays(scene['K'], scene['c2w'], H, W)
ref = torch.empty(n, 4, device=dev)
rend.render_rays(rays, or_rays, out=ref)
torch.cuda.synchronize()
out = torch.empty_like(ref)
bad = torch.zeros((), dtype=torch.int64, device=dev)
res = {}
t0 = time.perf_counter()
for i in range(a.frames):
    out.zero_()
    rend.render_rays(rays, or_rays, out=out)
    bad += (~torch.equal(out, ref)) if False else (out != ref).any().long()
    if i % 300 == 299:
        print(f'one-call frames {i + 1}: differing so far {int(bad)}', flush=True)
res['one_call_frames'] = a.frames; res['one_call_differing'] = int(bad)
bad.zero_()
bounds = [(k * n // 8, (k + 1) * n // 8) for k in range(8)]
for i in range(a.frames // 4):
    out.zero_()
    for lo, hi in bounds:
        rend.render_rays(rays[lo:hi], or_rays[lo:hi], out=out[lo:hi])
    bad += (out != ref).any().long()
res['shard_frames'] = a.frames // 4; res['shard_differing'] = int(bad)
print('shards done', res, flush=True)
bad.zero_()
ch = ChunkedRenderer(rend, 1024, 4)
for i in range(a.chunked):
    out.zero_()
    ch.render_rays(rays, or_rays, out)
    bad += (out != ref).any().long()
res['chunked_frames_4_streams'] = a.chunked; res['chunked_differing'] = int(bad)
print('chunked eager done', res, flush=True)
bad.zero_()
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    ch.render_rays(rays, or_rays, out)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
with torch.cuda.graph(g):
    ch.render_rays(rays, or_rays, out)
for i in range(a.chunked * 2):
    out.zero_()
    g.replay()
    bad += (out != ref).any().long()
res['chunked_graph_replays'] = a.chunked * 2; res['chunked_graph_differing'] = int(bad)
res['seconds'] = round(time.perf_counter() - t0, 1)
print(json.dumps(res))
