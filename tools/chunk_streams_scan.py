#!/usr/bin/env python3
"""configs[1] read literally (the Fern frame as 745 calls of <= 1024 rays): ms per frame against the number of HIP streams the chunks are
dealt over (pronerf_amd.render.ChunkedRenderer), eager and captured as ONE hipGraph; every variant is compared bit for bit with the one-call
frame.      python3 tools/chunk_streams_scan.py [--streams 1 2 4 8 16] [--chunk 1024]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                     # noqa: E402
from pronerf_amd import synthetic                # noqa: E402
from pronerf_amd.render import Renderer          # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--streams', type=int, nargs='+', default=[1, 2, 4, 8, 16])
ap.add_argument('--chunk', type=int, default=1024)
a = ap.parse_args()
H, W = 756, 1008
dev = torch.device('cuda:0')
weights = synthetic.make_weights(0, 'trained')
scene = synthetic.make_scene(0, H=H, W=W, focal=815.13, rotate=True)
rend = Renderer(weights, max_rays=H * W, device=dev)
rend.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
rays, or_rays = rend.frame_rays(scene['K'], scene['c2w'], H, W)
ref = torch.empty(H * W, 4, device=dev)
rend.render_rays(rays, or_rays, out=ref)
torch.cuda.synchronize()
for s in a.streams:
    r = bench.chunked_1024(rend, rays, or_rays, ref, chunk=a.chunk, reps=2, streams=s)
    k = f'streams{s}_' if s > 1 else ''
    print(json.dumps({'streams': s, 'calls_ms': round(r[k + 'calls_ms_per_frame'], 2), 'graph_ms': r.get(k + 'graph_ms_per_frame') and round(r[k + 'graph_ms_per_frame'], 2),
                      'bit_identical': [r[k + 'calls_bit_identical_to_one_call'], r.get(k + 'graph_bit_identical_to_one_call')], 'err': r.get(k + 'graph_error')}), flush=True)
