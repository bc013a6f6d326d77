"""Single-GPU rehearsal of the ray-sharded frame: time render_rays on the shard one rank gets at world sizes 1, 2, 4, 8 (no collective).
frame_ms(1) / shard_ms(N) bounds the strong-scaling speed-up bench.py can reach at N GPUs.  Prints one JSON line."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pronerf_amd import synthetic   # noqa: E402
from pronerf_amd.render import Renderer, shard_range   # noqa: E402

H, W, FOCAL = 756, 1008, 815.13
dev = torch.device('cuda:0')
weights = synthetic.make_weights(0, 'trained')
scene = synthetic.make_scene(0, H=H, W=W, focal=FOCAL, rotate=True)
res = {}
for world in (1, 2, 4, 8):
    worst = 0.0
    for rank in sorted({0, world // 2, world - 1}):
        first, count = shard_range(H * W, rank, world)
        rend = Renderer(weights, max_rays=count, device=dev)
        rend.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
        rays, orr = rend.frame_rays(scene['K'], scene['c2w'], H, W, first=first, count=count)
        out = torch.empty(count, 4, device=dev)
        for _ in range(5):
            rend.render_rays(rays, orr, out=out)
        torch.cuda.synchronize()
        reps = 40
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            rend.render_rays(rays, orr, out=out)
        e1.record()
        torch.cuda.synchronize()
        worst = max(worst, e0.elapsed_time(e1) / reps)
        del rend
    res[world] = round(worst, 4)
print(json.dumps({'shard_ms': res, 'speedup_bound': {k: round(res[1] / v, 2) for k, v in res.items()}}))
