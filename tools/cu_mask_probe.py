#!/usr/bin/env python3
"""Are the fused stages limited by the chip's power budget?  The same launch (256 persistent workgroups, 762 048 rays) on a stream whose CU mask
leaves 1/2 or 1/4 of each XCD's CUs enabled (hipExtStreamCreateWithCUMask): with a fixed clock the time would be 2x / 4x the full-chip time;
whatever it falls short of that is clock the power manager gives back when fewer CUs draw power.
    python3 tools/cu_mask_probe.py"""
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pronerf_amd import ops, synthetic           # noqa: E402
from pronerf_amd.render import Renderer          # noqa: E402

H, W = 756, 1008
dev = torch.device('cuda:0')
hip = C.CDLL('libamdhip64.so')
weights = synthetic.make_weights(0, 'trained')
scene = synthetic.make_scene(0, H=H, W=W, focal=815.13, rotate=True)
rend = Renderer(weights, max_rays=H * W, device=dev)
rend.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
rays, or_rays = rend.frame_rays(scene['K'], scene['c2w'], H, W)
depth, _, add, mul, _, _ = ops.sampler_fwd(rend.sampler, rays, want_idx=False, want_rgb=False)[:6]
rin = ops.refine_input(rays, or_rays, depth, rend.img4, rend.proj)
z, pts = ops.refine_fwd(rend.refine, rin, rays, depth)
torch.cuda.synchronize()
n_cu = torch.cuda.get_device_properties(0).multi_processor_count


def masked_stream(keep_of):                      # keep CU i iff (i // 8) % keep_of == 0  (the mask's bits go round the XCDs: bit i = CU i // 8 of XCD i % 8)
    words = (n_cu + 31) // 32
    m = (C.c_uint32 * words)()
    kept = 0
    for i in range(n_cu):
        if (i // 8) % keep_of == 0:
            m[i // 32] |= 1 << (i % 32); kept += 1
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), C.c_uint32(words), m)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev), kept


STAGES = {
    'nerf_kernel': lambda: ops.nerf_fwd(rend.nerf, pts, rays, z, add, mul),
    'refine_kernel (module level, no projection head)': lambda: ops.refine_fwd(rend.refine, rin, rays, depth),
    'sampler (two passes)': lambda: ops.sampler_fwd(rend.sampler, rays, want_idx=False, want_rgb=False),
}


def timed(fn, stream, reps=12):
    with torch.cuda.stream(stream):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


streams = {k: masked_stream(k) for k in (2, 4)}
out = {'cus': n_cu}
for name, fn in STAGES.items():
    full = timed(fn, torch.cuda.current_stream())
    r = {'full_chip_ms': round(full, 4)}
    for k, (st, kept) in streams.items():
        ms = timed(fn, st)
        r[f'{kept}_cus_ms'] = round(ms, 4)
        r[f'{kept}_cus_rate_per_cu_vs_full'] = round(full * n_cu / (ms * kept), 4)
    out[name] = r
print(json.dumps(out))
