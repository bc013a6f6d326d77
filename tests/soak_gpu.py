"""Soak script (not a pytest test): many frames / iterations / object lifetimes; prints one JSON line.  Checks for hangs, drift
between repeated identical launches, and device-memory growth between the second and the last create/use/free cycle of each object
(the first cycle pays one-time costs: code objects, the rocBLAS workspace)."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from pronerf_amd import ops, synthetic   # noqa: E402
from pronerf_amd.render import Renderer   # noqa: E402

dev = torch.device('cuda:0')
H, W = 378, 504
scene = synthetic.make_scene(0, H=H, W=W, rotate=True)
w = synthetic.make_weights(0, 'trained')


def free_bytes():
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    return torch.cuda.mem_get_info()[0]


free_r, free_t = [], []
t0 = time.time()
ref = None
for rep in range(6):                                   # object lifetimes: create / use / free
    rend = Renderer(w, max_rays=H * W, device=dev)
    rend.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
    rays, orr = rend.frame_rays(scene['K'], scene['c2w'], H, W)
    for _ in range(100):
        out = rend.render_rays(rays, orr)[0]
    torch.cuda.synchronize()
    if ref is None:
        ref = out.clone()
    assert torch.equal(out, ref), 'repeated identical launches differ'
    del rend, rays, orr, out
    free_r.append(free_bytes())
frames = 600
import test_train_gpu as T   # noqa: E402
from oracle import pronerf_oracle as orc   # noqa: E402  (batch construction helper only)
b = T._batch(0, 12, 16, 7)
layers = orc.trainer_layers(b['w'])
losses = []
for rep in range(5):
    tr = ops.Trainer([W_ for W_, _ in layers], [x for _, x in layers], max_rays=b['N'], device=dev, max_samples=32)
    img4 = ops.images_pack(T.cu(b['images'], dev))
    args = (T.cu(b['rays'], dev), T.cu(b['or_rays'], dev), T.cu(b['target'], dev), img4, T.cu(b['poses'], dev), T.cu(b['K'], dev), b['ref_nos'].to(dev).contiguous())
    rs = np.random.RandomState(rep)
    for it in range(90):
        if it % 2:
            jit = torch.from_numpy(np.minimum(np.abs(rs.randn(b['N'], 32)) / 5, 0.99).astype(np.float32)).to(dev)
            L, _ = tr.explore_fwd_bwd(*args, n_mult=4, dir1=1, jitter=jit, dir2=-1, raw_noise=None, want_rgb=False)
            tr.adam_step(5e-4, weight_decay=5e-8, nerf_only=True)
        else:
            L, _ = tr.fwd_bwd(*args, jitter=T.cu(b['jitter'], dev), jitter_dir=1, raw_noise=T.cu(b['noise'], dev), want_rgb=False)
            tr.adam_step(5e-4, weight_decay=5e-8)
    losses.append(float(L[0]))
    del tr, img4, args, L
    free_t.append(free_bytes())
# steady-state growth: skip the first cycle of each object (the HIP runtime may still enlarge its own pools there)
growth = max(free_r[1] - free_r[-1], free_t[1] - free_t[-1]) / 2 ** 20
print('free MiB after renderer cycles', [round(x / 2 ** 20) for x in free_r], 'after trainer cycles', [round(x / 2 ** 20) for x in free_t], file=sys.stderr)
assert growth < 1.0, f'device memory grows across object lifetimes: {growth:.1f} MiB'
print(json.dumps({'frames': frames, 'train_iterations': 450, 'seconds': round(time.time() - t0, 1), 'final_losses': [round(x, 5) for x in losses],
                  'device_memory_growth_MB': round(growth, 1), 'identical_repeats': True}))
