#!/usr/bin/env python3
"""(Checker-side script, not a pytest test: it times the oracle's eager torch graph next to the HIP trainer.)
Stage-2 training iteration at BASELINE config-4 size (N_rand = 4096 rays, 17 training views of 756x1008, 8 samples):
HIP trainer (pnrf_train_stage2_fwd_bwd + pnrf_trainer_adam_step) vs reference-style eager PyTorch on the same GPU (the
oracle's torch graph with autograd + torch.optim.Adam; note the reference itself additionally replicates all 17 images x8
per step, run_S_eS_eN_alter_base_refine2.py:602-604, which the oracle's projection does not).  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pronerf_oracle as orc   # noqa: E402  (baseline leg only)
from pronerf_amd import _lib, ops, synthetic     # noqa: E402

if '--lib' in sys.argv:                     # time another build: pronerf_amd/lib/libpronerf_hip_<name>.so (python -m pronerf_amd.build --variant <name> ...)
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'libpronerf_hip_' + sys.argv[sys.argv.index('--lib') + 1] + '.so')

dev = torch.device('cuda:0')
HIP_ONLY = '--hip-only' in sys.argv        # skip the eager legs (for rocprofv3 runs)
H, W, NV, N = 756, 1008, 17, 4096
scene = synthetic.make_scene(0, H=H, W=W, n_views=NV, sigma_t=0.2, rotate=True, focal=815.13)
w = synthetic.make_weights(0, 'trained'); w['nerfcls'] = synthetic.make_nerfcls_weights(0, head_scale=0.3)
own = 2
fr = orc.frame_setup({**scene, 'c2w': scene['poses'][own]})
rs = np.random.RandomState(0)
sel = torch.from_numpy(np.sort(rs.choice(H * W, N, replace=False)))
rays, or_rays = fr['rays'][sel].contiguous(), fr['or_rays'][sel].contiguous()
target = torch.from_numpy(scene['images'][own].reshape(-1, 3))[sel].contiguous()
poses = torch.from_numpy(scene['poses']); K = torch.from_numpy(scene['K'])
images = torch.from_numpy(scene['images']).permute(0, 3, 1, 2).contiguous()
order = np.sort(rs.choice(np.arange(0, NV - 1), 4, replace=False)).astype(np.int64)
ref_nos = orc.select_neighbors_train(poses[own][None].expand(N, -1, -1), poses, 4, order)
jitter = torch.from_numpy(np.minimum(np.abs(rs.randn(N, 8)) / 5, 1 - 2e-6).astype(np.float32))
noise = torch.from_numpy(rs.randn(N, 8).astype(np.float32))
cu = lambda x: torch.as_tensor(x, dtype=torch.float32).to(dev).contiguous()

layers = orc.trainer_layers(w)
tr = ops.Trainer([W_ for W_, _ in layers], [b for _, b in layers], max_rays=N, device=dev)
img4 = ops.images_pack(cu(images))
args = (cu(rays), cu(or_rays), cu(target), img4, cu(poses), cu(K), ref_nos.to(dev).contiguous())
kw = dict(jitter=cu(jitter), jitter_dir=1, raw_noise=cu(noise), want_rgb=False)


def timed(fn, iters, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters, (time.perf_counter() - t0) * 1e3 / iters


def hip_step():
    tr.fwd_bwd(*args, **kw)
    tr.adam_step(5e-4, weight_decay=5e-8)


hip_ms, hip_wall = timed(hip_step, 50, 5)
tr.set_products('f32')                      # the same iteration on the exact-fp32 MFMA products
f32_ms, f32_wall = timed(hip_step, 30, 5)
tr.set_products('f16x2')

g = lambda x: x.to(dev)
tl = [(torch.tensor(W_, device=dev, requires_grad=True), torch.tensor(b, device=dev, requires_grad=True)) for W_, b in layers]
opt = torch.optim.Adam([p for pair in tl for p in pair], lr=5e-4, betas=(0.9, 0.999), weight_decay=5e-8)
gi = dict(rays=g(rays), or_rays=g(or_rays), target=g(target), images=g(images), poses=g(poses), K=g(K), ref_nos=g(ref_nos), jitter=g(jitter), noise=g(noise))
torch.set_default_device(dev)


def eager_step():
    opt.zero_grad()
    loss, _, _ = orc.stage2_loss(tl, gi['rays'], gi['or_rays'], gi['target'], gi['images'], gi['poses'], gi['K'], gi['ref_nos'], jitter=gi['jitter'],
                                 jitter_dir=1, raw_noise=gi['noise'])
    loss.backward()
    opt.step()


eager_ms, eager_wall = timed(eager_step, 10, 2) if not HIP_ONLY else (float('nan'), float('nan'))

# stage-1 odd iteration (config 5): NeRF-only step on 8*n_mult explored samples per ray (run_S_eS_eN_alter_base.py:689-729, 929-944)
torch.set_default_device('cpu')
explore = {}
opt_n = torch.optim.Adam([p for pair in tl[14:] for p in pair], lr=5e-4, betas=(0.9, 0.999), weight_decay=5e-8)
for n_mult in (() if '--stage2-only' in sys.argv else (1, 4, 8)):          # --stage2-only: a clean kernel trace of the joint iteration
    S = 8 * n_mult
    jit = torch.from_numpy(np.minimum(np.abs(rs.randn(N, S)) / 5, 0.99).astype(np.float32))
    tr_x = ops.Trainer([W_ for W_, _ in layers], [b for _, b in layers], max_rays=N, device=dev, max_samples=S)
    jd = cu(jit)

    def hip_x():
        tr_x.explore_fwd_bwd(*args, n_mult=n_mult, dir1=1, jitter=jd, dir2=-1, raw_noise=None, want_rgb=False)
        tr_x.adam_step(5e-4, weight_decay=5e-8, nerf_only=True)

    hx_ms, hx_wall = timed(hip_x, 20, 3)
    torch.set_default_device(dev)

    def eager_x():
        opt_n.zero_grad()
        loss, _, _ = orc.stage1_loss(tl, gi['rays'], gi['or_rays'], gi['target'], gi['images'], gi['poses'], gi['K'], gi['ref_nos'], False, n_mult=n_mult,
                                     dir1=1, jitter=jd, dir2=-1)
        loss.backward()
        opt_n.step()

    ex_ms, ex_wall = timed(eager_x, 5, 1) if not HIP_ONLY else (float('nan'), float('nan'))
    torch.set_default_device('cpu')
    explore[S] = {'hip_ms': round(hx_wall, 3), 'eager_ms': round(ex_wall, 3), 'speedup': round(ex_wall / hx_wall, 2)}
    del tr_x
print(json.dumps({'workload': 'stage-2 training iteration, 4096 rays, 17 views 756x1008, 8 samples, NeRF-class fine net, fp32',
                  'products': 'split fp16 (default); fp32_products_ms: exact-fp32 MFMA products',
                  'hip_trainer_ms': round(hip_ms, 3), 'hip_trainer_wall_ms': round(hip_wall, 3), 'fp32_products_ms': round(f32_ms, 3), 'eager_torch_gpu_ms': round(eager_ms, 3),
                  'eager_torch_gpu_wall_ms': round(eager_wall, 3), 'speedup': round(eager_wall / hip_wall, 2),
                  'rays_per_s_hip': round(N / (hip_wall * 1e-3)), 'stage1_explore_by_samples_per_ray': explore}))
