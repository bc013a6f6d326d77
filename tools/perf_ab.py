#!/usr/bin/env python3
"""Interleaved A/B timing of library builds and kernel variants in ONE process (cdna guide §5.4 rule 24).

    python tools/perf_ab.py [--rounds 7] [--frames 10] --configs "lib=;lib=head;lib=,sampler=sampler_split;lib=,nerf=bf16_32x32"

A config selects a library build (lib=<name> -> pronerf_amd/lib/libpronerf_hip_<name>.so, empty = the default build; see
`python -m pronerf_amd.build --variant <name> [flags]`) and the kernel variants of its handles (pnrf_mlp_set_variant: sampler=
default (two passes) | sampler_split | sampler_f32 | sampler_f32_full, refine= default (fp16) | bf16 | refine_16x16, nerf= default (fp16) | bf16 | bf16_32x32 | nerf_4x64) — explicit configuration, the library reads no environment.  shape=wide|narrow|single forces the workgroup shape of all three
stages (pnrf_mlp_set_shape).
Every round renders `--frames` frames per config through pnrf_render_rays_fwd with the context's per-kernel events
(pnrf_ctx_profile_begin / _end); reports median / min over the rounds per stage kernel on the bench workload (one 1008x756 frame).
path=ops times the operator-level sequence instead (sampler, refine_input, refine on refine_in, NeRF: four kernels with the [n,144]
intermediate in HBM), frames back to back with torch events per stage."""
import argparse
import ctypes as C
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pronerf_amd import _lib, ops, synthetic    # noqa: E402
from pronerf_amd.render import Renderer         # noqa: E402

H, W = 756, 1008


def load_lib(name):
    path = _lib.LIB_PATH if not name else os.path.join(os.path.dirname(_lib.LIB_PATH), f'libpronerf_hip_{name}.so')
    lib = C.CDLL(path)
    for fn, (res, args) in _lib.SIGNATURES.items():
        f = getattr(lib, fn, None)            # an older build under comparison may lack the newest entry points
        if f is not None:
            f.restype = res; f.argtypes = args
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rounds', type=int, default=7)
    ap.add_argument('--frames', type=int, default=10)
    ap.add_argument('--configs', default='lib=;lib=,sampler=sampler_split;lib=,nerf=bf16_32x32')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    weights = synthetic.make_weights(0, 'trained')
    scene = synthetic.make_scene(0, H=H, W=W, focal=815.13, rotate=True)
    cfgs = [dict(kv.split('=') for kv in c.split(',') if kv) for c in a.configs.split(';')]
    libs, rends = {}, []
    for c in cfgs:
        n = c.get('lib', '')
        if n not in libs:
            libs[n] = load_lib(n)
        _lib._lib = libs[n]
        var = {k: v for k, v in (('sampler', c.get('sampler', 'default')), ('refine', c.get('refine', 'default')), ('nerf', c.get('nerf', 'default'))) if v != 'default'}
        r = Renderer(weights, max_rays=H * W, device=dev, variants=var, shape=(c.get('shape') or None))
        r.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
        rends.append(r)
    _lib._lib = libs[cfgs[0].get('lib', '')]
    rays, or_rays = rends[0].frame_rays(scene['K'], scene['c2w'], H, W)
    res = [dict() for _ in cfgs]
    for r in range(a.rounds + 1):
        for i, c in enumerate(cfgs):
            _lib._lib = libs[c.get('lib', '')]
            rend = rends[i]
            if c.get('path') == 'ops':
                ev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(a.frames)]
                for e in ev:
                    e[0].record()
                    depth, _, add, mul, _, _ = ops.sampler_fwd(rend.sampler, rays, want_idx=False, want_rgb=False)
                    e[1].record()
                    rin = ops.refine_input(rays, or_rays, depth, rend.img4, rend.proj)
                    e[2].record()
                    z, pts = ops.refine_fwd(rend.refine, rin, rays, depth)
                    e[3].record()
                    ops.nerf_fwd(rend.nerf, pts, rays, z, add, mul)
                    e[4].record()
                torch.cuda.synchronize()
                names = ('sampler_kernel', 'refine_input_kernel', 'refine_kernel', 'nerf_kernel')
                ms = {k: sum(e[i].elapsed_time(e[i + 1]) for e in ev) / a.frames for i, k in enumerate(names)}
            else:
                rend.ctx.profile_begin(a.frames)
                for _ in range(a.frames):
                    rend.render_rays(rays, or_rays)
                ms, _ = rend.ctx.profile_end()
                torch.cuda.synchronize()
            if r == 0:
                continue
            for k, v in ms.items():
                res[i].setdefault(k, []).append(v)
            res[i].setdefault('frame', []).append(sum(ms.values()))
    for i, c in enumerate(cfgs):
        line = ' '.join(f'{k}={statistics.median(v):.3f}/{min(v):.3f}' for k, v in res[i].items())
        print(f'{c}: median/min ms  {line}')


if __name__ == '__main__':
    main()
