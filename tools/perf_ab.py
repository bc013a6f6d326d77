#!/usr/bin/env python3
"""Interleaved A/B timing of kernel variants in ONE process (cdna guide §5.4 rule 24).

    python tools/perf_ab.py [--rounds 7] --configs "lib=,fold=1,var=1x8;lib=head,fold=1,var=1x8"

A config selects a library build (lib=<name> -> pronerf_amd/lib/libpronerf_hip_<name>.so, empty = the
default build; see `python -m pronerf_amd.build --variant <name> [flags]`) and the library's environment
knobs, which are re-read on every call (PNRF_SAMPLER_FOLD, PNRF_BF16_VARIANT).  Reports median / min per
stage kernel on the bench workload (one 1008x756 frame)."""
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
        f = getattr(lib, fn); f.restype = res; f.argtypes = args
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rounds', type=int, default=7)
    ap.add_argument('--configs', default='lib=,fold=1,var=1x8;lib=,fold=1,var=2x4;lib=,fold=0,var=1x8')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    weights = synthetic.make_weights(0, 'trained')
    scene = synthetic.make_scene(0, H=H, W=W, focal=815.13, rotate=True)
    cfgs = [dict(kv.split('=') for kv in c.split(',')) for c in a.configs.split(';')]
    libs, rends = {}, {}
    for c in cfgs:
        n = c.get('lib', '')
        if n not in libs:
            libs[n] = load_lib(n)
            _lib._lib = libs[n]
            r = Renderer(weights, max_rays=H * W, device=dev)
            r.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
            rends[n] = r
    rays, or_rays = rends[cfgs[0].get('lib', '')].frame_rays(scene['K'], scene['c2w'], H, W)
    res = {i: {'sampler': [], 'refine_in': [], 'refine': [], 'nerf': [], 'frame': []} for i in range(len(cfgs))}
    ev = lambda: torch.cuda.Event(enable_timing=True)
    for r in range(a.rounds + 1):
        for i, c in enumerate(cfgs):
            _lib._lib = libs[c.get('lib', '')]
            rend = rends[c.get('lib', '')]
            os.environ['PNRF_SAMPLER_FOLD'] = c.get('fold', '1')
            os.environ['PNRF_BF16_VARIANT'] = c.get('var', '16')
            os.environ['PNRF_SAMPLER_PREC'] = c.get('prec', 'h16')
            e = [ev() for _ in range(5)]
            e[0].record()
            depth, _, add, mul, _, _ = ops.sampler_fwd(rend.sampler, rays, want_idx=False, want_rgb=False)
            e[1].record()
            rin = ops.refine_input(rays, or_rays, depth, rend.img4, rend.proj)
            e[2].record()
            z, pts = ops.refine_fwd(rend.refine, rin, rays, depth)
            e[3].record()
            rgbd, _ = ops.nerf_fwd(rend.nerf, pts, rays, z, add, mul)
            e[4].record()
            torch.cuda.synchronize()
            if r == 0:
                continue
            t = [x.elapsed_time(y) for x, y in zip(e[:-1], e[1:])]
            for k, v in zip(('sampler', 'refine_in', 'refine', 'nerf'), t):
                res[i][k].append(v)
            res[i]['frame'].append(sum(t))
    for i, c in enumerate(cfgs):
        line = ' '.join(f'{k}={statistics.median(v):.3f}/{min(v):.3f}' for k, v in res[i].items())
        print(f'{c}: median/min ms  {line}')


if __name__ == '__main__':
    main()
