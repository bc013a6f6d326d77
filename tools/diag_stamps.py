#!/usr/bin/env python3
"""Diagnostic build (-DPNRF_DIAG): share of wave time spent in the slot wait + barrier, per kernel.
Builds pronerf_amd/lib/libpronerf_hip_diag.so, runs each MLP stage once on the bench frame and reads
the in-kernel s_memtime sums.  The diag build's run time itself is not meaningful (stamps drain LDS)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pronerf_amd import build as B   # noqa: E402

diag = os.path.join(B.LIBDIR, 'libpronerf_hip_diag.so')
if '--build' in sys.argv or not os.path.exists(diag):
    srcs = [os.path.join(B.CSRC, s) for s in B.SOURCES]
    subprocess.run([B._hipcc()] + B.FLAGS + ['-DPNRF_DIAG=1', '-shared', '-I', B.INCLUDE, '-o', diag] + srcs, check=True)
    if '--build' in sys.argv:
        sys.exit(0)
from pronerf_amd import _lib   # noqa: E402
_lib.LIB_PATH = diag
from pronerf_amd import ops, synthetic   # noqa: E402
from pronerf_amd.render import Renderer   # noqa: E402

lib = _lib.load()
H, W = 756, 1008
dev = torch.device('cuda:0')
rend = Renderer(synthetic.make_weights(0, 'trained'), max_rays=H * W, device=dev)
scene = synthetic.make_scene(0, H=H, W=W, focal=815.13, rotate=True)
rend.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
rays, or_rays = rend.frame_rays(scene['K'], scene['c2w'], H, W)


def read(nw):
    n = 256 * nw
    buf = (C.c_ulonglong * (4 * n))()
    lib.pnrf_diag_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    assert lib.pnrf_diag_read(buf, 4 * n) == 0
    a = np.array(buf[:], dtype=np.float64).reshape(n, 4)
    tot, vm, bar, nb = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
    return (f'waves={n} total={tot.mean():.0f} cyc | per slot: period={tot.mean() / nb.mean():.0f}  vmcnt wait={vm.mean() / nb.mean():.0f}  '
            f'barrier wait={bar.mean() / nb.mean():.0f} (min over waves {bar.min() / nb.mean():.0f}, max {bar.max() / nb.mean():.0f})  begins={nb.mean():.0f}')


for _ in range(2):
    depth, _, add, mul, _, _ = ops.sampler_fwd(rend.sampler, rays, want_idx=False, want_rgb=False)
print('sampler:', read(8))
rin = ops.refine_input(rays, or_rays, depth, rend.img4, rend.proj)
for _ in range(2):
    z, pts = ops.refine_fwd(rend.refine, rin, rays, depth)
print('refine :', read(8))
for var, nw in (('1x8', 8), ('2x4', 4)):
    os.environ['PNRF_BF16_VARIANT'] = var
    for _ in range(2):
        rgbd, _ = ops.nerf_fwd(rend.nerf, pts, rays, z, add, mul)
    print(f'nerf {var}:', read(nw))


def timeline(nwg=8):
    """Stamps of tiles 4 and 5 of the last full hidden layer (layer_bf16, -DPNRF_DIAG), per wave, in cycles after the
    workgroup's first stamp: arrive | barrier released | after MFMA 0 | 1 | 8 | 15 issued | (tile 5 likewise) | arrive at tile 6."""
    n = 64 * 8 * 16
    buf = (C.c_ulonglong * n)()
    lib.pnrf_diag_read_timeline.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    assert lib.pnrf_diag_read_timeline(buf, n) == 0
    a = np.array(buf[:], dtype=np.int64).reshape(64, 8, 16)[:, :, :13]
    names = ['arr4', 'rel4', 'm0', 'm1', 'm8', 'm15', 'arr5', 'rel5', 'm0', 'm1', 'm8', 'm15', 'arr6']
    print('      ' + ' '.join(f'{x:>6s}' for x in names))
    for wg in range(nwg):
        t0 = a[wg].min()
        for w in range(8):
            print(f'wg{wg} w{w} ' + ' '.join(f'{int(v - t0):6d}' for v in a[wg, w]))
    rel = a - a[:, :, :1].min(axis=1, keepdims=True)
    print('mean  ' + ' '.join(f'{v:6.0f}' for v in rel.reshape(-1, 13).mean(0)))


os.environ['PNRF_BF16_VARIANT'] = '1x8'
for _ in range(2):
    rgbd, _ = ops.nerf_fwd(rend.nerf, pts, rays, z, add, mul)
print('nerf 1x8 timeline (cycles):')
timeline()
