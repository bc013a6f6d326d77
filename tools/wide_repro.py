#!/usr/bin/env python3
"""Reproducer / localiser for differing rows in forced-WIDE 8192-ray calls (tools/coresidency_stress.py phase `wide_8192`: 64 rows of 41 M differed once
in round 6).  Renders 8192-ray calls walking through the frame, optionally beside the foreign kernels of that phase, compares every row with the one-call
frame and, for every call with a difference, reports which rows (index inside the call, modulo the batch sizes of the stages), which of the four output
channels, how large — and whether the sampler's intermediate outputs (depth / add / mul through the stage-level entry points on the same rays) differ too.

    python tools/wide_repro.py [--variant pre6] [--calls 20000] [--no-foreign] [--shape wide|auto|narrow] [--chunk 8192]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument('--variant', default=None)
ap.add_argument('--calls', type=int, default=20000)
ap.add_argument('--chunk', type=int, default=8192)
ap.add_argument('--shape', default='wide')
ap.add_argument('--no-foreign', action='store_true')
ap.add_argument('--kinds', type=int, nargs=3, default=[0, 2, 0])
ap.add_argument('--sampler', default='default', help="sampler variant ('default' two-pass | 'sampler_split')")
ap.add_argument('--refine', default='default', help="refine variant ('default' fp16 operands | 'bf16')")
ap.add_argument('--only', default=None, help="'refine': loop the projecting refine stage alone on the reference depths; 'refine_head0': the refine net on refine_in from memory; "
                                             "'sampler' / 'nerf': those stages alone")
a = ap.parse_args()

from pronerf_amd import _lib                         # noqa: E402
if a.variant:
    lib = C.CDLL(os.path.join(os.path.dirname(_lib.LIB_PATH), f'libpronerf_hip_{a.variant}.so'))
    for fn, (res, args) in _lib.SIGNATURES.items():
        f = getattr(lib, fn); f.restype = res; f.argtypes = args
    _lib._lib = lib
from pronerf_amd import synthetic                    # noqa: E402
from pronerf_amd.render import Renderer              # noqa: E402

fk = C.CDLL(os.path.join(ROOT, 'pronerf_amd', 'lib', 'libforeign_kernels.so'))
fk.foreign_launch.restype = C.c_int
fk.foreign_launch.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
dev = torch.device('cuda:0')
torch.cuda.set_device(dev)
H, W = 756, 1008
scene = synthetic.make_scene(0, H=H, W=W, focal=815.13, rotate=True)
weights = synthetic.make_weights(0, 'trained')
variants = {k: v for k, v in (('sampler', a.sampler), ('refine', a.refine)) if v != 'default'} or None
rend = Renderer(weights, max_rays=H * W, device=dev, variants=variants)
rend.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
rays, or_rays = rend.frame_rays(scene['K'], scene['c2w'], H, W)
N = rays.shape[0]
ref, ref_idx = rend.render_rays(rays, or_rays, want_idx=True)
ref, ref_idx = ref.clone(), ref_idx.clone()
rw = Renderer(weights, max_rays=a.chunk, device=dev, shape=None if a.shape == 'auto' else a.shape, variants=variants)
rw.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
torch.cuda.synchronize()

fbuf = torch.randint(0, 2 ** 31 - 1, (64 * 1024 * 1024 // 4,), dtype=torch.int32, device=dev)
sink = torch.zeros(4, dtype=torch.int32, device=dev)
side = [torch.cuda.Stream(device=dev) for _ in range(3)]
GR = {0: 2048, 1: 512, 2: 256, 3: 256}
IT = {}


def foreign(kind, stream, iters=None):
    assert fk.foreign_launch(kind, C.c_void_p(stream.cuda_stream), GR[kind], iters or IT[kind], C.c_void_p(fbuf.data_ptr()), fbuf.numel() * 4, C.c_void_p(sink.data_ptr()), None) == 0


def tune(kind, target_us=150.0):          # iterations for ~target microseconds per foreign kernel, measured alone (as tools/coresidency_stress.py)
    it = 8
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        foreign(kind, side[0], it); side[0].synchronize()
        e0.record(side[0]); foreign(kind, side[0], it); e1.record(side[0]); side[0].synchronize()
        us = e0.elapsed_time(e1) * 1e3
        if us > 0.7 * target_us:
            break
        it = max(it + 1, int(it * min(8.0, target_us / max(us, 1.0))))
    return it


for k in set(a.kinds):
    IT[k] = tune(k)


cur = torch.cuda.current_stream()
nch = N // a.chunk
lib = _lib.load()
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
ws_bytes = int(lib.pnrf_sampler_workspace_bytes(a.chunk))
NAMES = ('depth', 'add', 'mul', 'z', 'pts', 'rgbd')


class Buf:
    def __init__(self, n):
        f = lambda *sh: torch.empty(*sh, device=dev)
        self.depth, self.add, self.mul, self.z, self.pts, self.rgbd = f(n, 8), f(n, 8), f(n, 8), f(n, 8), f(n, 8, 3), f(n, 4)
        self.ws = torch.zeros(max(ws_bytes, 64) // 4, device=dev, dtype=torch.int32)


RIN = None


def stages(lo, b, off=0, ref=None):
    """the three stages of one call through the stage-level entry points on the current stream (what pnrf_render_rays_fwd runs), outputs at row `off` of b"""
    st = C.c_void_p(cur.cuda_stream)
    r, orr = rays[lo:lo + a.chunk], or_rays[lo:lo + a.chunk]
    sl = lambda t: t[off:off + a.chunk]
    if a.only and ref is not None:        # one stage alone, its inputs = the undisturbed reference's
        rs = lambda t: t[lo:lo + a.chunk]
        nb, Hf, Wf, _ = rw.img4.shape
        if a.only == 'refine':
            _lib.check(lib.pnrf_refine_project_fwd(rw.refine.handle, p(r), p(orr), p(rs(ref.depth)), p(rw.img4), p(rw.proj), nb, Hf, Wf, 1e-5, p(sl(b.z)), p(sl(b.pts)), a.chunk, st), 'refine')
        elif a.only == 'refine_head0':
            _lib.check(lib.pnrf_refine_fwd(rw.refine.handle, p(RIN[lo:lo + a.chunk]), p(r), p(rs(ref.depth)), p(sl(b.z)), p(sl(b.pts)), a.chunk, st), 'refine_fwd')
        elif a.only == 'sampler':
            _lib.check(lib.pnrf_sampler_fwd_ws(rw.sampler.handle, p(r), a.chunk, p(sl(b.depth)), p(sl(b.add)), p(sl(b.mul)), None, None, None, p(b.ws), ws_bytes, -1.0, st), 'sampler')
        else:
            _lib.check(lib.pnrf_nerf_fwd(rw.nerf.handle, p(rs(ref.pts)), p(r), p(rs(ref.z)), p(rs(ref.add)), p(rs(ref.mul)), p(sl(b.rgbd)), None, a.chunk, st), 'nerf')
        return
    _lib.check(lib.pnrf_sampler_fwd_ws(rw.sampler.handle, p(r), a.chunk, p(sl(b.depth)), p(sl(b.add)), p(sl(b.mul)), None, None, None, p(b.ws), ws_bytes, -1.0, st), 'sampler')
    nb, Hf, Wf, _ = rw.img4.shape
    _lib.check(lib.pnrf_refine_project_fwd(rw.refine.handle, p(r), p(orr), p(sl(b.depth)), p(rw.img4), p(rw.proj), nb, Hf, Wf, 1e-5, p(sl(b.z)), p(sl(b.pts)), a.chunk, st), 'refine')
    _lib.check(lib.pnrf_nerf_fwd(rw.nerf.handle, p(sl(b.pts)), p(r), p(sl(b.z)), p(sl(b.add)), p(sl(b.mul)), p(sl(b.rgbd)), None, a.chunk, st), 'nerf')


# debug build (-DPNRF_DEBUG_EHEAD): the refine kernel records, per (ray, lane half), what it held: [0,16) depth / ray rows, [16,24) placement, [24,32) xor of the
# B operand after every layer ([24] packed inputs, [25 + l] input of hidden layer l, [30] last hidden output), [32,48) last accumulators, [48,120) the 72 inputs
has_dbg = hasattr(lib, 'pnrf_debug_set_ehead')
dbuf = None
if has_dbg:
    lib.pnrf_debug_set_ehead.restype = C.c_int
    lib.pnrf_debug_set_ehead.argtypes = [C.c_void_p, C.c_void_p]
    dbuf = torch.zeros(N, 2, 128, device=dev)
    assert lib.pnrf_debug_set_ehead(dbuf.data_ptr(), rays.data_ptr()) == 0

# reference intermediates of every call position, rendered alone
REF = Buf(nch * a.chunk)
for c in range(nch):
    stages(c * a.chunk, REF, c * a.chunk)
torch.cuda.synchronize()
assert torch.equal(REF.rgbd, ref[:nch * a.chunk]), 'stage-level entry points and the context call disagree on the undisturbed frame'
if a.only == 'refine_head0':             # refine_in of the whole frame from the exact projection operator; its reference outputs replace the projecting stage's
    from pronerf_amd import ops
    RIN = torch.cat([ops.refine_input(rays[c * a.chunk:(c + 1) * a.chunk], or_rays[c * a.chunk:(c + 1) * a.chunk], REF.depth[c * a.chunk:(c + 1) * a.chunk], rw.img4, rw.proj) for c in range(nch)])
    for c in range(nch):
        stages(c * a.chunk, REF, c * a.chunk, ref=REF)
    torch.cuda.synchronize()
CHECK = {'refine': ('z', 'pts'), 'refine_head0': ('z', 'pts'), 'sampler': ('depth', 'add', 'mul'), 'nerf': ('rgbd',), None: ('rgbd',)}[a.only]
REFDBG = dbuf.clone() if has_dbg else None
GROUPS = {'epilogue inputs (depth, ray)': (0, 16), 'xor of packed inputs': (24, 25), 'xor in of hidden 0..4': (25, 30), 'xor last hidden output': (30, 31),
          'last accumulators': (32, 48), 'the 72 inputs': (48, 120)}
work = [Buf(a.chunk), Buf(a.chunk)]
t0 = time.time()
pending, report, total_bad = [], [], 0
for i in range(a.calls):
    c = (i * 7) % nch
    lo = c * a.chunk
    if i % 32 == 0:
        ev = torch.cuda.Event(); ev.record(cur)
        for s in side:
            s.wait_event(ev)
    if not a.no_foreign:
        for k, s in zip(a.kinds, side):
            foreign(k, s)
    b = work[i & 1]
    stages(lo, b, ref=REF if a.only else None)
    nbad = sum((getattr(b, n).reshape(a.chunk, -1) != getattr(REF, n)[lo:lo + a.chunk].reshape(a.chunk, -1)).any(1).sum() for n in CHECK[:1])
    keep = {n: getattr(b, n).clone() for n in (CHECK if a.only else NAMES)}
    if has_dbg:
        keep['dbg'] = dbuf[lo:lo + a.chunk].clone()
    pending.append((i, lo, nbad, keep))
    if len(pending) >= 48 or i == a.calls - 1:
        torch.cuda.synchronize()
        for (ci, clo, nb, keep) in pending:
            nb = int(nb)
            if nb:
                total_bad += nb
                rep = {'call': ci, 'first_ray_of_call': clo, 'rows_rgbd': nb}
                for n in (CHECK if a.only else NAMES):
                    df = (keep[n].reshape(a.chunk, -1) != getattr(REF, n)[clo:clo + a.chunk].reshape(a.chunk, -1))
                    rows = df.any(1).nonzero()[:, 0].tolist()
                    runs = []
                    if rows:
                        start = prev = rows[0]
                        for r in rows[1:]:
                            if r != prev + 1:
                                runs.append([start, prev - start + 1]); start = r
                            prev = r
                        runs.append([start, prev - start + 1])
                        cols = df[rows].any(0).nonzero()[:, 0].tolist()
                        d = (keep[n].reshape(a.chunk, -1)[rows] - getattr(REF, n)[clo:clo + a.chunk].reshape(a.chunk, -1)[rows]).abs()
                        rep[n] = {'rows': len(rows), 'runs': runs[:8], 'columns': cols[:24], 'max_abs_diff': float(d.nan_to_num(9e9).max()),
                                  'got': keep[n].reshape(a.chunk, -1)[rows[0]].tolist()[:8], 'want': getattr(REF, n)[clo + rows[0]].reshape(-1).tolist()[:8]}
                    else:
                        rep[n] = {'rows': 0}
                if has_dbg:
                    zrows = (keep['z'] != REF.z[clo:clo + a.chunk]).any(1).nonzero()[:, 0]
                    d, r = keep['dbg'][zrows].view(torch.int32), REFDBG[clo + zrows].view(torch.int32)
                    rep['dbg'] = {}
                    for g, (x0, x1) in GROUPS.items():
                        ne = (d[:, :, x0:x1] != r[:, :, x0:x1])
                        rep['dbg'][g] = {'rays_differing': int(ne.any(2).any(1).sum()), 'of': len(zrows), 'first_columns': ne.any(0).any(0).nonzero()[:, 0].tolist()[:12]}
                    pl = d[:, 0, 16:24]
                    rep['dbg']['placement HW_ID / LDS_ALLOC / GPR_ALLOC (hex) of the failing waves'] = sorted({(f'{int(x[0]) & 0xffffffff:08x}', f'{int(x[5]) & 0xffffffff:08x}', f'{int(x[6]) & 0xffffffff:08x}') for x in pl.tolist()})[:8]
                    ok_rows = torch.tensor([x for x in range(0, a.chunk, 32) if x not in set((zrows // 32 * 32).tolist())][:4], device=dev)
                    rep['dbg']['placement of four good waves of the call'] = sorted({(f'{int(x[0]) & 0xffffffff:08x}', f'{int(x[5]) & 0xffffffff:08x}', f'{int(x[6]) & 0xffffffff:08x}') for x in keep['dbg'][ok_rows, 0, 16:24].view(torch.int32).tolist()})
                report.append(rep)
                print(json.dumps(rep), flush=True)
        pending = []
    if i % 2000 == 1999:
        print(f'{i + 1} calls, {total_bad} rows differ, {time.time() - t0:.1f} s', file=sys.stderr, flush=True)
print(json.dumps({'library': a.variant or 'shipped', 'only': a.only, 'refine': a.refine, 'shape': a.shape, 'chunk': a.chunk, 'calls': a.calls, 'foreign': not a.no_foreign, 'kinds': a.kinds, 'sampler': a.sampler,
                  'rows_differ': total_bad, 'calls_with_differences': len(report), 'seconds': round(time.time() - t0, 1)}))
