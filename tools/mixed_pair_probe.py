#!/usr/bin/env python3
"""the ELU stages (sampler passes + projecting refine) of frame i+1 BESIDE the NeRF stage of frame i on one CU.

The ELU kernels keep the MFMA pipe 53 % busy (they are bound by VALU issue), the NeRF kernel 73 %: a co-resident pair of one
workgroup of each kind per CU is the only configuration that could fill those idle cycles.  Round 4 excluded it for wrong rows, round 5 found
the cause (packed-fp32 instructions) and removed it; its THROUGHPUT had never been measured.  This does, on a test-only build in which the narrow
launches do not pad their LDS request (so that two 4-wave fused workgroups fit one CU):

    python -m pronerf_amd.build --variant pair -DPNRF_NARROW_LDS_BYTES=0
    python tools/mixed_pair_probe.py [--seconds 2.0] [--out profiles/r06_mixed_pair.json]

Rows (762 048-ray frame, same weights and rays in every row; frames / s over >= `seconds` of back-to-back frames, socket power and shader clock
polled meanwhile; every row's last frame compared with the one-stream frame):
  one_stream_ctx          pnrf_render_rays_fwd on one stream, shipped shapes (the product path)                      <- the reference row
  one_stream_stages       the same three stages through the stage-level entry points on one stream (what the pipeline rows are made of)
  two_streams_wide        ELU stages of frame i+1 on stream B, NeRF stage of frame i on stream A, shipped (wide) shapes: no two fused
                          workgroups fit a CU, the streams only overlap at kernel tails
  one_stream_narrow       all stages forced to 4-wave workgroups, one stream: the price of the shape alone
  two_streams_narrow      the mixed pair: 4-wave ELU workgroups beside 4-wave NeRF workgroups (2 x 16 columns per wave), co-resident
  two_streams_4x64        the same with the NeRF stage's 4 x 64-column shape (every weight fragment read feeds four MFMAs)
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pronerf_amd import _lib                                         # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--variant', default='pair')
ap.add_argument('--seconds', type=float, default=2.0)
ap.add_argument('--weights', default='trained', help="'trained' (the bench's seeded DoNeRFTRT nets) or 'scene3d' (the scene-trained fixture)")
ap.add_argument('--out', default=None)
args = ap.parse_args()

if args.variant != 'shipped':
    lib = C.CDLL(os.path.join(os.path.dirname(_lib.LIB_PATH), f'libpronerf_hip_{args.variant}.so'))
    for fn, (res, a) in _lib.SIGNATURES.items():
        f = getattr(lib, fn); f.restype = res; f.argtypes = a
    _lib._lib = lib
lib = _lib.load()

from bench import PowerPoll                                          # noqa: E402
from pronerf_amd import ops, synthetic                               # noqa: E402
from pronerf_amd.render import Renderer                              # noqa: E402

H, W = 756, 1008
N = H * W
dev = torch.device('cuda:0')
if args.weights == 'scene3d':
    scene, w = synthetic.scene3d_frame(0, 4), synthetic.load_trained_fixture('scene3d')
else:
    scene, w = synthetic.make_scene(0, H=H, W=W, focal=815.13, rotate=True), synthetic.make_weights(0, 'trained')
w = {k: w[k] for k in ('sampler', 'refine', 'nerf')}


def renderer(shape=None, nerf_variant=None):
    r = Renderer(w, max_rays=N, device=dev, shape=shape, variants={'nerf': nerf_variant} if nerf_variant else None)
    r.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
    return r


ref = renderer()
rays, or_rays = ref.frame_rays(scene['K'], scene['c2w'], H, W)
want = torch.empty(N, 4, device=dev)
ref.render_rays(rays, or_rays, out=want)
torch.cuda.synchronize()
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
ws_bytes = int(lib.pnrf_sampler_workspace_bytes(N))


class Buffers:
    def __init__(self):
        f = lambda *s: torch.empty(*s, device=dev)
        self.depth, self.add, self.mul, self.z, self.pts, self.rgbd = f(N, 8), f(N, 8), f(N, 8), f(N, 8), f(N, 8, 3), f(N, 4)
        self.ws = torch.zeros(max(ws_bytes, 64) // 4, device=dev, dtype=torch.int32)


def elu_stages(r, b, stream):
    s = C.c_void_p(stream.cuda_stream)
    _lib.check(lib.pnrf_sampler_fwd_ws(r.sampler.handle, p(rays), N, p(b.depth), p(b.add), p(b.mul), None, None, None, p(b.ws), ws_bytes, -1.0, s), 'sampler')
    nb, Hf, Wf, _ = r.img4.shape
    _lib.check(lib.pnrf_refine_project_fwd(r.refine.handle, p(rays), p(or_rays), p(b.depth), p(r.img4), p(r.proj), nb, Hf, Wf, 1e-5, p(b.z), p(b.pts), N, s), 'refine')


def nerf_stage(r, b, stream):
    _lib.check(lib.pnrf_nerf_fwd(r.nerf.handle, p(b.pts), p(rays), p(b.z), p(b.add), p(b.mul), p(b.rgbd), None, N, C.c_void_p(stream.cuda_stream)), 'nerf')


def run(name, frame_fn, last_out, warm=10):
    for _ in range(warm):
        frame_fn(None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
        frame_fn(i)
    torch.cuda.synchronize()
    est = (time.perf_counter() - t0) / 20
    nfr = max(40, int(args.seconds / est) + 1)
    with PowerPoll() as pw:
        t0 = time.perf_counter()
        for i in range(nfr):
            frame_fn(i)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    out = last_out()
    row = {'row': name, 'frames': nfr, 'seconds': dt, 'ms_per_frame': dt / nfr * 1e3, 'frames_per_s': nfr / dt, 'rays_per_s': N * nfr / dt,
           'rows_differing_from_the_one_stream_frame': int((out != want).any(1).sum()), 'power': pw.result()}
    print(json.dumps(row), flush=True)
    return row


rows = []
out0 = torch.empty(N, 4, device=dev)
rows.append(run('one_stream_ctx', lambda i: ref.render_rays(rays, or_rays, out=out0), lambda: out0))

A, B = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
bufs = [Buffers(), Buffers()]


def staged(r):
    def f(i):
        cur = torch.cuda.current_stream()
        elu_stages(r, bufs[0], cur)
        nerf_stage(r, bufs[0], cur)
    return f


def pipelined(r):
    elu_done = [torch.cuda.Event(), torch.cuda.Event()]
    nerf_done = [torch.cuda.Event(), torch.cuda.Event()]
    state = {'n': 0}

    def f(i):
        k = state['n'] & 1
        if state['n'] >= 2:
            B.wait_event(nerf_done[k])                 # buffers k are free again once the NeRF stage of two frames ago has read them
        elu_stages(r, bufs[k], B)
        elu_done[k].record(B)
        A.wait_event(elu_done[k])
        nerf_stage(r, bufs[k], A)
        nerf_done[k].record(A)
        state['n'] += 1
    f.last = lambda: bufs[(state['n'] - 1) & 1].rgbd
    return f


rows.append(run('one_stream_stages', staged(ref), lambda: bufs[0].rgbd))
f = pipelined(ref)
rows.append(run('two_streams_wide', f, f.last))
nar = renderer(shape='narrow')
rows.append(run('one_stream_narrow', staged(nar), lambda: bufs[0].rgbd))
f = pipelined(nar)
rows.append(run('two_streams_narrow', f, f.last))
n464 = renderer(shape={'sampler': 'narrow', 'refine': 'narrow'}, nerf_variant='nerf_4x64')
rows.append(run('one_stream_4x64', staged(n464), lambda: bufs[0].rgbd))
f = pipelined(n464)
rows.append(run('two_streams_4x64', f, f.last))

base = rows[0]['frames_per_s']
for r in rows:
    r['vs_one_stream_ctx'] = r['frames_per_s'] / base
res = {'library': args.variant, 'weights': args.weights, 'rays': N, 'rows': rows}
print('\n'.join(f"{r['row']:22s} {r['ms_per_frame']:7.3f} ms/frame  x{r['vs_one_stream_ctx']:.3f}  "
                f"{(r['power'] or {}).get('socket_w_mean', float('nan')):7.0f} W  {(r['power'] or {}).get('sclk_mhz_mean') or float('nan'):6.0f} MHz  "
                f"differing rows {r['rows_differing_from_the_one_stream_frame']}" for r in rows))
if args.out:
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    json.dump(res, open(args.out, 'w'), indent=1)
