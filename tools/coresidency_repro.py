"""Reproducer + diagnosis of round 4's co-residency hazard (NOTEBOOK.md §12, §19): two 4-wave fused-MLP workgroups of DIFFERENT kernels on one CU.

Needs a diagnostic build of the library (never the shipped one: the shipped narrow launches ask for 84 KiB of LDS so that no two share a CU):

    python -m pronerf_amd.build --variant pairdbg -DPNRF_NARROW_LDS_BYTES=0 -DPNRF_DEBUG_EHEAD [more -D...]
    python tools/coresidency_repro.py pairdbg [frames] [--out profiles/<file>.txt]

The frame is rendered as 745 calls of 1024 rays on four streams (narrow shape forced), every row compared with the one-call frame.  With
-DPNRF_DEBUG_EHEAD the fused refine epilogue also records the ray / depth rows it holds in registers since its batch head and where the wave
ran (HW_ID, XCC_ID, LDS_ALLOC, GPR_ALLOC); for every wrong row the script reports which OTHER ray's rows the registers held, lane by lane,
and the placement of the wave — the evidence that tells a stale register from a wrong address from another wave's data."""
import argparse
import collections
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
from pronerf_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument('variant')
ap.add_argument('frames', nargs='?', type=int, default=12)
ap.add_argument('--streams', type=int, default=4)
ap.add_argument('--chunk', type=int, default=1024)
ap.add_argument('--out', default=None)
args = ap.parse_args()

lib = C.CDLL(os.path.join(os.path.dirname(_lib.LIB_PATH), f'libpronerf_hip_{args.variant}.so'))
for fn, (res, a) in _lib.SIGNATURES.items():
    f = getattr(lib, fn); f.restype = res; f.argtypes = a
_lib._lib = lib
has_dbg = hasattr(lib, 'pnrf_debug_set_ehead')
if has_dbg:
    lib.pnrf_debug_set_ehead.restype = C.c_int
    lib.pnrf_debug_set_ehead.argtypes = [C.c_void_p, C.c_void_p]

from pronerf_amd import synthetic                                    # noqa: E402
from pronerf_amd.render import ChunkedRenderer, Renderer             # noqa: E402

lines = []
def say(*a):
    s = ' '.join(str(x) for x in a)
    print(s, flush=True); lines.append(s)

H, W = 756, 1008
dev = torch.device('cuda:0')
scene = synthetic.make_scene(0, H=H, W=W, focal=815.13, rotate=True)
rend = Renderer(synthetic.make_weights(0, 'trained'), max_rays=H * W, device=dev, shape='narrow')
rend.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
rays, or_rays = rend.frame_rays(scene['K'], scene['c2w'], H, W)
N = rays.shape[0]
say(f'library variant {args.variant}; debug dump: {has_dbg}; {N} rays as {-(-N // args.chunk)} calls of {args.chunk} on {args.streams} streams, {args.frames} frames')

dbg = torch.zeros(N, 2, 128, device=dev) if has_dbg else None
if has_dbg:
    assert lib.pnrf_debug_set_ehead(dbg.data_ptr(), rays.data_ptr()) == 0
ref, _ = rend.render_rays(rays, or_rays)
ref = ref.clone()
torch.cuda.synchronize()
if has_dbg:
    dref = dbg.clone()
    # the one-call frame's own record must be the rows themselves
    ok_ray = bool((dref[:, :, 8:16] == rays[:, None, :8]).all())
    say(f'one-call frame: recorded ray rows equal the ray tensor: {ok_ray}')
    keys = {}
    dnp = dref[:, 0, :16].cpu().numpy()
    for j in range(N):
        keys.setdefault(dnp[j, :8].tobytes(), j)
    rkeys = {}
    rnp = rays[:, :8].cpu().numpy()
    for j in range(N):
        rkeys.setdefault(rnp[j].tobytes(), j)

ch = ChunkedRenderer(rend, args.chunk, args.streams)
tot_rows = tot_runs = 0
hist = collections.Counter()
layer_hist = collections.Counter()
for rep in range(args.frames):
    out = torch.zeros_like(ref)
    if has_dbg:
        dbg.zero_()
    ch.render_rays(rays, or_rays, out)
    torch.cuda.synchronize()
    bad = (out != ref).any(1)
    nb = int(bad.sum())
    tot_rows += nb
    if not nb:
        continue
    idx = bad.nonzero().flatten().cpu().numpy()
    starts = [0] + [k + 1 for k in range(len(idx) - 1) if idx[k + 1] != idx[k] + 1]
    tot_runs += len(starts)
    dmax = (out[bad] - ref[bad]).abs().max(0)[0].tolist()
    say(f'frame {rep}: {nb} rows differ in {len(starts)} runs; max |difference| per channel (r, g, b, depth): {[float("%.3g" % x) for x in dmax]}')
    if not has_dbg:
        continue
    d = dbg.cpu().numpy()
    di = d.view(np.int32)
    for s in starts[:6]:
        i0 = int(idx[s])
        run = [int(x) for x in idx[s:] if x - i0 < 32 and x // 32 == i0 // 32]
        loc = i0 % args.chunk
        say(f'  run at ray {i0} (+{len(run)}): call {i0 // args.chunk}, stream {(i0 // args.chunk) % args.streams}, workgroup {loc // 128}, wave {(loc % 128) // 32}, columns {loc % 32}..{loc % 32 + len(run) - 1}')
        for hh in (0, 1):
            hw, xcc, lds, gpr = (int(di[i0, hh, k]) & 0xffffffff for k in (16, 17, 21, 22))
            good = i0 - (i0 % 32)        # column 0 of the same wave (a lane of quad 0)
            say(f'    half {hh}: HW_ID {hw:08x} (wave {hw & 15} simd {(hw >> 4) & 3} cu {(hw >> 8) & 15} sh {(hw >> 12) & 1} se {(hw >> 13) & 7}) XCC {xcc & 15} '
                f'LDS_ALLOC {lds:08x} (base {(lds & 0xff) * 256} B... size field {(lds >> 12) & 0x1ff}) GPR_ALLOC {gpr:08x}; column 0 of the wave: LDS_ALLOC {int(di[good, hh, 21]) & 0xffffffff:08x}')
            if d.shape[2] >= 120:            # where the difference enters: inputs (feat), B operand after every layer (xor), last accumulators
                dr = dref[run[0]:run[-1] + 1, hh].cpu().numpy(); dri = dr.view(np.int32)
                dw = d[run[0]:run[-1] + 1, hh]; dwi = dw.view(np.int32)
                feat_bad = int((dw[:, 48:120] != dr[:, 48:120]).any(1).sum())
                cks_bad = [int((dwi[:, 24 + k] != dri[:, 24 + k]).sum()) for k in range(7)]
                fin_bad = int((dw[:, 32:48] != dr[:, 32:48]).any(1).sum())
                first = next((k for k, c in enumerate(cks_bad) if c), None)
                say(f'      lanes of the run with different inputs (72 features): {feat_bad}; with a different B-operand xor after [inputs, layer 0..5]: {cks_bad}; '
                    f'with different last accumulators: {fin_bad}  -> first difference at index {first}')
                layer_hist[first] += 1
                if feat_bad:
                    r0 = next(k for k in range(len(run)) if (dw[k, 48:120] != dr[k, 48:120]).any())
                    cols = np.nonzero(dw[r0, 48:120] != dr[r0, 48:120])[0]
                    say(f'      first lane with different inputs: ray {run[r0]}, features {cols[:24].tolist()}{"..." if len(cols) > 24 else ""}; got {dw[r0, 48 + cols[:4]].tolist()} want {dr[r0, 48 + cols[:4]].tolist()}')
            for i in run[:3] + run[-1:]:
                jd = keys.get(d[i, hh, :8].tobytes()); jr = rkeys.get(d[i, hh, 8:16].tobytes())
                okd = bool((d[i, hh, :8] == dnp[i, :8]).all()); okr = bool((d[i, hh, 8:16] == rnp[i]).all())
                hist[(('depth', None if jd is None else jd - i), ('ray', None if jr is None else jr - i))] += 1
                say(f'      ray {i} (col {i % 32}): depth row held = that of ray {jd} ({"own" if okd else "d=" + str(None if jd is None else jd - i)}), ray row held = that of ray {jr} ({"own" if okr else "d=" + str(None if jr is None else jr - i)})')
say(f'TOTAL: {tot_rows} differing rows in {tot_runs} runs over {args.frames} frames ({args.frames * -(-N // args.chunk)} calls)')
if layer_hist:
    say('first stage whose B operand differs (0 = packed inputs, 1..6 = output of layer 0..5, None = only the epilogue), per (run, half):', dict(layer_hist))
if hist:
    say('offsets (held row - own row) of the inspected wrong lanes:', dict(hist))
if args.out:
    open(args.out, 'w').write('\n'.join(lines) + '\n')
