"""Which STAGE produces the wrong rows of the paired-workgroup configuration (NOTEBOOK §12 / §19)?

The frame as 745 chunks of 1024 rays on four streams, but at OPERATOR level — pnrf_sampler_fwd_ws, pnrf_refine_project_fwd, pnrf_nerf_fwd with every
intermediate in a tensor of its own (depth / add / mul, z / pts, raw, rgbd per chunk) — with the narrow shape forced, in a build whose narrow launches
may share a CU (-DPNRF_NARROW_LDS_BYTES=0).  Every intermediate of every chunk is compared with the same call made alone on one stream; a stage is the
ORIGIN of a wrong row when its output differs while all of its inputs are identical.

    python -m pronerf_amd.build --variant pair -DPNRF_NARROW_LDS_BYTES=0
    python tools/coresidency_stage.py pair [frames] [--out profiles/<file>.txt]"""
import argparse
import collections
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.getcwd())
from pronerf_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument('variant')
ap.add_argument('frames', nargs='?', type=int, default=40)
ap.add_argument('--streams', type=int, default=4)
ap.add_argument('--chunk', type=int, default=1024)
ap.add_argument('--out', default=None)
args = ap.parse_args()
if args.variant != 'shipped':
    lib = C.CDLL(os.path.join(os.path.dirname(_lib.LIB_PATH), f'libpronerf_hip_{args.variant}.so'))
    for fn, (res, a) in _lib.SIGNATURES.items():
        f = getattr(lib, fn); f.restype = res; f.argtypes = a
    _lib._lib = lib
from pronerf_amd import ops, synthetic                      # noqa: E402
from pronerf_amd.render import Renderer                     # noqa: E402

lines = []
def say(*a):
    s = ' '.join(str(x) for x in a)
    print(s, flush=True); lines.append(s)

H, W = 756, 1008
dev = torch.device('cuda:0')
scene = synthetic.make_scene(0, H=H, W=W, focal=815.13, rotate=True)
rend = Renderer(synthetic.make_weights(0, 'trained'), max_rays=1024, device=dev, shape='narrow')
rend.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
rays, or_rays = rend.frame_rays(scene['K'], scene['c2w'], H, W)
N = rays.shape[0]
S, R, NF = rend.sampler, rend.refine, rend.nerf
img4, proj = rend.img4, rend.proj
chunks = [(a, min(N, a + args.chunk)) for a in range(0, N, args.chunk)]
NAMES = ('depth', 'add', 'mul', 'z', 'pts', 'raw', 'rgbd')


def call(a, b):
    r, o = rays[a:b], or_rays[a:b]
    d, _, ad, mu, _, _, n2, n3 = ops.sampler_fwd(S, r, want_idx=False, want_rgb=False, two_pass=True)
    z, p = ops.refine_project_fwd(R, r, o, d, img4, proj)
    rgbd, raw = ops.nerf_fwd(NF, p, r, z, ad, mu, want_raw=True)
    return [d, ad, mu, z, p.reshape(b - a, 24), raw.reshape(b - a, 32), rgbd]


say(f'library {args.variant}; {N} rays as {len(chunks)} operator-level chunks of {args.chunk} on {args.streams} streams, {args.frames} frames; narrow shape forced')
ref = [call(a, b) for a, b in chunks]                      # alone, one stream
torch.cuda.synchronize()
again = [call(a, b) for a, b in chunks[:40]]
torch.cuda.synchronize()
assert all(torch.equal(x, y) for c0, c1 in zip(ref, again) for x, y in zip(c0, c1)), 'the sequential calls are not deterministic'
streams = [torch.cuda.Stream(device=dev) for _ in range(args.streams)]
origin = collections.Counter()
differ = collections.Counter()
shown = 0
for rep in range(args.frames):
    got = [None] * len(chunks)
    ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream())
    for s in streams:
        s.wait_event(ev)
    for j, (a, b) in enumerate(chunks):
        with torch.cuda.stream(streams[j % args.streams]):
            got[j] = call(a, b)
    torch.cuda.synchronize()
    for j, (a, b) in enumerate(chunks):
        bad = [(g != r).any(1) for g, r in zip(got[j], ref[j])]
        nb = [int(x.sum()) for x in bad]
        if not any(nb):
            continue
        for k, name in enumerate(NAMES):
            if nb[k]:
                differ[name] += nb[k]
        # origin: first stage (sampler = 0..2, refine = 3..4, nerf = 5..6) with a differing output
        first = next(k for k in range(7) if nb[k])
        stage = 'sampler' if first < 3 else ('refine' if first < 5 else 'nerf')
        origin[stage] += 1
        if shown < 12:
            shown += 1
            rows = bad[first].nonzero().flatten().tolist()
            dmax = float((got[j][first] - ref[j][first]).abs().max())
            say(f'frame {rep} chunk {j} (rays {a}..{b - 1}, stream {j % args.streams}): differing rows per tensor {dict(zip(NAMES, nb))}; origin = {stage} ({NAMES[first]}); '
                f'rows {rows[:3]}..{rows[-1]} ({len(rows)} rows, first row % 128 = {rows[0] % 128}, % 16 = {rows[0] % 16}); max |diff| in {NAMES[first]} {dmax:.3g}')
            if stage == 'refine' and nb[4] and not nb[3]:
                # z is right, pts = o + d z + offset is not: pts_got - pts_ref = (o' - o) + (d' - d) z per component -> which ray's (o', d') did the epilogue use?
                zz = ref[j][3].double()                                            # [n, 8]
                dp = (got[j][4].double() - ref[j][4].double()).reshape(-1, 8, 3)   # [n, 8, 3]
                r0 = rows[0]
                torch.set_printoptions(precision=6, linewidth=220, sci_mode=False)
                say(f'      row {r0}: which of the 24 pts values differ: {(got[j][4][r0] != ref[j][4][r0]).int().tolist()}')
                say(f'      row {r0}: got  pts {got[j][4][r0].tolist()}')
                say(f'      row {r0}: want pts {ref[j][4][r0].tolist()}')
                say(f'      row {r0}: z {ref[j][3][r0].tolist()}; ray o {rays[a + r0, 0:3].tolist()} d {rays[a + r0, 3:6].tolist()}')
                say(f'      row {r0}: got - want {(got[j][4][r0] - ref[j][4][r0]).tolist()}')
                for rloc in rows[:1]:
                    A = torch.stack([torch.ones(8, dtype=torch.float64, device=dev), zz[rloc]], 1)      # [8, 2]
                    sol = torch.linalg.lstsq(A, dp[rloc]).solution                   # [2, 3]: delta o, delta d
                    resid = float((A @ sol - dp[rloc]).abs().max())
                    o_eff = rays[a + rloc, 0:3].double() + sol[0]; d_eff = rays[a + rloc, 3:6].double() + sol[1]
                    dist = (rays[:, 0:3].double() - o_eff).abs().sum(1) + (rays[:, 3:6].double() - d_eff).abs().sum(1)
                    jbest = int(dist.argmin())
                    say(f'      local row {rloc} (ray {a + rloc}): the epilogue used the (o, d) of ray {jbest} (offset {jbest - (a + rloc):+d} rays; match error {float(dist[jbest]):.2e}, '
                        f'fit residual {resid:.1e}); that ray is in chunk {jbest // args.chunk}, local row {jbest % args.chunk}')
            if stage == 'nerf':
                # does the difference sit in the MLP (raw) or only in the compositing epilogue?
                say(f'      raw rows differing: {nb[5]}, rgbd rows differing: {nb[6]}; max |raw diff| {float((got[j][5] - ref[j][5]).abs().max()):.3g}')
    del got
say(f'TOTAL over {args.frames} frames ({args.frames * len(chunks)} chunks): chunks with a wrong row by ORIGIN stage: {dict(origin)}; differing rows per tensor: {dict(differ)}')
if args.out:
    open(args.out, 'w').write('\n'.join(lines) + '\n')
