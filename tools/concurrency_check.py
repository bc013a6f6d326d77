"""Concurrency check of the fused stages: the frame as 1024- / 4096-ray calls round-robin over several HIP streams (ChunkedRenderer), workgroup
shapes forced or per launch, compared row by row with the one-call frame.  Round 4 found with it that two 4-wave workgroups of DIFFERENT
fused kernels on one CU corrupt each other's batch-head loads (NOTEBOOK.md); since then a CU holds at most one fused-MLP workgroup and this
prints 0 differing rows.      python tools/concurrency_check.py [library variant]"""
import sys, os, torch, ctypes as C
sys.path.insert(0, os.getcwd())
from pronerf_amd import _lib
if len(sys.argv) > 1:
    lib = C.CDLL(os.path.join(os.path.dirname(_lib.LIB_PATH), f'libpronerf_hip_{sys.argv[1]}.so'))
    for fn, (res, args) in _lib.SIGNATURES.items():
        f = getattr(lib, fn); f.restype = res; f.argtypes = args
    _lib._lib = lib
    print('library variant', sys.argv[1])
from pronerf_amd import synthetic
from pronerf_amd.render import Renderer, ChunkedRenderer
H, W = 756, 1008
dev = torch.device('cuda:0')
scene = synthetic.make_scene(0, H=H, W=W, focal=815.13, rotate=True)
NS = {'sampler': 'wide', 'refine': 'narrow', 'nerf': 'narrow'}
combos = [('all narrow, 4 streams', 'narrow', 4, 1024, None), ('auto, 4 streams', None, 4, 1024, None), ('all narrow, 6 streams x 4096', 'narrow', 6, 4096, None)]
for name, shape, streams, chunk, variants in combos:
    rend = Renderer(synthetic.make_weights(0, 'trained'), max_rays=H * W, device=dev, shape=shape, variants=variants)
    rend.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
    rays, or_rays = rend.frame_rays(scene['K'], scene['c2w'], H, W)
    ref, _ = rend.render_rays(rays, or_rays); ref = ref.clone()
    ch = ChunkedRenderer(rend, chunk, streams)
    tot = 0
    for rep in range(12):
        out = torch.zeros_like(ref)
        ch.render_rays(rays, or_rays, out)
        torch.cuda.synchronize()
        bad = (out != ref).any(1)
        nb = int(bad.sum()); tot += nb
        if nb:
            idx = bad.nonzero().flatten()
            runs = (idx[1:] != idx[:-1] + 1).sum().item() + 1
            print(f'   {name} rep {rep}: {nb} rows differ in {runs} runs; first rows {idx[:4].tolist()} max|d| {float((out[bad]-ref[bad]).abs().max()):.2e}', flush=True)
    print(f'{name}: {tot} differing rows over 6 frames', flush=True)
    del ch, rend
