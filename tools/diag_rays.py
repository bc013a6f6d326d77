import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from oracle import pronerf_oracle as orc, synth
from pronerf_amd import ops
dev = torch.device('cuda:0')
H, W = 756, 1008
scene = synth.make_scene(0, H=H, W=W, focal=815.13, rotate=True)
fr = orc.frame_setup(scene)
rays, orr = ops.frame_rays(scene['K'], scene['c2w'], H, W, device=dev)
for name, a, b in (('rays', rays.cpu().numpy(), fr['rays'].numpy()), ('or_rays', orr.cpu().numpy(), fr['or_rays'].numpy())):
    for c in range(11):
        d = a[:, c] != b[:, c]
        print(name, c, 'mismatch frac %.4f' % d.mean(), 'max abs %.3e' % np.abs(a[:, c] - b[:, c]).max(), 'max rel ulp %.2f' % (np.abs(a[:, c] - b[:, c]) / np.spacing(np.abs(b[:, c]).astype(np.float32) + 1e-30)).max())
