"""Host cost of one render_rays call (ctypes + argument checks + 4 launches), measured on a tiny batch so the GPU is never the limit."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pronerf_amd import synthetic
from pronerf_amd.render import Renderer
dev = torch.device('cuda:0')
scene = synthetic.make_scene(0, H=16, W=16, rotate=True)
rend = Renderer(synthetic.make_weights(0, 'trained'), max_rays=256, device=dev)
rend.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
rays, orr = rend.frame_rays(scene['K'], scene['c2w'], 16, 16)
out = torch.empty(256, 4, device=dev)
for _ in range(50):
    rend.render_rays(rays, orr, out=out)
torch.cuda.synchronize()
n = 3000
t0 = time.perf_counter()
for _ in range(n):
    rend.render_rays(rays, orr, out=out)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(json.dumps({'issue_us_per_call': round((t1 - t0) / n * 1e6, 1), 'total_us_per_call': round((t2 - t0) / n * 1e6, 1)}))
