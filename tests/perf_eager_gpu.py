"""Reference-style eager-PyTorch baseline on the GPU (NOT part of the product path).

BASELINE.md §4 item 2: "the reference single-GPU PyTorch rays/s that the >= 10x target is measured
against = the same restatement run with device='cuda' (fp32, unfused, whole frame in one call like
run_S_eS_eN_alter_trt.py:329), timed with device events over 20 repetitions after warm-up".  The
reference itself cannot travel to the GPU box, so its pinned restatement (oracle/) is run on the
device.  Lives under tests/ because only tests may execute oracle code.

    python tests/perf_eager_gpu.py [--rays N] [--reps R]     -> one JSON line
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pronerf_oracle as orc   # noqa: E402
from oracle import synth                    # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rays', type=int, default=756 * 1008)
    ap.add_argument('--reps', type=int, default=20)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    torch.backends.cuda.matmul.allow_tf32 = False
    weights = synth.make_weights(0, 'trained')
    wd = {k: {'W': [torch.from_numpy(w).to(dev) for w in v['W']], 'b': [torch.from_numpy(b).to(dev) for b in v['b']]} for k, v in weights.items()}
    scene = synth.make_scene(0, H=756, W=1008, focal=815.13, rotate=True)
    fr = orc.frame_setup(scene)
    n = min(a.rays, fr['rays'].shape[0])
    rays, or_rays = fr['rays'][:n].to(dev), fr['or_rays'][:n].to(dev)
    mm_input = fr['mm_input'][:n].to(dev)       # precomputed per frame outside the timed loop, as in the reference (trt.py:274-278)
    images, proj = fr['images'].to(dev), fr['proj'].to(dev)
    t1, t2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ms = []
    with torch.no_grad():
        for i in range(a.reps + 2):
            t1.record()
            out = orc.render_rays_infer(wd, rays, or_rays, images, proj, mm_input=mm_input)
            t2.record()
            torch.cuda.synchronize()
            if i >= 2:
                ms.append(t1.elapsed_time(t2))
            del out
    best, mean = min(ms), sum(ms) / len(ms)
    print(json.dumps({'what': 'reference-style eager PyTorch fp32 on one MI355X (oracle restatement, whole batch in one call)',
                      'rays': n, 'reps': a.reps, 'ms_best': best, 'ms_mean': mean, 'rays_per_s_best': n / best * 1e3,
                      'rays_per_s_mean': n / mean * 1e3, 'torch': torch.__version__}))


if __name__ == '__main__':
    main()
