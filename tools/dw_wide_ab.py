#!/usr/bin/env python3
"""Round 5: the grouped split-fp16 weight gradients on 256 x 128 tiles (dwh_body_wide) against the 128 x 128 tiles, same process, interleaved, on the three
BASELINE-size training workloads: ms per iteration, and the gradients of the two forms against each other (same arithmetic; the partial sums are grouped
differently, so the difference is fp32 round-off).      python3 tools/dw_wide_ab.py [--rounds 5] [--iters 30]"""
import argparse
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pronerf_amd import workloads as wl          # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--rounds', type=int, default=5)
ap.add_argument('--iters', type=int, default=30)
a = ap.parse_args()
out = {}
for name, S in (('stage2_iteration', 8), ('stage1_explore_64', 64), ('stage1_explore_256', 256)):
    wk = wl.TrainWorkload('cuda:0', max_samples=S)
    fn = wk.stage2_step if name == 'stage2_iteration' else (lambda: wk.explore_step(S // 8))
    tr = wk.trainer
    grads = {}
    nostep = (lambda: wk.stage2_step(adam=False)) if name == 'stage2_iteration' else (lambda: wk.explore_step(S // 8, adam=False))
    for form, tile in (('square_128', 255), ('wide_256x128', 256), ('square_again', 255)):
        tr.set_dw_kernel(tile, 0)
        nostep(); torch.cuda.synchronize()                     # same parameters for every form: no optimizer step
        grads[form] = tr.flat('grad').clone()
    ref = grads['square_128'].double()
    rel = float((grads['wide_256x128'].double() - ref).norm() / ref.norm())
    assert torch.equal(grads['square_again'], grads['square_128'])
    worst = float(((grads['wide_256x128'].double() - ref).abs() / (ref.abs() + 1e-6 * float(ref.abs().max()))).max())
    ms = {'square_128': [], 'wide_256x128': []}
    for r in range(a.rounds):
        for form, tile in (('square_128', 255), ('wide_256x128', 256)):
            tr.set_dw_kernel(tile, 0)
            ms[form].append(wl.timed_ms(fn, a.iters, 3)[0])
    out[name] = {f: {'median_ms': round(statistics.median(v), 4), 'min_ms': round(min(v), 4)} for f, v in ms.items()}
    out[name]['gradient_rel_l2_difference'] = rel
    out[name]['gradient_worst_elementwise_rel'] = worst
    out[name]['finite'] = bool(torch.isfinite(grads['wide_256x128']).all())
    del wk, tr
    torch.cuda.empty_cache()
print(json.dumps(out))
