#!/usr/bin/env python3
"""Per-tensor gradient difference between the engine path (forward + backward chains) and one launch per layer."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pronerf_amd import workloads as wl
wk = wl.TrainWorkload('cuda:0', max_samples=8)
tr = wk.trainer
rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
res = {}
for kind in ('f16x2', 'f16x2_unchained'):
    tr.set_products(kind)
    wk.stage2_step(want_rgb=True, adam=False)
    res[kind] = [[g.clone() for g in tr.read('grad', i)] for i in range(26)]
names = ['pts%d' % i for i in range(8)] + ['feature', 'alpha', 'views', 'rgb']
for i in range(14, 26):
    a, b = res['f16x2'][i], res['f16x2_unchained'][i]
    print(f'{names[i - 14]:8s} W rel {rel(a[0], b[0]):.2e} (norm {float(b[0].norm()):.2e} vs {float(a[0].norm()):.2e})   b rel {rel(a[1], b[1]):.2e}')
print('sampler/refine worst', max(rel(x, y) for i in range(14) for x, y in zip(res['f16x2'][i], res['f16x2_unchained'][i])))
