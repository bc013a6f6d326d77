#!/usr/bin/env python3
"""Fine net's forward on the fused-MLP engine (default products) against one product launch per layer: losses, images, gradients, time."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pronerf_amd import workloads as wl

wk = wl.TrainWorkload('cuda:0', max_samples=64)
tr = wk.trainer
rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
for name, step in (('stage2', lambda: wk.stage2_step(want_rgb=True, adam=False)), ('explore64', lambda: wk.explore_step(8, want_rgb=True, adam=False))):
    res = {}
    for kind in ('f16x2', 'f16x2_unchained', 'f32'):
        tr.set_products(kind)
        loss, rgb = step()
        res[kind] = (loss.clone(), rgb.clone(), [g.clone() for i in range(14, 26) for g in tr.read('grad', i)])
    for kind in ('f16x2', 'f16x2_unchained'):
        L, rgb, gr = res[kind]
        L32, rgb32, g32 = res['f32']
        print(name, kind, 'loss', float(L[1]), 'vs f32', float(L32[1]), 'rgb rel', rel(rgb, rgb32), 'worst grad rel vs f32', max(rel(a, b) for a, b in zip(gr, g32)))
    print(name, 'engine vs unchained: rgb rel', rel(res['f16x2'][1], res['f16x2_unchained'][1]), 'worst grad rel',
          max(rel(a, b) for a, b in zip(res['f16x2'][2], res['f16x2_unchained'][2])))
for kind in ('f16x2', 'f16x2_wchain', 'f16x2_unchained'):
    tr.set_products(kind)
    ms2, _ = wl.timed_ms(lambda: wk.stage2_step(), 30, 5)
    ms64, _ = wl.timed_ms(lambda: wk.explore_step(8), 20, 3)
    print(json.dumps({'products': kind, 'stage2_ms': ms2, 'explore64_ms': ms64}))
