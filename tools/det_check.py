#!/usr/bin/env python3
"""Run-to-run determinism of a training iteration's gradients, kernel by kernel and as a hipGraph."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pronerf_amd import _lib
if os.environ.get('DET_LIB'):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), f"libpronerf_hip_{os.environ['DET_LIB']}.so")
from pronerf_amd import workloads as wl
wk = wl.TrainWorkload('cuda:0', max_samples=8)
tr = wk.trainer
kinds = sys.argv[1:] or ['f16x2', 'f16x2_unchained']
def grads(): return [g.clone() for i in range(26) for g in tr.read('grad', i)]
for kind in kinds:
    tr.set_products(kind); tr.set_graph(False)
    wk.stage2_step(want_rgb=True, adam=False); g0 = grads()
    wk.stage2_step(want_rgb=True, adam=False); g1 = grads()
    tr.set_graph(True)
    wk.stage2_step(want_rgb=True, adam=False); g2 = grads()
    wk.stage2_step(want_rgb=True, adam=False); g3 = grads()
    tr.set_graph(False)
    d = lambda a, b: [i for i, (x, y) in enumerate(zip(a, b)) if not torch.equal(x, y)]
    rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
    print(kind, 'rel diffs run0/run1 of tensors 36..41', [f'{rel(g0[i], g1[i]):.1e}' for i in range(36, 42)])
    print(kind, 'run0 vs run1', d(g0, g1), 'run0 vs capture', d(g0, g2), 'capture vs replay', d(g2, g3))
