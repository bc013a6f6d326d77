#!/usr/bin/env python3
"""Determinism soak of the trainer at BASELINE size: the same N optimizer steps (stage-2 iterations, then stage-1 exploration iterations at 64 samples per
ray) run twice from the same parameters and moments — kernel by kernel and as hipGraph replays — must end in bit-identical parameters.
    python3 tools/train_soak.py [--steps 400]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pronerf_amd import workloads as wl          # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=400)
a = ap.parse_args()
wk = wl.TrainWorkload('cuda:0', max_samples=64)
tr = wk.trainer
tr.set_products('f16x2')
keep = {k: tr.flat(k).clone() for k in ('param', 'm', 'v', 'm_nerf', 'v_nerf')}
W0, b0 = tr.read('param', 0)


def restore():
    for k, v in keep.items():
        tr.flat(k).copy_(v)
    tr.write('param', 0, W0, b0)
    tr.set_step(0, 0)


def run(graph):
    restore()
    tr.set_graph(graph)
    losses = []
    for i in range(a.steps):
        loss, _ = wk.stage2_step(adam=True) if i % 2 == 0 else wk.explore_step(8, adam=True)
        if i % 50 == 0:
            losses.append(float(loss[0]))
    p = tr.flat('param').clone()
    tr.set_graph(False)
    return p, losses


t0 = time.perf_counter()
p1, l1 = run(False)
p2, l2 = run(False)
p3, l3 = run(True)
print(json.dumps({'steps': a.steps, 'losses_every_50': l1, 'finite': bool(torch.isfinite(p1).all()), 'run_to_run_identical': bool(torch.equal(p1, p2)),
                  'graph_replay_identical': bool(torch.equal(p1, p3)), 'seconds': round(time.perf_counter() - t0, 1)}))
