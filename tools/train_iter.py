#!/usr/bin/env python3
"""Run K training iterations of one BASELINE-size workload on the HIP trainer (product code only; for rocprofv3 runs).

    python3 tools/train_iter.py --workload stage2_iteration | stage1_explore_64 | stage1_explore_256 [--iters K] [--products f32]

Prints one JSON line with the mean device milliseconds per iteration."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pronerf_amd import _lib                     # noqa: E402
from pronerf_amd import workloads as wl          # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--workload', default='stage2_iteration')
ap.add_argument('--iters', type=int, default=10)
ap.add_argument('--warmup', type=int, default=2)
ap.add_argument('--products', default='f16x2')
ap.add_argument('--lib', default='', help='time another build: pronerf_amd/lib/libpronerf_hip_<name>.so (python -m pronerf_amd.build --variant <name> ...)')
args = ap.parse_args()
if args.lib:
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), f'libpronerf_hip_{args.lib}.so')
S = 8 if args.workload == 'stage2_iteration' else int(args.workload.rsplit('_', 1)[1])
wk = wl.TrainWorkload('cuda:0', max_samples=S)
wk.trainer.set_products(args.products)
fn = wk.stage2_step if args.workload == 'stage2_iteration' else (lambda: wk.explore_step(S // 8))
ms, wall = wl.timed_ms(fn, args.iters, args.warmup)
print(json.dumps({'workload': args.workload, 'iters': args.iters, 'warmup': args.warmup, 'ms': ms, 'host_ms': wall, 'products': args.products}))
