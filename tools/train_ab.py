#!/usr/bin/env python3
"""Interleaved timing of library builds on the two training workloads:  python tools/train_ab.py <lib> <lib> ...   ('' = the default build)"""
import ctypes, json, os, subprocess, sys
libs = sys.argv[1:] or ['']
res = {l: [] for l in libs}
for rnd in range(3):
    for l in libs:
        for w in ('stage2_iteration', 'stage1_explore_64'):
            cmd = [sys.executable, os.path.join(os.path.dirname(__file__), 'train_iter.py'), '--workload', w, '--iters', '40', '--warmup', '5'] + (['--lib', l] if l else [])
            out = subprocess.run(cmd, capture_output=True, text=True).stdout.strip().splitlines()[-1]
            res[l].append((w, json.loads(out)['ms']))
for l in libs:
    for w in ('stage2_iteration', 'stage1_explore_64'):
        v = sorted(ms for ww, ms in res[l] if ww == w)
        print(f'{l or "default":10s} {w:20s} median {v[len(v) // 2]:.4f} min {v[0]:.4f}')
