"""Single-GPU rehearsal of the ray-sharded frame and of small calls (pronerf_amd.workloads.shard_rehearsal — the block bench.py reports as
`shard_rehearsal`): render_rays on the shard one rank gets at world sizes 1, 2, 4, 8 (no collective) and on 1024- / 4096-ray calls.

    python tools/shard_scaling.py [--shapes auto wide narrow single] [--out profiles/r04_shard_rehearsal.json]

--shapes forces the workgroup shape of the three fused stages (pnrf_mlp_set_shape; several values = one block each, same process).
Prints one JSON line."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pronerf_amd import synthetic, workloads   # noqa: E402

H, W, FOCAL = 756, 1008, 815.13


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shapes', nargs='*', default=['auto'])
    ap.add_argument('--reps', type=int, default=40)
    ap.add_argument('--out', default=None)
    ap.add_argument('--stages', action='store_true', help='per-stage ms (context events) for rank 0 of every world size and for the small calls')
    a = ap.parse_args()
    weights = synthetic.make_weights(0, 'trained')
    scene = synthetic.make_scene(0, H=H, W=W, focal=FOCAL, rotate=True)
    res = {}
    for w in a.shapes:
        res[w] = workloads.shard_rehearsal(weights, scene, H, W, 'cuda:0', reps=a.reps, shape=(None if w == 'auto' else w), stages=a.stages)
        torch.cuda.empty_cache()
    if len(res) == 1:
        res = next(iter(res.values()))
    line = json.dumps(res)
    if a.out:
        open(a.out, 'w').write(line + '\n')
    print(line)


if __name__ == '__main__':
    main()
