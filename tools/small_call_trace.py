#!/usr/bin/env python3
"""Back-to-back small pnrf_render_rays_fwd calls for a rocprofv3 --kernel-trace timeline:

    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/small_call_trace.py --rays 1024 --calls 60 --shape narrow
    python3 tools/small_call_trace.py --parse <dir>      # per kernel: median duration; median gap to its predecessor; per-call span

What a chunked caller pays per call = kernel durations + the gaps between dependent launches (configs[1]'s 1024-ray chunks)."""
import argparse
import collections
import csv
import glob
import os
import re
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def short(name):
    m = re.search(r'(\w+)(<[^(]*>)?\(', name)
    return (m.group(1) + (m.group(2) or '')) if m else name[:40]


def parse(d, skip):
    f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    rows = [r for r in rows if any(k in r['Kernel_Name'] for k in ('sampler_', 'refine_kernel', 'nerf16_kernel'))]
    calls, cur = [], []
    for r in rows:
        if 'sampler_p1' in r['Kernel_Name'] and cur:
            calls.append(cur); cur = []
        cur.append(r)
    calls.append(cur)
    calls = [c for c in calls[skip:] if len(c) == 4]
    dur, gap, span = collections.defaultdict(list), collections.defaultdict(list), []
    prev_end = None
    for c in calls:
        for r in c:
            s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
            n = short(r['Kernel_Name'])[:60] + f" grid {int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])}x{r['Workgroup_Size_X']}"
            dur[n].append((e - s) / 1e3)
            if prev_end is not None:
                gap[n].append((s - prev_end) / 1e3)
            prev_end = e
        span.append((int(c[-1]['End_Timestamp']) - int(c[0]['Start_Timestamp'])) / 1e3)
    for n in dur:
        print(f'{n:90s} dur {statistics.median(dur[n]):7.2f} us   gap before {statistics.median(gap[n]) if gap[n] else 0:6.2f} us')
    print(f'calls {len(calls)}: first-kernel start -> last-kernel end, median {statistics.median(span):.2f} us; sum of kernel medians '
          f'{sum(statistics.median(v) for v in dur.values()):.2f} us')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--parse', default=None)
    ap.add_argument('--skip', type=int, default=10)
    ap.add_argument('--rays', type=int, default=1024)
    ap.add_argument('--calls', type=int, default=60)
    ap.add_argument('--shape', default='auto')
    a = ap.parse_args()
    if a.parse:
        return parse(a.parse, a.skip)
    import torch
    from pronerf_amd import synthetic
    from pronerf_amd.render import Renderer
    H, W = 756, 1008
    dev = torch.device('cuda:0')
    scene = synthetic.make_scene(0, H=H, W=W, focal=815.13, rotate=True)
    rend = Renderer(synthetic.make_weights(0, 'trained'), max_rays=a.rays, device=dev, shape=None if a.shape == 'auto' else a.shape)
    rend.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
    rays, or_rays = rend.frame_rays(scene['K'], scene['c2w'], H, W, first=0, count=a.rays)
    out = torch.empty(a.rays, 4, device=dev)
    for _ in range(a.calls):
        rend.render_rays(rays, or_rays, out=out)
    torch.cuda.synchronize()


if __name__ == '__main__':
    main()
