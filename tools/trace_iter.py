#!/usr/bin/env python3
"""Kernel timeline of the last complete training iteration in a rocprofv3 --kernel-trace CSV (iterations start at iter_prepare_kernel):
    python tools/trace_iter.py <dir with *kernel_trace.csv> [--which -2]
prints every launch with its grid, duration and the gap to its predecessor, then totals per kernel name."""
import collections
import csv
import glob
import re
import sys


def short(name):
    m = re.search(r'(\w+)(<[^(]*>)?\(', name)
    return (m.group(1) + (m.group(2) or '')) if m else name[:40]


def main():
    d = sys.argv[1]
    which = int(sys.argv[sys.argv.index('--which') + 1]) if '--which' in sys.argv else -2
    f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    idx = [i for i, r in enumerate(rows) if 'stage_batch' in r['Kernel_Name']]
    nxt = which + 1
    a, b = idx[which], (idx[nxt] if nxt != 0 and nxt < len(idx) else len(rows))
    prev, tot, gaps = None, 0.0, 0.0
    per = collections.defaultdict(lambda: [0, 0.0])
    for r in rows[a:b]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        n = short(r['Kernel_Name'])
        gap = (s - prev) / 1e3 if prev else 0.0
        tot += (e - s) / 1e3
        gaps += max(gap, 0.0)
        per[n][0] += 1
        per[n][1] += (e - s) / 1e3
        print(f"{n:44s} grid {r['Grid_Size_X']:>8}x{r['Grid_Size_Y']:<4} {(e - s) / 1e3:8.1f} us  gap {gap:6.1f}")
        prev = e
    print(f'-- {b - a} launches, kernel time {tot:.1f} us, gaps {gaps:.1f} us')
    for n, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print(f'{n:44s} x{c:3d} {t:8.1f} us  {100 * t / tot:5.1f} %')


if __name__ == '__main__':
    main()
