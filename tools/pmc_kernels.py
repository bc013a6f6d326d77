#!/usr/bin/env python3
"""Mean of every counter per kernel name from rocprofv3 --pmc output directories:  python tools/pmc_kernels.py <dir> [name filter ...]"""
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].replace('(anonymous namespace)::', '')[:60]
        if len(sys.argv) > 2 and not any(k in n for k in sys.argv[2:]):
            continue
        acc[n][r['Counter_Name']].append(float(r['Counter_Value']))
        acc[n]['dur_us'].append((float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3)
for n, c in acc.items():
    print(n, {k: round(sum(v) / len(v), 1) for k, v in c.items()}, 'launches', len(c['dur_us']))
