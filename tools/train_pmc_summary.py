#!/usr/bin/env python3
"""Fold tools/profile_train.sh's passes into profiles/<tag>_train_pmc_summary.json (read by bench.py's `train` block).

    python tools/train_pmc_summary.py <tag> gpurun_out/prof_train_<tag>

Per workload: HBM bytes per iteration = ((2 FETCH_SIZE + WRITE_SIZE) KiB summed over every dispatch of the 8-iteration run) minus (the same
of the 2-iteration run), / 6 — the set-up kernels cancel; gfx950 reports half of a wide coalesced read stream in FETCH_SIZE
(MI355X_MICROARCH.md, HBM).  Launches per iteration likewise.  The kernel-stats CSV of each workload is copied next to the summary."""
import csv
import glob
import json
import os
import shutil
import sys

tag, src = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir = os.path.join(ROOT, 'profiles')



def newest_per_dir(files):
    """gpurun merges every call's outputs into the same local tree: a pass directory that was profiled twice holds two runs' CSVs (one per
    profiler process id).  Keep, per directory, the newest."""
    best = {}
    for f in files:
        d = os.path.dirname(f)
        if d not in best or os.path.getmtime(f) > os.path.getmtime(best[d]):
            best[d] = f
    return sorted(best.values())

def provenance(src):
    """csrc digest recorded on the GPU box when the passes ran (tools/profile_*.sh) and the commit being summarised (with a dirty mark)."""
    import subprocess
    out = {}
    f = os.path.join(src, 'csrc_digest.txt')
    if os.path.exists(f):
        out['csrc_digest'] = open(f).read().strip()
    try:
        c = subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], cwd=ROOT, capture_output=True, text=True).stdout.strip()
        dirty = subprocess.run(['git', 'status', '--porcelain', '--', 'pronerf_amd/csrc', 'include'], cwd=ROOT, capture_output=True, text=True).stdout.strip()
        out['commit'] = c + ('+uncommitted csrc changes' if dirty else '')
    except Exception:
        pass
    return out

A, B = 2, 8


def total(workload, counter, iters):
    tot, n = 0.0, 0
    for f in newest_per_dir(glob.glob(os.path.join(src, f'pmc_{workload}_{counter}_{iters}', '**', '*counter_collection.csv'), recursive=True)):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r['Counter_Name'] == counter:
                    tot += float(r['Counter_Value']); n += 1
    return tot, n


res = {'tag': tag, 'method': f'(sum over all dispatches of the {B}-iteration run - sum of the {A}-iteration run) / {B - A}; bytes = (2 FETCH_SIZE + WRITE_SIZE) KiB',
       'command': 'rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-trace --output-format csv -d <dir> -- python3 tools/train_iter.py --workload <w> --iters <2|8> --warmup 0',
       'workloads': {}}
for w in ('stage2_iteration', 'stage1_explore_64', 'stage1_explore_256'):
    fa, na = total(w, 'FETCH_SIZE', A); fb, nb = total(w, 'FETCH_SIZE', B)
    wa, _ = total(w, 'WRITE_SIZE', A); wb, _ = total(w, 'WRITE_SIZE', B)
    if nb == 0:
        continue
    fetch = (fb - fa) / (B - A) * 1024; write = (wb - wa) / (B - A) * 1024
    e = {'fetch_bytes_per_iter_raw': int(fetch), 'write_bytes_per_iter': int(write), 'hbm_bytes_per_iter': int(2 * fetch + write),
         'launches_per_iter': (nb - na) / (B - A), 'source': f'profiles/{tag}_train_pmc_summary.json'}
    for name in (f'{w}.json', f'stats_{w}.json'):
        p = os.path.join(src, name)
        if os.path.exists(p):
            try:
                e['ms_unprofiled' if name == f'{w}.json' else 'ms_under_kernel_trace'] = json.loads(open(p).read().strip().splitlines()[-1])['ms']
            except Exception:
                pass
    st = newest_per_dir(glob.glob(os.path.join(src, f'stats_{w}', '**', '*kernel_stats.csv'), recursive=True))
    if st:
        shutil.copy(st[0], os.path.join(out_dir, f'{tag}_train_{w}_kernel_stats.csv'))
    res['workloads'][w] = e
res.update(provenance(src))
json.dump(res, open(os.path.join(out_dir, f'{tag}_train_pmc_summary.json'), 'w'), indent=1)
print(json.dumps(res, indent=1))
