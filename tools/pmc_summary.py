#!/usr/bin/env python3
"""Fold the rocprofv3 --pmc passes of one round into profiles/<tag>_pmc_summary.json.

    python tools/pmc_summary.py <tag> <dir with one sub-directory per pass>

Each pass directory holds rocprofv3's *_counter_collection.csv (one counter set per pass, collected with
--kernel-trace only, as MI355X_MICROARCH.md prescribes).  Per kernel (mean over its launches, copy kernels
dropped):  hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) KiB  — gfx950 reports half of a wide coalesced
read stream in FETCH_SIZE;  clock_GHz = GRBM_GUI_ACTIVE / 8 XCDs / duration;  mfma_busy_frac =
SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE/8);  valu_per_mfma = (SQ_INSTS_VALU - SQ_INSTS_MFMA) / SQ_INSTS_MFMA — SQ_INSTS_VALU
counts the MFMAs themselves as well (checked against the static instruction mix of the kernels' steady-state loops, tools/isa_table.py).  bench.py reads hbm_bytes_per_launch as
roofline.traffic.  The raw CSVs are copied next to the summary."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

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



def short(name):
    if 'sampler_kernel<2>' in name:
        return 'sampler_p3_kernel'               # the exact-fp32 third pass: a few workgroups that leave at once when nothing saturated
    if 'nerf16_kernel' in name:
        return 'nerf_kernel'
    for k in ('nerf_kernel', 'refine_input_kernel', 'refine_kernel', 'sampler_p1_kernel', 'sampler_h16_kernel', 'sampler_kernel', 'frame_rays_kernel',
              'images_pack_kernel'):
        if k in name:
            return 'sampler_kernel' if k == 'sampler_h16_kernel' else k
    return None


acc = defaultdict(lambda: defaultdict(list))
for f in newest_per_dir(glob.glob(os.path.join(src, '**', '*counter_collection.csv'), recursive=True)):
    names = set()
    disp = {}
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = short(r['Kernel_Name'])
            if not k:
                continue
            names.add(r['Counter_Name'])
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
            disp[(k, r['Dispatch_Id'])] = float(r['End_Timestamp']) - float(r['Start_Timestamp'])
    for (k, _), d in disp.items():
        acc[k]['dur_ns@' + '+'.join(sorted(names))].append(d)
    shutil.copy(f, os.path.join(out_dir, f'{tag}_pmc_' + '_'.join(sorted(names))[:60] + '_counter_collection.csv'))

per = {}
for k, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    e = {n: round(v, 1) for n, v in m.items() if not n.startswith('dur_ns@')}
    if 'FETCH_SIZE' in m and 'WRITE_SIZE' in m:
        e['hbm_bytes_per_launch'] = int((2 * m['FETCH_SIZE'] + m['WRITE_SIZE']) * 1024)
    durs = [v for n, v in m.items() if n.startswith('dur_ns@') and 'GRBM_GUI_ACTIVE' in n]
    if durs and 'GRBM_GUI_ACTIVE' in m:
        e['dur_ns_in_clock_pass'] = round(durs[0], 1)
        cyc = m['GRBM_GUI_ACTIVE'] / 8
        e['clock_GHz'] = round(cyc / durs[0], 3)
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in m:
            e['mfma_busy_frac'] = round(m['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc), 3)
    if m.get('SQ_WAVE_CYCLES', 0) > 0:           # where the waves' cycles go (quad-cycle units cancel in the ratios)
        for src_n, dst_n in (('SQ_WAIT_ANY', 'wave_frac_parked'), ('SQ_WAIT_INST_ANY', 'wave_frac_issue_stall'), ('SQ_WAIT_INST_LDS', 'wave_frac_lds_issue_stall'),
                             ('SQ_ACTIVE_INST_ANY', 'wave_frac_active'), ('SQ_ACTIVE_INST_VALU', 'wave_frac_valu'), ('SQ_ACTIVE_INST_LDS', 'wave_frac_lds')):
            if src_n in m:
                e[dst_n] = round(m[src_n] / m['SQ_WAVE_CYCLES'], 3)
    if m.get('SQ_INSTS_MFMA', 0) > 0 and 'SQ_INSTS_VALU' in m:
        e['valu_per_mfma'] = round((m['SQ_INSTS_VALU'] - m['SQ_INSTS_MFMA']) / m['SQ_INSTS_MFMA'], 3)
    per[k] = e
summary = {
    'round': tag,
    'command': 'rocprofv3 --pmc <C> --kernel-trace --output-format csv -d <pass dir> -- python3 bench.py --steps 3 --warmup 1 '
               '--no-cpu-baseline --no-gpu-eager-baseline --no-sustained --no-chunked --no-variants --no-train   (tools/profile_round.sh: one pass per counter set)',
    'units': 'FETCH_SIZE / WRITE_SIZE in KiB per launch (mean over launches); see tools/pmc_summary.py for the derived fields',
    'per_kernel': per,
}
summary.update(provenance(src))
path = os.path.join(out_dir, f'{tag}_pmc_summary.json')
json.dump(summary, open(path, 'w'), indent=1)
print(path)
print(json.dumps({k: {n: v for n, v in e.items() if n in ('hbm_bytes_per_launch', 'clock_GHz', 'mfma_busy_frac', 'valu_per_mfma')} for k, e in per.items()}, indent=1))
