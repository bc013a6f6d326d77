#!/usr/bin/env python3
"""Per-kernel static figures of the built library's gfx950 code objects: instruction count, MFMA count, and — from the kernel descriptors'
notes — VGPRs, scratch (spill) bytes, LDS.     python tools/kernel_stats.py [filter substring ...]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pronerf_amd import build

LLVM = '/opt/rocm/lib/llvm/bin'
lib = build.LIB
flt = [a for a in sys.argv[1:] if not a.startswith('--')]
for a in sys.argv[1:]:
    if a.startswith('--lib='):
        lib = a[6:]
rows = []
for co in build.device_code_objects(lib):
    with tempfile.NamedTemporaryFile(suffix='.co') as f:
        f.write(co); f.flush()
        dis = subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '-d', '--mcpu=gfx950', '-C', f.name], stdout=subprocess.PIPE, text=True).stdout
        notes = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', f.name], stdout=subprocess.PIPE, text=True).stdout
    cur, n, mf = None, collections.Counter(), collections.Counter()
    for line in dis.splitlines():
        m = re.match(r'^[0-9a-f]+ <(.+)>:', line)
        if m:
            cur = m.group(1); continue
        if cur and line.startswith('\t'):
            n[cur] += 1
            if 'v_mfma' in line:
                mf[cur] += 1
    meta = {}
    for blk in re.split(r'\n\s+- \.agpr_count', notes)[1:]:
        g = lambda k: (re.search(rf'\.{k}:\s+(\S+)', blk) or [None, '?'])[1]
        sym = g('symbol').replace('.kd', '')
        meta[sym] = (g('vgpr_count'), g('private_segment_fixed_size'), g('group_segment_fixed_size'), g('sgpr_count'))
    dem = {}
    if meta:
        out = subprocess.run(['c++filt'] + list(meta), stdout=subprocess.PIPE, text=True).stdout.split('\n')
        dem = dict(zip(out, meta))
    for k in n:
        if '.kd' in k or (flt and not any(s in k for s in flt)):
            continue
        v = meta.get(dem.get(k, k), meta.get(k, ('?', '?', '?', '?')))
        rows.append((k, n[k], mf[k], v))
for k, ni, m, v in sorted(rows):
    print(f'{ni:7d} instr {m:5d} mfma  vgpr {v[0]:>4s} scratch {v[1]:>5s} lds {v[2]:>6s}  {k[:120]}')
