#!/usr/bin/env python3
"""Static scan of the built library for VALU-write -> MFMA-read distances the compiler's hazard recogniser cannot see: the packed activations are produced
by INLINE-ASM conversions (v_cvt_pk_f16_f32, v_cvt_pk_bf16_f32, v_fma_mix{lo,hi}_f16, v_max_f32 ...), which hipcc does not treat as VALU writes when it
counts the wait states "VALU writes VGPR -> MFMA reads it".  For every such conversion in every kernel: the number of instructions up to the first v_mfma
that reads the written register (straight-line: the scan stops at a label or a branch), smallest first.

    python tools/hazard_scan.py [--lib=path] [kernel substring ...] [--max 2]"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pronerf_amd import build

lib = next((a[6:] for a in sys.argv[1:] if a.startswith('--lib=')), build.LIB)
SHOW = int(next((a[7:] for a in sys.argv[1:] if a.startswith('--show=')), 4))
mx = int(next((a[6:] for a in sys.argv[1:] if a.startswith('--max=')), 2))
flt = [a for a in sys.argv[1:] if not a.startswith('--')]
WRITERS = ('v_cvt_pk_f16_f32', 'v_cvt_pk_bf16_f32', 'v_fma_mixlo_f16', 'v_fma_mixhi_f16', 'v_cvt_f16_f32', 'v_pack_b32_f16', 'v_perm_b32', 'v_and_or_b32', 'v_lshl_or_b32',
           'v_max_f32', 'v_med3_f32', 'v_fma_f32', 'v_exp_f32', 'v_mov_b32', 'v_cndmask_b32', 'v_mul_f32', 'v_add_f32')


def regs(tok):
    """VGPR numbers named by an operand token: v12 | v[12:15]"""
    m = re.match(r'^v(\d+)$', tok)
    if m:
        return {int(m.group(1))}
    m = re.match(r'^v\[(\d+):(\d+)\]$', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


for dis, _ in build._disassembly(lib):
    cur, body = None, []
    kernels = {}
    for line in dis.splitlines():
        m = re.match(r'^[0-9a-f]+ <(.+)>:', line)
        if m:
            cur = m.group(1).replace('(anonymous namespace)::', '')
            kernels[cur] = []
        elif cur and line.startswith('\t'):
            kernels[cur].append(line.strip().split('//')[0].strip())
    for name, ins in kernels.items():
        if flt and not any(f in name for f in flt):
            continue
        if not any('v_mfma' in i for i in ins):
            continue
        hist = {}
        worst = []
        for k, i in enumerate(ins):
            t = i.replace(',', ' ').split()
            if not t or t[0] not in WRITERS:
                continue
            dst = regs(t[1]) if len(t) > 1 else set()
            if not dst:
                continue
            for d in range(1, 12):
                if k + d >= len(ins):
                    break
                u = ins[k + d].replace(',', ' ').split()
                if not u:
                    continue
                if u[0].startswith(('s_cbranch', 's_branch', 's_endpgm', 's_setpc')):
                    break
                if u[0].startswith('v_mfma'):
                    src = set().union(*[regs(x) for x in u[2:5]])          # srcA, srcB, srcC
                    if dst & src:
                        hist.setdefault((t[0], d), 0)
                        hist[(t[0], d)] += 1
                        if d <= mx:
                            worst.append((d, k, ' | '.join(ins[k:k + d + 1])))
                        break
                if len(u) > 1 and regs(u[1]) & dst and not u[0].startswith('v_mfma'):
                    break                                              # overwritten before any MFMA read it
        close = {f"{op} +{d}": c for (op, d), c in sorted(hist.items(), key=lambda x: x[0][1]) if d <= SHOW}
        if close:
            print(f'{name[:110]}\n   writer -> first MFMA reading it, distance in instructions: {close}')
            for d, k, txt in sorted(worst)[:6]:
                print(f'     +{d} @{k}: {txt[:230]}')
