#!/usr/bin/env python3
"""Instruction table of a kernel's steady state from hipcc's `-save-temps` assembly: what sits between consecutive MFMAs.

    hipcc ... -save-temps -c pronerf_amd/csrc/pnrf_mlp_kernels.hip     (python -m pronerf_amd.build --isa <dir> does this)
    python tools/isa_table.py <file.s> <kernel name substring> [--out profiles/<tag>_isa_<kernel>.md]

The kernel's code is cut into basic blocks at labels; the block (loop body) that holds the most MFMAs is the hidden-layer steady state
(the layer loop of the fused MLP kernels is a real loop over two ping-pong layers).  For that block and for the whole kernel the tool
prints: instruction counts per class, VALU per MFMA, issue cycles per MFMA (MI355X_MICROARCH.md cycle constants: an MFMA holds the
SIMD's vector issue for 8 cycles, plain VALU 4, transcendental / v_cvt_pk 4-8 — priced 8 for exp/log/rcp/rsq/sqrt/sin/cos, 4 otherwise)
and a histogram of the VALU opcodes, so that every non-MFMA vector instruction in the steady state is accounted for by name."""
import argparse
import collections
import re
import sys

TRANS = ('v_exp_', 'v_log_', 'v_rcp_', 'v_rsq_', 'v_sqrt_', 'v_sin_', 'v_cos_')


def classify(op):
    if op.startswith('v_mfma') or op.startswith('v_smfmac'):
        return 'mfma'
    if op.startswith('v_'):
        return 'valu'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        return 'vmem'
    if op.startswith('s_waitcnt'):
        return 'waitcnt'
    if op.startswith('s_barrier'):
        return 'barrier'
    if op.startswith('s_'):
        return 'salu'
    return 'other'


def kernel_lines(path, name):
    out, inside = [], False
    for line in open(path):
        if not inside:
            if re.match(r'^_Z\S*:', line) and name in line.split(':')[0]:
                inside = True
            continue
        if line.startswith('\t.amdhsa_kernel') or line.lstrip().startswith('.section') or re.match(r'^\t\.end_amdhsa_kernel', line):
            break
        out.append(line.rstrip('\n'))
        if line.strip().startswith('s_endpgm'):
            break
    if not out:
        sys.exit(f'kernel matching {name!r} not found in {path}')
    return out


def parse(lines):
    """-> list of blocks, each {'label', 'ins': [(op, text)]}"""
    blocks = [{'label': '<entry>', 'ins': []}]
    for l in lines:
        t = l.strip()
        if not t or t.startswith(';') or t.startswith('.'):
            if re.match(r'^\.LBB\S*:', t):
                blocks.append({'label': t.split(':')[0], 'ins': []})
            continue
        if re.match(r'^[.\w$]+:', t):
            blocks.append({'label': t.split(':')[0], 'ins': []})
            continue
        t = t.split(';')[0].strip()
        if not t:
            continue
        # inline-asm statements may carry several instructions separated by newlines already split by the printer
        op = t.split()[0]
        blocks[-1]['ins'].append((op, t))
    return blocks


def stats(ins):
    c = collections.Counter(classify(op) for op, _ in ins)
    valu = collections.Counter(op for op, _ in ins if classify(op) == 'valu')
    cyc_valu = sum((8 if op.startswith(TRANS) else 4) * n for op, n in valu.items())
    return c, valu, cyc_valu


def gaps(ins):
    """VALU / LDS / other counts between consecutive MFMAs"""
    out, cur = [], collections.Counter()
    for op, _ in ins:
        k = classify(op)
        if k == 'mfma':
            out.append(cur); cur = collections.Counter()
        else:
            cur[k] += 1
    return out


def report(title, ins, mfma_cycles, w):
    c, valu, cyc_valu = stats(ins)
    n = c['mfma']
    w(f'### {title}\n')
    w(f'{len(ins)} instructions: ' + ', '.join(f'{k} {v}' for k, v in sorted(c.items())) + '\n')
    if n:
        per = c['valu'] / n
        issue = (cyc_valu + 8 * n) / n
        w(f'* VALU per MFMA: **{per:.2f}** ({c["valu"]} / {n});  LDS instructions per MFMA: {c["lds"] / n:.2f}')
        w(f'* vector-issue cycles per MFMA (8 for the MFMA itself + its share of the VALU stream): **{issue:.1f}** of the {mfma_cycles} the MFMA occupies its pipe '
          f'-> one wave alone fills {min(1.0, mfma_cycles / issue):.2f} of the pipe; the two waves of a SIMD share the issue port, so the pipe can stay busy only while '
          f'this number is <= {mfma_cycles}')
        g = gaps(ins)
        hist = collections.Counter(x['valu'] for x in g)
        w('* VALU instructions in the gap before an MFMA (gap size: number of gaps): ' + ', '.join(f'{k}: {v}' for k, v in sorted(hist.items())))
    w('\n| VALU opcode | count | per MFMA |\n|---|---|---|')
    for op, k in valu.most_common():
        w(f'| `{op}` | {k} | {k / n:.3f} |' if n else f'| `{op}` | {k} | |')
    w('')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('asm'); ap.add_argument('kernel')
    ap.add_argument('--out', default=None)
    ap.add_argument('--mfma-cycles', type=int, default=16, help='pipe cycles per MFMA: 16 for 16x16x32, 32 for 32x32x16 / 16x16x4_f32')
    ap.add_argument('--dump-steady', action='store_true', help='also print the steady-state block instruction by instruction')
    a = ap.parse_args()
    blocks = parse(kernel_lines(a.asm, a.kernel))
    lines = []
    w = lines.append
    w(f'# ISA table: `{a.kernel}` ({a.asm.split("/")[-1]})\n')
    allins = [i for b in blocks for i in b['ins']]
    best = max(blocks, key=lambda b: sum(1 for op, _ in b['ins'] if classify(op) == 'mfma'))
    nb = sum(1 for op, _ in best['ins'] if classify(op) == 'mfma')
    w(f'{len(blocks)} basic blocks, {len(allins)} instructions; steady-state block `{best["label"]}` holds {nb} of the kernel\'s '
      f'{sum(1 for op, _ in allins if classify(op) == "mfma")} static MFMAs.\n')
    report(f'steady state: block `{best["label"]}`', best['ins'], a.mfma_cycles, w)
    report('whole kernel (static counts, every block once)', allins, a.mfma_cycles, w)
    if a.dump_steady:
        w('### steady-state block, instruction by instruction\n\n```')
        for op, t in best['ins']:
            w(t)
        w('```')
    text = '\n'.join(lines) + '\n'
    if a.out:
        open(a.out, 'w').write(text)
    print(text)


if __name__ == '__main__':
    main()
