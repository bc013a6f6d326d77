"""Build libpronerf_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).

    python -m pronerf_amd.build [--force]

The shared library lands in pronerf_amd/lib/ (git-ignored, but shipped to the GPU box by
gpurun).  A content hash of the sources is stored next to it so that rebuilds are skipped
when nothing changed.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
INCLUDE = os.path.join(os.path.dirname(HERE), 'include')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libpronerf_hip.so')
SOURCES = ['pnrf_pack.hip', 'pnrf_ops.hip', 'pnrf_mlp_kernels.hip', 'pnrf_train.hip']
LINK = []                    # no library dependencies: every kernel, the training step's layer products included, is in csrc/
ARCH = 'gfx950'
# No packed-fp32 instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 / v_pk_mov_b32) in any kernel of the library.  Round 5 traced round 4's "co-residency
# hazard" to them: in the fused refine kernel's scalar epilogue — where the SLP vectorizer had paired the fp32 arithmetic — a v_pk_mul_f32 returned 0 in one
# 16-lane quarter of a wave a few times per thousand calls, but only while a workgroup of a DIFFERENT fused kernel shared the CU (loads, every layer's B
# operand and the last accumulators bit-identical in the failing batches; 121 / 209 / 102 bad chunks per 29 800 with packed fp32, 0 per 283 100 without;
# tools/pkf32_coexec_probe.hip reproduces it in a micro-kernel: victim in MODE.FP16_OVFL, 240-register windows on both waves, a 16x16x32 bf16 MFMA partner:
# NOTEBOOK §19, profiles/r05_coresidency_*).  The feature is switched off for the device compilation (the host pass prints "not a recognized feature" for
# it, filtered below); tests/test_abi_cpu.py disassembles the built library and fails on any packed-fp32 opcode.
NO_PACKED_FP32 = ['-Xclang', '-target-feature', '-Xclang', '-packed-fp32-ops']
FLAGS = ['-O3', '-std=c++17', '-fPIC', f'--offload-arch={ARCH}', '-Wall', '-Wno-unused-function', '-Wno-pass-failed'] + NO_PACKED_FP32


def _quiet(out: str) -> str:
    """hipcc's output without the host pass's note about the device-only target feature."""
    return '\n'.join(l for l in out.splitlines() if "'-packed-fp32-ops' is not a recognized feature" not in l).strip()


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', shutil.which('hipcc')):
        if c and os.path.exists(c):
            return c
    raise RuntimeError('hipcc not found (looked at $HIPCC, /opt/rocm/bin/hipcc, PATH)')


def _layout_tag():
    """32-bit tag of the files that define the packed weight-stream layout; engine files carry it (pnrf_mlp_serialize)."""
    import zlib
    c = 0
    for f in ('pnrf_layout.h', 'pnrf_pack.hip'):
        c = zlib.crc32(open(os.path.join(CSRC, f), 'rb').read(), c)
    return c & 0xffffffff


TRAINER_ONLY = ('pnrf_train.hip', 'pnrf_tchain.h', 'pnrf_hgemm.h')     # sources no inference kernel is built from
INFERENCE_ONLY = ('pnrf_mlp_kernels.hip',)                           # ... and the one no trainer kernel is built from (the trainer has its own chains)


def _code_only(text: str) -> str:
    """C / C++ source without comments and with runs of white space folded: what the compiler sees.  String and character literals are kept verbatim."""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c == '/' and i + 1 < n and text[i + 1] == '/':
            while i < n and text[i] != '\n':
                i += 1
        elif c == '/' and i + 1 < n and text[i + 1] == '*':
            j = text.find('*/', i + 2)
            i = n if j < 0 else j + 2
            out.append(' ')
        elif c in '"\'':
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == '\\' else 1
            out.append(text[i:j + 1])
            i = j + 1
        else:
            out.append(c)
            i += 1
    return ' '.join(''.join(out).split())


def _digest(scope='all'):
    """sha256 over the kernel sources + build flags.  scope 'all': every file, byte for byte (the library's .sha256 stamp);
    'inference': without the trainer-only sources — what the rendering kernels are built from (tools/profile_round.sh records it, bench.py
    compares it before quoting a profile), so that work on the trainer does not void the frame's PMC profile; 'training': without the fused
    inference kernels' source, for the trainer's profile (tools/profile_train.sh).  The two profile scopes hash the CODE — comments stripped, white
    space folded (_code_only) — so that a comment in the header does not void a profile of kernels it did not change (round 5: twice)."""
    h = hashlib.sha256()
    files = sorted(os.listdir(CSRC)) + ['../../include/pronerf_hip.h']
    for f in files:
        p = os.path.join(CSRC, f)
        if os.path.isfile(p) and not (scope == 'inference' and f in TRAINER_ONLY) and not (scope == 'training' and f in INFERENCE_ONLY):
            h.update(f.encode())
            h.update(open(p, 'rb').read() if scope == 'all' else _code_only(open(p, encoding='utf-8', errors='replace').read()).encode())
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    stamp = LIB + '.sha256'
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return LIB
    hipcc = _hipcc()
    objs = []
    procs = []
    for s in SOURCES:
        o = os.path.join(LIBDIR, s.replace('.hip', '.o'))
        objs.append(o)
        cmd = [hipcc] + FLAGS + [f'-DPNRF_LAYOUT_TAG=0x{_layout_tag():08x}u', '-I', INCLUDE, '-c', os.path.join(CSRC, s), '-o', o]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f'hipcc failed on {s}:\n{out}')
        if verbose and _quiet(out):
            print(_quiet(out))
    cmd = [hipcc, '-shared', '-fPIC', f'--offload-arch={ARCH}', '-o', LIB] + objs + LINK
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    for o in objs:
        os.remove(o)
    open(stamp, 'w').write(dig)
    return LIB


def device_code_objects(lib: str = LIB):
    """The gfx950 code objects inside a built library: clang offload bundles ("__CLANG_OFFLOAD_BUNDLE__": entry count, then per entry offset / size /
    triple) found in the file, one per translation unit.  -> list of bytes.  (roc-obj-ls needs a perl module this image lacks.)"""
    import struct
    data = open(lib, 'rb').read()
    magic = b'__CLANG_OFFLOAD_BUNDLE__'
    out = []
    pos = data.find(magic)
    while pos >= 0:
        n = struct.unpack_from('<Q', data, pos + len(magic))[0]
        q = pos + len(magic) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from('<QQQ', data, q)
            triple = data[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if ARCH in triple and size:
                out.append(data[pos + off:pos + off + size])
        pos = data.find(magic, pos + len(magic))
    return out


_DISASM = {}


def _disassembly(lib: str = LIB):
    """[(llvm-objdump -d -C text, llvm-readelf --notes text)] per gfx950 code object of the library; cached per (path, mtime) — the two consumers below
    (and the tests that call both) disassemble 30 MB of text once."""
    import tempfile
    key = (os.path.abspath(lib), os.path.getmtime(lib))
    if key not in _DISASM:
        llvm = os.path.join(os.path.dirname(_hipcc()), '..', 'lib', 'llvm', 'bin')
        if not os.path.exists(os.path.join(llvm, 'llvm-objdump')):
            llvm = '/opt/rocm/lib/llvm/bin'
        out = []
        for co in device_code_objects(lib):
            with tempfile.NamedTemporaryFile(suffix='.co') as f:
                f.write(co); f.flush()
                dis = subprocess.run([os.path.join(llvm, 'llvm-objdump'), '-d', f'--mcpu={ARCH}', '-C', f.name], stdout=subprocess.PIPE, text=True, check=True).stdout
                notes = subprocess.run([os.path.join(llvm, 'llvm-readelf'), '--notes', f.name], stdout=subprocess.PIPE, text=True, check=True).stdout
            out.append((dis, notes))
        _DISASM.clear()
        _DISASM[key] = out
    return _DISASM[key]


def device_opcodes(lib: str = LIB):
    """Counter of the instruction mnemonics in every gfx950 code object of the library (llvm-objdump -d)."""
    import collections
    ops = collections.Counter()
    for txt, _ in _disassembly(lib):
        for line in txt.splitlines():
            t = line.split()
            if len(t) >= 2 and line.startswith('\t') and (t[0].startswith(('v_', 's_', 'ds_', 'global_', 'buffer_', 'flat_', 'scratch_'))):
                ops[t[0]] += 1
    return ops


def device_kernels(lib: str = LIB):
    """Static figures of every kernel in the library's gfx950 code objects: demangled name -> {'instr', 'mfma' (counts in the disassembly), 'vgpr',
    'scratch' (bytes of private segment: spills), 'lds' (static bytes), 'vmov' (registers moved by v_mov_b32 / v_mov_b64 between VGPRs: a cluster of
    them in a fused-MLP kernel is a copy of its operand registers)} from llvm-objdump and the kernel descriptors' notes (llvm-readelf)."""
    import collections
    import re
    out = {}
    for dis, notes in _disassembly(lib):
        cur, n, mf, mv = None, collections.Counter(), collections.Counter(), collections.Counter()
        for line in dis.splitlines():
            m = re.match(r'^[0-9a-f]+ <(.+)>:', line)
            if m:
                cur = m.group(1)
            elif cur and line.startswith('\t'):
                n[cur] += 1
                mf[cur] += 'v_mfma' in line
                mm = re.match(r'\tv_mov_b(32|64)_e32 v\[?\d+(?::\d+\])?, v', line)
                if mm:
                    mv[cur] += 2 if mm.group(1) == '64' else 1
        meta = {}
        for blk in re.split(r'\n\s+- \.agpr_count', notes)[1:]:
            g = lambda k: (re.search(rf'\.{k}:\s+(\S+)', blk) or [None, '-1'])[1]
            meta[g('symbol').replace('.kd', '')] = (int(g('vgpr_count')), int(g('private_segment_fixed_size')), int(g('group_segment_fixed_size')))
        names = list(meta)
        dem = subprocess.run(['c++filt'] + names, stdout=subprocess.PIPE, text=True).stdout.split('\n') if names else []
        for sym, d in zip(names, dem):
            k = d if d in n else sym
            out[k.replace('(anonymous namespace)::', '')] = {'instr': n.get(k, 0), 'mfma': mf.get(k, 0), 'vgpr': meta[sym][0], 'scratch': meta[sym][1], 'lds': meta[sym][2],
                                                             'vmov': mv.get(k, 0)}
    return out


PACKED_FP32_OPCODES = ('v_pk_mul_f32', 'v_pk_add_f32', 'v_pk_fma_f32', 'v_pk_mov_b32')


CEILING_SRC = os.path.join(os.path.dirname(HERE), 'tools', 'mfma_ceiling.hip')
CEILING_BIN = os.path.join(LIBDIR, 'mfma_ceiling')


def build_ceiling_probe(force: bool = False) -> str:
    """tools/mfma_ceiling.hip -> pronerf_amd/lib/mfma_ceiling: the pure-MFMA loop bench.py runs to report what the pipes sustain at the chip's
    power limit (roofline.sustained).  Not part of the product."""
    os.makedirs(LIBDIR, exist_ok=True)
    if not os.path.exists(CEILING_SRC):
        return ''
    if force or not os.path.exists(CEILING_BIN) or os.path.getmtime(CEILING_BIN) < os.path.getmtime(CEILING_SRC):
        subprocess.run([_hipcc(), '-O3', f'--offload-arch={ARCH}', CEILING_SRC, '-o', CEILING_BIN], check=True)
    return CEILING_BIN


FOREIGN_SRC = os.path.join(os.path.dirname(HERE), 'tools', 'foreign_kernels.hip')
FOREIGN_LIB = os.path.join(LIBDIR, 'libforeign_kernels.so')


def build_foreign_kernels(force: bool = False) -> str:
    """tools/foreign_kernels.hip -> pronerf_amd/lib/libforeign_kernels.so: the stand-in kernels of other streams that the co-residency stress
    test (tests/test_coresidency_gpu.py, tools/coresidency_stress.py) runs beside the renderer.  Test infrastructure, not part of the product."""
    os.makedirs(LIBDIR, exist_ok=True)
    if not os.path.exists(FOREIGN_SRC):
        return ''
    if force or not os.path.exists(FOREIGN_LIB) or os.path.getmtime(FOREIGN_LIB) < os.path.getmtime(FOREIGN_SRC):
        subprocess.run([_hipcc(), '-O3', f'--offload-arch={ARCH}', '-shared', '-fPIC', FOREIGN_SRC, '-o', FOREIGN_LIB], check=True)
    return FOREIGN_LIB


def build_variant(name: str, extra_flags=(), csrc: str = CSRC, include: str = INCLUDE) -> str:
    """Build pronerf_amd/lib/libpronerf_hip_<name>.so with extra compiler flags (A/B timing, diagnostics)."""
    os.makedirs(LIBDIR, exist_ok=True)
    out = os.path.join(LIBDIR, f'libpronerf_hip_{name}.so')
    srcs = [os.path.join(csrc, s) for s in SOURCES]
    cmd = [_hipcc()] + FLAGS + list(extra_flags) + ['-shared', '-I', include, '-o', out] + srcs + LINK
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if _quiet(r.stdout):
        print(_quiet(r.stdout))
    if r.returncode != 0:
        raise RuntimeError(f'hipcc failed building variant {name} (exit {r.returncode}); last lines of its output:\n' + '\n'.join(r.stdout.splitlines()[-30:]))
    return out


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == '--variant':
        # '+packed-fp32' as an extra flag re-enables the packed-fp32 instructions (reproducer builds of tools/coresidency_stage.py)
        extra = [f for f in sys.argv[3:] if f != '+packed-fp32']
        if '+packed-fp32' in sys.argv[3:]:
            extra += ['-Xclang', '-target-feature', '-Xclang', '+packed-fp32-ops']
        print(build_variant(sys.argv[2], extra))
    else:
        print(build(force='--force' in sys.argv))
        print(build_ceiling_probe(force='--force' in sys.argv))
        print(build_foreign_kernels(force='--force' in sys.argv))
