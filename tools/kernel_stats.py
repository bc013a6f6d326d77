#!/usr/bin/env python3
"""Per-kernel static figures of the built library's gfx950 code objects (pronerf_amd.build.device_kernels): instruction count, MFMA count, VGPRs, scratch
(spill) bytes, static LDS.     python tools/kernel_stats.py [--lib=path] [name substring ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pronerf_amd import build

lib = next((a[6:] for a in sys.argv[1:] if a.startswith('--lib=')), build.LIB)
flt = [a for a in sys.argv[1:] if not a.startswith('--')]
for k, v in sorted(build.device_kernels(lib).items()):
    if not flt or any(f in k for f in flt):
        print(f"{v['instr']:7d} instr {v['mfma']:5d} mfma  vgpr {v['vgpr']:4d} scratch {v['scratch']:5d} lds {v['lds']:6d} vmov {v['vmov']:4d}  {k[:130]}")
