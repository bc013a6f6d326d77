#!/usr/bin/env python3
"""Times the training iterations of tests/perf_train_gpu.py's workload against several library builds, each in its own child process
(python tools/train_ab.py <variant> [<variant> ...]; '' = the default build)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = '''
import ctypes as C, os, sys
sys.path.insert(0, %r)
from pronerf_amd import _lib
name = %r
if name:
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'libpronerf_hip_' + name + '.so')
sys.argv = ['perf_train_gpu.py', '--hip-only']
__file__ = os.path.join(%r, 'tests', 'perf_train_gpu.py')
exec(open(__file__).read())
'''
for v in sys.argv[1:] or ['']:
    r = subprocess.run([sys.executable, '-c', code % (ROOT, v, ROOT)], capture_output=True, text=True)
    print(repr(v), r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-500:])
