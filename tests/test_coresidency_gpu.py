"""GPU: the shipped fused launches beside foreign kernels that can share their CUs — every row of every call compared (tools/coresidency_stress.py).

Round 4's hazard (NOTEBOOK §12, §19: packed-fp32 VALU instructions of a wave in FP16_OVFL mode beside a wave issuing 16x16x32 bf16 MFMAs, both with 240
registers) showed with two fused-MLP workgroups on one CU, which the launcher excludes, and the library no longer contains the instruction class; what can
still become resident beside a fused workgroup is a kernel of another stream — at N > 1 the RCCL all-gather and the index_select of FrameGather, or somebody's
bf16 GEMM.  This test runs that configuration on the one-GPU box: narrow 1024-ray calls beside gather / LDS-DMA / small foreign kernels and beside a bf16
16x16x32 MFMA loop with 240 registers per wave (tools/foreign_kernels.hip), wide calls beside the two kinds that fit beside 2 x 240 registers, the chunked
frame on four streams, and whole frames through a one-rank RCCL FrameGather with a permutation index.  The long run (10^5 calls per narrow phase) is
profiles/r05_coresidency_stress.json; here ~6000 calls keep the suite short.

Round 6 (NOTEBOOK 22): the WIDE refine stage did return wrong rows beside the 1-wave small kernel — 2e-4 of the 8192-ray calls, always the slower half of a
workgroup's waves — while its two waves per SIMD left 32 registers for a third.  Wide fused kernels now allocate the whole register file (256 per wave,
checked statically by tests/test_abi_cpu.py): the phases `wide_8192_beside_small` / `narrow_1024_beside_small` run beside three streams of that kernel;
profiles/r06_wide_repro.txt holds the long runs (0 of 450 000 wide calls with the reservation; 106 of 200 000 without)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_row_changes_beside_foreign_kernels_and_the_rccl_gather():
    assert os.path.exists(os.path.join(ROOT, 'pronerf_amd', 'lib', 'libforeign_kernels.so')), 'run __graft_entry__.build() first (pronerf_amd.build.build_foreign_kernels)'
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE')}
    env.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    calls = int(os.environ.get('PNRF_STRESS_CALLS', '6000'))
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'coresidency_stress.py'), '--calls', str(calls)], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert lines, (r.returncode, r.stdout[-1000:], r.stderr[-3000:])
    j = json.loads(lines[-1])
    for phase in ('narrow_1024', 'narrow_1024_beside_mfma', 'narrow_1024_beside_small', 'wide_8192', 'wide_8192_beside_small', 'chunked_4_streams', 'frame_gather_rccl_ws1'):
        assert j[phase]['rows_differ'] == 0 and j[phase]['rows_compared'] > 0, (phase, j[phase])
    assert j['narrow_1024']['calls'] == calls and j['frame_gather_rccl_ws1']['backend'] == 'nccl'
    assert r.returncode == 0 and j['total_rows_differ'] == 0
