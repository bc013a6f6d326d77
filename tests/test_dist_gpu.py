"""GPU: the RCCL launch path on a one-GPU box.  bench.py at N > 1 and the sharded frame driver issue `all_gather_into_tensor` on device tensors
through `pronerf_amd.dist.FrameGather` with backend 'nccl' (= RCCL on ROCm); the multi-rank logic is covered with gloo ranks
(tests/test_dist_cpu.py, tests/test_bench_gpu.py), this test loads RCCL itself: a process group of ONE rank on cuda:0, frames rendered by the
HIP path and gathered by RCCL on its own stream while the next frame renders (the one exchange step of the path)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_all_gather_through_frame_gather_world_size_one():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE')}
    env.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'rccl_worker.py')], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert j['backend'] == 'nccl' and j['world'] == 1
    assert j['frames_equal'] == [True] * 5 and j['reordered_frames_equal'] == [True] * 5 and j['plain_collectives_ok']
