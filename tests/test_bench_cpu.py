"""CPU: bench.py's launcher for N > 1 (no torch.distributed.run).  Without a GPU the ranks cannot run; what is checked here is that the
parent refuses / fails loudly with a non-zero exit code instead of printing a line."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(torch.cuda.device_count() > 0, reason='CPU-only checks')


def _env():
    return {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}


def test_self_launch_refuses_rccl_without_enough_gpus():
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '2'], cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and 'needs 2 visible GPUs' in r.stderr and not r.stdout.strip()


def test_self_launch_reports_a_failed_rank():
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '2', '--backend', 'gloo', '--launch-timeout', '120'], cwd=ROOT, env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 1 and 'stopping the other ranks' in r.stderr and not r.stdout.strip()


def test_world_size_mismatch_is_an_error():
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '2'], cwd=ROOT, env=dict(_env(), WORLD_SIZE='4', RANK='0'), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and 'does not match' in r.stderr
