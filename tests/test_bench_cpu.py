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


def test_profile_numbers_are_refused_when_the_kernel_sources_changed():
    """roofline.traffic / train.hbm_bytes_per_iter come from committed rocprofv3 --pmc summaries: quoted only while the summary's csrc digest
    equals this tree's (pronerf_amd.build._digest), otherwise null with the reason."""
    sys.path.insert(0, ROOT)
    import bench
    now = bench.csrc_digest()
    assert isinstance(now, str) and len(now) == 64
    ok, prov = bench.profile_provenance({'csrc_digest': now, 'commit': 'abc1234', '_file': 'profiles/x.json'})
    assert ok and prov == {'file': 'profiles/x.json', 'commit': 'abc1234', 'csrc_digest': now}
    ok, prov = bench.profile_provenance({'csrc_digest': '0' * 64, 'commit': 'abc1234'})
    assert not ok and 'changed since the profile' in prov['refused']
    ok, prov = bench.profile_provenance({'round': 'r03_v3'})
    assert not ok and 'predates' in prov['refused']
    val, prov = bench.pmc_traffic('nerf_kernel')              # whatever is committed: either a byte count with its source, or a reason
    assert (val is None and 'refused' in prov) or (val > 0 and prov['csrc_digest'] == now)
    # the frame's profile is keyed to the sources the rendering kernels are built from; the trainer's to everything
    from pronerf_amd import build
    assert now == build._digest('inference') != build._digest('all') == bench.csrc_digest('all') != build._digest('training')
    # the profile scopes hash code, not comments or layout: a comment in the header must not void a profile of kernels it did not change
    assert build._code_only('int a = 1; // c\n/* x\n y */ const char* s = "//not /*c*/"; char q = \'"\';\n\n  int   b;') == \
        'int a = 1; const char* s = "//not /*c*/"; char q = \'"\'; int b;'
    assert build._code_only('x = 1; /* a */') == build._code_only('x = 1;   // b') != build._code_only('x = 2;')
