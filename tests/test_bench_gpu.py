"""GPU: bench.py's JSON contract, on small step counts — the N = 1 line with `roofline`, `cpu_baseline` and `gpu_eager_baseline`, and the
N = 2 path (two ranks sharing the one GPU of the box, `gloo` instead of RCCL) so that the multi-rank code cannot rot between the
driver's multi-GPU runs.  bench.py runs as a fresh child process, as the driver starts it."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(out):
    lines = [l for l in out.strip().splitlines() if l.startswith('{')]
    assert lines, out
    return json.loads(lines[-1])


QUICK = ['--no-cpu-baseline', '--no-gpu-eager-baseline', '--no-sustained', '--no-chunked', '--no-variants', '--no-shard-rehearsal', '--steady-seconds', '0',
         '--no-train', '--no-optimizer-weights']
_N1 = {}


def n1_frame_digest():
    """sha256 of the frame the N = 1 bench renders (bench.py --gpus 1, every extra block off): what every N > 1 line's assembled frame must hash to."""
    if 'sha' not in _N1:
        env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}
        r = subprocess.run([sys.executable, 'bench.py', '--steps', '2', '--warmup', '1'] + QUICK, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        _N1['sha'] = _last_json(r.stdout)['frame_sha256']
        assert len(_N1['sha']) == 64
    return _N1['sha']


def check_per_rank(j, n):
    pr = j['per_rank']
    assert [p['rank'] for p in pr] == list(range(n)) and [p['rays'] for p in pr] == j['config']['rays_per_rank']
    assert all(p['render_ms'] > 0 and p['gather_ms'] > 0 and 0 <= p['rays_second_pass'] <= p['rays'] for p in pr)


def test_bench_single_gpu_line():
    r = subprocess.run([sys.executable, 'bench.py', '--steps', '5', '--warmup', '3', '--cpu-sample-rays', '8192', '--eager-reps', '2'], cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _last_json(r.stdout)
    assert j['n_gpus'] == 1 and j['steps'] == 5 and j['warmup'] == 3 and j['unit'] == 'rays/s' and j['higher_is_better'] is True
    assert j['outputs_finite'] and j['value'] > 1e7 and abs(j['value'] - 762048 / (j['ms_per_step'] * 1e-3)) < 1e-3 * j['value']
    assert 'workload' in j['config'] and 'model' not in j['config']
    rf = j['roofline']
    assert rf['bound'] == 'mfma' and rf['unit'] == 'TFLOP/s' and 0 < rf['frac'] < 1 and abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-9
    assert sum(k['ms'] for k in j['kernels'].values()) <= j['ms_per_step'] * 1.02            # the kernels add up to (at most) the frame
    cb = j['cpu_baseline']
    assert cb['kind'] == 'port' and cb['cores'] >= 1 and cb['value'] > 0
    ge = j['gpu_eager_baseline']
    assert ge['rays'] == 762048 and ge['value'] > 0 and ge['hip_vs_eager_rgb_psnr_db'] > 46.4
    assert abs(j['vs_baseline'] - j['value'] / ge['value']) < 1e-6 * j['vs_baseline'] and j['vs_baseline'] > 10.0      # BASELINE.json: >= 10x
    wo = j['weights_optimizer']                                    # the hard weights are timed too
    assert wo['ms_per_frame'] > 0 and 0 < wo['sampler_two_pass']['fraction'] < 1 and wo['hip_vs_eager_rgb_psnr_db'] > 46.4
    assert len(j['frame_sha256']) == 64 and j['per_rank'] is None


def test_bench_two_ranks_on_one_gpu_gloo():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port', '29517',
           'bench.py', '--gpus', '2', '--backend', 'gloo', '--steps', '3', '--warmup', '1', '--no-cpu-baseline']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _last_json(r.stdout)
    assert j['n_gpus'] == 2 and j['scaling'] == 'strong' and j['steps'] == 3 and j['backend'] == 'gloo'
    assert j['outputs_finite'] and j['value'] > 0 and j['config']['rays_per_rank'] == [381120, 380928] and j['config']['ray_partition'].startswith('cyclic')
    assert 'roofline' not in j and 'cpu_baseline' not in j            # N = 1 only
    check_per_rank(j, 2)
    assert j['frame_sha256'] == n1_frame_digest()                     # shards + gather + reorder change no byte of the frame


def test_bench_two_ranks_without_a_launcher():
    """`python3 bench.py --gpus 2` as the driver starts the N = 1 bench (no torch.distributed.run): bench.py starts its own ranks before
    anything touches the GPU and relays rank 0's line."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}
    cmd = [sys.executable, 'bench.py', '--gpus', '2', '--backend', 'gloo', '--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--partition', 'contiguous']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _last_json(r.stdout)
    assert j['n_gpus'] == 2 and j['ranks'] == 2 and j['backend'] == 'gloo' and j['launcher'].startswith('self')
    assert j['config']['rays_per_rank'] == [381024, 381024] and j['config']['gather_bytes_per_rank_per_frame'] == 381024 * 16
    assert j['config']['ray_partition'] == 'contiguous'
    assert j['outputs_finite'] and j['value'] > 0
    check_per_rank(j, 2)
    assert j['frame_sha256'] == n1_frame_digest()


def test_bench_nccl_needs_one_gpu_per_rank():
    """--gpus 2 with the RCCL backend on a one-GPU box: refused with a clear message and a non-zero exit code, nothing started."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip('box has several GPUs')
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE')}
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '2', '--steps', '1', '--warmup', '0'], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and 'needs 2 visible GPUs' in r.stderr
