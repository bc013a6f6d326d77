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
import torch

pytestmark = pytest.mark.gpu
NDEV = torch.cuda.device_count()          # counting devices does not initialise the GPU (this process only starts children)

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


# ---- real ranks: these tests switch themselves on the moment the box has >= 2 GPUs (the driver's 8-GPU node); on the one-GPU box of a round
# they are collected and skipped, and their gloo twins (tests/test_bench_gpu.py, tests/test_mirror_gpu.py) run instead.
def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}
    env.update(MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    return env


@pytest.mark.skipif(NDEV < 2, reason='needs >= 2 GPUs (RCCL with one rank per GPU)')
@pytest.mark.parametrize('n', sorted({2, NDEV} & set(range(2, NDEV + 1))) or [2])
@pytest.mark.parametrize('launcher', ['torchrun', 'self'])
def test_bench_on_real_rccl_ranks(n, launcher):
    """bench.py --gpus n with the nccl backend, started as the driver starts it (torch.distributed.run) and without a launcher: RCCL group of n ranks,
    one GPU each, and the assembled frame of the last timed step hashes to the N = 1 frame's bytes; the line explains itself (per-rank render / gather ms)."""
    from test_bench_gpu import QUICK, _last_json, check_per_rank, n1_frame_digest
    base = ['bench.py', '--gpus', str(n), '--steps', '5', '--warmup', '3'] + QUICK
    if launcher == 'torchrun':
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1', '--master-port', str(_free_port())] + base
    else:
        cmd = [sys.executable] + base
    r = subprocess.run(cmd, cwd=ROOT, env=_clean_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    j = _last_json(r.stdout)
    assert j['n_gpus'] == n and j['backend'] == 'nccl' and j['rccl_ranks'] == n and j['scaling'] == 'strong' and j['outputs_finite']
    check_per_rank(j, n)
    assert len({p['device'].split(' ')[0] for p in j['per_rank']}) == n           # one GPU per rank
    assert j['frame_sha256'] == n1_frame_digest()
    print(f"\n[rccl x{n}, {launcher}] {j['ms_per_step']:.3f} ms/frame, {j['value'] / 1e6:.1f} M rays/s; per rank render "
          f"{[round(p['render_ms'], 3) for p in j['per_rank']]} ms, gather {[round(p['gather_ms'], 3) for p in j['per_rank']]} ms")


@pytest.mark.skipif(NDEV < 2, reason='needs >= 2 GPUs (RCCL with one rank per GPU)')
def test_frame_driver_under_torchrun_on_real_rccl_ranks(tmp_path):
    """The inference script under torchrun with min(4, #GPUs) RCCL ranks: every frame's rays sharded block-cyclically, gathered by RCCL; rank 0 writes the same
    PNG bytes as the single-process run (the reference's harness, run_S_eS_eN_alter_trt.py:327-332, 719-721, on real ranks)."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import llff_synth
    from pronerf_amd import synthetic as synth
    n = min(4, NDEV)
    root = llff_synth.make_dataset(str(tmp_path / 'scene'), seed=1, n=10, H=48, W=64, factor=4)
    sds = synth.state_dicts(synth.make_weights(0, 'trained'))
    ck = str(tmp_path / '000123.tar')
    torch.save({'global_step': 123, 'mmr_network_fn_state_dict': sds['sampler'], 'refine_net_state_dict': sds['refine'], 'network_fine_state_dict': sds['nerf']}, ck)
    body = (f'basedir = {tmp_path}/logs\ndatadir = {root}\nft_path = {ck}\nfactor = 4\nllffhold = 8\nN_samples = 8\nN_point_ray_enc = 48\n'
            'mmnetdepth = 6\nmmnetskips = [10000]\nnum_neighbor = 4\nuse_viewdirs = True\n')
    (tmp_path / 'one.txt').write_text('expname = one\n' + body)
    (tmp_path / 'many.txt').write_text('expname = many\n' + body)
    env = dict(_clean_env(), PYTHONPATH=ROOT)
    r1 = subprocess.run([sys.executable, '-m', 'pronerf_amd.run_S_eS_eN_alter_trt', '--config', str(tmp_path / 'one.txt'), '--render_test'], cwd=ROOT, env=env,
                        capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-3000:]
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1', '--master-port',
                        str(_free_port()), '-m', 'pronerf_amd.run_S_eS_eN_alter_trt', '--config', str(tmp_path / 'many.txt'), '--render_test'],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d1, d2 = tmp_path / 'logs' / 'one' / 'renderonly_test_000123', tmp_path / 'logs' / 'many' / 'renderonly_test_000123'
    assert sorted(os.listdir(d1)) == sorted(os.listdir(d2)) == ['000.png', '001.png', 'depth_000.png', 'depth_001.png']
    for f in os.listdir(d1):
        assert (d1 / f).read_bytes() == (d2 / f).read_bytes(), f
    assert r.stdout.count('Mean Test PSNR') == 1
