"""CPU (gloo, world_size 2 and 3): the ray-sharding + gather logic of pronerf_amd.dist.

The per-range renderer injected here is the CPU oracle (tests may use it as the checker); on the
GPU ranks it is Renderer.render_rays.  What is tested is the N>1 path: contiguous ranges cover
the frame exactly once, ragged splits, gather ordering, root-only gather."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pronerf_amd.render import RayPartition, shard_range


def test_shard_range_partitions():
    for n in (0, 1, 7, 8, 762048, 762049, 1000003):
        for ws in (1, 2, 3, 8):
            pos = 0
            for r in range(ws):
                f, c = shard_range(n, r, ws)
                assert f == pos and c >= 0
                pos += c
            assert pos == n
            sizes = [shard_range(n, r, ws)[1] for r in range(ws)]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(762048, 7, 8) == (7 * 95256, 95256)


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, ws, port, H, W, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=ws)
    try:
        from oracle import pronerf_oracle as orc
        from oracle import synth
        from pronerf_amd.dist import render_frame_sharded
        scene = synth.make_scene(0, H=H, W=W)
        weights = synth.make_weights(0, 'trained')
        fr = orc.frame_setup(scene)

        def render_fn(first, count):
            o = orc.render_rays_infer(weights, fr['rays'][first:first + count], fr['or_rays'][first:first + count], fr['images'], fr['proj'])
            return torch.cat([o['rgb'], o['depth'][:, None]], 1)

        full = render_frame_sharded(render_fn, H * W, gather='all')
        root = render_frame_sharded(render_fn, H * W, gather='root')
        q.put((rank, full.numpy(), None if root is None else root.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('ws,H,W', [(2, 6, 8), (3, 5, 7)])       # 48 rays / 2 (even) and 35 rays / 3 (ragged)
def test_sharded_render_matches_single_process(ws, H, W):
    from oracle import pronerf_oracle as orc
    from oracle import synth
    scene = synth.make_scene(0, H=H, W=W)
    fr = orc.frame_setup(scene)
    o = orc.render_rays_infer(synth.make_weights(0, 'trained'), fr['rays'], fr['or_rays'], fr['images'], fr['proj'])
    ref = torch.cat([o['rgb'], o['depth'][:, None]], 1).numpy()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, ws, port, H, W, q)) for r in range(ws)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(ws)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, full, root in res:
        np.testing.assert_allclose(full, ref, rtol=0, atol=2e-5)       # CPU BLAS rounding depends on the batch size
        if rank == 0:
            np.testing.assert_allclose(root, ref, rtol=0, atol=2e-5)
        else:
            assert root is None


def test_ray_partition_properties():
    """RayPartition: both kinds deal every ray exactly once; cyclic blocks of 1024 rays give the Fern frame 95 424 + 7 x 95 232 rays; the gather
    index rebuilds the frame from the rank-major gathered buffer; frame_rays_args describe the same rows."""
    for n, w, blk in ((762048, 8, 1024), (1000, 3, 64), (130, 4, 32), (64, 8, 16), (5, 8, 4), (762048, 2, 1024), (97, 1, 16), (768, 2, 1024)):
        for kind in ('contiguous', 'cyclic'):
            p = RayPartition(n, w, kind, blk)
            rows = [p.rows(r) for r in range(w)]
            assert [len(x) for x in rows] == p.counts and sum(p.counts) == n
            assert sorted(torch.cat(rows).tolist()) == list(range(n))
            buf = torch.full((w * p.cmax,), -1, dtype=torch.int64)
            for r in range(w):
                buf[r * p.cmax: r * p.cmax + p.counts[r]] = rows[r]
                a = p.frame_rays_args(r)
                q = torch.arange(a['count'])
                pix = a['first'] + q if 'block' not in a else a['first'] + (q // a['block']) * a['stride'] + q % a['block']
                assert torch.equal(pix, rows[r])
            gi = p.gather_index()
            assert torch.equal(buf if gi is None else buf[gi], torch.arange(n))
    assert RayPartition(762048, 8, 'cyclic').counts == [95424] + [95232] * 7
    assert RayPartition(762048, 8, 'contiguous').counts == [95256] * 8 and RayPartition(762048, 8, 'contiguous').gather_index() is None
    assert RayPartition(100, 1, 'cyclic').kind == 'contiguous'                      # one rank: the whole frame
    with pytest.raises(ValueError):
        RayPartition(10, 2, 'striped')


def test_ray_partition_properties_random_sizes():
    """The same properties on drawn (n, world, block, kind): every ray exactly once, counts consistent, the gather index inverts the rank-major buffer."""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=120, deadline=None)
    @given(n=st.integers(1, 5000), w=st.integers(1, 9), blk=st.sampled_from([4, 32, 64, 1024]), kind=st.sampled_from(['contiguous', 'cyclic']))
    def check(n, w, blk, kind):
        p = RayPartition(n, w, kind, blk)
        rows = [p.rows(r) for r in range(w)]
        assert [len(x) for x in rows] == p.counts and sum(p.counts) == n and max(p.counts) == p.cmax
        assert sorted(torch.cat(rows).tolist()) == list(range(n))
        buf = torch.full((w * p.cmax,), -1, dtype=torch.int64)
        for r in range(w):
            buf[r * p.cmax: r * p.cmax + p.counts[r]] = rows[r]
            a = p.frame_rays_args(r)
            q = torch.arange(a['count'])
            pix = a['first'] + q if 'block' not in a else a['first'] + (q // a['block']) * a['stride'] + q % a['block']
            assert torch.equal(pix, rows[r])
        gi = p.gather_index()
        assert torch.equal(buf if gi is None else buf[gi], torch.arange(n))

    check()


def _gather_worker(rank, ws, port, n_total, frames, pipelined, q, kind='contiguous'):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=ws)
    try:
        from pronerf_amd.dist import FrameGather
        part = RayPartition(n_total, ws, kind, 32)
        fg = FrameGather(n_total, 4, device='cpu', pipelined=pipelined, partition=part)
        assert fg.count == part.count(rank) and fg.depth == (2 if pipelined else 1)
        if kind == 'contiguous':
            assert (fg.first, fg.count) == shard_range(n_total, rank, ws)
        rows = part.rows(rank).to(torch.float32)[:, None] + torch.tensor([0., .25, .5, .75])
        got = []
        for f in range(frames):                       # frame f: pixel value = 1000 f + global row (+ channel / 4)
            b = fg.acquire()
            fg.outs[b][:fg.count] = rows + 1000. * f
            fg.submit(b)
            if f >= 1 and pipelined:                  # the previous frame sits in the other buffer, complete or in flight
                got.append(fg.frame(1 - b).clone())
        fg.fence()
        last = (frames - 1) % fg.depth
        got.append(fg.frame(last).clone())
        q.put((rank, [g.numpy() for g in got]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('ws,n_total,pipelined,kind', [(2, 48, True, 'contiguous'), (3, 35, True, 'contiguous'), (2, 35, False, 'contiguous'),
                                                       (3, 135, True, 'cyclic'), (2, 151, False, 'cyclic')])
def test_frame_gather_pipeline(ws, n_total, pipelined, kind):
    """pronerf_amd.dist.FrameGather (what bench.py times at N > 1): two buffers in flight, async all-gather, ragged shards; contiguous
    ranges and the block-cyclic partition (blocks of 32 rays here), whose frame is rebuilt through the partition's gather index."""
    frames = 5
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, ws, port, n_total, frames, pipelined, q, kind)) for r in range(ws)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(ws)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.arange(n_total, dtype=np.float32)[:, None] + np.array([0., .25, .5, .75], dtype=np.float32)
    for rank, got in res:
        seen = list(range(frames)) if pipelined else [frames - 1]
        assert len(got) == len(seen)
        for f, g in zip(seen, got):
            np.testing.assert_array_equal(g, want + 1000. * f)
