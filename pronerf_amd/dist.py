"""Ray-sharded multi-GPU rendering: one process per GPU, contiguous flat ray ranges, one
all-gather of the packed [n,4] rgb+depth tiles per frame (RCCL over xGMI with backend 'nccl').

The reference has no distributed path; rays are independent given the weights and
the four neighbour images, so the only exchange step is the gather of finished
pixels.  1.5 MB per rank at 8 GPUs: a single small collective per frame, no ring of buckets.

``render_fn(first, count) -> Tensor[count, C]`` is whatever renders a contiguous ray range on this
rank (on a GPU rank: ``Renderer.render_rays`` over ``Renderer.frame_rays(first=..., count=...)``);
keeping it a callable lets the sharding/gather logic be exercised on CPU with the ``gloo``
backend (tests/test_dist_cpu.py).
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from .render import RayPartition, shard_range


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def render_frame_sharded(render_fn, n_total: int, out_channels: int = 4, device=None, dtype=torch.float32, gather: str = 'all'):
    """Render this rank's share of a frame of ``n_total`` rays and gather the frame.

    gather='all'  : every rank returns the full [n_total, C] frame (all_gather);
    gather='root' : rank 0 returns the frame, other ranks return None (gather to rank 0).
    Shard sizes differ by at most one ray; equal shards use a single all_gather_into_tensor.
    """
    rank, ws = world()
    first, count = shard_range(n_total, rank, ws)
    part = render_fn(first, count)
    if part.shape[0] != count:
        raise ValueError(f'render_fn returned {part.shape[0]} rows for a shard of {count} rays')
    if ws == 1:
        return part
    part = part.contiguous()
    device = part.device if device is None else device
    counts = [shard_range(n_total, r, ws)[1] for r in range(ws)]
    cmax = max(counts)
    if cmax != min(counts):          # ragged split (sizes differ by one ray): pad to a common size
        pad = torch.zeros(cmax, out_channels, device=device, dtype=dtype)
        pad[:count] = part
        part = pad
    if gather == 'root':
        buf = [torch.empty(cmax, out_channels, device=device, dtype=dtype) for _ in range(ws)] if rank == 0 else None
        dist.gather(part, buf, dst=0)
        return torch.cat([b[:c] for b, c in zip(buf, counts)], 0) if rank == 0 else None
    if cmax == min(counts):
        full = torch.empty(n_total, out_channels, device=device, dtype=dtype)
        dist.all_gather_into_tensor(full, part)
        return full
    buf = torch.empty(ws * cmax, out_channels, device=device, dtype=dtype)
    dist.all_gather_into_tensor(buf, part)
    return torch.cat([buf[r * cmax:r * cmax + c] for r, c in enumerate(counts)], 0)


class FrameGather:
    """The per-frame exchange step for a stream of frames: every rank renders its contiguous shard into ``acquire()``'s buffer and calls
    ``submit()``; the all-gather of that frame then runs on the collective's stream while the next frame renders into the other buffer
    (``depth`` buffers, ``async_op=True``; ``acquire`` waits — a stream wait on GPU backends, the host does not block — until the gather
    that last used a buffer is done).  Shards are padded to the largest one (they differ by at most one ray); ``frame(b)`` is the
    assembled [n_total, C] frame of buffer ``b`` (in pixel order: see ``_set_index``).  ``bench.py`` times exactly this at N > 1."""

    def __init__(self, n_total: int, channels: int = 4, device=None, dtype=torch.float32, depth: int = 2, pipelined: bool = True,
                 collective=None, partition=None):
        """collective: None = the all-gather runs when there is more than one rank; True = also in a one-rank process group (the RCCL launch
        path on a one-GPU box: tests/test_dist_gpu.py).  partition: a ``RayPartition`` (how the frame's rays are dealt to the ranks: this
        rank renders ``partition.frame_rays_args(rank)``) or 'contiguous' / 'cyclic'; None = contiguous ranges (``shard_range``)."""
        self.rank, self.world = world()
        self.n_total, self.channels = int(n_total), int(channels)
        if partition is None or isinstance(partition, str):
            partition = RayPartition(n_total, self.world, partition or 'contiguous')
        if partition.n_total != self.n_total or partition.world != self.world:
            raise ValueError('FrameGather: the partition is for another frame size / world size')
        self.partition = partition
        self.counts = list(partition.counts)
        self.first = shard_range(n_total, self.rank, self.world)[0] if partition.kind == 'contiguous' else None
        self.count = partition.count(self.rank)
        self.cmax = partition.cmax
        self.collective = self.world > 1 if collective is None else bool(collective)
        if self.collective and not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError('FrameGather(collective=True) needs an initialised process group')
        self.pipelined = bool(pipelined) and self.collective
        self.depth = depth if self.pipelined else 1
        self.outs = [torch.zeros(self.cmax, channels, device=device, dtype=dtype) for _ in range(self.depth)]
        self.fulls = [torch.empty(self.world * self.cmax, channels, device=device, dtype=dtype) if self.collective else None for _ in range(self.depth)]
        self.pending = [None] * self.depth
        self._next = 0
        self._set_index(partition.gather_index(device) if self.collective else None, device, dtype)

    def _set_index(self, index, device, dtype):
        """Frame order of the gathered buffer (None: it already is the frame).  With an index and a GPU collective (RCCL) the reorder is part of
        the frame's pipeline: one index_select into ``frames[b]`` on a side stream that waits for the gather — stream order, the host does not
        block — so that what ``bench.py`` times at N > 1 ends with the frame assembled in pixel order; other backends reorder in ``frame()``."""
        self._index = index
        self._eager = index is not None and self.pipelined and dist.get_backend() == 'nccl'
        self.frames = [torch.empty(self.n_total, self.channels, device=device, dtype=dtype) for _ in range(self.depth)] if self._eager else None
        self._side = torch.cuda.Stream(device=device) if self._eager else None
        self._done = [None] * self.depth

    def acquire(self) -> int:
        """Index of the buffer the next frame renders into (``outs[b][:count]``), free of any gather still reading it."""
        b = self._next
        self._next = (self._next + 1) % self.depth
        self._wait(b)
        return b

    def submit(self, b: int):
        if not self.collective:
            return
        if self.pipelined:
            self.pending[b] = dist.all_gather_into_tensor(self.fulls[b], self.outs[b], async_op=True)
            if self._eager:
                with torch.cuda.stream(self._side):
                    self.pending[b].wait()                      # the side stream waits for the collective (no host block)
                    torch.index_select(self.fulls[b], 0, self._index, out=self.frames[b])
                    ev = torch.cuda.Event()
                    ev.record(self._side)
                self._done[b] = ev
        else:
            dist.all_gather_into_tensor(self.fulls[b], self.outs[b])

    def _wait(self, b: int):
        if self.pending[b] is not None:
            self.pending[b].wait()
            self.pending[b] = None
        if self._done[b] is not None:                           # the next gather into fulls[b] / reader of frames[b] comes behind the reorder
            torch.cuda.current_stream().wait_event(self._done[b])
            self._done[b] = None

    def fence(self):
        for b in range(self.depth):
            self._wait(b)

    def frame(self, b: int):
        """The gathered frame of buffer ``b`` ([n_total, C] in frame order; a view when the gathered buffer already is the frame: equal
        contiguous shards — otherwise one index_select with the partition's ``gather_index``)."""
        self._wait(b)
        if not self.collective:
            return self.outs[b][:self.count]
        if self._eager:
            return self.frames[b]
        full = self.fulls[b]
        return full if self._index is None else full.index_select(0, self._index)


def allreduce_gradients(trainer, group=None):
    """Data-parallel training (not in the reference): average the gradients of the replicas in place — one
    all-reduce of the trainer's flat gradient array (1.4 M floats = 5.5 MB, a single RCCL launch over xGMI) — so that every
    rank's following ``adam_step`` is identical.  Each rank runs ``fwd_bwd`` on its share of the global batch (equal shares:
    the losses are means over the rank's rays) with the per-batch draws (neighbour ranks, coin flips, n_mult) shared by all ranks."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return
    g = trainer.flat('grad')
    dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group)
    g.mul_(1.0 / world)
