"""Synthetic training workloads at the sizes BASELINE.json names (configs[3], configs[4]), built from product code only: seeded
weights (``synthetic``), device-generated rays of one training view (``ops.frame_rays``), the drivers' neighbour ranking
(run_S_eS_eN_alter_base_refine2.py:584-600) and packed views.  Used by ``bench.py`` (the ``train`` block), by the full-size tests and by
``tools/train_iter.py`` (profiling); nothing here imports ``oracle``.

    configs[3]  stage-2 iteration: N_rand = 4096 rays of one view, 17 training views of 756 x 1008, 8 samples, NeRF-class fine net,
                forward + backward + Adam  (run_S_eS_eN_alter_base_refine2.py:831-878)
    configs[4]  stage-1 exploration iteration: same batch, sampler / refine nets frozen, 8 n_mult samples per ray (the reference draws
                n_mult in 1..8, BASELINE.json names 256 samples = n_mult 32), NeRF-only Adam  (run_S_eS_eN_alter_base.py:689-729, 929-940)
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops, synthetic
from .run_S_eS_eN_alter_base_refine2 import neighbor_rank_table, trainer_layer_list

H, W, FOCAL, N_VIEWS, N_RAND = 756, 1008, 815.13, 17, 4096


def layer_macs():
    """(in, out, needs input gradient) of the 26 trainer layers, in trainer order."""
    mm = lambda fi, fo: [(fi, 256, False)] + [(256, 256, True)] * 5 + [(256, fo, True)]
    nerf = [(63, 256, True)] + [(256, 256, True)] * 4 + [(319, 256, True)] + [(256, 256, True)] * 2 + [(256, 256, True), (256, 1, True), (283, 128, True), (128, 3, True)]
    return mm(288, 27), mm(144, 35), nerf


def train_flop(n_rays: int, samples: int, nerf_only: bool = False):
    """Algorithmic FLOPs of one training iteration: per Linear layer 2 in out rows for the forward product, the same for the weight
    gradient and, where the layer's input carries gradient, for the input gradient (bias, activations, encodings, compositing, Adam
    excluded).  ``nerf_only``: stage-1 exploration iterations — sampler / refine nets forward only."""
    s, r, f = layer_macs()
    per = lambda layers, rows, train: sum(2 * fi * fo * rows * ((3 if dx else 2) if train else 1) for fi, fo, dx in layers)
    return per(s, n_rays, not nerf_only) + per(r, n_rays, not nerf_only) + per(f, n_rays * samples, True)


class TrainWorkload:
    """One fixed training batch at configs[3] / configs[4] size and a trainer sized for ``max_samples`` samples per ray."""

    def __init__(self, device='cuda:0', n_rays=N_RAND, n_views=N_VIEWS, H=H, W=W, focal=FOCAL, seed=0, max_samples=8, own=2):
        dev = torch.device(device)
        self.device, self.n, self.nv, self.H, self.W = dev, int(n_rays), int(n_views), int(H), int(W)
        scene = synthetic.make_scene(seed, H=H, W=W, n_views=n_views, sigma_t=0.2, rotate=True, focal=focal)
        w = synthetic.make_weights(seed, 'trained')
        sd = synthetic.state_dicts(w)
        fine_sd = synthetic.nerfcls_state_dict(synthetic.make_nerfcls_weights(seed, head_scale=0.3))
        self.layers = trainer_layer_list(sd['sampler'], sd['refine'], fine_sd)
        rs = np.random.RandomState(seed)
        sel = torch.from_numpy(np.sort(rs.choice(H * W, self.n, replace=False))).to(dev)
        with torch.cuda.device(dev):
            rays, or_rays = ops.frame_rays(scene['K'], scene['poses'][own], H, W, near=0., far=1., device=dev)
            self.rays, self.or_rays = rays[sel].contiguous(), or_rays[sel].contiguous()
            del rays, or_rays
            self.target = torch.from_numpy(scene['images'][own].reshape(-1, 3)).to(dev)[sel].contiguous()
            self.images_nchw = torch.from_numpy(scene['images']).to(dev).permute(0, 3, 1, 2).contiguous()
            self.img4 = ops.images_pack(self.images_nchw)
            self.poses = torch.from_numpy(scene['poses']).to(dev)[:, :3, :4].contiguous()
            self.K = torch.from_numpy(scene['K']).to(dev).contiguous()
            rank = torch.from_numpy(neighbor_rank_table(self.poses)).to(dev)
            order = torch.as_tensor(np.sort(rs.choice(np.arange(0, n_views - 1), 4, replace=False)), device=dev)
            self.ref_nos = rank[own][1:][order][None].expand(self.n, -1).contiguous()              # the batch comes from one view (:594-600)
            self.jitter = torch.from_numpy(np.minimum(np.abs(rs.randn(self.n, 8)) / 5, 1 - 2e-6).astype(np.float32)).to(dev)
            self.noise = torch.from_numpy(rs.randn(self.n, 8).astype(np.float32)).to(dev)
            self._rs = rs
            self.max_samples = int(max_samples)
            self.trainer = ops.Trainer([W_ for W_, _ in self.layers], [b for _, b in self.layers], max_rays=self.n, device=dev, max_samples=self.max_samples)
        self._xj = {}

    def batch_args(self):
        return (self.rays, self.or_rays, self.target, self.img4, self.poses, self.K, self.ref_nos)

    def stage2_step(self, lr=5e-4, want_rgb=False, adam=True):
        out = self.trainer.fwd_bwd(*self.batch_args(), jitter=self.jitter, jitter_dir=1, raw_noise=self.noise, want_rgb=want_rgb)
        if adam:
            self.trainer.adam_step(lr, weight_decay=5e-8)
        return out

    def explore_jitter(self, n_mult):
        if n_mult not in self._xj:
            j = np.minimum(np.abs(np.random.RandomState(1000 + n_mult).randn(self.n, 8 * n_mult)) / 5, 0.99).astype(np.float32)
            self._xj[n_mult] = torch.from_numpy(j).to(self.device)
        return self._xj[n_mult]

    def explore_step(self, n_mult, lr=5e-4, want_rgb=False, adam=True):
        if 8 * n_mult > self.max_samples:
            raise ops.PnrfError(f'TrainWorkload: {8 * n_mult} samples per ray, trainer sized for {self.max_samples}')
        out = self.trainer.explore_fwd_bwd(*self.batch_args(), n_mult=n_mult, dir1=1, jitter=self.explore_jitter(n_mult), dir2=-1, raw_noise=None,
                                           want_rgb=want_rgb)
        if adam:
            self.trainer.adam_step(lr, weight_decay=5e-8, nerf_only=True)
        return out


def timed_ms(fn, iters, warm, stream_sync=torch.cuda.synchronize):
    """Mean device milliseconds of ``fn`` over ``iters`` back-to-back calls (events on the current stream) and the host wall time."""
    import time
    for _ in range(warm):
        fn()
    stream_sync()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(iters):
        fn()
    e1.record(); stream_sync()
    return e0.elapsed_time(e1) / iters, (time.perf_counter() - t0) * 1e3 / iters


def settled_ms(fn, reps):
    """Device milliseconds per call once the chip has settled on this call size: the better of two measurements of ``reps`` back-to-back calls, each
    behind ``reps`` // 2 untimed ones.  (A measurement taken right after calls of another size can sit 5 % high for its whole length — the clock the
    power manager chose for the previous load — which is not what a rank that renders the same shard frame after frame sees.)"""
    return min(timed_ms(fn, reps, max(5, reps // 2))[0] for _ in range(2))


def shard_rehearsal(weights, scene, H, W, device='cuda:0', worlds=(1, 2, 4, 8), call_rays=(1024, 4096), reps=40, shape=None, check=True, stages=False):
    """One-GPU rehearsal of what a rank of an N-GPU run and a chunked caller execute, on the kernels of THIS build: ``pnrf_render_rays_fwd`` on the
    first / middle / last contiguous ray shard of 1/N of the frame (``shard_range``), no collective; and on single calls of ``call_rays`` rays
    (the frame's first rays), launched back to back on one stream.  Reports per world size the slowest of the three shards and the implied
    kernel-only strong-scaling bound frame_ms / shard_ms.  ``check``: every shard / call equals the same rows of the one-call frame bit for bit
    (rays are independent).  ``stages``: a second pass per size with the context's per-stage events (sampler / refine / NeRF ms)."""
    from .render import RayPartition, Renderer, shard_range

    def stage_ms(fn, n):
        rend.ctx.profile_begin(n)
        for _ in range(n):
            fn()
        return rend.ctx.profile_end()[0]
    dev = torch.device(device)
    n_total = H * W
    rend = Renderer(weights, max_rays=n_total, device=dev, shape=shape)
    rend.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
    rays, or_rays = rend.frame_rays(scene['K'], scene['c2w'], H, W)
    ref = torch.empty(n_total, 4, device=dev)
    rend.render_rays(rays, or_rays, out=ref)
    out = torch.empty_like(ref)
    res = {'what': 'one GPU, current kernels: ms per pnrf_render_rays_fwd call on the first / middle / last contiguous shard of 1/N of the 762 048-ray '
                   'frame (no collective) — and, cyclic_*, on the same ranks\' share of the frame dealt in blocks of 1024 rays round-robin (what bench.py and the frame driver use at N > 1) — and on single small calls; every figure the better of two measurements of `reps` back-to-back calls, each behind reps / 2 untimed ones; speedup_bound = frame ms / slowest shard ms',
           'reps': reps, 'shape': shape if shape is not None else 'auto (per launch: the default)', 'shards': {}, 'calls': {}}
    identical = True
    frame_ms = None
    for world in worlds:
        per = {}
        for rank in sorted({0, world // 2, world - 1}):
            first, count = shard_range(n_total, rank, world)
            r, o, dst = rays[first:first + count], or_rays[first:first + count], out[first:first + count]
            ms = settled_ms(lambda: rend.render_rays(r, o, out=dst), reps)
            per[f'rank{rank}'] = ms
            if stages:
                per[f'rank{rank}_stages'] = stage_ms(lambda: rend.render_rays(r, o, out=dst), min(reps, 64))
                per[f'rank{rank}_stages']['rays_second_pass'] = rend.ctx.sampler_stats()
            if check:
                identical = identical and bool(torch.equal(dst, ref[first:first + count]))
        worst = max(v for k, v in per.items() if not k.endswith('_stages'))
        if world == 1:
            frame_ms = worst
        res['shards'][str(world)] = {'rays': shard_range(n_total, 0, world)[1], 'ms': per, 'ms_slowest': worst,
                                     'speedup_bound': (frame_ms / worst) if frame_ms else None,
                                     'rays_per_s_x_world': n_total / worst * 1e3}
        if world > 1:
            # the same ranks with the frame dealt in blocks of 1024 rays round-robin (RayPartition 'cyclic': what bench.py and the frame driver use at
            # N > 1): every rank renders the frame's average share of second-pass rays
            part = RayPartition(n_total, world, 'cyclic')
            cyc = {}
            for rank in sorted({0, world // 2, world - 1}):
                rows = part.rows(rank).to(dev)
                r, o = rays.index_select(0, rows), or_rays.index_select(0, rows)
                dst = torch.empty(rows.shape[0], 4, device=dev)
                cyc[f'rank{rank}'] = settled_ms(lambda: rend.render_rays(r, o, out=dst), reps)
                if stages:
                    cyc[f'rank{rank}_stages'] = stage_ms(lambda: rend.render_rays(r, o, out=dst), min(reps, 64))
                    cyc[f'rank{rank}_stages']['rays_second_pass'] = rend.ctx.sampler_stats()
                if check:
                    identical = identical and bool(torch.equal(dst, ref.index_select(0, rows)))
            cw = max(v for k, v in cyc.items() if not k.endswith('_stages'))
            res['shards'][str(world)].update(cyclic_ms=cyc, cyclic_ms_slowest=cw, cyclic_speedup_bound=frame_ms / cw)
    for c in call_rays:
        r, o, dst = rays[:c], or_rays[:c], out[:c]
        ms, wall = timed_ms(lambda: rend.render_rays(r, o, out=dst), 200, 20)
        res['calls'][str(c)] = {'ms_per_call': ms, 'host_ms_per_call': wall, 'rays_per_s': c / ms * 1e3,
                                'frame_ms_at_this_rate': ms * ((n_total + c - 1) // c)}
        if stages:
            res['calls'][str(c)]['stages'] = stage_ms(lambda: rend.render_rays(r, o, out=dst), 64)
        if check:
            identical = identical and bool(torch.equal(dst, ref[:c]))
    if check:
        res['bit_identical_to_one_call_frame'] = identical
    del rend
    return res
