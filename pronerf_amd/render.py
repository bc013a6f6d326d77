"""Frame-level host logic of the inference path: neighbour selection, projection matrices,
image packing, ray-range sharding — the part of the reference's ``render_path`` that runs once
per frame outside its timed region (run_S_eS_eN_alter_trt.py:245-302) — and the ``Renderer``
object that owns the packed networks and the per-ray workspace.

All per-ray work is done by the HIP kernels behind ``pronerf_amd.ops``; what is computed here on
the host is O(number of cameras) (a 20-element sort and four 3x4 matrix products per frame).
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops

N_SAMPLES = 8
NUM_NEIGHBOR = 4
FLIP = np.diag([1.0, -1.0, -1.0]).astype(np.float32)

# Named operating points of the renderer (``Renderer(preset=...)``; the per-net kernel variants of include/pronerf_hip.h).
#   'default'  what bench.py's headline times: two-pass sampler (statistical index parity, fp16-grade depths on the undecided-free rays),
#              fp16 refine, bf16 NeRF MLP — rendering error 56 .. 59 dB against the fp32 reference renderer; holds the "PSNR within 0.05 dB
#              of the reference" gate for images of up to ~36 dB PSNR, with 2x headroom up to ~33 dB (real LLFF scenes: 25 .. 28 dB).
#   'quality'  every ray through the split-fp16 sampler (exact indices, fp32-grade depths) + fp16 NeRF operands: 63 .. 65 dB rendering error,
#              +0.8 ms per 762 048-ray frame; the choice for scenes a net fits beyond ~35 dB and wherever index parity must be exact.
PRESETS = {'default': {}, 'quality': {'sampler': 'sampler_split', 'nerf': 'f16'}}


def select_neighbors(c2w, poses, num_neighbor=NUM_NEIGHBOR):
    """Indices of the ``num_neighbor`` source cameras closest to the target camera centre,
    ascending (run_S_eS_eN_alter_trt.py:281-283)."""
    c2w = np.asarray(c2w, dtype=np.float32); poses = np.asarray(poses, dtype=np.float32)
    d = np.sqrt(((c2w[None, :3, 3] - poses[:, :3, 3]) ** 2).sum(1, dtype=np.float32))
    return np.argsort(d, kind='stable')[:num_neighbor]


def projection_matrices(K, poses):
    """M_k = K . diag(1,-1,-1) . pose_k, each 3x4 (run_S_eS_eN_alter_trt.py:289-294).
    The kernel is agnostic to what the 3x4 means; like the reference this passes the stored
    pose as-is."""
    K = np.asarray(K, dtype=np.float32); poses = np.asarray(poses, dtype=np.float32)
    return np.stack([K @ (FLIP @ p[:3, :4]) for p in poses], 0).astype(np.float32)


def shard_range(n_total: int, rank: int, world: int):
    """Contiguous flat ray range of ``rank``: [first, first+count).  Sizes differ by at most one
    ray; concatenating the ranges in rank order gives back [0, n_total)."""
    base, rem = divmod(int(n_total), int(world))
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


class RayPartition:
    """How the ``n_total`` rays of a frame are dealt to ``world`` ranks.

    'contiguous'  rank r renders the flat range ``shard_range(n_total, r, world)``.
    'cyclic'      blocks of ``block`` consecutive rays are dealt round-robin: rank r renders blocks r, r + world, r + 2 world, ...  Every rank
                  then sees every part of the image.  The per-ray work is not uniform — the two-pass sampler re-renders the rays whose depth
                  order its first pass cannot decide, 11 % of the rays in the top eighth of the bench frame and 36 % in the bottom eighth — so
                  contiguous shards finish 0.65 .. 0.71 ms apart and the frame waits for the slowest (profiles/r04_shard_rehearsal.json);
                  dealt cyclically every rank gets the frame's average.

    Rank r's rays are the rows of ``frame_rays(rank)``; after the all-gather of the per-rank [cmax, C] tiles, ``gather_index`` maps frame row g
    to its row in the gathered [world * cmax, C] buffer (None when the gathered buffer already is the frame)."""

    def __init__(self, n_total: int, world: int, kind: str = 'cyclic', block: int = 1024):
        if kind not in ('contiguous', 'cyclic'):
            raise ValueError(f"RayPartition: kind must be 'contiguous' or 'cyclic', got {kind!r}")
        self.n_total, self.world = int(n_total), int(world)
        # small frames: at least ~8 blocks per rank (blocks stay multiples of 32 rays = the NeRF stage's row granularity of 256 samples)
        self.block = max(32, min(int(block), self.n_total // (8 * self.world) // 32 * 32))
        self.kind = 'contiguous' if self.world == 1 else kind
        if self.kind == 'contiguous':
            self.counts = [shard_range(n_total, r, world)[1] for r in range(world)]
        else:
            nblk = (self.n_total + self.block - 1) // self.block
            tail = self.n_total - (nblk - 1) * self.block                 # rays of the last block
            self.counts = []
            for r in range(self.world):
                mine = len(range(r, nblk, self.world))
                self.counts.append(mine * self.block - ((self.block - tail) if (mine and (nblk - 1) % self.world == r) else 0))
        self.cmax = max(self.counts) if self.counts else 0

    def count(self, rank: int) -> int:
        return self.counts[rank]

    def frame_rays_args(self, rank: int):
        """kwargs of ``ops.frame_rays`` / ``Renderer.frame_rays`` for this rank's rays."""
        if self.kind == 'contiguous':
            first, count = shard_range(self.n_total, rank, self.world)
            return {'first': first, 'count': count}
        return {'first': rank * self.block, 'count': self.counts[rank], 'block': self.block, 'stride': self.world * self.block}

    def rows(self, rank: int):
        """The frame rows (flat ray indices) of this rank, in the order it renders them (host tensor, int64)."""
        if self.kind == 'contiguous':
            first, count = shard_range(self.n_total, rank, self.world)
            return torch.arange(first, first + count)
        q = torch.arange(self.counts[rank])
        return rank * self.block + (q // self.block) * (self.world * self.block) + q % self.block

    def gather_index(self, device=None):
        """[n_total] int64: row of frame ray g in the gathered [world * cmax, C] buffer; None if that buffer is the frame itself."""
        if self.kind == 'contiguous':
            if self.cmax * self.world == self.n_total:
                return None
            idx = torch.cat([r * self.cmax + torch.arange(c) for r, c in enumerate(self.counts)])
            return idx.to(device) if device is not None else idx
        g = torch.arange(self.n_total)
        b = g // self.block
        idx = (b % self.world) * self.cmax + (b // self.world) * self.block + g % self.block
        return idx.to(device) if device is not None else idx


class Renderer:
    """Packed networks + workspace for ``render_rays`` (inference).

    weights: dict with 'sampler'/'refine'/'nerf' -> {'W': [...], 'b': [...]} (torch layout
    ``W[out,in]``, numpy or torch) — e.g. ``pronerf_amd.synthetic.make_weights`` or the tensors of
    a checkpoint's state dicts (see ``run_nerf_helpers.weights_from_state_dicts``); an entry may also be an
    ``ops.PackedMLP`` (a module's ``packed()``, or one loaded from an engine file).
    """

    def __init__(self, weights, max_rays: int, device='cuda:0', variants=None, shape=None, preset='default'):
        """preset: 'default' | 'quality' (``PRESETS`` above); ``variants`` override single nets of it.  variants: optional {'sampler' | 'refine' | 'nerf': kernel variant} (``ops.PackedMLP.set_variant``; parity tests and A/B
        timing — the default kernels are the product path).  shape: optional workgroup shape for the three nets ('wide' | 'narrow' |
        'auto', or a dict per net; ``ops.PackedMLP.set_shape``) — default: chosen per launch from the ray count."""
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise ops.PnrfError('Renderer needs a GPU device (pronerf_amd has no CPU path)')
        if preset not in PRESETS:
            raise ops.PnrfError(f'Renderer: preset must be one of {sorted(PRESETS)}, got {preset!r}')
        self.preset = preset
        variants = {**PRESETS[preset], **(variants or {})}
        def pack(net, w):
            if isinstance(w, ops.PackedMLP):                  # already packed (module.packed(), an engine file)
                if w.net not in net:
                    raise ops.PnrfError(f'Renderer: packed network of kind {w.net} where one of {net} is needed')
                return w
            # DoNeRFTRT (any depth) ends in its 4-wide output layer; the NeRF class (pts0..7, feature, alpha, views, rgb) in its 3-wide rgb head
            kind = net[0] if len(net) == 1 else (ops.NET_NERFCLS if (len(w['W']) == 12 and tuple(w['W'][-1].shape)[0] == 3) else ops.NET_NERF)
            return ops.PackedMLP(kind, w['W'], w['b'])
        with torch.cuda.device(self.device):
            self.sampler = pack((ops.NET_SAMPLER,), weights['sampler'])
            self.refine = pack((ops.NET_REFINE,), weights['refine'])
            self.nerf = pack((ops.NET_NERF, ops.NET_NERFCLS), weights['nerf'])
            for k, v in (variants or {}).items():
                getattr(self, k).set_variant(v)
            if shape is not None:
                for k in ('sampler', 'refine', 'nerf'):
                    w = shape.get(k) if isinstance(shape, dict) else shape
                    if w is not None:
                        getattr(self, k).set_shape(w)
            self.ctx = ops.RenderContext(self.sampler, self.refine, self.nerf, max_rays)
        self.num_neighbor = (self.refine.in_dim - 48) // 24           # the refine net's input is 6 x 8 Pluecker values + 3 x 8 colours per neighbour view
        self.img4 = None
        self.proj = None
        self.ref_nos = None

    # ---- engine files (the reference's <export_dir>/*_fp16.trt, pronerf/tensorrt.py:8-14)
    ENGINE_FILES = {'nerf': 'nerf.pnrf', 'sampler': 'minmaxrays_net.pnrf', 'refine': 'refine_net.pnrf'}

    def save_engines(self, export_dir):
        import os
        os.makedirs(export_dir, exist_ok=True)
        paths = {k: os.path.join(export_dir, f) for k, f in self.ENGINE_FILES.items()}
        for k, path in paths.items():
            getattr(self, k).save(path)
        return paths

    @classmethod
    def from_engines(cls, export_dir, max_rays: int, device='cuda:0'):
        import os
        dev = torch.device(device)
        if dev.type != 'cuda':
            raise ops.PnrfError('Renderer needs a GPU device (pronerf_amd has no CPU path)')
        with torch.cuda.device(dev):
            packed = {k: ops.PackedMLP.load(os.path.join(export_dir, f)) for k, f in cls.ENGINE_FILES.items()}
        return cls(packed, max_rays, device)

    # ---- per frame (outside the timed region, like run_S_eS_eN_alter_trt.py:281-302)
    def set_views(self, c2w, poses, images_nhwc, K, num_neighbor=None):
        """Pick the neighbours of the target pose, upload + interleave their images, build the
        projection matrices.  images_nhwc: [n_views,H,W,3] numpy/torch in [0,1].  num_neighbor: None = what the refine net was built for
        (``self.num_neighbor``; 4 in the Fern configs) — any other value is refused by the kernels."""
        ref = select_neighbors(c2w, poses, self.num_neighbor if num_neighbor is None else num_neighbor)
        self.ref_nos = ref
        imgs = images_nhwc[ref] if isinstance(images_nhwc, np.ndarray) else images_nhwc[torch.as_tensor(ref)]
        nchw = torch.as_tensor(imgs, dtype=torch.float32).permute(0, 3, 1, 2).contiguous().to(self.device)   # H2D (trt.py:286)
        with torch.cuda.device(self.device):
            self.img4 = ops.images_pack(nchw)
        self.proj = torch.from_numpy(projection_matrices(K, np.asarray(poses)[ref])).to(self.device)
        return ref

    def frame_rays(self, K, c2w, H, W, first=0, count=None, block=None, stride=0):
        with torch.cuda.device(self.device):
            return ops.frame_rays(K, c2w, H, W, first=first, count=count, device=self.device, block=block, stride=stride)

    def calibrate(self, rays, or_rays, threshold=0.55):
        """Pick the sampler form for THESE nets from one rendered frame (once per checkpoint, outside any timed loop; waits for the device).
        A sampler that has learned surfaces bunches its 8 depths there and the two-pass form re-renders most rays (68.7 % on the scene-trained
        fixture: 1.46 ms for pass 1 + pass 2 against 1.30 ms for the split-fp16 kernel alone on 762 048 rays): from ``threshold`` of the rays
        on, the single exact pass is the faster AND the exact choice, and this switches the sampler handle to it (same rows as pass 2 renders
        them, bit for bit).  Returns (second-pass fraction, variant now in force).  Deterministic: the decision is a function of the frame."""
        if self.sampler.variant != 'default':
            return None, self.sampler.variant
        self.render_rays(rays, or_rays)
        frac = self.ctx.sampler_stats() / max(1, rays.shape[0])           # synchronises
        if frac > threshold:
            self.sampler.set_variant('sampler_split')
        return frac, self.sampler.variant

    # ---- the hot path (the reference's timed region, trt.py:327-332)
    def render_rays(self, rays, or_rays, eps=1e-5, want_idx=False, out=None):
        if self.img4 is None:
            raise ops.PnrfError('Renderer.render_rays: call set_views() first')
        with torch.cuda.device(self.device):
            return self.ctx.render_rays(rays, or_rays, self.img4, self.proj, eps=eps, want_idx=want_idx, out=out)


class ChunkedRenderer:
    """A frame as calls of at most ``chunk`` rays (BASELINE.json configs[1]: "1024-ray chunks"; the reference's ``chunk`` argument,
    run_S_eS_eN_alter_trt.py:223, which its TRT path accepts and ignores).  A small call is latency-bound — four dependent kernels of one
    batch each, a few of the 256 CUs busy — so the chunks go round-robin over ``streams`` HIP streams, each with its own context (workspace):
    ``streams`` chunks are in flight at a time, fork / join by events on the caller's stream, which also captures into one hipGraph.
    Same kernels, same rows: the result equals the one-call frame bit for bit."""

    def __init__(self, renderer: Renderer, chunk: int = 1024, streams: int = 4):
        self.r, self.chunk, self.k = renderer, int(chunk), max(1, int(streams))
        with torch.cuda.device(renderer.device):
            self.ctxs = [ops.RenderContext(renderer.sampler, renderer.refine, renderer.nerf, self.chunk) for _ in range(self.k)]
            self.streams = [torch.cuda.Stream(device=renderer.device) for _ in range(self.k)] if self.k > 1 else [None]
            self.fork = torch.cuda.Event()
            self.joins = [torch.cuda.Event() for _ in range(self.k)]

    def render_rays(self, rays, or_rays, out, eps=1e-5):
        r = self.r
        if r.img4 is None:
            raise ops.PnrfError('ChunkedRenderer.render_rays: call Renderer.set_views() first')
        n = rays.shape[0]
        with torch.cuda.device(r.device):
            cur = torch.cuda.current_stream()
            if self.k == 1:
                for a in range(0, n, self.chunk):
                    b = min(n, a + self.chunk)
                    self.ctxs[0].render_rays(rays[a:b], or_rays[a:b], r.img4, r.proj, eps=eps, out=out[a:b])
                return out
            self.fork.record(cur)
            for i, st in enumerate(self.streams):
                st.wait_event(self.fork)
            for j, a in enumerate(range(0, n, self.chunk)):
                b = min(n, a + self.chunk)
                i = j % self.k
                with torch.cuda.stream(self.streams[i]):
                    self.ctxs[i].render_rays(rays[a:b], or_rays[a:b], r.img4, r.proj, eps=eps, out=out[a:b])
            for i, st in enumerate(self.streams):
                self.joins[i].record(st)
                cur.wait_event(self.joins[i])
        return out
