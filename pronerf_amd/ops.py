"""Thin torch-tensor wrappers over the C ABI (device memory + stream plumbing only).

Every function takes contiguous fp32 CUDA(ROCm) tensors, allocates its outputs with torch and
launches on ``torch.cuda.current_stream()``.  No arithmetic happens on the Python side.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import NET_NERF, NET_NERFCLS, NET_REFINE, NET_SAMPLER, PnrfError, check

f32 = torch.float32


def _ptr(t):
    return C.c_void_p(0) if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t, name, shape_tail=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise PnrfError(f'{name}: expected a GPU tensor (pronerf_amd has no CPU path)')
    if t.dtype != f32:
        raise PnrfError(f'{name}: expected float32, got {t.dtype}')
    if shape_tail is not None and tuple(t.shape[-len(shape_tail):]) != tuple(shape_tail):
        raise PnrfError(f'{name}: expected trailing shape {shape_tail}, got {tuple(t.shape)}')
    return t.contiguous()


class PackedMLP:
    """Device-resident pre-tiled weights of one network (pnrf_mlp_pack)."""

    def __init__(self, net: int, weights, biases, variant=None):
        """variant: None / 'default', or one of ``_lib.VARIANTS`` ('sampler_f32', 'sampler_f32_full', 'bf16_32x32') — fixed for the
        life of the handle (pnrf_mlp_set_variant); nothing is read from the environment."""
        lib = _lib.load()
        n = len(weights)
        ws = [np.ascontiguousarray(w.detach().cpu().numpy() if isinstance(w, torch.Tensor) else w, dtype=np.float32) for w in weights]
        bs = [np.ascontiguousarray(b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else b, dtype=np.float32) for b in biases]
        Wp = (C.c_void_p * n)(*[w.ctypes.data for w in ws])
        bp = (C.c_void_p * n)(*[b.ctypes.data for b in bs])
        ind = (C.c_int * n)(*[w.shape[1] for w in ws])
        outd = (C.c_int * n)(*[w.shape[0] for w in ws])
        h = C.c_void_p()
        check(lib.pnrf_mlp_pack(net, Wp, bp, ind, outd, n, C.byref(h)), 'pnrf_mlp_pack')
        self.handle = h
        self.net = net
        self.in_dim = ws[0].shape[1]
        self.out_dim = 4 if net == NET_NERFCLS else ws[-1].shape[0]      # NeRF class: [rgb(3), alpha]
        self.variant = 'default'
        if variant not in (None, 'default'):
            self.set_variant(variant)

    def set_variant(self, variant):
        """Configuration step (before the handle is used by a context / stream)."""
        if variant not in _lib.VARIANTS:
            raise PnrfError(f'unknown kernel variant {variant!r}; one of {sorted(_lib.VARIANTS)}')
        check(_lib.load().pnrf_mlp_set_variant(self.handle, _lib.VARIANTS[variant]), 'pnrf_mlp_set_variant')
        self.variant = variant
        return self

    SHAPES = {'auto': 0, 'narrow': 4, 'wide': 8}

    def set_shape(self, shape):
        """Workgroup shape of the fused stages launched from this handle (pnrf_mlp_set_shape): 'auto' (per launch, the default), 'wide',
        'narrow', or the PNRF_SHAPE_* integer — bit-identical results, A/B timing."""
        v = self.SHAPES.get(shape, shape) if isinstance(shape, str) else shape
        if isinstance(v, str):
            raise PnrfError(f'unknown workgroup shape {shape!r}; one of {sorted(self.SHAPES)}')
        check(_lib.load().pnrf_mlp_set_shape(self.handle, int(v)), 'pnrf_mlp_set_shape')
        return self

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                _lib.load().pnrf_mlp_free(self.handle)
                self.handle = None
        except Exception:
            pass

    # ---- engine files: the packed stream as bytes (pnrf_mlp_serialize / pnrf_mlp_deserialize)
    def serialize(self) -> bytes:
        lib = _lib.load()
        size = C.c_int64()
        check(lib.pnrf_mlp_serialize(self.handle, None, 0, C.byref(size)), 'pnrf_mlp_serialize')
        buf = (C.c_char * size.value)()
        check(lib.pnrf_mlp_serialize(self.handle, buf, size.value, C.byref(size)), 'pnrf_mlp_serialize')
        return bytes(buf)

    @classmethod
    def deserialize(cls, data: bytes, expect_net=None):
        """New handle on the current device from ``serialize()`` output; ``expect_net`` (NET_*) guards against swapped files."""
        lib = _lib.load()
        data = bytes(data)
        h = C.c_void_p()
        check(lib.pnrf_mlp_deserialize(data, len(data), C.byref(h)), 'pnrf_mlp_deserialize')
        self = cls.__new__(cls)
        self.handle = h
        net, ind, indx, outd = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        check(lib.pnrf_mlp_kind(h, C.byref(net), C.byref(ind), C.byref(indx), C.byref(outd)), 'pnrf_mlp_kind')
        self.net, self.in_dim, self.out_dim = net.value, ind.value, outd.value
        self.variant = 'default'
        if expect_net is not None and self.net != expect_net:
            raise PnrfError(f'engine holds net kind {self.net}, expected {expect_net}')
        return self

    def save(self, path):
        with open(path, 'wb') as f:
            f.write(self.serialize())

    @classmethod
    def load(cls, path, expect_net=None):
        with open(path, 'rb') as f:
            return cls.deserialize(f.read(), expect_net)

    def forward(self, x, x_views=None, head_act=False):
        """Output of the last Linear, [m, out_dim]; head_act applies the TRT classes' sigmoid/tanh heads."""
        x = _chk(x, 'x', (self.in_dim,))
        m = x.shape[0]
        y = torch.empty(m, self.out_dim, device=x.device, dtype=f32)
        if x_views is not None:
            x_views = _chk(x_views, 'x_views', (27,))
        check(_lib.load().pnrf_mlp_fwd(self.handle, _ptr(x), _ptr(x_views), _ptr(y), m, int(bool(head_act)), _stream()), 'pnrf_mlp_fwd')
        return y


def posenc(x, n_freq):
    x = _chk(x, 'x', (3,))
    lead = x.shape[:-1]
    xf = x.reshape(-1, 3)
    out = torch.empty(xf.shape[0], 3 + 6 * n_freq, device=x.device, dtype=f32)
    check(_lib.load().pnrf_posenc_fwd(_ptr(xf), _ptr(out), xf.shape[0], n_freq, _stream()), 'pnrf_posenc_fwd')
    return out.reshape(*lead, 3 + 6 * n_freq)


def plucker(o, d):
    o = _chk(o, 'rays_o', (3,)); d = _chk(d, 'rays_d', (3,))
    if o.shape != d.shape:
        raise PnrfError(f'plucker: shape mismatch {tuple(o.shape)} vs {tuple(d.shape)}')
    lead = o.shape[:-1]
    of, df = o.reshape(-1, 3), d.reshape(-1, 3)
    out = torch.empty(of.shape[0], 6, device=o.device, dtype=f32)
    check(_lib.load().pnrf_plucker_fwd(_ptr(of), _ptr(df), _ptr(out), of.shape[0], _stream()), 'pnrf_plucker_fwd')
    return out.reshape(*lead, 6)


def ray_encode(rays, n_pts=48):
    rays = _chk(rays, 'rays', (11,))
    out = torch.empty(rays.shape[0], 6 * n_pts, device=rays.device, dtype=f32)
    check(_lib.load().pnrf_ray_encode_fwd(_ptr(rays), _ptr(out), rays.shape[0], n_pts, _stream()), 'pnrf_ray_encode_fwd')
    return out


def frame_rays(K, c2w, H, W, near=0.0, far=1.0, or_near=1.0, or_far=10.0, first=0, count=None, device='cuda', block=None, stride=0):
    """rays[count,11], or_rays[count,11] for flat pixel range [first, first+count); with ``block`` / ``stride``: row q is pixel
    first + (q // block) * stride + q % block (a rank's share of a block-cyclic partition, pnrf_frame_rays_blocks_fwd)."""
    count = H * W - first if count is None else count
    Kh = np.ascontiguousarray(np.asarray(K.detach().cpu() if isinstance(K, torch.Tensor) else K, dtype=np.float32).reshape(3, 3))
    Ch = np.ascontiguousarray(np.asarray(c2w.detach().cpu() if isinstance(c2w, torch.Tensor) else c2w, dtype=np.float32)[:3, :4])
    rays = torch.empty(count, 11, device=device, dtype=f32)
    orr = torch.empty(count, 11, device=device, dtype=f32)
    fp = C.POINTER(C.c_float)
    if block is None:
        check(_lib.load().pnrf_frame_rays_fwd(Kh.ctypes.data_as(fp), Ch.ctypes.data_as(fp), H, W, near, far, or_near, or_far,
                                              first, count, _ptr(rays), _ptr(orr), _stream()), 'pnrf_frame_rays_fwd')
    else:
        check(_lib.load().pnrf_frame_rays_blocks_fwd(Kh.ctypes.data_as(fp), Ch.ctypes.data_as(fp), H, W, near, far, or_near, or_far,
                                                     first, int(block), int(stride), count, _ptr(rays), _ptr(orr), _stream()), 'pnrf_frame_rays_blocks_fwd')
    return rays, orr


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    rays_o = _chk(rays_o, 'rays_o', (3,)); rays_d = _chk(rays_d, 'rays_d', (3,))
    lead = rays_o.shape
    of, df = rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)
    oo = torch.empty_like(of); od = torch.empty_like(df)
    check(_lib.load().pnrf_ndc_rays_fwd(_ptr(of), _ptr(df), int(H), int(W), float(focal), float(near), _ptr(oo), _ptr(od), of.shape[0], _stream()),
          'pnrf_ndc_rays_fwd')
    return oo.reshape(lead), od.reshape(lead)


def warp_trt(img, depth, ro1, rd1, w2c):
    """img [B,3,Hf,Wf]; depth [B,n]; ro1, rd1 [4,n] (shared) or [B,4,n]; w2c [B,3,4] -> [B,3,n]."""
    img = _chk(img, 'img'); depth = _chk(depth, 'depth'); w2c = _chk(w2c, 'w2c', (3, 4))
    B, _, Hf, Wf = img.shape
    n = depth.shape[-1]
    if ro1.dim() == 3 and ro1.stride(0) == 0:       # the reference's expand(): one copy shared by all B
        ro1, rd1 = ro1[0], rd1[0]
    ro1 = _chk(ro1, 'ro1'); rd1 = _chk(rd1, 'rd1')
    bstride = 4 * n if ro1.dim() == 3 else 0
    out = torch.empty(B, 3, n, device=img.device, dtype=f32)
    check(_lib.load().pnrf_warp_trt_fwd(_ptr(img), _ptr(depth), _ptr(ro1), _ptr(rd1), bstride, _ptr(w2c), _ptr(out), B, Hf, Wf, n, _stream()),
          'pnrf_warp_trt_fwd')
    return out


def warp_train(img, depth, ro1, rd1, c2w2, K):
    """Training warp: img [B,3,Hf,Wf]; depth [B,n]; ro1, rd1 [3,n] (shared) or [B,3,n]; c2w2 [B,3,4]; K [B,3,3] -> [B,3,n]."""
    img = _chk(img, 'img'); depth = _chk(depth, 'depth'); c2w2 = _chk(c2w2, 'c2w2', (3, 4)); K = _chk(K, 'intrinsics', (3, 3))
    B, _, Hf, Wf = img.shape
    n = depth.shape[-1]
    if ro1.dim() == 3 and ro1.stride(0) == 0:
        ro1, rd1 = ro1[0], rd1[0]
    ro1 = _chk(ro1, 'ro1'); rd1 = _chk(rd1, 'rd1')
    bstride = 3 * n if ro1.dim() == 3 else 0
    out = torch.empty(B, 3, n, device=img.device, dtype=f32)
    check(_lib.load().pnrf_warp_train_fwd(_ptr(img), _ptr(depth), _ptr(ro1), _ptr(rd1), bstride, _ptr(c2w2), _ptr(K), _ptr(out), B, Hf, Wf, n, _stream()),
          'pnrf_warp_train_fwd')
    return out


def refine_input_train(rays, or_rays, depth_sorted, img4, poses, K, ref_nos, eps=1e-5, layout=0):
    rays = _chk(rays, 'rays', (11,)); or_rays = _chk(or_rays, 'or_rays', (11,)); depth_sorted = _chk(depth_sorted, 'depth_sorted', (8,))
    img4 = _chk(img4, 'img4', (4,)); poses = _chk(poses, 'poses', (3, 4)); K = _chk(K, 'K', (3, 3))
    if ref_nos.dtype != torch.int64 or not ref_nos.is_cuda or ref_nos.shape != (rays.shape[0], 4):
        raise PnrfError(f'refine_input_train: ref_nos must be a GPU int64 tensor [n,4], got {ref_nos.dtype} {tuple(ref_nos.shape)}')
    ref_nos = ref_nos.contiguous()
    nv, Hf, Wf, _ = img4.shape
    n = rays.shape[0]
    out = torch.empty(n, 144, device=rays.device, dtype=f32)
    check(_lib.load().pnrf_refine_input_train_fwd(_ptr(rays), _ptr(or_rays), _ptr(depth_sorted), _ptr(img4), _ptr(poses), _ptr(K), _ptr(ref_nos),
                                                  nv, 4, Hf, Wf, eps, int(layout), _ptr(out), n, _stream()), 'pnrf_refine_input_train_fwd')
    return out


def refine_train_fwd(mlp, refine_in, rays, depth_sorted, jitter=None, jitter_dir=1, want_rgb0=True):
    refine_in = _chk(refine_in, 'refine_in', (144,)); rays = _chk(rays, 'rays', (11,)); depth_sorted = _chk(depth_sorted, 'depth_sorted', (8,))
    jitter = None if jitter is None else _chk(jitter, 'jitter', (8,))
    n, dev = rays.shape[0], rays.device
    z = torch.empty(n, 8, device=dev, dtype=f32); pts = torch.empty(n, 8, 3, device=dev, dtype=f32)
    rgb0 = torch.empty(n, 3, device=dev, dtype=f32) if want_rgb0 else None
    check(_lib.load().pnrf_refine_train_fwd(mlp.handle, _ptr(refine_in), _ptr(rays), _ptr(depth_sorted), _ptr(jitter), int(jitter_dir),
                                            _ptr(z), _ptr(pts), _ptr(rgb0), n, _stream()), 'pnrf_refine_train_fwd')
    return z, pts, rgb0


def nerf_train_fwd(mlp, pts, rays, z=None, add=None, mul=None, noise=None, clamp=0.0, white_bkgd=False, want_raw=False):
    """pts [n,S,3].  S == 8: fused compositing -> (rgbd [n,4], raw|None).  S != 8: -> (None, raw [n,S,4])."""
    pts = _chk(pts, 'pts', (3,)); rays = _chk(rays, 'rays', (11,))
    n, S, dev = rays.shape[0], pts.shape[1], rays.device
    if pts.shape[0] != n:
        raise PnrfError(f'nerf_train_fwd: pts {tuple(pts.shape)} does not match {n} rays')
    z = None if z is None else _chk(z, 'z'); add = None if add is None else _chk(add, 'add'); mul = None if mul is None else _chk(mul, 'mul')
    noise = None if noise is None else _chk(noise, 'noise')
    fused = S == 8
    rgbd = torch.empty(n, 4, device=dev, dtype=f32) if fused else None
    raw = torch.empty(n, S, 4, device=dev, dtype=f32) if (want_raw or not fused) else None
    check(_lib.load().pnrf_nerf_train_fwd(mlp.handle, _ptr(pts), _ptr(rays), _ptr(z), _ptr(add), _ptr(mul), _ptr(noise), float(clamp),
                                          int(bool(white_bkgd)), int(S), _ptr(rgbd), _ptr(raw), n, _stream()), 'pnrf_nerf_train_fwd')
    return rgbd, raw


def explore(z8, rays, jitter, n_mult, dir1, dir2):
    """Stage-1 exploration: z8 [n,8], jitter [n, 8*n_mult] -> (z [n, 8*n_mult], pts [n, 8*n_mult, 3])."""
    z8 = _chk(z8, 'z8', (8,)); rays = _chk(rays, 'rays', (11,)); jitter = _chk(jitter, 'jitter', (8 * n_mult,))
    n, dev = rays.shape[0], rays.device
    z = torch.empty(n, 8 * n_mult, device=dev, dtype=f32); pts = torch.empty(n, 8 * n_mult, 3, device=dev, dtype=f32)
    check(_lib.load().pnrf_explore_fwd(_ptr(z8), _ptr(rays), _ptr(jitter), int(n_mult), int(dir1), int(dir2), _ptr(z), _ptr(pts), n, _stream()),
          'pnrf_explore_fwd')
    return z, pts


def images_pack(img_nchw):
    img = _chk(img_nchw, 'images')
    nv, c, Hf, Wf = img.shape
    if c != 3:
        raise PnrfError(f'images_pack: expected [nv,3,H,W], got {tuple(img.shape)}')
    out = torch.empty(nv, Hf, Wf, 4, device=img.device, dtype=f32)
    check(_lib.load().pnrf_images_pack(_ptr(img), _ptr(out), nv, Hf, Wf, _stream()), 'pnrf_images_pack')
    return out


def refine_input(rays, or_rays, depth_sorted, img4, proj, eps=1e-5):
    rays = _chk(rays, 'rays', (11,)); or_rays = _chk(or_rays, 'or_rays', (11,)); depth_sorted = _chk(depth_sorted, 'depth_sorted', (8,))
    img4 = _chk(img4, 'img4', (4,)); proj = _chk(proj, 'proj', (3, 4))
    nb, Hf, Wf, _ = img4.shape                   # any 1 .. 8 neighbour views: refine_in is [n, 48 + 24 nb] (nb = 4: 144)
    if proj.shape[0] != nb:
        raise PnrfError(f'refine_input: {nb} packed images but {proj.shape[0]} projection matrices')
    n = rays.shape[0]
    out = torch.empty(n, 48 + 24 * nb, device=rays.device, dtype=f32)
    check(_lib.load().pnrf_refine_input_fwd(_ptr(rays), _ptr(or_rays), _ptr(depth_sorted), _ptr(img4), _ptr(proj), nb, Hf, Wf, eps,
                                            _ptr(out), n, _stream()), 'pnrf_refine_input_fwd')
    return out


def composite(raw, z, rays_d, add=None, mul=None, noise=None, clamp=0.0, white_bkgd=False):
    raw = _chk(raw, 'raw', (4,)); z = _chk(z, 'z_vals'); rays_d = _chk(rays_d, 'rays_d', (3,))
    n, s = z.shape
    add = None if add is None else _chk(add, 'mm_density_add')
    mul = None if mul is None else _chk(mul, 'mm_density_mul')
    noise = None if noise is None else _chk(noise, 'noise')
    dev = raw.device
    rgb = torch.empty(n, 3, device=dev, dtype=f32); disp = torch.empty(n, device=dev, dtype=f32)
    acc = torch.empty(n, device=dev, dtype=f32); w = torch.empty(n, s, device=dev, dtype=f32); depth = torch.empty(n, device=dev, dtype=f32)
    check(_lib.load().pnrf_composite_fwd(_ptr(raw), _ptr(z), _ptr(rays_d), 3, _ptr(add), _ptr(mul), _ptr(noise), float(clamp),
                                         int(bool(white_bkgd)), _ptr(rgb), _ptr(disp), _ptr(acc), _ptr(w), _ptr(depth), n, s, _stream()),
          'pnrf_composite_fwd')
    return rgb, disp, acc, w, depth


def sampler_fwd(mlp: PackedMLP, rays, want_idx=True, want_rgb=True, want_raw=False, two_pass=False, kappa=None):
    """pnrf_sampler_fwd; two_pass=True: pnrf_sampler_fwd_ws (plain-fp16 pass for every ray + split-fp16 pass for the undecided ones, what
    the fused path runs) and the numbers of rays of the second (split fp16) and third (exact fp32: saturated activations) pass as a seventh
    and eighth return value (0-d int32 tensors on the device)."""
    rays = _chk(rays, 'rays', (11,))
    n, dev = rays.shape[0], rays.device
    depth = torch.empty(n, 8, device=dev, dtype=f32); add = torch.empty_like(depth); mul = torch.empty_like(depth)
    idx = torch.empty(n, 8, device=dev, dtype=torch.int64) if want_idx else None
    rgb = torch.empty(n, 3, device=dev, dtype=f32) if want_rgb else None
    draw = torch.empty(n, 8, device=dev, dtype=f32) if want_raw else None
    if two_pass:
        lib = _lib.load()
        nb = int(lib.pnrf_sampler_workspace_bytes(n))
        ws = torch.empty(max(nb, 64) // 4, device=dev, dtype=torch.int32)
        check(lib.pnrf_sampler_fwd_ws(mlp.handle, _ptr(rays), n, _ptr(depth), _ptr(add), _ptr(mul), _ptr(idx), _ptr(rgb), _ptr(draw), _ptr(ws), nb,
                                      -1.0 if kappa is None else float(kappa), _stream()), 'pnrf_sampler_fwd_ws')
        return depth, idx, add, mul, rgb, draw, ws[1], ws[5]
    check(_lib.load().pnrf_sampler_fwd(mlp.handle, _ptr(rays), n, _ptr(depth), _ptr(add), _ptr(mul), _ptr(idx), _ptr(rgb), _ptr(draw), _stream()),
          'pnrf_sampler_fwd')
    return depth, idx, add, mul, rgb, draw


def refine_fwd(mlp: PackedMLP, refine_in, rays, depth_sorted):
    refine_in = _chk(refine_in, 'refine_in', (mlp.in_dim,)); rays = _chk(rays, 'rays', (11,)); depth_sorted = _chk(depth_sorted, 'depth_sorted', (8,))
    n, dev = rays.shape[0], rays.device
    z = torch.empty(n, 8, device=dev, dtype=f32); pts = torch.empty(n, 8, 3, device=dev, dtype=f32)
    check(_lib.load().pnrf_refine_fwd(mlp.handle, _ptr(refine_in), _ptr(rays), _ptr(depth_sorted), _ptr(z), _ptr(pts), n, _stream()), 'pnrf_refine_fwd')
    return z, pts


def refine_project_fwd(mlp: PackedMLP, rays, or_rays, depth_sorted, img4, proj, eps=1e-5):
    """Projection into the neighbour views + refine MLP + interval refinement in one kernel (pnrf_refine_project_fwd) -> (z [n,8], pts [n,8,3])."""
    rays = _chk(rays, 'rays', (11,)); or_rays = _chk(or_rays, 'or_rays', (11,)); depth_sorted = _chk(depth_sorted, 'depth_sorted', (8,))
    img4 = _chk(img4, 'img4', (4,)); proj = _chk(proj, 'proj', (3, 4))
    nb, Hf, Wf, _ = img4.shape
    if proj.shape[0] != nb:
        raise PnrfError(f'refine_project_fwd: {nb} packed images but {proj.shape[0]} projection matrices')
    n, dev = rays.shape[0], rays.device
    z = torch.empty(n, 8, device=dev, dtype=f32); pts = torch.empty(n, 8, 3, device=dev, dtype=f32)
    check(_lib.load().pnrf_refine_project_fwd(mlp.handle, _ptr(rays), _ptr(or_rays), _ptr(depth_sorted), _ptr(img4), _ptr(proj), nb, Hf, Wf, eps,
                                              _ptr(z), _ptr(pts), n, _stream()), 'pnrf_refine_project_fwd')
    return z, pts


def nerf_fwd(mlp: PackedMLP, pts, rays, z, add, mul, want_raw=False):
    pts = _chk(pts, 'pts', (8, 3)); rays = _chk(rays, 'rays', (11,)); z = _chk(z, 'z', (8,)); add = _chk(add, 'add', (8,)); mul = _chk(mul, 'mul', (8,))
    n, dev = rays.shape[0], rays.device
    rgbd = torch.empty(n, 4, device=dev, dtype=f32)
    raw = torch.empty(n, 8, 4, device=dev, dtype=f32) if want_raw else None
    check(_lib.load().pnrf_nerf_fwd(mlp.handle, _ptr(pts), _ptr(rays), _ptr(z), _ptr(add), _ptr(mul), _ptr(rgbd), _ptr(raw), n, _stream()), 'pnrf_nerf_fwd')
    return rgbd, raw


class RenderContext:
    """pnrf_ctx: per-ray workspace for the whole inference path (sized once)."""

    def __init__(self, sampler: PackedMLP, refine: PackedMLP, nerf: PackedMLP, max_rays: int):
        self._keep = (sampler, refine, nerf)
        h = C.c_void_p()
        check(_lib.load().pnrf_ctx_create(sampler.handle, refine.handle, nerf.handle, int(max_rays), C.byref(h)), 'pnrf_ctx_create')
        self.handle = h
        self.max_rays = int(max_rays)

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                _lib.load().pnrf_ctx_free(self.handle)
                self.handle = None
        except Exception:
            pass

    STAGES = ('sampler_kernel', 'refine_kernel', 'nerf_kernel')

    def sampler_stats(self):
        """Rays the sampler's second (split-fp16) pass rendered in the most recent render_rays call; waits for the device."""
        v = C.c_int64()
        check(_lib.load().pnrf_ctx_sampler_stats(self.handle, C.byref(v)), 'pnrf_ctx_sampler_stats')
        return int(v.value)

    def sampler_saturated(self):
        """Rays the exact-fp32 third pass rendered in the most recent render_rays call (activations at the fp16 limit); waits for the device."""
        v = C.c_int64()
        check(_lib.load().pnrf_ctx_sampler_saturated(self.handle, C.byref(v)), 'pnrf_ctx_sampler_saturated')
        return int(v.value)

    def set_sampler_kappa(self, kappa):
        """Threshold of the two-pass sampler for this context (pnrf_ctx_set_sampler_kappa): negative = the library default (PNRF_SAMPLER_KAPPA of
        include/pronerf_hip.h; ``sampler_kappa()`` reports it), 0 = only the fp32 round-off allowance; NaN / inf are refused."""
        check(_lib.load().pnrf_ctx_set_sampler_kappa(self.handle, float(kappa)), 'pnrf_ctx_set_sampler_kappa')

    def sampler_kappa(self):
        """The threshold this context's calls run with (pnrf_ctx_get_sampler_kappa)."""
        v = C.c_float()
        check(_lib.load().pnrf_ctx_get_sampler_kappa(self.handle, C.byref(v)), 'pnrf_ctx_get_sampler_kappa')
        return float(v.value)

    def profile_begin(self, max_frames=64):
        """Record per-stage events on the next ``max_frames`` render_rays calls (pnrf_ctx_profile_begin)."""
        check(_lib.load().pnrf_ctx_profile_begin(self.handle, int(max_frames)), 'pnrf_ctx_profile_begin')

    def profile_end(self):
        """-> ({stage: mean ms}, frames recorded); waits for the last recorded call."""
        ms = (C.c_float * len(self.STAGES))()
        frames = C.c_int()
        check(_lib.load().pnrf_ctx_profile_end(self.handle, ms, C.byref(frames)), 'pnrf_ctx_profile_end')
        return dict(zip(self.STAGES, (float(v) for v in ms))), frames.value

    def render_rays(self, rays, or_rays, img4, proj, eps=1e-5, want_idx=False, out=None):
        rays = _chk(rays, 'rays', (11,)); or_rays = _chk(or_rays, 'or_rays', (11,))
        img4 = _chk(img4, 'img4', (4,)); proj = _chk(proj, 'proj', (3, 4))
        n = rays.shape[0]
        nb, Hf, Wf, _ = img4.shape
        if proj.shape[0] != nb:
            raise PnrfError(f'render_rays: {nb} packed neighbour images but {proj.shape[0]} projection matrices')
        if out is not None:
            if not isinstance(out, torch.Tensor) or out.device != rays.device or out.dtype != f32 or tuple(out.shape) != (n, 4) or not out.is_contiguous():
                raise PnrfError(f'render_rays: out must be a contiguous float32 tensor [{n}, 4] on {rays.device}, got '
                                f'{getattr(out, "dtype", type(out))} {tuple(getattr(out, "shape", ()))} on {getattr(out, "device", "?")}')
        rgbd = out if out is not None else torch.empty(n, 4, device=rays.device, dtype=f32)
        idx = torch.empty(n, 8, device=rays.device, dtype=torch.int64) if want_idx else None
        check(_lib.load().pnrf_render_rays_fwd(self.handle, _ptr(rays), _ptr(or_rays), _ptr(img4), _ptr(proj), nb, Hf, Wf, eps,
                                               _ptr(rgbd), _ptr(idx), n, _stream()), 'pnrf_render_rays_fwd')
        return rgbd, idx


def linspace(start, end, n):
    out = (C.c_float * n)()
    check(_lib.load().pnrf_linspace(start, end, n, out), 'pnrf_linspace')
    return np.array(out[:], dtype=np.float32)


# ------------------------------------------------------------------------------------------ stage-2 training step
def composite_bwd(raw, z, rays_d, d_rgb, add=None, mul=None, noise=None, clamp=0.0, white_bkgd=False):
    """raw2outputs backward for d rgb_map -> (d_raw, d_z, d_add, d_mul); d_add / d_mul are None without add / mul."""
    raw = _chk(raw, 'raw', (4,)); z = _chk(z, 'z_vals'); rays_d = _chk(rays_d, 'rays_d', (3,)); d_rgb = _chk(d_rgb, 'd_rgb', (3,))
    n, s = z.shape
    add = None if add is None else _chk(add, 'mm_density_add')
    mul = None if mul is None else _chk(mul, 'mm_density_mul')
    noise = None if noise is None else _chk(noise, 'noise')
    d_raw = torch.empty_like(raw); d_z = torch.empty_like(z)
    d_add = None if add is None else torch.empty_like(z)
    d_mul = None if mul is None else torch.empty_like(z)
    check(_lib.load().pnrf_composite_bwd(_ptr(raw), _ptr(z), _ptr(rays_d), 3, _ptr(add), _ptr(mul), _ptr(noise), float(clamp), int(bool(white_bkgd)),
                                         _ptr(d_rgb), _ptr(d_raw), _ptr(d_z), _ptr(d_add), _ptr(d_mul), n, s, _stream()), 'pnrf_composite_bwd')
    return d_raw, d_z, d_add, d_mul


def posenc_bwd(x, d_out, n_freq):
    x = _chk(x, 'x', (3,)); d_out = _chk(d_out, 'd_out', (3 + 6 * n_freq,))
    d_x = torch.empty_like(x)
    check(_lib.load().pnrf_posenc_bwd(_ptr(x), _ptr(d_out), _ptr(d_x), x.numel() // 3, n_freq, _stream()), 'pnrf_posenc_bwd')
    return d_x


def sampler_head_fwd(y, rays):
    y = _chk(y, 'y', (27,)); rays = _chk(rays, 'rays', (11,))
    n, dev = y.shape[0], y.device
    depth = torch.empty(n, 8, device=dev, dtype=f32); add = torch.empty_like(depth); mul = torch.empty_like(depth)
    idx = torch.empty(n, 8, device=dev, dtype=torch.int64); rgb = torch.empty(n, 3, device=dev, dtype=f32)
    check(_lib.load().pnrf_sampler_head_fwd(_ptr(y), _ptr(rays), _ptr(depth), _ptr(idx), _ptr(add), _ptr(mul), _ptr(rgb), n, _stream()), 'pnrf_sampler_head_fwd')
    return depth, idx, add, mul, rgb


def sampler_head_bwd(y, rays, idx, d_depth, d_add, d_mul, d_rgb=None):
    y = _chk(y, 'y', (27,)); rays = _chk(rays, 'rays', (11,))
    d_y = torch.empty_like(y)
    check(_lib.load().pnrf_sampler_head_bwd(_ptr(y), _ptr(rays), _ptr(idx.contiguous()), _ptr(_chk(d_depth, 'd_depth', (8,))), _ptr(_chk(d_add, 'd_add', (8,))),
                                            _ptr(_chk(d_mul, 'd_mul', (8,))), _ptr(d_rgb), _ptr(d_y), y.shape[0], _stream()), 'pnrf_sampler_head_bwd')
    return d_y


def refine_head_fwd(y, rays, depth_sorted, jitter=None, jitter_dir=1):
    y = _chk(y, 'y', (35,)); rays = _chk(rays, 'rays', (11,)); depth_sorted = _chk(depth_sorted, 'depth_sorted', (8,))
    n, dev = y.shape[0], y.device
    z_pre = torch.empty(n, 8, device=dev, dtype=f32); z = torch.empty_like(z_pre)
    pts = torch.empty(n, 8, 3, device=dev, dtype=f32); rgb0 = torch.empty(n, 3, device=dev, dtype=f32)
    jitter = None if jitter is None else _chk(jitter, 'jitter', (8,))
    check(_lib.load().pnrf_refine_head_fwd(_ptr(y), _ptr(rays), _ptr(depth_sorted), _ptr(jitter), int(jitter_dir), _ptr(z_pre), _ptr(z), _ptr(pts), _ptr(rgb0),
                                           n, _stream()), 'pnrf_refine_head_fwd')
    return z_pre, z, pts, rgb0


def refine_head_bwd(y, rays, depth_sorted, z_pre, d_pts, d_z=None, d_rgb0=None, jitter=None, jitter_dir=1):
    y = _chk(y, 'y', (35,)); rays = _chk(rays, 'rays', (11,)); depth_sorted = _chk(depth_sorted, 'depth_sorted', (8,))
    d_y = torch.empty_like(y); d_depth = torch.empty_like(depth_sorted)
    jitter = None if jitter is None else _chk(jitter, 'jitter', (8,))
    check(_lib.load().pnrf_refine_head_bwd(_ptr(y), _ptr(rays), _ptr(depth_sorted), _ptr(_chk(z_pre, 'z_pre', (8,))), _ptr(jitter), int(jitter_dir),
                                           _ptr(_chk(d_pts, 'd_pts', (3,))), _ptr(d_z), _ptr(d_rgb0), _ptr(d_y), _ptr(d_depth), y.shape[0], _stream()),
          'pnrf_refine_head_bwd')
    return d_y, d_depth


TRAINER_LAYERS = 26          # 7 sampler + 7 refine + 12 NeRF class (pts_linears.0..7, feature, alpha, views, rgb)


class Trainer:
    """fp32 parameters, gradients, Adam state and workspaces of the stage-2 training step (pnrf_trainer_*).

    weights / biases: 26 arrays in the order of ``TRAINER_LAYERS`` (torch layout ``W[out, in]``)."""

    def __init__(self, weights, biases, max_rays, device='cuda:0', max_samples=8):
        lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise PnrfError('Trainer needs a GPU device (pronerf_amd has no CPU path)')
        n = len(weights)
        ws = [np.ascontiguousarray(w.detach().cpu().numpy() if isinstance(w, torch.Tensor) else w, dtype=np.float32) for w in weights]
        bs = [np.ascontiguousarray(b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else b, dtype=np.float32) for b in biases]
        self.shapes = [w.shape for w in ws]
        Wp = (C.c_void_p * n)(*[w.ctypes.data for w in ws]); bp = (C.c_void_p * n)(*[b.ctypes.data for b in bs])
        ind = (C.c_int * n)(*[w.shape[1] for w in ws]); outd = (C.c_int * n)(*[w.shape[0] for w in ws])
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(lib.pnrf_trainer_create(Wp, bp, ind, outd, n, int(max_rays), int(max_samples), C.byref(h)), 'pnrf_trainer_create')
        self.handle = h
        self.max_rays = int(max_rays)
        self.max_samples = int(max_samples)

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                _lib.load().pnrf_trainer_free(self.handle)
                self.handle = None
        except Exception:
            pass

    def read(self, kind, layer):
        """kind: 'param' | 'grad' | 'm' | 'v' (joint Adam) | 'm_nerf' | 'v_nerf' (NeRF-only Adam) -> (W, b) as new device tensors."""
        k = self._KINDS[kind]
        out_d, in_d = self.shapes[layer]
        W = torch.empty(out_d, in_d, device=self.device, dtype=f32); b = torch.empty(out_d, device=self.device, dtype=f32)
        with torch.cuda.device(self.device):
            check(_lib.load().pnrf_trainer_read(self.handle, k, layer, _ptr(W), _ptr(b), _stream()), 'pnrf_trainer_read')
        return W, b

    def write(self, kind, layer, W=None, b=None):
        k = self._KINDS[kind]
        W = None if W is None else torch.as_tensor(W, dtype=f32).contiguous()
        b = None if b is None else torch.as_tensor(b, dtype=f32).contiguous()
        with torch.cuda.device(self.device):
            check(_lib.load().pnrf_trainer_write(self.handle, k, layer, _ptr(W), _ptr(b), _stream()), 'pnrf_trainer_write')

    _KINDS = {'param': 0, 'grad': 1, 'm': 2, 'v': 3, 'm_nerf': 4, 'v_nerf': 5}

    def flat(self, kind='grad'):
        """Zero-copy torch view [nparam] of one of the trainer's flat device arrays (all layers, trainer order) — e.g. for
        ``torch.distributed.all_reduce`` of the gradients of data-parallel replicas (pronerf_amd.dist.allreduce_gradients)."""
        ptr, n = C.c_void_p(), C.c_int64()
        check(_lib.load().pnrf_trainer_flat(self.handle, self._KINDS[kind], C.byref(ptr), C.byref(n)), 'pnrf_trainer_flat')

        class _Dev:                                      # __cuda_array_interface__ v3: torch wraps the memory without copying
            __cuda_array_interface__ = {'shape': (int(n.value),), 'typestr': '<f4', 'data': (int(ptr.value), False), 'version': 3}
        with torch.cuda.device(self.device):
            t = torch.as_tensor(_Dev(), device=self.device)
        t._pnrf_owner = self                             # keep the trainer alive as long as the view is
        return t

    def set_dw_kernel(self, tile=0, min_rows_128=0):
        """Weight-gradient kernel of the square layers: 0 = by shape and row count, 64 / 128 = forced; 256 / 255 = the grouped gradients' 256 x 128 tiles
        on from ``min_rows_128`` rows / off (pnrf_trainer_set_dw_kernel)."""
        check(_lib.load().pnrf_trainer_set_dw_kernel(self.handle, int(tile), int(min_rows_128)), 'pnrf_trainer_set_dw_kernel')

    def dw_group_info(self):
        """(gradients in the most recent iteration's grouped weight-gradient launch, bit mask of those that ran on 256 x 128 tiles)
        (pnrf_trainer_dw_group_info)."""
        n, m = C.c_int(), C.c_uint()
        check(_lib.load().pnrf_trainer_dw_group_info(self.handle, C.byref(n), C.byref(m)), 'pnrf_trainer_dw_group_info')
        return int(n.value), int(m.value)

    def set_graph(self, enable=True):
        """Replay the iterations as hipGraphs, or (default) launch their kernels one by one (pnrf_trainer_set_graph)."""
        check(_lib.load().pnrf_trainer_set_graph(self.handle, int(bool(enable))), 'pnrf_trainer_set_graph')

    def set_products(self, kind):
        """'f16x2' (default): split-fp16 MFMA layer products (fp32-grade), the fine net's forward as one launch on the fused-MLP engine; 'f32':
        exact-fp32 MFMA products; 'f16x2_unchained': split fp16 with one launch per layer of the fine net's forward; 'f16x2_wchain': with its
        256 -> 256 layers as two 64-row layer chains (bit-identical to 'f16x2_unchained'; pnrf_trainer_set_products)."""
        kinds = {'f16x2': 0, 'f32': 1, 'f16x2_unchained': 2, 'f16x2_wchain': 3}
        if kind not in kinds:
            raise PnrfError(f"Trainer.set_products: kind must be one of {sorted(kinds)} ('f16x2': split-fp16 MFMA products, the default; 'f32': exact-fp32 "
                            f'MFMA products), got {kind!r} (the training drivers read it from PNRF_TRAIN_PRODUCTS)')
        k = kinds[kind]
        check(_lib.load().pnrf_trainer_set_products(self.handle, k), 'pnrf_trainer_set_products')

    def set_step(self, step, step_nerf=0):
        check(_lib.load().pnrf_trainer_set_step(self.handle, int(step), int(step_nerf)), 'pnrf_trainer_set_step')

    def _batch(self, rays, or_rays, target, img4, poses, K, ref_nos, jitter, jitter_dir, raw_noise, white_bkgd, eps, a_mmrgb, clamp, layout, S):
        rays = _chk(rays, 'rays', (11,)); or_rays = _chk(or_rays, 'or_rays', (11,)); target = _chk(target, 'target', (3,))
        img4 = _chk(img4, 'img4', (4,)); poses = _chk(poses, 'poses', (3, 4)); K = _chk(K, 'K', (3, 3))
        if ref_nos.dtype != torch.int64 or not ref_nos.is_cuda or tuple(ref_nos.shape) != (rays.shape[0], 4):
            raise PnrfError('ref_nos: expected an int64 GPU tensor [n, 4]')
        ref_nos = ref_nos.contiguous()
        jitter = None if jitter is None else _chk(jitter, 'jitter', (S,))
        raw_noise = None if raw_noise is None else _chk(raw_noise, 'raw_noise', (S,))
        keep = (rays, or_rays, target, img4, poses, K, ref_nos, jitter, raw_noise)          # alive until the call returns
        bt = _lib.TrainBatch(rays=rays.data_ptr(), or_rays=or_rays.data_ptr(), target=target.data_ptr(), img4=img4.data_ptr(), poses=poses.data_ptr(),
                             K=K.data_ptr(), ref_nos=ref_nos.data_ptr(), jitter=None if jitter is None else jitter.data_ptr(),
                             raw_noise=None if raw_noise is None else raw_noise.data_ptr(), n=rays.shape[0], nv=img4.shape[0], Hf=img4.shape[1],
                             Wf=img4.shape[2], jitter_dir=int(jitter_dir), white_bkgd=int(bool(white_bkgd)), eps=float(eps), a_mmrgb=float(a_mmrgb),
                             clamp=float(clamp), layout=int(layout))
        return bt, keep

    def fwd_bwd(self, rays, or_rays, target, img4, poses, K, ref_nos, jitter=None, jitter_dir=1, raw_noise=None, white_bkgd=False, eps=1e-5,
                a_mmrgb=0.0, want_rgb=True, clamp=0.0, layout=0):
        """One joint forward + backward (stage 2; stage-1 even iterations with clamp=10, layout=1, eps=1e-6, a_mmrgb=1);
        returns (loss[4] device tensor = total, img, rgb0, mm_rgb; rgb_map1 [n,3] or None)."""
        bt, keep = self._batch(rays, or_rays, target, img4, poses, K, ref_nos, jitter, jitter_dir, raw_noise, white_bkgd, eps, a_mmrgb, clamp, layout, 8)
        dev = keep[0].device
        loss = torch.empty(4, device=dev, dtype=f32)
        rgb = torch.empty(keep[0].shape[0], 3, device=dev, dtype=f32) if want_rgb else None
        with torch.cuda.device(dev):
            check(_lib.load().pnrf_train_stage2_fwd_bwd(self.handle, C.byref(bt), _ptr(loss), _ptr(rgb), _stream()), 'pnrf_train_stage2_fwd_bwd')
        return loss, rgb

    def explore_fwd_bwd(self, rays, or_rays, target, img4, poses, K, ref_nos, n_mult, dir1, jitter, dir2, raw_noise=None, white_bkgd=False, eps=1e-6,
                        clamp=10.0, layout=1, want_rgb=True):
        """Stage-1 odd iteration: NeRF-only forward + backward on 8*n_mult explored samples per ray (jitter, raw_noise: [n, 8*n_mult])."""
        bt, keep = self._batch(rays, or_rays, target, img4, poses, K, ref_nos, jitter, dir2, raw_noise, white_bkgd, eps, 0.0, clamp, layout, 8 * int(n_mult))
        dev = keep[0].device
        loss = torch.empty(4, device=dev, dtype=f32)
        rgb = torch.empty(keep[0].shape[0], 3, device=dev, dtype=f32) if want_rgb else None
        with torch.cuda.device(dev):
            check(_lib.load().pnrf_train_explore_fwd_bwd(self.handle, C.byref(bt), int(n_mult), int(dir1), _ptr(loss), _ptr(rgb), _stream()),
                  'pnrf_train_explore_fwd_bwd')
        return loss, rgb

    def adam_step(self, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, nerf_only=False):
        """optimizer.step(); nerf_only: the stage-1 NeRF-only optimizer (own moments / step count) instead of the joint one."""
        with torch.cuda.device(self.device):
            check(_lib.load().pnrf_trainer_adam_step(self.handle, int(bool(nerf_only)), float(lr), float(betas[0]), float(betas[1]), float(eps),
                                                     float(weight_decay), _stream()), 'pnrf_trainer_adam_step')
