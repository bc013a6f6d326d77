"""TEST INFRASTRUCTURE ONLY — CPU fp32 restatement of the ProNeRF render hot path.

This file restates, from scratch, the algorithm of KAIST-VICLab/pronerf's inference
``render_rays`` (run_S_eS_eN_alter_trt.py:599-696) and the operators it calls.  It is
the *checker* for the HIP path: only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it.  The product path
(``pronerf_amd``) never does and fails loudly when its HIP library is missing.

Parity status: PINNED.  The reference has no tests/golden vectors of its own
(SURVEY.md §4); this restatement is pinned against outputs of the reference itself,
imported on CPU in the build container by ``oracle/gen_golden.py`` and committed as
``tests/golden/*.npz`` (``tests/test_oracle_golden.py`` replays them).

All tensors are fp32 torch CPU tensors (the reference is fp32 torch; the GEMMs here
go through ``torch.nn.functional.linear`` exactly like ``nn.Linear`` there, so the
agreement with the reference is at fp32 round-off).  Each function cites the
reference ``file:line`` it follows (paths relative to the reference root).
"""
from __future__ import annotations

import math
import torch
import torch.nn.functional as F

f32 = torch.float32


def _t(x):
    return x if isinstance(x, torch.Tensor) else torch.as_tensor(x, dtype=f32)


# ----------------------------------------------------------------------------- rays
def get_rays(H, W, K, c2w):
    """Pixel grid -> world rays.  run_nerf_helpers.py:2705-2714 (get_rays)."""
    K, c2w = _t(K), _t(c2w)
    jj, ii = torch.meshgrid(torch.arange(H, dtype=f32), torch.arange(W, dtype=f32), indexing='ij')
    dirs = torch.stack([(ii - K[0, 2]) / K[0, 0], -(jj - K[1, 2]) / K[1, 1], -torch.ones_like(ii)], -1)
    rays_d = (dirs[..., None, :] * c2w[:3, :3]).sum(-1)
    rays_o = c2w[:3, 3].expand(rays_d.shape)
    return rays_o, rays_d


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """Shift to the near plane and project to NDC.  run_nerf_helpers.py:2776-2793."""
    t = -(near + rays_o[..., 2]) / rays_d[..., 2]
    o = rays_o + t[..., None] * rays_d
    sx = -1.0 / (W / (2.0 * focal))
    sy = -1.0 / (H / (2.0 * focal))
    o0 = sx * o[..., 0] / o[..., 2]
    o1 = sy * o[..., 1] / o[..., 2]
    o2 = 1.0 + 2.0 * near / o[..., 2]
    d0 = sx * (rays_d[..., 0] / rays_d[..., 2] - o[..., 0] / o[..., 2])
    d1 = sy * (rays_d[..., 1] / rays_d[..., 2] - o[..., 1] / o[..., 2])
    d2 = -2.0 * near / o[..., 2]
    return torch.stack([o0, o1, o2], -1), torch.stack([d0, d1, d2], -1)


def pluecker(o, d):
    """[normalize(d), o x normalize(d)].  run_nerf_helpers.py:629-632 (Pluecker.forward)."""
    dn = d / d.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    m = torch.linalg.cross(o, dn, dim=-1)
    return torch.cat([dn, m], -1)


def ray_points(o, d, t0, t1, n):
    """o + t*d at t=linspace(t0,t1,n).  run_S_eS_eN_alter_trt.py:546-562."""
    t = torch.linspace(t0, t1, n, dtype=f32).to(o.device)      # computed on CPU: the reference's linspace runs there too before .to()
    return o[..., None, :] + d[..., None, :] * t[None, :, None]


def mm_input_from_rays(o, d, n_pts=48):
    """Sampler input [N, 6*n_pts].  run_S_eS_eN_alter_trt.py:274-277."""
    pts = ray_points(o, d, 0.0, 1.0, n_pts)
    pl = pluecker(pts, d[:, None, :].expand(-1, n_pts, -1))
    return pl.reshape(o.shape[0], 6 * n_pts)


def posenc(x, n_freq):
    """[x, sin(2^k x), cos(2^k x)]_k.  run_nerf_helpers.py:666-671 / :677-692."""
    out = [x]
    for k in range(n_freq):
        f = float(2.0 ** k)
        out.append(torch.sin(x * f))
        out.append(torch.cos(x * f))
    return torch.cat(out, -1)


# ----------------------------------------------------------------------------- MLPs
def _lin(x, W, b):
    return F.linear(x, _t(W), _t(b))


def mlp_elu_backbone(x, Ws, bs):
    """6x[Linear+ELU] + Linear, no skip (mmnetskips=[10000]).  run_nerf_helpers.py:1490-1497."""
    h = x
    for W, b in zip(Ws[:-1], bs[:-1]):
        h = F.elu(_lin(h, W, b))
    return _lin(h, Ws[-1], bs[-1])


def sampler_forward(w, mm_input, n_samples=8):
    """MinMaxRaySamplerTRT_Net.forward.  run_nerf_helpers.py:1490-1507.
    Returns (mm_rgb, density_add, density_mul, depth)."""
    S = n_samples
    y = mlp_elu_backbone(mm_input, w['W'], w['b'])
    return torch.sigmoid(y[:, 3 * S:]), y[:, S:2 * S], y[:, 2 * S:3 * S], torch.sigmoid(y[:, :S])


def refine_forward(w, x, n_samples=8):
    """MinMaxRayEpiSamplerTRT_Net.forward.  run_nerf_helpers.py:1526-1540.
    Returns (refine_depth, refine_rgb, points_offset)."""
    S = n_samples
    y = mlp_elu_backbone(x, w['W'], w['b'])
    return torch.sigmoid(y[:, :S]), torch.sigmoid(y[:, 4 * S:]), torch.tanh(y[:, S:4 * S])


def nerf_forward(w, emb_pts, emb_dirs):
    """DoNeRFTRT.forward (skip='auto': view encoding concatenated before the last layer).
    run_nerf_helpers.py:1331-1343; layer table :1190-1239."""
    h = emb_pts
    n = len(w['W'])
    for i in range(n):
        if i == n - 1:
            h = torch.cat([h, emb_dirs], -1)
        h = _lin(h, w['W'][i], w['b'][i])
        if i + 1 < n:
            h = F.relu(h)
    return h


def nerfcls_forward(w, x, input_ch=63, input_ch_views=27, skips=(4,)):
    """NeRF.forward (use_viewdirs=True).  run_nerf_helpers.py:824-847."""
    pts, views = x[..., :input_ch], x[..., input_ch:input_ch + input_ch_views]
    h = pts
    for i, (W, b) in enumerate(w['pts_linears']):
        h = F.relu(_lin(h, W, b))
        if i in skips:
            h = torch.cat([pts, h], -1)
    alpha = _lin(h, *w['alpha_linear'])
    feat = _lin(h, *w['feature_linear'])
    h = torch.cat([feat, views], -1)
    h = F.relu(_lin(h, *w['views_linears'][0]))
    rgb = _lin(h, *w['rgb_linear'])
    return torch.cat([rgb, alpha], -1)


# ----------------------------------------------------------------------------- sort
def sort_gather(depth, add, mul, near, far):
    """depth affine + ascending sort + permute add/mul.  run_S_eS_eN_alter_trt.py:631-635.
    ``torch.sort`` on CPU is stable (ties: lower original index first)."""
    depth = depth * (far - near) + near
    d_sorted, idx = torch.sort(depth, dim=-1, stable=True)
    return d_sorted, idx, torch.gather(add, 1, idx), torch.gather(mul, 1, idx)


# ----------------------------------------------------------------------------- projection
def bilinear_zeros(img, X, Y):
    """Bilinear fetch at pixel coordinates, zero padding per tap — what
    ``grid_sample(bilinear, padding_mode='zeros', align_corners=True)`` computes after the
    reference's normalisation.  inverse_warp.py:607-615.

    img [C,Hf,Wf]; X,Y [n]  ->  [C,n].  The normalise/unnormalise round trip of the
    reference (2X/(Wf-1)-1, then ((x+1)/2)*(Wf-1)) is replayed so that fp32 rounding agrees.
    """
    C, Hf, Wf = img.shape
    xn = 2 * X / (Wf - 1) - 1
    yn = 2 * Y / (Hf - 1) - 1
    ix = ((xn + 1) / 2) * (Wf - 1)
    iy = ((yn + 1) / 2) * (Hf - 1)
    x0 = torch.floor(ix); y0 = torch.floor(iy)
    x1 = x0 + 1; y1 = y0 + 1
    wx1 = ix - x0; wx0 = x1 - ix
    wy1 = iy - y0; wy0 = y1 - iy
    flat = img.reshape(C, Hf * Wf)
    finite = torch.isfinite(ix) & torch.isfinite(iy)

    def tap(xx, yy, wgt):
        ok = finite & (xx >= 0) & (xx <= Wf - 1) & (yy >= 0) & (yy <= Hf - 1)
        xi = torch.where(ok, xx, torch.zeros_like(xx)).long()
        yi = torch.where(ok, yy, torch.zeros_like(yy)).long()
        v = flat[:, yi * Wf + xi]
        return torch.where(ok, wgt, torch.zeros_like(wgt))[None] * v

    return tap(x0, y0, wx0 * wy0) + tap(x1, y0, wx1 * wy0) + tap(x0, y1, wx0 * wy1) + tap(x1, y1, wx1 * wy1)


def project_trt(images_nchw, proj, or_o, or_d, depth_ndc, eps=1e-5):
    """NDC depth -> metric depth -> world point -> pixel in each neighbour -> bilinear RGB.
    run_S_eS_eN_alter_trt.py:637-655 + inverse_warp.py:584-619.

    images_nchw [NB,3,Hf,Wf]; proj [NB,3,4] (= K.diag(1,-1,-1).pose, trt.py:289-294);
    or_o/or_d [N,3] un-normalised camera rays; depth_ndc [N,S] sorted sampler depths.
    Returns epi [N, NB*S*3] with index (k*S+s)*3+c (neighbour-major, trt.py:653-655).
    Non-finite pixel coordinates (p.z == 0) contribute 0 — the HIP kernel's defined behaviour.
    """
    N, S = depth_ndc.shape
    NB = images_nchw.shape[0]
    z3d = 1.0 / (1.0 - depth_ndc - eps)                                     # trt.py:637
    ro1 = torch.cat([or_o, torch.ones(N, 1, device=or_o.device)], -1)       # trt.py:256-258
    rd1 = torch.cat([or_d, torch.zeros(N, 1, device=or_o.device)], -1)
    out = torch.zeros(N, NB, S, 3, device=or_o.device)
    for k in range(NB):
        for s in range(S):
            w = ro1 + rd1 * z3d[:, s:s + 1]                                 # inverse_warp.py:600
            p = w @ proj[k].T                                               # inverse_warp.py:601 (bmm)
            X = p[:, 0] / p[:, 2]; Y = p[:, 1] / p[:, 2]                    # :603-605
            out[:, k, s, :] = bilinear_zeros(images_nchw[k], X, Y).T
    return out.reshape(N, NB * S * 3)


# ----------------------------------------------------------------------------- refine / composite
def interval_refine(depth_sorted, refine, near, far):
    """z = lower + (upper-lower)*refine over the sorted-depth midpoints.  trt.py:673-677."""
    mids = 0.5 * (depth_sorted[..., 1:] + depth_sorted[..., :-1])
    upper = torch.cat([mids, 0.5 * (far + depth_sorted[..., -1:])], -1)
    lower = torch.cat([0.5 * (near + depth_sorted[..., :1]), mids], -1)
    return lower + (upper - lower) * refine


def raw2outputs(raw, z_vals, rays_d, add=None, mul=None, noise=None, clamp=0.0, white_bkgd=False):
    """Alpha compositing with the sampler's density modulation.
    Infer variant: run_S_eS_eN_alter_trt.py:564-597.  Training variants (clamp +-10,
    additive sigma noise, white background; run_S_eS_eN_alter_base.py:501-551,
    run_S_eS_eN_alter_base_refine2.py:475-522) are selected by the keyword arguments;
    ``noise`` is the already-scaled [N,S] sample added to sigma_raw.
    Returns (rgb_map, disp_map, acc_map, weights, depth_map)."""
    if clamp > 0:
        raw = raw.clamp(-clamp, clamp)
    dists = z_vals[..., 1:] - z_vals[..., :-1]
    dists = torch.cat([dists, torch.full_like(dists[..., :1], 1e10)], -1)
    dists = dists * rays_d[..., None, :].norm(dim=-1)
    rgb = torch.sigmoid(raw[..., :3])
    sig = raw[..., 3]
    if noise is not None:
        sig = sig + noise
    if add is not None:
        sig = sig + add
    alpha = 1.0 - torch.exp(-F.relu(sig) * dists)
    if mul is not None:
        alpha = alpha * F.relu(mul)
    T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha + 1e-10], -1), -1)[:, :-1]
    weights = alpha * T
    rgb_map = (weights[..., None] * rgb).sum(-2)
    depth_map = (weights * z_vals).sum(-1)
    acc_map = weights.sum(-1)
    disp_map = 1.0 / torch.max(1e-10 * torch.ones_like(depth_map), depth_map / acc_map)
    if white_bkgd:
        rgb_map = rgb_map + (1.0 - acc_map[..., None])
    return rgb_map, disp_map, acc_map, weights, depth_map


# ----------------------------------------------------------------------------- frame setup + render
def frame_setup(scene, n_samples=8, num_neighbor=4, n_pts=48, near=0.0, far=1.0, or_near=1.0, or_far=10.0):
    """Per-frame inputs of ``render_rays`` built as ``render_path`` does.
    run_S_eS_eN_alter_trt.py:245-302.  Returns rays [N,11], or_rays [N,11], mm_input
    [N,288], ref_nos, neighbour images [NB,3,Hf,Wf] and projection matrices [NB,3,4]."""
    H, W, K = scene['H'], scene['W'], _t(scene['K'])
    c2w, poses, images = _t(scene['c2w']), _t(scene['poses']), _t(scene['images'])
    rays_o, rays_d = get_rays(H, W, K, c2w)                                    # :245
    viewdirs = (rays_d / rays_d.norm(dim=-1, keepdim=True)).reshape(-1, 3)     # :246-248
    or_o = rays_o.reshape(-1, 3); or_d = rays_d.reshape(-1, 3)
    N = or_o.shape[0]
    or_rays = torch.cat([or_o, or_d, torch.full((N, 1), or_near), torch.full((N, 1), or_far), viewdirs], -1)
    o, d = ndc_rays(H, W, float(K[0, 0]), 1.0, rays_o, rays_d)                 # :265
    o = o.reshape(-1, 3); d = d.reshape(-1, 3)
    rays = torch.cat([o, d, torch.full((N, 1), near), torch.full((N, 1), far), viewdirs], -1)
    mm_input = mm_input_from_rays(o, d, n_pts)                                 # :274-277
    dist = ((c2w[None, :, 3] - poses[:, :, 3]) ** 2).sum(1) ** 0.5             # :281
    ref_nos = torch.sort(dist, dim=0, stable=True)[1][:num_neighbor]           # :282-283
    nb_img = images[ref_nos].permute(0, 3, 1, 2).contiguous()                  # :286,296
    flip = torch.diag(torch.tensor([1.0, -1.0, -1.0]))
    proj = K[None] @ (flip[None] @ poses[ref_nos])                             # :289-294
    return {'rays': rays.contiguous(), 'or_rays': or_rays.contiguous(), 'mm_input': mm_input.contiguous(),
            'ref_nos': ref_nos, 'images': nb_img, 'proj': proj.contiguous(), 'sh': (H, W, 3)}


def render_rays_infer(weights, rays, or_rays, images, proj, mm_input=None, n_samples=8, n_pts=48,
                      multires=10, multires_views=4, eps=1e-5, nerf='donerf'):
    """Inference ``render_rays``.  run_S_eS_eN_alter_trt.py:599-696.  Returns every
    intermediate so each HIP stage can be checked on its own."""
    S = n_samples
    N = rays.shape[0]
    o, d = rays[:, 0:3], rays[:, 3:6]
    viewdirs = rays[:, -3:]
    near, far = rays[:, 6:7], rays[:, 7:8]
    if mm_input is None:
        mm_input = mm_input_from_rays(o, d, n_pts)
    mm_rgb, add, mul, depth = sampler_forward(weights['sampler'], mm_input, S)          # :628
    depth_sorted, idx, add_s, mul_s = sort_gather(depth, add, mul, near, far)           # :631-635
    epi = project_trt(images, proj, or_rays[:, 0:3], or_rays[:, 3:6], depth_sorted, eps)  # :637-655
    epi_pts = o[:, None, :] + d[:, None, :] * depth_sorted[..., None]                   # :656
    pl = pluecker(epi_pts, d[:, None, :].expand(-1, S, -1)).reshape(N, 6 * S)           # :657-658
    refine_in = torch.cat([pl, epi], 1)                                                 # :661
    rdepth, _, offs = refine_forward(weights['refine'], refine_in, S)                   # :668
    z = interval_refine(depth_sorted, rdepth, near, far)                                # :673-677
    pts = o[:, None, :] + d[:, None, :] * z[..., None] + 1e-2 * offs.reshape(N, S, 3)   # :679-681
    emb = posenc(pts.reshape(-1, 3), multires)                                          # :198
    emb_d = posenc(viewdirs[:, None, :].expand(-1, S, -1).reshape(-1, 3), multires_views)  # :200-204
    if nerf == 'donerf':
        raw = nerf_forward(weights['nerf'], emb, emb_d).reshape(N, S, 4)                # :206
    else:
        raw = nerfcls_forward(weights['nerfcls'], torch.cat([emb, emb_d], -1)).reshape(N, S, 4)
    rgb, disp, acc, wts, dmap = raw2outputs(raw, z, d, add_s, mul_s)                    # :694
    return {'mm_rgb': mm_rgb, 'depth_raw': depth, 'depth_sorted': depth_sorted, 'sort_idx': idx,
            'add_sorted': add_s, 'mul_sorted': mul_s, 'epi': epi, 'refine_in': refine_in,
            'refine_depth': rdepth, 'offsets': offs, 'z': z, 'pts': pts, 'raw': raw,
            'rgb': rgb, 'disp': disp, 'acc': acc, 'weights': wts, 'depth': dmap}


# ----------------------------------------------------------------------------- stage-2 (training) forward
def project_train(images_nchw, poses, K, or_o, or_d, depth_ndc, ref_nos, eps=1e-5):
    """Training-variant neighbour projection with valid-mask mean fill.
    inverse_warp.py:515-581 (inverse_warp_rod1_rt2_coords) + run_S_eS_eN_alter_base_refine2.py:570-626.

    images_nchw [nv,3,Hf,Wf] (all training views); poses [nv,3,4] camera-to-world; K [3,3];
    ref_nos [N,nb] per-ray indices of the source views; depth_ndc [N,S] sorted sampler depths.
    c2 = R^T w - R^T t;  c2 /= (|c2.z| + 1e-8);  c2.z = 1;  c2.y = -c2.y;  p = K c2;  samples whose
    normalised X or Y leaves [-1,1] give 0;  valid = (sum_c rgb > 0);  invalid (k,s) entries are replaced
    by the mean over the valid neighbours of that sample.
    Returns (epi [N, nb*S*3] with index (k*S+s)*3+c, margin [N]): margin = the smallest distance of any of the
    ray's normalised coordinates to the in/out boundary |x| = 1 — the mask is discontinuous there, so
    comparisons at fp32 round-off are only meaningful for rays with margin > ~1e-5."""
    N, S = depth_ndc.shape
    nb = ref_nos.shape[1]
    _, _, Hf, Wf = images_nchw.shape
    poses, K = _t(poses), _t(K)
    z3d = 1.0 / (1.0 - depth_ndc - eps)                                         # refine2.py:570
    Rt = poses[:, :, :3].transpose(1, 2)                                        # inverse_warp.py:530-533
    tt = -torch.bmm(Rt, poses[:, :, 3:4])[:, :, 0]
    vals = torch.zeros(N, nb, S, 3)
    margin = torch.full((N,), float('inf'))
    for k in range(nb):
        v = ref_nos[:, k].long()
        for s in range(S):
            w = or_o + or_d * z3d[:, s:s + 1]                                   # inverse_warp.py:536
            c2 = torch.einsum('nij,nj->ni', Rt[v], w) + tt[v]                   # :539
            zz = c2[:, 2:3].abs()
            c2n = c2 / (zz + 1e-8)                                              # :543-544
            c2n = torch.stack([c2n[:, 0], -c2n[:, 1], torch.ones_like(c2n[:, 0])], -1)   # :545-546
            p = c2n @ K.T                                                       # :547
            X, Y = p[:, 0], p[:, 1]
            xn = 2 * X / (Wf - 1) - 1; yn = 2 * Y / (Hf - 1) - 1                # :552-553
            inside = (xn <= 1) & (xn >= -1) & (yn <= 1) & (yn >= -1)            # :558-562 (outside -> coordinate 2 -> zero)
            margin = torch.minimum(margin, torch.minimum((xn.abs() - 1).abs(), (yn.abs() - 1).abs()))
            out = torch.zeros(N, 3)
            for vi in v.unique().tolist():
                m = (v == vi) & inside
                if m.any():
                    out[m] = bilinear_zeros(images_nchw[vi], X[m], Y[m]).T
            vals[:, k, s, :] = out
    valid = (vals.sum(-1, keepdim=True) > 0).to(vals.dtype).expand(-1, -1, -1, 3)   # refine2.py:622
    mean = (valid * vals).sum(1, keepdim=True) / (valid.sum(1, keepdim=True) + 1e-6)    # :623
    vals = vals * valid + mean * (1 - valid)                                            # :624
    return vals.reshape(N, nb * S * 3), margin                                          # :626


def warp_train(img, depth, ro1, rd1, c2w2, K):
    """The training warp as an operator, inverse_warp.py:515-581 (inverse_warp_rod1_rt2_coords), padding_mode='zeros', scale=1.
    img [B,3,Hf,Wf]; depth [B,n]; ro1, rd1 [3,n] (the reference's repeat()-ed operands, stored once); c2w2 [B,3,4]; K [B,3,3].
    Returns (projected [B,3,n], margin [B,n]): margin = distance of the sample's normalised coordinates to the in/out boundary."""
    img, depth, ro1, rd1, c2w2, K = (_t(x) for x in (img, depth, ro1, rd1, c2w2, K))
    B, _, Hf, Wf = img.shape
    Rt = c2w2[:, :, :3].transpose(1, 2)                                         # :530-533
    tt = -torch.bmm(Rt, c2w2[:, :, 3:4])
    w = ro1[None] + rd1[None] * depth[:, None, :]                               # :536
    c2 = torch.bmm(Rt, w) + tt                                                  # :539
    c2n = c2 / (c2[:, 2:3].abs() + 1e-8)                                        # :543-544
    c2n = torch.stack([c2n[:, 0], -c2n[:, 1], torch.ones_like(c2n[:, 0])], 1)   # :545-546
    p = torch.bmm(K, c2n)                                                       # :547
    X, Y = p[:, 0], p[:, 1]
    xn = 2 * X / (Wf - 1) - 1; yn = 2 * Y / (Hf - 1) - 1                        # :552-553
    inside = (xn <= 1) & (xn >= -1) & (yn <= 1) & (yn >= -1)                    # :557-561
    out = torch.stack([bilinear_zeros(img[b], X[b], Y[b]) * inside[b][None] for b in range(B)], 0)
    return out, torch.minimum((xn.abs() - 1).abs(), (yn.abs() - 1).abs())


def select_neighbors_train(target_poses, poses, num_neighbor, order_idx=None):
    """Per-ray ranking of the training cameras by distance to the ray's own camera.
    refine2.py:590-600: randomize -> drop rank 0 (self) and take the rank positions ``order_idx`` (a sorted random
    subset drawn once per batch); evaluation (order_idx None) -> ranks 0..num_neighbor-1."""
    d = ((target_poses[:, None, :, 3] - poses[None, :, :, 3]) ** 2).sum(2) ** 0.5
    idx = torch.sort(d, dim=1, stable=True)[1]
    if order_idx is None:
        return idx[:, :num_neighbor]
    return idx[:, 1:][:, torch.as_tensor(order_idx).long()]


def render_rays_stage2(weights, rays, or_rays, images_nchw, poses, K, ref_nos, jitter=None, jitter_dir=1, raw_noise=None,
                       white_bkgd=False, n_samples=8, n_pts=48, eps=1e-5):
    """Stage-2 (refine) training-time ``render_rays`` forward with the random draws made explicit.
    run_S_eS_eN_alter_base_refine2.py:525-680.  weights: 'sampler', 'refine' (stacks) and 'nerfcls' (NeRF class).
    jitter [N,S] = min(|N(0,1)|/5, 1-2e-6) (refine2.py:649-653), jitter_dir +1 = toward the next sample / far,
    -1 = toward the previous / near (the coin flip, :654-661); raw_noise [N,S] = randn*raw_noise_std (:497)."""
    S = n_samples
    N = rays.shape[0]
    o, d = rays[:, 0:3], rays[:, 3:6]
    viewdirs = rays[:, -3:]
    near, far = rays[:, 6:7], rays[:, 7:8]
    mm_rgb, add, mul, depth = sampler_forward(weights['sampler'], mm_input_from_rays(o, d, n_pts), S)     # :551-563
    depth_sorted, idx, add_s, mul_s = sort_gather(depth, add, mul, near, far)                               # :563-568
    with torch.no_grad():                      # the reference builds the epipolar features under torch.no_grad() (:576-626)
        epi, margin = project_train(images_nchw, poses, K, or_rays[:, 0:3], or_rays[:, 3:6], depth_sorted, ref_nos, eps)   # :570-626
    pl = pluecker(o[:, None, :] + d[:, None, :] * depth_sorted[..., None], d[:, None, :].expand(-1, S, -1)).reshape(N, 6 * S)
    refine_in = torch.cat([pl, epi], 1)                                                                      # :634
    rdepth, refine_rgb, offs = refine_forward(weights['refine'], refine_in, S)                               # :635-638
    z = interval_refine(depth_sorted, rdepth, near, far)                                                     # :640-643
    if jitter is not None:                                                                                   # :646-662
        if jitter_dir > 0:
            diff = (z - torch.cat([z[:, 1:], far * torch.ones(N, 1)], 1)).abs()
            z = z + jitter * diff
        else:
            diff = (z - torch.cat([near * torch.ones(N, 1), z[:, :-1]], 1)).abs()
            z = z - jitter * diff
    pts = o[:, None, :] + d[:, None, :] * z[..., None] + 1e-2 * offs.reshape(N, S, 3)                        # :666-668
    emb = torch.cat([posenc(pts.reshape(-1, 3), 10), posenc(viewdirs[:, None, :].expand(-1, S, -1).reshape(-1, 3), 4)], -1)
    raw = nerfcls_forward(weights['nerfcls'], emb).reshape(N, S, 4)                                          # :669
    rgb, disp, acc, wts, dmap = raw2outputs(raw, z, d, add_s, mul_s, noise=raw_noise, white_bkgd=white_bkgd)  # :674
    return {'rgb_map0': refine_rgb, 'rgb_map1': rgb, 'depth_map': dmap, 'mm_rgb': mm_rgb, 'z_vals': z.mean(-1),
            'z_vals0': depth_sorted.mean(-1), 'depth_sorted': depth_sorted, 'sort_idx': idx, 'add_sorted': add_s, 'mul_sorted': mul_s,
            'epi': epi, 'edge_margin': margin, 'refine_in': refine_in, 'z': z, 'pts': pts, 'raw': raw, 'acc': acc, 'weights': wts}


def explore_samples(z, near, far, n_mult, dir1, jitter, dir2):
    """Stage-1 exploration of the refined depths (run_S_eS_eN_alter_base.py:689-729) with its draws made explicit:
    each of the 8 refined depths is replicated n_mult times toward the next (dir1 > 0) or previous (dir1 < 0) sample,
    offsets linspace(0, 1-1/n_mult, n_mult) * |gap|, sorted; then z += dir2 * jitter * |z - neighbour| with
    jitter [N, 8*n_mult] = min(|N(0,1)|/5, 0.99)."""
    N = z.shape[0]
    if n_mult > 1:
        mults = torch.linspace(0, 1 - 1 / n_mult, n_mult)[None]
        if dir1 > 0:
            diff = (z - torch.cat([z[:, 1:], far * torch.ones(N, 1)], 1)).abs()
            noise = mults[:, None, :] * diff[:, :, None]
        else:
            diff = (z - torch.cat([near * torch.ones(N, 1), z[:, :-1]], 1)).abs()
            noise = -mults[:, None, :] * diff[:, :, None]
        z = (z[:, :, None] + noise).reshape(N, -1)
        z, _ = torch.sort(z, dim=-1)
    if dir2 > 0:
        diff = (z - torch.cat([z[:, 1:], far * torch.ones(N, 1)], 1)).abs()
        return z + jitter * diff
    diff = (z - torch.cat([near * torch.ones(N, 1), z[:, :-1]], 1)).abs()
    return z - jitter * diff


def render_rays_stage1(weights, rays, or_rays, images_nchw, poses, K, ref_nos, train_sampler, n_mult=1, dir1=1, jitter=None, dir2=1,
                       raw_noise=None, white_bkgd=False, n_samples=8, n_pts=48):
    """Stage-1 training-time ``render_rays`` forward (run_S_eS_eN_alter_base.py:554-761), random draws explicit.
    train_sampler=True  (even steps): offsets added, compositing with the sampler's add/mul, no noise;
    train_sampler=False (odd steps, randomize): exploration path (n_mult, dir1, jitter, dir2), no offsets, compositing
    without add/mul and with sigma noise.  raw is clamped to +-10 (base.py:523); eps = 1e-6 (:607); epi is sample-major (:664-665)."""
    S = n_samples
    N = rays.shape[0]
    o, d = rays[:, 0:3], rays[:, 3:6]
    viewdirs = rays[:, -3:]
    near, far = rays[:, 6:7], rays[:, 7:8]
    import contextlib
    frozen = contextlib.nullcontext if train_sampler else torch.no_grad      # odd steps: sampler / refine nets under no_grad (base.py:592-596, 675-679)
    with frozen():
        mm_rgb, add, mul, depth = sampler_forward(weights['sampler'], mm_input_from_rays(o, d, n_pts), S)
    depth_sorted, idx, add_s, mul_s = sort_gather(depth, add, mul, near, far)
    with torch.no_grad():                                                    # epipolar features: always under no_grad (base.py:612-665)
        epi, margin = project_train(images_nchw, poses, K, or_rays[:, 0:3], or_rays[:, 3:6], depth_sorted, ref_nos, 1e-6)
        epi = epi.reshape(N, -1, S, 3).permute(0, 2, 1, 3).reshape(N, -1)                   # neighbour-major -> sample-major (:664-665)
    pl = pluecker(o[:, None, :] + d[:, None, :] * depth_sorted[..., None], d[:, None, :].expand(-1, S, -1)).reshape(N, 6 * S)
    refine_in = torch.cat([pl, epi], 1)
    with frozen():
        rdepth, refine_rgb, offs = refine_forward(weights['refine'], refine_in, S)
    z = interval_refine(depth_sorted, rdepth, near, far)
    if not train_sampler and jitter is not None:
        z = explore_samples(z, near, far, n_mult, dir1, jitter, dir2)
    Sx = z.shape[1]
    pts = o[:, None, :] + d[:, None, :] * z[..., None]
    if train_sampler:
        pts = pts + 1e-2 * offs.reshape(N, S, 3)
    emb = torch.cat([posenc(pts.reshape(-1, 3), 10), posenc(viewdirs[:, None, :].expand(-1, Sx, -1).reshape(-1, 3), 4)], -1)
    raw = nerfcls_forward(weights['nerfcls'], emb).reshape(N, Sx, 4)
    if train_sampler:
        rgb, disp, acc, wts, dmap = raw2outputs(raw, z, d, add_s, mul_s, clamp=10.0, white_bkgd=white_bkgd)
    else:
        rgb, disp, acc, wts, dmap = raw2outputs(raw, z, d, noise=raw_noise, clamp=10.0, white_bkgd=white_bkgd)
    return {'rgb_map0': refine_rgb, 'rgb_map1': rgb, 'depth_map': dmap, 'mm_rgb': mm_rgb, 'depth_map0': z.mean(-1), 'sigma1': raw[..., 3],
            'depth_sorted': depth_sorted, 'sort_idx': idx, 'add_sorted': add_s, 'mul_sorted': mul_s, 'epi': epi, 'edge_margin': margin,
            'refine_in': refine_in, 'z8': interval_refine(depth_sorted, rdepth, near, far), 'z': z, 'pts': pts, 'raw': raw}


def psnr(a, b, peak=1.0):
    mse = torch.mean((a.double() - b.double()) ** 2).item()
    return float('inf') if mse == 0 else 10.0 * math.log10(peak * peak / mse)


# ----------------------------------------------------------------------------- stage-2 training step
def trainer_layers(weights):
    """The 26 (W, b) pairs of the stage-2 optimizer in the trainer's order: sampler fc_backbone.0..5 + fc_output, refine net
    likewise, fine net (class NeRF) pts_linears.0..7, feature_linear, alpha_linear, views_linears.0, rgb_linear
    (run_S_eS_eN_alter_base_refine2.py:358-392).  ``weights``: 'sampler', 'refine' stacks and 'nerfcls'."""
    wc = weights['nerfcls']
    out = list(zip(weights['sampler']['W'], weights['sampler']['b'])) + list(zip(weights['refine']['W'], weights['refine']['b']))
    out += list(wc['pts_linears']) + [wc['feature_linear'], wc['alpha_linear'], wc['views_linears'][0], wc['rgb_linear']]
    return out


def weights_from_layers(layers):
    """Inverse of ``trainer_layers`` (entries may be tensors that require grad)."""
    L = list(layers)
    return {'sampler': {'W': [w for w, _ in L[0:7]], 'b': [b for _, b in L[0:7]]},
            'refine': {'W': [w for w, _ in L[7:14]], 'b': [b for _, b in L[7:14]]},
            'nerfcls': {'pts_linears': L[14:22], 'feature_linear': L[22], 'alpha_linear': L[23], 'views_linears': [L[24]], 'rgb_linear': L[25]}}


def stage2_loss(layers, rays, or_rays, target, images_nchw, poses, K, ref_nos, jitter=None, jitter_dir=1, raw_noise=None, white_bkgd=False,
                a_mmrgb=0.0):
    """loss of one stage-2 iteration (run_S_eS_eN_alter_base_refine2.py:855-866): img2mse(rgb_map1, target)
    [+ a_mmrgb (img2mse(rgb_map0) + img2mse(mm_rgb))].  Returns (loss, img_loss, outputs)."""
    o = render_rays_stage2(weights_from_layers(layers), rays, or_rays, images_nchw, poses, K, ref_nos, jitter=jitter, jitter_dir=jitter_dir,
                           raw_noise=raw_noise, white_bkgd=white_bkgd)
    img_loss = torch.mean((o['rgb_map1'] - target) ** 2)
    loss = img_loss
    if a_mmrgb > 0:
        loss = loss + a_mmrgb * (torch.mean((o['rgb_map0'] - target) ** 2) + torch.mean((o['mm_rgb'] - target) ** 2))
    return loss, img_loss, o


def stage1_loss(layers, rays, or_rays, target, images_nchw, poses, K, ref_nos, train_sampler, n_mult=1, dir1=1, jitter=None, dir2=1, raw_noise=None,
                white_bkgd=False):
    """loss of one stage-1 iteration (run_S_eS_eN_alter_base.py:929-958): odd (train_sampler False) img2mse(rgb_map1) on the
    explored samples; even img2mse(rgb_map1) + img2mse(rgb_map0) + img2mse(mm_rgb).  Returns (loss, img_loss, outputs)."""
    o = render_rays_stage1(weights_from_layers(layers), rays, or_rays, images_nchw, poses, K, ref_nos, train_sampler, n_mult=n_mult, dir1=dir1,
                           jitter=jitter, dir2=dir2, raw_noise=raw_noise, white_bkgd=white_bkgd)
    img_loss = torch.mean((o['rgb_map1'] - target) ** 2)
    loss = img_loss
    if train_sampler:
        loss = loss + torch.mean((o['rgb_map0'] - target) ** 2) + torch.mean((o['mm_rgb'] - target) ** 2)
    return loss, img_loss, o
