"""Deterministic synthetic weights / scenes (no dataset or checkpoint ships with the repo).

Shared by ``bench.py``, ``__graft_entry__.smoke()``, the tests and the golden generator
(``oracle/gen_golden.py``, via ``oracle/synth.py``) so that weights never have to be committed as
fixtures: everything is regenerated from a seed with ``numpy.random.RandomState`` (MT19937, stable
across numpy versions).  Pure numpy — no compute path lives here.  Shapes follow the reference's ``create_nerf``
(run_S_eS_eN_alter_trt.py:412-458) with the fern_trt.txt hyper-parameters
(configs/llff/fern/fern_trt.txt:14-34).
"""
from __future__ import annotations

import numpy as np

# fern_trt.txt / script defaults (run_S_eS_eN_alter_trt.py:62-65,132-135)
N_SAMPLES = 8
NUM_NEIGHBOR = 4
N_POINT_RAY_ENC = 48
MULTIRES = 10
MULTIRES_VIEWS = 4
NETDEPTH = 8
NETWIDTH = 256
MMNETDEPTH = 6
MMNETWIDTH = 256

SAMPLER_DIMS = [6 * N_POINT_RAY_ENC] + [MMNETWIDTH] * MMNETDEPTH + [3 * N_SAMPLES + 3]
REFINE_DIMS = [6 * N_SAMPLES + 3 * NUM_NEIGHBOR * N_SAMPLES] + [MMNETWIDTH] * MMNETDEPTH + [4 * N_SAMPLES + 3]
POS_CH = 3 + 6 * MULTIRES          # 63
DIR_CH = 3 + 6 * MULTIRES_VIEWS    # 27


def nerf_layer_dims(netdepth=NETDEPTH):
    """(in,out) of DoNeRFTRT(D=netdepth,W=256,skip='auto') (run_nerf_helpers.py:1190-1239)."""
    dims = [(POS_CH, NETWIDTH)] + [(NETWIDTH, NETWIDTH)] * (netdepth - 2) + [(NETWIDTH + DIR_CH, 4)]
    return dims


def nerfcls_layer_dims():
    """Layer table of the ``NeRF`` class (run_nerf_helpers.py:792-822), skips=[4]."""
    W = NETWIDTH
    pts = [(POS_CH, W)] + [((W + POS_CH) if i == 4 else W, W) for i in range(NETDEPTH - 1)]
    return {
        'pts_linears': pts,
        'views_linears': [(DIR_CH + W, W // 2)],
        'feature_linear': (W, W),
        'alpha_linear': (W, 1),
        'rgb_linear': (W // 2, 3),
    }


def _uniform(rs, shape, bound):
    return rs.uniform(-bound, bound, size=shape).astype(np.float32)


def _mlp(rs, dims, kind):
    Ws, bs = [], []
    for i in range(len(dims) - 1):
        fi, fo = dims[i], dims[i + 1]
        if kind == 'default':
            wb = 1.0 / np.sqrt(fi)          # nn.Linear default (kaiming_uniform, a=sqrt(5))
        else:
            wb = np.sqrt(6.0 / fi)          # signal-preserving: every layer matters to the output
        Ws.append(_uniform(rs, (fo, fi), wb))
        bs.append(_uniform(rs, (fo,), 1.0 / np.sqrt(fi)))
    return Ws, bs


def make_weights(seed: int = 0, kind: str = 'trained', n_pts: int = N_POINT_RAY_ENC, mmnetdepth: int = MMNETDEPTH, num_neighbor: int = NUM_NEIGHBOR,
                 netdepth: int = NETDEPTH):
    """Three weight sets as lists of numpy arrays (torch layout ``W[out,in]``).  n_pts / mmnetdepth / num_neighbor / netdepth: the reference's free
    shape arguments (``--N_point_ray_enc --mmnetdepth --num_neighbor --netdepth``, run_S_eS_eN_alter_trt.py:62-82, 427-457); defaults = fern_trt.txt.

    kind:
      'default'  module default initialisation (sampler/refine: nn.Linear default;
                 nerf: kaiming-normal weights, run_nerf_helpers.py:1243-1244).  Sampler
                 depths collapse into a narrow band -> worst case for sort ties.
      'spread'   'default' with the sampler head scaled x8 and a spread depth bias
                 (tie-free sort-index test).
      'trained'  signal-preserving hidden layers and head biases chosen so that depths
                 spread over (0,1), alpha/weights are non-degenerate and rgb spans [0,1].
      'heavy'    'trained' with Student-t(3) heavy-tailed sampler hidden weights (same variance);
      'x4'       'trained' with the sampler's hidden layers scaled x4 each (activations up to the fp16
                 range) and its output layer scaled back: adversarial sets for the two-pass sampler.
    """
    assert kind in ('default', 'spread', 'trained', 'heavy', 'x4')
    adversarial, kind = kind, ('trained' if kind in ('heavy', 'x4') else kind)
    rs = np.random.RandomState(1000003 * (seed + 1) + {'default': 0, 'spread': 1, 'trained': 2}[kind])
    S = N_SAMPLES
    sampler_dims = [6 * n_pts] + [MMNETWIDTH] * mmnetdepth + [3 * N_SAMPLES + 3]
    refine_dims = [6 * N_SAMPLES + 3 * num_neighbor * N_SAMPLES] + [MMNETWIDTH] * mmnetdepth + [4 * N_SAMPLES + 3]
    sW, sb = _mlp(rs, sampler_dims, 'default' if kind != 'trained' else 'trained')
    rW, rb = _mlp(rs, refine_dims, 'default' if kind != 'trained' else 'trained')
    if adversarial == 'heavy':
        # the sampler's hidden weights redrawn from Student's t with 3 degrees of freedom at the variance of the 'trained' draw: a few
        # weights per row 5-20 x the rest (what the column-norm constants of the two-pass sampler's error model have to cover)
        ts = np.random.RandomState(7000003 * (seed + 1))
        for i in range(len(sW) - 1):
            fi = sampler_dims[i]
            sW[i] = (ts.standard_t(3, size=sW[i].shape) * np.sqrt(2.0 / fi / 3.0)).astype(np.float32)
    if adversarial == 'x4':
        # the sampler's hidden layers x4 each (weights and biases), its output layer / 4^6: activations up to the fp16 range (the saturated
        # rays go through the exact-fp32 pass), logits of the usual size
        for i in range(len(sW) - 1):
            sW[i] = sW[i] * 4.0; sb[i] = sb[i] * 4.0
        sW[-1] = sW[-1] / 4.0 ** (len(sW) - 1)
    nW, nb = [], []
    for fi, fo in nerf_layer_dims(netdepth):
        nW.append((rs.randn(fo, fi) * np.sqrt(2.0 / fi)).astype(np.float32))
        nb.append(_uniform(rs, (fo,), 1.0 / np.sqrt(fi)))
    perm = np.array([5, 2, 7, 0, 3, 6, 1, 4])
    if kind == 'spread':
        sW[-1] = sW[-1].copy(); sb[-1] = sb[-1].copy()
        sW[-1][:S] *= 8.0
        sb[-1][:S] = np.linspace(-2.5, 2.5, S, dtype=np.float32)[perm]
    if kind == 'trained':
        sW[-1] = sW[-1].copy(); sb[-1] = sb[-1].copy()
        sW[-1] *= 0.35
        sb[-1][:S] = np.linspace(-2.0, 2.0, S, dtype=np.float32)[perm]
        sb[-1][S:2 * S] += 2.0           # density add
        sb[-1][2 * S:3 * S] += 1.0       # density mul
        rW[-1] = rW[-1] * 0.35
        nW[-1] = nW[-1] * 0.5
        nb[-1] = nb[-1].copy(); nb[-1][3] += 1.0
    return {
        'sampler': {'W': sW, 'b': sb},
        'refine': {'W': rW, 'b': rb},
        'nerf': {'W': nW, 'b': nb},
    }


def make_nerfcls_weights(seed: int = 0, head_scale: float = 1.0):
    """Weights of the ``NeRF`` class fine net (default nn.Linear-style init, gain 2).  ``head_scale`` < 1
    shrinks the rgb / alpha heads to trained-like magnitudes (logits of a few units)."""
    rs = np.random.RandomState(7919 * (seed + 1))
    t = nerfcls_layer_dims()
    out = {}

    def lin(fi, fo):
        return _uniform(rs, (fo, fi), np.sqrt(6.0 / fi)), _uniform(rs, (fo,), 1.0 / np.sqrt(fi))

    out['pts_linears'] = [lin(fi, fo) for fi, fo in t['pts_linears']]
    out['views_linears'] = [lin(*t['views_linears'][0])]
    out['feature_linear'] = lin(*t['feature_linear'])
    out['alpha_linear'] = lin(*t['alpha_linear'])
    out['rgb_linear'] = lin(*t['rgb_linear'])
    if head_scale != 1.0:
        for k in ('alpha_linear', 'rgb_linear'):
            W, b = out[k]
            out[k] = ((W * head_scale).astype(np.float32), b)
    return out


def state_dicts(weights):
    """numpy weights -> the reference modules' ``state_dict`` key layout.

    sampler/refine: ``fc_backbone.{i}.*``, ``fc_output.*`` (run_nerf_helpers.py:1484-1488);
    DoNeRFTRT: ``layers.{i}.*`` (run_nerf_helpers.py:1235-1239).
    """
    import torch
    out = {}
    for name in ('sampler', 'refine'):
        W, b = weights[name]['W'], weights[name]['b']
        sd = {}
        for i in range(len(W) - 1):
            sd[f'fc_backbone.{i}.weight'] = torch.from_numpy(W[i].copy())
            sd[f'fc_backbone.{i}.bias'] = torch.from_numpy(b[i].copy())
        sd['fc_output.weight'] = torch.from_numpy(W[-1].copy())
        sd['fc_output.bias'] = torch.from_numpy(b[-1].copy())
        out[name] = sd
    W, b = weights['nerf']['W'], weights['nerf']['b']
    sd = {}
    for i in range(len(W)):
        sd[f'layers.{i}.weight'] = torch.from_numpy(W[i].copy())
        sd[f'layers.{i}.bias'] = torch.from_numpy(b[i].copy())
    out['nerf'] = sd
    return out


def nerfcls_state_dict(w):
    import torch
    sd = {}
    for i, (W, b) in enumerate(w['pts_linears']):
        sd[f'pts_linears.{i}.weight'] = torch.from_numpy(W.copy()); sd[f'pts_linears.{i}.bias'] = torch.from_numpy(b.copy())
    W, b = w['views_linears'][0]
    sd['views_linears.0.weight'] = torch.from_numpy(W.copy()); sd['views_linears.0.bias'] = torch.from_numpy(b.copy())
    for k in ('feature_linear', 'alpha_linear', 'rgb_linear'):
        W, b = w[k]
        sd[f'{k}.weight'] = torch.from_numpy(W.copy()); sd[f'{k}.bias'] = torch.from_numpy(b.copy())
    return sd


def make_scene(seed: int = 0, H: int = 24, W: int = 32, Hf: int | None = None, Wf: int | None = None,
               focal: float | None = None, n_views: int = NUM_NEIGHBOR, sigma_t: float = 0.05,
               rotate: bool = False):
    """A synthetic forward-facing scene (SURVEY.md §8(d) / BASELINE.md §4).

    Returns a dict with ``H, W, focal, K[3,3], c2w[3,4]`` (target), ``poses[n,3,4]``
    (neighbour cameras) and ``images[n,Hf,Wf,3]`` in [0,1).  The Fern geometry is
    H=756, W=1008, focal=815.13.
    """
    rs = np.random.RandomState(424243 * (seed + 1))
    Hf = H if Hf is None else Hf
    Wf = W if Wf is None else Wf
    if focal is None:
        focal = 815.13 * W / 1008.0
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]], dtype=np.float32)

    def pose(sig):
        R = np.eye(3, dtype=np.float64)
        if rotate:
            a = rs.randn(3) * 0.03
            cx, sx, cy, sy, cz, sz = np.cos(a[0]), np.sin(a[0]), np.cos(a[1]), np.sin(a[1]), np.cos(a[2]), np.sin(a[2])
            Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
            Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
            Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
            R = Rz @ Ry @ Rx
        t = rs.randn(3) * sig
        return np.concatenate([R, t[:, None]], 1).astype(np.float32)

    c2w = pose(sigma_t)
    poses = np.stack([pose(sigma_t) for _ in range(n_views)], 0)
    images = rs.rand(n_views, Hf, Wf, 3).astype(np.float32)
    return {'H': H, 'W': W, 'focal': float(focal), 'K': K, 'c2w': c2w, 'poses': poses, 'images': images}


# optimizer-trained nets in the tree (tools/make_trained_fixture.py): 'pictures' = fitted to twenty independent pictures on a rig (round 4);
# 'scene3d' = fitted to ONE ray-cast 3-D scene (round 6, tests/llff_synth.py Scene3D): hold-out views 33.8 .. 37.3 dB, a sampler that has learned surfaces
FIXTURES = {'pictures': 'trained_synth_scene.npz', 'scene3d': 'trained_scene3d.npz'}


def scene3d_frame(view=0, scale=4):
    """The consistent scene of the 'scene3d' fixture as a frame dict like ``make_scene``'s: hold-out pose ``view`` of its LLFF directory (rebuilt in a
    temporary directory, once per process) rendered at ``scale`` x the training resolution (4: the Fern frame, 756 x 1008 — NDC rays do not depend on
    the pixel grid), neighbour cameras / images = the training views in the loader's greedy COLMAP order (189 x 252 pictures)."""
    import os
    import sys
    import tempfile
    key = (view, scale)
    if key not in _SCENE3D:
        tests = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests')
        if tests not in sys.path:
            sys.path.insert(0, tests)
        import llff_synth
        from . import load_llff as L
        if 'root' not in _SCENE3D:
            _SCENE3D['tmp'] = tempfile.TemporaryDirectory()
            _SCENE3D['root'] = llff_synth.make_dataset(os.path.join(_SCENE3D['tmp'].name, 'scene'), seed=2, n=20, H=189, W=252, factor=4, consistent=True, n_points=3000)
        images, poses, bds, _, i_test, i_ref = L.load_llff_data_infer(_SCENE3D['root'], factor=4, llffhold=8)
        H, W, focal = int(poses[0, 0, 4]) * scale, int(poses[0, 1, 4]) * scale, float(poses[0, 2, 4]) * scale
        K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]], dtype=np.float32)
        _SCENE3D[key] = {'H': H, 'W': W, 'focal': focal, 'K': K, 'c2w': poses[view, :3, :4].astype(np.float32), 'poses': poses[i_ref][:, :3, :4].astype(np.float32),
                         'images': images[i_ref].astype(np.float32), 'gt_small': images[view].astype(np.float32)}
    return _SCENE3D[key]


_SCENE3D = {}


def load_trained_fixture(path=None):
    """The optimizer-trained nets of tests/golden/trained_synth_scene.npz (tools/make_trained_fixture.py: this package's stage-1 and stage-2
    drivers on the synthetic LLFF scene) as weight dicts: 'sampler', 'refine' (stacks of 7), 'nerf' = the NeRF-class fine net in pack order
    (pts_linears 0..7, feature, alpha, views, rgb: what ``Renderer`` takes) and 'nerfcls' = the same by name (what the oracle takes)."""
    import os
    if path is None or path in FIXTURES:
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', FIXTURES[path or 'pictures'])
    g = np.load(path)
    L = [(g[f'W{i}'], g[f'b{i}']) for i in range(26)]
    stack = lambda a, b: {'W': [W for W, _ in L[a:b]], 'b': [x for _, x in L[a:b]]}
    return {'sampler': stack(0, 7), 'refine': stack(7, 14), 'nerf': stack(14, 26),
            'nerfcls': {'pts_linears': L[14:22], 'feature_linear': L[22], 'alpha_linear': L[23], 'views_linears': [L[24]], 'rgb_linear': L[25]},
            'info': {k: g[k] for k in ('stage1_iters', 'stage2_iters', 'stage1_loss', 'stage2_loss')}}


def scene_for(seed: int, kind: str, **kw):
    """The frame a weight set is tested on: ``make_scene(seed, **kw)``, or — kind 'scene' — the scene those nets were trained on (hold-out view 0 at the Fern frame size)."""
    return scene3d_frame(0, 4) if kind == 'scene' else make_scene(seed, **kw)


def weight_set(seed: int, kind: str):
    """``make_weights(seed, kind)``, or the optimizer-trained fixtures: kind 'optimizer' (independent pictures) / 'scene' (the consistent 3-D scene); the seed
    then only names the frame."""
    if kind == 'scene':
        return load_trained_fixture('scene3d')
    return load_trained_fixture() if kind == 'optimizer' else make_weights(seed, kind)
