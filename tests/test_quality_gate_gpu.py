"""GPU: the literal quality gate of BASELINE.json — "PSNR within 0.05 dB of reference".

The reference computes its PSNR against the GROUND-TRUTH image of every hold-out view (run_S_eS_eN_alter_trt.py:351-353, 368-373:
``mse2psnr(img2mse(rgb, gt_imgs[i]))``, then the mean over the views, :719-721).  The other full-frame tests bound the error against the
oracle's rendering (>= 46.4 dB, which moves a 27 dB image PSNR by <= 0.05 dB *if* the rendering error is uncorrelated with the image error);
this file drops the "if": scenes with ground truth, nets an optimizer fitted to them, both renderers measured against the same ground truth.

Two scenes (tests/llff_synth.py, seed 2, 20 views of 189 x 252), each with the nets this package's stage-1 + stage-2 drivers trained on it
(tools/make_trained_fixture.py), rebuilt deterministically here and read by the LLFF loader:

  scene3d   (round 6) ONE 3-D scene ray-cast from the rig (``Scene3D``: wall, receding floor, two layers of occluding discs, fine texture),
            COLMAP points = real scene points with the tracks of the views that see them.  The nets generalise: the HOLD-OUT views (every 8th,
            the reference's protocol) come out at 33 .. 37 dB — the reference's own quality number is meaningful here, and it is the gate.
  pictures  (round 5) twenty independent pictures on the rig: nothing to generalise to (hold-out ~10 dB); its four TRAINING views at 35 .. 37 dB
            remain a sensitive memorisation case.

Gates, per view and on the mean over the views:
  * both renderers >= 20 dB on scene3d's hold-out views (the fit is a real one);
  * 'default' preset (what bench.py's headline times): | PSNR(HIP vs GT) - PSNR(oracle vs GT) | <= 0.05 dB;
  * 'quality' preset (split-fp16 sampler for every ray + fp16 NeRF operands, ``pronerf_amd.render.PRESETS``): <= 0.025 dB — 2x headroom at the
    sensitive views, the documented choice above ~35 dB image PSNR (DESIGN.md §2 states the ceiling of the default preset)."""
import os

import numpy as np
import pytest
import torch

from oracle import pronerf_oracle as orc
from oracle import synth

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')
SCENES = {'scene3d': ({'consistent': True, 'n_points': 3000}, 'trained_scene3d.npz'),       # = tools/make_trained_fixture.py SCENE_KW
          'pictures': ({}, 'trained_synth_scene.npz')}


def _gate_rows(tmp_path, which, presets):
    import llff_synth
    from pronerf_amd import load_llff as L
    from pronerf_amd.render import Renderer
    assert torch.cuda.is_available()
    dev = torch.device('cuda:0')
    torch.backends.cuda.matmul.allow_tf32 = False
    kw, fixture = SCENES[which]
    root = llff_synth.make_dataset(str(tmp_path / 'scene'), seed=2, n=20, H=189, W=252, factor=4, **kw)
    images, poses, bds, _, i_test, i_ref = L.load_llff_data_infer(root, factor=4, llffhold=8)
    H, W, focal = int(poses[0, 0, 4]), int(poses[0, 1, 4]), float(poses[0, 2, 4])
    assert (H, W) == (189, 252) and list(i_test) == [0, 8, 16]
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]], dtype=np.float32)             # trt.py:746-751
    w = synth.load_trained_fixture(os.path.join(GOLDEN, fixture))
    rends = {p: Renderer({k: w[k] for k in ('sampler', 'refine', 'nerf')}, max_rays=H * W, device=dev, preset=p) for p in presets}
    td = lambda x: torch.as_tensor(x).to(dev)
    wd = {k: {'W': [td(x) for x in w[k]['W']], 'b': [td(x) for x in w[k]['b']]} for k in ('sampler', 'refine')}
    c = w['nerfcls']
    pair = lambda p: (td(p[0]), td(p[1]))
    wd['nerfcls'] = {'pts_linears': [pair(p) for p in c['pts_linears']], 'feature_linear': pair(c['feature_linear']), 'alpha_linear': pair(c['alpha_linear']),
                     'views_linears': [pair(c['views_linears'][0])], 'rgb_linear': pair(c['rgb_linear'])}
    i_train = [int(i) for i in i_ref[[0, len(i_ref) // 3, 2 * len(i_ref) // 3, len(i_ref) - 1]]]
    rows = []
    for kind, views in (('hold-out', [int(i) for i in i_test]), ('training', i_train)):
        for v in views:
            scene = {'H': H, 'W': W, 'K': K, 'c2w': poses[v, :3, :4], 'poses': poses[i_ref][:, :3, :4], 'images': images[i_ref]}   # trt.py:773-787
            fr = orc.frame_setup(scene)
            gt = torch.as_tensor(images[v], dtype=torch.float32).reshape(-1, 3).to(dev)
            ref = None
            for p, rend in rends.items():
                rend.set_views(scene['c2w'], scene['poses'], scene['images'], K)
                rays, or_rays = rend.frame_rays(K, scene['c2w'], H, W)
                assert torch.equal(rays.cpu(), fr['rays']) and torch.equal(or_rays.cpu(), fr['or_rays'])
                rgbd, _ = rend.render_rays(rays, or_rays)
                if ref is None:
                    with torch.no_grad():
                        ref = orc.render_rays_infer(wd, rays, or_rays, fr['images'].to(dev), fr['proj'].to(dev), mm_input=fr['mm_input'].to(dev), nerf='nerfcls')
                    p_orc = orc.psnr(ref['rgb'], gt)                                                   # mse2psnr(img2mse(rgb, gt)), trt.py:351-353
                rows.append((p, kind, v, orc.psnr(rgbd[:, :3], gt), p_orc, orc.psnr(rgbd[:, :3], ref['rgb']), rend.ctx.sampler_stats() / (H * W)))
    print()
    for p, kind, v, p_hip, p_orc, p_err, f2 in rows:
        print(f'[quality gate {which}] {p:8s} {kind:8s} view {v:2d}: PSNR vs ground truth  HIP {p_hip:7.3f} dB   oracle {p_orc:7.3f} dB   difference {p_hip - p_orc:+.4f} dB'
              f'   (HIP vs oracle {p_err:.1f} dB; second pass {100 * f2:.1f} % of the rays)')
    return rows


def _check(rows, which, bars):
    for p, bar in bars.items():
        for kind in ('hold-out', 'training'):
            sel = [r for r in rows if r[0] == p and r[1] == kind]
            m_hip, m_orc = np.mean([r[3] for r in sel]), np.mean([r[4] for r in sel])
            print(f'[quality gate {which}] {p}: mean over the {len(sel)} {kind} views: HIP {m_hip:.3f} dB, oracle {m_orc:.3f} dB, difference {m_hip - m_orc:+.4f} dB (bar {bar})')
            assert abs(m_hip - m_orc) <= bar, (p, kind, m_hip, m_orc)
            for r in sel:
                assert abs(r[3] - r[4]) <= bar, r


def test_holdout_psnr_of_a_real_scene_within_0p05_db_of_the_oracle(tmp_path):
    """The reference's own protocol on the geometrically consistent scene: hold-out views, PSNR against their ground-truth pictures."""
    rows = _gate_rows(tmp_path, 'scene3d', ('default', 'quality'))
    hold = [r for r in rows if r[1] == 'hold-out']
    assert len(hold) == 6 and min(min(r[3], r[4]) for r in hold) >= 20.0, 'the scene3d fixture no longer generalises to its hold-out views'
    _check(rows, 'scene3d', {'default': 0.05, 'quality': 0.025})


def test_psnr_against_ground_truth_within_0p05_db_of_the_oracle(tmp_path):
    """Round 5's gate: the independent-pictures scene; its training views (35 .. 37 dB) are the sensitive half."""
    rows = _gate_rows(tmp_path, 'pictures', ('default', 'quality'))
    _check(rows, 'pictures', {'default': 0.05, 'quality': 0.025})
    assert max(r[4] for r in rows if r[1] == 'training') > 28.0, 'the fixture no longer fits its own training views: the sensitive half of the gate is gone'
