"""GPU: the literal quality gate of BASELINE.json — "PSNR within 0.05 dB of reference".

The reference computes its PSNR against the GROUND-TRUTH image of every hold-out view (run_S_eS_eN_alter_trt.py:351-353, 368-373:
``mse2psnr(img2mse(rgb, gt_imgs[i]))``, then the mean over the views).  The other full-frame tests bound the error against the oracle's
rendering (>= 46.4 dB, which moves a 27 dB image PSNR by <= 0.05 dB *if* the rendering error is uncorrelated with the image error); this test
drops the "if": a scene with ground truth, nets an optimizer fitted to it, both renderers measured against the same ground truth.

Scene and nets: tests/llff_synth.py (seed 2, 20 views 189 x 252) is the LLFF directory tools/make_trained_fixture.py trained
tests/golden/trained_synth_scene.npz on (stage-1 + stage-2 drivers of this package, 33.8 dB on its training rays).  Train-free here: the
directory is rebuilt (deterministic), read by the loader, and rendered
  * at the three hold-out poses (every 8th view, the reference's protocol; the synthetic views are independent pictures, so the fit does not
    generalise to them — PSNR ~ 10 dB, where a rendering error hardly moves the figure), and
  * at four TRAINING poses, where the nets fit the ground truth to 30 dB and more — the sensitive case: there the rendering error is a
    visible share of the image error.
Gate per view: | PSNR(HIP path vs GT) - PSNR(oracle vs GT) | <= 0.05 dB, and the same for the mean over the views."""
import os

import numpy as np
import pytest
import torch

from oracle import pronerf_oracle as orc
from oracle import synth

pytestmark = pytest.mark.gpu


def test_psnr_against_ground_truth_within_0p05_db_of_the_oracle(tmp_path):
    import llff_synth
    from pronerf_amd import load_llff as L
    from pronerf_amd.render import Renderer
    assert torch.cuda.is_available()
    dev = torch.device('cuda:0')
    torch.backends.cuda.matmul.allow_tf32 = False
    root = llff_synth.make_dataset(str(tmp_path / 'scene'), seed=2, n=20, H=189, W=252, factor=4)        # = tools/make_trained_fixture.py's scene
    images, poses, bds, _, i_test, i_ref = L.load_llff_data_infer(root, factor=4, llffhold=8)
    H, W, focal = int(poses[0, 0, 4]), int(poses[0, 1, 4]), float(poses[0, 2, 4])
    assert (H, W) == (189, 252) and list(i_test) == [0, 8, 16]
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]], dtype=np.float32)             # trt.py:746-751
    w = synth.load_trained_fixture()
    rend = Renderer({k: w[k] for k in ('sampler', 'refine', 'nerf')}, max_rays=H * W, device=dev)
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
            rend.set_views(scene['c2w'], scene['poses'], scene['images'], K)
            fr = orc.frame_setup(scene)
            rays, or_rays = rend.frame_rays(K, scene['c2w'], H, W)
            assert torch.equal(rays.cpu(), fr['rays']) and torch.equal(or_rays.cpu(), fr['or_rays'])
            rgbd, _ = rend.render_rays(rays, or_rays)
            with torch.no_grad():
                ref = orc.render_rays_infer(wd, rays, or_rays, fr['images'].to(dev), fr['proj'].to(dev), mm_input=fr['mm_input'].to(dev), nerf='nerfcls')
            gt = torch.as_tensor(images[v], dtype=torch.float32).reshape(-1, 3).to(dev)
            p_hip, p_orc = orc.psnr(rgbd[:, :3], gt), orc.psnr(ref['rgb'], gt)                          # mse2psnr(img2mse(rgb, gt)), trt.py:351-353
            p_err = orc.psnr(rgbd[:, :3], ref['rgb'])
            rows.append((kind, v, p_hip, p_orc, p_err))
    print()
    for kind, v, p_hip, p_orc, p_err in rows:
        print(f'[quality gate] {kind:8s} view {v:2d}: PSNR vs ground truth  HIP {p_hip:7.3f} dB   oracle {p_orc:7.3f} dB   difference {p_hip - p_orc:+.4f} dB'
              f'   (HIP vs oracle {p_err:.1f} dB)')
    for kind in ('hold-out', 'training'):
        sel = [r for r in rows if r[0] == kind]
        m_hip, m_orc = np.mean([r[2] for r in sel]), np.mean([r[3] for r in sel])
        print(f'[quality gate] mean over the {len(sel)} {kind} views: HIP {m_hip:.3f} dB, oracle {m_orc:.3f} dB, difference {m_hip - m_orc:+.4f} dB')
        assert abs(m_hip - m_orc) <= 0.05, (kind, m_hip, m_orc)
    for kind, v, p_hip, p_orc, p_err in rows:
        assert abs(p_hip - p_orc) <= 0.05, (kind, v, p_hip, p_orc)
    assert max(r[3] for r in rows if r[0] == 'training') > 28.0, 'the fixture no longer fits its own training views: the sensitive half of the gate is gone'
