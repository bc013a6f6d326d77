"""CPU: the geometrically consistent fixture scene (tests/llff_synth.py ``Scene3D``, ``make_dataset(.., consistent=True)``) IS a scene.

The nets of tests/golden/trained_scene3d.npz were trained on it and the hold-out quality gate, the kappa margin and the bench leg are anchored on it
(tests/test_quality_gate_gpu.py, tests/test_fullframe_gpu.py, bench.py ``weights_scene3d``); VERDICT r5 asked for exactly this: textured surfaces at different
depths ray-cast from the forward-facing rig, COLMAP points that are real scene points seen by the views that see them, so that the loader's visibility
ranking (reference load_llff.py:496-547) selects real neighbours.  Checked here, on a small instance, without a GPU:

  * every view is a picture of the same geometry: a pixel of view A, lifted to its ray-cast surface point and projected into view B, lands on the same
    colour wherever B sees that point (B's own ray to it hits it first) — and the scene HAS parallax and occlusion (some points are hidden in B);
  * the pictures the loader reads are the ray-cast renders (8-bit), the depth bounds are the 0.1 / 99.9 percentiles of each view's visible point depths;
  * the COLMAP model: every track member really sees its point (projects inside the image, first hit = the point), every non-member does not; the
    2-D observations are the projections; names / ids / file order are permuted as in a real model and survive the binary round trip;
  * ``load_llff_data_infer`` on it: hold-out = every 8th view, and the greedy reference ranking equals a brute-force greedy set cover of the TRUE
    visibility matrix; ``pronerf_amd.synthetic.scene3d_frame`` hands that scene to the GPU tests at the Fern frame size (rays of the small picture = every
    4th pixel)."""
import os

import numpy as np
import pytest

import llff_synth
from pronerf_amd import colmap_utils as cu
from pronerf_amd import load_llff as L

N, H, W, F, NP = 9, 24, 32, 4, 400


@pytest.fixture(scope='module')
def ds(tmp_path_factory):
    root = llff_synth.make_dataset(str(tmp_path_factory.mktemp('scene3d') / 'scene'), seed=5, n=N, H=H, W=W, factor=F, n_points=NP, consistent=True)
    arr = np.load(os.path.join(root, 'poses_bounds.npy'))
    cams = [arr[i, :15].reshape(3, 5) for i in range(N)]
    return root, arr, cams, llff_synth.Scene3D(5)


def test_views_are_pictures_of_one_geometry(ds):
    root, arr, cams, scene = ds
    f = llff_synth.FOCAL_PER_WIDTH * W
    seen_frac, occluded = [], 0
    for a, b in ((0, 1), (2, 5), (7, 3), (4, 8)):
        rgb_a, _ = scene.render(cams[a], H, W, f, ss=1)
        right, up, back, c = llff_synth.Scene3D.camera_axes(cams[a])
        j, i = np.mgrid[0:H, 0:W].astype(np.float64)
        d = ((i - 0.5 * W) / f)[..., None] * right - ((j - 0.5 * H) / f)[..., None] * up - back
        col, _, pts = scene.cast(c, d)
        np.testing.assert_allclose(col, rgb_a, atol=1e-12)                       # render() at ss = 1 is cast() along the pixel rays
        pts = pts.reshape(-1, 3)
        uv, zc = scene.project(cams[b], H, W, f, pts)
        col_b, _, first = scene.cast(cams[b][:, 3], pts - cams[b][:, 3])         # what B sees in the direction of A's surface points
        vis = np.linalg.norm(first - pts, axis=-1) < 1e-6
        np.testing.assert_allclose(col_b[vis], col.reshape(-1, 3)[vis], atol=1e-9)   # the same surface point has the same colour from anywhere
        inside = (zc > 0) & (uv[:, 0] >= 0) & (uv[:, 0] <= W - 1) & (uv[:, 1] >= 0) & (uv[:, 1] <= H - 1)
        seen_frac.append(float((vis & inside).mean()))
        occluded += int((~vis & inside).sum())
        # parallax: the disparity of the reprojection differs between near and far surface points
        depth_a = -pts[:, 2]
        shift = np.linalg.norm(uv - np.stack([i.reshape(-1), j.reshape(-1)], -1), axis=-1)
        near, far = shift[(depth_a < 8) & inside & vis], shift[(depth_a > 12) & inside & vis]
        if len(near) > 10 and len(far) > 10:
            assert abs(np.median(near) - np.median(far)) > 0.3, 'no parallax between the layers'
    assert min(seen_frac) > 0.3 and occluded > 20, (seen_frac, occluded)        # overlapping views, and real occlusion


def test_pictures_and_bounds_are_the_ray_cast(ds):
    from PIL import Image
    root, arr, cams, scene = ds
    names = sorted(os.listdir(os.path.join(root, f'images_{F}')))
    assert len(names) == N and len(os.listdir(os.path.join(root, 'images'))) == N
    for k in (0, 4, 8):
        img = np.asarray(Image.open(os.path.join(root, f'images_{F}', names[k])), dtype=np.float64) / 255
        want, _ = scene.render(cams[k], H, W, llff_synth.FOCAL_PER_WIDTH * W, ss=2)
        assert np.abs(img - want).max() <= 0.5 / 255 + 1e-9
        # depth bounds = percentiles of the CAMERA-axis depths of what the view sees (LLFF's pose tool): inside the range of its own pixels' surface depths
        right, up, back, c = llff_synth.Scene3D.camera_axes(cams[k])
        j, i = np.mgrid[0:H, 0:W].astype(np.float64)
        f = llff_synth.FOCAL_PER_WIDTH * W
        _, _, pts = scene.cast(c, ((i - 0.5 * W) / f)[..., None] * right - ((j - 0.5 * H) / f)[..., None] * up - back)
        zc = -((pts - c) @ back)
        assert zc.min() * 0.9 <= arr[k, 15] < arr[k, 16] <= zc.max() * 1.1, (arr[k, 15:], zc.min(), zc.max())
    assert arr[:, 15].min() > 2.5 and arr[:, 16].max() <= 14.0 * 1.5                   # floor edge .. wall (oblique rays reach the wall farther out)


def test_colmap_model_holds_real_visibility(ds):
    root, arr, cams, scene = ds
    images = cu.read_images_binary(os.path.join(root, 'sparse', '0', 'images.bin'))
    points = cu.read_points3d_binary(os.path.join(root, 'sparse', '0', 'points3D.bin'))
    assert len(images) == N and len(points) == NP
    by_name = sorted(images.values(), key=lambda im: im.name)
    assert [im.id for im in by_name] != sorted(im.id for im in by_name)                 # ids are a permutation, not the file order
    view_of = {im.id: k for k, im in enumerate(by_name)}
    Hf, Wf, f_hi = H * F, W * F, llff_synth.FOCAL_PER_WIDTH * W * F
    xyz = np.stack([p.xyz for p in points.values()])
    member = np.zeros((N, NP), bool)
    for col, p in enumerate(points.values()):
        assert len(p.image_ids) >= 1
        member[[view_of[int(i)] for i in p.image_ids], col] = True
    for k in range(N):
        uv, zc = scene.project(cams[k], Hf, Wf, f_hi, xyz)
        inside = (zc > 0) & (uv[:, 0] >= 0) & (uv[:, 0] <= Wf - 1) & (uv[:, 1] >= 0) & (uv[:, 1] <= Hf - 1)
        _, _, first = scene.cast(cams[k][:, 3], xyz - cams[k][:, 3])
        truly = inside & (np.linalg.norm(first - xyz, axis=-1) < 1e-6 * (1 + np.abs(xyz).max()))
        np.testing.assert_array_equal(member[k], truly)                               # a view is in a point's track iff it really sees the point
        im = by_name[k]
        pid = {int(p.id): c for c, p in enumerate(points.values())}
        cols = [pid[int(q)] for q in im.point3D_ids]
        np.testing.assert_allclose(im.xys, uv[cols], atol=1e-9)                       # the 2-D observations are the projections
        R = cu.qvec2rotmat(im.qvec)                                                    # COLMAP pose: x_cam = R x_world + t, camera looks along +z
        pc = (R @ xyz[cols].T).T + im.tvec
        np.testing.assert_allclose(pc[:, 2], zc[cols], rtol=1e-9, atol=1e-9)
    assert member.sum(0).min() >= 1 and member.sum(0).max() >= 4 and not member.all()    # tracks of different lengths; nobody sees everything


def test_loader_ranks_real_neighbours(ds):
    root, arr, cams, scene = ds
    images, poses, bds, _, i_test, i_ref = L.load_llff_data_infer(root, factor=F, llffhold=8, num_neighbor=4)
    assert images.shape == (N, H, W, 3) and list(i_test) == [0, 8] and len(i_ref) == 4 and not set(i_ref) & set(i_test)
    assert abs(float(bds.min()) - 1.0 / 0.75) < 1e-5                                   # bd_factor: nearest bound -> 1 / 0.75
    # brute force on the TRUE visibility (recomputed from the geometry): repeatedly the training view that sees the most uncovered points
    points = cu.read_points3d_binary(os.path.join(root, 'sparse', '0', 'points3D.bin'))
    xyz = np.stack([p.xyz for p in points.values()])
    Hf, Wf, f_hi = H * F, W * F, llff_synth.FOCAL_PER_WIDTH * W * F
    i_train = [i for i in range(N) if i not in i_test]
    vis = np.zeros((len(i_train), NP), bool)
    for r, k in enumerate(i_train):
        uv, zc = scene.project(cams[k], Hf, Wf, f_hi, xyz)
        _, _, first = scene.cast(cams[k][:, 3], xyz - cams[k][:, 3])
        vis[r] = (zc > 0) & (uv[:, 0] >= 0) & (uv[:, 0] <= Wf - 1) & (uv[:, 1] >= 0) & (uv[:, 1] <= Hf - 1) & (np.linalg.norm(first - xyz, axis=-1) < 1e-6 * (1 + np.abs(xyz).max()))
    picks = []
    for _ in range(4):
        best = int(np.argmax(vis.sum(1)))
        picks.append(i_train[best])
        vis &= ~vis[best][None]
    assert list(i_ref) == picks


def test_scene3d_frame_is_that_scene_at_the_fern_frame_size():
    from pronerf_amd import synthetic
    fr = synthetic.scene3d_frame(0, 4)
    assert (fr['H'], fr['W']) == (756, 1008) and fr['poses'].shape == (17, 3, 4) and fr['images'].shape == (17, 189, 252, 3) and fr['gt_small'].shape == (189, 252, 3)
    assert abs(fr['K'][0, 0] / 1008 - llff_synth.FOCAL_PER_WIDTH) < 1e-4 and fr['K'][0, 2] == 504 and fr['K'][1, 2] == 378
    f1 = synthetic.scene3d_frame(0, 1)
    assert (f1['H'], f1['W']) == (189, 252) and np.array_equal(f1['c2w'], fr['c2w']) and abs(fr['K'][0, 0] - 4 * f1['K'][0, 0]) < 1e-3
    w = synthetic.weight_set(0, 'scene')
    assert len(w['sampler']['W']) == 7 and len(w['nerf']['W']) == 12 and float(w['info']['stage1_iters']) == 20000
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'trained_scene3d.npz'))
    assert str(g['scene']) == 'consistent' and g['psnr_holdout'].min() > 30.0            # what the training run measured on the hold-out views
