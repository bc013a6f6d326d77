"""LLFF / COLMAP loader against goldens produced by the reference's own loader on the same synthetic
scene (oracle/gen_golden_llff.py).  Host-side numpy: CPU only."""
import os

import numpy as np
import pytest

import llff_synth
from pronerf_amd import colmap_utils as cu
from pronerf_amd import load_llff as L

GOLD = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'llff_loader.npz'))


@pytest.fixture(scope='module')
def scene(tmp_path_factory):
    return llff_synth.make_dataset(str(tmp_path_factory.mktemp('llff')), seed=0)


def test_colmap_readers_match_reference(scene):
    im = cu.read_images_binary(os.path.join(scene, 'sparse/0/images.bin'))
    pt = cu.read_points3d_binary(os.path.join(scene, 'sparse/0/points3D.bin'))
    assert np.array_equal(np.array(list(im.keys())), GOLD['colmap_image_ids'])           # file order kept
    assert np.array_equal(np.array([np.concatenate([v.qvec, v.tvec]) for v in im.values()]), GOLD['colmap_image_qt'])
    assert [v.name for v in im.values()] == list(GOLD['colmap_image_names'])
    assert np.array_equal(np.array([len(v.point3D_ids) for v in im.values()]), GOLD['colmap_image_npts'])
    assert np.array_equal(np.array([int(v.point3D_ids.sum()) for v in im.values()]), GOLD['colmap_image_p3d_sum'])
    assert np.array_equal(np.array(list(pt.keys())), GOLD['colmap_point_ids'])
    assert np.array_equal(np.array([v.xyz for v in pt.values()]), GOLD['colmap_point_xyz'])
    assert np.array_equal(np.array([int(v.image_ids.sum()) * 1000 + int(v.point2D_idxs.sum()) for v in pt.values()]), GOLD['colmap_point_track_sum'])


def test_colmap_write_read_round_trip(scene, tmp_path):
    im = cu.read_images_binary(os.path.join(scene, 'sparse/0/images.bin'))
    pt = cu.read_points3d_binary(os.path.join(scene, 'sparse/0/points3D.bin'))
    cu.write_images_binary(tmp_path / 'i.bin', im)
    cu.write_points3d_binary(tmp_path / 'p.bin', pt)
    assert open(tmp_path / 'i.bin', 'rb').read() == open(os.path.join(scene, 'sparse/0/images.bin'), 'rb').read()
    assert open(tmp_path / 'p.bin', 'rb').read() == open(os.path.join(scene, 'sparse/0/points3D.bin'), 'rb').read()
    R = cu.qvec2rotmat(np.array([0.5, 0.5, -0.5, 0.5]))
    assert np.allclose(R @ R.T, np.eye(3)) and np.isclose(np.linalg.det(R), 1.0)


@pytest.mark.parametrize('tag,kw', [('std', {}), ('sph', {'spherify': True})])
def test_load_llff_data_matches_reference(scene, tag, kw):
    images, poses, bds, render_poses, i_test = L.load_llff_data(scene, factor=4, recenter=True, bd_factor=.75, **kw)
    assert poses.dtype == np.float32 and images.dtype == np.float32
    np.testing.assert_allclose(poses, GOLD[f'{tag}_poses'], rtol=0, atol=2e-6)
    np.testing.assert_allclose(bds, GOLD[f'{tag}_bds'], rtol=0, atol=2e-6)
    np.testing.assert_allclose(render_poses, GOLD[f'{tag}_render_poses'], rtol=0, atol=2e-5)
    assert int(i_test) == int(GOLD[f'{tag}_i_test'])
    if tag == 'std':
        assert list(images.shape) == list(GOLD['images_shape'])
        np.testing.assert_allclose(images.astype(np.float64).sum((1, 2, 3)), GOLD['images_sum'], rtol=1e-12)
        assert poses[0, 0, 4] == 24 and poses[0, 1, 4] == 32 and np.isclose(poses[0, 2, 4], llff_synth.FOCAL_PER_WIDTH * 32)     # H, W, focal/factor


@pytest.mark.parametrize('nn', [1, 3, 5])
def test_reference_view_selection_matches_reference(scene, nn):
    images, poses, bds, render_poses, i_test, i_ref = L.load_llff_data_infer(scene, factor=4, num_neighbor=nn)
    assert np.array_equal(i_test, GOLD['infer_i_test'])
    assert np.array_equal(i_ref, GOLD[f'infer_i_ref_{nn}'])
    np.testing.assert_allclose(poses, GOLD['infer_poses'], rtol=0, atol=2e-6)


def test_reference_view_selection_none_ranks_all_training_views(scene):
    *_, i_test, i_ref = L.load_llff_data_infer(scene, factor=4, num_neighbor=None)
    i_train = np.array([i for i in range(10) if i not in i_test])
    assert sorted(i_ref.tolist()) == i_train.tolist()                          # a permutation of the training views
    assert np.array_equal(i_ref[:5], GOLD['infer_i_ref_5'])                   # same greedy prefix as the reference


def test_path_zflat_and_minify(scene, tmp_path):
    *_, render_poses, _ = L.load_llff_data(scene, factor=4, path_zflat=True)
    assert render_poses.shape == (60, 3, 5)
    import shutil
    d = tmp_path / 's'
    shutil.copytree(scene, d)
    shutil.rmtree(d / 'images_4')
    images, poses, *_ = L.load_llff_data(str(d), factor=4)                     # images_4 rebuilt from images/
    assert images.shape == (10, 24, 32, 3) and (d / 'images_4').is_dir()
    with pytest.raises(ValueError):
        os.remove(d / 'images_4' / sorted(os.listdir(d / 'images_4'))[0])
        L.load_llff_data(str(d), factor=4)
