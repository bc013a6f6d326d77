#!/usr/bin/env python3
"""Golden outputs of the reference's LLFF loader on the synthetic scene of tests/llff_synth.py.

Runs /root/reference/load_llff.py (``load_llff_data``, ``load_llff_data_infer``) and colmap_utils.py in THIS container and
writes tests/golden/llff_loader.npz.  The reference imports ``imageio`` at module level (absent here): it is stubbed with a
PIL-backed ``imread`` — pixel decoding is not what these goldens pin; poses, bounds, the spiral path, hold-out / reference
view indices and the COLMAP records are.  Test infrastructure only; the reference never leaves this container.

    python oracle/gen_golden_llff.py
"""
import os
import sys
import tempfile
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.dont_write_bytecode = True


def main():
    from PIL import Image
    import llff_synth
    imageio = types.ModuleType('imageio')
    imageio.imread = lambda f, **kw: np.asarray(Image.open(f))
    sys.modules['imageio'] = imageio
    sys.modules.setdefault('cv2', types.ModuleType('cv2'))          # imported at module level by the reference loader, unused by these calls
    sys.path.insert(0, REF)
    import load_llff as ref
    import colmap_utils as ref_cu
    out = {}
    with tempfile.TemporaryDirectory() as d:
        llff_synth.make_dataset(d, seed=0)
        # path_zflat=True is not pinned: the reference passes N_views/2 (a float) to np.linspace, a TypeError under numpy 2
        for tag, kw in (('std', {}), ('sph', {'spherify': True})):
            images, poses, bds, render_poses, i_test = ref.load_llff_data(d, factor=4, recenter=True, bd_factor=.75, **kw)
            out[f'{tag}_poses'], out[f'{tag}_bds'], out[f'{tag}_render_poses'], out[f'{tag}_i_test'] = poses, bds, render_poses, np.int64(i_test)
            if tag == 'std':
                out['images_sum'] = images.astype(np.float64).sum((1, 2, 3))
                out['images_shape'] = np.array(images.shape)
        for nn in (1, 3, 5):
            images, poses, bds, render_poses, i_test, i_ref = ref.load_llff_data_infer(d, factor=4, recenter=True, bd_factor=.75, num_neighbor=nn)
            out[f'infer_i_ref_{nn}'] = np.asarray(i_ref, dtype=np.int64)
        out['infer_i_test'] = np.asarray(i_test, dtype=np.int64)
        out['infer_poses'] = poses
        im = ref_cu.read_images_binary(os.path.join(d, 'sparse/0/images.bin'))
        pt = ref_cu.read_points3d_binary(os.path.join(d, 'sparse/0/points3D.bin'))
        out['colmap_image_ids'] = np.array(list(im.keys()), dtype=np.int64)
        out['colmap_image_qt'] = np.array([np.concatenate([v.qvec, v.tvec]) for v in im.values()])
        out['colmap_image_names'] = np.array([v.name for v in im.values()])
        out['colmap_image_npts'] = np.array([len(v.point3D_ids) for v in im.values()], dtype=np.int64)
        out['colmap_image_p3d_sum'] = np.array([int(v.point3D_ids.sum()) for v in im.values()], dtype=np.int64)
        out['colmap_point_ids'] = np.array(list(pt.keys()), dtype=np.int64)
        out['colmap_point_xyz'] = np.array([v.xyz for v in pt.values()])
        out['colmap_point_track_sum'] = np.array([int(v.image_ids.sum()) * 1000 + int(v.point2D_idxs.sum()) for v in pt.values()], dtype=np.int64)
    path = os.path.join(ROOT, 'tests', 'golden', 'llff_loader.npz')
    np.savez_compressed(path, **out)
    print(path, {k: getattr(v, 'shape', None) for k, v in out.items()})


if __name__ == '__main__':
    main()
