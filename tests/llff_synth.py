"""Synthetic LLFF scene on disk (poses_bounds.npy, images/, images_<factor>/, sparse/0/*.bin) for the loader tests and for
oracle/gen_golden.py --llff (which runs the reference's loader on the same directory).  Deterministic in `seed`."""
import os

import numpy as np

from pronerf_amd import colmap_utils as cu


FOCAL_PER_WIDTH = 0.8086          # Fern: 3260.5 px focal on 4032 px wide images


def make_dataset(root, seed=0, n=10, H=24, W=32, factor=4, n_points=400):
    from PIL import Image
    rng = np.random.RandomState(seed)
    os.makedirs(os.path.join(root, 'images'), exist_ok=True)
    os.makedirs(os.path.join(root, f'images_{factor}'), exist_ok=True)
    os.makedirs(os.path.join(root, 'sparse', '0'), exist_ok=True)
    names = [f'IMG_{4000 + 3 * i:04d}' for i in range(n)]
    def picture(h, w):           # smooth colour field + a little noise (pure noise would make bilinear taps chaotic)
        y, x = np.mgrid[0:h, 0:w].astype(np.float64)
        f = rng.uniform(0.5, 2.5, (3, 2)); ph = rng.uniform(0, 6.28, (3, 2))
        img = np.stack([0.5 + 0.25 * np.sin(f[c, 0] * 6.28 * x / w + ph[c, 0]) + 0.2 * np.cos(f[c, 1] * 6.28 * y / h + ph[c, 1]) for c in range(3)], -1)
        return np.clip((img + rng.normal(0, 0.01, img.shape)) * 255, 0, 255).astype(np.uint8)

    for nm in names:
        Image.fromarray(picture(H * factor, W * factor)).save(os.path.join(root, 'images', nm + '.JPG'), quality=90)
        Image.fromarray(picture(H, W)).save(os.path.join(root, f'images_{factor}', nm + '.png'))
    # forward-facing rig: small rotations about a common direction, cameras spread on a plane
    arr = np.zeros((n, 17))
    for i in range(n):
        a = rng.normal(0, 0.08, 3)
        Rx = np.array([[1, 0, 0], [0, np.cos(a[0]), -np.sin(a[0])], [0, np.sin(a[0]), np.cos(a[0])]])
        Ry = np.array([[np.cos(a[1]), 0, np.sin(a[1])], [0, 1, 0], [-np.sin(a[1]), 0, np.cos(a[1])]])
        Rz = np.array([[np.cos(a[2]), -np.sin(a[2]), 0], [np.sin(a[2]), np.cos(a[2]), 0], [0, 0, 1]])
        R = Rz @ Ry @ Rx
        t = np.array([rng.uniform(-1.5, 1.5), rng.uniform(-1.0, 1.0), rng.normal(0, 0.1)])
        p = np.concatenate([R, t[:, None], np.array([[H * factor], [W * factor], [FOCAL_PER_WIDTH * W * factor]])], 1)      # [3,5]
        arr[i, :15] = p.reshape(-1)
        arr[i, 15:] = [rng.uniform(3.5, 5.0), rng.uniform(40.0, 60.0)]
    np.save(os.path.join(root, 'poses_bounds.npy'), arr)
    # COLMAP model: image ids are a permutation (not the file order); tracks of 2..6 images per point
    ids = rng.permutation(n) + 1
    images = {}
    for k in rng.permutation(n):                       # file order of the records differs from the name order too
        m = rng.randint(3, 9)
        images[int(ids[k])] = cu.Image(id=int(ids[k]), qvec=rng.normal(size=4), tvec=rng.normal(size=3), camera_id=1,
                                      name=names[k] + '.JPG', xys=rng.uniform(0, 100, (m, 2)), point3D_ids=rng.randint(-1, n_points, m))
    points = {}
    for pid in rng.permutation(n_points) + 1:
        # views near each other see the same points: pick a window of consecutive views
        c, w = rng.randint(0, n), rng.randint(2, 7)
        members = sorted({int(np.clip(c + d, 0, n - 1)) for d in range(-(w // 2), w - w // 2)})
        points[int(pid)] = cu.Point3D(id=int(pid), xyz=rng.normal(size=3), rgb=rng.randint(0, 256, 3), error=np.array(rng.uniform(0, 2)),
                                      image_ids=np.array([ids[m] for m in members]), point2D_idxs=rng.randint(0, 8, len(members)))
    cu.write_images_binary(os.path.join(root, 'sparse', '0', 'images.bin'), images)
    cu.write_points3d_binary(os.path.join(root, 'sparse', '0', 'points3D.bin'), points)
    return root
