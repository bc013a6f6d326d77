"""Synthetic LLFF scene on disk (poses_bounds.npy, images/, images_<factor>/, sparse/0/*.bin) for the loader tests and for
oracle/gen_golden.py --llff (which runs the reference's loader on the same directory).  Deterministic in `seed`.

Two kinds.  ``make_dataset(..)`` (rounds 1-5): twenty INDEPENDENT pictures on a camera rig with a random COLMAP model — a file-format
fixture; a net fitted to it memorises pictures and has nothing to generalise to.  ``make_dataset(.., consistent=True)`` (round 6): ONE
3-D scene (``Scene3D``: a textured back wall, a floor receding in depth, two layers of occluding discs) ray-cast analytically from the
same forward-facing rig — every view is a picture of the same geometry, hold-out views are predictable from the training views, the
neighbour-view projection of the renderer finds the colour of the same surface point, the depth bounds are the views' real depth
ranges, and the COLMAP model holds real scene points with the tracks of the views that really see them (inside the image and not
occluded), so that ``load_llff_data_infer``'s greedy visibility ranking (reference load_llff.py:496-547) selects real neighbours."""
import os

import numpy as np

from pronerf_amd import colmap_utils as cu


FOCAL_PER_WIDTH = 0.8086          # Fern: 3260.5 px focal on 4032 px wide images


class Scene3D:
    """World frame = the frame of poses_bounds.npy; the cameras sit near the origin and look down -z.  LLFF stores a camera's axes as the
    columns [down, right, backwards] (reference load_llff.py:354-356 turns them into NeRF's [right, up, backwards]); with the rig's small
    rotations: down ~ +x, right ~ +y, backwards ~ +z.  Surfaces (all Lambertian, colours a function of the surface point only):
      wall   the plane z = -z_wall, opaque everywhere;
      floor  the plane x = x_floor (below the cameras), from z = -z_floor0 back to the wall: continuous depths;
      discs  two layers z = -z_l of opaque discs with soft-textured faces: occluders with real parallax against the wall."""

    def __init__(self, seed=0, detail=0.12):
        """detail: amplitude of the fine texture (periods of 2 .. 8 pixels in the 189 x 252 views) — what a real photograph has and an 8-sample
        NeRF does not resolve: it puts the fitted nets' PSNR where real scenes are (25 .. 33 dB) instead of at 42 dB (detail = 0)."""
        rng = np.random.RandomState(1000 + seed)
        self.detail = float(detail)
        self.dph = np.random.RandomState(2000 + seed).uniform(0, 6.28, (3, 3, 2))
        self.z_wall, self.x_floor, self.z_floor0 = 14.0, 2.6, 3.2
        self.layers = []
        for z, k, r0, r1 in ((7.0, 7, 0.7, 1.3), (4.2, 4, 0.3, 0.55)):
            ext = 0.45 * z                                             # half-extent of the area the rig sees at that depth
            self.layers.append({'z': z, 'c': rng.uniform(-ext, ext, (k, 2)) * np.array([0.8, 1.0]) - np.array([0.1 * z, 0.0]), 'r': rng.uniform(r0, r1, k),
                                'hue': rng.uniform(0, 1, (k, 3)), 'f': rng.uniform(1.5, 4.0, (k, 2)), 'ph': rng.uniform(0, 6.28, (k, 2))})
        self.wf, self.wp = rng.uniform(0.35, 1.4, (3, 2)), rng.uniform(0, 6.28, (3, 2))
        self.ff, self.fp = rng.uniform(0.5, 1.6, (3, 2)), rng.uniform(0, 6.28, (3, 2))

    @staticmethod
    def _soft_checker(u, v, period):
        return 0.5 + 0.5 * np.tanh(4.0 * np.sin(np.pi * u / period) * np.sin(np.pi * v / period))

    def _fine(self, which, u, v):
        return self.detail * np.stack([np.sin(u + self.dph[which, c, 0]) * np.sin(v + self.dph[which, c, 1]) for c in range(3)], -1)

    def _wall_rgb(self, x, y):
        base = np.stack([0.5 + 0.22 * np.sin(self.wf[c, 0] * x + self.wp[c, 0]) + 0.18 * np.cos(self.wf[c, 1] * y + self.wp[c, 1]) for c in range(3)], -1)
        return np.clip(base + 0.10 * (self._soft_checker(x, y, 2.2)[..., None] - 0.5) + self._fine(0, 17.0 * x, 19.0 * y), 0, 1)

    def _floor_rgb(self, y, z):
        base = np.stack([0.45 + 0.2 * np.sin(self.ff[c, 0] * y + self.fp[c, 0]) * np.cos(self.ff[c, 1] * z + self.fp[c, 1]) for c in range(3)], -1)
        return np.clip(base + 0.22 * (self._soft_checker(y, z, 1.3)[..., None] - 0.5) + self._fine(1, 23.0 * y, 13.0 * z), 0, 1)

    def cast(self, o, d):
        """First hit of the rays o + t d (d_z < 0): (rgb [...,3], depth = -z of the hit point [...], hit point [...,3])."""
        o, d = np.broadcast_arrays(np.asarray(o, np.float64), np.asarray(d, np.float64))
        t_best = (-self.z_wall - o[..., 2]) / d[..., 2]
        p = o + t_best[..., None] * d
        rgb = self._wall_rgb(p[..., 0], p[..., 1])
        with np.errstate(divide='ignore', invalid='ignore'):
            t = np.where(d[..., 0] > 1e-9, (self.x_floor - o[..., 0]) / d[..., 0], np.inf)
        q = o + np.where(np.isfinite(t), t, 0.0)[..., None] * d
        hit = np.isfinite(t) & (t > 0) & (t < t_best) & (-q[..., 2] >= self.z_floor0)
        rgb = np.where(hit[..., None], self._floor_rgb(q[..., 1], q[..., 2]), rgb)
        t_best = np.where(hit, t, t_best)
        for L in self.layers:
            t = (-L['z'] - o[..., 2]) / d[..., 2]
            q = o + t[..., None] * d
            for k in range(len(L['r'])):
                du, dv = q[..., 0] - L['c'][k, 0], q[..., 1] - L['c'][k, 1]
                hit = (du * du + dv * dv <= L['r'][k] ** 2) & (t < t_best) & (t > 0)
                if hit.any():
                    tex = 0.55 + 0.3 * np.sin(L['f'][k, 0] * du + L['ph'][k, 0]) * np.sin(L['f'][k, 1] * dv + L['ph'][k, 1])
                    rgb = np.where(hit[..., None], np.clip(L['hue'][k] * 0.6 + 0.4 * tex[..., None] + self._fine(2, 29.0 * du, 31.0 * dv), 0, 1), rgb)
                    t_best = np.where(hit, t, t_best)
        p = o + t_best[..., None] * d
        return rgb, -p[..., 2], p

    @staticmethod
    def camera_axes(p35):
        """(right, up, backwards, centre) of a poses_bounds camera [3,5] (columns down, right, backwards, centre, hwf)."""
        return p35[:, 1], -p35[:, 0], p35[:, 2], p35[:, 3]

    def render(self, p35, H, W, focal, ss=1):
        """Picture [H,W,3] in [0,1] and depth map [H,W] of the camera: pixel (i, j) looks along right (i - W/2)/f - up (j - H/2)/f - backwards,
        the rays of the reference's get_rays (run_nerf_helpers.py:2705-2714); ss x ss sub-pixel samples around that direction, box-filtered."""
        right, up, back, c = self.camera_axes(p35)
        acc, depth = 0.0, None
        offs = (np.arange(ss) + 0.5) / ss - 0.5
        j, i = np.mgrid[0:H, 0:W].astype(np.float64)
        for dy in offs:
            for dx in offs:
                d = ((i + dx - 0.5 * W) / focal)[..., None] * right - ((j + dy - 0.5 * H) / focal)[..., None] * up - back
                rgb, z, _ = self.cast(c, d)
                acc = acc + rgb
                depth = z if depth is None else np.minimum(depth, z)
        return acc / (ss * ss), depth

    def project(self, p35, H, W, focal, pts):
        """Pixel coordinates [n,2] and camera depth [n] of world points in the camera."""
        right, up, back, c = self.camera_axes(p35)
        v = pts - c
        zc = -(v @ back)
        with np.errstate(divide='ignore', invalid='ignore'):
            return np.stack([(v @ right) / zc * focal + 0.5 * W, -(v @ up) / zc * focal + 0.5 * H], -1), zc


def _rotmat2qvec(R):
    """COLMAP quaternion (w, x, y, z) of a rotation matrix (inverse of colmap_utils.qvec2rotmat)."""
    K = np.array([[R[0, 0] - R[1, 1] - R[2, 2], 0, 0, 0], [R[1, 0] + R[0, 1], R[1, 1] - R[0, 0] - R[2, 2], 0, 0],
                  [R[2, 0] + R[0, 2], R[2, 1] + R[1, 2], R[2, 2] - R[0, 0] - R[1, 1], 0],
                  [R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1], R[0, 0] + R[1, 1] + R[2, 2]]]) / 3.0
    w, v = np.linalg.eigh(K)
    q = v[[3, 0, 1, 2], np.argmax(w)]
    return -q if q[0] < 0 else q


def _consistent_dataset(root, rng, seed, names, arr, n, H, W, factor, n_points, hires):
    """Pictures, depth bounds and the COLMAP model of ``Scene3D(seed)`` for the rig in ``arr`` (filled in place: the two bound columns).
    The loader reads ``images_<factor>``: those are ray-cast at their own pixel grid (2 x 2 sub-samples).  The full-size JPEGs exist for the
    directory layout only (nothing reads them while ``images_<factor>`` exists); ``hires`` ray-casts them too (20 s for twenty 756 x 1008
    views), otherwise they are the small pictures enlarged."""
    from PIL import Image
    scene = Scene3D(seed)
    Hf, Wf, f_hi = H * factor, W * factor, FOCAL_PER_WIDTH * W * factor
    cams = [arr[i, :15].reshape(3, 5) for i in range(n)]
    for i, nm in enumerate(names):
        lo, _ = scene.render(cams[i], H, W, f_hi / factor, ss=2)               # the picture the loader reads: rendered at ITS pixel grid
        lo8 = Image.fromarray(np.round(lo * 255).astype(np.uint8))
        lo8.save(os.path.join(root, f'images_{factor}', nm + '.png'))
        hi8 = Image.fromarray(np.round(scene.render(cams[i], Hf, Wf, f_hi, ss=1)[0] * 255).astype(np.uint8)) if hires else lo8.resize((Wf, Hf), Image.BICUBIC)
        hi8.save(os.path.join(root, 'images', nm + '.JPG'), quality=92)
    # scene points: the first hit of random pixels of random views — visible from at least that view by construction
    src = rng.randint(0, n, n_points)
    pix = np.stack([rng.uniform(0, Wf - 1, n_points), rng.uniform(0, Hf - 1, n_points)], -1)
    xyz, col = np.zeros((n_points, 3)), np.zeros((n_points, 3))
    for i in range(n):
        m = src == i
        right, up, back, c = Scene3D.camera_axes(cams[i])
        d = ((pix[m, 0] - 0.5 * Wf) / f_hi)[:, None] * right - ((pix[m, 1] - 0.5 * Hf) / f_hi)[:, None] * up - back
        col[m], _, xyz[m] = scene.cast(c, d)
    obs = {}                                                            # view -> list of (pixel, point index)
    tracks = [[] for _ in range(n_points)]
    for i in range(n):
        uv, zc = scene.project(cams[i], Hf, Wf, f_hi, xyz)
        inside = (zc > 0) & (uv[:, 0] >= 0) & (uv[:, 0] <= Wf - 1) & (uv[:, 1] >= 0) & (uv[:, 1] <= Hf - 1)
        _, _, first = scene.cast(cams[i][:, 3], xyz - cams[i][:, 3])       # what the view really sees in the point's direction
        seen = inside & (np.linalg.norm(first - xyz, axis=-1) < 1e-6 * (1 + np.abs(xyz).max()))
        obs[i] = [(uv[k], k) for k in np.nonzero(seen)[0]]
        for k in np.nonzero(seen)[0]:
            tracks[k].append(i)
        zs = zc[seen]
        arr[i, 15:] = [np.percentile(zs, 0.1), np.percentile(zs, 99.9)]     # LLFF's pose tool: the 0.1 / 99.9 percentiles of the view's point depths
    ids = rng.permutation(n) + 1                                        # image ids are a permutation, the records in another order again
    pid = rng.permutation(n_points) + 1
    images = {}
    for k in rng.permutation(n):
        right, up, back, c = Scene3D.camera_axes(cams[k])
        Rw2c = np.stack([right, -up, -back], 0)                         # COLMAP camera axes: right, down, forwards
        images[int(ids[k])] = cu.Image(id=int(ids[k]), qvec=_rotmat2qvec(Rw2c), tvec=-Rw2c @ c, camera_id=1, name=names[k] + '.JPG',
                                      xys=np.array([u for u, _ in obs[k]]).reshape(-1, 2), point3D_ids=np.array([pid[j] for _, j in obs[k]], dtype=np.int64))
    slot = {k: {j: s for s, (_, j) in enumerate(obs[k])} for k in range(n)}
    points = {}
    for j in rng.permutation(n_points):
        points[int(pid[j])] = cu.Point3D(id=int(pid[j]), xyz=xyz[j], rgb=np.round(col[j] * 255).astype(np.int64), error=np.array(rng.uniform(0.2, 1.2)),
                                        image_ids=np.array([ids[v] for v in tracks[j]]), point2D_idxs=np.array([slot[v][j] for v in tracks[j]]))
    return images, points


def make_dataset(root, seed=0, n=10, H=24, W=32, factor=4, n_points=400, consistent=False, hires=False):
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

    for nm in ([] if consistent else names):
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
    if consistent:
        images, points = _consistent_dataset(root, rng, seed, names, arr, n, H, W, factor, n_points, hires)
        np.save(os.path.join(root, 'poses_bounds.npy'), arr)
        cu.write_images_binary(os.path.join(root, 'sparse', '0', 'images.bin'), images)
        cu.write_points3d_binary(os.path.join(root, 'sparse', '0', 'points3D.bin'), points)
        return root
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
