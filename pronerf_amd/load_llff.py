"""LLFF scene loader: ``poses_bounds.npy`` + ``images_<factor>/`` (+ COLMAP ``sparse/0`` for the reference-view
selection) -> images, poses, bounds, spiral render path, hold-out and reference view indices.

Same entry points, argument meaning and return order as the reference's ``load_llff.py``
(``load_llff_data`` :349-421, ``load_llff_data_infer`` :423-547, ``recenter_poses`` :189-203, ``poses_avg`` :163-172,
``render_path_spiral`` :176-185, ``spherify_poses`` :207-262) so its driver scripts can import this module instead.
Host-side numpy only: this is the data-format row of SURVEY.md §8(f)3, it feeds ``render_path`` and is not on the
per-ray path.

Differences that are decisions, not accidents:
  * images are decoded with PIL (the reference uses imageio; identical for 8-bit PNG/JPEG);
  * a missing ``images_<factor>`` directory is created with PIL box-filter resizing instead of shelling out to
    ImageMagick ``mogrify`` (load_llff.py:12-60) — no subprocess, no mutation of the original ``images`` directory;
  * ``load_llff_data_infer(num_neighbor=None)``: the reference then fails in ``range(None)``; here
    ``None`` ranks *all* training views greedily by newly covered COLMAP points;
  * the greedy selection runs on a boolean visibility matrix built with one scatter per point track instead of a
    Python loop with ``list.index`` per observation (O(points·track·views) in the reference).
"""
from __future__ import annotations

import os

import numpy as np

from .colmap_utils import read_images_binary, read_points3d_binary

_IMG_EXT = ('JPG', 'jpg', 'png')


def _imread(path):
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert('RGB') if im.mode not in ('RGB', 'RGBA') else im)


def _image_files(d):
    return [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith(_IMG_EXT)]


def _minify(basedir, factor):
    """Create ``images_<factor>`` (PNG) from ``images`` if it does not exist yet."""
    from PIL import Image
    src, dst = os.path.join(basedir, 'images'), os.path.join(basedir, f'images_{factor}')
    if os.path.exists(dst):
        return
    os.makedirs(dst)
    for f in _image_files(src):
        with Image.open(f) as im:
            w, h = im.size
            im.convert('RGB').resize((w // factor, h // factor), Image.BOX).save(
                os.path.join(dst, os.path.splitext(os.path.basename(f))[0] + '.png'))


def _load_data(basedir, factor=None, load_imgs=True):
    """poses [3,5,N] (columns: 3x3 rotation in LLFF order, translation, [H, W, focal]), bds [2,N], imgs [H,W,3,N] in [0,1]
    (load_llff.py:66-124; the width=/height= variants of the reference are not used by any driver and are not provided)."""
    arr = np.load(os.path.join(basedir, 'poses_bounds.npy'))
    poses = arr[:, :-2].reshape([-1, 3, 5]).transpose([1, 2, 0])
    bds = arr[:, -2:].transpose([1, 0])
    sfx = ''
    if factor is not None and factor != 1:
        sfx = f'_{factor}'
        _minify(basedir, factor)
    else:
        factor = 1
    imgdir = os.path.join(basedir, 'images' + sfx)
    if not os.path.isdir(imgdir):
        raise FileNotFoundError(f'{imgdir} does not exist')
    files = _image_files(imgdir)
    if poses.shape[-1] != len(files):
        raise ValueError(f'mismatch between images ({len(files)}) and poses ({poses.shape[-1]}) in {basedir}')
    sh = _imread(files[0]).shape
    poses[:2, 4, :] = np.array(sh[:2]).reshape([2, 1])
    poses[2, 4, :] = poses[2, 4, :] * 1. / factor
    if not load_imgs:
        return poses, bds
    imgs = np.stack([_imread(f)[..., :3] / 255. for f in files], -1)
    return poses, bds, imgs


# ------------------------------------------------------------------------------------------ pose algebra
# Everything below is built on two batched helpers: `_frames` (camera frames from look directions) and `_homog` (3x4 -> 4x4).  The public
# names are the reference's (its drivers import them); paths are generated for all angles at once rather than camera by camera.
def _unit(v):
    v = np.asarray(v, dtype=np.float64) if not isinstance(v, np.ndarray) else v
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def _frames(back, up, origin):
    """Camera-to-world frames [..., 3, 4] with columns (right, true up, backwards, origin): `back` is the direction the camera's +z axis
    points along, `up` an approximate up vector; all broadcast over leading dimensions."""
    zc = _unit(back)
    xc = _unit(np.cross(np.broadcast_to(up, zc.shape), zc))
    yc = _unit(np.cross(zc, xc))
    return np.stack([xc, yc, zc, np.broadcast_to(origin, zc.shape)], axis=-1)


def _homog(m34):
    """[..., 3, 4] -> [..., 4, 4] by appending the row (0, 0, 0, 1)."""
    m34 = np.asarray(m34)
    out = np.zeros(m34.shape[:-2] + (4, 4), dtype=m34.dtype)
    out[..., :3, :] = m34
    out[..., 3, 3] = 1
    return out


def _with_hwf(frames, hwf):
    """Append the [H, W, focal] column to every frame: [..., 3, 4] -> [..., 3, 5]."""
    return np.concatenate([frames, np.broadcast_to(hwf, frames.shape[:-1] + (1,))], axis=-1)


def normalize(x):
    return x / np.linalg.norm(x)


def viewmatrix(z, up, pos):
    return _frames(np.asarray(z), np.asarray(up), np.asarray(pos))


def poses_avg(poses):
    """Average camera [3,5]: mean centre, summed z axis, summed y axis as up (load_llff.py:163-172)."""
    rot = poses[:, :3, :3].sum(0)
    return _with_hwf(_frames(rot[:, 2], rot[:, 1], poses[:, :3, 3].mean(0)), poses[0, :3, 4:5])


def recenter_poses(poses):
    """Express all poses in the frame of the average camera (load_llff.py:189-203)."""
    out = poses.copy()
    out[:, :3, :4] = (np.linalg.inv(_homog(poses_avg(poses)[:, :4])) @ _homog(poses[:, :3, :4]))[:, :3, :4]
    return out


def render_path_spiral(c2w, up, rads, focal, zdelta, zrate, rots, N):
    """Spiral of N cameras around the average pose looking at depth ``focal`` (load_llff.py:176-185; ``zdelta`` is unused there too)."""
    th = np.linspace(0., 2. * np.pi * rots, int(N) + 1)[:-1]
    local = np.stack([np.cos(th), -np.sin(th), -np.sin(th * zrate), np.ones_like(th)], -1) * np.append(np.asarray(rads, dtype=np.float64), 1.)
    m = c2w[:3, :4]
    eye = local @ m.T                                              # [N, 3] camera centres in world coordinates
    target = m @ np.array([0, 0, -focal, 1.])
    return list(_with_hwf(_frames(eye - target, up, eye), c2w[:, 4:5]))


def spherify_poses(poses, bds):
    """360-degree captures: recentre on the point closest to all optical axes, unit mean radius, circular path
    (load_llff.py:207-262)."""
    hwf = poses[0, :3, 4:5]
    d, o = poses[:, :3, 2], poses[:, :3, 3]
    # least squares: the point minimising the summed squared distance to the optical axes o + t d
    P = np.eye(3) - d[:, :, None] * d[:, None, :]
    centre = np.linalg.inv((np.swapaxes(P, 1, 2) @ P).mean(0)) @ (P @ o[:, :, None]).mean(0)[:, 0]
    e_up = _unit((o - centre).mean(0))
    e_1 = _unit(np.cross([.1, .2, .3], e_up))
    e_2 = _unit(np.cross(e_up, e_1))
    world = np.stack([e_1, e_2, e_up, centre], 1)
    moved = np.linalg.inv(_homog(world)) @ _homog(poses[:, :3, :4])
    scale = 1. / np.sqrt(np.mean(np.sum(np.square(moved[:, :3, 3]), -1)))
    moved[:, :3, 3] *= scale
    bds *= scale                                                   # in place, as the reference does
    height = moved[:, 2, 3].mean()
    ring = np.sqrt(1. - height ** 2)                               # mean radius is 1 after scaling
    th = np.linspace(0., 2. * np.pi, 120)
    eye = np.stack([ring * np.cos(th), ring * np.sin(th), np.full_like(th, height)], -1)
    zc = _unit(eye)
    xc = _unit(np.cross(zc, np.array([0, 0, -1.])))
    yc = _unit(np.cross(zc, xc))
    path = _with_hwf(np.stack([xc, yc, zc, eye], -1), hwf)
    return _with_hwf(moved[:, :3, :4], hwf), path, bds


# ------------------------------------------------------------------------------------------ scene assembly
def _llff_to_nerf_axes(poses):
    """LLFF stores camera axes as [down, right, backwards]; NeRF wants [right, up, backwards] (load_llff.py:354-356).  [3,5,N] -> [N,3,5]"""
    return np.moveaxis(np.concatenate([poses[:, 1:2], -poses[:, 0:1], poses[:, 2:]], 1), -1, 0).astype(np.float32)


def _forward_facing_path(poses, bds, path_zflat, n_views=120):
    """The reference's spiral for forward-facing captures (load_llff.py:376-407): focus at the 1/4-3/4 mix of the near and far disparities,
    radii = 90th percentile of the camera offsets, two turns (one flat half-length turn with ``path_zflat``)."""
    mean_cam = poses_avg(poses)
    up = _unit(poses[:, :3, 1].sum(0))
    near, far = bds.min() * .9, bds.max() * 5.
    focus = 1. / (.25 / near + .75 / far)
    radii = np.percentile(np.abs(poses[:, :3, 3]), 90, 0)
    turns = 2
    if path_zflat:
        mean_cam[:3, 3] += -near * .1 * mean_cam[:3, 2]
        radii[2] = 0.
        turns, n_views = 1, n_views / 2
    return render_path_spiral(mean_cam, up, radii, focus, near * .2, zrate=.5, rots=turns, N=n_views)


def _assemble(basedir, factor, recenter, bd_factor, spherify, path_zflat):
    raw_poses, raw_bds, imgs = _load_data(basedir, factor=factor)
    poses = _llff_to_nerf_axes(raw_poses)
    images = np.moveaxis(imgs, -1, 0).astype(np.float32)
    bds = np.moveaxis(raw_bds, -1, 0).astype(np.float32)
    if bd_factor is not None:                                               # nearest bound -> 1/bd_factor
        rescale = 1. / (bds.min() * bd_factor)
        poses[:, :3, 3] *= rescale
        bds *= rescale
    if recenter:
        poses = recenter_poses(poses)
    if spherify:
        poses, render_poses, bds = spherify_poses(poses, bds)
    else:
        render_poses = _forward_facing_path(poses, bds, path_zflat)
    render_poses = np.asarray(render_poses, dtype=np.float32)
    dist2 = np.sum(np.square(poses_avg(poses)[:3, 3] - poses[:, :3, 3]), -1)
    return images, poses.astype(np.float32), bds, render_poses, np.argmin(dist2)   # hold-out = view closest to the average pose


def load_llff_data(basedir, factor=8, recenter=True, bd_factor=.75, spherify=False, path_zflat=False):
    """-> images [N,H,W,3], poses [N,3,5], bds [N,2], render_poses [120,3,5], i_test (load_llff.py:349-421)."""
    return _assemble(basedir, factor, recenter, bd_factor, spherify, path_zflat)


def select_reference_views(basedir, i_train, num_neighbor=None):
    """Greedy set cover of the COLMAP points by training views (load_llff.py:496-544): repeatedly take the view that sees
    the most not-yet-covered points.  Returns indices into ``i_train`` in pick order."""
    imdata = read_images_binary(os.path.join(basedir, 'sparse/0/images.bin'))
    by_name = sorted(imdata.values(), key=lambda im: im.name)            # file order == image index
    index_of = {im.id: i for i, im in enumerate(by_name)}
    train_pos = {int(v): k for k, v in enumerate(i_train)}
    pts = read_points3d_binary(os.path.join(basedir, 'sparse/0/points3D.bin'))
    vis = np.zeros((len(i_train), len(pts)), dtype=bool)
    for col, p in enumerate(pts.values()):
        rows = [train_pos[index_of[int(j)]] for j in p.image_ids if index_of[int(j)] in train_pos]
        vis[rows, col] = True
    picks = []
    n_pick = len(i_train) if num_neighbor is None else int(num_neighbor)
    for _ in range(n_pick):
        total = vis.sum(-1)
        best = int(np.argmax(total))
        if total[best] <= 0:
            if num_neighbor is None:                                       # everything covered: keep the rest in index order
                picks += [k for k in range(len(i_train)) if k not in picks]
                break
            raise ValueError('reference-view selection: no uncovered COLMAP point left for another view')
        picks.append(best)
        vis &= ~vis[best][None]
    return np.asarray(picks, dtype=np.int64)


def load_llff_data_infer(basedir, factor=8, recenter=True, bd_factor=.75, spherify=False, path_zflat=False, num_neighbor=None,
                         llffhold=8):
    """``load_llff_data`` + every ``llffhold``-th view held out + the greedily selected reference views
    -> images, poses, bds, render_poses, i_test [n_test], i_ref (load_llff.py:423-547)."""
    images, poses, bds, render_poses, _ = _assemble(basedir, factor, recenter, bd_factor, spherify, path_zflat)
    i_test = np.arange(images.shape[0])[::llffhold]
    i_train = np.array([i for i in np.arange(int(images.shape[0])) if i not in i_test])
    i_ref = i_train[select_reference_views(basedir, i_train, num_neighbor)]
    return images, poses, bds, render_poses, i_test, i_ref
