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
def normalize(x):
    return x / np.linalg.norm(x)


def viewmatrix(z, up, pos):
    vec2 = normalize(z)
    vec0 = normalize(np.cross(up, vec2))
    vec1 = normalize(np.cross(vec2, vec0))
    return np.stack([vec0, vec1, vec2, pos], 1)


def poses_avg(poses):
    """Average camera [3,5]: mean centre, summed z axis, summed y axis as up (load_llff.py:163-172)."""
    hwf = poses[0, :3, -1:]
    center = poses[:, :3, 3].mean(0)
    vec2 = normalize(poses[:, :3, 2].sum(0))
    up = poses[:, :3, 1].sum(0)
    return np.concatenate([viewmatrix(vec2, up, center), hwf], 1)


def recenter_poses(poses):
    """Express all poses in the frame of the average camera (load_llff.py:189-203)."""
    out = poses + 0
    bottom = np.reshape([0, 0, 0, 1.], [1, 4])
    c2w = np.concatenate([poses_avg(poses)[:3, :4], bottom], -2)
    p44 = np.concatenate([poses[:, :3, :4], np.tile(bottom[None], [poses.shape[0], 1, 1])], -2)
    p44 = np.linalg.inv(c2w) @ p44
    out[:, :3, :4] = p44[:, :3, :4]
    return out


def render_path_spiral(c2w, up, rads, focal, zdelta, zrate, rots, N):
    """Spiral of N cameras around the average pose looking at depth ``focal`` (load_llff.py:176-185)."""
    out = []
    rads = np.array(list(rads) + [1.])
    hwf = c2w[:, 4:5]
    for theta in np.linspace(0., 2. * np.pi * rots, int(N) + 1)[:-1]:
        c = np.dot(c2w[:3, :4], np.array([np.cos(theta), -np.sin(theta), -np.sin(theta * zrate), 1.]) * rads)
        z = normalize(c - np.dot(c2w[:3, :4], np.array([0, 0, -focal, 1.])))
        out.append(np.concatenate([viewmatrix(z, up, c), hwf], 1))
    return out


def spherify_poses(poses, bds):
    """360-degree captures: recentre on the point closest to all optical axes, unit mean radius, circular path
    (load_llff.py:207-262)."""
    def p44(p):
        return np.concatenate([p, np.tile(np.reshape(np.eye(4)[-1, :], [1, 1, 4]), [p.shape[0], 1, 1])], 1)

    rays_d, rays_o = poses[:, :3, 2:3], poses[:, :3, 3:4]
    A = np.eye(3) - rays_d * np.transpose(rays_d, [0, 2, 1])
    b = -A @ rays_o
    center = np.squeeze(-np.linalg.inv((np.transpose(A, [0, 2, 1]) @ A).mean(0)) @ b.mean(0))
    up = (poses[:, :3, 3] - center).mean(0)
    vec0 = normalize(up)
    vec1 = normalize(np.cross([.1, .2, .3], vec0))
    vec2 = normalize(np.cross(vec0, vec1))
    c2w = np.stack([vec1, vec2, vec0, center], 1)
    reset = np.linalg.inv(p44(c2w[None])) @ p44(poses[:, :3, :4])
    rad = np.sqrt(np.mean(np.sum(np.square(reset[:, :3, 3]), -1)))
    sc = 1. / rad
    reset[:, :3, 3] *= sc
    bds *= sc
    rad *= sc
    zh = np.mean(reset[:, :3, 3], 0)[2]
    radcircle = np.sqrt(rad ** 2 - zh ** 2)
    path = []
    for th in np.linspace(0., 2. * np.pi, 120):
        cam = np.array([radcircle * np.cos(th), radcircle * np.sin(th), zh])
        v2 = normalize(cam)
        v0 = normalize(np.cross(v2, np.array([0, 0, -1.])))
        v1 = normalize(np.cross(v2, v0))
        path.append(np.stack([v0, v1, v2, cam], 1))
    path = np.stack(path, 0)
    path = np.concatenate([path, np.broadcast_to(poses[0, :3, -1:], path[:, :3, -1:].shape)], -1)
    reset = np.concatenate([reset[:, :3, :4], np.broadcast_to(poses[0, :3, -1:], reset[:, :3, -1:].shape)], -1)
    return reset, path, bds


# ------------------------------------------------------------------------------------------ scene assembly
def _assemble(basedir, factor, recenter, bd_factor, spherify, path_zflat):
    poses, bds, imgs = _load_data(basedir, factor=factor)
    # LLFF stores [down, right, backwards]; NeRF wants [right, up, backwards] (load_llff.py:354-356)
    poses = np.concatenate([poses[:, 1:2, :], -poses[:, 0:1, :], poses[:, 2:, :]], 1)
    poses = np.moveaxis(poses, -1, 0).astype(np.float32)
    images = np.moveaxis(imgs, -1, 0).astype(np.float32)
    bds = np.moveaxis(bds, -1, 0).astype(np.float32)
    sc = 1. if bd_factor is None else 1. / (bds.min() * bd_factor)          # nearest bound -> 1/bd_factor
    poses[:, :3, 3] *= sc
    bds *= sc
    if recenter:
        poses = recenter_poses(poses)
    if spherify:
        poses, render_poses, bds = spherify_poses(poses, bds)
    else:
        c2w = poses_avg(poses)
        up = normalize(poses[:, :3, 1].sum(0))
        close_depth, inf_depth = bds.min() * .9, bds.max() * 5.
        dt = .75
        focal = 1. / ((1. - dt) / close_depth + dt / inf_depth)            # focus depth of the spiral
        zdelta = close_depth * .2
        rads = np.percentile(np.abs(poses[:, :3, 3]), 90, 0)
        n_views, n_rots = 120, 2
        if path_zflat:
            c2w[:3, 3] = c2w[:3, 3] + (-close_depth * .1) * c2w[:3, 2]
            rads[2] = 0.
            n_rots, n_views = 1, n_views / 2
        render_poses = render_path_spiral(c2w, up, rads, focal, zdelta, zrate=.5, rots=n_rots, N=n_views)
    render_poses = np.array(render_poses).astype(np.float32)
    c2w = poses_avg(poses)
    i_test = np.argmin(np.sum(np.square(c2w[:3, 3] - poses[:, :3, 3]), -1))   # view closest to the average pose
    return images.astype(np.float32), poses.astype(np.float32), bds, render_poses, i_test


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
