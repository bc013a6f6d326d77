"""COLMAP sparse-model readers used by the LLFF loader's reference-view selection.

Same names and record fields as the reference's ``colmap_utils.py`` (``read_images_binary`` :168-201,
``read_points3d_binary`` :230-258, ``qvec2rotmat`` :272-283) so ``load_llff_data_infer`` reads the same
``sparse/0/{images,points3D}.bin`` files.  Parsing is done on one ``bytes`` object with ``struct.unpack_from`` /
``numpy.frombuffer`` (the track arrays in one call each) instead of a file read per field.

File formats (COLMAP ``src/base/reconstruction.cc``, little endian):
  images.bin   : u64 n | n x { i32 image_id, f64 qvec[4], f64 tvec[3], i32 camera_id, char name[] '\\0',
                               u64 n2d, n2d x { f64 x, f64 y, i64 point3D_id } }
  points3D.bin : u64 n | n x { u64 point3D_id, f64 xyz[3], u8 rgb[3], f64 error, u64 track_len,
                               track_len x { i32 image_id, i32 point2D_idx } }
"""
from __future__ import annotations

import collections
import struct

import numpy as np

Image = collections.namedtuple('Image', ['id', 'qvec', 'tvec', 'camera_id', 'name', 'xys', 'point3D_ids'])
Point3D = collections.namedtuple('Point3D', ['id', 'xyz', 'rgb', 'error', 'image_ids', 'point2D_idxs'])

_P2D = np.dtype([('x', '<f8'), ('y', '<f8'), ('id', '<i8')])
_TRK = np.dtype([('image_id', '<i4'), ('point2D_idx', '<i4')])


def read_images_binary(path_to_model_file):
    """dict image_id -> Image, in file order."""
    buf = open(path_to_model_file, 'rb').read()
    (n,) = struct.unpack_from('<Q', buf, 0)
    off = 8
    images = {}
    for _ in range(n):
        image_id, q0, q1, q2, q3, t0, t1, t2, camera_id = struct.unpack_from('<idddddddi', buf, off)
        off += 64
        end = buf.index(b'\x00', off)
        name = buf[off:end].decode('utf-8')
        off = end + 1
        (n2d,) = struct.unpack_from('<Q', buf, off)
        off += 8
        rec = np.frombuffer(buf, dtype=_P2D, count=n2d, offset=off)
        off += 24 * n2d
        images[image_id] = Image(id=image_id, qvec=np.array([q0, q1, q2, q3]), tvec=np.array([t0, t1, t2]), camera_id=camera_id,
                                 name=name, xys=np.column_stack([rec['x'], rec['y']]).astype(np.float64),
                                 point3D_ids=rec['id'].astype(np.int64))
    return images


def read_points3d_binary(path_to_model_file):
    """dict point3D_id -> Point3D, in file order."""
    buf = open(path_to_model_file, 'rb').read()
    (n,) = struct.unpack_from('<Q', buf, 0)
    off = 8
    points = {}
    for _ in range(n):
        pid, x, y, z, r, g, b, err = struct.unpack_from('<QdddBBBd', buf, off)
        off += 43
        (tl,) = struct.unpack_from('<Q', buf, off)
        off += 8
        trk = np.frombuffer(buf, dtype=_TRK, count=tl, offset=off)
        off += 8 * tl
        points[pid] = Point3D(id=pid, xyz=np.array([x, y, z]), rgb=np.array([r, g, b]), error=np.array(err),
                              image_ids=trk['image_id'].astype(np.int64), point2D_idxs=trk['point2D_idx'].astype(np.int64))
    return points


def qvec2rotmat(qvec):
    """Rotation matrix of the COLMAP quaternion (w, x, y, z)."""
    w, x, y, z = qvec
    return np.array([
        [1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * w * z, 2 * z * x + 2 * w * y],
        [2 * x * y + 2 * w * z, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * w * x],
        [2 * z * x - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * x * x - 2 * y * y]])


def write_images_binary(path, images):
    """Inverse of ``read_images_binary`` (fixtures, exporting a subset of a model)."""
    with open(path, 'wb') as f:
        f.write(struct.pack('<Q', len(images)))
        for im in images.values():
            f.write(struct.pack('<idddddddi', im.id, *[float(v) for v in im.qvec], *[float(v) for v in im.tvec], im.camera_id))
            f.write(im.name.encode('utf-8') + b'\x00')
            f.write(struct.pack('<Q', len(im.point3D_ids)))
            rec = np.empty(len(im.point3D_ids), dtype=_P2D)
            rec['x'], rec['y'], rec['id'] = im.xys[:, 0], im.xys[:, 1], im.point3D_ids
            f.write(rec.tobytes())


def write_points3d_binary(path, points):
    """Inverse of ``read_points3d_binary``."""
    with open(path, 'wb') as f:
        f.write(struct.pack('<Q', len(points)))
        for p in points.values():
            f.write(struct.pack('<QdddBBBd', p.id, *[float(v) for v in p.xyz], *[int(v) for v in p.rgb], float(p.error)))
            f.write(struct.pack('<Q', len(p.image_ids)))
            rec = np.empty(len(p.image_ids), dtype=_TRK)
            rec['image_id'], rec['point2D_idx'] = p.image_ids, p.point2D_idxs
            f.write(rec.tobytes())
