"""COLMAP sparse-model files without ``pycolmap``: ``cameras.bin`` / ``images.bin`` / ``points3D.bin`` (and the ``.txt`` forms).

Upstream's CLI entry point reads the scene with ``pycolmap.Reconstruction(sparse/0)`` (densify.py:54-56) and only ever touches
    rec.cameras[id].model.name / .params / .width / .height                     (core/geometry.py:10-30)
    rec.images[id].camera_id / .name / .cam_from_world.rotation.matrix() / .translation   (core/geometry.py:33-42, densify.py:59-88)
    rec.points3D (truthiness) and images[id].points2D[i].has_point3D() / .point3D_id       (core/selection.py:10-33)
so that is the surface the small classes below offer; ``densify.load_reconstruction`` uses them when ``pycolmap`` is not installed
(it is absent from the ROCm image).  The binary layouts are COLMAP's documented ones (little endian):

    cameras.bin   u64 n | per camera: i32 id, i32 model_id, u64 width, u64 height, f64 params[n_params(model)]
    images.bin    u64 n | per image:  i32 id, f64 qvec[4] (w x y z), f64 tvec[3], i32 camera_id, name\\0, u64 n2D, n2D x (f64 x, f64 y, i64 point3D_id)
    points3D.bin  u64 n | per point:  u64 id, f64 xyz[3], u8 rgb[3], f64 error, u64 track, track x (i32 image_id, i32 point2D_idx)

The writers exist for fixtures (tests/golden/make_colmap_fixture.py) and round-trip tests.
"""
from __future__ import annotations

import os
import struct
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

# model id -> (name, number of parameters): COLMAP's camera models
CAMERA_MODELS = {0: ("SIMPLE_PINHOLE", 3), 1: ("PINHOLE", 4), 2: ("SIMPLE_RADIAL", 4), 3: ("RADIAL", 5), 4: ("OPENCV", 8),
                 5: ("OPENCV_FISHEYE", 8), 6: ("FULL_OPENCV", 12), 7: ("FOV", 5), 8: ("SIMPLE_RADIAL_FISHEYE", 4), 9: ("RADIAL_FISHEYE", 5),
                 10: ("THIN_PRISM_FISHEYE", 12)}
_MODEL_IDS = {name: mid for mid, (name, _n) in CAMERA_MODELS.items()}


class _Model:
    def __init__(self, name: str):
        self.name = name


class Camera:
    def __init__(self, camera_id: int, model: str, width: int, height: int, params: Sequence[float]):
        self.camera_id, self.model, self.width, self.height = int(camera_id), _Model(model), int(width), int(height)
        self.params = np.asarray(params, np.float64)


class _Rotation:
    def __init__(self, qvec):
        self.quat = np.asarray(qvec, np.float64)          # w x y z

    def matrix(self) -> np.ndarray:
        """3x3 f64 rotation of the NORMALISED quaternion (what Eigen's toRotationMatrix gives pycolmap)."""
        w, x, y, z = self.quat / np.linalg.norm(self.quat)
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                         [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                         [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]], np.float64)


class _Rigid3d:
    def __init__(self, qvec, tvec):
        self.rotation = _Rotation(qvec)
        self.translation = np.asarray(tvec, np.float64)


class Point2D:
    __slots__ = ("xy", "point3D_id")

    def __init__(self, x: float, y: float, point3D_id: int):
        self.xy, self.point3D_id = (float(x), float(y)), int(point3D_id)

    def has_point3D(self) -> bool:
        return self.point3D_id != -1 and self.point3D_id != 0xFFFFFFFFFFFFFFFF


class Image:
    def __init__(self, image_id: int, qvec, tvec, camera_id: int, name: str, xys: Optional[np.ndarray] = None, point3D_ids: Optional[np.ndarray] = None):
        self.image_id, self.camera_id, self.name = int(image_id), int(camera_id), str(name)
        self.cam_from_world = _Rigid3d(qvec, tvec)
        self._xys = np.zeros((0, 2)) if xys is None else np.asarray(xys, np.float64).reshape(-1, 2)
        self._p3d = np.zeros((0,), np.int64) if point3D_ids is None else np.asarray(point3D_ids, np.int64).reshape(-1)

    @property
    def points2D(self) -> List[Point2D]:
        return [Point2D(x, y, p) for (x, y), p in zip(self._xys, self._p3d)]

    def observed_point3D_ids(self) -> np.ndarray:
        """The ids ``[p.point3D_id for p in points2D if p.has_point3D() and p.point3D_id != -1]`` as one array (what the visibility-based
        reference selection asks of every image, core/selection.py: millions of observations in a real scene - not one Python object each)."""
        p = self._p3d
        return p[(p != -1) & (p.astype(np.uint64) != np.uint64(0xFFFFFFFFFFFFFFFF))]


class Reconstruction:
    """``cameras`` / ``images`` / ``points3D`` dictionaries of a sparse model directory (binary files preferred, text otherwise)."""

    def __init__(self, sparse_dir: str):
        self.cameras: Dict[int, Camera] = {}
        self.images: Dict[int, Image] = {}
        self.points3D: Dict[int, Tuple[np.ndarray, np.ndarray, float]] = {}
        d = str(sparse_dir)
        if os.path.isfile(os.path.join(d, "cameras.bin")) and os.path.isfile(os.path.join(d, "images.bin")):
            self.cameras = read_cameras_bin(os.path.join(d, "cameras.bin"))
            self.images = read_images_bin(os.path.join(d, "images.bin"))
            if os.path.isfile(os.path.join(d, "points3D.bin")):
                self.points3D = read_points3D_bin(os.path.join(d, "points3D.bin"))
        elif os.path.isfile(os.path.join(d, "cameras.txt")) and os.path.isfile(os.path.join(d, "images.txt")):
            self.cameras = read_cameras_txt(os.path.join(d, "cameras.txt"))
            self.images = read_images_txt(os.path.join(d, "images.txt"))
            if os.path.isfile(os.path.join(d, "points3D.txt")):
                self.points3D = read_points3D_txt(os.path.join(d, "points3D.txt"))
        else:
            raise FileNotFoundError(f"no COLMAP model (cameras/images .bin or .txt) under {d}")


# ---- binary ------------------------------------------------------------------------------------------------------------------------------
def _unpack(fh, fmt: str):
    size = struct.calcsize(fmt)
    data = fh.read(size)
    if len(data) != size:
        raise ValueError("truncated COLMAP file")
    return struct.unpack(fmt, data)


def read_cameras_bin(path: str) -> Dict[int, Camera]:
    out: Dict[int, Camera] = {}
    with open(path, "rb") as fh:
        (n,) = _unpack(fh, "<Q")
        for _ in range(n):
            cid, mid, w, h = _unpack(fh, "<iiQQ")
            if mid not in CAMERA_MODELS:
                raise ValueError(f"unknown COLMAP camera model id {mid}")
            name, npar = CAMERA_MODELS[mid]
            out[cid] = Camera(cid, name, w, h, _unpack(fh, f"<{npar}d"))
    return out


def read_images_bin(path: str) -> Dict[int, Image]:
    out: Dict[int, Image] = {}
    with open(path, "rb") as fh:
        (n,) = _unpack(fh, "<Q")
        for _ in range(n):
            iid, qw, qx, qy, qz, tx, ty, tz, cid = _unpack(fh, "<i7di")
            name = bytearray()
            while True:
                c = fh.read(1)
                if not c:
                    raise ValueError("truncated COLMAP file")
                if c == b"\0":
                    break
                name += c
            (n2d,) = _unpack(fh, "<Q")
            rec = np.frombuffer(fh.read(24 * n2d), dtype=np.dtype([("xy", "<f8", 2), ("p", "<i8")]))
            if rec.shape[0] != n2d:
                raise ValueError("truncated COLMAP file")
            out[iid] = Image(iid, (qw, qx, qy, qz), (tx, ty, tz), cid, name.decode("utf-8"), rec["xy"].copy(), rec["p"].copy())
    return out


def read_points3D_bin(path: str) -> Dict[int, Tuple[np.ndarray, np.ndarray, float]]:
    out: Dict[int, Tuple[np.ndarray, np.ndarray, float]] = {}
    with open(path, "rb") as fh:
        (n,) = _unpack(fh, "<Q")
        for _ in range(n):
            pid, x, y, z, r, g, b, err, track = _unpack(fh, "<Q3d3BdQ")
            fh.seek(8 * track, os.SEEK_CUR)
            out[pid] = (np.array([x, y, z]), np.array([r, g, b], np.uint8), float(err))
    return out


def write_cameras_bin(path: str, cameras: Sequence[Camera]) -> None:
    with open(path, "wb") as fh:
        fh.write(struct.pack("<Q", len(cameras)))
        for c in cameras:
            mid = _MODEL_IDS[c.model.name]
            assert len(c.params) == CAMERA_MODELS[mid][1]
            fh.write(struct.pack("<iiQQ", c.camera_id, mid, c.width, c.height))
            fh.write(struct.pack(f"<{len(c.params)}d", *[float(v) for v in c.params]))


def write_images_bin(path: str, images: Sequence[Image]) -> None:
    with open(path, "wb") as fh:
        fh.write(struct.pack("<Q", len(images)))
        for im in images:
            q, t = im.cam_from_world.rotation.quat, im.cam_from_world.translation
            fh.write(struct.pack("<i7di", im.image_id, *[float(v) for v in q], *[float(v) for v in t], im.camera_id))
            fh.write(im.name.encode("utf-8") + b"\0")
            fh.write(struct.pack("<Q", im._xys.shape[0]))
            rec = np.zeros(im._xys.shape[0], dtype=np.dtype([("xy", "<f8", 2), ("p", "<i8")]))
            rec["xy"], rec["p"] = im._xys, im._p3d
            fh.write(rec.tobytes())


def write_colmap_points3D_bin(path: str, points: Dict[int, Tuple[np.ndarray, np.ndarray, float]], tracks: Optional[Dict[int, List[Tuple[int, int]]]] = None) -> None:
    with open(path, "wb") as fh:
        fh.write(struct.pack("<Q", len(points)))
        for pid, (xyz, rgb, err) in points.items():
            tr = (tracks or {}).get(pid, [])
            fh.write(struct.pack("<Q3d3BdQ", pid, *[float(v) for v in xyz], *[int(v) for v in rgb], float(err), len(tr)))
            for iid, idx in tr:
                fh.write(struct.pack("<ii", iid, idx))


# ---- text --------------------------------------------------------------------------------------------------------------------------------
def _lines(path: str):
    with open(path, "r") as fh:
        for line in fh:
            line = line.strip()
            if line and not line.startswith("#"):
                yield line


def read_cameras_txt(path: str) -> Dict[int, Camera]:
    out = {}
    for line in _lines(path):
        f = line.split()
        out[int(f[0])] = Camera(int(f[0]), f[1], int(f[2]), int(f[3]), [float(v) for v in f[4:]])
    return out


def read_images_txt(path: str) -> Dict[int, Image]:
    out = {}
    with open(path, "r") as fh:
        rows = [l.rstrip("\n") for l in fh if not l.startswith("#")]
    rows = [r for r in rows if r.strip() or True]
    i = 0
    while i < len(rows):
        if not rows[i].strip():
            i += 1
            continue
        f = rows[i].split()
        pts = rows[i + 1].split() if i + 1 < len(rows) else []
        xys = np.array([[float(pts[j]), float(pts[j + 1])] for j in range(0, len(pts) - 2, 3)], np.float64).reshape(-1, 2)
        p3d = np.array([int(pts[j + 2]) for j in range(0, len(pts) - 2, 3)], np.int64)
        out[int(f[0])] = Image(int(f[0]), [float(v) for v in f[1:5]], [float(v) for v in f[5:8]], int(f[8]), " ".join(f[9:]), xys, p3d)
        i += 2
    return out


def read_points3D_txt(path: str):
    out = {}
    for line in _lines(path):
        f = line.split()
        out[int(f[0])] = (np.array([float(v) for v in f[1:4]]), np.array([int(v) for v in f[4:7]], np.uint8), float(f[7]))
    return out


def quaternion_from_rotation(R) -> np.ndarray:
    """(w, x, y, z) of a rotation matrix (fixtures): the branch with the largest diagonal term."""
    R = np.asarray(R, np.float64)
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = [0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s]
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = np.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        q = [(R[2, 1] - R[1, 2]) / s, 0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s]
    elif R[1, 1] > R[2, 2]:
        s = np.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        q = [(R[0, 2] - R[2, 0]) / s, (R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s]
    else:
        s = np.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        q = [(R[1, 0] - R[0, 1]) / s, (R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s]
    q = np.asarray(q, np.float64)
    return q / np.linalg.norm(q)
