"""Synthetic scenes for tests, smoke() and bench.py.

Real MipNeRF360 images, COLMAP models and RoMa-v2 weights are not available offline, and the hot
path is per-correspondence arithmetic, so its inputs are generated analytically: pinhole cameras on a
ring (the `garden` capture pattern: 185 views circling a table), a ground plane with smooth relief,
and for every (reference, neighbour) pair the exact dense warp of the reference grid into the
neighbour plus pixel noise / gross outliers, together with a spatially coherent certainty field of
the kind RoMa's overlap head produces.  Everything is torch so the same code fills HBM directly on
the GPU for the benchmark and produces small CPU arrays for fixtures.
"""
from __future__ import annotations

import dataclasses
import math
from typing import List, Optional, Sequence

import numpy as np
import torch

from .core.types import CameraRecord

# (H_lr, W_lr, H_out, W_out) per RoMa preset: upstream core/matcher.py:82-94 and
# RoMaV2/src/romav2/romav2.py:118-160.  The output grid is the high-res size when one is set.
ROMA_PRESETS = {
    "turbo": (320, 320, 320, 320),
    "fast": (512, 512, 512, 512),
    "base": (640, 640, 640, 640),
    "high": (640, 640, 960, 960),
    "precise": (800, 800, 1280, 1280),
}


def ring_cameras(n: int, width: int = 1297, height: int = 840, focal: float = 960.0,
                 radius: float = 4.0, cam_height: float = 2.0, target_height: float = 0.0,
                 wobble: float = 0.05, seed: int = 0, arc: float = 2.0 * math.pi) -> List[CameraRecord]:
    """``n`` pinhole cameras on a circle of ``radius`` at ``cam_height`` looking at the origin.
    ``arc`` < 2*pi places them on a partial arc (small baselines for a few-camera ROI)."""
    rng = np.random.RandomState(seed)
    cams: List[CameraRecord] = []
    K = np.array([[focal, 0.0, width / 2.0 + 3.25], [0.0, focal * 1.003, height / 2.0 - 1.75], [0.0, 0.0, 1.0]])
    for i in range(n):
        th = arc * i / n
        r = radius * (1.0 + wobble * rng.uniform(-1, 1))
        c = np.array([r * math.cos(th), r * math.sin(th), cam_height * (1.0 + wobble * rng.uniform(-1, 1))])
        fwd = np.array([0.0, 0.0, target_height]) - c
        fwd /= np.linalg.norm(fwd)
        right = np.cross(fwd, np.array([0.0, 0.0, 1.0]))
        right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        R = np.stack([right, down, fwd], axis=0)
        t = -R @ c
        cams.append(CameraRecord.from_krt(i + 1, K, R, t, width, height, image_path=f"synthetic/{i:04d}.png"))
    return cams


def identity_axis_torch(n: int, device) -> torch.Tensor:
    """The A-grid axis exactly as the matcher builds it (upstream core/matcher.py:132-133)."""
    return torch.linspace(-1 + 1 / n, 1 - 1 / n, n, device=device)


@dataclasses.dataclass
class SyntheticReference:
    """Hot-path inputs of one reference view with its k neighbours (all tensors on one device)."""
    ref_index: int
    nbr_indices: List[int]
    warp: torch.Tensor        # (k,H,W,C) f32, C=2: [xB,yB]; C=4: [xA,yA,xB,yB], normalised [-1,1]
    cert: torch.Tensor        # (k,H,W) f32 in [0,1]
    image: torch.Tensor       # (h_match,w_match,3) u8
    w_match: int
    h_match: int


def _cam_t(cam: CameraRecord, device):
    f64 = dict(dtype=torch.float64, device=device)
    return (torch.as_tensor(np.asarray(cam.K, np.float64), **f64), torch.as_tensor(np.asarray(cam.R, np.float64), **f64),
            torch.as_tensor(np.asarray(cam.t, np.float64).reshape(3), **f64),
            torch.as_tensor(np.asarray(cam.C, np.float64), **f64))


def synth_image(h: int, w: int, seed: int, device="cpu") -> torch.Tensor:
    g = torch.Generator(device="cpu").manual_seed(int(seed) + 977)
    yy = torch.linspace(0, 1, h).view(h, 1).expand(h, w)
    xx = torch.linspace(0, 1, w).view(1, w).expand(h, w)
    base = torch.stack([0.5 + 0.4 * torch.sin(6.0 * xx + 2.0 * yy), 0.5 + 0.4 * torch.cos(5.0 * yy - 3.0 * xx),
                        0.5 + 0.4 * torch.sin(9.0 * xx * yy + 1.0)], dim=-1)
    tex = torch.rand((h, w, 3), generator=g) * 0.2 - 0.1
    return ((base + tex).clamp(0, 1) * 255.0).round().to(torch.uint8).to(device)


def synth_reference(cams: Sequence[CameraRecord], ref_index: int, nbr_indices: Sequence[int],
                    H: int, W: int, w_match: int, h_match: int, *, noise_px: float = 0.3,
                    outlier_frac: float = 0.0, channels: int = 2, seed: int = 0,
                    cert_mode: str = "smooth", device="cpu", far_depth: float = 25.0,
                    low_parallax_patch: Optional[Sequence[float]] = None,
                    patch_depths: Sequence[float] = (6.0, 60.0), out_of_range: float = 0.0,
                    occlusion_steps: bool = False) -> SyntheticReference:
    """Dense warp + certainty of ``ref_index`` into each neighbour.

    low_parallax_patch: (y0, y1, x0, x1) as fractions of the grid.  Inside it the surface is replaced by a ramp of
    depths ``patch_depths[0] .. patch_depths[1]`` along x (distance from the reference camera), so that the angle
    between the two viewing rays falls through the min-parallax threshold (0.5 degrees by default) inside the patch:
    SURVEY 8d config 3 asks for "a patch with sub-0.5 degree parallax" to exercise that reject reason.

    cert_mode: "smooth"  - low-frequency overlap-like field + 2 % jitter (coherent best-neighbour regions)
               "tiefree" - distinct uniform values in (0.2, 0.9) (no ties under floor/cap clamps)
               "beta"    - iid Beta(2,2) (floor and cap clamps produce massive ties, like real data edges)
               "bimodal" - what ``sigmoid(confidence logits)`` gives (RoMaV2's overlap head): most cells near 0.02 or near 0.98, coherent regions
                           of each, a thin transition - after the 0.2 floor and the 0.9 cap nearly EVERY cell sits on a clamp

    out_of_range: that fraction of the grid's columns (on its right) and, at half that width, of its rows (at the top) is warped past the edge
               of the neighbour - x (y) beyond [-1, 1], through the border and on to 2.5 image half-widths outside - as the real matcher does for
               content the neighbour does not see; upstream does not reject such coordinates (core/pipeline.py:697-703), the reprojection test does.
    occlusion_steps: the surface gets depth discontinuities (a checker of foreground slabs at 0.65 of the depth): the warp jumps at their edges.
    """
    dev = torch.device(device)
    gen = torch.Generator(device="cpu").manual_seed(int(seed) * 7919 + int(ref_index) * 104729 + 13)
    camA = cams[ref_index]
    KA, RA, tA, CA = _cam_t(camA, dev)
    ax = identity_axis_torch(W, dev).to(torch.float64)
    ay = identity_axis_torch(H, dev).to(torch.float64)
    xA = (ax + 1.0) * 0.5 * (w_match - 1)
    yA = (ay + 1.0) * 0.5 * (h_match - 1)
    uA = (xA * (camA.width / float(w_match))).view(1, W).expand(H, W)
    vA = (yA * (camA.height / float(h_match))).view(H, 1).expand(H, W)
    # back-project onto the ground plane z = relief(x, y)
    dx = (uA - KA[0, 2]) / KA[0, 0]
    dy = (vA - KA[1, 2]) / KA[1, 1]
    dirs_c = torch.stack([dx, dy, torch.ones_like(dx)], dim=-1)           # camera frame
    dirs_w = dirs_c @ RA                                                    # R^T d  (row-vector form)
    s_plane = -CA[2] / dirs_w[..., 2].clamp(max=-1e-6)
    hits = (dirs_w[..., 2] < -1e-6) & (s_plane < far_depth)
    s = torch.where(hits, s_plane, torch.full_like(s_plane, far_depth))
    Xw = CA.view(1, 1, 3) + s.unsqueeze(-1) * dirs_w
    relief = 0.08 * torch.sin(2.1 * Xw[..., 0] + 0.3) * torch.cos(1.7 * Xw[..., 1] - 0.2)
    s = s * (1.0 - relief / CA[2].clamp(min=0.5))
    if occlusion_steps:
        gy_ = torch.linspace(0, 1, H, dtype=torch.float64, device=dev).view(H, 1)
        gx_ = torch.linspace(0, 1, W, dtype=torch.float64, device=dev).view(1, W)
        slab = (torch.floor(5.0 * gx_ + 0.8 * gy_) + torch.floor(3.0 * gy_ - 0.4 * gx_)).to(torch.int64) % 2 == 1
        s = torch.where(slab, s * 0.65, s)
    if low_parallax_patch is not None:
        fy0, fy1, fx0, fx1 = [float(v) for v in low_parallax_patch]
        iy0, iy1, ix0, ix1 = int(fy0 * H), max(int(fy1 * H), int(fy0 * H) + 1), int(fx0 * W), max(int(fx1 * W), int(fx0 * W) + 1)
        ramp = torch.linspace(float(patch_depths[0]), float(patch_depths[1]), ix1 - ix0, dtype=torch.float64, device=dev)
        s = s.clone()
        s[iy0:iy1, ix0:ix1] = ramp.view(1, -1).expand(iy1 - iy0, ix1 - ix0) / dirs_w[iy0:iy1, ix0:ix1].norm(dim=-1)
    Xw = CA.view(1, 1, 3) + s.unsqueeze(-1) * dirs_w

    warps, certs = [], []
    for slot, nb in enumerate(nbr_indices):
        camB = cams[nb]
        KB, RB, tB, _ = _cam_t(camB, dev)
        Xc = Xw @ RB.T + tB.view(1, 1, 3)
        z = Xc[..., 2]
        uB = KB[0, 0] * Xc[..., 0] / z + KB[0, 2]
        vB = KB[1, 1] * Xc[..., 1] / z + KB[1, 2]
        if noise_px > 0:      # matching noise, in full-resolution camera pixels of the neighbour
            nz = torch.randn((2, H, W), generator=gen, dtype=torch.float64).to(dev) * noise_px
            uB = uB + nz[0]
            vB = vB + nz[1]
        xB = uB / (camB.width / float(w_match))
        yB = vB / (camB.height / float(h_match))
        xBn = xB / (0.5 * (w_match - 1)) - 1.0
        yBn = yB / (0.5 * (h_match - 1)) - 1.0
        if out_of_range > 0:
            gxo = torch.linspace(0, 1, W, dtype=torch.float64, device=dev).view(1, W)
            gyo = torch.linspace(0, 1, H, dtype=torch.float64, device=dev).view(H, 1)
            fx_, fy_ = float(out_of_range), 0.5 * float(out_of_range)
            xBn = xBn + 3.5 * ((gxo - (1.0 - fx_)) / fx_).clamp(min=0.0)           # from where it was, across the border, to well outside
            yBn = yBn - 3.5 * (((fy_ - gyo)) / fy_).clamp(min=0.0)
        if outlier_frac > 0:
            pick = (torch.rand((H, W), generator=gen) < outlier_frac).to(dev)
            rnd = (torch.rand((2, H, W), generator=gen, dtype=torch.float64) * 2.0 - 1.0).to(dev)
            xBn = torch.where(pick, rnd[0], xBn)
            yBn = torch.where(pick, rnd[1], yBn)
        inside = ((xBn.abs() < 1.0) & (yBn.abs() < 1.0) & (z > 0)).to(torch.float64)
        gy = torch.linspace(0, 1, H, dtype=torch.float64, device=dev).view(H, 1)
        gx = torch.linspace(0, 1, W, dtype=torch.float64, device=dev).view(1, W)
        if cert_mode == "smooth":
            ph = 2.399963 * (slot + 1) + 0.37 * ref_index
            field = 0.55 + 0.40 * torch.sin(3.0 * gx + ph) * torch.sin(2.0 * gy + 1.3 * ph)
            jit = (torch.rand((H, W), generator=gen, dtype=torch.float64) * 0.04 - 0.02).to(dev)
            c = (field + jit).clamp(0.0, 1.0) * (0.15 + 0.85 * inside)
        elif cert_mode == "tiefree":
            perm = torch.randperm(H * W, generator=gen).to(dev).to(torch.float64).view(H, W)
            c = 0.2 + 0.7 * (perm + 0.5 + 0.1 * slot) / (H * W)
        elif cert_mode == "beta":
            a = torch.distributions.Beta(2.0, 2.0)
            torch.manual_seed(int(seed) * 31 + slot + 1000 * int(ref_index))
            c = a.sample((H, W)).to(torch.float64).to(dev)
        elif cert_mode == "bimodal":
            ph = 2.399963 * (slot + 1) + 0.37 * ref_index
            logit = 18.0 * torch.sin(3.0 * gx + ph) * torch.sin(2.0 * gy + 1.3 * ph) + 1.5 * torch.randn((H, W), generator=gen, dtype=torch.float64).to(dev)
            c = torch.sigmoid(logit - 6.0 * (1.0 - inside))
        else:
            raise ValueError(cert_mode)
        wp = torch.stack([xBn, yBn], dim=-1)
        if channels == 4:
            ida = torch.stack([identity_axis_torch(W, dev).view(1, W).expand(H, W),
                               identity_axis_torch(H, dev).view(H, 1).expand(H, W)], dim=-1)
            wp = torch.cat([ida, wp.to(torch.float32)], dim=-1)
        warps.append(wp.to(torch.float32))
        certs.append(c.to(torch.float32))
    return SyntheticReference(
        ref_index=int(ref_index), nbr_indices=[int(n) for n in nbr_indices],
        warp=torch.stack(warps, 0).contiguous(), cert=torch.stack(certs, 0).contiguous(),
        image=synth_image(h_match, w_match, seed * 1000 + ref_index, dev), w_match=int(w_match), h_match=int(h_match))


def ring_neighbours(n_cams: int, ref_index: int, k: int) -> List[int]:
    """The k nearest cameras on the ring: +1, -1, +2, -2, ..."""
    out: List[int] = []
    step = 1
    while len(out) < k:
        for sgn in (1, -1):
            if len(out) < k:
                cand = (ref_index + sgn * step) % n_cams
                if cand != ref_index and cand not in out:
                    out.append(cand)
        step += 1
        if step > n_cams:
            break
    return out


# ---- a whole scene on disk + a matcher that knows it (bench.py's pipeline leg, tests/golden/time_reference.py) -------------------------------
def write_colmap_scene(root: str, n_cams: int = 185, width: int = 1297, height: int = 840, images_subdir: str = "images_4", fmt: str = "jpg",
                       seed: int = 0, n_points: int = 600, workers: int = 8) -> List[CameraRecord]:
    """A garden-like scene as upstream's CLI reads it (densify.py:54-88): ``root/sparse/0/{cameras,images,points3D}.bin`` (one PINHOLE camera per
    view, principal point off the centre; ``n_points`` ground points with their tracks, which the visibility-based reference selection needs) and
    ``root/<images_subdir>/view_NNNN.<fmt>`` - ``n_cams`` images of ``width`` x ``height`` (MipNeRF360 garden: 185 views, ``images_4`` 1297 x 840).
    Returns the cameras with their image paths.  Files that exist already are kept (the images take ~10 s to make)."""
    import os
    from concurrent.futures import ThreadPoolExecutor

    from PIL import Image

    from .core import colmap_io as cio
    sparse, img_dir = os.path.join(root, "sparse", "0"), os.path.join(root, images_subdir)
    os.makedirs(sparse, exist_ok=True)
    os.makedirs(img_dir, exist_ok=True)
    cams = ring_cameras(n_cams, width=width, height=height, seed=seed)
    rs = np.random.RandomState(seed + 7)
    pts = {pid: (rs.uniform(-1.0, 1.0, 3) * np.array([1.8, 1.8, 0.15]), rs.randint(0, 255, 3).astype(np.uint8), float(rs.uniform(0.1, 1.0)))
           for pid in range(1, n_points + 1)}
    tracks = {pid: [] for pid in pts}
    cameras, images = [], []
    xyz = np.stack([pts[p][0] for p in sorted(pts)])
    for i, c in enumerate(cams):
        name = f"view_{i:04d}.{fmt}"
        c.image_path = os.path.join(img_dir, name)
        K = np.asarray(c.K, np.float64)
        cameras.append(cio.Camera(i + 1, "PINHOLE", c.width, c.height, [K[0, 0], K[1, 1], K[0, 2], K[1, 2]]))
        proj = (np.asarray(c.P, np.float64) @ np.concatenate([xyz, np.ones((xyz.shape[0], 1))], 1).T).T
        uv = proj[:, :2] / proj[:, 2:3]
        seen = (proj[:, 2] > 0) & (uv[:, 0] >= 0) & (uv[:, 0] < c.width) & (uv[:, 1] >= 0) & (uv[:, 1] < c.height) & (rs.rand(xyz.shape[0]) < 0.7)
        ids = (np.flatnonzero(seen) + 1).astype(np.int64)
        for j, pid in enumerate(ids):
            tracks[int(pid)].append((i + 1, j))
        images.append(cio.Image(i + 1, cio.quaternion_from_rotation(np.asarray(c.R, np.float64)), np.asarray(c.t, np.float64).reshape(3), i + 1, name,
                                uv[seen], ids))
    cio.write_cameras_bin(os.path.join(sparse, "cameras.bin"), cameras)
    cio.write_images_bin(os.path.join(sparse, "images.bin"), images)
    cio.write_colmap_points3D_bin(os.path.join(sparse, "points3D.bin"), pts, tracks)

    def one(i):
        if not os.path.exists(cams[i].image_path):
            im = Image.fromarray(synth_image(height, width, seed * 1000 + i).numpy())
            im.save(cams[i].image_path, **({"quality": 90} if fmt.lower() in ("jpg", "jpeg") else {}))
    with ThreadPoolExecutor(max_workers=max(1, int(workers))) as pool:
        list(pool.map(one, range(n_cams)))
    return cams


class SyntheticMatcher:
    """Stand-in for upstream's ``RomaMatcher`` (core/matcher.py:74-211: ``w_resized / h_resized / sample_thresh / match_grids_batch / close``) on a
    scene whose cameras it knows: ``match_grids_batch`` returns the analytic warp + certainty of ``synth_reference`` for the reference and the
    neighbours it is handed, resident on ``device``.  The fields of the references named to ``precompute`` are made ahead of the run (a table, so the
    matcher costs a dictionary lookup); ``latency_s_per_pair`` then stands in for the model's forward (a host sleep: RoMa-v2 itself takes tens of
    ms per pair).  Which cameras a call is about comes from ``keys`` (this package's driver names them: ``supports_feature_keys``) or, for a driver
    that hands over nothing but images (upstream's), from a fingerprint of the match-size image registered with ``register_image``."""
    sample_thresh = 0.9
    accepts_device_images = True
    supports_feature_keys = True

    def __init__(self, cams: Sequence[CameraRecord], setting: str = "fast", device="cpu", *, noise_px: float = 0.5, outlier_frac: float = 0.05,
                 channels: int = 2, seed: int = 0, latency_s_per_pair: float = 0.0, cert_mode: str = "smooth"):
        self.cams, self.device = list(cams), torch.device(device)
        h_lr, w_lr, self.H, self.W = ROMA_PRESETS[setting]
        self.w_resized, self.h_resized = int(w_lr), int(h_lr)
        self.kw = dict(noise_px=noise_px, outlier_frac=outlier_frac, channels=int(channels), seed=int(seed), cert_mode=cert_mode)
        self.latency = float(latency_s_per_pair)
        self.table: dict = {}
        self._axes: dict = {}
        self.fingerprints: dict = {}
        self.calls = self.pairs = 0
        self.seconds = 0.0                 # wall time spent inside match_grids_batch (the stand-in latency included)

    def set_feature_cache(self, cache) -> None:
        pass

    def reference_axes(self, H: int, W: int):
        key = (int(H), int(W))
        if key not in self._axes:
            self._axes[key] = (identity_axis_torch(W, self.device), identity_axis_torch(H, self.device))
        return self._axes[key]

    @staticmethod
    def _fingerprint(img) -> bytes:
        a = np.asarray(img) if not isinstance(img, torch.Tensor) else img.cpu().numpy()
        h, w = a.shape[:2]
        ys, xs = np.linspace(1, h - 2, 6).astype(int), np.linspace(1, w - 2, 6).astype(int)
        return a[np.ix_(ys, xs)].tobytes()

    def register_image(self, cam_index: int, image) -> None:
        """``image``: the camera's match-size image exactly as the driver's loader prepares it"""
        self.fingerprints[self._fingerprint(image)] = int(cam_index)

    def fields(self, ref: int, nbrs: Sequence[int]):
        key = (int(ref), tuple(int(n) for n in nbrs))
        hit = self.table.get(key)
        if hit is None:
            s = synth_reference(self.cams, key[0], list(key[1]), self.H, self.W, self.w_resized, self.h_resized, device=self.device, **self.kw)
            hit = [(s.warp[j].contiguous(), s.cert[j].contiguous()) for j in range(len(key[1]))]
        return key, hit

    def precompute(self, refs: Sequence[int], nn_table, nns_per_ref: int) -> int:
        """The fields of these references with the neighbours the driver will load (``nn_table[r][:nns_per_ref]`` without r itself)."""
        for r in refs:
            nbrs = [int(n) for n in nn_table[int(r)][:int(nns_per_ref)] if self.cams[int(n)].uid != self.cams[int(r)].uid]
            if nbrs:
                key, val = self.fields(int(r), nbrs)
                self.table[key] = val
        return len(self.table)

    def match_grids_batch(self, imA, imB_list, keys=None):
        import time
        t0 = time.perf_counter()
        if keys is not None:
            ref, nbrs = int(keys[0]), [int(k) for k in keys[1]]
        else:
            ref, nbrs = self.fingerprints[self._fingerprint(imA)], [self.fingerprints[self._fingerprint(b)] for b in imB_list]
        _key, out = self.fields(ref, nbrs)
        self.calls += 1
        self.pairs += len(out)
        if self.latency > 0.0:
            time.sleep(max(0.0, t0 + self.latency * len(out) - time.perf_counter()))
        self.seconds += time.perf_counter() - t0
        return list(out)

    def close(self) -> None:
        pass
