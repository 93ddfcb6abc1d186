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
                    patch_depths: Sequence[float] = (6.0, 60.0)) -> SyntheticReference:
    """Dense warp + certainty of ``ref_index`` into each neighbour.

    low_parallax_patch: (y0, y1, x0, x1) as fractions of the grid.  Inside it the surface is replaced by a ramp of
    depths ``patch_depths[0] .. patch_depths[1]`` along x (distance from the reference camera), so that the angle
    between the two viewing rays falls through the min-parallax threshold (0.5 degrees by default) inside the patch:
    SURVEY 8d config 3 asks for "a patch with sub-0.5 degree parallax" to exercise that reject reason.

    cert_mode: "smooth"  - low-frequency overlap-like field + 2 % jitter (coherent best-neighbour regions)
               "tiefree" - distinct uniform values in (0.2, 0.9) (no ties under floor/cap clamps)
               "beta"    - iid Beta(2,2) (floor and cap clamps produce massive ties, like real data edges)
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
