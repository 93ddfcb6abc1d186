"""CPU oracle for the dense-initialisation hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This module restates, in NumPy, the algorithm of the upstream plugin's per-reference
"aggregate -> select -> Sampson -> DLT -> reprojection / cheirality / parallax -> colour" stage.
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
it, and only as the checker.  The shipped path (``lichtfeld-densification-plugin_amd``) never imports
anything from ``oracle/`` and raises if its HIP library is missing.

Parity pinning
--------------
The upstream repository ships no tests, golden vectors or fixtures for this path ("parity unpinned"
by upstream tests).  The oracle is therefore pinned by *import*: ``tests/golden/make_golden.py``
imports the upstream modules in the development container (with stubbed host modules), runs them on
seeded synthetic inputs and commits inputs + outputs as ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` requires this restatement to reproduce those outputs bit for bit.

Third-party arithmetic on the path (not under /root/reference): NumPy 2.2.6 ``linalg.svd`` /
``linalg.inv`` (LAPACK ``sgesdd`` / ``sgesv`` from the OpenBLAS bundled in the NumPy wheel), the
legacy MT19937 ``numpy.random.choice`` and torch 2.10 ``Tensor.sum`` (f32 cascade sum).  The oracle
calls the same library entry points the upstream call sites use (geometry.py:79,84,127-128,
sampling.py:19,32,38, pipeline.py:634-636) so that it is bit-faithful on the same machine.

All citations are ``file:line`` relative to the upstream checkout.

dtype ladder reproduced here:
  f32  certainty floor / masks / arg-max / coordinate conversion / DLT rows / SVD / reprojection /
       cheirality / parallax
  f64  Sampson error, bilinear colour weights and accumulation (rounded to f32 on emit)
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

__all__ = [
    "OracleParams", "OracleCamera", "SegmentResult", "ReferenceResult",
    "identity_axis", "identity_axis_scalar", "certainty_prologue", "nearest_resize_mask", "warp_mask_nearest",
    "aggregate_best", "select_samples", "fundamental_matrix", "sampson_error",
    "dlt_triangulate", "reprojection_error", "depth_positive", "parallax_ok",
    "bilinear_colour", "match_pixels", "triangulate_selected", "triangulate_reference",
    "triangulate_dense", "cell_diagnostics", "cell_diagnostics_exact", "classify_flips", "FLIP_REASONS",
    "to_uint8_rgb", "ply_bytes", "points3d_bin_bytes",
]


# --------------------------------------------------------------------------------------------
# carriers
# --------------------------------------------------------------------------------------------
@dataclass
class OracleParams:
    """Thresholds of ``DensePipelineConfig`` that reach the hot path (core/config.py:7-26)."""
    certainty_thresh: float = 0.20
    reproj_thresh: float = 0.8
    sampson_thresh: float = 5.0
    min_parallax_deg: float = 0.5
    no_filter: bool = False
    matches_per_ref: int = 10000
    sample_cap: float = 0.9          # RomaMatcher.sample_thresh, core/matcher.py:92


@dataclass
class OracleCamera:
    """f32 camera block as carried by ``_CameraLookup`` (core/pipeline.py:68-78)."""
    K: np.ndarray   # (3,3) f32
    R: np.ndarray   # (3,3) f32
    t: np.ndarray   # (3,1) f32
    P: np.ndarray   # (3,4) f32
    C: np.ndarray   # (3,)  f32
    width: int
    height: int


@dataclass
class SegmentResult:
    """Survivors of one (reference, neighbour) group, in the order upstream emits them."""
    nbr_slot: int
    xyz: np.ndarray          # (n,3) f32
    rgb: np.ndarray          # (n,3) f32 in [0,1]
    err: np.ndarray          # (n,)  f32
    sel_pos: np.ndarray      # (n,)  i64 position inside sel_idx
    cell: np.ndarray         # (n,)  i64 flat grid index y*W+x
    matches_px: np.ndarray   # (n,4) f32 clipped [xA,yA,xB,yB] in match pixels (debug previews)
    cert_norm: np.ndarray    # (n,)  f32 clip(cert / cap, 0, 1)


@dataclass
class ReferenceResult:
    segments: List[SegmentResult] = field(default_factory=list)

    @property
    def xyz(self) -> np.ndarray:
        return (np.concatenate([s.xyz for s in self.segments], axis=0)
                if self.segments else np.zeros((0, 3), np.float32))

    @property
    def rgb(self) -> np.ndarray:
        return (np.concatenate([s.rgb for s in self.segments], axis=0)
                if self.segments else np.zeros((0, 3), np.float32))

    @property
    def err(self) -> np.ndarray:
        return (np.concatenate([s.err for s in self.segments], axis=0)
                if self.segments else np.zeros((0,), np.float32))

    @property
    def cell(self) -> np.ndarray:
        return (np.concatenate([s.cell for s in self.segments], axis=0)
                if self.segments else np.zeros((0,), np.int64))

    @property
    def nbr(self) -> np.ndarray:
        return (np.concatenate([np.full(s.cell.shape, s.nbr_slot, np.int64) for s in self.segments])
                if self.segments else np.zeros((0,), np.int64))

    @property
    def count(self) -> int:
        return int(sum(s.xyz.shape[0] for s in self.segments))


# --------------------------------------------------------------------------------------------
# producer-side conventions (RomaMatcher)
# --------------------------------------------------------------------------------------------
def identity_axis(n: int) -> np.ndarray:
    """A-grid coordinate of column/row ``j``: ``torch.linspace(-1+1/n, 1-1/n, n)`` in f32
    (core/matcher.py:132-133).  torch's result depends on the device kernel (the CPU kernel is
    vectorised: ``base + step*lane`` per vector; the GPU kernel evaluates ``start + step*j`` /
    ``end - step*(n-1-j)`` per element), differing by at most one ulp, so the oracle calls the same
    library routine and every comparison passes the axes explicitly."""
    import torch
    return torch.linspace(-1 + 1 / n, 1 - 1 / n, n).numpy()


def identity_axis_scalar(n: int) -> np.ndarray:
    """Per-element form of ``torch.linspace`` (the device kernel's formula without contraction):
    ``start + step*j`` below the midpoint, ``end - step*(n-1-j)`` from it on, ``step`` in f32."""
    start = np.float32(-1 + 1 / n)
    end = np.float32(1 - 1 / n)
    if n == 1:
        return np.array([start], np.float32)
    step = np.float32((end - start) / np.float32(n - 1))
    j = np.arange(n)
    lo = (start + (step * j.astype(np.float32)).astype(np.float32)).astype(np.float32)
    hi = (end - (step * (n - 1 - j).astype(np.float32)).astype(np.float32)).astype(np.float32)
    return np.where(j < n // 2, lo, hi).astype(np.float32)


# --------------------------------------------------------------------------------------------
# P1: certainty prologue (core/pipeline.py:405-430)
# --------------------------------------------------------------------------------------------
def nearest_resize_mask(mask01: np.ndarray, out_hw: Tuple[int, int]) -> np.ndarray:
    """``F.interpolate(mode="nearest")`` of a {0,1} mask to ``out_hw`` (core/pipeline.py:372-378):
    ``src = min(floor(dst * (in/out as f32)), in-1)``."""
    m = np.asarray(mask01, np.float32)
    ih, iw = m.shape
    oh, ow = out_hw
    if (ih, iw) == (oh, ow):
        return m
    sy = np.float32(ih) / np.float32(oh)
    sx = np.float32(iw) / np.float32(ow)
    yy = np.minimum(np.floor(np.arange(oh, dtype=np.float32) * sy).astype(np.int64), ih - 1)
    xx = np.minimum(np.floor(np.arange(ow, dtype=np.float32) * sx).astype(np.int64), iw - 1)
    return m[yy[:, None], xx[None, :]]


def warp_mask_nearest(mask_hw: np.ndarray, xb_norm: np.ndarray, yb_norm: np.ndarray) -> np.ndarray:
    """``F.grid_sample(mask, warp[...,2:4], mode="nearest", padding_mode="zeros",
    align_corners=False)`` (core/pipeline.py:419-430).  Un-normalise ``((g+1)*size-1)/2`` in f32,
    round half to even, out-of-range -> 0."""
    h, w = mask_hw.shape
    gx = np.asarray(xb_norm, np.float32)
    gy = np.asarray(yb_norm, np.float32)
    with np.errstate(invalid="ignore", over="ignore"):
        fx = ((gx + np.float32(1)) * np.float32(w) - np.float32(1)) / np.float32(2)
        fy = ((gy + np.float32(1)) * np.float32(h) - np.float32(1)) / np.float32(2)
        ix = np.rint(fx)
        iy = np.rint(fy)
        ok = (ix >= 0) & (ix <= w - 1) & (iy >= 0) & (iy <= h - 1)   # NaN compares false
    ixc = np.where(ok, ix, 0).astype(np.int64)
    iyc = np.where(ok, iy, 0).astype(np.int64)
    return np.where(ok, mask_hw[iyc, ixc].astype(np.float32), np.float32(0)).astype(np.float32)


def certainty_prologue(cert_hw: np.ndarray, warp_hw: np.ndarray, certainty_thresh: float,
                       mask_a: Optional[np.ndarray] = None,
                       mask_b: Optional[np.ndarray] = None) -> np.ndarray:
    """Raise certainty to the floor (``torch.clamp(min=...)``: a floor, not a reject), then zero it
    under reference mask A (nearest-resized to the grid) and under neighbour mask B sampled at the
    warped position (core/pipeline.py:407-430)."""
    c = np.maximum(np.asarray(cert_hw, np.float32), np.float32(certainty_thresh))   # NaN propagates, as in torch
    h, w = c.shape
    if mask_a is not None:
        c = (c * nearest_resize_mask(mask_a, (h, w))).astype(np.float32)
    if mask_b is not None:
        mb = nearest_resize_mask(mask_b, (h, w))
        c = (c * warp_mask_nearest(mb, warp_hw[..., -2], warp_hw[..., -1])).astype(np.float32)
    return c


# --------------------------------------------------------------------------------------------
# F1: aggregate (core/pipeline.py:632-640)
# --------------------------------------------------------------------------------------------
def aggregate_best(cert_list: Sequence[np.ndarray], warp_list: Sequence[np.ndarray]):
    """Per-pixel maximum certainty over the k neighbours and the warp of the winner.
    ``torch.max(dim=0)`` returns the FIRST maximal index on ties, and a NaN wins over any number
    (first NaN).  Returns ``best_cert (H,W) f32, best_k (H,W) i64, agg (H*W,C) f32``."""
    cert = np.stack([np.asarray(c, np.float32) for c in cert_list], axis=0)
    k, h, w = cert.shape
    best = cert[0].copy()
    best_k = np.zeros((h, w), np.int64)
    for j in range(1, k):
        c = cert[j]
        take = (c > best) | (np.isnan(c) & ~np.isnan(best))
        best = np.where(take, c, best)
        best_k = np.where(take, j, best_k)
    warp = np.stack([np.asarray(wp, np.float32) for wp in warp_list], axis=0)
    ys, xs = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    agg = warp[best_k, ys, xs].reshape(h * w, warp.shape[-1])
    return best.astype(np.float32), best_k, agg


# --------------------------------------------------------------------------------------------
# S: coverage sampling (core/sampling.py:8-53)
# --------------------------------------------------------------------------------------------
def _torch_sum_f32(x: np.ndarray) -> np.float32:
    """``weights.sum()`` is a torch CPU f32 reduction (core/sampling.py:27); its cascade order is
    not NumPy's pairwise order, so the same library routine is called."""
    import torch
    return np.float32(torch.from_numpy(np.ascontiguousarray(x, np.float32)).sum().item())


def select_samples(best_cert: np.ndarray, M: int, cap: float = 0.9, border: int = 2,
                   tiles: int = 24, no_filter: bool = False,
                   rng: Optional[np.random.RandomState] = None, s_override: Optional[float] = None) -> np.ndarray:
    """Which grid cells get triangulated.

    no_filter: the ``M`` largest capped certainties in ``argsort(-flat)`` order (unsorted index
    order; tie order is whatever NumPy's unstable quicksort yields).
    filter   : ``int(0.85*M)`` cells drawn without replacement with probability proportional to
    the capped certainty inside a 2-px border (legacy ``RandomState.choice``), plus, walking cells by
    descending weight, the first cell of every not-yet-seen ``W//24``-pixel tile; the union is
    returned sorted ascending (``np.unique``).

    ``rng`` stands in for the process-global legacy NumPy RNG the upstream code consumes
    (core/sampling.py:32; seeded at core/pipeline.py:793)."""
    cert = np.minimum(np.asarray(best_cert, np.float32), np.float32(cap))   # torch.clamp(max=cap); NaN propagates
    h, w = cert.shape
    if no_filter:
        flat = cert.reshape(-1)
        if flat.size == 0:
            return np.zeros((0,), np.int64)
        order = np.argsort(-flat)
        return order[:min(M, flat.size)]
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    inside = (xx >= border) & (xx <= w - 1 - border) & (yy >= border) & (yy <= h - 1 - border)
    weights = (cert * inside.astype(np.float32)).reshape(-1).astype(np.float32)
    # s_override: the normaliser another implementation used (torch's own f32 sum depends on the host's
    # thread count and vector ISA, so it is an input of the comparison, not part of the algorithm)
    s = _torch_sum_f32(weights) if s_override is None else np.float32(s_override)
    if not (s > 0):
        return np.zeros((0,), np.int64)
    weights = (weights / s).astype(np.float32)
    m_main = int(M * 0.85)
    chooser = rng if rng is not None else np.random
    idx_main = chooser.choice(weights.size, size=min(m_main, weights.size), replace=False, p=weights)
    tile = max(1, w // tiles)
    bins = ((xx // tile) * 100000 + (yy // tile)).reshape(-1)
    order = np.argsort(-weights)
    budget = M - len(idx_main)
    seen = set()
    cov: List[int] = []
    for i in order:
        if weights[i] <= 0:
            break
        b = int(bins[i])
        if b in seen:
            continue
        seen.add(b)
        cov.append(int(i))
        if len(cov) >= budget:
            break
    return np.unique(np.concatenate([idx_main, np.asarray(cov, np.int64)]))


# --------------------------------------------------------------------------------------------
# F5-F9: two-view geometry (core/geometry.py)
# --------------------------------------------------------------------------------------------
def fundamental_matrix(K1, R1, t1, K2, R2, t2) -> np.ndarray:
    """``F = K2^-T [t]x R K1^-1`` with ``R = R2 R1^T``, ``t = t2 - R t1``; all f32, inverses by
    LAPACK (core/geometry.py:53-55,122-130)."""
    R = R2 @ R1.T
    t = (t2 - R @ t1).reshape(3)
    tx, ty, tz = t
    cross = np.array([[0, -tz, ty], [tz, 0, -tx], [-ty, tx, 0]], dtype=np.float32)
    E = cross @ R
    return np.linalg.inv(K2).T @ E @ np.linalg.inv(K1)


def sampson_error(F: np.ndarray, uv1: np.ndarray, uv2: np.ndarray) -> np.ndarray:
    """First-order geometric error; the homogeneous ``ones`` column is f64 so the whole expression
    runs in f64 from f32 inputs (core/geometry.py:133-141)."""
    n = uv1.shape[0]
    x1 = np.concatenate([uv1, np.ones((n, 1))], axis=1)
    x2 = np.concatenate([uv2, np.ones((n, 1))], axis=1)
    Fx1 = (F @ x1.T).T
    Ftx2 = (F.T @ x2.T).T
    num = np.sum(x2 * Fx1, axis=1)
    den = Fx1[:, 0] ** 2 + Fx1[:, 1] ** 2 + Ftx2[:, 0] ** 2 + Ftx2[:, 1] ** 2 + 1e-12
    return (num ** 2) / den


def dlt_rows(P1, P2, uv1, uv2) -> np.ndarray:
    """The (n,4,4) f32 DLT system: rows ``u*p2-p0``, ``v*p2-p1`` per view (core/geometry.py:63-75)."""
    n = uv1.shape[0]
    A = np.empty((n, 4, 4), np.float32)
    A[:, 0, :] = uv1[:, 0:1] * P1[2] - P1[0]
    A[:, 1, :] = uv1[:, 1:2] * P1[2] - P1[1]
    A[:, 2, :] = uv2[:, 0:1] * P2[2] - P2[0]
    A[:, 3, :] = uv2[:, 1:2] * P2[2] - P2[1]
    return A


def dlt_triangulate(P1, P2, uv1, uv2) -> np.ndarray:
    """Homogeneous point = right singular vector of the smallest singular value (f32 ``sgesdd``),
    divided by its 4th component, which is replaced by +1e-12 when |w| < 1e-12
    (core/geometry.py:58-87; the n==1 branch at :77-82 uses ``>`` instead of ``<``, identical
    except at |w| == 1e-12 exactly)."""
    n = uv1.shape[0]
    if n == 0:
        return np.zeros((0, 4), np.float32)
    A = dlt_rows(P1, P2, uv1, uv2)
    if n == 1:
        Vt = np.linalg.svd(A[0])[2]
        Xh = Vt[-1]
        wv = Xh[3] if abs(Xh[3]) > 1e-12 else 1e-12
        return (Xh / wv)[None, :]
    Vt = np.linalg.svd(A)[2]
    Xh = Vt[:, -1, :]
    wv = np.where(np.abs(Xh[:, 3:4]) < 1e-12, 1e-12, Xh[:, 3:4])
    return Xh / wv


def reprojection_error(P, X, uv) -> np.ndarray:
    """Pixel distance between ``P X`` (depth floored at 1e-12) and ``uv`` (core/geometry.py:91-104)."""
    proj = X @ P.T
    z = np.maximum(proj[:, 2], 1e-12)
    du = proj[:, 0] / z - uv[:, 0]
    dv = proj[:, 1] / z - uv[:, 1]
    return np.sqrt(du * du + dv * dv)


def depth_positive(P, X) -> np.ndarray:
    """Cheirality: third row of ``P X^T`` > 0 (core/geometry.py:107-110)."""
    return (P @ X.T)[2, :] > 0.0


def parallax_angle_deg(C1, C2, X) -> np.ndarray:
    v1 = X[:, :3] - C1.reshape(1, 3)
    v2 = X[:, :3] - C2.reshape(1, 3)
    v1 /= np.linalg.norm(v1, axis=1, keepdims=True) + 1e-12
    v2 /= np.linalg.norm(v2, axis=1, keepdims=True) + 1e-12
    return np.degrees(np.arccos(np.clip(np.sum(v1 * v2, axis=1), -1.0, 1.0)))


def parallax_ok(C1, C2, X, min_deg: float) -> np.ndarray:
    """Angle between the unit rays from both centres to X >= ``min_deg`` (core/geometry.py:113-119)."""
    return parallax_angle_deg(C1, C2, X) >= float(min_deg)


# --------------------------------------------------------------------------------------------
# F2/F3: coordinates and colour (core/pipeline.py:653-683)
# --------------------------------------------------------------------------------------------
def match_pixels(norm: np.ndarray, size_match: int) -> np.ndarray:
    """Normalised [-1,1] -> match-image pixels ``(n+1)*0.5*(size-1)`` in f32
    (core/pipeline.py:655-656,701-702)."""
    return (norm + 1.0) * 0.5 * (size_match - 1)


def bilinear_colour(img_u8: np.ndarray, x_px: np.ndarray, y_px: np.ndarray,
                    w_match: int, h_match: int) -> np.ndarray:
    """Bilinear RGB from the (already resized) reference image; integer-minus-f32 weights promote
    to f64 and the 4-tap sum runs in f64; returns f64 in [0,1] (core/pipeline.py:661-679)."""
    hi, wi = img_u8.shape[0], img_u8.shape[1]
    xi = x_px * (wi / float(w_match))
    yi = y_px * (hi / float(h_match))
    with np.errstate(invalid="ignore"):
        x0 = np.clip(np.floor(xi).astype(np.int32), 0, wi - 1)
        y0 = np.clip(np.floor(yi).astype(np.int32), 0, hi - 1)
    x1 = np.clip(x0 + 1, 0, wi - 1)
    y1 = np.clip(y0 + 1, 0, hi - 1)
    wa = (x1 - xi) * (y1 - yi)
    wb = (xi - x0) * (y1 - yi)
    wc = (x1 - xi) * (yi - y0)
    wd = (xi - x0) * (yi - y0)
    Ia = img_u8[y0, x0].astype(np.float32)
    Ib = img_u8[y0, x1].astype(np.float32)
    Ic = img_u8[y1, x0].astype(np.float32)
    Id = img_u8[y1, x1].astype(np.float32)
    return (Ia * wa[:, None] + Ib * wb[:, None] + Ic * wc[:, None] + Id * wd[:, None]) / 255.0


# --------------------------------------------------------------------------------------------
# F2-F10: per-reference triangulation of a given selection (core/pipeline.py:653-780)
# --------------------------------------------------------------------------------------------
def triangulate_selected(sel_idx: np.ndarray, best_cert: np.ndarray, best_k: np.ndarray,
                         agg: np.ndarray, img_a: np.ndarray, cam_a: OracleCamera,
                         cams_b: Sequence[OracleCamera], w_match: int, h_match: int,
                         params: OracleParams, axes=None) -> ReferenceResult:
    """Everything after selection.  ``agg`` has 4 columns [xA,yA,xB,yB] (or 2 columns [xB,yB]; the
    A columns are then the analytic identity grid).  Groups are formed per neighbour in order of first
    appearance while scanning ``sel_idx`` in the order given; members keep that order."""
    sel_idx = np.asarray(sel_idx, np.int64)
    out = ReferenceResult()
    if sel_idx.size == 0:
        return out
    h, w = best_cert.shape
    if agg.shape[1] == 2:
        ax, ay = axes if axes is not None else (identity_axis(w), identity_axis(h))
        agg = np.concatenate([ax[np.arange(h * w) % w, None], ay[np.arange(h * w) // w, None], agg],
                             axis=1).astype(np.float32)
    nbr_of = best_k.reshape(-1)[sel_idx]
    sel = agg[sel_idx]
    xA = match_pixels(sel[:, 0], w_match)
    yA = match_pixels(sel[:, 1], h_match)
    xBn, yBn = sel[:, 2], sel[:, 3]
    cert_sel = best_cert.reshape(-1)[sel_idx]
    rgb_all = bilinear_colour(img_a, xA, yA, w_match, h_match)
    uvA_all = np.stack([xA * (cam_a.width / float(w_match)), yA * (cam_a.height / float(h_match))], axis=1)

    order: List[int] = []
    members: Dict[int, List[int]] = {}
    for pos, kk in enumerate(nbr_of):
        kk = int(kk)
        if kk not in members:
            members[kk] = []
            order.append(kk)
        members[kk].append(pos)

    cap = float(params.sample_cap) if float(params.sample_cap) > 1e-6 else 1.0
    for kk in order:
        pos = np.asarray(members[kk], np.int64)
        cb = cams_b[kk]
        xB = match_pixels(xBn[pos], w_match)
        yB = match_pixels(yBn[pos], h_match)
        uvB = np.stack([xB * (cb.width / float(w_match)), yB * (cb.height / float(h_match))], axis=1)
        if (not params.no_filter) and params.sampson_thresh > 0:
            F = fundamental_matrix(cam_a.K, cam_a.R, cam_a.t, cb.K, cb.R, cb.t)
            good = sampson_error(F, uvA_all[pos], uvB) < float(params.sampson_thresh)
            if not np.any(good):
                continue
            pos, xB, yB, uvB = pos[good], xB[good], yB[good], uvB[good]
        if pos.size == 0:
            continue
        uvA = uvA_all[pos]
        X = dlt_triangulate(cam_a.P, cb.P, uvA, uvB)
        with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
            err = np.maximum(reprojection_error(cam_a.P, X, uvA), reprojection_error(cb.P, X, uvB))
            if params.no_filter:
                keep = np.isfinite(X).all(axis=1) & np.isfinite(err)
            else:
                keep = err <= float(params.reproj_thresh)
                keep &= depth_positive(cam_a.P, X)
                keep &= depth_positive(cb.P, X)
                if params.min_parallax_deg > 0:
                    keep &= parallax_ok(cam_a.C, cb.C, X, params.min_parallax_deg)
        if not np.any(keep):
            continue
        kp = pos[keep]
        m = np.stack([np.clip(xA[kp], 0.0, float(w_match - 1)), np.clip(yA[kp], 0.0, float(h_match - 1)),
                      np.clip(xB[keep], 0.0, float(w_match - 1)), np.clip(yB[keep], 0.0, float(h_match - 1))],
                     axis=1).astype(np.float32)
        out.segments.append(SegmentResult(
            nbr_slot=kk,
            xyz=X[keep][:, :3].astype(np.float32),
            rgb=rgb_all[kp].astype(np.float32),
            err=err[keep].astype(np.float32),
            sel_pos=kp,
            cell=sel_idx[kp],
            matches_px=m,
            cert_norm=np.clip(cert_sel[kp] / cap, 0.0, 1.0).astype(np.float32),
        ))
    return out


def prepare_reference(cert_list, warp_list, params: OracleParams, mask_a=None, mask_b_list=None):
    """P1 + F1 for one reference: returns ``best_cert, best_k, agg``."""
    certs = []
    for j, (c, wp) in enumerate(zip(cert_list, warp_list)):
        mb = None if mask_b_list is None else mask_b_list[j]
        certs.append(certainty_prologue(c, wp, params.certainty_thresh, mask_a, mb))
    return aggregate_best(certs, warp_list)


def triangulate_reference(cert_list, warp_list, img_a, cam_a, cams_b, w_match, h_match,
                          params: OracleParams, rng=None, sel_idx=None, mask_a=None,
                          mask_b_list=None, apply_prologue=True, axes=None):
    """The whole per-reference stage: prologue (optional: upstream applies it in
    ``_collect_reference_matches`` before ``_triangulate_ref`` sees the maps), aggregate, select
    (unless ``sel_idx`` is supplied) and triangulate.  Returns ``(ReferenceResult, sel_idx)``."""
    if apply_prologue:
        best_cert, best_k, agg = prepare_reference(cert_list, warp_list, params, mask_a, mask_b_list)
    else:
        best_cert, best_k, agg = aggregate_best(cert_list, warp_list)
    if sel_idx is None:
        sel_idx = select_samples(best_cert, params.matches_per_ref, cap=params.sample_cap, border=2,
                                 tiles=24, no_filter=params.no_filter, rng=rng)
    res = triangulate_selected(sel_idx, best_cert, best_k, agg, img_a, cam_a, cams_b, w_match, h_match, params,
                               axes=axes)
    return res, np.asarray(sel_idx, np.int64)


def triangulate_dense(cert_list, warp_list, img_a, cam_a, cams_b, w_match, h_match,
                      params: OracleParams, mask_a=None, mask_b_list=None, axes=None):
    """Dense extension used by the fused kernel: every grid cell upstream's sampler could ever draw is "selected" - the cells whose
    best certainty after floor and masks is not <= 0 (core/sampling.py:24-27, 41-43 upstream: p = weights / sum, the coverage
    pass stops at weights <= 0; a masked-out cell has weight 0; the border upstream excludes from sampling stays in) - the selection
    stage is skipped, certainty otherwise only decides the winning neighbour.  Survivors are returned in raster order together
    with their cell index and neighbour slot.  With upstream's rules this is exactly
    ``triangulate_selected(nonzero(~(best_cert <= 0)))`` re-sorted by cell."""
    best_cert, best_k, agg = prepare_reference(cert_list, warp_list, params, mask_a, mask_b_list)
    h, w = best_cert.shape
    with np.errstate(invalid="ignore"):
        candidates = np.nonzero(~(best_cert.reshape(-1) <= 0))[0].astype(np.int64)      # (a NaN best certainty stays a candidate, as before)
    res = triangulate_selected(candidates, best_cert, best_k, agg, img_a, cam_a,
                               cams_b, w_match, h_match, params, axes=axes)
    cell = res.cell
    order = np.argsort(cell, kind="stable")
    return {
        "xyz": res.xyz[order], "rgb": res.rgb[order], "err": res.err[order],
        "cell": cell[order], "nbr": res.nbr[order],
        "seg_counts": np.array([sum(s.xyz.shape[0] for s in res.segments if s.nbr_slot == j)
                                for j in range(len(cert_list))], np.int64),
        "best_cert": best_cert, "best_k": best_k,
    }


def _reproj_noise(P, X) -> np.ndarray:
    """First-order bound on the f32 rounding noise of one view's reprojection error, in pixels: the
    three dot products of ``X @ P.T`` each carry ~eps32 * sum|terms| of absolute error and the
    division by the depth amplifies the depth's share by |u|/z.  For an ordinary cell this is ~1e-4 px;
    for a point that almost lies in the neighbour's principal plane (z -> 0, |u| ~ 1e5 px) it reaches
    a pixel, i.e. upstream's own error value is then not reproducible by ANY other evaluation order."""
    eps = np.float64(np.finfo(np.float32).eps)
    Xd = np.abs(X.astype(np.float64))
    Pd = np.abs(P.astype(np.float64))
    mag = Xd @ Pd.T                                    # sum of |terms| per row
    pr = X.astype(np.float64) @ P.astype(np.float64).T
    z = np.maximum(np.abs(pr[:, 2]), 1e-30)
    uv = np.hypot(pr[:, 0], pr[:, 1]) / z
    return (4.0 * eps * ((mag[:, 0] + mag[:, 1]) / z + mag[:, 2] / z * uv)).astype(np.float32)


def cell_diagnostics(cells: np.ndarray, best_k: np.ndarray, agg: np.ndarray, cam_a: OracleCamera,
                     cams_b: Sequence[OracleCamera], w_match: int, h_match: int,
                     axes=None) -> Dict[str, np.ndarray]:
    """Per-cell decision variables (Sampson error, reprojection error, both depths, parallax angle)
    for guard-band checks: a cell whose value lies within epsilon of a threshold may legitimately
    flip between two correct implementations of the f32 SVD."""
    cells = np.asarray(cells, np.int64)
    h, w = best_k.shape
    if agg.shape[1] == 2:
        ax, ay = axes if axes is not None else (identity_axis(w), identity_axis(h))
        agg = np.concatenate([ax[np.arange(h * w) % w, None], ay[np.arange(h * w) // w, None], agg],
                             axis=1).astype(np.float32)
    n = cells.size
    se = np.full(n, np.nan)
    err = np.full(n, np.nan, np.float32)
    z1 = np.full(n, np.nan, np.float32)
    z2 = np.full(n, np.nan, np.float32)
    ang = np.full(n, np.nan, np.float32)
    sv_gap = np.full(n, np.nan, np.float32)
    noise = np.full(n, np.nan, np.float32)
    sel = agg[cells]
    xA = match_pixels(sel[:, 0], w_match)
    yA = match_pixels(sel[:, 1], h_match)
    uvA_all = np.stack([xA * (cam_a.width / float(w_match)), yA * (cam_a.height / float(h_match))], axis=1)
    kk_all = best_k.reshape(-1)[cells]
    for kk in np.unique(kk_all):
        pos = np.nonzero(kk_all == kk)[0]
        cb = cams_b[int(kk)]
        xB = match_pixels(sel[pos, 2], w_match)
        yB = match_pixels(sel[pos, 3], h_match)
        uvB = np.stack([xB * (cb.width / float(w_match)), yB * (cb.height / float(h_match))], axis=1)
        uvA = uvA_all[pos]
        F = fundamental_matrix(cam_a.K, cam_a.R, cam_a.t, cb.K, cb.R, cb.t)
        se[pos] = sampson_error(F, uvA, uvB)
        with np.errstate(all="ignore"):
            A = dlt_rows(cam_a.P, cb.P, uvA, uvB)
            sv = np.linalg.svd(A.astype(np.float64), compute_uv=False)
            sv_gap[pos] = (sv[:, 3] / np.maximum(sv[:, 2], 1e-300)).astype(np.float32)
            X = dlt_triangulate(cam_a.P, cb.P, uvA, uvB)
            err[pos] = np.maximum(reprojection_error(cam_a.P, X, uvA), reprojection_error(cb.P, X, uvB))
            z1[pos] = (cam_a.P @ X.T)[2, :]
            z2[pos] = (cb.P @ X.T)[2, :]
            noise[pos] = np.maximum(_reproj_noise(cam_a.P, X), _reproj_noise(cb.P, X))
            ang[pos] = parallax_angle_deg(cam_a.C, cb.C, X)
    return {"sampson": se, "err": err, "z1": z1, "z2": z2, "parallax_deg": ang, "sv_ratio": sv_gap,
            "err_noise": noise}


def cell_diagnostics_exact(cells: np.ndarray, best_k: np.ndarray, agg: np.ndarray, cam_a: OracleCamera,
                           cams_b: Sequence[OracleCamera], w_match: int, h_match: int, axes=None) -> Dict[str, np.ndarray]:
    """The same decision variables as ``cell_diagnostics`` evaluated WITHOUT upstream's f32 rounding noise downstream
    of the DLT matrix: the f32 matrix A is upstream's (core/geometry.py:63-75), its smallest right singular vector
    comes from an f64 SVD, the point is rounded to f32 once (upstream's X is f32), and reprojection / depth / parallax
    are then evaluated in f64 on that f32 point.  ``cell_diagnostics`` (upstream's arithmetic) and this function
    bracket what ANY correct evaluation of upstream's formulas on the same inputs can return; a cell whose threshold
    lies between the two is decided by rounding, not by the algorithm (see ``classify_flips``)."""
    cells = np.asarray(cells, np.int64)
    h, w = best_k.shape
    if agg.shape[1] == 2:
        ax, ay = axes if axes is not None else (identity_axis(w), identity_axis(h))
        agg = np.concatenate([ax[np.arange(h * w) % w, None], ay[np.arange(h * w) // w, None], agg], axis=1).astype(np.float32)
    n = cells.size
    out = {k: np.full(n, np.nan) for k in ("err", "z1", "z2", "parallax_deg", "z1_noise", "z2_noise")}
    sel = agg[cells]
    xA = match_pixels(sel[:, 0], w_match)
    yA = match_pixels(sel[:, 1], h_match)
    uvA_all = np.stack([xA * (cam_a.width / float(w_match)), yA * (cam_a.height / float(h_match))], axis=1)
    kk_all = best_k.reshape(-1)[cells]
    eps = np.float64(np.finfo(np.float32).eps)
    for kk in np.unique(kk_all):
        pos = np.nonzero(kk_all == kk)[0]
        cb = cams_b[int(kk)]
        xB = match_pixels(sel[pos, 2], w_match)
        yB = match_pixels(sel[pos, 3], h_match)
        uvB = np.stack([xB * (cb.width / float(w_match)), yB * (cb.height / float(h_match))], axis=1)
        uvA = uvA_all[pos]
        with np.errstate(all="ignore"):
            A = dlt_rows(cam_a.P, cb.P, uvA, uvB).astype(np.float64)
            Vt = np.linalg.svd(A)[2]
            Xh = Vt[:, -1, :]
            wv = np.where(np.abs(Xh[:, 3:4]) < 1e-12, 1e-12, Xh[:, 3:4])
            X = (Xh / wv).astype(np.float32).astype(np.float64)          # upstream's point is f32
            e = []
            for P, uv, zk in ((cam_a.P, uvA, "z1"), (cb.P, uvB, "z2")):
                pr = X @ P.astype(np.float64).T
                z = np.maximum(pr[:, 2], 1e-12)
                e.append(np.hypot(pr[:, 0] / z - uv[:, 0].astype(np.float64), pr[:, 1] / z - uv[:, 1].astype(np.float64)))
                out[zk][pos] = pr[:, 2]
                out[zk + "_noise"][pos] = 4.0 * eps * (np.abs(X) @ np.abs(P[2].astype(np.float64)))
            out["err"][pos] = np.maximum(e[0], e[1])
            v1 = X[:, :3] - cam_a.C.astype(np.float64).reshape(1, 3)
            v2 = X[:, :3] - cb.C.astype(np.float64).reshape(1, 3)
            v1 = v1 / (np.linalg.norm(v1, axis=1, keepdims=True) + 1e-12)
            v2 = v2 / (np.linalg.norm(v2, axis=1, keepdims=True) + 1e-12)
            out["parallax_deg"][pos] = np.degrees(np.arccos(np.clip(np.sum(v1 * v2, axis=1), -1.0, 1.0)))
    return out


FLIP_REASONS = ("sampson", "reproj", "cheirality", "parallax", "non_finite")


def classify_flips(cells: np.ndarray, best_k: np.ndarray, agg: np.ndarray, cam_a: OracleCamera,
                   cams_b: Sequence[OracleCamera], w_match: int, h_match: int, params: OracleParams, axes=None,
                   sampson_rel: float = 1e-12):
    """For cells that another implementation decided differently from this oracle: which threshold explains each one.

    Noise model (stated, per reject reason; eps = 2^-23):
      sampson     upstream evaluates the test in f64 from f32 inputs.  An implementation that is handed upstream's F may
                  differ by the association order of ~20 f64 operations: |se - thr| <= sampson_rel * max(1, thr) with
                  sampson_rel = 1e-12.  (Without upstream's F - the closed-form K^-1 instead of np.linalg.inv - F moves
                  by ~2e-6 relative and the caller passes sampson_rel = 1e-5.)
      reproj      the threshold lies between upstream's f32 value (LAPACK sgesdd + f32 reprojection) and the value of
                  the same formulas free of f32 rounding noise (f64 SVD of the same f32 matrix, f64 reprojection),
                  widened by the first-order bound on the f32 rounding of ONE reprojection, ``_reproj_noise``
                  (4 eps sum|terms| / z): that is what any f32 evaluation of the formula may add.
      cheirality  zero lies between upstream's f32 depth and the f64 depth, widened by 4 eps sum|terms| of the depth row.
      parallax    the threshold lies between upstream's f32 angle and the f64 angle, widened by the f32 rounding of the
                  normalised dot product seen through arccos: 4 eps / sin(theta_min) radians.
      non_finite  upstream's value is NaN / Inf (w -> 0 guard, overflow): not reproducible by construction.
    Returns ``(reason (n,) array of indices into FLIP_REASONS, -1 = out of band, details dict)``."""
    cells = np.asarray(cells, np.int64)
    n = cells.size
    reason = np.full(n, -1, np.int64)
    if n == 0:
        return reason, {}
    with np.errstate(all="ignore"):
        up = cell_diagnostics(cells, best_k, agg, cam_a, cams_b, w_match, h_match, axes=axes)
        ex = cell_diagnostics_exact(cells, best_k, agg, cam_a, cams_b, w_match, h_match, axes=axes)
    eps = float(np.finfo(np.float32).eps)

    def between(a, b, thr, pad):
        lo, hi = np.minimum(a, b) - pad, np.maximum(a, b) + pad
        return (thr >= lo) & (thr <= hi)

    with np.errstate(all="ignore"):
        if not params.no_filter:
            thr_r = float(np.float32(params.reproj_thresh))
            in_par = np.zeros(n, bool)
            if params.min_parallax_deg > 0:
                pad_par = np.degrees(4.0 * eps / np.sin(np.radians(max(float(params.min_parallax_deg), 1e-3))))
                in_par = between(up["parallax_deg"].astype(np.float64), ex["parallax_deg"], float(params.min_parallax_deg), pad_par)
            in_z = between(up["z1"].astype(np.float64), ex["z1"], 0.0, ex["z1_noise"]) | \
                between(up["z2"].astype(np.float64), ex["z2"], 0.0, ex["z2_noise"])
            in_rep = between(up["err"].astype(np.float64), ex["err"], thr_r, up["err_noise"].astype(np.float64))
            in_sam = np.zeros(n, bool)
            if params.sampson_thresh > 0:
                in_sam = np.abs(up["sampson"] - params.sampson_thresh) <= sampson_rel * max(1.0, params.sampson_thresh)
            bad = ~np.isfinite(up["err"].astype(np.float64)) | ~np.isfinite(ex["err"])
            for idx, m in ((3, in_par), (2, in_z), (1, in_rep), (0, in_sam), (4, bad)):     # later entries take precedence
                reason[m] = idx
        else:
            reason[~np.isfinite(up["err"].astype(np.float64)) | ~np.isfinite(ex["err"])] = 4
    return reason, {"upstream": up, "exact": ex}


# --------------------------------------------------------------------------------------------
# N1: writers (core/writers.py:15-46, core/image_utils.py:24-26)
# --------------------------------------------------------------------------------------------
def to_uint8_rgb(rgb01: np.ndarray) -> np.ndarray:
    """``clip(round(x*255), 0, 255)`` with NumPy's round-half-to-even (core/image_utils.py:24-26)."""
    return np.clip(np.round(rgb01 * 255.0), 0, 255).astype(np.uint8)


def ply_bytes(xyz: np.ndarray, rgb_u8: np.ndarray) -> bytes:
    """Binary little-endian PLY, 15 bytes per vertex: 3 x f32 + 3 x u8 (core/writers.py:29-46)."""
    n = xyz.shape[0]
    head = ("ply\nformat binary_little_endian 1.0\n"
            f"element vertex {n}\n"
            "property float x\nproperty float y\nproperty float z\n"
            "property uchar red\nproperty uchar green\nproperty uchar blue\n"
            "end_header\n").encode("ascii")
    rec = np.zeros(n, dtype=np.dtype([("p", "<f4", 3), ("c", "u1", 3)]))
    rec["p"] = np.asarray(xyz, np.float32)
    rec["c"] = np.asarray(rgb_u8, np.uint8)
    return head + rec.tobytes()


def points3d_bin_bytes(xyz: np.ndarray, rgb_u8: np.ndarray, err: Optional[np.ndarray] = None) -> bytes:
    """Upstream's truncated COLMAP ``points3D.bin``: u64 count, then per point u64 id (1-based),
    3 x f64 xyz, 3 x u8 rgb, f64 error = 43 bytes, no track (core/writers.py:15-26)."""
    n = xyz.shape[0]
    if err is None:
        err = np.zeros((n,), np.float32)
    rec = np.zeros(n, dtype=np.dtype([("id", "<u8"), ("p", "<f8", 3), ("c", "u1", 3), ("e", "<f8")]))
    rec["id"] = np.arange(1, n + 1, dtype=np.uint64)
    rec["p"] = np.asarray(xyz).astype(np.float64)
    rec["c"] = np.asarray(rgb_u8, np.uint8)
    rec["e"] = np.asarray(err).astype(np.float64)
    return np.uint64(n).tobytes() + rec.tobytes()
