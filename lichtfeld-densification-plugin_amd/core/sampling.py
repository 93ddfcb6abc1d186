"""Coverage sampling of the aggregated certainty map (upstream core/sampling.py:8-53).

HOST stage of the "sampled" (upstream-equivalent) mode: it consumes the ``best_cert`` map produced by
the HIP aggregate kernel and returns the flat cell indices the indexed kernel triangulates.  It uses
the same library entry points upstream uses for the parts whose results depend on the library
(legacy ``RandomState.choice``, unstable ``argsort``, torch's f32 ``sum``) so that, given the same
seed and reference order, the selection is the upstream selection.  The per-cell Python loop of the
tile-coverage pass (the dominant cost upstream, ~H*W iterations) is replaced by an equivalent
first-occurrence-per-bin computation.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch


_NORMALISER_CACHE: dict = {}


def upstream_weight_sum(cert_map, cap: float = 0.9, border: int = 2) -> float:
    """Upstream's normaliser of the sampling weights (core/sampling.py:23-31 there): the torch CPU f32 ``sum`` of the capped,
    border-masked certainty map - the same library calls on the same values (``clamp(max=cap)``, ``* inside.float()``, ``sum``), so the same
    rounding as upstream on this machine.  The 0 / 1 border mask is built once per grid size and the two elementwise steps write into a
    kept buffer: three torch operations per map instead of a dozen (each one wakes torch's whole thread pool: 0.6 ms per map on a
    128-thread host with the dozen; single-threaded passes are slower still there - profiles/r4/sampled_default_breakdown.txt)."""
    cert = torch.as_tensor(cert_map).detach().to("cpu", torch.float32)
    H, W = int(cert.shape[0]), int(cert.shape[1])
    key = (H, W, int(border))
    kept = _NORMALISER_CACHE.get(key)
    if kept is None:
        ys = torch.arange(H).view(H, 1)
        xs = torch.arange(W).view(1, W)
        inside = (xs >= border) & (xs <= W - 1 - border) & (ys >= border) & (ys <= H - 1 - border)
        if len(_NORMALISER_CACHE) > 8:
            _NORMALISER_CACHE.clear()
        kept = _NORMALISER_CACHE[key] = (inside.to(torch.float32), torch.empty((H, W), dtype=torch.float32))     # (one caller at a time: the driver thread)
    mask, buf = kept
    torch.clamp(cert, max=cap, out=buf)
    buf.mul_(mask)
    return float(buf.reshape(-1).sum())


def select_samples_with_coverage(cert_map, M: int, cap: float = 0.9, border: int = 2, tiles: int = 24,
                                 no_filter: bool = False, rng: Optional[np.random.RandomState] = None) -> np.ndarray:
    """Flat indices (int64) of the cells to triangulate.

    ``rng``: the legacy MT19937 stream to draw from; ``None`` uses NumPy's process-global one exactly
    like upstream (seeded by the pipeline with ``config.seed``)."""
    cert = torch.as_tensor(cert_map).detach().to("cpu", torch.float32)
    cert = torch.clamp(cert, max=cap)
    H, W = cert.shape
    M = int(M)
    if no_filter:
        flat = cert.reshape(-1).numpy()
        if flat.size == 0:
            return np.zeros((0,), dtype=np.int64)
        return np.argsort(-flat)[:min(M, flat.size)]

    ys = torch.arange(H).view(H, 1)
    xs = torch.arange(W).view(1, W)
    inside = (xs >= border) & (xs <= W - 1 - border) & (ys >= border) & (ys <= H - 1 - border)
    weights_t = (cert * inside.to(torch.float32)).reshape(-1)
    total = weights_t.sum()
    if not bool(total > 0):
        return np.zeros((0,), dtype=np.int64)
    weights = (weights_t / total).numpy()

    n_main = min(int(M * 0.85), weights.size)
    chooser = rng if rng is not None else np.random
    idx_main = chooser.choice(weights.size, size=n_main, replace=False, p=weights)

    # tile coverage: walking cells by descending weight, keep the first cell of every tile bin
    tile = max(1, W // tiles)
    bins = ((xs // tile) * 100000 + (ys // tile)).expand(H, W).reshape(-1).numpy()
    order = np.argsort(-weights)
    n_pos = int(np.count_nonzero(weights > 0))          # descending order: positives form a prefix
    budget = max(M - idx_main.size, 1)                   # the walk always admits its first pick
    _, first_at = np.unique(bins[order[:n_pos]], return_index=True)
    first_at.sort()
    idx_cov = order[first_at[:budget]]
    return np.unique(np.concatenate([idx_main, idx_cov.astype(np.int64)]))
