"""Pair scheduling: which cameras are references and which are their neighbours
(upstream core/selection.py:10-70).  Runs on the host: n ~ 200 cameras."""
from __future__ import annotations

from typing import Dict, List

import numpy as np


def select_cameras_kcenters(flat_poses: np.ndarray, k: int) -> List[int]:
    """Greedy k-centres on the z-scored 16-vector poses: start from the pose farthest from the mean,
    then repeatedly add the pose farthest from the chosen set.  Returns sorted indices."""
    X = np.asarray(flat_poses, dtype=np.float32)
    n = X.shape[0]
    k = max(1, min(int(k), n))
    Z = (X - X.mean(axis=0, keepdims=True)) / (X.std(axis=0, keepdims=True) + 1e-8)
    seed = int(np.argmax(np.einsum("nd,nd->n", Z, Z)))
    chosen = [seed]
    nearest = np.linalg.norm(Z - Z[seed], axis=1)
    nearest[seed] = -np.inf
    while len(chosen) < k:
        nxt = int(np.argmax(nearest))
        chosen.append(nxt)
        nearest = np.minimum(nearest, np.linalg.norm(Z - Z[nxt], axis=1))
        nearest[nxt] = -np.inf
    return sorted(chosen)


def select_cameras_by_visibility(rec, k: int) -> List[int]:
    """Greedy set cover over the sparse model's 3-D points (upstream core/selection.py:10-33): the first pick is the image with the most
    OBSERVATIONS (upstream's first scores are list lengths: a 3-D point observed twice in one image counts twice - found by
    tests/golden/check_oracle_fuzz.py), every later pick the image with the most distinct points not covered yet; ties go to the image that comes
    first in ``rec.images``; returns the picked image ids sorted.

    Same picks as upstream's loop, computed incrementally: upstream recomputes ``len(set(points) - covered)`` for every remaining image after
    every pick (O(picks x observations) set operations: minutes on a real scene's millions of observations); here each newly covered point
    decrements the gain of the images that observe it, once (O(observations) in all, NumPy)."""
    if not rec.points3D:
        raise ValueError("Visibility-based selection requires a sparse point cloud.")
    ids, lists = [], []
    for img in rec.images.values():
        fast = getattr(img, "observed_point3D_ids", None)
        obs = fast() if callable(fast) else np.asarray([p.point3D_id for p in img.points2D if p.has_point3D() and p.point3D_id != -1], np.int64)
        ids.append(img.image_id)
        lists.append(np.asarray(obs, np.int64).reshape(-1))
    n = len(ids)
    k = min(int(k), n)
    if k <= 0 or n == 0:
        return []
    uniq = [np.unique(a) for a in lists]
    all_pts = np.unique(np.concatenate(uniq)) if n else np.zeros(0, np.int64)
    dense = [np.searchsorted(all_pts, u) for u in uniq]                  # every image's distinct points as indices into all_pts
    # inverted index (point -> images observing it) in CSR form
    img_of = np.concatenate([np.full(d.size, i, np.int64) for i, d in enumerate(dense)]) if n else np.zeros(0, np.int64)
    pt_of = np.concatenate(dense) if n else np.zeros(0, np.int64)
    order = np.argsort(pt_of, kind="stable")
    img_sorted = img_of[order]
    starts = np.searchsorted(pt_of[order], np.arange(all_pts.size + 1))
    covered = np.zeros(all_pts.size, bool)
    alive = np.ones(n, bool)
    score = np.array([a.size for a in lists], np.int64)                  # first pick: observations, duplicates included
    gain = np.array([d.size for d in dense], np.int64)                   # afterwards: distinct points not covered yet
    picked: List[int] = []
    for it in range(k):
        cand = np.where(alive, score if it == 0 else gain, -1)
        best = int(np.argmax(cand))                                      # the first maximum: rec.images' order, like max() over the dict
        picked.append(ids[best])
        alive[best] = False
        new = dense[best][~covered[dense[best]]]
        if new.size:
            covered[new] = True
            # every image observing a newly covered point loses one uncovered point per such point
            lo, hi = starts[new], starts[new + 1]
            tot = int((hi - lo).sum())
            if tot:
                take = np.repeat(lo - np.concatenate([[0], np.cumsum(hi - lo)[:-1]]), hi - lo) + np.arange(tot)
                gain -= np.bincount(img_sorted[take], minlength=n)
    return sorted(picked)


def nearest_neighbors(flat_poses: np.ndarray, k: int) -> np.ndarray:
    """(n, k) indices of the k cameras with the closest flattened pose (Euclidean, f32, via
    ``torch.cdist`` + ``topk`` like upstream so ties resolve the same way)."""
    import torch
    M = torch.from_numpy(np.asarray(flat_poses).astype(np.float32))
    n = int(M.shape[0])
    if n <= 1:
        return np.empty((n, 0), dtype=np.int64)
    k = max(1, min(int(k), n - 1))
    with torch.no_grad():
        d = torch.cdist(M, M, p=2)
        d.fill_diagonal_(float("inf"))
        idx = torch.topk(d, k, largest=False, dim=1).indices
    return idx.cpu().numpy()
