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
    """Greedy set cover over the sparse model's 3-D points (needs a pycolmap Reconstruction)."""
    if not rec.points3D:
        raise ValueError("Visibility-based selection requires a sparse point cloud.")
    observed = {img.image_id: [p.point3D_id for p in img.points2D if p.has_point3D() and p.point3D_id != -1] for img in rec.images.values()}
    seen_by: Dict[int, set] = {iid: set(pts) for iid, pts in observed.items()}
    k = min(k, len(seen_by))
    # upstream's first scores count OBSERVATIONS (a list: a 3-D point observed twice in one image counts twice), the later ones distinct
    # uncovered points (core/selection.py:14-33 upstream) - found by tests/golden/check_oracle_fuzz.py
    gain = {iid: len(pts) for iid, pts in observed.items()}
    covered: set = set()
    picked: List[int] = []
    for _ in range(k):
        if not gain:
            break
        best = max(gain, key=gain.get)
        picked.append(best)
        covered |= seen_by[best]
        del gain[best]
        for iid in gain:
            gain[iid] = len(seen_by[iid] - covered)
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
