"""Multi-GPU sharding of the dense-initialisation path (one process per GPU, torch.distributed).

The path shards by REFERENCE VIEW: the arg-max across neighbours couples the k pairs of one
reference, nothing couples two references.  References are dealt round-robin to the
ranks, every rank runs the whole per-reference path on its share with no collective, and there is
exactly one exchange step at the end: a variable-length all-gather of the survivors
(28 B per point: xyz f32x3, rgb f32x3, err f32) over RCCL/xGMI.  Payloads are small (<= tens of MB
per rank), so the exchange is two collectives - counts, then one padded all-gather - rather than a
chain of point-to-point sends.

The upstream plugin has no multi-GPU code; the ordering rule below is this implementation's:
the gathered sequence is ordered by position in the global reference list, so 1-GPU and N-GPU runs
produce the same sequence.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch


def shard_references(n_refs: int, rank: int, world: int) -> List[int]:
    """Positions (in the global reference list) owned by ``rank``: round-robin, which balances the
    k-centres ordering of the list."""
    return list(range(rank, n_refs, world))


def _pack(xyz: torch.Tensor, rgb: torch.Tensor, err: torch.Tensor, rows: int) -> torch.Tensor:
    buf = torch.zeros((rows, 7), dtype=torch.float32, device=xyz.device)
    n = xyz.shape[0]
    if n:
        buf[:n, 0:3] = xyz
        buf[:n, 3:6] = rgb
        buf[:n, 6] = err
    return buf


def _collective_device(t: torch.Tensor, dist, group=None) -> torch.device:
    """RCCL ("nccl") moves device buffers; gloo (CPU tests, or several ranks sharing one GPU) host ones."""
    try:
        backend = str(dist.get_backend(group)).lower()
    except Exception:
        backend = "nccl"
    return torch.device("cpu") if "gloo" in backend else t.device


def _all_gather_rows(mine: torch.Tensor, world: int, dist, group=None) -> torch.Tensor:
    """``[world, rows, cols]`` from every rank's ``[rows, cols]`` with ONE collective into ONE pre-sized buffer
    (``all_gather_into_tensor``); backends without it take the list form."""
    out = torch.empty((world,) + tuple(mine.shape), dtype=mine.dtype, device=mine.device)
    try:
        dist.all_gather_into_tensor(out.view(-1), mine.reshape(-1), group=group)
    except (RuntimeError, NotImplementedError, AttributeError):
        dist.all_gather([out[r] for r in range(world)], mine, group=group)
    return out


def all_gather_points(xyz: torch.Tensor, rgb: torch.Tensor, err: torch.Tensor, dist, group=None
                      ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, List[int]]:
    """Concatenate every rank's survivors in rank order.  Returns (xyz, rgb, err, counts).

    Two collectives: the counts (8 B per rank), then the survivors packed 28 B per point and padded to the largest
    rank's count, gathered into one ``[world, rows, 7]`` buffer.  Under RCCL the tensors never leave the device."""
    world = dist.get_world_size(group)
    home = xyz.device
    dev = _collective_device(xyz, dist, group)
    xyz, rgb, err = xyz.to(dev), rgb.to(dev), err.to(dev)
    n_local = torch.tensor([xyz.shape[0]], dtype=torch.int64, device=dev)
    counts = [int(c) for c in _all_gather_rows(n_local, world, dist, group).reshape(-1).tolist()]
    rows = max(max(counts), 1)
    gathered = _all_gather_rows(_pack(xyz, rgb, err, rows), world, dist, group)
    cat = torch.cat([gathered[r, :c] for r, c in enumerate(counts)], dim=0) if sum(counts) else gathered[0, :0]
    cat = cat.to(home)
    return cat[:, 0:3].contiguous(), cat[:, 3:6].contiguous(), cat[:, 6].contiguous(), counts


def agree_on_status(local_code: int, dist, device=None, group=None) -> int:
    """MAX over the ranks of a small status code (0 = fine).  Called before the final exchange so that a rank that was
    cancelled or failed does not leave the others blocked in the collective: every rank learns the worst status and
    raises the same exception."""
    try:
        backend = str(dist.get_backend(group)).lower()
    except Exception:
        backend = "nccl"
    dev = torch.device("cpu") if ("gloo" in backend or device is None) else device
    t = torch.tensor([int(local_code)], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t.item())


def all_gather_by_reference(xyz: torch.Tensor, rgb: torch.Tensor, err: torch.Tensor,
                            ref_counts: Sequence[int], n_refs_global: int, dist, group=None
                            ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, np.ndarray]:
    """All-gather and restore the global reference order.

    ``ref_counts[i]`` = survivors of the i-th LOCAL reference (global position ``rank + i*world``),
    local points being stored reference after reference.  Returns the points ordered by global
    reference position and the per-reference counts ``(n_refs_global,)``."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    home = xyz.device
    dev = _collective_device(xyz, dist, group)
    per_rank = (n_refs_global + world - 1) // world
    local = torch.zeros(per_rank, dtype=torch.int64, device=dev)
    expected = len(shard_references(n_refs_global, rank, world))
    if len(ref_counts) != expected:
        raise ValueError(f"rank {rank} owns {expected} references, got {len(ref_counts)} counts")
    if len(ref_counts):
        local[:len(ref_counts)] = torch.as_tensor(list(ref_counts), dtype=torch.int64, device=dev)
    if int(local.sum().item()) != xyz.shape[0]:
        raise ValueError("ref_counts do not add up to the number of local points")
    table = _all_gather_rows(local, world, dist, group).cpu().numpy()                  # [world, per_rank]
    gx, gc, ge, counts = all_gather_points(xyz, rgb, err, dist, group)
    rank_base = np.concatenate([[0], np.cumsum(counts)])[:-1]
    within = np.concatenate([np.zeros((world, 1), np.int64), np.cumsum(table, axis=1)], axis=1)
    pieces, global_counts = [], np.zeros(n_refs_global, np.int64)
    for g in range(n_refs_global):
        r, i = g % world, g // world
        lo = rank_base[r] + within[r, i]
        hi = rank_base[r] + within[r, i + 1]
        global_counts[g] = hi - lo
        if hi > lo:
            pieces.append(torch.arange(lo, hi, device=dev))
    if pieces:
        order = torch.cat(pieces).to(gx.device)
        gx, gc, ge = gx[order], gc[order], ge[order]
    return gx.to(home), gc.to(home), ge.to(home), global_counts
