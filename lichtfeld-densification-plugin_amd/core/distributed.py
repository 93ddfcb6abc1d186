"""Multi-GPU sharding of the dense-initialisation path (one process per GPU, torch.distributed).

The path shards by REFERENCE VIEW: the arg-max across neighbours couples the k pairs of one
reference, nothing couples two references.  References are dealt round-robin to the
ranks, every rank runs the whole per-reference path on its share with no collective, and there is
exactly one exchange step at the end, over RCCL/xGMI, of the survivors (28 B per point: xyz f32x3,
rgb f32x3, err f32), in one of two forms:

  * ``all_gather_by_reference``  - what BASELINE's north star names: counts, then ONE padded all-gather; every rank ends
    up with the whole cloud in global reference order;
  * ``gather_to_root_by_reference`` - when only rank 0 writes the file: counts, then every rank's records travel ONCE,
    point to point (grouped send / receive), straight to their reference-ordered place in rank 0's pre-sized buffer - no
    padding to the largest rank, no concatenation, no second index gather, and 1/world of the all-gather's traffic.

``ShardedPlyStream`` is the streamed writer of a sharded run (BASELINE config 5): every rank packs the PLY records of its
finished references on the device and rank 0 appends them to the output file in global reference order as they arrive.

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
    return torch.device("cpu") if "gloo" in _backend_name(dist, group) else t.device


def _backend_name(dist, group=None) -> str:
    try:
        return str(dist.get_backend(group)).lower()
    except Exception:
        return "nccl"


def _all_gather_rows(mine: torch.Tensor, world: int, dist, group=None) -> torch.Tensor:
    """``[world, rows, cols]`` from every rank's ``[rows, cols]`` with ONE collective into ONE pre-sized buffer
    (``all_gather_into_tensor``).  Which collective is used is decided by what the installed torch offers - the same answer on
    every rank - never by catching a failure: an error of the collective (a rank out of memory, a communicator fault)
    propagates instead of sending this rank into a different collective than its peers."""
    out = torch.empty((world,) + tuple(mine.shape), dtype=mine.dtype, device=mine.device)
    if hasattr(dist, "all_gather_into_tensor"):
        dist.all_gather_into_tensor(out.view(-1), mine.reshape(-1).contiguous(), group=group)
    else:
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
    dev = torch.device("cpu") if ("gloo" in _backend_name(dist, group) or device is None) else device
    t = torch.tensor([int(local_code)], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t.item())


def all_gather_by_reference(xyz: torch.Tensor, rgb: torch.Tensor, err: torch.Tensor,
                            ref_counts: Sequence[int], n_refs_global: int, dist, group=None
                            ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, np.ndarray]:
    """All-gather and restore the global reference order.

    ``ref_counts[i]`` = survivors of the i-th LOCAL reference (global position ``rank + i*world``),
    local points being stored reference after reference.  Returns the points ordered by global
    reference position and the per-reference counts ``(n_refs_global,)``.

    Two collectives (the per-reference counts, then the records padded to the largest rank in ONE ``all_gather_into_tensor``);
    the gathered block is then read ONCE: every reference's slice is copied from its rank's row to its place in the ordered
    result (n_refs_global slice copies - no concatenation of the trimmed rows, no index gather over the cloud)."""
    world = dist.get_world_size(group)
    home = xyz.device
    dev = _collective_device(xyz, dist, group)
    table = _reference_table(ref_counts, int(xyz.shape[0]), n_refs_global, dist, dev, group)       # [world, per_rank]
    per_rank_points = table.sum(axis=1)
    rows = max(int(per_rank_points.max()) if per_rank_points.size else 0, 1)
    gathered = _all_gather_rows(_pack(xyz.to(dev), rgb.to(dev), err.to(dev), rows), world, dist, group)     # [world, rows, 7]
    global_counts = np.array([table[g % world, g // world] for g in range(n_refs_global)], np.int64)
    offsets = np.concatenate([[0], np.cumsum(global_counts)])
    within = np.concatenate([np.zeros((world, 1), np.int64), np.cumsum(table, axis=1)], axis=1)
    out = torch.empty((int(offsets[-1]), 7), dtype=torch.float32, device=dev)
    for g in range(n_refs_global):
        r, i, n = g % world, g // world, int(global_counts[g])
        if n:
            out[int(offsets[g]):int(offsets[g]) + n] = gathered[r, int(within[r, i]):int(within[r, i]) + n]
    out = out.to(home)
    return out[:, 0:3].contiguous(), out[:, 3:6].contiguous(), out[:, 6].contiguous(), global_counts


def _reference_table(ref_counts: Sequence[int], n_local_points: int, n_refs_global: int, dist, dev, group=None):
    """Every rank's per-reference survivor counts: ``table[rank, i]`` = survivors of that rank's i-th local reference (global
    position ``rank + i * world``).  One small all-gather; checks that the local counts describe the local points."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    per_rank = (n_refs_global + world - 1) // world
    expected = len(shard_references(n_refs_global, rank, world))
    if len(ref_counts) != expected:
        raise ValueError(f"rank {rank} owns {expected} references, got {len(ref_counts)} counts")
    if int(sum(int(c) for c in ref_counts)) != int(n_local_points):
        raise ValueError("ref_counts do not add up to the number of local points")
    local = torch.zeros(per_rank, dtype=torch.int64)
    if len(ref_counts):
        local[:len(ref_counts)] = torch.as_tensor([int(c) for c in ref_counts], dtype=torch.int64)
    return _all_gather_rows(local.to(dev), world, dist, group).cpu().numpy()


def gather_to_root_by_reference(xyz: torch.Tensor, rgb: torch.Tensor, err: torch.Tensor, ref_counts: Sequence[int],
                                n_refs_global: int, dist, group=None, root: int = 0):
    """The exchange for runs whose cloud is consumed by ONE rank (the writer).  Counts first (every rank learns the global
    per-reference counts), then each rank's 28-byte records go point to point - one grouped batch of sends / receives, one
    message per reference - into rank ``root``'s buffer, which is sized exactly and filled at the reference-ordered offsets.

    Returns ``(xyz, rgb, err, global_counts)``: on ``root`` the whole cloud in global reference order (the 1-GPU sequence), on
    the other ranks their own shard, untouched.  Bytes on the wire: the survivors once (all-gather: world times, plus padding)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    home = xyz.device
    dev = _collective_device(xyz, dist, group)
    table = _reference_table(ref_counts, int(xyz.shape[0]), n_refs_global, dist, dev, group)       # [world, per_rank]
    global_counts = np.array([table[g % world, g // world] for g in range(n_refs_global)], np.int64)
    offsets = np.concatenate([[0], np.cumsum(global_counts)])
    within = np.concatenate([np.zeros((world, 1), np.int64), np.cumsum(table, axis=1)], axis=1)
    n_local = int(xyz.shape[0])
    local = torch.empty((n_local, 7), dtype=torch.float32, device=dev)        # the records as they travel: one pass over the shard
    if n_local:
        local[:, 0:3] = xyz.to(dev)
        local[:, 3:6] = rgb.to(dev)
        local[:, 6] = err.to(dev)
    ops = []
    if rank == root:
        buf = torch.empty((int(offsets[-1]), 7), dtype=torch.float32, device=dev)
        for g in range(n_refs_global):
            r, i, n = g % world, g // world, int(global_counts[g])
            if n == 0:
                continue
            dst = buf[int(offsets[g]):int(offsets[g]) + n]
            if r == root:
                dst.copy_(local[int(within[r, i]):int(within[r, i]) + n])
            else:
                ops.append(dist.P2POp(dist.irecv, dst, r, group))
    else:
        for i, n in enumerate(int(c) for c in ref_counts):
            if n:
                ops.append(dist.P2POp(dist.isend, local[int(within[rank, i]):int(within[rank, i]) + n], root, group))
    if ops:
        for work in dist.batch_isend_irecv(ops):
            work.wait()
    if rank != root:
        return xyz, rgb, err, global_counts
    buf = buf.to(home)
    return buf[:, 0:3].contiguous(), buf[:, 3:6].contiguous(), buf[:, 6].contiguous(), global_counts


class ShardedPlyStream:
    """Streamed PLY output of a sharded run.  Every rank calls ``push(local_index, body)`` with the device-packed 15-byte records
    (lfd_pack_ply) of each of its references that produced points, in local order, and ``finish()`` at the end.  Rank 0 owns the
    ``StreamedPlyWriter``: whenever it pushes one of its own references it first receives - point to point, a count then the
    records - every reference of the other ranks that precedes it in the global order, so the file grows in the 1-GPU sequence
    while the run proceeds.  The other ranks never wait: their sends are asynchronous and kept alive until ``finish``.
    A reference without points (skipped, failed, nothing survived) travels as a count of 0, which keeps the ranks in step; a rank
    that stops early still calls ``finish`` (the pipeline does so in its error path), which sends 0 for what is left."""

    def __init__(self, dist, n_refs_global: int, writer, device, group=None, root: int = 0):
        self.dist, self.group, self.root = dist, group, int(root)
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.n_refs = int(n_refs_global)
        self.writer = writer if self.rank == self.root else None
        self.dev = torch.device("cpu") if "gloo" in _backend_name(dist, group) else torch.device(device)
        self._next_local = 0           # non-root: first local reference not sent yet
        self._next_global = 0          # root: first global position not written yet
        self._pending = []             # non-root: (work, tensor) of sends in flight
        self._n_local = len(shard_references(self.n_refs, self.rank, self.world))
        self.finished = False

    # -- non-root ---------------------------------------------------------------------------------------------------------
    def _send(self, body: Optional[torch.Tensor]) -> None:
        n = 0 if body is None else int(body.numel())
        head = torch.tensor([n], dtype=torch.int64, device=self.dev)
        self._pending.append((self.dist.isend(head, self.root, group=self.group), head))
        if n:
            payload = body.to(self.dev).contiguous()
            self._pending.append((self.dist.isend(payload, self.root, group=self.group), payload))
        self._pending = [(w, t) for w, t in self._pending if not w.is_completed()]

    # -- root ---------------------------------------------------------------------------------------------------------------
    def _recv_into_file(self, src: int) -> None:
        head = torch.zeros(1, dtype=torch.int64, device=self.dev)
        self.dist.recv(head, src, group=self.group)
        n = int(head.item())
        if n > 0:
            payload = torch.empty(n, dtype=torch.uint8, device=self.dev)
            self.dist.recv(payload, src, group=self.group)
            if self.writer is not None:           # (no writer: the file could not be opened - the records are received and dropped)
                self.writer.append_packed(payload.cpu().numpy().tobytes())

    def _advance_to(self, global_pos: int) -> None:
        """Write every reference of the other ranks that precedes ``global_pos`` (own references before it had no points)."""
        while self._next_global < global_pos:
            owner = self._next_global % self.world
            if owner != self.root:
                self._recv_into_file(owner)
            self._next_global += 1

    def push(self, local_index: int, body) -> None:
        """``body``: uint8 tensor (device or host) or bytes: the 15-byte records of this rank's ``local_index``-th reference."""
        if self.finished:
            raise RuntimeError("push after finish")
        if self.rank == self.root:
            g = self.root + int(local_index) * self.world
            self._advance_to(g)
            if self.writer is not None:
                self.writer.append_packed(body if isinstance(body, (bytes, bytearray)) else body.cpu().numpy().tobytes())
            self._next_global = g + 1
            return
        if isinstance(body, (bytes, bytearray)):
            body = torch.frombuffer(bytearray(body), dtype=torch.uint8)
        while self._next_local < int(local_index):
            self._send(None)
            self._next_local += 1
        self._send(body)
        self._next_local = int(local_index) + 1

    def finish(self) -> None:
        """Flush: the root receives what is left, the others send 0 for the references they never pushed and wait for their sends."""
        if self.finished:
            return
        self.finished = True
        if self.rank == self.root:
            self._advance_to(self.n_refs)
            return
        while self._next_local < self._n_local:
            self._send(None)
            self._next_local += 1
        for w, _t in self._pending:
            w.wait()
        self._pending = []
