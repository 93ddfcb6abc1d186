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

``OverlappedExchange`` is the same exchange in ROUNDS that run beside the compute: every ``refs_per_round`` local references the
finished references' records are handed to an asynchronous collective (all-gather, or gather to the writer rank) while the next batch
computes; the result is the same ordered sequence.  When the consumer is the PLY writer the 15-byte device-packed records travel
instead of the 28-byte rows.

``plan_replication`` / ``split_replicated``: recompute instead of communicate.  One GPU triangulates a reference faster than its survivors cross an
xGMI link, so a sharded dense run is bound by the link; the references of a scene are therefore split into a SHARDED prefix (dealt over the ranks,
exchanged) and a REPLICATED suffix (computed by every rank that needs the cloud, never sent), sized so that the redundant compute and the exchange of
the rest take the same time.  With a matcher in the loop (tens of ms per pair) the plan is "shard everything"; for the bare hot path it is not.

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


def split_replicated(n_refs: int, n_replicated: int, rank: int, world: int, replicas_here: bool = True) -> Tuple[List[int], int]:
    """Positions this rank processes when the LAST ``n_replicated`` references of the global list are computed by every rank instead of
    exchanged: its round-robin share of the sharded prefix ``[0, n_sharded)`` followed by the whole replicated suffix (``replicas_here=False``:
    a rank that does not consume the cloud - ``gather_to_root``, not the root - skips the suffix).  Returns ``(positions, n_sharded)``."""
    n_rep = max(0, min(int(n_replicated), int(n_refs)))
    n_sh = int(n_refs) - n_rep
    mine = shard_references(n_sh, rank, world)
    if replicas_here:
        mine = mine + list(range(n_sh, int(n_refs)))
    return mine, n_sh


def plan_replication(n_refs: int, world: int, ref_ms: float, ref_bytes: float, *, launch_ms: float = 0.0, link_gbps: float = 122.0,
                     collective_ms: float = 0.05, copy_gbps: float = 2000.0) -> dict:
    """How many of a scene's ``n_refs`` references to REPLICATE (compute on every rank) instead of exchanging them, for an all-gather (or a
    gather to one rank) of the survivors over point-to-point links.

    Cost model (one step = the scene once; the sharded part is launched first, its rounds travel while the replicated part computes):
        compute(n)   = launch_ms + ref_ms * n                              one rank's launches over n references
        exchange(m)  = collective_ms + m * ref_bytes / link                one rank's shard of m references crossing ONE link - what a direct
                                                                           all-gather and a gather need, every peer on its own link at once
        merge(n)     = n * ref_bytes / copy                                the gathered records copied to their ordered place
        T(n_sh)      = compute(ceil(n_sh / world)) + max(compute(n_refs - n_sh), exchange(ceil(n_sh / world))) + merge(n_sh)
    Returns the ``n_sh`` of the smallest T (ties: the larger sharded part - less redundant work) as
    ``{"n_sharded", "n_replicated", "step_ms", "pure_sharding_ms", "single_rank_ms", "inputs"}``.  ``ref_ms`` is the per-reference cost of
    EVERYTHING a rank does for a reference (with a matcher in the loop: tens of ms - the plan is then n_replicated = 0); ``link_gbps`` the
    measured all-gather bandwidth per peer.  world == 1: nothing is exchanged, nothing replicated."""
    n_refs, world = int(n_refs), max(1, int(world))
    link = max(float(link_gbps), 1e-6) * 1e6          # bytes per ms
    copy = max(float(copy_gbps), 1e-6) * 1e6

    def compute(n):
        return (float(launch_ms) + float(ref_ms) * n) if n > 0 else 0.0

    def step(n_sh):
        m = -(-n_sh // world)
        ex = (float(collective_ms) + m * float(ref_bytes) / link) if n_sh > 0 else 0.0
        return compute(m) + max(compute(n_refs - n_sh), ex) + n_sh * float(ref_bytes) / copy

    inputs = {"n_refs": n_refs, "world": world, "ref_ms": float(ref_ms), "ref_bytes": float(ref_bytes), "launch_ms": float(launch_ms),
              "link_gbps": float(link_gbps), "collective_ms": float(collective_ms), "copy_gbps": float(copy_gbps)}
    single = compute(n_refs)
    if world == 1 or n_refs == 0:
        return {"n_sharded": n_refs, "n_replicated": 0, "step_ms": single, "pure_sharding_ms": single, "single_rank_ms": single, "inputs": inputs}
    best = min(range(n_refs + 1), key=lambda n_sh: (round(step(n_sh), 9), -n_sh))
    return {"n_sharded": best, "n_replicated": n_refs - best, "step_ms": step(best), "pure_sharding_ms": step(n_refs), "single_rank_ms": single,
            "inputs": inputs}


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
    """Backend of the (initialised) process group.  An error of ``get_backend`` - a group that was never initialised, a rank that is not
    a member - propagates: answering "nccl" for it would send host tensors down the device branch instead of failing."""
    return str(dist.get_backend(group)).lower()


def _global_rank(dist, group, r: int) -> int:
    """Point-to-point calls take GLOBAL ranks as peers even when ``group=`` is given; ``r`` is a rank inside ``group``."""
    return int(r) if group is None else int(dist.get_global_rank(group, int(r)))


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
                ops.append(dist.P2POp(dist.irecv, dst, _global_rank(dist, group, r), group))
    else:
        for i, n in enumerate(int(c) for c in ref_counts):
            if n:
                ops.append(dist.P2POp(dist.isend, local[int(within[rank, i]):int(within[rank, i]) + n], _global_rank(dist, group, root), group))
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
    while the run proceeds.  The other ranks do not wait on the HOST: their sends are asynchronous and kept alive until ``finish``.
    On the DEVICE they are coupled to rank 0 under RCCL: a point-to-point send is a rendezvous kernel on the communicator's stream that
    spins until rank 0 posts the matching receive - which it does when it reaches its own reference at that global position - so one
    header and one payload send per reference queue up there, and anything that synchronises the whole device on a non-root rank
    (a ``hipFree`` behind a reallocation, ``torch.cuda.empty_cache``) waits until rank 0 has caught up.  (Under gloo the sends are
    buffered by the transport.)
    A reference without points (skipped, failed, nothing survived) travels as a count of 0, which keeps the ranks in step; a rank
    that stops early still calls ``finish`` (the pipeline does so in its error path), which sends 0 for what is left.
    A writer that fails on rank 0 (disk full, I/O error) does not stop the protocol: the first error is kept, the writer is dropped,
    the remaining payloads are received and discarded, and ``finish`` raises the kept error once everything has been drained - the other
    ranks' sends complete and the run's status agreement reports the failure."""

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
        self.writer_error: Optional[BaseException] = None      # root: first failure of the writer (re-raised by finish() after the drain)
        self._root_global = _global_rank(dist, group, self.root)

    # -- non-root ---------------------------------------------------------------------------------------------------------
    def _send(self, body: Optional[torch.Tensor]) -> None:
        n = 0 if body is None else int(body.numel())
        head = torch.tensor([n], dtype=torch.int64, device=self.dev)
        self._pending.append((self.dist.isend(head, self._root_global, group=self.group), head))
        if n:
            payload = body.to(self.dev).contiguous()
            self._pending.append((self.dist.isend(payload, self._root_global, group=self.group), payload))
        self._pending = [(w, t) for w, t in self._pending if not w.is_completed()]

    # -- root ---------------------------------------------------------------------------------------------------------------
    def _write(self, data: bytes) -> None:
        """Append to the file; the FIRST failure is kept and the writer dropped - receiving goes on, so that the protocol completes."""
        if self.writer is None:                   # (no writer: the file could not be opened, or an earlier append failed - the records are dropped)
            return
        try:
            self.writer.append_packed(data)
        except BaseException as exc:              # noqa: BLE001 - re-raised by finish()
            self.writer_error = exc
            self.writer = None

    def _recv_into_file(self, src: int) -> None:
        peer = _global_rank(self.dist, self.group, src)
        head = torch.zeros(1, dtype=torch.int64, device=self.dev)
        self.dist.recv(head, peer, group=self.group)
        n = int(head.item())
        if n > 0:
            payload = torch.empty(n, dtype=torch.uint8, device=self.dev)
            self.dist.recv(payload, peer, group=self.group)
            self._write(payload.cpu().numpy().tobytes())

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
            self._write(body if isinstance(body, (bytes, bytearray)) else body.cpu().numpy().tobytes())
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
            self._advance_to(self.n_refs)         # drains every peer even when the writer has failed
            if self.writer_error is not None:
                raise self.writer_error
            return
        while self._next_local < self._n_local:
            self._send(None)
            self._next_local += 1
        for w, _t in self._pending:
            w.wait()
        self._pending = []


RECORD_F32 = "f32"     # 28-byte rows: xyz f32 x 3, rgb f32 x 3, err f32 (what PipelineResult holds)
RECORD_PLY = "ply"     # 15-byte PLY vertex records packed on the device (lfd_pack_ply): xyz f32 LE x 3, rgb u8 x 3 - for runs whose consumer is the writer


class OverlappedExchange:
    """The exchange of a sharded run in ROUNDS that overlap the compute (no upstream counterpart: upstream has no multi-GPU code).

    Local references are taken in rounds of ``refs_per_round`` (round c = local references [c*B, (c+1)*B), global positions
    ``rank + i*world``).  ``push(local_index, records)`` hands over one finished reference; when a push (or ``finish``) moves past the end of
    a round, the round is CLOSED: its per-reference counts go into a small asynchronous all-gather, and the round BEFORE it - whose counts
    have long arrived - sends its records, padded to that round's largest rank, as one asynchronous collective (``all_gather``: every rank
    receives every rank's records; ``gather_to_root``: only rank ``root`` does).  Nothing on the host waits for a collective until ``finish``,
    and the collectives run on the communicator's own stream beside the kernels of the next batch (under gloo - CPU tests, ranks sharing a
    GPU - the records cross to host tensors first).  Every rank runs the same number of rounds (ranks that own fewer references close empty
    ones in ``finish``), so the collectives stay matched whatever a rank skipped.

    ``record``: RECORD_F32 - ``records`` is an (n, 7) float32 tensor; RECORD_PLY - an (n * 15,) uint8 tensor of device-packed PLY records.
    ``finish()`` returns ``(records, counts)``: the records of ALL references in global reference order - the 1-rank sequence - as one tensor
    of the same kind ((N, 7) float32 / (N * 15,) uint8; on every rank for ``all_gather``, on ``root`` for ``gather_to_root``, where the other
    ranks get their own shard back) and the (n_refs_global,) survivor counts.

    Nothing is copied that need not be: records pushed as consecutive views of one buffer (what a launch of the fused kernel leaves: the references
    of a batch one after the other) travel from where they are - no concatenation - and, when that buffer has room for the round's padded size
    (it has: a launch's buffer holds H * W records per reference), without a padded copy either; the receivers never look beyond the counts.
    ``push_many`` hands over such a run of references in one call.  The gathered blocks reach their ordered places in ONE launch per exchange
    (``lfd_copy_segments`` through ``hip_backend.copy_segments``; per-reference tensor copies on host tensors).

    ``form="counts_only"``: nothing but the counts travels - every rank keeps its own shard and learns, round by round, where its references
    sit in the global sequence (``_round_known`` is called with each round's table: ``SharedFilePlyStream`` writes its byte ranges of the
    output file from there).  ``finish()`` then returns this rank's own records and the global counts."""

    def __init__(self, dist, n_refs_global: int, refs_per_round: int, device, form: str = "all_gather", record: str = RECORD_F32,
                 group=None, root: int = 0, eager: bool = False, dest: Optional[torch.Tensor] = None):
        """``eager``: a round's records leave as soon as the round is complete - its counts are waited for on the spot (a host wait of one small
        collective) instead of a round later; for a caller whose next launch is already queued (bench.py), so that the records travel while it
        runs.  ``dest``: a (rows, cols) tensor on the exchange's device with room for every record: the ordered records are placed into it FROM
        ROW 0 as the rounds complete (each round's placement queued behind its collective) instead of all at the end into a new tensor; ``finish``
        then returns ``dest[:total]``."""
        if form not in ("all_gather", "gather_to_root", "counts_only"):
            raise ValueError("form must be 'all_gather', 'gather_to_root' or 'counts_only'")
        if record not in (RECORD_F32, RECORD_PLY):
            raise ValueError("record must be 'f32' or 'ply'")
        self.dist, self.group, self.root = dist, group, int(root)
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.n_refs = int(n_refs_global)
        self.B = max(1, int(refs_per_round))
        self.form, self.record = form, record
        self.cols, self.dtype = (7, torch.float32) if record == RECORD_F32 else (15, torch.uint8)
        self.home = torch.device(device)
        self.dev = torch.device("cpu") if "gloo" in _backend_name(dist, group) else self.home
        self.n_local = len(shard_references(self.n_refs, self.rank, self.world))
        n_local_max = (self.n_refs + self.world - 1) // self.world
        self.n_rounds = (n_local_max + self.B - 1) // self.B
        self.eager = bool(eager)
        self._dest = dest
        if dest is not None and (dest.dim() != 2 or dest.shape[1] != self.cols or dest.dtype != self.dtype or not dest.is_contiguous()
                                 or dest.device != self.dev):
            raise ValueError(f"dest must be a contiguous (rows, {self.cols}) {self.dtype} tensor on {self.dev}")
        self._placed = 0                   # rounds whose records are at their ordered place in dest already
        self._placed_rows = 0
        self._block: dict = {}             # round -> (records of the whole round as one view, per-reference counts): push_many's fast path
        self._open: dict = {}              # local index -> records of the round being filled
        self._cur = 0                      # round being filled
        self._rounds: list = []            # closed rounds: dict(counts_out, counts_work, payload, n_local, work, gathered, table)
        self._last_local = -1
        self.finished = False
        self.bytes_sent = 0                # this rank's padded payload bytes handed to collectives (bench)

    # ---------------------------------------------------------------------------------------------------------------------------------
    def push(self, local_index: int, records: Optional[torch.Tensor]) -> None:
        """Records of this rank's ``local_index``-th reference (None / empty: no survivors); local indices must increase."""
        if self.finished:
            raise RuntimeError("push after finish")
        li = int(local_index)
        if li <= self._last_local or li >= self.n_local:
            raise ValueError(f"local reference {li} out of order or out of range (last {self._last_local}, owned {self.n_local})")
        self._last_local = li
        while self._cur < li // self.B:
            self._close_round()
        if records is not None and records.numel():
            rec = records.reshape(-1, self.cols)
            if rec.dtype != self.dtype:
                raise ValueError(f"records must be {self.dtype} with {self.cols} columns per point")
            self._open[li] = rec
        if self._cur < self.n_rounds and li == min((self._cur + 1) * self.B, self.n_local) - 1:
            self._close_round()                # the round is complete: its counts leave now, not when the next round's first reference arrives

    def push_many(self, first_local_index: int, records: torch.Tensor, counts: Sequence[int]) -> None:
        """``len(counts)`` consecutive local references at once: ``records`` holds their records one after the other (``counts[i]`` each) - the
        buffer a launch of the fused kernel wrote.  The same as one ``push`` per reference with the corresponding slice (views: nothing is copied);
        a run that is exactly one round is taken over as ONE view."""
        rec = records.reshape(-1, self.cols)
        first, m = int(first_local_index), len(counts)
        c = first // self.B if self.B else 0
        whole_round = (m > 0 and first == c * self.B and first + m == min((c + 1) * self.B, self.n_local) and first == self._last_local + 1
                       and not self.finished and rec.dtype == self.dtype)
        if not whole_round:
            lo = 0
            for i, n in enumerate(counts):
                n = int(n)
                self.push(first + i, rec[lo:lo + n] if n else None)
                lo += n
            return
        while self._cur < c:
            self._close_round()
        cnt = np.zeros(self.B, np.int64)
        cnt[:m] = np.asarray(counts, np.int64)
        self._block[c] = (rec[:int(cnt.sum())], cnt)
        self._last_local = first + m - 1
        self._close_round()

    def _joined(self, parts):
        """One (rows, cols) tensor of the parts in order: a VIEW when they already lie one after the other in one buffer, else a concatenation."""
        if len(parts) == 1:
            return parts[0]
        es = parts[0].element_size()
        run = all(p.is_contiguous() for p in parts) and all(
            parts[i + 1].untyped_storage().data_ptr() == parts[0].untyped_storage().data_ptr()
            and parts[i + 1].data_ptr() == parts[i].data_ptr() + parts[i].numel() * es for i in range(len(parts) - 1))
        if run:
            rows = sum(int(p.shape[0]) for p in parts)
            return torch.as_strided(parts[0], (rows, self.cols), (self.cols, 1))
        return torch.cat(parts, 0)

    def _close_round(self) -> None:
        c = self._cur
        blk = self._block.pop(c, None)
        if blk is not None:
            payload, counts = blk[0], torch.from_numpy(blk[1])
        else:
            counts = torch.zeros(self.B, dtype=torch.int64)
            parts = []
            for j in range(self.B):
                rec = self._open.pop(c * self.B + j, None)
                if rec is not None:
                    counts[j] = rec.shape[0]
                    parts.append(rec)
            payload = self._joined(parts) if parts else torch.empty((0, self.cols), dtype=self.dtype, device=self.home)
        cnt_in = counts.to(self.dev)
        cnt_out = torch.empty(self.world * self.B, dtype=torch.int64, device=self.dev)
        work = self.dist.all_gather_into_tensor(cnt_out, cnt_in, group=self.group, async_op=True)
        self._rounds.append(dict(counts_in=cnt_in, counts_out=cnt_out, counts_work=work, payload=payload, n_local=int(counts.sum()),
                                 local_counts=counts, work=None, gathered=None, table=None, padded=None))
        self._cur += 1
        if self.eager:
            self._send_round(c)               # (waits for the counts it has just sent off: the caller's next launch is already queued)
        elif c >= 1:
            self._send_round(c - 1)           # its counts were gathered a whole round ago

    def _round_segments(self, c: int, base_row: int, row_bytes: int):
        """(source byte offset in the round's gathered block, destination byte offset, bytes) of every reference of round ``c``; the round's
        references are consecutive in the global order (local index first, then rank), the first of them at ``base_row``."""
        st = self._rounds[c]
        table = st["table"]
        within = np.concatenate([np.zeros((self.world, 1), np.int64), np.cumsum(table, axis=1)], axis=1)
        rows = int(st["gathered"].shape[1])
        segs, at = [], int(base_row)
        for j in range(self.B):
            for r in range(self.world):
                n = int(table[r, j])
                if n and r + (c * self.B + j) * self.world < self.n_refs:
                    segs.append(((r * rows + int(within[r, j])) * row_bytes, at * row_bytes, n * row_bytes))
                    at += n
        return segs

    def _place_round(self, c: int) -> None:
        """Round ``c``'s records to their ordered place in ``dest`` (queued behind the round's collective; rounds are placed in order)."""
        st = self._rounds[c]
        if st["gathered"] is not None:
            st["work"].wait()
            self._place(st["gathered"], self._dest, self._round_segments(c, self._placed_rows, self.cols * self._dest.element_size()))
        self._placed_rows += int(st["table"].sum())
        self._placed += 1

    def _send_round(self, c: int) -> None:
        st = self._rounds[c]
        if st.get("sent"):
            return
        st["sent"] = True
        st["counts_work"].wait()
        # the gathered counts as a host table: a few dozen integers (under RCCL their copy is ordered behind the collective by wait())
        table = st["counts_out"].cpu().view(self.world, self.B).numpy().copy()
        st["table"] = table
        self._round_known(c, table, st)
        rows = int(table.sum(axis=1).max())
        early = self._dest is not None and (self.form == "all_gather" or (self.form == "gather_to_root" and self.rank == self.root))
        if early:
            if int(self._placed_rows + sum(int(self._rounds[i]["table"].sum()) for i in range(self._placed, c + 1))) > int(self._dest.shape[0]):
                raise ValueError("dest has no room for the records gathered so far")
            while self._placed < c:                 # the rounds before this one: their collectives were launched a round ago
                self._place_round(self._placed)
        if rows == 0 or self.form == "counts_only":
            return
        pay = st["payload"]
        room = (pay.untyped_storage().nbytes() - pay.storage_offset() * pay.element_size()) // (self.cols * pay.element_size()) if st["n_local"] else 0
        if st["n_local"] and pay.device == self.dev and pay.is_contiguous() and room >= rows:
            # the records' own buffer has room for the round's padded size: they travel from where they are (what lies behind them is
            # whatever the buffer held - the receivers never look beyond the counts)
            padded = torch.as_strided(pay, (rows, self.cols), (self.cols, 1))
        else:
            padded = torch.zeros((rows, self.cols), dtype=self.dtype, device=self.dev)
            if st["n_local"]:
                padded[:st["n_local"]] = pay.to(self.dev)
        st["padded"] = padded
        self.bytes_sent += padded.numel() * padded.element_size()
        if self.form == "all_gather":
            out = torch.empty((self.world, rows, self.cols), dtype=self.dtype, device=self.dev)
            st["gathered"] = out
            st["work"] = self.dist.all_gather_into_tensor(out.view(-1), padded.view(-1), group=self.group, async_op=True)
        else:
            if self.rank == self.root:
                out = torch.empty((self.world, rows, self.cols), dtype=self.dtype, device=self.dev)
                st["gathered"] = out
                st["work"] = self.dist.gather(padded, [out[r] for r in range(self.world)], dst=_global_rank(self.dist, self.group, self.root),
                                              group=self.group, async_op=True)
            else:
                st["work"] = self.dist.gather(padded, None, dst=_global_rank(self.dist, self.group, self.root), group=self.group, async_op=True)

    def _round_known(self, c: int, table: np.ndarray, st: dict) -> None:
        """Hook: the counts of round ``c`` of every rank (``table[rank, j]``) have arrived (called in round order)."""

    def finish(self, place=None, concat: bool = True):
        """Close what is left (every rank runs ``n_rounds`` rounds), wait for the collectives, return the ordered records and counts.
        ``concat=False``: where this rank only gets its own shard back (``counts_only``, or ``gather_to_root`` off the root) the records are
        returned as the list of the rounds' payloads, where they are, instead of one concatenated copy.
        ``place(n_rows)``: optional, returns the (n_rows, cols) tensor on this exchange's device the ordered records are to be written INTO (a
        slice of a larger buffer: the caller's own records sit next to it) instead of a new allocation; used where this rank assembles the cloud."""
        if self.finished:
            raise RuntimeError("finish called twice")
        self.finished = True
        while self._cur < self.n_rounds:
            self._close_round()
        if self.n_rounds:
            self._send_round(self.n_rounds - 1)
        for st in self._rounds:
            if st["work"] is not None:
                st["work"].wait()
        global_counts = np.zeros(self.n_refs, np.int64)
        for c, st in enumerate(self._rounds):
            for r in range(self.world):
                for j in range(self.B):
                    g = r + (c * self.B + j) * self.world
                    if g < self.n_refs:
                        global_counts[g] = st["table"][r, j]
        have_all = self.form == "all_gather" or (self.form == "gather_to_root" and self.rank == self.root)
        if not have_all:            # gather_to_root, another rank: its own shard, untouched
            own = [st["payload"] for st in self._rounds if st["n_local"]]
            if not concat:
                return [self._shape(p_.to(self.home)) for p_ in own], global_counts
            mine = (torch.cat(own, 0) if len(own) > 1 else own[0]) if own else torch.empty((0, self.cols), dtype=self.dtype, device=self.home)
            return self._shape(mine.to(self.home)), global_counts
        if self._dest is not None:
            while self._placed < len(self._rounds):
                self._place_round(self._placed)
            return self._shape(self._dest[:self._placed_rows].to(self.home)), global_counts
        offsets = np.concatenate([[0], np.cumsum(global_counts)])
        if place is not None:
            out = place(int(offsets[-1]))
            if tuple(out.shape) != (int(offsets[-1]), self.cols) or out.dtype != self.dtype or out.device != self.dev:
                raise ValueError(f"place() must return a ({int(offsets[-1])}, {self.cols}) {self.dtype} tensor on {self.dev}")
        else:
            out = torch.empty((int(offsets[-1]), self.cols), dtype=self.dtype, device=self.dev)
        row_bytes = self.cols * out.element_size()
        base = 0
        for c, st in enumerate(self._rounds):
            if st["gathered"] is not None:
                self._place(st["gathered"], out, self._round_segments(c, base, row_bytes))
            base += int(st["table"].sum())
        return self._shape(out.to(self.home)), global_counts

    def _place(self, src: torch.Tensor, dst: torch.Tensor, segs) -> None:
        """dst.bytes[d : d + n] = src.bytes[s : s + n] for every (s, d, n): one launch on the device, tensor copies on the host."""
        if not segs:
            return
        if src.is_cuda:
            from . import hip_backend as hb
            # (dst may be a view into a larger buffer - finish(place=...): offsets are relative to ITS first byte)
            hb.copy_segments(src, dst, segs)
            return
        sb, db = src.reshape(-1).view(torch.uint8), dst.reshape(-1).view(torch.uint8)
        for s_, d_, n_ in segs:
            db[d_:d_ + n_] = sb[s_:s_ + n_]

    def _shape(self, t: torch.Tensor) -> torch.Tensor:
        return t if self.record == RECORD_F32 else t.reshape(-1)


def rows_from_points(xyz: torch.Tensor, rgb: torch.Tensor, err: torch.Tensor) -> torch.Tensor:
    """(n, 7) float32 rows [x y z r g b err] of one reference, as RECORD_F32 travels."""
    n = int(xyz.shape[0])
    rows = torch.empty((n, 7), dtype=torch.float32, device=xyz.device)
    if n:
        rows[:, 0:3] = xyz
        rows[:, 3:6] = rgb
        rows[:, 6] = err
    return rows


def points_from_ply_records(records: torch.Tensor):
    """(xyz f32 (n,3), rgb f32 (n,3) = u8 / 255) out of 15-byte PLY records: positions exactly as computed, colours as the writer
    quantised them (re-quantising them gives the same bytes: |k/255 * 255 - k| < 1e-4)."""
    rec = records.reshape(-1, 15)
    xyz = rec[:, :12].contiguous().view(torch.float32).reshape(-1, 3)
    rgb = rec[:, 12:15].to(torch.float32) / 255.0
    return xyz, rgb


class SharedFilePlyStream(OverlappedExchange):
    """The exchange-free streamed output of a sharded run on ONE node (a file system every rank sees): only the per-reference COUNTS cross a
    link.  Every rank packs its finished references' 15-byte PLY records on its own GPU; when the counts of a round are known (one small
    asynchronous all-gather per round, a round behind the compute like every OverlappedExchange) it knows the byte offset of each of its
    references in the global reference order and writes them there itself (``os.pwrite``) - N ranks copy over N PCIe links and write N
    disjoint byte ranges of one file; no survivor crosses xGMI.  Rank ``root`` creates the file (the fixed-width header of
    ``StreamedPlyWriter``) before anybody writes and patches the vertex count into it at the end.  File bytes = the 1-rank ``write_ply``
    output (``tests/test_distributed_cpu.py::test_shared_file_stream...``).  No upstream counterpart."""

    def __init__(self, dist, n_refs_global: int, refs_per_round: int, path: str, device, group=None, root: int = 0):
        super().__init__(dist, n_refs_global, refs_per_round, device, form="counts_only", record=RECORD_PLY, group=group, root=root)
        import os
        from .writers import ensure_dir, streamed_ply_header
        self.path = path
        self._base = 0                    # records of the rounds already placed
        self._data_offset = len(streamed_ply_header(0))
        self.error: Optional[BaseException] = None
        self._fd = None
        if self.rank == self.root:
            try:
                ensure_dir(path)
                with open(path, "wb") as fh:
                    fh.write(streamed_ply_header(0))
            except BaseException as exc:               # noqa: BLE001 - kept: the collectives below must still match on every rank
                self.error = exc
        # the file exists before anybody opens it - or the root could not create it, and then NOBODY opens whatever stale file may sit at that
        # path: the root's status travels in the collective that doubles as the barrier
        made = torch.tensor([0 if self.error is None else 1], dtype=torch.int32, device=self.dev)
        dist.all_reduce(made, op=dist.ReduceOp.MAX, group=group)
        if int(made.item()):
            self.error = self.error or RuntimeError(f"the writer rank could not create {path}")
            return
        try:
            self._fd = os.open(path, os.O_RDWR)
        except BaseException as exc:                   # noqa: BLE001
            self.error = self.error or exc

    def _round_known(self, c: int, table: np.ndarray, st: dict) -> None:
        import os
        # global order inside a round: local index first, then rank (g = rank + local * world)
        order = [(j, r) for j in range(self.B) for r in range(self.world)]
        pos = self._base
        mine = {}
        for j, r in order:
            if r == self.rank:
                mine[j] = pos
            pos += int(table[r, j])
        self._base = pos
        if self._fd is None or self.error is not None or not st["n_local"]:
            return
        try:
            payload = st["payload"].reshape(-1).cpu().numpy()
            off = 0
            for j in range(self.B):
                n = int(st["local_counts"][j])
                if n:
                    view = memoryview(payload[off * 15:(off + n) * 15])
                    at, done = self._data_offset + 15 * mine[j], 0
                    while done < len(view):            # pwrite may write less than asked (signals, 2 GiB limit)
                        w = os.pwrite(self._fd, view[done:], at + done)
                        if w <= 0:
                            raise OSError(f"pwrite wrote {w} bytes at offset {at + done} of {self.path}")
                        done += w
                    off += n
        except BaseException as exc:                   # noqa: BLE001 - the rounds go on (the counts must stay matched); raised by finish()
            self.error = exc

    def finish(self):
        import os
        # (concat=False: this rank's records have been written where they belong; nothing concatenates them into a second copy of the shard)
        recs, counts = super().finish(concat=False)
        if self._fd is not None:
            try:
                os.fsync(self._fd)
            except OSError:
                pass
            os.close(self._fd)
            self._fd = None
        # every byte range is in place - or somebody failed: the ranks agree (one small all-reduce instead of a barrier), and the vertex count is
        # patched into the header only for a complete file (a failed run leaves "element vertex 0": no reader takes the holes for points)
        flag = torch.tensor([0 if self.error is None else 1], dtype=torch.int32, device=self.dev)
        self.dist.all_reduce(flag, op=self.dist.ReduceOp.MAX, group=self.group)
        if int(flag.item()) and self.error is None:
            self.error = RuntimeError(f"another rank failed to write its records into {self.path}")
        if self.rank == self.root and self.error is None:
            from .writers import streamed_ply_header
            try:
                with open(self.path, "r+b") as fh:     # the vertex count, patched into the fixed-width header
                    fh.write(streamed_ply_header(int(counts.sum())))
                    if int(counts.sum()) == 0:
                        fh.truncate(self._data_offset)
            except BaseException as exc:               # noqa: BLE001
                self.error = exc
        if self.error is not None:
            raise self.error
        return recs, counts


# ---- dry run: what a sharded run WILL do, without touching a GPU or a communicator -------------------------------------------------------------
def exchange_schedule(n_refs: int, world: int, refs_per_round: int, form: str = "all_gather", record: str = RECORD_F32, counts: Optional[Sequence[int]] = None,
                      eager: bool = False, root: int = 0) -> dict:
    """The collectives an ``OverlappedExchange(n_refs, refs_per_round, form, record)`` issues, in issue order - the SAME list on every rank, which is
    the point: RCCL matches (and executes) the operations of a communicator in issue order, so a rank that would issue another sequence hangs the
    job.  ``counts``: survivors per GLOBAL reference position (None: sizes of the record collectives are left open).  Returns

        {"rounds": n, "refs_per_round": B, "ranks": [{"rank", "positions", "rounds": [[positions of round c], ...]}, ...],
         "collectives": [{"op", "what": "counts" | "records", "round", "numel_in", "numel_out" (per rank, in elements of the tensor handed over),
                          "dtype", "rows" (records: the round's padded rows per rank)}, ...]}

    Order (``_close_round`` / ``_send_round`` / ``finish``): the counts of round c leave when the round closes, its records one round LATER (when the
    counts have long arrived) - C0 C1 R0 C2 R1 ... C(n-1) R(n-2) R(n-1); ``eager``: C0 R0 C1 R1 ...; a round without a single survivor on any
    rank sends no records (every rank sees that in the gathered counts).  tests/test_distributed_cpu.py records a gloo run's calls and compares."""
    if form not in ("all_gather", "gather_to_root", "counts_only"):
        raise ValueError("form must be 'all_gather', 'gather_to_root' or 'counts_only'")
    n_refs, world, B = int(n_refs), int(world), max(1, int(refs_per_round))
    cols, dtype = (7, "float32") if record == RECORD_F32 else (15, "uint8")
    n_local_max = (n_refs + world - 1) // world
    n_rounds = (n_local_max + B - 1) // B
    ranks = []
    for r in range(world):
        pos = shard_references(n_refs, r, world)
        ranks.append({"rank": r, "positions": pos, "rounds": [pos[c * B:(c + 1) * B] for c in range(n_rounds)]})

    def rows_of(c):
        if counts is None:
            return None
        per_rank = [sum(int(counts[g]) for g in ranks[r]["rounds"][c]) for r in range(world)]
        return max(per_rank) if per_rank else 0

    def count_step(c):
        return {"op": "all_gather_into_tensor", "what": "counts", "round": c, "numel_in": B, "numel_out": world * B, "dtype": "int64"}

    def record_step(c):
        rows = rows_of(c)
        if form == "counts_only" or rows == 0:
            return None
        n = None if rows is None else rows * cols
        if form == "all_gather":
            return {"op": "all_gather_into_tensor", "what": "records", "round": c, "rows": rows, "numel_in": n, "numel_out": None if n is None else world * n, "dtype": dtype}
        return {"op": "gather", "what": "records", "round": c, "rows": rows, "numel_in": n, "numel_out": None if n is None else world * n, "dtype": dtype, "dst": int(root)}

    seq = []
    for c in range(n_rounds):
        seq.append(count_step(c))
        send = c if eager else c - 1
        if send >= 0:
            seq.append(record_step(send))
    if n_rounds and not eager:
        seq.append(record_step(n_rounds - 1))
    return {"rounds": n_rounds, "refs_per_round": B, "form": form, "record": record, "record_bytes": cols * (4 if record == RECORD_F32 else 1),
            "ranks": ranks, "collectives": [s for s in seq if s is not None]}


class RecordingDist:
    """A stand-in for the ``torch.distributed`` module handed to the classes of this file that forwards every call and logs the collectives - op name,
    elements in / out, dtype - so that a run's ACTUAL sequence can be compared with ``exchange_schedule`` (CPU tests, gloo)."""
    _LOGGED = ("all_gather_into_tensor", "gather", "all_reduce", "all_gather", "barrier", "isend", "recv", "batch_isend_irecv")

    def __init__(self, dist):
        self._dist = dist
        self.log: List[dict] = []

    def __getattr__(self, name):
        fn = getattr(self._dist, name)
        if name not in self._LOGGED:
            return fn

        def logged(*a, **kw):
            entry = {"op": name}
            tens = [x for x in a if isinstance(x, torch.Tensor)]
            if name == "all_gather_into_tensor" and len(tens) >= 2:
                entry.update(numel_out=int(tens[0].numel()), numel_in=int(tens[1].numel()), dtype=str(tens[1].dtype).replace("torch.", ""))
            elif name == "gather" and tens:
                entry.update(numel_in=int(tens[0].numel()), dtype=str(tens[0].dtype).replace("torch.", ""), dst=kw.get("dst"))
            elif tens:
                entry.update(numel_in=int(tens[0].numel()), dtype=str(tens[0].dtype).replace("torch.", ""))
            self.log.append(entry)
            return fn(*a, **kw)
        return logged
