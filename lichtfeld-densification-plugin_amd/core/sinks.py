"""Where a run's survivors go: the accumulator behind ``PipelineResult``, the intermediate previews, the streamed output file, the debug
previews, and - in a sharded run - the exchange.

Replaces upstream core/pipeline.py:880-898 (the per-reference ``extend`` of three Python lists, the debug preview and the intermediate
PLY of ``_emit_intermediate``, :508-532) and :909-928 (the final concatenation).  The strategies of core/strategies.py hand every finished
reference to ``RunOutputs.emit`` as an ``Emission`` (its survivors still where the kernels wrote them); what each consumer needs of it -
the 15-byte records for the file and the previews, the rows for the exchange - is made once and shared."""
from __future__ import annotations

import dataclasses
import os
import time
from typing import Callable, List, Optional, Tuple

import numpy as np
import torch

from . import distributed as lfd_dist
from .debug_viz import MatchPreview
from .hostlog import log
from .image_io import to_uint8_rgb
from .packing import PackedReference, cancelled, raise_if_cancelled
from .writers import CumulativePlyBody, StreamedPlyWriter, ensure_dir, ply_records

_DEBUG_PREVIEW_INTERVAL = 3      # upstream core/pipeline.py:34
_PREVIEW_MAX_MATCHES = 10000     # upstream core/pipeline.py:50


class PipelineResult:
    """upstream core/pipeline.py:37-43 (``xyz``, ``rgb``, ``err``, ``elapsed_seconds``, ``pairs_processed``) plus what this implementation
    knows beyond it.  The three arrays are made on FIRST ACCESS: from the device tensors (``device_points``: one copy across PCIe), or - when the run
    streamed its output as file records (dense mode + ``stream_output``) - read back from the written file: positions exact, colours as quantised
    (u8 / 255), no reprojection error (a PLY vertex has none).  A caller that only wants the file, or keeps working on the GPU, never pays for them."""

    def __init__(self, xyz=None, rgb=None, err=None, elapsed_seconds: float = 0.0, pairs_processed: int = 0, pairs_matched: int = 0,
                 points_per_reference: Optional[np.ndarray] = None, device_points=None, streamed_path: Optional[str] = None,
                 clock=None, loader: Optional[Callable[[], Tuple[np.ndarray, np.ndarray, np.ndarray]]] = None):
        self._arrays = (xyz, rgb, err) if loader is None else None
        self._loader = loader
        self.elapsed_seconds = float(elapsed_seconds)
        self.pairs_processed = int(pairs_processed)           # upstream's name; counts REFERENCES that produced points
        self.pairs_matched = int(pairs_matched)               # actual (reference, neighbour) pairs matched
        self.points_per_reference = points_per_reference
        self.device_points = device_points                    # the same points, still on the GPU (None when only records were made)
        self.streamed_path = streamed_path                    # config.stream_output: the PLY already written while the run proceeded
        self._clock = clock

    def _get(self, i: int) -> np.ndarray:
        if self._arrays is None:
            self._arrays = tuple(self._loader())
        return self._arrays[i]

    @property
    def stages(self) -> Optional[dict]:
        """run_dense_pipeline(stage_clock=...): StageClock.report() as it stands NOW (what the caller does with the result afterwards - bringing
        the arrays to the host, writing the file - is charged to the same clock); None without a clock"""
        return self._clock.report() if hasattr(self._clock, "report") else None

    xyz = property(lambda self: self._get(0))      # (N,3) f32
    rgb = property(lambda self: self._get(1))      # (N,3) f32 in [0,1]
    err = property(lambda self: self._get(2))      # (N,)  f32

    @property
    def n_points(self) -> int:
        """points this result holds (a rank of a gather_to_root run that is not the root holds its own shard), without materialising the arrays"""
        if self.device_points is not None:
            return int(self.device_points[0].shape[0])
        if self._arrays is not None:
            return int(self._arrays[0].shape[0])
        return int(np.sum(self.points_per_reference)) if self.points_per_reference is not None else int(self.xyz.shape[0])


def arrays_from_ply(path: str):
    """(xyz, rgb = u8 / 255, err = 0) of a PLY this package's writers made (binary little-endian, 15-byte vertices)."""
    with open(path, "rb") as f:
        blob = f.read()
    body = blob.split(b"end_header\n", 1)[1]
    rec = np.frombuffer(body, dtype=np.dtype([("xyz", "<f4", 3), ("rgb", "u1", 3)]))
    return (np.ascontiguousarray(rec["xyz"]), (rec["rgb"].astype(np.float32) / np.float32(255.0)),
            np.zeros((rec.shape[0],), np.float32))


@dataclasses.dataclass
class ShardPlan:
    """Which positions of the reference list this rank processes (core/distributed.py: round-robin over the sharded prefix, then the
    replicated suffix on every rank that receives the cloud).  Decided from the configuration and the world ALONE, before anything can fail."""
    world: int
    rank: int
    n_refs: int
    my_positions: List[int]
    n_sharded: int
    n_sharded_mine: int
    n_rep: int
    consumes_cloud: bool

    @staticmethod
    def make(config, n_refs: int, world: int, rank: int) -> "ShardPlan":
        consumes = world == 1 or config.exchange == "all_gather" or rank == 0
        n_rep = 0
        if world > 1 and config.exp("exchange_overlap") and not config.stream_output:
            n_rep = int(round(float(config.exp("exchange_replicate")) * n_refs))
        mine, n_sharded = lfd_dist.split_replicated(n_refs, n_rep, rank, world, replicas_here=consumes)
        return ShardPlan(world, rank, int(n_refs), mine, n_sharded, sum(1 for g in mine if g < n_sharded), n_rep, consumes)


@dataclasses.dataclass
class Emission:
    """One finished reference: its survivors where the kernels wrote them (CPU tensors on the host backend)."""
    local_i: int
    packed: PackedReference
    points: Tuple[torch.Tensor, torch.Tensor, torch.Tensor]
    dbg: Optional[dict] = None
    _ply: Optional[torch.Tensor] = None
    _body: Optional[bytes] = None

    @property
    def count(self) -> int:
        return int(self.points[0].shape[0])

    def ply_tensor(self, hot) -> torch.Tensor:
        """the 15-byte PLY records as a uint8 tensor that stays where the points are, packed once"""
        if self._ply is None:
            self._ply = hot.pack_ply_tensor(self.points[0], self.points[1])
        return self._ply

    def ply_bytes(self, hot) -> bytes:
        if self._body is None:
            if self._ply is not None:
                with hot.clock.stage("d2h"):
                    self._body = self._ply.cpu().numpy().tobytes()
            else:
                self._body = hot.pack_ply_bytes(self.points[0], self.points[1])
        return self._body


def build_preview(packed: PackedReference, slot: int, cams, matches: np.ndarray, cert_norm: np.ndarray, pair_index: int,
                  total_pairs: int) -> Optional[MatchPreview]:
    """upstream core/pipeline.py:460-505 (``_build_match_preview``): at most 10000 matches, picked by a generator seeded with the pair"""
    if matches.size == 0:
        return None
    total = int(matches.shape[0])
    nbr = cams[packed.nbr_indices[slot]]
    if matches.shape[0] > _PREVIEW_MAX_MATCHES:
        seed = ((int(packed.ref_uid) & 0xFFFFFFFF) * 73856093) ^ ((int(nbr.uid) & 0xFFFFFFFF) * 19349663)
        pick = np.random.default_rng(seed & 0xFFFFFFFF).choice(matches.shape[0], size=_PREVIEW_MAX_MATCHES, replace=False)
        matches, cert_norm = matches[pick], cert_norm[pick]
    return MatchPreview(ref_id=packed.ref_uid, nbr_id=int(nbr.uid),
                        ref_label=os.path.basename(cams[packed.ref_index].image_path),
                        nbr_label=os.path.basename(nbr.image_path), left_image=packed.image,
                        right_image=packed.nbr_images[slot], matches=matches.astype(np.float32, copy=False),
                        cert_norm=cert_norm.astype(np.float32, copy=False), match_count=total,
                        pair_index=int(pair_index), total_pairs=int(total_pairs))


class ShardLink:
    """Everything a sharded run (world > 1) exchanges.  ONE of three arrangements, chosen from the configuration alone and set up before
    anything can fail - every rank then reaches the matching ``finish`` whatever went wrong on it (it sends empty references / closes empty
    rounds) and nobody is left blocked in a receive or a collective ahead of the status agreement:

      * ``stream_output``: the streamed file (``ShardedPlyStream``: records to rank 0, or ``SharedFilePlyStream``: every rank writes its
        own byte ranges) on a process group of its own, and - AFTER that stream has been drained - one exchange of the result;
      * otherwise, by default, the exchange in rounds beside the compute (``OverlappedExchange``);
      * experimental['exchange_overlap'] off: one exchange after the last reference.

    Never two communicators in flight at once: ranks issue a stream's operations and an exchange's rounds in different orders (a rank with
    fewer references closes its last rounds in finish()), which RCCL does not tolerate across concurrent communicators."""

    def __init__(self, dist, plan: ShardPlan, config, dev):
        self.dist, self.plan, self.config, self.dev = dist, plan, config, dev
        self.stream_group = None
        self.shard_stream: Optional[lfd_dist.ShardedPlyStream] = None
        self.shared_file: Optional[lfd_dist.SharedFilePlyStream] = None
        self.xchg: Optional[lfd_dist.OverlappedExchange] = None
        self.xchg_result = None
        self.rep_parts: List[torch.Tensor] = []        # replicated references' records (the exchange's format), in reference order
        per_round = int(config.exp("exchange_round")) or max(int(config.refs_per_launch), 4)
        if config.stream_output:
            self.stream_group = dist.new_group()
            if config.exp("stream_shared_file"):
                self.shared_file = lfd_dist.SharedFilePlyStream(dist, plan.n_refs, per_round, config.output_path, dev, group=self.stream_group)
            else:
                # rank 0 opens the file inside the run's try: if that fails it still receives (and drops) what the others send
                self.shard_stream = lfd_dist.ShardedPlyStream(dist, plan.n_refs, None, dev, group=self.stream_group)
        elif config.exp("exchange_overlap"):
            self.xchg = lfd_dist.OverlappedExchange(dist, plan.n_sharded, per_round, dev, form=config.exchange,
                                                    record=config.exchange_record_format())

    @property
    def streams_file(self) -> bool:
        return self.shard_stream is not None or self.shared_file is not None

    @property
    def wants_ply(self) -> bool:
        return self.streams_file or (self.xchg is not None and self.xchg.record == lfd_dist.RECORD_PLY)

    def push(self, em: Emission, hot, replicated: bool) -> None:
        if self.shard_stream is not None:
            self.shard_stream.push(em.local_i, em.ply_tensor(hot))     # the records stay where they were packed until they travel to rank 0
        if self.shared_file is not None:
            self.shared_file.push(em.local_i, em.ply_tensor(hot))      # ... or until this rank writes them into its own byte range of the file
        if self.xchg is not None:
            # this reference's records join the round being filled; a complete round leaves in an asynchronous collective while the next batch computes
            rec = em.ply_tensor(hot) if self.xchg.record == lfd_dist.RECORD_PLY else lfd_dist.rows_from_points(*em.points)
            if replicated:
                self.rep_parts.append(rec.to(self.dev))                # a replicated reference: every rank that receives the cloud has it already
            else:
                self.xchg.push(em.local_i, rec)

    def finish(self) -> Optional[BaseException]:
        """Drain what this rank still owes its peers; returns the first failure instead of raising (the caller is in ``finally``)."""
        first: Optional[BaseException] = None
        for what, obj in (("sharded output stream", self.shard_stream), ("shared-file output stream", self.shared_file)):
            if obj is None:
                continue
            try:
                obj.finish()          # rank 0 receives what is left / the last byte ranges and the vertex count (incl. a writer failure kept until the peers were drained)
            except Exception as exc:
                log.error(f"The {what} failed: {exc}")
                first = first or exc
        if self.xchg is not None:
            try:
                self.xchg_result = self.xchg.finish()   # closes the rounds that are left (empty ones on a rank that stopped early) and waits for the collectives
            except Exception as exc:
                log.error(f"The overlapped exchange failed: {exc}")
                first = first or exc
        if self.stream_group is not None:
            try:
                self.dist.destroy_process_group(self.stream_group)     # one communicator per run would otherwise stay behind in a long-lived host process
            except Exception as exc:
                log.warn(f"Releasing the stream's process group failed: {exc}")
            self.stream_group = None
        return first

    def gather(self, out: "RunOutputs"):
        """(xyz, rgb, err) as device tensors in global reference order, the global per-reference counts, and the run's counters summed
        over the ranks (replicated references and their pairs count once, on rank 0)."""
        plan, dist = self.plan, self.dist
        if self.xchg_result is not None:
            recs, counts = self.xchg_result         # the rounds travelled beside the compute; what is left is to name the parts of the ordered records
            ply = self.xchg.record == lfd_dist.RECORD_PLY
            if plan.n_rep and plan.consumes_cloud:
                # sharded part | replicated part: the replicated references are the LAST of the list, so the ordered cloud is a concatenation
                if self.rep_parts:
                    recs = torch.cat([recs.reshape(-1)] + [r.reshape(-1) for r in self.rep_parts])
                if not ply:
                    recs = recs.reshape(-1, 7)
                counts = np.concatenate([counts, np.asarray(out.counts_local[plan.n_sharded_mine:], np.int64)])
            elif plan.n_rep:
                counts = np.concatenate([counts, np.zeros(plan.n_rep, np.int64)])      # (a rank that does not receive the cloud did not compute them)
            if ply:
                gx, gc = lfd_dist.points_from_ply_records(recs)
                ge = torch.zeros((int(gx.shape[0]),), dtype=torch.float32, device=gx.device)
            else:
                gx, gc, ge = recs[:, 0:3].contiguous(), recs[:, 3:6].contiguous(), recs[:, 6].contiguous()
            mine_once = (out.refs_with_points - (out.rep_refs_with_points if plan.rank else 0), out.pair_counter - (out.rep_pairs if plan.rank else 0))
        else:
            # the one exchange step: the survivors travel over RCCL from where they already are (HBM), ordered by reference
            if out.dev_parts:
                lx, lc, le = (torch.cat([p[i] for p in out.dev_parts], 0) for i in range(3))
            else:
                lx, lc, le = (torch.zeros((0, 3), device=self.dev), torch.zeros((0, 3), device=self.dev), torch.zeros((0,), device=self.dev))
            # gather_to_root: only rank 0 consumes the cloud (it writes the file) - every record travels once, straight to its place on rank 0;
            # the other ranks return their own shard
            fn = lfd_dist.gather_to_root_by_reference if self.config.exchange == "gather_to_root" else lfd_dist.all_gather_by_reference
            gx, gc, ge, counts = fn(lx, lc, le, out.counts_local, plan.n_refs, dist)
            mine_once = (out.refs_with_points, out.pair_counter)
        t = torch.tensor(list(mine_once), dtype=torch.int64, device=lfd_dist._collective_device(gx, dist))
        dist.all_reduce(t)
        return (gx, gc, ge), counts, int(t[0].item()), int(t[1].item())


class RunOutputs:
    """The consumers of a run's finished references, in upstream's order per reference: accumulate (core/pipeline.py:880-884) - the sharded
    run's exchange - the preview body and the streamed file - the debug previews (:886-895) - the intermediate PLY (:896-898)."""

    def __init__(self, config, plan: ShardPlan, dist, dev, cams, *, on_sequential_viz=None, debug_state=None, cancel_requested=None,
                 total_pairs_est: int = 0):
        self.config, self.plan, self.dev, self.cams = config, plan, dev, cams
        self.on_sequential_viz, self.debug_state, self.cancel = on_sequential_viz, debug_state, cancel_requested
        self.total_pairs_est = int(total_pairs_est)
        self.dev_parts: List[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = []
        self.counts_local = [0] * len(plan.my_positions)
        self.refs_with_points = 0
        self.pair_counter = 0
        self.rep_refs_with_points = 0
        self.rep_pairs = 0
        self.records_only = False                        # DensePlyStreamer: the survivors exist as file records only
        self.cum_body: Optional[CumulativePlyBody] = None          # bytes of the cloud so far, for the intermediate previews
        self.stream_writer: Optional[StreamedPlyWriter] = None     # config.stream_output: the output file grows while the run proceeds
        self.intermediate_base: Optional[str] = None
        if on_sequential_viz and config.viz_interval > 0:
            ensure_dir(config.output_path)
            self.intermediate_base = os.path.splitext(config.output_path)[0] + "_intermediate"
        self.link: Optional[ShardLink] = ShardLink(dist, plan, config, dev) if plan.world > 1 else None

    def open(self) -> None:
        """What can fail (files): called inside the run's try."""
        if self.intermediate_base:
            self.cum_body = CumulativePlyBody()
        if self.config.stream_output and (self.link is None or self.link.shard_stream is not None) and self.plan.rank == 0:
            self.stream_writer = StreamedPlyWriter(self.config.output_path)
            if self.link is not None:
                self.link.shard_stream.writer = self.stream_writer

    # -- per reference ----------------------------------------------------------------------------------------------------------------
    def note_pairs(self, local_i: int, n_pairs: int) -> int:
        """``n_pairs`` more (reference, neighbour) pairs matched (upstream's ``pair_counter``, core/pipeline.py:125-129); returns the first one's number."""
        first = self.pair_counter + 1
        self.pair_counter += int(n_pairs)
        if local_i >= self.plan.n_sharded_mine:
            self.rep_pairs += int(n_pairs)
        return first

    def count_reference(self, local_i: int, n_points: int) -> None:
        self.counts_local[local_i] = int(n_points)
        self.refs_with_points += 1
        if local_i >= self.plan.n_sharded_mine:
            self.rep_refs_with_points += 1

    def emit(self, em: Emission, hot) -> None:
        self.dev_parts.append(em.points)
        self.count_reference(em.local_i, em.count)
        if self.link is not None:
            self.link.push(em, hot, replicated=em.local_i >= self.plan.n_sharded_mine)
        own_file = self.stream_writer if (self.link is None) else None       # (a sharded stream feeds rank 0's writer itself)
        for sink in (self.cum_body, own_file):
            if sink is not None:
                # this reference's PLY records, packed once (on the device when the points are there): the previews and the streamed
                # output are made of these bytes, nothing is re-concatenated or re-quantised later
                body = em.ply_bytes(hot)
                with hot.clock.stage("write", sync=False):
                    sink.append_packed(body)
        if em.dbg is not None and self.debug_state is not None:
            self._debug_previews(em)
        if self.intermediate_base and self.refs_with_points % self.config.viz_interval == 0:
            with hot.clock.stage("write", sync=False):       # the cumulative preview file + whatever the caller's on_sequential_viz does with it
                self._snapshot()

    def _debug_previews(self, em: Emission) -> None:
        total_val = self.total_pairs_est if self.total_pairs_est > 0 else max(self.pair_counter, 1)
        for slot, (m, cn) in em.dbg["matches"].items():
            raise_if_cancelled(self.cancel)
            pair_idx = em.dbg["pair_index"][slot]
            show = (not self.debug_state.is_auto_step()) or _DEBUG_PREVIEW_INTERVAL <= 0 or pair_idx % _DEBUG_PREVIEW_INTERVAL == 1
            if not show:
                continue
            try:
                pv = build_preview(em.packed, slot, self.cams, m, cn, pair_idx, total_val)
                if pv:
                    self.debug_state.submit_preview(pv)
            except Exception as exc:
                log.warn(f"Debug preview failed: {exc}")

    def _snapshot(self) -> None:
        raise_if_cancelled(self.cancel)
        try:
            path = f"{self.intermediate_base}_{self.refs_with_points}.ply"
            self.cum_body.snapshot(path)
            log.debug(f"Live update: {self.cum_body.count:,} points after {self.refs_with_points} refs")
            self.on_sequential_viz(path)
        except Exception as exc:
            log.warn(f"Failed to emit intermediate PLY: {exc}")

    # -- end of run -------------------------------------------------------------------------------------------------------------------
    def finish(self) -> Optional[BaseException]:
        """In the run's ``finally``: the peers are drained, the streamed file gets its vertex count."""
        first = self.link.finish() if self.link is not None else None
        if self.stream_writer is not None:
            try:
                self.stream_writer.close()
            except Exception as exc:
                log.warn(f"Closing the streamed output failed: {exc}")
        return first

    @property
    def streamed(self) -> bool:
        return self.stream_writer is not None or (self.link is not None and self.link.streams_file)

    def result(self, t0: float, clock) -> PipelineResult:
        """upstream core/pipeline.py:909-928: the concatenated cloud - here ONE copy across PCIe of what stayed on the device all along."""
        counts = np.asarray(self.counts_local, np.int64)
        refs_with_points, pairs = self.refs_with_points, self.pair_counter
        streamed_path = self.config.output_path if self.streamed else None
        loader, arrays, device_points = None, None, None
        if self.link is not None:
            device_points, counts, refs_with_points, pairs = self.link.gather(self)
        elif self.dev_parts:
            device_points = tuple(torch.cat([p[i] for p in self.dev_parts], 0) for i in range(3))
        if int(counts.sum()) == 0:
            raise RuntimeError("No points triangulated. Try adjusting parameters.")
        if self.records_only:
            loader = lambda path=self.config.output_path: arrays_from_ply(path)       # noqa: E731
        elif device_points is not None and device_points[0].is_cuda:
            # the f32 cloud crosses PCIe when - and only if - somebody asks for the arrays: a caller that writes the file (densify.py packs the
            # records on the device) or keeps working on the GPU never pays for it
            def loader(pts=device_points, clk=clock):
                with clk.stage("d2h"):
                    return tuple(t.cpu().numpy() for t in pts)
        elif device_points is not None:
            arrays = tuple(t.numpy() for t in device_points)
        else:
            arrays = (np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32), np.zeros((0,), np.float32))
        xyz, rgb, err = arrays if arrays is not None else (None, None, None)
        return PipelineResult(xyz=xyz, rgb=rgb, err=err, elapsed_seconds=time.time() - t0, pairs_processed=refs_with_points,
                              pairs_matched=pairs, points_per_reference=counts, device_points=device_points, streamed_path=streamed_path,
                              clock=clock, loader=loader)


__all__ = ["PipelineResult", "ShardPlan", "Emission", "ShardLink", "RunOutputs", "arrays_from_ply", "build_preview", "cancelled"]
