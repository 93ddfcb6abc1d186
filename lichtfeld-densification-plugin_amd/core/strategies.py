"""How the matched references of a run are triangulated: one strategy object per run, fed by the driver loop with ``submit(Matched)`` and
closed with ``drain()``.

  * ``SampledLoop``      - upstream's mode (core/pipeline.py:602-780 per reference): coverage sampling, then the selected cells.  Five
                           schedules, picked per reference from what the configuration allows: several references per fused call chained on
                           upstream's one RNG stream (the default where nobody waits for intermediate previews) or on one stream each,
                           upstream's normaliser pipelined over a side stream, launch-ahead, or one synchronous reference.
  * ``DenseBatcher``     - every candidate cell of ``refs_per_launch`` references through one launch of the fused kernel.
  * ``DensePlyStreamer`` - dense mode whose only consumer is the streamed output file: the kernel writes the 15-byte PLY records itself,
                           the records cross PCIe on a side stream into pinned double buffers while the next launch computes, and a
                           writer thread appends them to the file.

Per-reference failures are logged and skipped, never fatal (upstream core/pipeline.py:874-879)."""
from __future__ import annotations

import dataclasses
import queue
import threading
from typing import List, Optional

import numpy as np
import torch

from . import hip_backend as hb
from .hostlog import log
from .hotpath import HotPath
from .packing import PackedReference
from .sinks import Emission, RunOutputs


@dataclasses.dataclass
class Matched:
    """One reference the matcher is through with: what the hot path consumes."""
    local_i: int
    packed: PackedReference
    ref: hb.ReferenceInputs
    axes: Optional[tuple]
    H: int
    W: int
    first_pair: int                  # number of its first (reference, neighbour) pair in the run (debug previews)
    want_debug: bool = False


# refs_per_launch = 0 (automatic): the group is AUTO_REFS_PER_LAUNCH references unless their buffers would be large - what ONE launch / fused call
# may hold on the device (survivor rows, weight maps) and in one pinned landing area on the host (hipHostMalloc'ed, page-locked memory)
AUTO_DEVICE_BYTES = 1 << 30
AUTO_PINNED_BYTES = 256 << 20


def bounded_group(n: int, cells: int, device_bytes_per_cell: int, pinned_bytes_per_cell: int) -> int:
    """The automatic group of a grid of ``cells`` cells: ``n`` references, fewer where n of them would exceed the budgets above (a 1280^2 grid
    in dense mode: 45 MB of survivor rows per reference), never less than one.  An explicit ``refs_per_launch`` is not touched."""
    if device_bytes_per_cell:
        n = min(n, AUTO_DEVICE_BYTES // max(1, cells * device_bytes_per_cell))
    if pinned_bytes_per_cell:
        n = min(n, AUTO_PINNED_BYTES // max(1, cells * pinned_bytes_per_cell))
    return max(1, int(n))


def reference_seed(seed: int, uid: int) -> int:
    return (int(seed) * 1000003 + int(uid) * 7919 + 12345) & 0xFFFFFFFF


def _trimmed(res: hb.TriangulationOutput, lo: int, hi: int):
    """copies: a slice would pin the whole capacity-sized buffer of the launch until the run ends"""
    return res.xyz[lo:hi].clone(), res.rgb[lo:hi].clone(), res.err[lo:hi].clone()


class SampledLoop:
    def __init__(self, hot: HotPath, outputs: RunOutputs, config, per_ref_rng: bool, auto_group: bool = False):
        self.hot, self.out, self.config, self.per_ref_rng = hot, outputs, config, bool(per_ref_rng)
        self.auto_group = bool(auto_group)    # refs_per_launch was 0: the group is bounded by bytes (``bounded_group``)
        self.stream_rng = np.random.RandomState(int(config.seed))     # upstream: np.random.seed(config.seed), global stream
        self.group: List[tuple] = []          # several references per fused call: (Matched, seed)
        self.pend_norm: List[tuple] = []      # default mode: aggregated map on its way to the host: (Matched, handle)
        self.inflight: List[tuple] = []       # launched, not yet read back: (Matched, handle)
        self.chain: List[Matched] = []        # single stream, several references per fused call: the group that is filling up
        self.chain_ready: List[tuple] = []    # ... groups whose weight maps are on their way to the host: (items, batch, normaliser slot or None)
        self.chain_fly: List[tuple] = []      # ... launched groups: (items, handle, (checkpoint place, epoch))
        self._ckpt, self._epoch = 0, 0        # next checkpoint place of the device's stream; recoveries so far (a checkpoint older than one is stale)

    # -- the five schedules ---------------------------------------------------------------------------------------------------------
    def submit(self, m: Matched) -> None:
        hot, cfg = self.hot, self.config
        # (a group's aggregated maps and weights on the device, its weights in one pinned landing area: 8 and 4 bytes per cell and reference)
        n_group = bounded_group(int(cfg.refs_per_launch), m.H * m.W, 8, 4) if self.auto_group else int(cfg.refs_per_launch)
        serial = bool(hot.clock.serialising)      # a stage-attribution run takes the unfused calls: `select` and `kernel` are then separate stages
        need_best = m.want_debug or serial
        dseed = reference_seed(cfg.seed, m.packed.ref_uid) if self.per_ref_rng else None
        if self.per_ref_rng and n_group > 1 and hot.can_launch_ahead(need_best, True, m.H, m.W):
            # every reference has its own stream: refs_per_launch of them share one fused call (lfd_triangulate_sampled_multi)
            self.group.append((m, dseed))
            if len(self.group) >= n_group:
                self._flush_group()
            return
        self._flush_group()
        if not self.per_ref_rng and n_group > 1 and hot.can_chain(need_best, m.H, m.W):
            # upstream's ONE stream, refs_per_launch references per fused call (lfd_triangulate_sampled_chain): they draw one after the other, in
            # this order, everything else of their selections runs side by side
            while self.pend_norm:
                self._promote_one()
            while self.inflight:
                self._finish_one()
            if self.chain and (self.chain[0].H, self.chain[0].W) != (m.H, m.W):
                self._launch_chain()
            self.chain.append(m)
            if len(self.chain) >= n_group:
                self._launch_chain()
            return
        self._drain_chain()
        if hot.can_pipeline_normaliser(need_best, self.per_ref_rng, m.H, m.W):
            # upstream's normaliser (the default) without a host wait in the launch stream: this reference's aggregated map starts its way to
            # the host; the reference before it - whose map has arrived meanwhile - gets its sum and its fused launch; the one before that is
            # collected.  The fused calls are issued in reference order: one MT19937 stream, as upstream.
            try:
                self.pend_norm.append((m, hot.begin_normaliser(m.ref, m.axes)))
            except Exception as ex:
                log.error(f"Triangulation error for ref {m.packed.ref_uid}: {ex}")
            while len(self.pend_norm) > 1:
                self._promote_one()
            while len(self.inflight) > 1:
                self._finish_one()
            return
        while self.pend_norm:         # (a run that leaves the pipelined mode - a debug preview switched on - first issues what is pending, in order)
            self._promote_one()
        if hot.can_launch_ahead(need_best, self.per_ref_rng, m.H, m.W):
            # reference i is launched (asynchronously, counts read back behind an event) BEFORE reference i-1 is collected: the host side
            # of one reference - packing, descriptor upload, Python - runs under the kernels of the other
            try:
                self.inflight.append((m, hot.launch_sampled(m.ref, m.axes, dseed)))
            except Exception as ex:
                log.error(f"Triangulation error for ref {m.packed.ref_uid}: {ex}")
            while len(self.inflight) > 1:
                self._finish_one()
            return
        while self.inflight:
            self._finish_one()
        self._one_synchronously(m, dseed, need_best)

    def drain(self) -> None:
        self._flush_group()
        self._drain_chain()
        while self.pend_norm:
            self._promote_one()
        while self.inflight:
            self._finish_one()

    def close(self) -> None:
        pass

    # -- pieces -----------------------------------------------------------------------------------------------------------------------
    def _one_synchronously(self, m: Matched, dseed, need_best: bool) -> None:
        rng = np.random.RandomState(dseed) if self.per_ref_rng else self.stream_rng
        try:
            res, best = self.hot.sampled(m.ref, m.axes, rng, dseed, need_best=need_best)
        except Exception as ex:
            log.error(f"Triangulation error for ref {m.packed.ref_uid}: {ex}")
            return
        if res is None:
            return
        dbg = None
        if m.want_debug:
            dbg = {"matches": self.hot.debug_matches(m.ref, res.cell, res.slot, m.axes, best),
                   "pair_index": {j: m.first_pair + j for j in range(len(m.ref.cert))}}
        self.out.emit(Emission(m.local_i, m.packed, (res.xyz.clone(), res.rgb.clone(), res.err.clone()), dbg), self.hot)

    def _promote_one(self) -> None:
        """The oldest reference whose aggregated map has reached the host: upstream's normaliser from it, then its fused call."""
        m, handle = self.pend_norm.pop(0)
        try:
            s_up = self.hot.finish_normaliser(handle)
            self.inflight.append((m, self.hot.launch_sampled(m.ref, m.axes, None, s_override=s_up, batch=handle[0])))
        except Exception as ex:
            log.error(f"Triangulation error for ref {m.packed.ref_uid}: {ex}")

    def _finish_one(self) -> None:
        """Collect the oldest launched reference and emit it: references are emitted in launch order.  The survivors stay where they are:
        the trimmed device copy is what the previews, the streamed output, the exchange and the device-side writers consume."""
        m, handle = self.inflight.pop(0)
        try:
            res = self.hot.finish_sampled(handle)
        except Exception as ex:
            log.error(f"Triangulation error for ref {m.packed.ref_uid}: {ex}")
            return
        if res is not None:
            self.out.emit(Emission(m.local_i, m.packed, (res.xyz, res.rgb, res.err)), self.hot)

    def _launch_chain(self) -> None:
        """The group that has filled up.  With upstream's normaliser its weight maps start their way to the host now (one aggregate launch, one
        copy on the side stream) and the group BEFORE it - whose maps have arrived meanwhile - gets its sums and its fused call: copy and sums of
        a group hide under the kernels of the other.  With the device's own sums the group is launched at once."""
        if not self.chain:
            return
        items, self.chain = list(self.chain), []
        try:
            batch = self.hot.prepare_chain([m.ref for m in items], items[0].axes)
            slot = self.hot.begin_chain_normalisers(batch) if self.hot.chain_uses_upstream_normaliser() else None
        except Exception as ex:
            # nothing of the group has drawn from the stream: upstream isolates failures per reference (core/pipeline.py:874-879), so the group is
            # redone one reference at a time, in order, behind the groups before it
            log.warn(f"Grouped triangulation of refs {[m.packed.ref_uid for m in items]} failed ({ex}); retrying one by one")
            self._drain_ready()
            for m in items:
                self._one_synchronously(m, None, False)
            return
        self.chain_ready.append((items, batch, slot))
        while len(self.chain_ready) > (1 if slot is not None else 0):
            self._promote_chain()

    def _promote_chain(self) -> None:
        """The oldest group whose weight maps are on their way: its sums, a checkpoint of the stream as it finds it, its fused call."""
        items, batch, slot = self.chain_ready.pop(0)
        place = None
        try:
            sums = self.hot.finish_chain_normalisers(slot) if slot is not None else None
            self.hot.checkpoint_rng(self._ckpt)
            place, self._ckpt = (self._ckpt, self._epoch), (self._ckpt + 1) % hb.RNG_CHECKPOINTS
            self.chain_fly.append((items, self.hot.launch_sampled_chain(batch, sums), place))
        except Exception as ex:
            # the call may have been enqueued in part: the groups before it are collected first (in order), then the stream goes back to where
            # this group found it and its references are redone one at a time - upstream isolates failures per reference (core/pipeline.py:874-879)
            while self.chain_fly:
                self._finish_chain()
            self._recover(items, place, ex)
        while len(self.chain_fly) > 1:
            self._finish_chain()

    def _drain_ready(self) -> None:
        while self.chain_ready:
            self._promote_chain()
        while self.chain_fly:
            self._finish_chain()

    def _finish_chain(self) -> None:
        items, handle, place = self.chain_fly.pop(0)
        try:
            res = self.hot.finish_sampled(handle, check_selection=False)
        except Exception as ex:
            self._recover(items, place, ex)
            return
        # 1-3 are upstream's own refusals (np.random.choice raises before it draws: that reference's error, the stream untouched, there as here).
        # Anything else is THIS implementation's: a chain whose bounded waits expired commits nothing although its first references have drawn,
        # a reference refused for inexactness draws nothing although upstream would have - the stream of the whole call, and of every call
        # launched behind it, is not upstream's any more.
        void = [int(st) for st in res.sel_status[:len(items)] if int(st) in hb.SELECT_VOIDS_STREAM]
        if void:
            self._recover(items, place, hb.selection_error(void[0]))
            return
        for bi, m in enumerate(items):
            st = int(res.sel_status[bi])
            if st != 0:     # what upstream's sampling stage raises for this reference (it has drawn nothing); the others are not affected
                log.error(f"Triangulation error for ref {m.packed.ref_uid}: {hb.selection_error(st)}")
                continue
            lo, hi = int(res.ref_offsets[bi]), int(res.ref_offsets[bi + 1])
            if hi > lo:
                self.out.emit(Emission(m.local_i, m.packed, _trimmed(res, lo, hi)), self.hot)

    def _recover(self, items: List[Matched], place, why) -> None:
        """A fused call of several references failed as a whole.  Nothing of it has been emitted; whatever was launched behind it drew from a
        stream that is void, so those calls are collected and discarded too.  The stream is rolled back to the checkpoint taken before the call
        (on the device, in stream order) and every reference concerned is redone alone, in order: the cloud and the stream afterwards are what
        the one-reference schedule gives.  If the stream cannot be restored the run fails - a wrong stream would silently change every later
        selection."""
        todo = list(items)
        while self.chain_fly:
            later, handle, _place = self.chain_fly.pop(0)
            try:
                self.hot.finish_sampled(handle, check_selection=False)       # waited for, its buffers back in the pool, its results dropped
            except Exception:
                pass
            todo += later
        log.warn(f"Grouped triangulation of refs {[m.packed.ref_uid for m in items]} failed ({why}); redoing refs "
                 f"{[m.packed.ref_uid for m in todo]} one by one")
        # (a checkpoint taken before an EARLIER recovery is stale - and not needed: that recovery's own rollback, later in stream order than
        # anything this call enqueued, has already put the stream right)
        if place is not None and place[1] == self._epoch:
            try:
                self.hot.rollback_rng(place[0])
            except Exception as ex:
                raise RuntimeError(f"the random stream could not be restored after a failed grouped call ({why}): {ex}") from ex
        self._epoch += 1
        for m in todo:
            self._one_synchronously(m, None, False)

    def _drain_chain(self) -> None:
        self._launch_chain()
        self._drain_ready()

    def _flush_group(self) -> None:
        if not self.group:
            return
        items, self.group = list(self.group), []
        try:
            res = self.hot.finish_sampled(self.hot.launch_sampled_multi([m.ref for m, _ in items], items[0][0].axes, [sd for _, sd in items]))
        except Exception as ex:
            # upstream isolates failures per reference (core/pipeline.py:874-879): redo the group one reference at a time, so that only the
            # reference that cannot be processed is dropped
            log.warn(f"Grouped triangulation of refs {[m.packed.ref_uid for m, _ in items]} failed ({ex}); retrying one by one")
            for m, sd in items:
                try:
                    one = self.hot.finish_sampled(self.hot.launch_sampled(m.ref, m.axes, sd))
                except Exception as ex1:
                    log.error(f"Triangulation error for ref {m.packed.ref_uid}: {ex1}")
                    continue
                if one is not None:
                    self.out.emit(Emission(m.local_i, m.packed, (one.xyz, one.rgb, one.err)), self.hot)
            return
        if res is None:
            return
        for bi, (m, _sd) in enumerate(items):
            lo, hi = int(res.ref_offsets[bi]), int(res.ref_offsets[bi + 1])
            if hi > lo:
                self.out.emit(Emission(m.local_i, m.packed, _trimmed(res, lo, hi)), self.hot)


class DenseBatcher:
    def __init__(self, hot: HotPath, outputs: RunOutputs, config, auto_group: bool = False):
        self.hot, self.out, self.config, self.auto_group = hot, outputs, config, bool(auto_group)
        self.pending: List[Matched] = []

    def submit(self, m: Matched) -> None:
        if self.pending and (self.pending[0].H, self.pending[0].W) != (m.H, m.W):
            self.drain()                  # one launch, one grid
        self.pending.append(m)
        n = int(self.config.refs_per_launch)
        if self.auto_group:               # 33 bytes per cell: xyz, rgb, err + the cell and slot columns of the C-ABI
            n = bounded_group(n, m.H * m.W, 33, 0)
        if len(self.pending) >= n:
            self.drain()

    def drain(self) -> None:
        if not self.pending:
            return
        items, self.pending = self.pending, []
        try:
            res = self.hot.dense([m.ref for m in items], items[0].axes)
        except Exception as ex:
            log.error(f"Triangulation error for refs {[m.packed.ref_uid for m in items]}: {ex}")
            return
        offs = res.ref_offsets          # the only read-back of a launch: R + 1 offsets
        for bi, m in enumerate(items):
            lo, hi = int(offs[bi]), int(offs[bi + 1])
            if hi > lo:
                self.out.emit(Emission(m.local_i, m.packed, _trimmed(res, lo, hi)), self.hot)

    def close(self) -> None:
        pass


class _Slot:
    """One of the streamer's buffer pairs: the device records a launch writes, their pinned landing area, and the events between them."""

    def __init__(self, n_refs: int, cells: int, dev, tiles_per_ref: int = 0):
        self.n_refs, self.points = int(n_refs), int(n_refs) * int(cells)
        cap = self.points
        # unordered retirement: where each tile went (not read by this side: a reference's records are contiguous anyway)
        self.table = torch.zeros((max(1, int(n_refs) * int(tiles_per_ref)), 2), dtype=torch.int32, device=dev) if tiles_per_ref else None
        self.records = torch.empty((max(cap * 15, 4),), dtype=torch.uint8, device=dev)
        self.offsets = torch.zeros((n_refs + 1,), dtype=torch.int64, device=dev)
        self.h_records = torch.empty((max(cap * 15, 4),), dtype=torch.uint8).pin_memory()
        self.h_offsets = torch.zeros((n_refs + 1,), dtype=torch.int64).pin_memory()
        self.kernel_done, self.offsets_here = torch.cuda.Event(), torch.cuda.Event()
        self.copy_start, self.copied = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.free = threading.Event()
        self.free.set()


class DensePlyStreamer:
    """Dense mode, ``stream_output``, one process on the device, nobody else looking at the points (no intermediate previews, no debug
    previews): ``lfd_triangulate_dense_ply`` writes the file's 15-byte vertex records; per launch the R + 1 offsets come back first (a few
    bytes, behind an event), then exactly the records that exist cross PCIe on a SIDE stream into a pinned buffer - while the launch stream
    already runs the next batch - and a writer thread appends them to the file.  Two buffer pairs: a launch may start as soon as the
    records of the launch before last have been written out.  15 instead of 28 bytes per survivor cross the bus, once, and no f32 cloud is
    ever assembled (``PipelineResult`` reads the file back if somebody asks for arrays).

    experimental['dense_tile_segments']: the kernel form without the look-back (lfd_triangulate_dense_ply_segments).  Every reference's records
    are then its point SET in tile-retirement order - the same points, not upstream's byte sequence, and not the same sequence from run to run -
    and cross in one copy per reference.

    Replaces upstream core/pipeline.py:753-780,880-884,917-919 + core/writers.py:29-46 for that consumer."""

    def __init__(self, hot: HotPath, outputs: RunOutputs, config, auto_group: bool = False):
        self.hot, self.out, self.config, self.auto_group = hot, outputs, config, bool(auto_group)
        self.unordered = bool(config.exp("dense_tile_segments"))
        self.pending: List[Matched] = []
        self.launched: List[tuple] = []
        self.slots: List[_Slot] = []
        self.side = torch.cuda.Stream(device=hot.dev)
        self.jobs: "queue.Queue" = queue.Queue()
        self.error: Optional[BaseException] = None
        self.thread = threading.Thread(target=self._write_loop, name="lfd-ply-writer", daemon=True)
        self.thread.start()
        outputs.records_only = True

    @staticmethod
    def applies(config, plan, on_host: bool, outputs: RunOutputs, debug_enabled: bool) -> bool:
        return (config.triangulation_mode == "dense" and bool(config.stream_output) and plan.world == 1 and not on_host
                and outputs.intermediate_base is None and not debug_enabled)

    def _group(self, cells: int) -> int:
        n = int(self.config.refs_per_launch)         # 15-byte records: once on the device, once in the pinned landing area of a buffer pair
        return bounded_group(n, cells, 15, 15) if self.auto_group else n

    def submit(self, m: Matched) -> None:
        if self.pending and (self.pending[0].H, self.pending[0].W) != (m.H, m.W):
            self._launch()                # one launch, one grid
        self.pending.append(m)
        if len(self.pending) >= self._group(m.H * m.W):
            self._launch()

    def _slot(self, n_refs: int, cells: int) -> _Slot:
        """A free buffer pair with room for the launch (the run's first launch sizes both: a last, smaller batch fits the same pairs)."""
        fits = [s for s in self.slots if s.n_refs >= n_refs and s.points >= n_refs * cells]
        for s in fits:
            if s.free.is_set():
                return s
        if len(fits) < 2:
            tpr = self.hot.dens.tiles_per_ref(cells, 1) if self.unordered else 0
            self.slots.append(_Slot(max(n_refs, self._group(cells)), cells, self.hot.dev, tpr))
            return self.slots[-1]
        # Both pairs are busy.  One of them may belong to a launch that is still in flight: its records reach the writer - and the pair comes back -
        # only after that launch has been COLLECTED, so everything in flight is collected first (waiting for such a pair without doing so would
        # wait forever); from then on every busy pair is in the writer's queue and the file is what the run waits for.
        while self.launched:
            self._collect()
        with self.hot.clock.stage("write", sync=False):
            while True:
                for s in fits:
                    if s.free.wait(timeout=0.002):
                        return s
                if not self.thread.is_alive():
                    raise RuntimeError("the PLY writer thread has stopped")

    def _launch(self) -> None:
        if not self.pending:
            return
        items, self.pending = self.pending, []
        while len(self.launched) > 1:
            self._collect()
        slot = self._slot(len(items), items[0].H * items[0].W)
        slot.free.clear()
        try:
            if self.unordered:       # (slot.offsets receives the per-reference COUNTS: reference r's records start at byte 15 * r * H * W)
                batch = self.hot.launch_dense_ply([m.ref for m in items], items[0].axes, slot.records, slot.offsets, table=slot.table)
            else:
                batch = self.hot.launch_dense_ply([m.ref for m in items], items[0].axes, slot.records, slot.offsets)
        except Exception as ex:
            slot.free.set()
            log.error(f"Triangulation error for refs {[m.packed.ref_uid for m in items]}: {ex}")
            return
        slot.kernel_done.record(self.hot.dens.stream)
        with torch.cuda.stream(self.side):
            self.side.wait_event(slot.kernel_done)
            slot.h_offsets.copy_(slot.offsets, non_blocking=True)
            slot.offsets_here.record(self.side)
        self.launched.append((slot, items, batch))
        while len(self.launched) > 1:         # one launch in flight beside the one just issued
            self._collect()

    def _collect(self) -> None:
        slot, items, _batch = self.launched.pop(0)
        with self.hot.clock.stage("kernel", sync=False):     # the host waits here for the launch's offsets, i.e. for its kernel
            slot.offsets_here.synchronize()
        offs = slot.h_offsets.numpy()[:len(items) + 1].copy()       # (a pair sized for refs_per_launch serves a last, smaller batch too)
        if self.unordered:
            counts = offs[:len(items)]
            offs = np.concatenate([[0], np.cumsum(counts)])
        n = int(offs[-1])
        with torch.cuda.stream(self.side):
            slot.copy_start.record(self.side)
            if n and not self.unordered:
                slot.h_records[:n * 15].copy_(slot.records[:n * 15], non_blocking=True)
            elif n:                      # one copy per reference, out of its own region of the launch's buffer
                cells = items[0].H * items[0].W
                for bi in range(len(items)):
                    lo, hi = int(offs[bi]), int(offs[bi + 1])
                    if hi > lo:
                        slot.h_records[lo * 15:hi * 15].copy_(slot.records[bi * cells * 15:(bi * cells + hi - lo) * 15], non_blocking=True)
            slot.copied.record(self.side)
        for bi, m in enumerate(items):
            if int(offs[bi + 1]) > int(offs[bi]):
                self.out.count_reference(m.local_i, int(offs[bi + 1]) - int(offs[bi]))
        self.jobs.put((slot, n))

    def _write_loop(self) -> None:
        clock = self.hot.clock
        while True:
            job = self.jobs.get()
            if job is None:
                return
            slot, n = job
            try:
                slot.copied.synchronize()
                clock.add("d2h", slot.copy_start.elapsed_time(slot.copied) * 1e-3)
                clock.count("d2h_bytes", n * 15)
                if n and self.error is None and self.out.stream_writer is not None:
                    with clock.stage("write", sync=False):
                        self.out.stream_writer.append_packed(memoryview(slot.h_records.numpy())[:n * 15])
            except BaseException as exc:          # noqa: BLE001 - raised by drain()
                self.error = self.error or exc
            finally:
                slot.free.set()

    def drain(self) -> None:
        try:
            self._launch()
            while self.launched:
                self._collect()
        finally:
            self.close()
        self.hot.dens.check_launches()
        if self.error is not None:
            raise self.error

    def close(self) -> None:
        """Stop the writer thread (after what is queued has been written); idempotent - the driver calls it in its ``finally``."""
        if self.thread.is_alive():
            self.jobs.put(None)
            self.thread.join()
