"""Per-run state of the per-reference hot path: the context of the C-ABI, the camera table, the prepared-image cache, and the ways of
triangulating a reference (the fused sampled call, its pipelined forms, the unfused three calls, the dense launch).

Replaces upstream core/pipeline.py:405-442 (``_collect_reference_matches`` epilogue) + :602-780 (``_triangulate_ref``): what those do on
the CPU per reference happens here through ``core/hip_backend.py`` (device) or the CPU twin (``backend="host"``)."""
from __future__ import annotations

import collections
import dataclasses
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import hip_backend as hb
from .image_io import to_uint8_rgb
from .packing import PackedReference
from .sampling import select_samples_with_coverage, upstream_weight_sum
from .stages import NULL_CLOCK
from .types import CameraRecord, DensePipelineConfig
from .writers import ply_records

# device_image_prep: bytes of prepared match-size images / masks kept on the device per run (LFD_PREPARED_CACHE_MB; 0 = keep none)
PREPARED_CACHE_BYTES = int(os.environ.get("LFD_PREPARED_CACHE_MB", "4096")) << 20


def _upload_u8(a, dev) -> torch.Tensor:
    """A u8 array to where the kernels run, without a host copy in between.  The loaders' caches hand out READ-ONLY arrays (a cached image must not be
    written to); torch warns when it wraps one - the wrapper here is only ever read (uploaded, or handed to the CPU twin, which takes const pointers)."""
    import warnings
    arr = np.ascontiguousarray(a, dtype=np.uint8)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", UserWarning)
        t = torch.from_numpy(arr)
    return t if torch.device(dev).type == "cpu" else t.to(dev)


class HotPath:
    """Per-run state of the hot path: context, camera table, and the ways of triangulating a reference."""

    def __init__(self, cams: Sequence[CameraRecord], config: DensePipelineConfig, sample_cap: float, w_match: int,
                 h_match: int, dev: torch.device, densifier: Optional[hb.HipDensifier], clock=NULL_CLOCK):
        self.dev = dev
        self.clock = clock
        self.config = config
        self.sample_cap = float(sample_cap)
        self.w_match, self.h_match = int(w_match), int(h_match)
        self.on_host = dev.type == "cpu"          # config.backend == "host": the CPU twin, chosen by the caller
        if densifier is not None:
            self.dens = densifier
        else:
            self.dens = hb.HostDensifier(int(config.exp("host_threads"))) if self.on_host else hb.HipDensifier(dev)
        if self.dens.device.type != dev.type:
            raise ValueError(f"backend runs on {dev} but the densifier handed in lives on {self.dens.device}")
        self._own = densifier is None
        self._prepared: "collections.OrderedDict" = collections.OrderedDict()      # device_image_prep: camera -> (image, mask) at match size
        self._prepared_bytes = 0
        self._buf_pool: dict = {}              # recycled survivor buffers of the fused sampled calls
        self._norm_side = None                 # upstream's normaliser: side stream, free slots, border masks (begin_normaliser)
        self._norm_free: list = []
        self._norm_group, self._norm_grid = 1, None
        self._norm_masks: dict = {}
        self.dens.upload_cameras(cams)
        self.cams = list(cams) if bool(config.exp("upstream_fundamental")) else None
        self.params = hb.make_params(config, sample_cap)

    def close(self) -> None:
        self._prepared.clear()
        self._prepared_bytes = 0
        if self._own:
            self.dens.close()

    def stage_decoded(self, cam_index: int, size_wh: Tuple[int, int], img, mask_l):
        """Called on a PACK thread with a freshly decoded view (device_image_prep): unless the camera's prepared tensors are already on the device,
        the decoded bytes start their way there now - a 3-4 MB pageable copy per image that would otherwise sit on the driver's thread between
        two matcher calls.  (Two packages that need the same new camera at the same moment both upload it: harmless.)"""
        if (int(cam_index), int(size_wh[0]), int(size_wh[1]), mask_l is not None) in self._prepared:
            return img, mask_l
        with self.clock.stage("prepare", sync=False):
            return _upload_u8(img, self.dev), (_upload_u8(mask_l, self.dev) if mask_l is not None else None)

    def prepare_on_device(self, packed: PackedReference, size_wh: Tuple[int, int], need_host: bool) -> PackedReference:
        """device_image_prep: the decoded arrays of ``packed`` are uploaded and resized / thresholded / blacked out by
        lfd_prepare_mask + lfd_prepare_image (Pillow's arithmetic, bit for bit); the result replaces the host-prepared arrays
        (``need_host``: also as NumPy copies, for a matcher that wants PIL images or for debug previews)."""
        dev = self.dev

        def up(a):
            return a if isinstance(a, torch.Tensor) else _upload_u8(a, dev)          # (a pack thread may have uploaded it already: stage_decoded)

        def one(cam_index, img, mask_l):
            # A camera is prepared the same way whether it is the reference or a neighbour, and it appears in ~k + 1 packages of a
            # run: the prepared match-size tensors (0.8 MB + 0.26 MB at 512^2) stay on the device, least recently used first out
            # beyond PREPARED_CACHE_BYTES - upstream keeps its resized images the same way (core/image_utils.py lru caches) - so a
            # decoded full-resolution image (36-72 MB at 12-24 MP) is uploaded and resized once per run, not once per appearance.
            key = (int(cam_index), int(size_wh[0]), int(size_wh[1]), mask_l is not None)
            hit = self._prepared.get(key)
            if hit is not None:
                self._prepared.move_to_end(key)
                return hit
            m01 = self.dens.prepare_mask(up(mask_l), size_wh) if mask_l is not None else None
            entry = (self.dens.prepare_image(up(img), size_wh, m01), m01)
            nbytes = entry[0].numel() + (m01.numel() if m01 is not None else 0)
            if nbytes <= PREPARED_CACHE_BYTES:
                self._prepared[key] = entry
                self._prepared_bytes += nbytes
                while self._prepared_bytes > PREPARED_CACHE_BYTES:
                    _k, old = self._prepared.popitem(last=False)
                    self._prepared_bytes -= old[0].numel() + (old[1].numel() if old[1] is not None else 0)
            return entry

        with self.clock.stage("prepare"):
            img_a, mask_a = one(packed.ref_index, packed.image, packed.mask_a)
            nbrs = [one(ci, im, mk) for ci, im, mk in zip(packed.nbr_indices, packed.nbr_images, packed.nbr_masks)]
        out = dataclasses.replace(packed, raw=False, dev={"image": img_a, "mask_a": mask_a, "nbr_images": [n[0] for n in nbrs],
                                                          "nbr_masks": [n[1] for n in nbrs]})
        if need_host:
            out.image = img_a.cpu().numpy()
            out.mask_a = mask_a.cpu().numpy() if mask_a is not None else None
            out.nbr_images = [n[0].cpu().numpy() for n in nbrs]
            out.nbr_masks = [n[1].cpu().numpy() if n[1] is not None else None for n in nbrs]
        else:
            out.mask_a = True if mask_a is not None else None           # only "is there a mask" is asked of these below
            out.nbr_masks = [True if n[1] is not None else None for n in nbrs]
        return out

    def inputs(self, packed: PackedReference, warps, certs) -> hb.ReferenceInputs:
        dev = self.dev
        if packed.dev is not None:                 # prepared on the device: nothing to upload
            d = packed.dev
            use_masks = d["mask_a"] is not None or any(m is not None for m in d["nbr_masks"])
            return hb.ReferenceInputs(ref_cam=packed.ref_index, nbr_cams=list(packed.nbr_indices), cert=certs, warp=warps,
                                      image=d["image"], mask_a=d["mask_a"], mask_b=list(d["nbr_masks"]) if use_masks else None)
        use_masks = packed.mask_a is not None or any(m is not None for m in packed.nbr_masks)
        mask_b = None
        with self.clock.stage("prepare"):        # the host-prepared image and masks cross to where the kernels run
            if use_masks:
                mask_b = [_upload_u8(m, dev) if m is not None else None for m in packed.nbr_masks]
            image = _upload_u8(packed.image, dev)
            mask_a = _upload_u8(packed.mask_a, dev) if packed.mask_a is not None else None
        return hb.ReferenceInputs(ref_cam=packed.ref_index, nbr_cams=list(packed.nbr_indices), cert=certs, warp=warps, image=image,
                                  mask_a=mask_a, mask_b=mask_b)

    def sampled(self, ref: hb.ReferenceInputs, axes, rng, device_seed: Optional[int], need_best: bool = False
                ) -> Tuple[Optional[hb.TriangulationOutput], Optional[torch.Tensor]]:
        """aggregate kernel -> coverage sampling -> indexed kernel.  The sampling stage runs on the
        device (lfd_select_samples consuming the context's MT19937 stream; lfd_select_top_m for
        no_filter) unless the configuration asks for the host stage (core/sampling.py).  With the device
        stage and no debug preview to feed (``need_best``), the three steps are ONE asynchronous call
        (lfd_triangulate_sampled): the selection count never visits the host."""
        batch = hb.PreparedBatch([ref], self.w_match, self.h_match, axes=axes, cameras=self.cams)
        on_device = self.config.selection_backend == "device" and not self.on_host
        fusable = on_device and (not self.config.no_filter or self.config.matches_per_ref <= self.dens.TOP_M_MAX)
        # upstream's own normaliser (torch's f32 sum of the weights, on this host) for the single-stream, filtered selection
        torch_sum = on_device and not self.config.no_filter and device_seed is None and bool(self.config.upstream_normaliser)
        if fusable and not need_best and not torch_sum:
            if device_seed is not None and not self.config.no_filter:
                self.dens.seed_rng(device_seed)
            try:
                with self.clock.stage("kernel"):
                    out = self.dens.triangulate_sampled(batch, self.params, self.config.matches_per_ref, cap=self.sample_cap, border=2, tiles=24)
                return (out if out.count else None), None
            except hb.SelectionInexact:
                pass         # weights below 2^-29 (certainty_thresh ~ 0): the host stage below, on the device's RNG stream
        with self.clock.stage("select"):
            best, _ = self.dens.aggregate(batch, self.params)
            sel_t = self._select(best, rng, device_seed, on_device, torch_sum)
        if sel_t.numel() == 0:
            return None, best[0]
        with self.clock.stage("kernel"):
            out = self.dens.triangulate_indexed(batch, self.params, sel_t, [0, int(sel_t.numel())])
        return (out if out.count else None), best[0]

    def _select(self, best: torch.Tensor, rng, device_seed: Optional[int], on_device: bool, torch_sum: bool) -> torch.Tensor:
        """The cells of one reference's aggregated map the sampling stage picks (int64, where the kernels run)."""
        sel_t = None
        if on_device and self.config.no_filter and self.config.matches_per_ref <= self.dens.TOP_M_MAX:
            sel_t = self.dens.select_top_m(best[0], self.config.matches_per_ref, cap=self.sample_cap)
        elif on_device and not self.config.no_filter:
            if device_seed is not None:
                self.dens.seed_rng(device_seed)
            try:
                s_up = upstream_weight_sum(best[0], cap=self.sample_cap, border=2) if torch_sum else 0.0
                # (a sum <= 0 is upstream's "nothing to sample" case, which the device stage reports itself from its exact sum)
                sel_t = self.dens.select_samples(best[0], self.config.matches_per_ref, cap=self.sample_cap, border=2, tiles=24,
                                                 s_override=s_up if s_up > 0.0 else 0.0)
            except hb.SelectionInexact:
                # upstream handles such maps normally (core/sampling.py:27-32): run its host stage on the stream the device
                # holds (the refused call consumed nothing) and hand the advanced stream back
                key, pos = self.dens.rng_state()
                rs = np.random.RandomState()
                rs.set_state(("MT19937", key, pos, 0, 0.0))
                sel = select_samples_with_coverage(best[0], self.config.matches_per_ref, cap=self.sample_cap, border=2,
                                                   tiles=24, no_filter=False, rng=rs)
                st = rs.get_state()
                self.dens.set_rng_state(st[1], int(st[2]))
                sel_t = torch.from_numpy(np.ascontiguousarray(sel, dtype=np.int64)).to(self.dev)
        if sel_t is None:        # host stage by configuration (selection_backend="host", or no_filter beyond the device's top-M limit)
            sel = select_samples_with_coverage(best[0], self.config.matches_per_ref, cap=self.sample_cap, border=2,
                                               tiles=24, no_filter=self.config.no_filter, rng=rng)
            sel_t = torch.from_numpy(np.ascontiguousarray(sel, dtype=np.int64)).to(self.dev)
        return sel_t

    def can_launch_ahead(self, need_best: bool, per_ref_rng: bool, H: int, W: int) -> bool:
        """The fused sampled call of reference i+1 may be launched before reference i is read back when nothing on the host
        depends on i's result: the selection runs on the device, no debug preview wants the aggregated map, and the device
        selection cannot refuse its input (every weight >= 2^-29 or 0, i.e. certainty_thresh >= 2^-29 * H*W*cap; a refused
        call consumes no random numbers and falls back to the host stage, which would then see the stream AFTER i+1's draws) -
        or every reference has its own stream anyway."""
        cfg = self.config
        if self.on_host:
            return False
        if cfg.selection_backend == "device" and not cfg.no_filter and not per_ref_rng and bool(cfg.upstream_normaliser):
            return False         # the normaliser comes from the host: one reference at a time
        fusable = cfg.selection_backend == "device" and (not cfg.no_filter or cfg.matches_per_ref <= self.dens.TOP_M_MAX)
        exact_ok = cfg.no_filter or per_ref_rng or float(cfg.certainty_thresh) >= 2.0 ** -29 * H * W * max(self.sample_cap, 1e-6)
        return fusable and not need_best and exact_ok

    def launch_sampled(self, ref: hb.ReferenceInputs, axes, device_seed: Optional[int], s_override: float = 0.0, batch=None):
        """Enqueue one reference's fused call and the read-back of its counts; returns what ``finish_sampled`` needs."""
        if batch is None:
            batch = hb.PreparedBatch([ref], self.w_match, self.h_match, axes=axes, cameras=self.cams)
        if device_seed is not None and not self.config.no_filter:
            self.dens.seed_rng(device_seed)
        M = self.config.matches_per_ref
        with self.clock.stage("kernel"):
            out = self._take_buffers(int(M) + 24 * 24 + 64, 1, batch.k)
            self.dens.launch_sampled(batch, self.params, M, out, cap=self.sample_cap, border=2, tiles=24, s_override=float(s_override))
            out.begin_collect(self.dens.stream)
        return batch, out

    def _take_buffers(self, capacity: int, n_refs: int, k: int) -> hb.OutputBuffers:
        """Survivor buffers of the fused sampled calls, recycled: a fresh OutputBuffers costs two device allocations and - on its first
        read-back - a pinned host allocation (hipHostMalloc: milliseconds), per reference; ``finish_sampled`` hands a buffer back once the
        reference's survivors have been copied out of it."""
        free = self._buf_pool.setdefault((int(capacity), int(n_refs), int(k)), [])
        return free.pop() if free else hb.OutputBuffers(int(capacity), int(n_refs), int(k), self.dev)

    # -- upstream's normaliser without stalling the launch stream ---------------------------------------------------------------
    def can_pipeline_normaliser(self, need_best: bool, per_ref_rng: bool, H: int, W: int) -> bool:
        """The default single-stream sampled mode (``upstream_normaliser``): the aggregated map of reference i is copied to the host on
        a SIDE stream while the host is busy with reference i - 1 (its torch sum, its fused launch) and the matcher with reference
        i + 1; the launch stream never waits for the host.  Same preconditions as the launch-ahead (the device selection must not be
        able to refuse its input), plus: one RNG stream, filter mode."""
        cfg = self.config
        if self.on_host or cfg.selection_backend != "device" or cfg.no_filter or per_ref_rng or need_best:
            return False
        if not bool(cfg.upstream_normaliser):
            return False
        return float(cfg.certainty_thresh) >= 2.0 ** -29 * H * W * max(self.sample_cap, 1e-6)

    def _normaliser_slot(self, R: int, H: int, W: int) -> dict:
        """Buffers of a group's weight maps on their way to the host: aggregated map, weights, pinned landing area, two events.  A slot is
        allocated for the run's group size and SLICED for whatever is smaller (a last short group, a single reference between groups): a run
        allocates two of them - hipHostMalloc of tens of MB costs milliseconds - and a slot of another grid is dropped, not kept."""
        for i, cand in enumerate(self._norm_free):
            if cand["grid"] == (H, W) and cand["cap"] >= R:
                slot = self._norm_free.pop(i)
                break
        else:
            self._norm_free.clear()            # a stale grid's - or too small - slots go back to the driver
            cap = self._norm_group = max(int(R), self._norm_group if self._norm_grid == (H, W) else 1)       # the largest group this grid has seen
            self._norm_grid = (H, W)
            slot = {"grid": (H, W), "cap": cap,
                    "best": torch.empty((cap, H, W), dtype=torch.float32, device=self.dev), "w": torch.empty((cap, H, W), dtype=torch.float32, device=self.dev),
                    "host": torch.empty((cap, H, W), dtype=torch.float32).pin_memory(), "agg_done": torch.cuda.Event(), "copied": torch.cuda.Event()}
        slot["n"] = int(R)
        return slot

    def _border_mask(self, H: int, W: int) -> torch.Tensor:
        mask = self._norm_masks.get((H, W))
        if mask is None:
            ys = torch.arange(H, device=self.dev).view(H, 1)
            xs = torch.arange(W, device=self.dev).view(1, W)
            mask = ((xs >= 2) & (xs <= W - 1 - 2) & (ys >= 2) & (ys <= H - 1 - 2)).to(torch.float32)          # border = 2 (core/pipeline.py:642-649 upstream)
            self._norm_masks = {(H, W): mask}
        return mask

    def _start_normalisers(self, batch: hb.PreparedBatch, slot: dict) -> None:
        """Aggregate on the launch stream; the capped, border-masked WEIGHTS (upstream's ``clamp(max=cap) * inside.float()``: exactly rounded
        element by element, so the device gives the host's values) right behind it; then the weight maps to pinned host memory on the side
        stream, an event behind them.  What is left for the host is upstream's one library-dependent step: torch's f32 ``sum``."""
        if self._norm_side is None:
            self._norm_side = torch.cuda.Stream(device=self.dev)
        n = slot["n"]
        best, w, host = slot["best"][:n], slot["w"][:n], slot["host"][:n]
        self.dens.launch_aggregate(batch, self.params, best, None)
        with torch.cuda.stream(self.dens.stream):
            torch.clamp(best, max=self.sample_cap, out=w)
            w.mul_(self._border_mask(batch.H, batch.W))
            slot["agg_done"].record(self.dens.stream)
        with torch.cuda.stream(self._norm_side):
            self._norm_side.wait_event(slot["agg_done"])
            host.copy_(w, non_blocking=True)
            slot["copied"].record(self._norm_side)

    def begin_normaliser(self, ref: hb.ReferenceInputs, axes):
        """One reference's weight map on its way to the host (``_start_normalisers``); returns what ``finish_normaliser`` needs."""
        batch = hb.PreparedBatch([ref], self.w_match, self.h_match, axes=axes, cameras=self.cams)
        slot = self._normaliser_slot(1, batch.H, batch.W)
        with self.clock.stage("select", sync=False):
            self._start_normalisers(batch, slot)
        return batch, slot

    def finish_normaliser(self, handle) -> float:
        """upstream's torch f32 sum (core/sampling.py:27 there) of the weight map that has arrived; the slot goes back to the pool"""
        _batch, slot = handle
        return self.finish_chain_normalisers(slot)[0]

    def launch_sampled_multi(self, refs: List[hb.ReferenceInputs], axes, seeds: List[int]):
        """``refs_per_launch`` references through ONE fused call, each on its own stream (per_reference_rng)."""
        batch = hb.PreparedBatch(refs, self.w_match, self.h_match, axes=axes, cameras=self.cams)
        M = self.config.matches_per_ref
        with self.clock.stage("kernel"):
            out = self._take_buffers(len(refs) * (int(M) + 24 * 24 + 64), len(refs), batch.k)
            self.dens.launch_sampled_multi(batch, self.params, M, out, seeds, cap=self.sample_cap, border=2, tiles=24)
            out.begin_collect(self.dens.stream)
        return batch, out

    # -- several references per fused call on upstream's ONE stream ------------------------------------------------------------------
    def can_chain(self, need_best: bool, H: int, W: int) -> bool:
        """``refs_per_launch`` references of the single-stream sampled mode may share one fused call (lfd_triangulate_sampled_chain: they draw one
        after the other from the context's MT19937 stream, everything else runs side by side) under the launch-ahead's preconditions: device
        selection that cannot refuse its input for inexactness, no debug preview that wants the aggregated map."""
        cfg = self.config
        if self.on_host or cfg.selection_backend != "device" or need_best:
            return False
        if cfg.no_filter:
            return cfg.matches_per_ref <= self.dens.TOP_M_MAX
        return float(cfg.certainty_thresh) >= 2.0 ** -29 * H * W * max(self.sample_cap, 1e-6)

    def chain_uses_upstream_normaliser(self) -> bool:
        return bool(self.config.upstream_normaliser) and not self.config.no_filter

    def prepare_chain(self, refs: List[hb.ReferenceInputs], axes) -> hb.PreparedBatch:
        """(separate from the launch: a reference that cannot be batched fails HERE, before anything has drawn from the stream)"""
        return hb.PreparedBatch(refs, self.w_match, self.h_match, axes=axes, cameras=self.cams)

    def begin_chain_normalisers(self, batch: hb.PreparedBatch):
        """``begin_normaliser`` for a whole group: ONE aggregate launch, the capped, border-masked weights of all its references in one pass, one
        copy to pinned host memory on the side stream.  The caller launches the group BEFORE this one next (its sums have long arrived), so that
        the copy and torch's sums of this group hide under that group's kernels."""
        slot = self._normaliser_slot(batch.n_refs, batch.H, batch.W)
        with self.clock.stage("select", sync=False):
            self._start_normalisers(batch, slot)
        return slot

    def finish_chain_normalisers(self, slot) -> List[float]:
        """upstream's torch f32 sum (core/sampling.py:27 there) of every reference's weight map, each over its own contiguous (H, W) block"""
        try:
            with self.clock.stage("select", sync=False):
                slot["copied"].synchronize()
                sums = [float(slot["host"][r].reshape(-1).sum()) for r in range(slot["n"])]
        finally:
            self._norm_free.append(slot)
        # (a sum <= 0 is upstream's "nothing to sample" case, which the device stage reports itself from its exact sum)
        return [v if v > 0.0 else 0.0 for v in sums]

    def checkpoint_rng(self, place: int) -> None:
        """The device's random stream put aside in stream order, before a fused call of several references (core/strategies.py::SampledLoop)."""
        self.dens.checkpoint_rng(place)

    def rollback_rng(self, place: int) -> None:
        self.dens.rollback_rng(place)

    def launch_sampled_chain(self, batch: hb.PreparedBatch, s_overrides: Optional[List[float]]):
        M = self.config.matches_per_ref
        with self.clock.stage("kernel"):
            out = self._take_buffers(batch.n_refs * (int(M) + 24 * 24 + 64), batch.n_refs, batch.k)
            self.dens.launch_sampled_chain(batch, self.params, M, out, s_overrides=s_overrides, cap=self.sample_cap, border=2, tiles=24)
            out.begin_collect(self.dens.stream)
        return batch, out

    def finish_sampled(self, handle, check_selection: bool = True) -> Optional[hb.TriangulationOutput]:
        """Wait for the reference's counts, copy its survivors out of the (recycled) buffers: the result owns trimmed tensors.
        ``check_selection=False``: the caller reads ``sel_status`` reference by reference (a chained group: a refused reference is that
        reference's error, not the group's)."""
        _batch, out = handle
        try:
            with self.clock.stage("d2h"):          # the counts: whatever the device still had to do for this reference shows here
                res = out.collect(indexed=True, check_selection=check_selection)
            if res.launch_status != 0:
                self.dens.check_launches()
            if not res.count and check_selection:
                return None
            return dataclasses.replace(res, xyz=res.xyz.clone(), rgb=res.rgb.clone(), err=res.err.clone(),
                                       cell=res.cell.clone() if res.cell is not None else None,
                                       slot=res.slot.clone() if res.slot is not None else None, _packed=None)
        finally:
            self._buf_pool.setdefault((out.capacity, out._n_refs, out._k), []).append(out)

    def pack_ply_tensor(self, xyz: torch.Tensor, rgb: torch.Tensor) -> torch.Tensor:
        """The same records as a uint8 tensor that stays where the points are (what a sharded run sends to the writer rank)."""
        if self.on_host:
            return torch.from_numpy(ply_records(xyz.numpy(), to_uint8_rgb(rgb.numpy())).view(np.uint8).reshape(-1).copy())
        return self.dens.pack_ply(xyz, rgb)

    def pack_ply_bytes(self, xyz: torch.Tensor, rgb: torch.Tensor) -> bytes:
        """The survivors' 15-byte PLY records, quantised and packed on the device (only file payload crosses PCIe)."""
        if self.on_host:
            return ply_records(xyz.numpy(), to_uint8_rgb(rgb.numpy())).tobytes()
        with self.clock.stage("d2h"):
            return self.dens.pack_ply(xyz, rgb).cpu().numpy().tobytes()

    def dense(self, refs: List[hb.ReferenceInputs], axes) -> hb.TriangulationOutput:
        batch = hb.PreparedBatch(refs, self.w_match, self.h_match, axes=axes, cameras=self.cams)
        with self.clock.stage("kernel"):
            if bool(self.config.exp("dense_tile_segments")):
                # unordered retirement (no look-back), raster order restored from the tile table: the same result, bit for bit
                return self.dens.order_segments(self.dens.triangulate_dense_segments(batch, self.params))
            return self.dens.triangulate_dense(batch, self.params)

    def launch_dense_ply(self, refs: List[hb.ReferenceInputs], axes, records: torch.Tensor, ref_offsets: torch.Tensor,
                         table: Optional[torch.Tensor] = None) -> hb.PreparedBatch:
        """The dense kernel writing the 15-byte PLY records itself (lfd_triangulate_dense_ply), asynchronously, into the caller's buffers
        (the streamed output of a dense run: core/strategies.py::DensePlyStreamer).  ``table``: the form without the look-back
        (lfd_triangulate_dense_ply_segments) - ``ref_offsets[:n]`` then receives the references' COUNTS and reference r's records start at
        byte 15 * r * H * W.  Returns the batch, which keeps the inputs alive."""
        batch = hb.PreparedBatch(refs, self.w_match, self.h_match, axes=axes, cameras=self.cams)
        with self.clock.stage("kernel"):
            if table is not None:
                self.dens.launch_dense_ply_segments(batch, self.params, records, ref_offsets, table)
            else:
                self.dens.launch_dense_ply(batch, self.params, records, ref_offsets)
        return batch

    def debug_matches(self, ref: hb.ReferenceInputs, out_cell: torch.Tensor, out_slot: torch.Tensor, axes,
                      best_cert: Optional[torch.Tensor]):
        """Per neighbour slot: clipped [xA,yA,xB,yB] in match pixels + certainty/cap of the survivors
        (upstream core/pipeline.py:761-769), gathered on the GPU from the maps the kernel consumed."""
        res = {}
        H, W = ref.cert[0].shape
        wm1, hm1 = float(self.w_match - 1), float(self.h_match - 1)
        cells = out_cell.long()
        for j in range(len(ref.cert)):
            sel = cells[out_slot == j]
            if sel.numel() == 0:
                continue
            wp = ref.warp[j].reshape(H * W, -1)[sel]
            if wp.shape[1] == 4:
                xan, yan, xbn, ybn = wp[:, 0], wp[:, 1], wp[:, 2], wp[:, 3]
            else:
                ax, ay = axes if axes is not None else (torch.from_numpy(hb.identity_axis(W)).to(self.dev),
                                                        torch.from_numpy(hb.identity_axis(H)).to(self.dev))
                xan, yan, xbn, ybn = ax[sel % W], ay[sel // W], wp[:, 0], wp[:, 1]
            m = torch.stack([((xan + 1.0) * 0.5 * wm1).clamp(0.0, wm1), ((yan + 1.0) * 0.5 * hm1).clamp(0.0, hm1),
                             ((xbn + 1.0) * 0.5 * wm1).clamp(0.0, wm1), ((ybn + 1.0) * 0.5 * hm1).clamp(0.0, hm1)], dim=1)
            denom = self.sample_cap if self.sample_cap > 1e-6 else 1.0
            if best_cert is not None:
                # gathered on the device, divided on the host with NumPy like upstream (core/pipeline.py:766-768): the GPU's f32
                # division may differ from IEEE by an ulp, and these few thousand values are a preview, not a hot path
                cn = np.clip(best_cert.reshape(-1)[sel].cpu().numpy() / denom, 0.0, 1.0).astype(np.float32)
            else:
                cn = np.ones(int(sel.numel()), np.float32)
            res[j] = (m.cpu().numpy().astype(np.float32), cn)
        return res

