"""Dense-initialisation pipeline driver with the per-reference hot path on the GPU.

Drop-in for upstream ``core/pipeline.py::run_dense_pipeline`` (:783-928): same signature, same
callbacks, same result type, same exceptions.  What changes is where the work happens:

    upstream                                     here
    --------------------------------------       ----------------------------------------------
    RoMa outputs copied to the host, sync        stay on the GPU, consumed in place
    _collect_reference_matches epilogue (CPU)    folded into the HIP kernels (floor, masks)
    _triangulate_ref (torch-CPU + NumPy)         lfd_aggregate + lfd_triangulate_indexed  ("sampled")
                                                 or lfd_triangulate_dense                 ("dense")
    4 pack threads, completion order             ordered prefetch (results do not depend on timing)

Two triangulation modes (``DensePipelineConfig.triangulation_mode``):
  * "sampled" (default) reproduces upstream: coverage sampling picks ~0.85*M + <=625 cells per
    reference from the aggregated certainty, those cells are triangulated and emitted in upstream's
    per-neighbour group order.  With ``per_reference_rng=False`` and one GPU the legacy NumPy stream
    is consumed exactly as upstream consumes it (one stream, reference after reference).
  * "dense" sends every grid cell through the fused kernel (survivors in raster order per reference).

Multi-GPU: when ``torch.distributed`` is initialised with world_size > 1 the reference list is dealt
round-robin to the ranks and the survivors are all-gathered in reference order at the end
(core/distributed.py); sampling then uses one RNG stream per reference.
"""
from __future__ import annotations

import collections
import dataclasses
import gc
import os
import time
from concurrent.futures import Future, ThreadPoolExecutor
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import distributed as lfd_dist
from . import hip_backend as hb
from .debug_viz import MatchDebugState, MatchPreview
from .hostlog import log
from .image_io import black_out, decode_mask_l, decode_rgb_u8, load_mask01, load_rgb_u8, to_uint8_rgb
from .matcher import RomaMatcher, has_cached_romav2_weights, romav2_cached_weights_paths
from .sampling import select_samples_with_coverage, upstream_weight_sum
from .scheduler import FeatureCache, PairSchedule
from .types import CameraRecord, DensePipelineConfig
from .writers import CumulativePlyBody, StreamedPlyWriter, ensure_dir, ply_records, write_ply

# device_image_prep: bytes of prepared match-size images / masks kept on the device per run (LFD_PREPARED_CACHE_MB; 0 = keep none)
PREPARED_CACHE_BYTES = int(os.environ.get("LFD_PREPARED_CACHE_MB", "4096")) << 20

_DEBUG_PREVIEW_INTERVAL = 3      # upstream core/pipeline.py:34
_PREVIEW_MAX_MATCHES = 10000     # upstream core/pipeline.py:50


@dataclasses.dataclass
class PipelineResult:
    xyz: np.ndarray                 # (N,3) f32
    rgb: np.ndarray                 # (N,3) f32 in [0,1]
    err: np.ndarray                 # (N,)  f32
    elapsed_seconds: float
    pairs_processed: int            # upstream's name; counts REFERENCES that produced points
    # extras (not present upstream)
    pairs_matched: int = 0          # actual (reference, neighbour) pairs matched
    points_per_reference: Optional[np.ndarray] = None
    device_points: Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = None   # same points, still on the GPU
    streamed_path: Optional[str] = None   # config.stream_output: the PLY already written while the run proceeded (complete, header patched)


class PipelineCancelled(RuntimeError):
    """Raised when a running dense pipeline is cancelled."""


@dataclasses.dataclass
class _PackedReference:
    position: int                   # position in refs_local
    ref_index: int                  # index into camera_records
    ref_uid: int
    image: np.ndarray               # (h,w,3) u8, masked pixels blacked out
    mask_a: Optional[np.ndarray]
    nbr_indices: List[int]
    nbr_images: List[np.ndarray]
    nbr_masks: List[Optional[np.ndarray]]
    # device_image_prep: ``image`` / ``nbr_images`` / masks above hold the DECODED arrays (any size, masks as "L") until
    # _HotPath.prepare_on_device has resized them on the GPU; ``dev`` then holds the prepared device tensors
    raw: bool = False
    dev: Optional[dict] = None


def _cancelled(cb: Optional[Callable[[], bool]]) -> bool:
    if cb is None:
        return False
    try:
        return bool(cb())
    except Exception as exc:
        log.warn(f"Cancellation callback failed: {exc}")
        return False


def _raise_if_cancelled(cb) -> None:
    if _cancelled(cb):
        raise PipelineCancelled("Cancelled")


def _estimate_total_pairs(refs_local, nn_table, uids, nns_per_ref) -> int:
    return sum(sum(1 for n in nn_table[r][:nns_per_ref] if uids[n] != uids[r]) for r in refs_local)


def _pack_reference(position: int, ref_index: int, cams: Sequence[CameraRecord], nn_table, nns_per_ref: int,
                    size_wh: Tuple[int, int], cancel, raw: bool = False) -> Optional[_PackedReference]:
    """Load and pre-process one reference and its neighbours (upstream core/pipeline.py:132-227).  ``raw``: decode only; the
    resize / mask / black-out steps then run on the GPU (``_HotPath.prepare_on_device``)."""
    if _cancelled(cancel):
        return None
    if raw:
        return _pack_reference_raw(position, ref_index, cams, nn_table, nns_per_ref, cancel)
    cam = cams[ref_index]
    try:
        img_a = load_rgb_u8(cam.image_path, size_wh)
    except Exception as exc:
        log.warn(f"Failed to load reference {cam.image_path}: {exc}")
        return None
    mask_a = None
    if getattr(cam, "mask_path", None):
        try:
            mask_a = load_mask01(cam.mask_path, size_wh)
            img_a = black_out(img_a, mask_a)
        except Exception as exc:
            log.warn(f"Failed to load/apply mask for reference {cam.uid}: {exc}")
            mask_a = None
    local = nn_table[ref_index][:nns_per_ref]
    if len(local) == 0:
        return None
    nbr_indices, nbr_images, nbr_masks = [], [], []
    for n in local:
        n = int(n)
        if _cancelled(cancel):
            return None
        nb = cams[n]
        if nb.uid == cam.uid:
            continue
        try:
            img_b = load_rgb_u8(nb.image_path, size_wh)
            mask_b = None
            if getattr(nb, "mask_path", None):
                try:
                    mask_b = load_mask01(nb.mask_path, size_wh)
                    img_b = black_out(img_b, mask_b)
                except Exception as exc:
                    log.warn(f"Failed to load/apply mask for neighbor {nb.uid}: {exc}")
                    mask_b = None
            nbr_indices.append(n)
            nbr_images.append(np.asarray(img_b, dtype=np.uint8))
            nbr_masks.append(mask_b)
        except Exception as exc:
            log.warn(f"Failed to load neighbor {nb.uid}: {exc}")
    if not nbr_images:
        return None
    return _PackedReference(position=position, ref_index=ref_index, ref_uid=int(cam.uid),
                            image=np.asarray(img_a, dtype=np.uint8), mask_a=mask_a, nbr_indices=nbr_indices,
                            nbr_images=nbr_images, nbr_masks=nbr_masks)


def _pack_reference_raw(position: int, ref_index: int, cams: Sequence[CameraRecord], nn_table, nns_per_ref: int,
                        cancel) -> Optional[_PackedReference]:
    """The decode half of _pack_reference: same skip / warn rules, images and masks left as decoded."""
    cam = cams[ref_index]
    try:
        img_a = decode_rgb_u8(cam.image_path)
    except Exception as exc:
        log.warn(f"Failed to load reference {cam.image_path}: {exc}")
        return None
    mask_a = None
    if getattr(cam, "mask_path", None):
        try:
            mask_a = decode_mask_l(cam.mask_path)
        except Exception as exc:
            log.warn(f"Failed to load/apply mask for reference {cam.uid}: {exc}")
    local = nn_table[ref_index][:nns_per_ref]
    if len(local) == 0:
        return None
    nbr_indices, nbr_images, nbr_masks = [], [], []
    for n in local:
        n = int(n)
        if _cancelled(cancel):
            return None
        nb = cams[n]
        if nb.uid == cam.uid:
            continue
        try:
            img_b = decode_rgb_u8(nb.image_path)
            mask_b = None
            if getattr(nb, "mask_path", None):
                try:
                    mask_b = decode_mask_l(nb.mask_path)
                except Exception as exc:
                    log.warn(f"Failed to load/apply mask for neighbor {nb.uid}: {exc}")
            nbr_indices.append(n)
            nbr_images.append(img_b)
            nbr_masks.append(mask_b)
        except Exception as exc:
            log.warn(f"Failed to load neighbor {nb.uid}: {exc}")
    if not nbr_images:
        return None
    return _PackedReference(position=position, ref_index=ref_index, ref_uid=int(cam.uid), image=img_a, mask_a=mask_a,
                            nbr_indices=nbr_indices, nbr_images=nbr_images, nbr_masks=nbr_masks, raw=True)


class _OrderedPrefetcher:
    """Bounded look-ahead over an indexable job list; results come back in submission order, so the
    order references are consumed in (and therefore the RNG stream and the output order) does not
    depend on thread timing - unlike upstream's completion-ordered pool
    (core/threaded_dataloader.py:190-222)."""

    def __init__(self, jobs: Sequence[Callable[[], object]], workers: int, window: int):
        self._jobs = list(jobs)
        self._pool = ThreadPoolExecutor(max_workers=max(1, int(workers)), thread_name_prefix="lfd-pack")
        self._window = max(1, int(window))
        self._futures: Dict[int, Future] = {}
        self._next_submit = 0
        self._next_yield = 0
        self._closed = False

    def _fill(self) -> None:
        while self._next_submit < len(self._jobs) and self._next_submit - self._next_yield < self._window:
            self._futures[self._next_submit] = self._pool.submit(self._jobs[self._next_submit])
            self._next_submit += 1

    def __iter__(self):
        return self

    def __next__(self):
        if self._closed or self._next_yield >= len(self._jobs):
            raise StopIteration
        self._fill()
        fut = self._futures.pop(self._next_yield)
        self._next_yield += 1
        res = fut.result()
        self._fill()
        return res

    def close(self) -> None:
        if self._closed:
            return
        self._closed = True
        for f in self._futures.values():
            f.cancel()
        self._pool.shutdown(wait=True, cancel_futures=True)


def _as_device_maps(results, dev) -> Tuple[List[torch.Tensor], List[torch.Tensor]]:
    warps, certs = [], []
    for warp, cert in results:
        warps.append(torch.as_tensor(warp).detach().to(dev, torch.float32).contiguous())
        certs.append(torch.as_tensor(cert).detach().to(dev, torch.float32).contiguous())
    return warps, certs


def _reference_seed(seed: int, uid: int) -> int:
    return (int(seed) * 1000003 + int(uid) * 7919 + 12345) & 0xFFFFFFFF


def _reference_rng(seed: int, uid: int) -> np.random.RandomState:
    return np.random.RandomState(_reference_seed(seed, uid))


def _build_preview(packed: _PackedReference, slot: int, cams, matches: np.ndarray, cert_norm: np.ndarray,
                   pair_index: int, total_pairs: int) -> Optional[MatchPreview]:
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


class _HotPath:
    """Per-run GPU state: context, camera table, and the two ways of triangulating a reference."""

    def __init__(self, cams: Sequence[CameraRecord], config: DensePipelineConfig, sample_cap: float, w_match: int,
                 h_match: int, dev: torch.device, densifier: Optional[hb.HipDensifier]):
        self.dev = dev
        self.config = config
        self.sample_cap = float(sample_cap)
        self.w_match, self.h_match = int(w_match), int(h_match)
        self.on_host = dev.type == "cpu"          # config.backend == "host": the CPU twin, chosen by the caller
        if densifier is not None:
            self.dens = densifier
        else:
            self.dens = hb.HostDensifier(int(getattr(config, "host_threads", 0))) if self.on_host else hb.HipDensifier(dev)
        if self.dens.device.type != dev.type:
            raise ValueError(f"backend runs on {dev} but the densifier handed in lives on {self.dens.device}")
        self._own = densifier is None
        self._prepared: "collections.OrderedDict" = collections.OrderedDict()      # device_image_prep: camera -> (image, mask) at match size
        self._prepared_bytes = 0
        self.dens.upload_cameras(cams)
        self.cams = list(cams) if bool(getattr(config, "upstream_fundamental", True)) else None
        self.params = hb.make_params(config, sample_cap)

    def close(self) -> None:
        self._prepared.clear()
        self._prepared_bytes = 0
        if self._own:
            self.dens.close()

    def prepare_on_device(self, packed: _PackedReference, size_wh: Tuple[int, int], need_host: bool) -> _PackedReference:
        """device_image_prep: the decoded arrays of ``packed`` are uploaded and resized / thresholded / blacked out by
        lfd_prepare_mask + lfd_prepare_image (Pillow's arithmetic, bit for bit); the result replaces the host-prepared arrays
        (``need_host``: also as NumPy copies, for a matcher that wants PIL images or for debug previews)."""
        dev = self.dev

        def up(a):
            return torch.from_numpy(np.array(a, dtype=np.uint8, copy=True)).to(dev)      # the decode cache hands out read-only arrays

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

    def inputs(self, packed: _PackedReference, warps, certs) -> hb.ReferenceInputs:
        dev = self.dev
        if packed.dev is not None:                 # prepared on the device: nothing to upload
            d = packed.dev
            use_masks = d["mask_a"] is not None or any(m is not None for m in d["nbr_masks"])
            return hb.ReferenceInputs(ref_cam=packed.ref_index, nbr_cams=list(packed.nbr_indices), cert=certs, warp=warps,
                                      image=d["image"], mask_a=d["mask_a"], mask_b=list(d["nbr_masks"]) if use_masks else None)
        use_masks = packed.mask_a is not None or any(m is not None for m in packed.nbr_masks)
        mask_b = None
        if use_masks:
            mask_b = [torch.from_numpy(np.array(m, dtype=np.uint8, copy=True)).to(dev) if m is not None else None
                      for m in packed.nbr_masks]
        return hb.ReferenceInputs(
            ref_cam=packed.ref_index, nbr_cams=list(packed.nbr_indices), cert=certs, warp=warps,
            image=torch.from_numpy(np.array(packed.image, dtype=np.uint8, copy=True)).to(dev),
            mask_a=torch.from_numpy(np.array(packed.mask_a, dtype=np.uint8, copy=True)).to(dev) if packed.mask_a is not None else None,
            mask_b=mask_b)

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
        torch_sum = on_device and not self.config.no_filter and device_seed is None and bool(getattr(self.config, "upstream_normaliser", True))
        if fusable and not need_best and not torch_sum:
            if device_seed is not None and not self.config.no_filter:
                self.dens.seed_rng(device_seed)
            try:
                out = self.dens.triangulate_sampled(batch, self.params, self.config.matches_per_ref, cap=self.sample_cap, border=2, tiles=24)
                return (out if out.count else None), None
            except hb.SelectionInexact:
                pass         # weights below 2^-29 (certainty_thresh ~ 0): the host stage below, on the device's RNG stream
        best, _ = self.dens.aggregate(batch, self.params)
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
        if sel_t.numel() == 0:
            return None, best[0]
        out = self.dens.triangulate_indexed(batch, self.params, sel_t, [0, int(sel_t.numel())])
        return (out if out.count else None), best[0]

    def can_launch_ahead(self, need_best: bool, per_ref_rng: bool, H: int, W: int) -> bool:
        """The fused sampled call of reference i+1 may be launched before reference i is read back when nothing on the host
        depends on i's result: the selection runs on the device, no debug preview wants the aggregated map, and the device
        selection cannot refuse its input (every weight >= 2^-29 or 0, i.e. certainty_thresh >= 2^-29 * H*W*cap; a refused
        call consumes no random numbers and falls back to the host stage, which would then see the stream AFTER i+1's draws) -
        or every reference has its own stream anyway."""
        cfg = self.config
        if self.on_host:
            return False
        if cfg.selection_backend == "device" and not cfg.no_filter and not per_ref_rng and bool(getattr(cfg, "upstream_normaliser", True)):
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
        out = self._take_buffers(int(M) + 24 * 24 + 64, 1, batch.k)
        self.dens.launch_sampled(batch, self.params, M, out, cap=self.sample_cap, border=2, tiles=24, s_override=float(s_override))
        out.begin_collect(self.dens.stream)
        return batch, out

    def _take_buffers(self, capacity: int, n_refs: int, k: int) -> hb.OutputBuffers:
        """Survivor buffers of the fused sampled calls, recycled: a fresh OutputBuffers costs two device allocations and - on its first
        read-back - a pinned host allocation (hipHostMalloc: milliseconds), per reference; ``finish_sampled`` hands a buffer back once the
        reference's survivors have been copied out of it."""
        pool = self.__dict__.setdefault("_buf_pool", {})
        free = pool.setdefault((int(capacity), int(n_refs), int(k)), [])
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
        if not bool(getattr(cfg, "upstream_normaliser", True)):
            return False
        return float(cfg.certainty_thresh) >= 2.0 ** -29 * H * W * max(self.sample_cap, 1e-6)

    def begin_normaliser(self, ref: hb.ReferenceInputs, axes):
        """Aggregate on the launch stream; the capped, border-masked WEIGHTS (upstream's ``clamp(max=cap) * inside.float()``: exactly rounded
        element by element, so the device gives the host's values) right behind it; then the 1 MB weight map to pinned host memory on the side
        stream, an event behind it.  What is left for the host is upstream's one library-dependent step: torch's f32 ``sum``."""
        batch = hb.PreparedBatch([ref], self.w_match, self.h_match, axes=axes, cameras=self.cams)
        if getattr(self, "_norm_side", None) is None:
            self._norm_side = torch.cuda.Stream(device=self.dev)
            self._norm_free: list = []
            self._norm_masks: dict = {}
        H, W = batch.H, batch.W
        mask = self._norm_masks.get((H, W))
        if mask is None:
            ys = torch.arange(H, device=self.dev).view(H, 1)
            xs = torch.arange(W, device=self.dev).view(1, W)
            mask = ((xs >= 2) & (xs <= W - 1 - 2) & (ys >= 2) & (ys <= H - 1 - 2)).to(torch.float32)          # border = 2 (core/pipeline.py:642-649 upstream)
            self._norm_masks[(H, W)] = mask
        slot = None
        for i, cand in enumerate(self._norm_free):
            if tuple(cand["best"].shape) == (1, H, W):
                slot = self._norm_free.pop(i)
                break
        if slot is None:
            slot = {"best": torch.empty((1, H, W), dtype=torch.float32, device=self.dev), "w": torch.empty((H, W), dtype=torch.float32, device=self.dev),
                    "host": torch.empty((H, W), dtype=torch.float32).pin_memory(), "agg_done": torch.cuda.Event(), "copied": torch.cuda.Event()}
        self.dens.launch_aggregate(batch, self.params, slot["best"], None)
        with torch.cuda.stream(self.dens.stream):
            torch.clamp(slot["best"][0], max=self.sample_cap, out=slot["w"])
            slot["w"].mul_(mask)
            slot["agg_done"].record(self.dens.stream)
        with torch.cuda.stream(self._norm_side):
            self._norm_side.wait_event(slot["agg_done"])
            slot["host"].copy_(slot["w"], non_blocking=True)
            slot["copied"].record(self._norm_side)
        return batch, slot

    def finish_normaliser(self, handle) -> float:
        """upstream's torch f32 sum (core/sampling.py:27 there) of the weight map that has arrived; the slot goes back to the pool"""
        _batch, slot = handle
        slot["copied"].synchronize()
        s_up = float(slot["host"].reshape(-1).sum())
        self._norm_free.append(slot)
        return s_up if s_up > 0.0 else 0.0      # (a sum <= 0 is upstream's "nothing to sample" case, which the device stage reports from its exact sum)

    def launch_sampled_multi(self, refs: List[hb.ReferenceInputs], axes, seeds: List[int]):
        """``refs_per_launch`` references through ONE fused call, each on its own stream (per_reference_rng)."""
        batch = hb.PreparedBatch(refs, self.w_match, self.h_match, axes=axes, cameras=self.cams)
        M = self.config.matches_per_ref
        out = self._take_buffers(len(refs) * (int(M) + 24 * 24 + 64), len(refs), batch.k)
        self.dens.launch_sampled_multi(batch, self.params, M, out, seeds, cap=self.sample_cap, border=2, tiles=24)
        out.begin_collect(self.dens.stream)
        return batch, out

    def finish_sampled(self, handle) -> Optional[hb.TriangulationOutput]:
        """Wait for the reference's counts, copy its survivors out of the (recycled) buffers: the result owns trimmed tensors."""
        _batch, out = handle
        try:
            res = out.collect(indexed=True, check_selection=True)
            if res.launch_status != 0:
                self.dens.check_launches()
            if not res.count:
                return None
            return dataclasses.replace(res, xyz=res.xyz.clone(), rgb=res.rgb.clone(), err=res.err.clone(),
                                       cell=res.cell.clone() if res.cell is not None else None,
                                       slot=res.slot.clone() if res.slot is not None else None, _packed=None)
        finally:
            self.__dict__.setdefault("_buf_pool", {}).setdefault((out.capacity, out._n_refs, out._k), []).append(out)

    def pack_ply_tensor(self, xyz: torch.Tensor, rgb: torch.Tensor) -> torch.Tensor:
        """The same records as a uint8 tensor that stays where the points are (what a sharded run sends to the writer rank)."""
        if self.on_host:
            return torch.from_numpy(ply_records(xyz.numpy(), to_uint8_rgb(rgb.numpy())).view(np.uint8).reshape(-1).copy())
        return self.dens.pack_ply(xyz, rgb)

    def pack_ply_bytes(self, xyz: torch.Tensor, rgb: torch.Tensor) -> bytes:
        """The survivors' 15-byte PLY records, quantised and packed on the device (only file payload crosses PCIe)."""
        if self.on_host:
            return ply_records(xyz.numpy(), to_uint8_rgb(rgb.numpy())).tobytes()
        return self.dens.pack_ply(xyz, rgb).cpu().numpy().tobytes()

    def dense(self, refs: List[hb.ReferenceInputs], axes) -> hb.TriangulationOutput:
        batch = hb.PreparedBatch(refs, self.w_match, self.h_match, axes=axes, cameras=self.cams)
        if bool(getattr(self.config, "dense_tile_segments", False)) and not self.on_host:
            # unordered retirement (no look-back), raster order restored from the tile table: the same result, bit for bit
            return self.dens.order_segments(self.dens.triangulate_dense_segments(batch, self.params))
        return self.dens.triangulate_dense(batch, self.params)

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


def run_dense_pipeline(
    camera_records: List[CameraRecord],
    refs_local: List[int],
    nn_table: np.ndarray,
    config: DensePipelineConfig,
    progress_callback: Optional[Callable[[float, str], None]] = None,
    on_sequential_viz: Optional[Callable[[str], None]] = None,
    debug_state: Optional[MatchDebugState] = None,
    cancel_requested: Optional[Callable[[], bool]] = None,
    *,
    matcher=None,
    densifier: Optional[hb.HipDensifier] = None,
    device: Optional[torch.device] = None,
    backend: Optional[str] = None,
) -> PipelineResult:
    """See module docstring.  ``matcher`` / ``densifier`` / ``device`` are injection points for
    tests and for callers that keep a warm model; by default a RomaMatcher is created (and released)
    per run exactly like upstream.  ``backend`` ("device" | "host", default ``config.backend``): "host" runs the per-reference
    path on the CPU twin of the C-ABI (HostDensifier + core/sampling.py) - upstream's CPU-only configuration, chosen by the
    caller; a "device" run without a GPU raises HipBackendError, it never turns into a host run."""
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0

    uids = [c.uid for c in camera_records]
    total_pairs_est = _estimate_total_pairs(refs_local, nn_table, uids, config.nns_per_ref)
    if debug_state:
        debug_state.set_total_pairs(total_pairs_est)
    viz_interval = config.viz_interval
    intermediate_base = None
    if on_sequential_viz and viz_interval > 0:
        ensure_dir(config.output_path)
        intermediate_base = os.path.splitext(config.output_path)[0] + "_intermediate"

    backend = str(backend if backend is not None else getattr(config, "backend", "device"))
    if backend not in ("device", "host"):
        raise ValueError("backend must be 'device' or 'host'")
    if backend == "host":
        dev = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise hb.HipBackendError("no GPU visible: the dense-initialisation hot path has no CPU fallback "
                                     "(backend=\"host\" selects the CPU twin explicitly)")
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if dev.type != "cuda":
            raise hb.HipBackendError(f"backend=\"device\" needs a cuda (HIP) device, got {dev}")

    per_ref_rng = bool(config.per_reference_rng) or world > 1
    stream_rng = np.random.RandomState(int(config.seed))     # upstream: np.random.seed(config.seed), global stream
    # Sharded runs may REPLICATE the last references of the list (config.exchange_replicate: computed by every rank that receives the cloud, never
    # sent - core/distributed.py::plan_replication says when that pays: never with a real matcher in the loop); the others are dealt round-robin.
    stream_wanted = (bool(getattr(config, "stream_output", False)) and str(config.output_path).lower().endswith(".ply")
                     and int(config.max_points) <= 0 and float(config.voxel_size) <= 0.0)
    consumes_cloud = world == 1 or str(getattr(config, "exchange", "all_gather")) == "all_gather" or rank == 0
    n_rep = 0
    if world > 1 and bool(getattr(config, "exchange_overlap", True)) and not stream_wanted:
        n_rep = int(round(float(getattr(config, "exchange_replicate", 0.0)) * len(refs_local)))
    my_positions, n_sharded = lfd_dist.split_replicated(len(refs_local), n_rep, rank, world, replicas_here=consumes_cloud)
    n_sharded_mine = sum(1 for g in my_positions if g < n_sharded)
    rep_parts: List[torch.Tensor] = []                     # replicated references' records (the exchange's format), in reference order
    rep_refs_with_points = 0
    rep_pairs = 0

    xyz_parts: List[np.ndarray] = []
    rgb_parts: List[np.ndarray] = []
    err_parts: List[np.ndarray] = []
    dev_parts: List[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = []
    counts_local = [0] * len(my_positions)
    refs_with_points = 0
    pair_counter = 0
    t0 = time.time()
    own_matcher = matcher is None
    cum_body: Optional[CumulativePlyBody] = None          # bytes of the cloud so far, for the intermediate previews
    stream_writer: Optional[StreamedPlyWriter] = None     # config.stream_output: the output file grows while the run proceeds
    prefetch: Optional[_OrderedPrefetcher] = None
    hot: Optional[_HotPath] = None
    feat_cache: Optional[FeatureCache] = None
    shard_stream: Optional[lfd_dist.ShardedPlyStream] = None

    rank_status = 0          # 0 fine, 1 cancelled, 2 failed: agreed on by all ranks before the exchange step (core/distributed.py)
    rank_error: Optional[BaseException] = None
    # What a sharded run exchanges while it proceeds is decided from the configuration and the world ALONE and set up here, before
    # anything can fail: every rank then reaches the matching finish() in `finally` whatever went wrong on it (a matcher that cannot be
    # built, no memory for the context, a cancellation) - it sends empty references / closes empty rounds - and nobody is left blocked
    # in a receive or a collective ahead of the status agreement.
    xchg: Optional[lfd_dist.OverlappedExchange] = None
    xchg_result = None
    shared_file: Optional[lfd_dist.SharedFilePlyStream] = None
    if world > 1:
        # The streamed output runs on a process group OF ITS OWN: its messages (point-to-point sends to rank 0, or its rounds of counts) and
        # the rounds of the overlapped exchange are issued in different orders on different ranks (a rank with fewer references closes its
        # last rounds in finish()), and operations on ONE communicator are matched - under RCCL also executed - in issue order.
        stream_group = dist.new_group() if stream_wanted else None
        if stream_wanted and bool(getattr(config, "stream_shared_file", False)):
            # exchange-free streamed output on one node: only counts travel, every rank writes its own byte ranges of the file
            per_round = int(getattr(config, "exchange_round", 0)) or max(int(config.refs_per_launch), 4)
            shared_file = lfd_dist.SharedFilePlyStream(dist, len(refs_local), per_round, config.output_path, dev, group=stream_group)
        elif stream_wanted:
            # sharded streamed output (BASELINE config 5): every rank packs its finished references' records on the device, rank 0 appends
            # them to the file in global reference order as they arrive.  Rank 0 opens the file inside the try: if that fails it still
            # receives (and drops) what the others send.
            shard_stream = lfd_dist.ShardedPlyStream(dist, len(refs_local), None, dev, group=stream_group)
        if bool(getattr(config, "exchange_overlap", True)):
            rec = str(getattr(config, "exchange_records", "f32"))
            if rec == "auto":
                rec = "ply" if (str(config.output_path).lower().endswith(".ply") and float(config.voxel_size) <= 0.0) else "f32"
            per_round = int(getattr(config, "exchange_round", 0)) or max(int(config.refs_per_launch), 4)
            xchg = lfd_dist.OverlappedExchange(dist, n_sharded, per_round, dev, form=str(getattr(config, "exchange", "all_gather")), record=rec)
    try:
        cached = has_cached_romav2_weights() if own_matcher else True
        msg = "Initializing RoMa v2 model..." if cached else "Installing model weights..."
        if progress_callback is not None:
            progress_callback(10.0, msg)
        log.info(msg)
        if not cached:
            log.info("RoMaV2 weights not found in cache; expected cache paths: " + ", ".join(romav2_cached_weights_paths()))
        if own_matcher:
            matcher = RomaMatcher(device=str(dev), mode="outdoor", setting=config.roma_setting,
                                  pairs_per_forward=int(getattr(config, "pairs_per_forward", 1)))
            if not cached and progress_callback is not None:
                progress_callback(10.0, "RoMa v2 model installation complete. Starting matching...")
        _raise_if_cancelled(cancel_requested)

        w_match, h_match = int(matcher.w_resized), int(matcher.h_resized)
        hot = _HotPath(camera_records, config, float(matcher.sample_thresh), w_match, h_match, dev, densifier)
        if not hot.on_host:
            hot.dens.seed_rng(int(config.seed))      # upstream: np.random.seed(config.seed) (core/pipeline.py:793); the host backend draws from stream_rng
        # N4: every camera's backbone features once per run, kept exactly until their last use (core/scheduler.py)
        schedule = PairSchedule(refs_local, nn_table, uids, config.nns_per_ref, positions=my_positions)
        if bool(getattr(matcher, "supports_feature_keys", False)):
            if bool(getattr(config, "share_features", True)):
                feat_cache = FeatureCache(schedule.last_use)
            # always (re)set: an injected, warm matcher may still hold the cache of an earlier run - other cameras under the same keys
            matcher.set_feature_cache(feat_cache)
        device_prep = bool(getattr(config, "device_image_prep", False)) and backend == "device"
        jobs = [(lambda p=p: _pack_reference(p, refs_local[p], camera_records, nn_table, config.nns_per_ref,
                                             (w_match, h_match), cancel_requested, raw=device_prep)) for p in my_positions]
        prefetch = _OrderedPrefetcher(jobs, workers=int(getattr(config, "pack_workers", 4)),
                                      window=int(getattr(config, "prefetch_packages", 8)))
        total_refs = len(my_positions)
        pending: List[Tuple[int, _PackedReference, hb.ReferenceInputs, object]] = []   # dense mode batching
        inflight: List[Tuple[int, _PackedReference, object]] = []                       # sampled mode: launched, not yet read back
        if on_sequential_viz and viz_interval > 0 and intermediate_base:
            cum_body = CumulativePlyBody()
        if stream_wanted and shared_file is None:
            if rank == 0:
                stream_writer = StreamedPlyWriter(config.output_path)
                if shard_stream is not None:
                    shard_stream.writer = stream_writer

        def emit(local_i: int, packed: _PackedReference, xyz, rgb, err, dbg, dev_pts=None) -> None:
            nonlocal refs_with_points, rep_refs_with_points
            replicated = local_i >= n_sharded_mine
            if replicated:
                rep_refs_with_points += 1
            # (dense mode hands over the device tensors only: the host arrays of the result are one copy at the end of the run)
            xyz_parts.append(xyz)
            rgb_parts.append(rgb)
            err_parts.append(err)
            counts_local[local_i] = int(xyz.shape[0]) if xyz is not None else int(dev_pts[0].shape[0])
            refs_with_points += 1
            packed_t = None
            if shard_stream is not None or shared_file is not None or (xchg is not None and xchg.record == lfd_dist.RECORD_PLY):
                packed_t = (hot.pack_ply_tensor(dev_pts[0], dev_pts[1]) if dev_pts is not None
                            else torch.from_numpy(ply_records(xyz, to_uint8_rgb(rgb)).view(np.uint8).reshape(-1).copy()))
            if shard_stream is not None:
                # sharded streamed output: the records stay where they were packed until they travel to rank 0
                shard_stream.push(local_i, packed_t)
            if shared_file is not None:
                shared_file.push(local_i, packed_t)      # ... or until this rank writes them into its own byte range of the file
            if xchg is not None:
                # the overlapped exchange: this reference's records join the round being filled; a round that is complete leaves in an
                # asynchronous collective while the next batch computes
                if xchg.record == lfd_dist.RECORD_PLY:
                    rec_t = packed_t
                elif dev_pts is not None:
                    rec_t = lfd_dist.rows_from_points(dev_pts[0], dev_pts[1], dev_pts[2])
                else:
                    rec_t = lfd_dist.rows_from_points(torch.from_numpy(xyz), torch.from_numpy(rgb), torch.from_numpy(err)).to(dev)
                if replicated:
                    rep_parts.append(rec_t.to(dev))         # a replicated reference: every rank that receives the cloud has it already
                else:
                    xchg.push(local_i, rec_t)
            if cum_body is not None or (stream_writer is not None and shard_stream is None):
                # this reference's PLY records, packed once (on the device when the points are there): the previews and the
                # streamed output are made of these bytes, nothing is re-concatenated or re-quantised later
                body = hot.pack_ply_bytes(dev_pts[0], dev_pts[1]) if dev_pts is not None else None
                for sink in (cum_body, stream_writer if shard_stream is None else None):
                    if sink is None:
                        continue
                    if body is not None:
                        sink.append_packed(body)
                    else:
                        sink.append(xyz, to_uint8_rgb(rgb))
            if dbg is not None and debug_state is not None:
                total_val = total_pairs_est if total_pairs_est > 0 else max(pair_counter, 1)
                for slot, (m, cn) in dbg["matches"].items():
                    _raise_if_cancelled(cancel_requested)
                    pair_idx = dbg["pair_index"][slot]
                    show = (not debug_state.is_auto_step()) or _DEBUG_PREVIEW_INTERVAL <= 0 or pair_idx % _DEBUG_PREVIEW_INTERVAL == 1
                    if not show:
                        continue
                    try:
                        pv = _build_preview(packed, slot, camera_records, m, cn, pair_idx, total_val)
                        if pv:
                            debug_state.submit_preview(pv)
                    except Exception as exc:
                        log.warn(f"Debug preview failed: {exc}")
            if on_sequential_viz and viz_interval > 0 and intermediate_base and refs_with_points % viz_interval == 0:
                _raise_if_cancelled(cancel_requested)
                try:
                    path = f"{intermediate_base}_{refs_with_points}.ply"
                    cum_body.snapshot(path)
                    log.debug(f"Live update: {cum_body.count:,} points after {refs_with_points} refs")
                    on_sequential_viz(path)
                except Exception as exc:
                    log.warn(f"Failed to emit intermediate PLY: {exc}")

        def flush_dense() -> None:
            if not pending:
                return
            axes = pending[0][3]
            try:
                out = hot.dense([p[2] for p in pending], axes)
            except Exception as ex:
                log.error(f"Triangulation error for refs {[p[1].ref_uid for p in pending]}: {ex}")
                pending.clear()
                return
            offs = out.ref_offsets          # the only read-back of a flush: R + 1 offsets
            for bi, (local_i, packed, ref, _axes) in enumerate(pending):
                lo, hi = int(offs[bi]), int(offs[bi + 1])
                if hi > lo:     # trimmed copies: a slice would pin the whole capacity-sized buffer of this flush until the run ends
                    dev_parts.append((out.xyz[lo:hi].clone(), out.rgb[lo:hi].clone(), out.err[lo:hi].clone()))
                    emit(local_i, packed, None, None, None, None, dev_parts[-1])
            pending.clear()

        group: List[Tuple[int, _PackedReference, hb.ReferenceInputs, object, int]] = []     # sampled mode, several references per call

        def flush_group() -> None:
            if not group:
                return
            items = list(group)
            group.clear()
            try:
                handle = hot.launch_sampled_multi([g[2] for g in items], items[0][3], [g[4] for g in items])
                res = hot.finish_sampled(handle)
            except Exception as ex:
                # upstream isolates failures per reference (core/pipeline.py:874-879): redo the group one reference at a time,
                # so that only the reference that cannot be processed is dropped
                log.warn(f"Grouped triangulation of refs {[g[1].ref_uid for g in items]} failed ({ex}); retrying one by one")
                for li, pk, rf, ax_, sd in items:
                    try:
                        one = hot.finish_sampled(hot.launch_sampled(rf, ax_, sd))
                    except Exception as ex1:
                        log.error(f"Triangulation error for ref {pk.ref_uid}: {ex1}")
                        continue
                    if one is not None:
                        dev_parts.append((one.xyz, one.rgb, one.err))
                        emit(li, pk, None, None, None, None, dev_parts[-1])
                return
            if res is None:
                return
            for bi, (li, pk, _ref, _axes, _seed) in enumerate(items):
                lo, hi = int(res.ref_offsets[bi]), int(res.ref_offsets[bi + 1])
                if hi > lo:
                    dev_parts.append((res.xyz[lo:hi].clone(), res.rgb[lo:hi].clone(), res.err[lo:hi].clone()))
                    emit(li, pk, None, None, None, None, dev_parts[-1])

        pend_norm: List[Tuple[int, _PackedReference, hb.ReferenceInputs, object, object]] = []   # default sampled mode: aggregated map on its way to the host

        def promote_one() -> None:
            """The oldest reference whose aggregated map has reached the host: upstream's normaliser from it, then its fused call."""
            li, pk, rf, ax_, handle = pend_norm.pop(0)
            try:
                s_up = hot.finish_normaliser(handle)
                inflight.append((li, pk, hot.launch_sampled(rf, ax_, None, s_override=s_up, batch=handle[0])))
            except Exception as ex:
                log.error(f"Triangulation error for ref {pk.ref_uid}: {ex}")

        def drain_pipelined() -> None:
            while pend_norm:
                promote_one()
            while inflight:
                finish_one()

        def finish_one() -> None:
            """Collect the oldest launched reference (sampled mode) and emit it: references are emitted in launch order."""
            li, pk, handle = inflight.pop(0)
            try:
                res = hot.finish_sampled(handle)
            except Exception as ex:
                log.error(f"Triangulation error for ref {pk.ref_uid}: {ex}")
                return
            if res is None:
                return
            # the survivors stay where they are: the trimmed device copy (finish_sampled) is what the previews, the streamed output, the exchange
            # and the device-side writers consume; the host arrays of the result are ONE copy at the end of the run
            dev_parts.append((res.xyz, res.rgb, res.err))
            emit(li, pk, None, None, None, None, dev_parts[-1])

        for local_i, packed in enumerate(prefetch):
            _raise_if_cancelled(cancel_requested)
            if progress_callback is not None:
                done = local_i + 1
                progress_callback(10.0 + (float(done - 1) / max(1, total_refs)) * 80.0,
                                  f"Matching {done}/{total_refs} | {done / max(0.001, time.time() - t0):.1f} it/s")
            if packed is None:
                if feat_cache is not None:
                    feat_cache.advance(local_i)       # a skipped reference is a schedule position too: its cameras' last uses pass
                continue
            _raise_if_cancelled(cancel_requested)
            from PIL import Image
            want_debug = debug_state is not None and debug_state.is_enabled()
            dev_images = bool(getattr(matcher, "accepts_device_images", False))
            if packed.raw:
                packed = hot.prepare_on_device(packed, (w_match, h_match), need_host=want_debug or not dev_images)
            kw_keys = {"keys": (packed.ref_index, list(packed.nbr_indices))} if feat_cache is not None else {}
            if packed.dev is not None and dev_images:
                results = matcher.match_grids_batch(packed.dev["image"], list(packed.dev["nbr_images"]), **kw_keys)
            else:
                results = matcher.match_grids_batch(Image.fromarray(np.ascontiguousarray(packed.image)),
                                                    [Image.fromarray(np.ascontiguousarray(a)) for a in packed.nbr_images], **kw_keys)
            if feat_cache is not None:
                feat_cache.advance(local_i)
            _raise_if_cancelled(cancel_requested)
            if not results:
                continue
            first_pair = pair_counter + 1
            pair_counter += len(results)
            if local_i >= n_sharded_mine:
                rep_pairs += len(results)
            warps, certs = _as_device_maps(results, dev)
            H, W = certs[0].shape
            axes = None
            if warps[0].shape[-1] == 2:
                ax = getattr(matcher, "reference_axes", None)
                if callable(ax):
                    a0, a1 = ax(H, W)
                    axes = (torch.as_tensor(a0).to(dev, torch.float32).contiguous(), torch.as_tensor(a1).to(dev, torch.float32).contiguous())
            ref = hot.inputs(packed, warps, certs)

            if config.triangulation_mode == "dense":
                pending.append((local_i, packed, ref, axes))
                if len(pending) >= int(config.refs_per_launch):
                    flush_dense()
                continue

            rng = _reference_rng(config.seed, packed.ref_uid) if per_ref_rng else stream_rng
            dseed = _reference_seed(config.seed, packed.ref_uid) if per_ref_rng else None
            if per_ref_rng and int(config.refs_per_launch) > 1 and hot.can_launch_ahead(want_debug, True, int(H), int(W)):
                # every reference has its own stream: refs_per_launch of them share one fused call (lfd_triangulate_sampled_multi)
                group.append((local_i, packed, ref, axes, dseed))
                if len(group) >= int(config.refs_per_launch):
                    flush_group()
                continue
            flush_group()
            if hot.can_pipeline_normaliser(want_debug, per_ref_rng, int(H), int(W)):
                # upstream's normaliser (the default) without a host wait in the launch stream: this reference's aggregated map starts
                # its way to the host; the reference before it - whose map has arrived meanwhile - gets its sum and its fused launch; the
                # one before that is collected.  The fused calls are issued in reference order: one MT19937 stream, as upstream.
                try:
                    pend_norm.append((local_i, packed, ref, axes, hot.begin_normaliser(ref, axes)))
                except Exception as ex:
                    log.error(f"Triangulation error for ref {packed.ref_uid}: {ex}")
                while len(pend_norm) > 1:
                    promote_one()
                while len(inflight) > 1:
                    finish_one()
                continue
            while pend_norm:            # (a run that leaves the pipelined mode - a debug preview switched on - first issues what is pending, in order)
                promote_one()
            if hot.can_launch_ahead(want_debug, per_ref_rng, int(H), int(W)):
                # reference i is launched (asynchronously, counts read back behind an event) BEFORE reference i-1 is collected:
                # the host side of one reference - packing, descriptor upload, Python - runs under the kernels of the other
                try:
                    inflight.append((local_i, packed, hot.launch_sampled(ref, axes, dseed)))
                except Exception as ex:
                    log.error(f"Triangulation error for ref {packed.ref_uid}: {ex}")
                while len(inflight) > 1:
                    finish_one()
                continue
            while inflight:
                finish_one()
            try:
                out, best = hot.sampled(ref, axes, rng, dseed, need_best=want_debug)
            except Exception as ex:
                log.error(f"Triangulation error for ref {packed.ref_uid}: {ex}")
                out = None
            if out is None:
                continue
            dbg = None
            if want_debug:
                dbg = {"matches": hot.debug_matches(ref, out.cell, out.slot, axes, best),
                       "pair_index": {j: first_pair + j for j in range(len(certs))}}
            dev_parts.append((out.xyz.clone(), out.rgb.clone(), out.err.clone()))
            emit(local_i, packed, None, None, None, dbg, dev_parts[-1])
        flush_group()
        drain_pipelined()
        flush_dense()
    except BaseException as exc:
        if world == 1:
            raise
        # sharded run: the other ranks are heading for the collectives below; tell them instead of leaving them blocked
        rank_status = 1 if isinstance(exc, PipelineCancelled) else 2
        rank_error = exc
    finally:
        if prefetch is not None:
            prefetch.close()
        if feat_cache is not None:           # the features of this run's cameras: nothing of them outlives the run
            feat_cache.clear()
            try:
                matcher.set_feature_cache(None)
            except Exception as exc:
                log.warn(f"Releasing the feature cache failed: {exc}")
        if own_matcher and matcher is not None:
            try:
                matcher.close()
            except Exception as exc:
                log.warn(f"Matcher cleanup failed: {exc}")
        if shard_stream is not None:
            try:
                shard_stream.finish()         # rank 0 receives what is left (a rank that stopped early sends empty references)
            except Exception as exc:          # (incl. a writer failure kept until the peers were drained: the run has failed on this rank)
                log.error(f"The sharded output stream failed: {exc}")
                if rank_status == 0:
                    rank_status, rank_error = 2, exc
        if shared_file is not None:
            try:
                shared_file.finish()          # the rounds that are left, the last byte ranges, the vertex count in the header (rank 0)
            except Exception as exc:
                log.error(f"The shared-file output stream failed: {exc}")
                if rank_status == 0:
                    rank_status, rank_error = 2, exc
        if xchg is not None:
            try:
                xchg_result = xchg.finish()   # closes the rounds that are left (empty ones on a rank that stopped early) and waits for the collectives
            except Exception as exc:
                log.error(f"The overlapped exchange failed: {exc}")
                if rank_status == 0:
                    rank_status, rank_error = 2, exc
        if stream_writer is not None:
            try:
                stream_writer.close()         # patches the vertex count into the header
            except Exception as exc:
                log.warn(f"Closing the streamed output failed: {exc}")
        if hot is not None:
            hot.close()
        if debug_state:
            debug_state.release_waiters()
        gc.collect()
        if torch.cuda.is_available():
            torch.cuda.empty_cache()

    if world > 1:
        if rank_status == 0 and _cancelled(cancel_requested):
            rank_status = 1
        worst = lfd_dist.agree_on_status(rank_status, dist, dev)
        if worst:
            if rank_error is not None:
                raise rank_error
            raise PipelineCancelled("Cancelled") if worst == 1 else RuntimeError("dense pipeline failed on another rank")
    else:
        _raise_if_cancelled(cancel_requested)
    if progress_callback:
        progress_callback(90.0, "Finalizing triangulation...")

    counts = np.asarray(counts_local, np.int64)
    device_points = None
    if world == 1 and dev_parts:
        device_points = (torch.cat([p[0] for p in dev_parts], 0), torch.cat([p[1] for p in dev_parts], 0),
                         torch.cat([p[2] for p in dev_parts], 0))
    if xyz_parts and all(x is not None for x in xyz_parts):
        xyz, rgb, err = np.concatenate(xyz_parts, 0), np.concatenate(rgb_parts, 0), np.concatenate(err_parts, 0)
    elif xyz_parts and device_points is not None:      # the survivors cross PCIe once, here
        xyz, rgb, err = (t.cpu().numpy() for t in device_points)
    else:
        xyz, rgb, err = np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32), np.zeros((0,), np.float32)
    n_points_global = int(xyz.shape[0])
    if world > 1 and xchg_result is not None:
        # the rounds travelled beside the compute; what is left is to name the parts of the ordered records
        recs, counts = xchg_result
        if n_rep and consumes_cloud:
            # sharded part | replicated part: the replicated references are the LAST of the list, so the ordered cloud is a concatenation
            recs = torch.cat([recs.reshape(-1)] + [r_.reshape(-1) for r_ in rep_parts]) if rep_parts else recs
            if xchg.record != lfd_dist.RECORD_PLY:
                recs = recs.reshape(-1, 7)
            counts = np.concatenate([counts, np.asarray(counts_local[n_sharded_mine:], np.int64)])
        elif n_rep:
            counts = np.concatenate([counts, np.zeros(n_rep, np.int64)])      # (a rank that does not receive the cloud did not compute them)
        if xchg.record == lfd_dist.RECORD_PLY:
            gx, gc_ = lfd_dist.points_from_ply_records(recs)
            ge = torch.zeros((int(gx.shape[0]),), dtype=torch.float32, device=gx.device)
        else:
            gx, gc_, ge = recs[:, 0:3].contiguous(), recs[:, 3:6].contiguous(), recs[:, 6].contiguous()
        xyz, rgb, err = gx.cpu().numpy(), gc_.cpu().numpy(), ge.cpu().numpy()
        device_points = (gx, gc_, ge)
        n_points_global = int(counts.sum())
        # (replicated references were processed by several ranks: they count once, on rank 0)
        mine_once = (refs_with_points - (rep_refs_with_points if rank else 0), pair_counter - (rep_pairs if rank else 0))
        t = torch.tensor(list(mine_once), dtype=torch.int64, device=lfd_dist._collective_device(gx, dist))
        dist.all_reduce(t)
        refs_with_points, pair_counter = int(t[0].item()), int(t[1].item())
    elif world > 1:       # the one exchange step: the survivors travel over RCCL from where they already are (HBM), ordered by reference
        if dev_parts:
            lx, lc, le = (torch.cat([p[i] for p in dev_parts], 0) for i in range(3))
        else:
            lx, lc, le = (torch.zeros((0, 3), device=dev), torch.zeros((0, 3), device=dev), torch.zeros((0,), device=dev))
        if str(getattr(config, "exchange", "all_gather")) == "gather_to_root":
            # only rank 0 consumes the cloud (it writes the file): every record travels once, straight to its place on rank 0;
            # the other ranks return their own shard
            gx, gc_, ge, counts = lfd_dist.gather_to_root_by_reference(lx, lc, le, counts_local, len(refs_local), dist)
        else:
            gx, gc_, ge, counts = lfd_dist.all_gather_by_reference(lx, lc, le, counts_local, len(refs_local), dist)
        xyz, rgb, err = gx.cpu().numpy(), gc_.cpu().numpy(), ge.cpu().numpy()
        device_points = (gx, gc_, ge)
        n_points_global = int(counts.sum())
        t = torch.tensor([refs_with_points, pair_counter], dtype=torch.int64,
                         device=lfd_dist._collective_device(gx, dist))
        dist.all_reduce(t)
        refs_with_points, pair_counter = int(t[0].item()), int(t[1].item())

    if n_points_global == 0:
        raise RuntimeError("No points triangulated. Try adjusting parameters.")
    return PipelineResult(xyz=xyz, rgb=rgb, err=err, elapsed_seconds=time.time() - t0,
                          pairs_processed=refs_with_points, pairs_matched=pair_counter, points_per_reference=counts,
                          device_points=device_points,
                          streamed_path=config.output_path if (stream_writer is not None or shard_stream is not None or shared_file is not None) else None)
