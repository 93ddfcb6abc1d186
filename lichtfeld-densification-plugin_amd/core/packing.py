"""Loading side of the driver loop: one package per reference (its image, masks and neighbours), made by a bounded, ORDERED prefetcher.

Replaces upstream core/pipeline.py:132-252 (``_pack_reference_batch``) and core/threaded_dataloader.py:42-241 (a completion-ordered
thread pool: with several workers upstream consumes references in whatever order their packages finish, so its RNG stream and output order
depend on thread timing).  Here results come back in submission order."""
from __future__ import annotations

import dataclasses
from concurrent.futures import Future, ThreadPoolExecutor
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import stages
from .hostlog import log
from .image_io import black_out, decode_mask_l, decode_rgb_u8, load_mask01, load_rgb_u8
from .types import CameraRecord


class PipelineCancelled(RuntimeError):
    """Raised when a running dense pipeline is cancelled."""


def cancelled(cb: Optional[Callable[[], bool]]) -> bool:
    """upstream core/pipeline.py:255-267: an exception of the callback is a warning, not a cancellation"""
    if cb is None:
        return False
    try:
        return bool(cb())
    except Exception as exc:
        log.warn(f"Cancellation callback failed: {exc}")
        return False


def raise_if_cancelled(cb) -> None:
    if cancelled(cb):
        raise PipelineCancelled("Cancelled")


@dataclasses.dataclass
class PackedReference:
    position: int                   # position in refs_local
    ref_index: int                  # index into camera_records
    ref_uid: int
    image: np.ndarray               # (h,w,3) u8, masked pixels blacked out
    mask_a: Optional[np.ndarray]
    nbr_indices: List[int]
    nbr_images: List[np.ndarray]
    nbr_masks: List[Optional[np.ndarray]]
    # device_image_prep: ``image`` / ``nbr_images`` / masks above hold the DECODED arrays (any size, masks as "L") until
    # HotPath.prepare_on_device has resized them on the GPU; ``dev`` then holds the prepared device tensors
    raw: bool = False
    dev: Optional[dict] = None


def _load_view(cam: CameraRecord, size_wh: Tuple[int, int], raw: bool, who: str):
    """(image, mask) of one camera: prepared on the host (resized, masked pixels black, mask as {0,1}) or - ``raw`` - as decoded.  An
    unreadable image raises; an unreadable mask is a warning and no mask (upstream core/pipeline.py:163-171,196-204)."""
    img = decode_rgb_u8(cam.image_path) if raw else load_rgb_u8(cam.image_path, size_wh)
    mask = None
    if getattr(cam, "mask_path", None):
        try:
            if raw:
                mask = decode_mask_l(cam.mask_path)
            else:
                mask = load_mask01(cam.mask_path, size_wh)
                img = black_out(img, mask)
        except Exception as exc:
            log.warn(f"Failed to load/apply mask for {who} {cam.uid}: {exc}")
            mask = None
    return img, mask


def pack_reference(position: int, ref_index: int, cams: Sequence[CameraRecord], nn_table, nns_per_ref: int,
                   size_wh: Tuple[int, int], cancel, raw: bool = False, stage=None) -> Optional[PackedReference]:
    """Load and pre-process one reference and its neighbours (upstream core/pipeline.py:132-227): same skip / warn rules.  ``raw``: decode
    only; the resize / mask / black-out steps then run on the GPU (``HotPath.prepare_on_device``).  ``stage(cam_index, image, mask)``: what a
    pack thread does with a freshly decoded view before it hands the package over (``HotPath.stage_decoded``: the upload, here instead of on the
    driver's thread)."""
    if cancelled(cancel):
        return None
    cam = cams[ref_index]
    try:
        img_a, mask_a = _load_view(cam, size_wh, raw, "reference")
        if stage is not None:
            img_a, mask_a = stage(ref_index, img_a, mask_a)
    except Exception as exc:
        log.warn(f"Failed to load reference {cam.image_path}: {exc}")
        return None
    local = nn_table[ref_index][:nns_per_ref]
    if len(local) == 0:
        return None
    nbr_indices, nbr_images, nbr_masks = [], [], []
    for n in local:
        n = int(n)
        if cancelled(cancel):
            return None
        nb = cams[n]
        if nb.uid == cam.uid:
            continue
        try:
            img_b, mask_b = _load_view(nb, size_wh, raw, "neighbor")
            if stage is not None:
                img_b, mask_b = stage(n, img_b, mask_b)
        except Exception as exc:
            log.warn(f"Failed to load neighbor {nb.uid}: {exc}")
            continue
        nbr_indices.append(n)
        nbr_images.append(img_b if raw else np.asarray(img_b, dtype=np.uint8))
        nbr_masks.append(mask_b)
    if not nbr_images:
        return None
    return PackedReference(position=position, ref_index=ref_index, ref_uid=int(cam.uid),
                           image=img_a if raw else np.asarray(img_a, dtype=np.uint8), mask_a=mask_a, nbr_indices=nbr_indices,
                           nbr_images=nbr_images, nbr_masks=nbr_masks, raw=raw)


class OrderedPrefetcher:
    """Bounded look-ahead over an indexable job list; results come back in submission order, so the
    order references are consumed in (and therefore the RNG stream and the output order) does not
    depend on thread timing - unlike upstream's completion-ordered pool
    (core/threaded_dataloader.py:190-222).  Every job runs with the run's stage clock bound to its thread."""

    def __init__(self, jobs: Sequence[Callable[[], object]], workers: int, window: int, clock=None):
        self._jobs = list(jobs)
        self._pool = ThreadPoolExecutor(max_workers=max(1, int(workers)), thread_name_prefix="lfd-pack")
        self._window = max(1, int(window))
        self._futures: Dict[int, Future] = {}
        self._next_submit = 0
        self._next_yield = 0
        self._closed = False
        self._clock = clock

    def _run(self, job):
        with stages.bound(self._clock):
            return job()

    def _fill(self) -> None:
        while self._next_submit < len(self._jobs) and self._next_submit - self._next_yield < self._window:
            self._futures[self._next_submit] = self._pool.submit(self._run, self._jobs[self._next_submit])
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
