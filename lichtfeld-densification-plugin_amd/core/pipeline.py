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

The driver is four parts, each in a module of its own:
    core/packing.py     load / decode one reference and its neighbours; the ordered prefetcher
    core/hotpath.py     the per-run context of the C-ABI and the calls that triangulate a reference
    core/strategies.py  SampledLoop | DenseBatcher | DensePlyStreamer: how matched references are scheduled onto the hot path
    core/sinks.py       RunOutputs: accumulator, previews, streamed file, debug previews, the sharded run's exchange; PipelineResult
and this file is the loop that connects them (``run_dense_pipeline``).

Two triangulation modes (``DensePipelineConfig.triangulation_mode``):
  * "sampled" (default) reproduces upstream: coverage sampling picks ~0.85*M + <=625 cells per
    reference from the aggregated certainty, those cells are triangulated and emitted in upstream's
    per-neighbour group order.  With ``per_reference_rng=False`` and one GPU the legacy NumPy stream
    is consumed exactly as upstream consumes it (one stream, reference after reference).
  * "dense" sends every grid cell through the fused kernel (survivors in raster order per reference).

Multi-GPU: when ``torch.distributed`` is initialised with world_size > 1 the reference list is dealt
round-robin to the ranks and the survivors are exchanged in reference order
(core/distributed.py); sampling then uses one RNG stream per reference.
"""
from __future__ import annotations

import dataclasses
import gc
import time
from typing import Callable, List, Optional

import numpy as np
import torch

from . import distributed as lfd_dist
from . import hip_backend as hb
from .debug_viz import MatchDebugState
from .hostlog import log
from .hotpath import HotPath
from .matcher import RomaMatcher, has_cached_romav2_weights, romav2_cached_weights_paths
from .packing import OrderedPrefetcher, PackedReference, PipelineCancelled, cancelled, pack_reference, raise_if_cancelled
from .scheduler import FeatureCache, PairSchedule
from .sinks import PipelineResult, RunOutputs, ShardPlan
from .stages import NULL_CLOCK
from .strategies import DenseBatcher, DensePlyStreamer, Matched, SampledLoop, reference_seed
from .types import CameraRecord, DensePipelineConfig

_HotPath = HotPath                       # (the name tests and profiles patch)
_reference_seed = reference_seed


def _estimate_total_pairs(refs_local, nn_table, uids, nns_per_ref) -> int:
    return sum(sum(1 for n in nn_table[r][:nns_per_ref] if uids[n] != uids[r]) for r in refs_local)


def _resolve_backend(config: DensePipelineConfig, backend: Optional[str], device):
    """(config with the effective backend, torch device).  A "device" run without a GPU raises: it never turns into a host run."""
    if backend is not None and str(backend) != config.backend:
        if backend not in ("device", "host"):
            raise ValueError("backend must be 'device' or 'host'")
        config = dataclasses.replace(config, backend=str(backend))        # (validated again: device-only settings on the host backend raise)
    if config.backend == "host":
        return config, torch.device("cpu")
    if not torch.cuda.is_available():
        raise hb.HipBackendError("no GPU visible: the dense-initialisation hot path has no CPU fallback "
                                 "(backend=\"host\" selects the CPU twin explicitly)")
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    if dev.type != "cuda":
        raise hb.HipBackendError(f"backend=\"device\" needs a cuda (HIP) device, got {dev}")
    return config, dev


def _make_matcher(config, dev, progress_callback):
    """upstream core/pipeline.py:795-812: the progress text depends on whether the weights are cached"""
    cached = has_cached_romav2_weights()
    _announce(progress_callback, cached)
    if not cached:
        log.info("RoMaV2 weights not found in cache; expected cache paths: " + ", ".join(romav2_cached_weights_paths()))
    matcher = RomaMatcher(device=str(dev), mode="outdoor", setting=config.roma_setting, pairs_per_forward=int(config.pairs_per_forward))
    if not cached and progress_callback is not None:
        progress_callback(10.0, "RoMa v2 model installation complete. Starting matching...")
    return matcher


def _announce(progress_callback, cached: bool) -> None:
    msg = "Initializing RoMa v2 model..." if cached else "Installing model weights..."
    if progress_callback is not None:
        progress_callback(10.0, msg)
    log.info(msg)


def _match_reference(local_i: int, packed: PackedReference, matcher, hot: HotPath, outputs: RunOutputs, feat_cache, size_wh, want_debug: bool,
                     cancel) -> Optional[Matched]:
    """One package through the matcher (upstream core/pipeline.py:856-872): prepared on the device first when it came decoded; the maps stay
    where the matcher left them."""
    from PIL import Image
    dev, clock = hot.dev, hot.clock
    dev_images = bool(getattr(matcher, "accepts_device_images", False))
    if packed.raw:
        packed = hot.prepare_on_device(packed, size_wh, need_host=want_debug or not dev_images)
    kw_keys = {"keys": (packed.ref_index, list(packed.nbr_indices))} if feat_cache is not None else {}
    with clock.stage("match"):
        if packed.dev is not None and dev_images:
            results = matcher.match_grids_batch(packed.dev["image"], list(packed.dev["nbr_images"]), **kw_keys)
        else:
            results = matcher.match_grids_batch(Image.fromarray(np.ascontiguousarray(packed.image)),
                                                [Image.fromarray(np.ascontiguousarray(a)) for a in packed.nbr_images], **kw_keys)
        if feat_cache is not None:
            feat_cache.advance(local_i)
        raise_if_cancelled(cancel)
        if not results:
            return None
        first_pair = outputs.note_pairs(local_i, len(results))
        warps = [_as_device_map(w, dev) for w, _c in results]
        certs = [_as_device_map(c, dev) for _w, c in results]
    H, W = certs[0].shape
    axes = None
    ax = getattr(matcher, "reference_axes", None)
    if warps[0].shape[-1] == 2 and callable(ax):
        a0, a1 = ax(H, W)
        axes = (torch.as_tensor(a0).to(dev, torch.float32).contiguous(), torch.as_tensor(a1).to(dev, torch.float32).contiguous())
    return Matched(local_i, packed, hot.inputs(packed, warps, certs), axes, int(H), int(W), first_pair, want_debug)


def _as_device_map(t, dev) -> torch.Tensor:
    """A matcher output as the contiguous f32 tensor on ``dev`` the kernels read in place - which it normally already is (no call, no copy then)."""
    if isinstance(t, torch.Tensor) and t.device == dev and t.dtype == torch.float32 and t.is_contiguous() and not t.requires_grad:
        return t
    return torch.as_tensor(t).detach().to(dev, torch.float32).contiguous()


def _make_strategy(config, plan: ShardPlan, hot: HotPath, outputs: RunOutputs, per_ref_rng: bool, debug_state, auto_group: bool = False):
    """``auto_group``: refs_per_launch was 0 - the strategy bounds the automatic group by the bytes its buffers take at the run's grid."""
    debug_on = debug_state is not None and debug_state.is_enabled()
    if DensePlyStreamer.applies(config, plan, hot.on_host, outputs, debug_on):
        return DensePlyStreamer(hot, outputs, config, auto_group)
    if config.triangulation_mode == "dense":
        return DenseBatcher(hot, outputs, config, auto_group)
    return SampledLoop(hot, outputs, config, per_ref_rng, auto_group)


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
    stage_clock=None,
) -> PipelineResult:
    """See module docstring.  ``matcher`` / ``densifier`` / ``device`` are injection points for
    tests and for callers that keep a warm model; by default a RomaMatcher is created (and released)
    per run exactly like upstream.  ``backend`` ("device" | "host", default ``config.backend``): "host" runs the per-reference
    path on the CPU twin of the C-ABI (HostDensifier + core/sampling.py) - upstream's CPU-only configuration, chosen by the
    caller; a "device" run without a GPU raises HipBackendError, it never turns into a host run.  ``stage_clock``: a
    core.stages.StageClock that receives the run's time per stage (``PipelineResult.stages``)."""
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    config.validate()
    config, dev = _resolve_backend(config, backend, device)
    per_ref_rng = bool(config.per_reference_rng) or world > 1
    # refs_per_launch = 0 (the default): several references per launch where the results do not depend on it and nobody watches the run proceed
    previews = (on_sequential_viz is not None and int(config.viz_interval) > 0) or debug_state is not None
    auto_group = int(config.refs_per_launch) == 0
    if int(config.refs_per_launch) != config.launch_group(world, previews):
        config = dataclasses.replace(config, refs_per_launch=config.launch_group(world, previews))
    clock = stage_clock if stage_clock is not None else NULL_CLOCK
    uids = [c.uid for c in camera_records]
    total_pairs_est = _estimate_total_pairs(refs_local, nn_table, uids, config.nns_per_ref)
    if debug_state:
        debug_state.set_total_pairs(total_pairs_est)
    # What a sharded run exchanges while it proceeds is decided from the configuration and the world ALONE and set up here, before anything
    # can fail: every rank then reaches the matching finish() in `finally` whatever went wrong on it (core/sinks.py::ShardLink).
    plan = ShardPlan.make(config, len(refs_local), world, rank)
    outputs = RunOutputs(config, plan, dist, dev, camera_records, on_sequential_viz=on_sequential_viz, debug_state=debug_state,
                         cancel_requested=cancel_requested, total_pairs_est=total_pairs_est)
    t0 = time.time()
    own_matcher = matcher is None
    prefetch = hot = feat_cache = strategy = None
    rank_status, rank_error = 0, None         # 0 fine, 1 cancelled, 2 failed: agreed on by all ranks before the exchange step (core/distributed.py)
    try:
        if own_matcher:
            matcher = _make_matcher(config, dev, progress_callback)
        else:
            _announce(progress_callback, True)
        raise_if_cancelled(cancel_requested)
        size_wh = (int(matcher.w_resized), int(matcher.h_resized))
        hot = HotPath(camera_records, config, float(matcher.sample_thresh), size_wh[0], size_wh[1], dev, densifier, clock=clock)
        if not hot.on_host:
            hot.dens.seed_rng(int(config.seed))      # upstream: np.random.seed(config.seed) (core/pipeline.py:793); the host backend draws from its own RandomState
        # N4: every camera's backbone features once per run, kept exactly until their last use (core/scheduler.py)
        schedule = PairSchedule(refs_local, nn_table, uids, config.nns_per_ref, positions=plan.my_positions)
        if bool(getattr(matcher, "supports_feature_keys", False)):
            feat_cache = FeatureCache(schedule.last_use) if config.share_features else None
            matcher.set_feature_cache(feat_cache)    # always (re)set: an injected, warm matcher may still hold the cache of an earlier run
        stage = (lambda ci, im, mk: hot.stage_decoded(ci, size_wh, im, mk)) if config.device_image_prep else None
        jobs = [(lambda p=p: pack_reference(p, refs_local[p], camera_records, nn_table, config.nns_per_ref, size_wh, cancel_requested,
                                            raw=bool(config.device_image_prep), stage=stage)) for p in plan.my_positions]
        # (upstream's `prefetch_packages` bounds the queue of FINISHED packages while all of its workers keep loading: the look-ahead here is at
        # least one package per worker, so that none of them idles)
        prefetch = OrderedPrefetcher(jobs, workers=int(config.pack_workers), window=max(int(config.prefetch_packages), int(config.pack_workers)), clock=clock)
        outputs.open()
        strategy = _make_strategy(config, plan, hot, outputs, per_ref_rng, debug_state, auto_group)
        total_refs = len(plan.my_positions)
        for local_i, packed in enumerate(prefetch):
            raise_if_cancelled(cancel_requested)
            if progress_callback is not None:
                done = local_i + 1
                progress_callback(10.0 + (float(done - 1) / max(1, total_refs)) * 80.0,
                                  f"Matching {done}/{total_refs} | {done / max(0.001, time.time() - t0):.1f} it/s")
            if packed is None:
                if feat_cache is not None:
                    feat_cache.advance(local_i)       # a skipped reference is a schedule position too: its cameras' last uses pass
                continue
            raise_if_cancelled(cancel_requested)
            want_debug = debug_state is not None and debug_state.is_enabled()
            m = _match_reference(local_i, packed, matcher, hot, outputs, feat_cache, size_wh, want_debug, cancel_requested)
            if m is not None:
                strategy.submit(m)
        strategy.drain()
    except BaseException as exc:
        if world == 1:
            raise
        # sharded run: the other ranks are heading for the collectives below; tell them instead of leaving them blocked
        rank_status, rank_error = (1 if isinstance(exc, PipelineCancelled) else 2), exc
    finally:
        failed = _release(prefetch, strategy, feat_cache, matcher if own_matcher else None, matcher, outputs, hot, debug_state)
        if failed is not None and rank_status == 0:
            rank_status, rank_error = 2, failed

    if world > 1:
        if rank_status == 0 and cancelled(cancel_requested):
            rank_status = 1
        worst = lfd_dist.agree_on_status(rank_status, dist, dev)
        if worst:
            if rank_error is not None:
                raise rank_error
            raise PipelineCancelled("Cancelled") if worst == 1 else RuntimeError("dense pipeline failed on another rank")
    else:
        raise_if_cancelled(cancel_requested)
    if progress_callback:
        progress_callback(90.0, "Finalizing triangulation...")
    return outputs.result(t0, clock)


def _release(prefetch, strategy, feat_cache, own_matcher, matcher, outputs: RunOutputs, hot, debug_state) -> Optional[BaseException]:
    """The run's ``finally`` (upstream core/pipeline.py:900-907 + what a sharded run owes its peers).  Returns a failure of the exchange."""
    if prefetch is not None:
        prefetch.close()
    if strategy is not None:
        strategy.close()
    if feat_cache is not None:           # the features of this run's cameras: nothing of them outlives the run
        feat_cache.clear()
        try:
            matcher.set_feature_cache(None)
        except Exception as exc:
            log.warn(f"Releasing the feature cache failed: {exc}")
    if own_matcher is not None:
        try:
            own_matcher.close()
        except Exception as exc:
            log.warn(f"Matcher cleanup failed: {exc}")
    failed = outputs.finish()
    if hot is not None:
        hot.close()
    if debug_state:
        debug_state.release_waiters()
    # upstream's `gc.collect(); torch.cuda.empty_cache()` (core/pipeline.py:904-907) exists to give the model's memory back: the collection runs where a
    # model was torn down (this run built it; RomaMatcher.close collects too), the run's own buffers - no reference cycles - go back to the driver either way
    if own_matcher is not None:
        gc.collect()
    if torch.cuda.is_available():
        torch.cuda.empty_cache()
    return failed
