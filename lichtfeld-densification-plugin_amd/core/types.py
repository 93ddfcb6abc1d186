"""Boundary types of the dense-initialisation path.

``DensePipelineConfig`` keeps the 18 fields of the upstream dataclass with the same names, order and
defaults (upstream core/config.py:7-26) so that the GUI panel / CLI can construct it unchanged, and
adds twelve settings of this implementation *after* them (all defaulted, so positional construction still
works) plus one ``experimental`` dict for the knobs that exist only because an experiment was run.
``problem()`` names every combination that cannot run: nothing is silently ignored.
``CameraRecord`` keeps upstream's per-camera record (core/camera_models.py:10-28): f32 intrinsics and
world-to-camera pose, the projection ``P = K [R|t]`` and the centre ``C = -R^T t``.
"""
from __future__ import annotations

import dataclasses
from typing import Dict, Optional

import numpy as np

TRIANGULATION_MODES = ("sampled", "dense")
AUTO_REFS_PER_LAUNCH = 16         # refs_per_launch = 0 (the default) where several references per launch are possible and nobody waits for previews

# Knobs that exist because an experiment was run: they stay reachable for the profiles and the tests that pin them, but they are not
# part of the configuration surface a caller is expected to touch.  ``DensePipelineConfig(experimental={...})``; an unknown key is an error.
EXPERIMENTAL_DEFAULTS = {
    # sharded runs (torch.distributed, world > 1): the exchange happens in ROUNDS beside the compute (core/distributed.py::OverlappedExchange)
    # instead of ONE exchange after the last reference; same sequence either way.  Not used together with stream_output (one communicator at a time).
    "exchange_overlap": True,
    # references per round of the overlapped exchange (0: refs_per_launch, at least 4)
    "exchange_round": 0,
    # what the overlapped exchange moves: "f32" the 28-byte rows of the result; "ply" the 15-byte PLY vertex records packed on the device
    # (positions exact, colours as the writer quantises them, no reprojection error: the result's rgb is then u8 / 255 and err zero);
    # "auto": "ply" when the output is a .ply and no voxel filter has to see f32 colours
    "exchange_records": "f32",
    # fraction of the reference list (its LAST references) computed by every rank that receives the cloud instead of exchanged
    # (core/distributed.py::plan_replication says when that pays: never with a real matcher in the loop)
    "exchange_replicate": 0.0,
    # sharded run + stream_output on ONE node: every rank writes its own byte ranges of the output file, only counts travel
    "stream_shared_file": False,
    # dense mode: the kernel with UNORDERED retirement + lfd_order_segments (bit-identical result, ~6 % less kernel time)
    "dense_tile_segments": False,
    # hand upstream's own fundamental matrices (np.linalg.inv products, computed on the host exactly as upstream computes them) to the
    # kernels instead of the closed-form F the library derives from the camera table (agrees to ~2e-6 relative, not bit for bit)
    "upstream_fundamental": True,
    # backend="host": threads of the CPU twin (0: all hardware threads)
    "host_threads": 0,
}


@dataclasses.dataclass
class DensePipelineConfig:
    output_path: str
    roma_setting: str = "fast"
    roi_only_selected: bool = False
    num_refs: float = 0.8
    nns_per_ref: int = 3
    matches_per_ref: int = 10000
    certainty_thresh: float = 0.20
    reproj_thresh: float = 0.8
    sampson_thresh: float = 5.0
    min_parallax_deg: float = 0.5
    max_points: int = 0
    no_filter: bool = False
    use_masks: bool = True
    voxel_size: float = 0.0
    seed: int = 0
    viz_interval: int = 3
    prefetch_packages: int = 8
    pack_workers: int = 4
    # ---- extensions of this implementation (not present upstream; 12 + the experimental dict) -------------------------------------------
    # "sampled": upstream behaviour - coverage sampling picks ~0.85*M+tiles cells per reference and
    #            only those are triangulated.  "dense": every grid cell upstream's sampler COULD draw (best certainty after
    #            floor and masks not <= 0: a masked-out cell never is) goes through the fused kernel.
    triangulation_mode: str = "sampled"
    # references whose RoMa outputs are kept resident and triangulated by ONE kernel launch (dense mode) or ONE fused call (sampled mode: on
    # upstream's single RNG stream - lfd_triangulate_sampled_chain - or on per-reference streams; same results either way).  0 = automatic
    # (``launch_group``): AUTO_REFS_PER_LAUNCH where that is possible and nobody waits for intermediate results, else 1.
    refs_per_launch: int = 0
    # per-reference RNG stream (seed ^ uid) instead of upstream's single process-global stream;
    # forced on when references are sharded over several GPUs (results then do not depend on
    # the shard count).
    per_reference_rng: bool = False
    # where the coverage-sampling stage of the "sampled" mode runs: "device" (lfd_select_samples: the
    # whole per-reference path stays where the backend runs) or "host" (core/sampling.py: the library calls
    # upstream makes, including torch's own f32 sum as the normaliser).
    selection_backend: str = "device"
    # selection on the device, one RNG stream (upstream's mode), filter mode: normalise the sampling weights with upstream's OWN normaliser -
    # torch's CPU f32 `sum` of the weight map, computed on this host exactly as core/sampling.py:27-31 computes it - instead of the device's
    # correctly rounded exact sum (the cells drawn are then the ones upstream draws ON THE SAME MACHINE, bit for bit).  Per-reference
    # streams / sharded runs / no_filter / dense mode have no such normaliser and are not affected.
    upstream_normaliser: bool = True
    # dense mode: blend colours with upstream's f64 arithmetic (bit-identical rgb) instead of f32 (within 2.5e-7); sampled mode always does
    exact_colour: bool = False
    # write the output PLY while the run proceeds: every completed reference's survivors leave as 15-byte records packed on the device
    # (dense mode: written by the kernel itself, copied out on a side stream while the next launch computes) and are appended to
    # ``output_path``; the vertex count in the header is patched at the end.  Needs a .ply output and neither a point cap nor a voxel
    # filter (both have to see the whole cloud first).
    stream_output: bool = False
    # resize / mask / black-out the decoded images on the GPU (lfd_prepare_image / lfd_prepare_mask: Pillow's BILINEAR and
    # NEAREST arithmetic, bit for bit) instead of with PIL on the host pack threads; decoding stays on the host
    device_image_prep: bool = False
    # compute every camera's backbone (DINOv3) features once per run and share them between the references that list the camera
    # (core/scheduler.py); upstream recomputes a neighbour's features for every reference
    share_features: bool = True
    # neighbours of a reference matched per RoMa-v2 forward (1 = upstream's loop, core/matcher.py:175-188)
    pairs_per_forward: int = 1
    # where the per-reference hot path runs.  "device": the HIP kernels (needs a GPU; raises HipBackendError without one - there is
    # no fallback).  "host": the CPU twin of the C-ABI (lfd_create_host) with the host sampling stage - upstream's CPU-only configuration
    # (densify.py:148-212 run without a GPU, BASELINE config 1); chosen by the caller, never automatically.
    backend: str = "device"
    # how the survivors of a sharded run (torch.distributed, world > 1) reach the writer: "all_gather" leaves the whole cloud on
    # every rank (what BASELINE's north star names), "gather_to_root" sends every rank's records straight to their place in rank
    # 0's buffer (the other ranks return their own shard only)
    exchange: str = "all_gather"
    experimental: Dict[str, object] = dataclasses.field(default_factory=dict)

    def __post_init__(self) -> None:
        self.validate()

    def exp(self, key: str):
        """The value of an experimental knob (EXPERIMENTAL_DEFAULTS lists them)."""
        return self.experimental.get(key, EXPERIMENTAL_DEFAULTS[key])

    def exchange_record_format(self) -> str:
        rec = str(self.exp("exchange_records"))
        if rec == "auto":
            return "ply" if (str(self.output_path).lower().endswith(".ply") and float(self.voxel_size) <= 0.0) else "f32"
        return rec

    def launch_group(self, world: int = 1, previews: bool = False) -> int:
        """The references per launch / fused call a run uses.  An explicit ``refs_per_launch`` is taken as it is.  Automatic (0): one reference at
        a time - upstream's cadence - when somebody watches the run proceed (intermediate previews), on the host backend, with the host selection
        stage in sampled mode, and in sharded runs (every rank has to derive the same number from the configuration alone); else
        AUTO_REFS_PER_LAUNCH: the results are the same, the references' kernels run side by side instead of one after the other."""
        n = int(self.refs_per_launch)
        if n > 0:
            return n
        if previews or int(world) > 1 or self.backend != "device":
            return 1
        if self.triangulation_mode != "dense" and self.selection_backend != "device":
            return 1
        return AUTO_REFS_PER_LAUNCH

    def problem(self) -> Optional[str]:
        """Why this combination of settings cannot run, or None.  Every pair of settings either works together or is named here: nothing
        is silently ignored (tests/test_config_matrix.py generates the pairs)."""
        if self.triangulation_mode not in TRIANGULATION_MODES:
            return f"triangulation_mode must be one of {TRIANGULATION_MODES}, got {self.triangulation_mode!r}"
        if int(self.refs_per_launch) < 0:
            return "refs_per_launch must be >= 1 (or 0: automatic)"
        if int(self.pairs_per_forward) < 1:
            return "pairs_per_forward must be >= 1"
        if self.selection_backend not in ("device", "host"):
            return "selection_backend must be 'device' or 'host'"
        if self.backend not in ("device", "host"):
            return "backend must be 'device' or 'host'"
        if self.exchange not in ("all_gather", "gather_to_root"):
            return "exchange must be 'all_gather' or 'gather_to_root'"
        if not isinstance(self.experimental, dict):
            return "experimental must be a dict"
        unknown = sorted(set(self.experimental) - set(EXPERIMENTAL_DEFAULTS))
        if unknown:
            return f"unknown experimental setting(s) {unknown}; known: {sorted(EXPERIMENTAL_DEFAULTS)}"
        if self.exp("exchange_records") not in ("f32", "ply", "auto"):
            return "experimental['exchange_records'] must be 'f32', 'ply' or 'auto'"
        if int(self.exp("exchange_round")) < 0:
            return "experimental['exchange_round'] must be >= 0"
        if not (0.0 <= float(self.exp("exchange_replicate")) <= 1.0):
            return "experimental['exchange_replicate'] must be a fraction in [0, 1]"
        dense = self.triangulation_mode == "dense"
        if self.stream_output:
            if not str(self.output_path).lower().endswith(".ply"):
                return "stream_output writes a PLY while the run proceeds: output_path must end in .ply"
            if int(self.max_points) > 0:
                return "stream_output cannot be combined with max_points: the point cap has to see the whole cloud before anything is written"
            if float(self.voxel_size) > 0.0:
                return "stream_output cannot be combined with voxel_size: the voxel filter has to see the whole cloud before anything is written"
        if self.device_image_prep and self.backend != "device":
            return "device_image_prep resizes the images with the HIP kernels: it needs backend='device'"
        if dense and self.selection_backend == "host":
            return "selection_backend='host' names where the coverage sampling of the sampled mode runs; dense mode has no sampling stage"
        if not dense and int(self.refs_per_launch) > 1:
            if self.backend != "device" or self.selection_backend != "device":
                return ("refs_per_launch > 1 in sampled mode puts several references into one fused device call: it needs backend='device' and "
                        "selection_backend='device'")
        if not self.upstream_normaliser and not dense and (self.selection_backend == "host" or self.backend == "host"):
            return ("upstream_normaliser=False asks for the device selection's exact weight sum; the host selection stage "
                    "(selection_backend='host' or backend='host') always normalises with torch's own sum, as upstream does")
        if self.exp("dense_tile_segments"):
            if not dense:
                return "experimental['dense_tile_segments'] is a form of the dense kernel: it needs triangulation_mode='dense'"
            if self.backend != "device":
                return "experimental['dense_tile_segments'] is a form of the HIP kernel: it needs backend='device'"
        if float(self.exp("exchange_replicate")) > 0.0:
            if not self.exp("exchange_overlap"):
                return "experimental['exchange_replicate'] needs the overlapped exchange (experimental['exchange_overlap'])"
            if self.stream_output:
                return "experimental['exchange_replicate'] cannot be combined with stream_output: a streamed sharded run sends every reference to the writer"
        if self.exp("stream_shared_file") and not self.stream_output:
            return "experimental['stream_shared_file'] is a form of the streamed output: it needs stream_output"
        if self.exp("exchange_records") == "ply":
            if not self.exp("exchange_overlap"):
                return "experimental['exchange_records']='ply' is a format of the overlapped exchange (experimental['exchange_overlap'])"
            if self.stream_output:
                return ("experimental['exchange_records']='ply' cannot be combined with stream_output: a streamed sharded run exchanges its result "
                        "once, at the end, as f32 rows")
        return None

    def validate(self) -> None:
        msg = self.problem()
        if msg:
            raise ValueError(msg)


# Settings that were dataclass fields of this package before its round-5 layout and live in ``experimental`` now: still accepted as keywords and
# still readable as attributes (both with a DeprecationWarning), so that a caller written against the older surface keeps running.
_MOVED_TO_EXPERIMENTAL = ("exchange_overlap", "exchange_round", "exchange_records", "exchange_replicate", "stream_shared_file",
                          "dense_tile_segments", "upstream_fundamental")


def _accept_moved_settings(cls):
    import functools
    import warnings
    plain_init = cls.__init__

    @functools.wraps(plain_init)
    def __init__(self, *args, **kwargs):
        moved = {k: kwargs.pop(k) for k in _MOVED_TO_EXPERIMENTAL if k in kwargs}
        if moved:
            warnings.warn(f"DensePipelineConfig({', '.join(moved)}=...) moved to experimental={{...}}", DeprecationWarning, stacklevel=2)
            kwargs["experimental"] = {**moved, **dict(kwargs.get("experimental") or {})}
        plain_init(self, *args, **kwargs)

    cls.__init__ = __init__
    for name in _MOVED_TO_EXPERIMENTAL:
        def getter(self, _n=name):
            warnings.warn(f"DensePipelineConfig.{_n} moved to experimental[{_n!r}] (config.exp({_n!r}))", DeprecationWarning, stacklevel=2)
            return self.exp(_n)
        setattr(cls, name, property(getter))
    return cls


DensePipelineConfig = _accept_moved_settings(DensePipelineConfig)


@dataclasses.dataclass
class CameraRecord:
    uid: int
    image_path: str
    width: int
    height: int
    K: np.ndarray
    R: np.ndarray
    t: np.ndarray
    P: np.ndarray
    C: np.ndarray
    mask_path: Optional[str] = None

    def flat_pose(self) -> np.ndarray:
        """Row-major 4x4 world-to-camera matrix as a 16-vector (f64), the feature used for
        k-centres reference selection and nearest-neighbour lookup (upstream
        core/camera_models.py:23-28)."""
        pose = np.eye(4)
        pose[:3, :3] = self.R
        pose[:3, 3] = np.asarray(self.t).reshape(3)
        return pose.reshape(-1)

    @staticmethod
    def from_krt(uid: int, K, R, t, width: int, height: int, image_path: str = "",
                 mask_path: Optional[str] = None) -> "CameraRecord":
        """Build a record the way both upstream entry points do (densify.py:59-88,215-245):
        f32 ``K,R,t``; ``P = K @ [R|t]`` and ``C = -R^T t`` evaluated in f32."""
        K = np.asarray(K, np.float32).reshape(3, 3)
        R = np.asarray(R, np.float32).reshape(3, 3)
        t = np.asarray(t, np.float32).reshape(3, 1)
        P = K @ np.concatenate([R, t], axis=1)
        C = (-R.T @ t).reshape(3)
        return CameraRecord(uid=int(uid), image_path=image_path, width=int(width), height=int(height),
                            K=K, R=R, t=t, P=P, C=C, mask_path=mask_path)
