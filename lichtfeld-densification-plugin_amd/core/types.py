"""Boundary types of the dense-initialisation path.

``DensePipelineConfig`` keeps the 18 fields of the upstream dataclass with the same names, order and
defaults (upstream core/config.py:7-26) so that the GUI panel / CLI can construct it unchanged, and
adds MI355X-specific knobs *after* them (all defaulted, so positional construction still works).
``CameraRecord`` keeps upstream's per-camera record (core/camera_models.py:10-28): f32 intrinsics and
world-to-camera pose, the projection ``P = K [R|t]`` and the centre ``C = -R^T t``.
"""
from __future__ import annotations

import dataclasses
from typing import Optional

import numpy as np

TRIANGULATION_MODES = ("sampled", "dense")


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
    # ---- extensions of this implementation (not present upstream) --------------------------
    # "sampled": upstream behaviour - coverage sampling picks ~0.85*M+tiles cells per reference and
    #            only those are triangulated.  "dense": every grid cell upstream's sampler COULD draw (best certainty after
    #            floor and masks not <= 0: a masked-out cell never is) goes through the fused kernel.
    triangulation_mode: str = "sampled"
    # references whose RoMa outputs are kept resident and triangulated by ONE kernel launch
    refs_per_launch: int = 1
    # per-reference RNG stream (seed ^ uid) instead of upstream's single process-global stream;
    # forced on when references are sharded over several GPUs (results then do not depend on
    # the shard count).
    per_reference_rng: bool = False
    # where the coverage-sampling stage of the "sampled" mode runs: "device" (lfd_select_samples: the
    # whole per-reference path stays on the GPU) or "host" (core/sampling.py: the library calls
    # upstream makes, including torch's own f32 sum as the normaliser).
    selection_backend: str = "device"
    # selection_backend="device", one RNG stream (upstream's mode): normalise the sampling weights with upstream's OWN normaliser - torch's
    # CPU f32 `sum` of the weight map, computed on this host exactly as core/sampling.py:27-31 computes it - instead of the device's
    # correctly rounded exact sum.  torch's sum rounds differently from the exact one on most maps (by 1-2 ulp, depending on the
    # host's thread count and vector ISA), which moves a cumulative-sum boundary under a draw on up to a quarter of the maps; with
    # this on, the cells drawn are the ones upstream draws ON THE SAME MACHINE, bit for bit, at the price of one 1 MB read-back and a
    # host reduction per reference (~0.2 ms; the fused asynchronous call and the launch-ahead are then not used).  Off: everything
    # stays on the device (0.2 ms per reference, pipelined).  Sharded / per-reference-stream runs always use the exact sum: they do not
    # reproduce upstream's single stream anyway.
    upstream_normaliser: bool = True
    # dense mode only: blend colours with upstream's f64 arithmetic (bit-identical rgb) instead of f32 (within 2.5e-7)
    exact_colour: bool = False
    # hand upstream's own fundamental matrices (np.linalg.inv products, computed on the host exactly as upstream computes
    # them) to the kernels instead of the closed-form F the library derives from the camera table
    upstream_fundamental: bool = True
    # write the output PLY while the run proceeds: every completed reference's survivors are packed on the device
    # (lfd_pack_ply, 15 B per point across PCIe) and appended to ``output_path``; the vertex count in the header is patched at
    # the end.  Honoured when the output is a .ply and neither a point cap nor a voxel filter has to see the whole cloud first.
    stream_output: bool = False
    # resize / mask / black-out the decoded images on the GPU (lfd_prepare_image / lfd_prepare_mask: Pillow's BILINEAR and
    # NEAREST arithmetic, bit for bit) instead of with PIL on the host pack threads; decoding stays on the host
    device_image_prep: bool = False
    # compute every camera's backbone (DINOv3) features once per run and share them between the references that list the camera
    # (core/scheduler.py); upstream recomputes a neighbour's features for every reference
    share_features: bool = True
    # neighbours of a reference matched per RoMa-v2 forward (one batched pass through the model instead of one pass per pair, which is
    # what upstream's loop does, core/matcher.py:175-188).  1 = upstream's behaviour; batched GEMMs may round differently from single ones
    pairs_per_forward: int = 1
    # where the per-reference hot path runs.  "device": the HIP kernels (needs a GPU; raises HipBackendError without one - there is
    # no fallback).  "host": the CPU twin of the C-ABI (lfd_create_host: the host build of the kernels' per-cell source on the host
    # cores) with the host sampling stage - upstream's CPU-only configuration (densify.py:148-212 run without a GPU, BASELINE
    # config 1); chosen by the caller, never automatically.
    backend: str = "device"
    # how the survivors of a sharded run (torch.distributed, world > 1) reach the writer: "all_gather" leaves the whole cloud on
    # every rank (what BASELINE's north star names), "gather_to_root" sends every rank's records straight to their place in rank
    # 0's buffer (the other ranks return their own shard only)
    exchange: str = "all_gather"
    # sharded runs: the exchange happens in ROUNDS beside the compute (core/distributed.py::OverlappedExchange) - every
    # ``exchange_round`` local references (0: ``refs_per_launch``, at least 4) the finished references' records go into an asynchronous
    # collective while the next batch computes - instead of ONE exchange after the last reference.  The result is the same sequence.
    exchange_overlap: bool = True
    exchange_round: int = 0
    # what travels: "f32" the 28-byte rows the result holds (xyz, rgb, err as f32); "ply" the 15-byte PLY vertex records packed on the
    # device (positions exact, colours as the writer quantises them, no reprojection error: ``PipelineResult.rgb`` is then u8 / 255 and
    # ``err`` zero - the written file is the same bytes); "auto": "ply" when the output is a .ply and no voxel filter has to see f32
    # colours, else "f32".  Only used by the overlapped exchange of a sharded run.
    exchange_records: str = "f32"
    # sharded runs: the fraction of the reference list (its LAST references) computed by EVERY rank that receives the cloud instead of being
    # exchanged - recompute instead of communicate.  core/distributed.py::plan_replication says when it pays: for the bare hot path (one GPU
    # triangulates a reference faster than its survivors cross an xGMI link), never with a real matcher in the loop (default 0).  Needs the
    # overlapped exchange; ignored with stream_output.  The result is the same sequence.
    exchange_replicate: float = 0.0
    # sharded run + stream_output on ONE node: every rank writes its own byte ranges of the output file (core/distributed.py::SharedFilePlyStream) -
    # only the per-reference counts cross a link - instead of sending its records to rank 0 (ShardedPlyStream).  Needs a file system all ranks see.
    stream_shared_file: bool = False
    # dense mode: the kernel with UNORDERED retirement (lfd_triangulate_dense_segments: no look-back, ~6 % less kernel time); raster order is
    # restored from the tile table by lfd_order_segments (bit-identical result).  Opt-in.
    dense_tile_segments: bool = False

    def __post_init__(self) -> None:
        if self.triangulation_mode not in TRIANGULATION_MODES:
            raise ValueError(f"triangulation_mode must be one of {TRIANGULATION_MODES}, "
                             f"got {self.triangulation_mode!r}")
        if int(self.refs_per_launch) < 1:
            raise ValueError("refs_per_launch must be >= 1")
        if self.selection_backend not in ("device", "host"):
            raise ValueError("selection_backend must be 'device' or 'host'")
        if self.backend not in ("device", "host"):
            raise ValueError("backend must be 'device' or 'host'")
        if self.exchange not in ("all_gather", "gather_to_root"):
            raise ValueError("exchange must be 'all_gather' or 'gather_to_root'")
        if self.exchange_records not in ("f32", "ply", "auto"):
            raise ValueError("exchange_records must be 'f32', 'ply' or 'auto'")
        if int(self.exchange_round) < 0:
            raise ValueError("exchange_round must be >= 0")
        if not (0.0 <= float(self.exchange_replicate) <= 1.0):
            raise ValueError("exchange_replicate must be a fraction in [0, 1]")


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
