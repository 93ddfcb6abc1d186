#!/usr/bin/env python3
"""Entry points of the dense-initialisation pass (upstream densify.py:148-420).

``dense_init`` (COLMAP scene on disk / CLI), ``dense_init_from_lfs`` (camera nodes handed over by
LichtFeld Studio's GUI job) and ``build_argparser`` keep upstream's signatures, flags, return codes
and progress milestones; the body is host glue around ``core.pipeline.run_dense_pipeline`` whose
per-reference hot path runs on the GPU.
"""
from __future__ import annotations

import argparse
import os
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np

from .core.hostlog import log
from .core.image_io import find_image, image_dir, to_uint8_rgb
from .core.pipeline import PipelineCancelled, run_dense_pipeline
from .core.selection import nearest_neighbors, select_cameras_by_visibility, select_cameras_kcenters
from .core.types import CameraRecord, DensePipelineConfig, TRIANGULATION_MODES
from .core.writers import write_ply, write_points3D_bin


# ---- COLMAP -> CameraRecord (upstream core/geometry.py:10-50, densify.py:53-88) ----------------------
def K_from_camera(cam) -> np.ndarray:
    """3x3 f32 intrinsics from a pycolmap camera: (fx, fy, cx, cy) by model family; distortion
    parameters are ignored exactly as upstream ignores them."""
    model = str(cam.model.name).upper()
    p = np.asarray(cam.params, dtype=np.float32)
    # upstream's chain IN ITS ORDER (core/geometry.py:15-30): SIMPLE_RADIAL_FISHEYE is a SIMPLE_RADIAL before it is a FISHEYE
    # (found by tests/golden/check_oracle_fuzz.py: the two 4-parameter tests had been merged in front of the 3-parameter ones)
    if "PINHOLE" in model and "SIMPLE" not in model:
        fx, fy, cx, cy = p[0], p[1], p[2], p[3]
    elif "SIMPLE_PINHOLE" in model or "SIMPLE_RADIAL" in model or model == "RADIAL":
        fx = fy = p[0]
        cx, cy = p[1], p[2]
    elif "OPENCV" in model or "FISHEYE" in model:
        fx, fy, cx, cy = p[0], p[1], p[2], p[3]
    else:
        fx = fy = p[0]
        cx = p[1] if len(p) > 1 else cam.width / 2
        cy = p[2] if len(p) > 2 else cam.height / 2
    K = np.eye(3, dtype=np.float32)
    K[0, 0], K[1, 1], K[0, 2], K[1, 2] = fx, fy, cx, cy
    return K


def pose_world2cam(im) -> Tuple[np.ndarray, np.ndarray]:
    if hasattr(im, "cam_from_world"):
        cfw = im.cam_from_world
        cfw = cfw() if callable(cfw) else cfw
        R = np.asarray(cfw.rotation.matrix(), dtype=np.float32)
        t = np.asarray(cfw.translation, dtype=np.float32).reshape(3, 1)
    else:
        R = im.qvec.to_rotation_matrix()
        t = np.asarray(im.tvec, dtype=np.float32).reshape(3, 1)
    return R, t


def load_reconstruction(sparse_dir: str):
    """``pycolmap.Reconstruction`` like upstream (densify.py:54-56) when pycolmap is installed; otherwise the same three files read by
    core/colmap_io.py (cameras / images / points3D, .bin or .txt), which offers the attributes upstream touches."""
    try:
        import pycolmap
    except ImportError:
        from .core.colmap_io import Reconstruction
        rec = Reconstruction(sparse_dir)
        log.info(f"pycolmap is not installed: read {len(rec.images)} images / {len(rec.cameras)} cameras from {sparse_dir} with the built-in reader")
        return rec, rec.cameras, rec.images
    rec = pycolmap.Reconstruction(sparse_dir)
    return rec, rec.cameras, rec.images


def camera_records_from_colmap(cams: Dict, imgs: Dict, images_dir: str) -> Tuple[List[CameraRecord], List[int]]:
    records: List[CameraRecord] = []
    img_ids = sorted(imgs.keys())
    for iid in img_ids:
        im = imgs[iid]
        cam = cams[im.camera_id]
        R, t = pose_world2cam(im)
        rec = CameraRecord.from_krt(iid, K_from_camera(cam), R, t, cam.width, cam.height,
                                    image_path=find_image(images_dir, im.name))
        records.append(rec)
    return records, img_ids


def extract_cameras_from_lfs(camera_nodes) -> List[CameraRecord]:
    """Scene camera nodes -> records; the principal point is assumed at the image centre
    (upstream densify.py:215-245)."""
    records: List[CameraRecord] = []
    for node in camera_nodes:
        if not getattr(node, "has_camera", False):
            continue
        w, h = node.camera_width, node.camera_height
        K = np.array([[node.camera_focal_x, 0.0, w / 2.0], [0.0, node.camera_focal_y, h / 2.0], [0.0, 0.0, 1.0]],
                     dtype=np.float32)
        rec = CameraRecord.from_krt(node.camera_uid, K, node.camera_R, node.camera_T, w, h, image_path=node.image_path,
                                    mask_path=(node.mask_path if getattr(node, "has_mask", False) else None))
        records.append(rec)
    return records


# ---- post-processing ------------------------------------------------------------------------------
def _flat_pose_stack(records: List[CameraRecord]) -> np.ndarray:
    return np.stack([c.flat_pose() for c in records], axis=0)


def _num_refs(fraction_or_count: float, n: int) -> int:
    return int(round(fraction_or_count * n)) if fraction_or_count <= 1.0 else int(fraction_or_count)


def _effective_neighbor_count(requested: int, camera_count: int) -> int:
    if camera_count <= 1:
        return 0
    return max(1, min(int(requested), camera_count - 1))


def _apply_point_cap(xyz, rgb, err, max_points: int, seed: int):
    if max_points > 0 and xyz.shape[0] > max_points:
        keep = np.random.default_rng(seed).choice(xyz.shape[0], size=max_points, replace=False)
        return xyz[keep], rgb[keep], err[keep]
    return xyz, rgb, err


def _finish_on_device(result, path: str, max_points: int, seed: int, clock=None) -> Optional[int]:
    """Point cap + output file for a result whose points are still on the GPU, without ever bringing the f32 cloud to the host: the same subset
    ``_apply_point_cap`` picks (the same generator, the same call), applied to the device tensors, then the file payload packed on the device
    (``_write_output``).  Returns the number of points written, or None when the result is not on a GPU (the host path applies)."""
    pts = result.device_points
    if pts is None or not pts[0].is_cuda:
        return None
    pts = _cap_device_points(pts, int(pts[0].shape[0]), max_points, seed)
    n = int(pts[0].shape[0])
    _write_output(path, np.empty((n, 0), np.float32), None, None, pts, clock=clock)
    return n


def _voxel_downsample(xyz: np.ndarray, rgb: np.ndarray, voxel_size: float) -> Tuple[np.ndarray, np.ndarray]:
    """One averaged point (and colour) per occupied voxel.  Uses Open3D when installed (upstream
    densify.py:29-50); otherwise an equivalent NumPy voxel-grid average (voxel order then follows the
    sorted voxel index instead of Open3D's hash order)."""
    col = rgb[:, :3].astype(np.float64) / (255.0 if rgb.size and rgb.max() > 1.0 else 1.0)
    try:
        import open3d as o3d
        pcd = o3d.geometry.PointCloud()
        pcd.points = o3d.utility.Vector3dVector(xyz.astype(np.float64))
        pcd.colors = o3d.utility.Vector3dVector(col)
        down = pcd.voxel_down_sample(voxel_size=float(voxel_size))
        return np.asarray(down.points, dtype=np.float32), np.asarray(down.colors, dtype=np.float32)
    except ImportError:
        pts = xyz.astype(np.float64)
        origin = pts.min(axis=0) - 0.5 * voxel_size
        key = np.floor((pts - origin) / float(voxel_size)).astype(np.int64)
        _, inv, cnt = np.unique(key, axis=0, return_inverse=True, return_counts=True)
        inv = inv.reshape(-1)
        p = np.zeros((cnt.size, 3))
        c = np.zeros((cnt.size, 3))
        np.add.at(p, inv, pts)
        np.add.at(c, inv, col)
        return (p / cnt[:, None]).astype(np.float32), (c / cnt[:, None]).astype(np.float32)


def _is_writer_rank() -> bool:
    """True unless this process is a non-zero rank of an initialised torch.distributed job."""
    try:
        import torch.distributed as dist
        return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0
    except Exception:
        return True


def _write_output(path: str, xyz, rgb, err, device_points=None, clock=None) -> None:
    """``.ply`` -> upstream's PLY, anything else -> upstream's points3D.bin (densify.py:129-135).  When
    the points are still on the GPU the records are quantised and packed there (lfd_pack_*) and only
    the final bytes are copied to the host; the files are byte-identical either way."""
    if not _is_writer_rank():        # sharded run: every rank holds the gathered cloud, rank 0 alone writes the file
        return
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    as_ply = path.lower().endswith(".ply")
    if device_points is not None and device_points[0].is_cuda and int(device_points[0].shape[0]) == int(xyz.shape[0]):
        from .core import hip_backend as hb
        from .core.writers import write_ply_packed, write_points3D_bin_packed
        from .core.stages import NULL_CLOCK
        clock = clock if clock is not None else NULL_CLOCK
        dens = hb.HipDensifier(device_points[0].device)
        try:
            n = int(xyz.shape[0])
            with clock.stage("d2h"):            # the file payload - 15 / 43 bytes per point - is what crosses PCIe
                packed = dens.pack_ply(device_points[0], device_points[1]) if as_ply else dens.pack_points3d(*device_points)
                body = packed.cpu().numpy().tobytes()
            with clock.stage("write", sync=False):
                (write_ply_packed if as_ply else write_points3D_bin_packed)(path, n, body)
        finally:
            dens.close()
        return
    rgb8 = to_uint8_rgb(rgb)
    if as_ply:
        write_ply(path, xyz, rgb8)
    else:
        write_points3D_bin(path, xyz, rgb8, err)


def _cap_device_points(device_points, n_before: int, max_points: int, seed: int):
    """The same subset _apply_point_cap picks, applied to the GPU copy."""
    if device_points is None or not (max_points > 0 and n_before > max_points):
        return device_points
    import torch
    keep = np.random.default_rng(seed).choice(n_before, size=max_points, replace=False)
    idx = torch.from_numpy(keep).to(device_points[0].device)
    return tuple(t[idx] for t in device_points)


def _was_cancelled(cb) -> bool:
    if cb is None:
        return False
    try:
        return bool(cb())
    except Exception as exc:
        log.warn(f"Cancellation callback failed: {exc}")
        return False


# ---- entry points ------------------------------------------------------------------------------------
def plan_scene(args):
    """What upstream's CLI derives from the scene before the pipeline starts (densify.py:154-165 there): the camera records, the reference
    views (greedy visibility cover of the sparse points, k-centres on the poses if that fails) and the neighbour table.
    Returns ``(records, refs_local, nn_table, sparse_dir)``."""
    scene_root = os.path.abspath(args.scene_root)
    sparse_dir = os.path.join(scene_root, "sparse", "0")
    images_dir = image_dir(scene_root, args.images_subdir)
    rec, cams, imgs = load_reconstruction(sparse_dir)
    records, img_ids = camera_records_from_colmap(cams, imgs, images_dir)
    flat = _flat_pose_stack(records)
    want = max(1, _num_refs(args.num_refs, len(img_ids)))
    try:
        by_id = {iid: i for i, iid in enumerate(img_ids)}
        refs_local = [by_id[r] for r in select_cameras_by_visibility(rec, want) if r in by_id]
    except Exception as exc:
        log.warn(f"Visibility-based selection failed: {exc}")
        refs_local = select_cameras_kcenters(flat, want)
    nn_table = nearest_neighbors(flat, max(1, args.nns_per_ref))
    return records, refs_local, nn_table, sparse_dir


def dense_init(args, progress_callback: Optional[Callable[[float, str], None]] = None, debug_state=None,
               cancel_requested: Optional[Callable[[], bool]] = None, **pipeline_kwargs) -> int:
    """CLI / COLMAP entry point.  Returns 0 on success, 2 when cancelled; raises on error.  ``pipeline_kwargs``: the injection points of
    ``run_dense_pipeline`` (a warm matcher, a stage clock), as ``dense_init_from_lfs`` takes them."""
    records, refs_local, nn_table, sparse_dir = plan_scene(args)
    config = DensePipelineConfig(
        output_path=os.path.join(sparse_dir, args.out_name), roma_setting=args.roma_setting, num_refs=args.num_refs,
        nns_per_ref=args.nns_per_ref, matches_per_ref=args.matches_per_ref, certainty_thresh=args.certainty_thresh,
        reproj_thresh=args.reproj_thresh, sampson_thresh=args.sampson_thresh, min_parallax_deg=args.min_parallax_deg,
        max_points=args.max_points, no_filter=args.no_filter, seed=args.seed, viz_interval=0,
        prefetch_packages=args.prefetch_packages, pack_workers=args.pack_workers,
        triangulation_mode=getattr(args, "triangulation_mode", "sampled"),
        refs_per_launch=getattr(args, "refs_per_launch", 0), backend=getattr(args, "backend", "device"),
        stream_output=bool(getattr(args, "stream_output", False)), device_image_prep=bool(getattr(args, "device_image_prep", False)))
    try:
        result = run_dense_pipeline(records, refs_local, nn_table, config, progress_callback=progress_callback,
                                    on_sequential_viz=None, debug_state=debug_state, cancel_requested=cancel_requested, **pipeline_kwargs)
    except PipelineCancelled:
        if progress_callback:
            progress_callback(0.0, "Cancelled")
        return 2
    if _was_cancelled(cancel_requested):
        if progress_callback:
            progress_callback(0.0, "Cancelled")
        return 2
    if progress_callback:
        progress_callback(95.0, "Writing output...")
    if result.streamed_path == config.output_path:      # config.stream_output: the file is already complete (and no cap applies to it)
        n_points = result.n_points
    else:
        n_points = _finish_on_device(result, config.output_path, args.max_points, args.seed, clock=pipeline_kwargs.get("stage_clock"))
        if n_points is None:
            xyz, rgb, err = _apply_point_cap(result.xyz, result.rgb, result.err, args.max_points, args.seed)
            _write_output(config.output_path, xyz, rgb, err, None)
            n_points = int(xyz.shape[0])
    log.info(f"Dense reconstruction finished: {n_points:,} points -> {config.output_path}")
    if progress_callback:
        progress_callback(100.0, f"Done! {n_points:,} points")
    return 0


def dense_init_from_lfs(camera_nodes, config: DensePipelineConfig,
                        progress_callback: Optional[Callable[[float, str], None]] = None,
                        on_sequential_viz: Optional[Callable[[str], None]] = None, debug_state=None,
                        cancel_requested: Optional[Callable[[], bool]] = None, **pipeline_kwargs
                        ) -> Tuple[int, Optional[str]]:
    """GUI entry point.  Returns ``(0, output_path)``, ``(1, message)`` or ``(2, "Cancelled")``."""
    if progress_callback:
        progress_callback(2.0, "Extracting camera data from scene...")
    records = extract_cameras_from_lfs(camera_nodes)
    if not config.use_masks:
        for r in records:
            r.mask_path = None
    if len(records) < 2:
        return 1, "Need at least 2 cameras for dense initialization"
    flat = _flat_pose_stack(records)
    refs_local = select_cameras_kcenters(flat, max(1, _num_refs(config.num_refs, len(records))))
    nns = _effective_neighbor_count(config.nns_per_ref, len(records))
    if nns < 1:
        return 1, "Need at least 2 cameras for dense initialization"
    if nns != int(config.nns_per_ref):
        log.info(f"Clamping neighbors per reference from {config.nns_per_ref} to {nns} for {len(records)} ROI cameras")
    nn_table = nearest_neighbors(flat, nns)
    log.info(f"Prepared {len(records)} cameras (refs={len(refs_local)})")
    try:
        result = run_dense_pipeline(records, refs_local, nn_table, config, progress_callback=progress_callback,
                                    on_sequential_viz=on_sequential_viz, debug_state=debug_state,
                                    cancel_requested=cancel_requested, **pipeline_kwargs)
    except PipelineCancelled:
        return 2, "Cancelled"
    except RuntimeError as exc:
        return 1, str(exc)
    if _was_cancelled(cancel_requested):
        return 2, "Cancelled"
    if result.streamed_path == config.output_path:      # config.stream_output: the PLY is complete; neither a cap nor a voxel filter applies to it
        if progress_callback:
            progress_callback(95.0, "Writing output PLY...")
        log.info(f"Dense point cloud saved to {config.output_path} ({result.n_points:,} points)")
        if progress_callback:
            progress_callback(100.0, f"Done! {result.n_points:,} points")
        return 0, config.output_path
    if config.voxel_size <= 0.0 and config.output_path.lower().endswith(".ply") and result.device_points is not None and result.device_points[0].is_cuda:
        # nothing has to see the cloud on the host: the cap is applied and the file payload packed where the points are
        if progress_callback:
            progress_callback(95.0, "Writing output PLY...")
        n_written = _finish_on_device(result, config.output_path, config.max_points, config.seed, clock=pipeline_kwargs.get("stage_clock"))
        log.info(f"Dense point cloud saved to {config.output_path} ({n_written:,} points)")
        if progress_callback:
            progress_callback(100.0, f"Done! {n_written:,} points")
        return 0, config.output_path
    xyz, rgb, err = _apply_point_cap(result.xyz, result.rgb, result.err, config.max_points, config.seed)
    dev_pts = _cap_device_points(result.device_points, result.xyz.shape[0], config.max_points, config.seed)
    if config.voxel_size > 0.0:
        if progress_callback:
            progress_callback(93.0, "Applying distance filter...")
        xyz, rgb = _voxel_downsample(xyz, rgb, config.voxel_size)
        dev_pts = None                                   # the voxel average lives on the host
        log.info(f"Distance filter ({config.voxel_size:.4f}): {xyz.shape[0]:,} points remaining")
    if progress_callback:
        progress_callback(95.0, "Writing output PLY...")
    out_path = config.output_path if config.output_path.lower().endswith(".ply") else config.output_path + ".ply"
    if out_path != config.output_path:                  # upstream always writes a PLY here, whatever the name
        d = os.path.dirname(config.output_path)
        if d:
            os.makedirs(d, exist_ok=True)
        if _is_writer_rank():
            write_ply(config.output_path, xyz, to_uint8_rgb(rgb))
    else:
        _write_output(config.output_path, xyz, rgb, err, dev_pts)
    log.info(f"Dense point cloud saved to {config.output_path} ({xyz.shape[0]:,} points)")
    if progress_callback:
        progress_callback(100.0, f"Done! {xyz.shape[0]:,} points")
    return 0, config.output_path


def build_argparser() -> argparse.ArgumentParser:
    """Upstream's CLI flags and defaults (densify.py:318-415) plus the two launch-shape extensions."""
    ap = argparse.ArgumentParser("Dense COLMAP initializer (RoMa v2 matching + fused HIP filter/triangulate on MI355X)")
    ap.add_argument("--scene_root", type=str, required=True, help="Path containing images*/ and sparse/0/")
    ap.add_argument("--images_subdir", type=str, default="images_2", help="Which images dir to read under scene_root")
    ap.add_argument("--out_name", type=str, default="points3D_dense.ply", help="Output filename under sparse/0/")
    ap.add_argument("--roma_setting", type=str, default="fast", choices=["precise", "high", "base", "fast", "turbo"],
                    help="RoMaV2 quality/speed setting")
    ap.add_argument("--roma_model", type=str, default="outdoor", choices=["outdoor", "indoor"],
                    help="Legacy flag for compatibility (RoMaV2 is unified)")
    ap.add_argument("--num_refs", type=float, default=0.75, help="Fraction (<=1) or count (>1) of frames to use as references")
    ap.add_argument("--nns_per_ref", type=int, default=4, help="Nearest neighbors per reference (3-5 is robust)")
    ap.add_argument("--matches_per_ref", type=int, default=12000, help="Samples per ref after aggregation")
    ap.add_argument("--certainty_thresh", type=float, default=0.20, help="Min certainty floor before selection")
    ap.add_argument("--reproj_thresh", type=float, default=1.5, help="Max reprojection error (px)")
    ap.add_argument("--sampson_thresh", type=float, default=5.0, help="Max Sampson error (px^2) pre-triangulation (<=0 disables)")
    ap.add_argument("--min_parallax_deg", type=float, default=0.5, help="Min parallax angle in degrees")
    ap.add_argument("--no_filter", action="store_true", help="Disable geometric filtering (debug only)")
    ap.add_argument("--max_points", type=int, default=0, help="Optional cap on total points (0 = unlimited)")
    ap.add_argument("--prefetch_packages", type=int, default=8, help="Reference packages prefetched ahead of the GPU")
    ap.add_argument("--pack_workers", type=int, default=4, help="Threads used to load/resize images")
    ap.add_argument("--seed", type=int, default=0, help="Random seed")
    ap.add_argument("--triangulation_mode", type=str, default="sampled", choices=list(TRIANGULATION_MODES),
                    help="sampled = upstream behaviour; dense = every grid cell through the fused kernel")
    ap.add_argument("--refs_per_launch", type=int, default=0, help="references per kernel launch (dense mode) or per fused call (sampled mode, device backend: same results, same RNG stream); 0 = automatic (16 where possible, 1 when intermediate previews are written)")
    ap.add_argument("--backend", type=str, default="device", choices=["device", "host"],
                    help="device = the HIP kernels (needs a GPU); host = the CPU twin of the C-ABI + the host sampling stage "
                         "(upstream's CPU-only configuration); never chosen automatically")
    ap.add_argument("--stream_output", action="store_true",
                    help="write the PLY while the run proceeds (15-byte records packed on the device; needs a .ply --out_name and no --max_points)")
    ap.add_argument("--device_image_prep", action="store_true", help="resize / mask the decoded images on the GPU (Pillow's arithmetic, bit for bit)")
    ap.add_argument("--keep_threads", action="store_true",
                    help="leave torch's intra-op thread count alone (by default it is lowered to the container's CPU quota; the count decides the last "
                         "bits of upstream's sampling normaliser, so a run compared bit for bit with upstream keeps upstream's setting)")
    return ap


def main(argv=None) -> int:
    """The command line (upstream densify.py:418-420).  As its own process it fits torch's intra-op threads to the container's CPU quota first
    (core/hostenv.py: a pool sized by the CPUs the container SEES gets the whole process throttled); inside LichtFeld Studio - ``dense_init`` /
    ``dense_init_from_lfs`` called by the plugin - process-wide settings are the host application's and nothing is touched."""
    from .core import hostenv
    args = build_argparser().parse_args(argv)
    if not args.keep_threads:
        import torch
        before = int(torch.get_num_threads())
        after = hostenv.fit_threads_to_quota(log=log.info)
        if after != before:
            log.info("note: upstream's sampling normaliser is a torch f32 sum whose last bits depend on the thread count - for draws bit-identical "
                     "to an upstream run on this machine pass --keep_threads or set OMP_NUM_THREADS to upstream's count")
    return dense_init(args)


if __name__ == "__main__":
    raise SystemExit(main())
