#!/usr/bin/env python3
"""bench.py's `pipeline` leg: SURVEY 8d's points/s definition (ii) - the hot path INCLUDING image loading, preparation, selection, compaction,
the survivors' way to the host and the written PLY - measured through ``densify.dense_init`` end to end on a generated on-disk scene.

What it replaces upstream: the driver loop core/pipeline.py:783-928 with core/threaded_dataloader.py:42-241 (loading), core/image_utils.py:85-91
(resize / mask), core/pipeline.py:602-780 (per reference), core/writers.py:29-46 (the PLY).  The matcher is a stand-in (RoMa-v2's weights are not
on the box): ``synthetic.SyntheticMatcher`` hands out the analytic warp + certainty fields of the scene, resident on the GPU, from a table made
before the run - zero latency - or after a stated sleep per pair.

    python bench_pipeline.py [--cams 185] [--latency-ms 20]          # prints the leg's JSON object alone
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PCIE_GEN5_X16_GBPS = 64.0        # nominal, one direction (32 GT/s x 16 lanes, 128b/130b)


def _clear_image_caches() -> None:
    """Every run of the leg decodes its images again: the loaders' caches live in the process, a second run would find them warm."""
    from lichtfeld_densification_plugin_amd.core import image_io
    for fn in (image_io.load_rgb_u8, image_io.load_mask01, image_io.decode_rgb_u8, image_io.decode_mask_l):
        fn.cache_clear()


def pinned_d2h_GBps(dev, n_bytes: int = 1 << 28, reps: int = 5) -> float:
    """What a plain device -> pinned host copy reaches on this box: the ceiling the streamed records are priced against."""
    src = torch.empty((n_bytes,), dtype=torch.uint8, device=dev)
    dst = torch.empty((n_bytes,), dtype=torch.uint8).pin_memory()
    dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        dst.copy_(src, non_blocking=True)
    e1.record()
    torch.cuda.synchronize(dev)
    return n_bytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def _ply_vertices(path: str) -> int:
    with open(path, "rb") as fh:
        head = fh.read(512).split(b"end_header\n", 1)[0].decode("ascii", "replace")
    return int([ln for ln in head.split("\n") if ln.startswith("element vertex")][0].split()[-1])


def run_once(scene_root: str, matcher, *, mode: str, device_prep: bool, backend: str = "device", refs_per_launch: int = 16, sync_stages: bool = False,
             roma_setting: str = "fast", num_refs: float = 0.8, nns: int = 3, matches: int = 10000, pack_workers: int = 4, sampled_group: int = 0) -> dict:
    """One ``dense_init`` on the scene, down to the written PLY.  GUI defaults (reprojection 0.8 px, 3 neighbours, 0.8 of the cameras as references).
    dense mode streams its output (``stream_output``: 15-byte records written by the kernel, copied out beside the next launch); sampled mode takes
    upstream's own flow - result arrays, then the writer (records packed on the device)."""
    from lichtfeld_densification_plugin_amd import densify
    from lichtfeld_densification_plugin_amd.core.stages import StageClock
    tag = f"{mode}_{'dev' if device_prep else 'host'}prep"
    argv = ["--scene_root", scene_root, "--images_subdir", "images_4", "--roma_setting", roma_setting, "--num_refs", str(num_refs), "--nns_per_ref", str(nns),
            "--matches_per_ref", str(matches), "--reproj_thresh", "0.8", "--out_name", f"bench_{tag}.ply", "--triangulation_mode", mode,
            "--pack_workers", str(pack_workers), "--backend", backend]
    if mode == "dense":
        argv += ["--refs_per_launch", str(refs_per_launch), "--stream_output"]
    elif sampled_group > 0:      # (0: the CLI's default - automatic: 16 references per fused call on upstream's one RNG stream, lfd_triangulate_sampled_chain)
        argv += ["--refs_per_launch", str(sampled_group)]
    if device_prep:
        argv += ["--device_image_prep"]
    args = densify.build_argparser().parse_args(argv)
    out_path = os.path.join(scene_root, "sparse", "0", f"bench_{tag}.ply")
    if os.path.exists(out_path):
        os.remove(out_path)
    _clear_image_caches()
    on_gpu = backend == "device"
    clock = StageClock(sync=(torch.cuda.synchronize if (sync_stages and on_gpu) else None))
    matcher.calls = matcher.pairs = 0
    matcher.seconds = 0.0
    if on_gpu:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    rc = densify.dense_init(args, matcher=matcher, stage_clock=clock)
    if on_gpu:
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rc != 0:
        raise RuntimeError(f"dense_init returned {rc}")
    n = _ply_vertices(out_path)
    rep = clock.report()
    stages = {k: v["seconds"] for k, v in rep.items() if isinstance(v, dict)}
    res = {"seconds": dt, "references": matcher.calls, "pairs": matcher.pairs, "points": n, "refs_per_s": matcher.calls / dt, "pairs_per_s": matcher.pairs / dt,
           "points_per_s": n / dt, "matcher_seconds": matcher.seconds, "file_bytes": os.path.getsize(out_path), "stage_seconds": stages}
    if "d2h_bytes" in rep and stages.get("d2h"):
        res["d2h_bytes"] = rep["d2h_bytes"]
        res["d2h_GBps"] = rep["d2h_bytes"] / stages["d2h"] / 1e9
    os.remove(out_path)
    return res


class _SceneNode:
    """What ``densify.extract_cameras_from_lfs`` reads of a LichtFeld Studio camera node (upstream densify.py:215-245): the GUI entry point's input."""

    def __init__(self, rec):
        self.has_camera, self.camera_uid = True, int(rec.uid)
        self.camera_width, self.camera_height = int(rec.width), int(rec.height)
        self.camera_focal_x, self.camera_focal_y = float(rec.K[0, 0]), float(rec.K[1, 1])
        self.camera_R, self.camera_T = np.asarray(rec.R, np.float32), np.asarray(rec.t, np.float32).reshape(3)
        self.image_path, self.mask_path, self.has_mask = rec.image_path, rec.mask_path, rec.mask_path is not None


def run_gui_once(nodes, matcher, out_dir: str, *, viz_interval: int, device_prep: bool, pack_workers: int, roma_setting: str = "fast", num_refs: float = 0.8,
                 nns: int = 3, matches: int = 10000) -> dict:
    """One ``dense_init_from_lfs`` - the entry point LichtFeld Studio's panel calls (upstream densify.py:248-315, panels/densification.py:278-285) - with
    the GUI's defaults: intermediate previews every ``viz_interval`` references that produced points (a cumulative PLY each, handed to
    ``on_sequential_viz``; the panel loads it into the scene - here its size is read), one reference per fused call while somebody watches
    (``DensePipelineConfig.launch_group``), progress callbacks, the final PLY.  ``viz_interval = 0``: the same run without previews."""
    import lichtfeld_densification_plugin_amd as lfd
    from lichtfeld_densification_plugin_amd import densify
    from lichtfeld_densification_plugin_amd.core.stages import StageClock
    out_path = os.path.join(out_dir, "gui_dense.ply")
    cfg = lfd.DensePipelineConfig(output_path=out_path, roma_setting=roma_setting, num_refs=num_refs, nns_per_ref=nns, matches_per_ref=matches,
                                  viz_interval=int(viz_interval), pack_workers=int(pack_workers), device_image_prep=bool(device_prep))
    _clear_image_caches()
    clock = StageClock()
    seen = {"previews": 0, "preview_bytes": 0, "progress": 0}

    def on_viz(path):
        seen["previews"] += 1
        seen["preview_bytes"] += os.path.getsize(path)
        os.remove(path)                 # (the panel replaces its preview node by the new file; the old one is of no further use)

    def on_progress(_pct, _msg):
        seen["progress"] += 1
    matcher.calls = matcher.pairs = 0
    matcher.seconds = 0.0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    code, info = densify.dense_init_from_lfs(nodes, cfg, progress_callback=on_progress, on_sequential_viz=(on_viz if viz_interval > 0 else None),
                                             matcher=matcher, stage_clock=clock)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if code != 0:
        raise RuntimeError(f"dense_init_from_lfs returned {code}: {info}")
    n = _ply_vertices(out_path)
    rep = clock.report()
    res = {"seconds": dt, "references": matcher.calls, "pairs": matcher.pairs, "points": n, "refs_per_s": matcher.calls / dt, "pairs_per_s": matcher.pairs / dt,
           "points_per_s": n / dt, "previews": seen["previews"], "preview_bytes": seen["preview_bytes"], "progress_callbacks": seen["progress"],
           "file_bytes": os.path.getsize(out_path), "stage_seconds": {k: v["seconds"] for k, v in rep.items() if isinstance(v, dict)}}
    os.remove(out_path)
    return res


def gui_leg(dev, records, *, roma_setting: str, num_refs: float, nns: int, pack_workers: int, out_dir: str) -> dict:
    """The GUI entry point on the scene's cameras as scene nodes: with the panel's previews (every third reference) and without."""
    from lichtfeld_densification_plugin_amd import densify, synthetic
    from lichtfeld_densification_plugin_amd.core.selection import nearest_neighbors, select_cameras_kcenters
    nodes = [_SceneNode(r) for r in records]
    recs = densify.extract_cameras_from_lfs(nodes)              # (principal point at the image centre: the GUI path's assumption)
    flat = np.stack([c.flat_pose() for c in recs], axis=0)
    refs = select_cameras_kcenters(flat, max(1, int(round(num_refs * len(recs))) if num_refs <= 1.0 else int(num_refs)))
    nn_table = nearest_neighbors(flat, max(1, min(int(nns), len(recs) - 1)))
    matcher = synthetic.SyntheticMatcher(recs, setting=roma_setting, device=dev, noise_px=0.5, outlier_frac=0.05, channels=2, seed=0)
    matcher.precompute(refs, nn_table, nns)
    kw = dict(device_prep=True, pack_workers=pack_workers, roma_setting=roma_setting, num_refs=num_refs, nns=nns)
    run_gui_once(nodes, matcher, out_dir, viz_interval=0, **kw)                      # untimed: the new matcher's table is touched once
    out = {"what": "densify.dense_init_from_lfs (the panel's entry point: k-centres references, principal point at the image centre) on the same cameras as scene "
                   "nodes, sampled mode, device image preparation; `previews_every_3`: the GUI's default viz_interval = 3 - a cumulative PLY after every third "
                   "reference, handed to on_sequential_viz (its size is read, the file removed), one reference per fused call while previews are watched",
           "previews_every_3": run_gui_once(nodes, matcher, out_dir, viz_interval=3, **kw),
           "no_previews": run_gui_once(nodes, matcher, out_dir, viz_interval=0, **kw)}
    return out


def pipeline_leg(dev, *, n_cams: int = 185, latency_ms: float = 20.0, scene_root: str = None, backend: str = "device", roma_setting: str = "fast",
                 width: int = 1297, height: int = 840, refs_per_launch: int = 16, runs=("zero", "latency", "stages", "gui"), num_refs: float = 0.8,
                 nns: int = 3) -> dict:
    """The `pipeline` object of the bench line: {"scene", "sampled": {...}, "dense": {...}, "pcie"}."""
    from lichtfeld_densification_plugin_amd import densify, synthetic
    t_setup = time.perf_counter()
    own_tmp = None
    if scene_root is None:
        own_tmp = tempfile.TemporaryDirectory(prefix="lfd_bench_scene_")
        scene_root = own_tmp.name
    synthetic.write_colmap_scene(scene_root, n_cams=n_cams, width=width, height=height, images_subdir="images_4", fmt="jpg", seed=0)
    plan_args = densify.build_argparser().parse_args(["--scene_root", scene_root, "--images_subdir", "images_4", "--num_refs", str(num_refs), "--nns_per_ref", str(nns)])
    records, refs_local, nn_table, _sparse = densify.plan_scene(plan_args)
    on_gpu = backend == "device"
    matcher = synthetic.SyntheticMatcher(records, setting=roma_setting, device=dev if on_gpu else "cpu", noise_px=0.5, outlier_frac=0.05, channels=2, seed=0)
    matcher.precompute(refs_local, nn_table, nns)
    setup_s = time.perf_counter() - t_setup
    kw = dict(backend=backend, roma_setting=roma_setting, refs_per_launch=refs_per_launch, num_refs=num_refs, nns=nns)
    leg = {"scene": {"cameras": n_cams, "image_size": [width, height], "image_format": "jpeg q90", "references": len(refs_local), "neighbours": nns,
                     "pairs": sum(len(k[1]) for k in matcher.table), "grid": [matcher.H, matcher.W], "setup_seconds": round(setup_s, 2),
                     "what": "synthetic garden-like COLMAP scene on disk (sparse/0 + images_4), densify.dense_init with GUI defaults "
                             f"({num_refs:g} of the cameras as references x {nns} neighbours, M = 10000, reprojection 0.8 px), matcher = analytic fields from a table on the GPU"},
           "note": "seconds = wall time of densify.dense_init (COLMAP read, reference / neighbour selection, image decode + preparation, matcher stand-in, "
                   "hot path, survivors to the host, PLY written and closed); image caches cleared before every run; `stage_seconds` of the plain runs are "
                   "host wall time per stage (launches are asynchronous: device time shows where the host next waits), `stages` is a separate run that "
                   "drains the device after every stage (a split, not a throughput); decode / prepare on the pack threads and write on the writer thread "
                   "overlap the loop, so the stages do not add up to the wall time"}
    if on_gpu:
        run_once(scene_root, matcher, mode="sampled", device_prep=True, **kw)      # untimed: module load, pinned pools, page cache of the scene's files
    for mode in ("sampled", "dense"):
        m = {}
        if "zero" in runs:
            m["host_prep"] = run_once(scene_root, matcher, mode=mode, device_prep=False, **kw)
            if on_gpu:
                m["device_prep"] = run_once(scene_root, matcher, mode=mode, device_prep=True, **kw)
        if "zero" in runs and on_gpu:
            # the zero-latency runs are bound by the JPEG decode on upstream's default of 4 pack threads: the same run with one thread per core
            # the container may use (the decode is the only stage that scales with it)
            from lichtfeld_densification_plugin_amd.core import hostenv
            cores = max(4, hostenv.usable_cores())
            if cores > 4:
                m[f"device_prep_{cores}_pack_workers"] = run_once(scene_root, matcher, mode=mode, device_prep=True, pack_workers=cores, **kw)
            if mode == "sampled":
                # the same run one reference per fused call (what a run with intermediate previews uses): the same cloud
                r = run_once(scene_root, matcher, mode=mode, device_prep=True, pack_workers=cores, sampled_group=1, **kw)
                m[f"device_prep_{cores}_pack_workers_1_ref_per_call"] = r
                same = m.get(f"device_prep_{cores}_pack_workers", m["device_prep"])
                if r["points"] != same["points"]:
                    raise RuntimeError(f"the one-reference-per-call run wrote {r['points']} points, the grouped (default) run {same['points']}")
        if "latency" in runs and latency_ms > 0:
            matcher.latency = latency_ms * 1e-3
            m[f"device_prep_matcher_{latency_ms:g}ms_per_pair" if on_gpu else f"host_prep_matcher_{latency_ms:g}ms_per_pair"] = \
                run_once(scene_root, matcher, mode=mode, device_prep=on_gpu, **kw)
            matcher.latency = 0.0
        if "stages" in runs and on_gpu:
            st = run_once(scene_root, matcher, mode=mode, device_prep=True, sync_stages=True, **kw)
            m["stages"] = {"seconds_per_stage": st["stage_seconds"], "run_seconds": st["seconds"],
                           "ms_per_reference": {k: round(v / max(1, st["references"]) * 1e3, 4) for k, v in st["stage_seconds"].items()}}
        leg[mode] = m
    if on_gpu and "gui" in runs:
        from lichtfeld_densification_plugin_amd.core import hostenv
        leg["gui"] = gui_leg(dev, records, roma_setting=roma_setting, num_refs=num_refs, nns=nns, pack_workers=max(4, hostenv.usable_cores()), out_dir=scene_root)
    if on_gpu:
        ceiling = pinned_d2h_GBps(dev)
        best = max((r.get("d2h_GBps", 0.0) for r in leg["dense"].values() if isinstance(r, dict)), default=0.0)
        leg["pcie"] = {"d2h_pinned_copy_GBps": ceiling, "nominal_gen5_x16_GBps": PCIE_GEN5_X16_GBPS, "dense_records_d2h_GBps": best or None,
                       "frac_of_measured_copy": (best / ceiling) if best else None, "frac_of_nominal": (best / PCIE_GEN5_X16_GBPS) if best else None,
                       "what": "dense mode's 15-byte records, device -> pinned host on a side stream while the next launch computes (HIP events around each copy)"}
    if own_tmp is not None:
        own_tmp.cleanup()
    return leg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cams", type=int, default=185)
    ap.add_argument("--latency-ms", type=float, default=20.0)
    ap.add_argument("--scene-root", default=None)
    ap.add_argument("--backend", default="device", choices=["device", "host"])
    ap.add_argument("--setting", default="fast")
    ap.add_argument("--size", default="1297x840")
    ap.add_argument("--num-refs", type=float, default=0.8, help="fraction (<= 1) or count of reference views (GUI default 0.8)")
    ap.add_argument("--nns", type=int, default=3, help="neighbours per reference (GUI default 3)")
    ap.add_argument("--refs-per-launch", type=int, default=16)
    a = ap.parse_args()
    from lichtfeld_densification_plugin_amd.core import hostenv
    hostenv.fit_threads_to_quota()
    w, h = (int(v) for v in a.size.split("x"))
    dev = torch.device("cuda", 0) if a.backend == "device" else torch.device("cpu")
    print(json.dumps({"pipeline": pipeline_leg(dev, n_cams=a.cams, latency_ms=a.latency_ms, scene_root=a.scene_root, backend=a.backend,
                                               roma_setting=a.setting, width=w, height=h, num_refs=a.num_refs, nns=a.nns, refs_per_launch=a.refs_per_launch)}),
          flush=True)


if __name__ == "__main__":
    main()
