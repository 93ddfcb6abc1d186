#!/usr/bin/env python3
"""UPSTREAM'S OWN code timed on the benchmark's workloads - development container only (it imports /root/reference through
tests/golden/ref_import.py; nothing of it is copied, the GPU box never sees it).

    python tests/golden/time_reference.py [--write] [--cams 185] [--quick]

SURVEY 8d: "the reference's Python is timed here ... on identical fixtures; report reference-Python time (this container, core count)".  Two legs:

  A  ``_triangulate_ref`` (core/pipeline.py:602-780) on references of bench.py's headline workload (`config2`: garden-like ring of 185 cameras,
     `fast` 512 x 512 grid, noise 0.5 px, 5 % outliers, smooth certainty; 3 neighbours with the GUI's thresholds and M = 10000, and 4 neighbours
     with the CLI's: reprojection 1.5 px, M = 12000) - seconds per reference, points, points/s.
  B  ``run_dense_pipeline`` (core/pipeline.py:783-928: threaded loader, PIL resize, matcher call, `_collect_reference_matches`, `_triangulate_ref`,
     accumulation) + ``write_ply`` (core/writers.py:29-46) on the on-disk scene of bench.py's `pipeline` leg (synthetic.write_colmap_scene, the same
     cameras / images / reference plan / matcher fields), with ``pack_workers`` 1 and 4 - seconds, references/s, pairs/s, points/s, and the
     matcher's own share of the wall time - and once more with the GUI's intermediate previews (viz_interval = 3: 49 cumulative PLYs).  Upstream's loader hands packages over in completion order, so the stand-in matcher recognises the
     images it is handed by a fingerprint (synthetic.SyntheticMatcher.register_image) instead of counting calls.

``--write`` stores the record as tests/golden/g11_reference_timing.json; bench.py quotes it as ``cpu_baseline.reference_python``."""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

from ref_import import load_reference  # noqa: E402
from lichtfeld_densification_plugin_amd import densify, synthetic  # noqa: E402
from lichtfeld_densification_plugin_amd.core import hostenv  # noqa: E402


class _Table:
    sample_thresh = 0.9

    def __init__(self, wm, hm, table):
        self.w_resized, self.h_resized, self.table, self.calls = wm, hm, table, 0

    def match_grids_batch(self, imA, imB_list):
        res = self.table[self.calls]
        self.calls += 1
        return [(w.clone(), c.clone()) for (w, c) in res]

    def close(self):
        pass


def leg_triangulate_ref(ns, n_refs: int, k: int, cfg_kw: dict) -> dict:
    """bench.py::build_workload's references (same cameras, neighbours, seeds, noise) through upstream's per-reference stage."""
    P = ns.pipeline
    cams = synthetic.ring_cameras(185, seed=0)
    H = W = wm = hm = 512
    cfg = ns.config.DensePipelineConfig(output_path="/tmp/x.ply", nns_per_ref=k, **cfg_kw)
    lookup = P._build_camera_lookup(cams)
    ids = [c.uid for c in cams]
    ctx = P._TriangulationContext(cameras=lookup, config=cfg, matcher_sample_cap=0.9, w_match=wm, h_match=hm)
    secs, pts, collect_s = [], [], []
    np.random.seed(0)
    for gi in range(n_refs):
        ref = (gi * 3) % 185
        nbrs = synthetic.ring_neighbours(185, ref, k)
        s = synthetic.synth_reference(cams, ref, nbrs, H, W, wm, hm, noise_px=0.5, outlier_frac=0.05, channels=4, seed=1000 + gi, cert_mode="smooth")
        packed = P._PackedReferenceBatch(ref_id=ids[ref], ref_path=cams[ref].image_path, imA_np=s.image.numpy(), maskA_np=None, wA_cam=cams[ref].width,
                                         hA_cam=cams[ref].height, nn_ids=[ids[n] for n in nbrs], nn_masks=[None] * k,
                                         nn_arrays=[np.zeros_like(s.image.numpy()) for _ in nbrs])
        fm = _Table(wm, hm, [[(s.warp[j], s.cert[j]) for j in range(k)]])
        t0 = time.perf_counter()
        matched, _ = P._collect_reference_matches(packed, fm, cfg, 0, None)
        t1 = time.perf_counter()
        with np.errstate(all="ignore"):
            tri = P._triangulate_ref(matched, ctx, collect_debug_matches=False)
        t2 = time.perf_counter()
        collect_s.append(t1 - t0)
        secs.append(t2 - t1)
        pts.append(0 if tri is None else int(tri.xyz.shape[0]))
    tot = float(np.sum(secs))
    return {"references": n_refs, "neighbours": k, "grid": [H, W], "matches_per_ref": cfg.matches_per_ref, "reproj_thresh": cfg.reproj_thresh,
            "seconds_per_reference": {"mean": float(np.mean(secs)), "min": float(np.min(secs)), "max": float(np.max(secs))},
            "collect_seconds_per_reference": float(np.mean(collect_s)),        # the epilogue of _collect_reference_matches (floor, masks, the copy to the host)
            "points_per_reference": float(np.mean(pts)), "points_per_s": float(np.sum(pts)) / tot, "pairs_per_s": n_refs * k / tot, "refs_per_s": n_refs / tot}


def make_scene(scene_root: str, n_cams: int, width: int, height: int):
    """bench_pipeline.py's scene and reference plan (before the stub `pycolmap` of ref_import is in sys.modules: the scene is read with this
    package's own COLMAP reader, as on the GPU box)"""
    synthetic.write_colmap_scene(scene_root, n_cams=n_cams, width=width, height=height, images_subdir="images_4", fmt="jpg", seed=0)
    args = densify.build_argparser().parse_args(["--scene_root", scene_root, "--images_subdir", "images_4", "--num_refs", "0.8", "--nns_per_ref", "3"])
    return densify.plan_scene(args)


def leg_pipeline(ns, plan, n_cams: int, width: int, height: int, workers: int, setting: str = "fast", viz_interval: int = 0) -> dict:
    """upstream's run_dense_pipeline + write_ply on the pipeline leg's scene.  ``viz_interval`` > 0: with the GUI's intermediate previews
    (core/pipeline.py:508-532 upstream: the cloud so far re-concatenated and written by write_ply after every viz_interval-th reference that produced
    points, the path handed to on_sequential_viz - here a callback that reads the file's size and removes it, as bench_pipeline.py's GUI leg does)"""
    P = ns.pipeline
    records, refs_local, nn_table, sparse = plan
    matcher = synthetic.SyntheticMatcher(records, setting=setting, device="cpu", noise_px=0.5, outlier_frac=0.05, channels=4, seed=0)
    matcher.precompute(refs_local, nn_table, 3)
    size = (matcher.w_resized, matcher.h_resized)
    for i, r in enumerate(records):                       # what upstream's loader will hand the matcher: its own PIL BILINEAR resize
        matcher.register_image(i, np.asarray(ns.image_utils.load_rgb_resized(r.image_path, size), dtype=np.uint8))
    for fn in (ns.image_utils.load_rgb_resized, getattr(ns.image_utils, "load_mask_resized_np", None)):
        if fn is not None and hasattr(fn, "cache_clear"):
            fn.cache_clear()                              # the timed run decodes every image itself
    P.RomaMatcher = lambda device="cpu", mode="outdoor", setting="fast", _m=matcher: _m
    P.has_cached_romav2_weights = lambda: True
    out_path = os.path.join(sparse, "reference_timing.ply")
    cfg = ns.config.DensePipelineConfig(output_path=out_path, roma_setting=setting, num_refs=0.8, nns_per_ref=3, matches_per_ref=10000, reproj_thresh=0.8,
                                        viz_interval=int(viz_interval), pack_workers=workers, seed=0)
    matcher.calls = matcher.pairs = 0
    matcher.seconds = 0.0
    seen = {"previews": 0, "preview_bytes": 0, "seconds": 0.0, "last": None}

    def on_viz(path):
        seen["previews"] += 1
        seen["preview_bytes"] += os.path.getsize(path)
        os.remove(path)
    t0 = time.perf_counter()
    with np.errstate(all="ignore"):
        res = P.run_dense_pipeline(records, refs_local, nn_table, cfg, on_sequential_viz=(on_viz if viz_interval > 0 else None))
    t1 = time.perf_counter()
    ns.writers.write_ply(out_path, res.xyz, ns.image_utils.to_uint8_rgb(res.rgb))
    t2 = time.perf_counter()
    n = int(res.xyz.shape[0])
    os.remove(out_path)
    dt = t2 - t0
    return {"pack_workers": workers, "cameras": n_cams, "image_size": [width, height], "references": matcher.calls, "pairs": matcher.pairs, "points": n,
            "seconds": dt, "pipeline_seconds": t1 - t0, "write_ply_seconds": t2 - t1, "matcher_seconds": matcher.seconds,
            "refs_per_s": matcher.calls / dt, "pairs_per_s": matcher.pairs / dt, "points_per_s": n / dt,
            "seconds_per_reference_without_matcher": (dt - matcher.seconds) / max(1, matcher.calls),
            "viz_interval": int(viz_interval), "previews": seen["previews"], "preview_bytes": seen["preview_bytes"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write", action="store_true")
    ap.add_argument("--cams", type=int, default=185)
    ap.add_argument("--refs", type=int, default=6, help="references of leg A per configuration")
    ap.add_argument("--quick", action="store_true", help="a small scene (16 cameras) for leg B: a smoke run of this script")
    ap.add_argument("--no-previews", action="store_true", help="skip the run with the GUI's intermediate previews (minutes: upstream writes every preview point by point)")
    a = ap.parse_args()
    threads = hostenv.fit_threads_to_quota()
    n_cams = 16 if a.quick else a.cams
    tmp = tempfile.TemporaryDirectory(prefix="lfd_ref_scene_")
    plan = make_scene(tmp.name, n_cams, 1297, 840)
    ns = load_reference()
    rec = {"what": "upstream's own Python (imported from /root/reference, nothing copied) timed in the development container on the benchmark's workloads",
           "host": {"cores_usable": hostenv.usable_cores(), "cpus_visible": os.cpu_count(), "torch_threads": threads},
           "versions": {"numpy": np.__version__, "torch": torch.__version__, "python": sys.version.split()[0]}}
    rec["triangulate_ref"] = {"gui_k3": leg_triangulate_ref(ns, a.refs, 3, dict(matches_per_ref=10000, reproj_thresh=0.8)),
                              "cli_k4": leg_triangulate_ref(ns, a.refs, 4, dict(matches_per_ref=12000, reproj_thresh=1.5))}
    print(json.dumps(rec["triangulate_ref"], indent=1), flush=True)
    rec["run_dense_pipeline"] = {}
    for workers in (1, 4):
        rec["run_dense_pipeline"][f"pack_workers_{workers}"] = leg_pipeline(ns, plan, n_cams, 1297, 840, workers)
        print(json.dumps(rec["run_dense_pipeline"][f"pack_workers_{workers}"]), flush=True)
    if not a.no_previews:
        # the GUI's default (DensePipelineConfig.viz_interval = 3, panels/densification.py hands on_sequential_viz in): what bench_pipeline.py's `gui` leg runs
        rec["run_dense_pipeline"]["pack_workers_4_previews_every_3"] = leg_pipeline(ns, plan, n_cams, 1297, 840, 4, viz_interval=3)
        print(json.dumps(rec["run_dense_pipeline"]["pack_workers_4_previews_every_3"]), flush=True)
    tmp.cleanup()
    if a.write:
        with open(os.path.join(HERE, "g11_reference_timing.json"), "w") as fh:
            json.dump(rec, fh, indent=1)
            fh.write("\n")
        print("wrote g11_reference_timing.json")


if __name__ == "__main__":
    main()
