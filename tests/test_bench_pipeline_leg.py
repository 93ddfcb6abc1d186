"""The `pipeline` leg of bench.py (bench_pipeline.py): densify.dense_init end to end on a generated on-disk scene with the analytic matcher - the leg's
contract on a small scene, on the host backend here and on the device with ``-m gpu``.  Reference: upstream core/pipeline.py:783-928 (the loop),
core/threaded_dataloader.py:42-241 (loading), core/writers.py:29-46 (the file) - SURVEY 8d's points/s definition (ii)."""
import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

RUN_KEYS = {"seconds", "references", "pairs", "points", "refs_per_s", "pairs_per_s", "points_per_s", "matcher_seconds", "file_bytes", "stage_seconds"}


def _check_runs(leg, expect_runs):
    assert leg["scene"]["cameras"] == 8 and leg["scene"]["references"] == 6 and leg["scene"]["pairs"] == 18 and leg["scene"]["neighbours"] == 3
    for mode in ("sampled", "dense"):
        runs = leg[mode]
        assert expect_runs <= set(runs), (mode, sorted(runs))
        pts = set()
        for name, r in runs.items():
            if name == "stages":
                continue
            assert RUN_KEYS <= set(r), (mode, name, sorted(r))
            assert r["references"] == 6 and r["pairs"] == 18 and r["points"] > 0 and r["seconds"] > 0
            assert abs(r["points_per_s"] - r["points"] / r["seconds"]) < 1e-6 * r["points_per_s"]
            assert r["file_bytes"] > r["points"] * 15 and r["file_bytes"] < r["points"] * 15 + 400         # header + 15-byte vertices
            assert {"decode", "prepare", "match", "kernel"} <= set(r["stage_seconds"])
            pts.add(r["points"])
        assert len(pts) == 1, (mode, pts)            # host / device preparation, with or without matcher latency: the same cloud
    lat = [n for n in leg["sampled"] if "ms_per_pair" in n][0]
    assert leg["sampled"][lat]["matcher_seconds"] >= 18 * 0.002 * 0.9           # the stand-in latency was served (2 ms per pair)
    assert leg["dense"][[n for n in leg["dense"] if n != "stages"][0]]["points"] > 5 * leg["sampled"][lat]["points"]


def test_pipeline_leg_contract_on_the_host_backend(tmp_path):
    import bench_pipeline
    leg = bench_pipeline.pipeline_leg(torch.device("cpu"), n_cams=8, latency_ms=2.0, scene_root=str(tmp_path), backend="host", roma_setting="turbo",
                                      width=320, height=208)
    _check_runs(leg, {"host_prep", "host_prep_matcher_2ms_per_pair"})
    assert "select" in leg["sampled"]["host_prep"]["stage_seconds"] and "write" in leg["dense"]["host_prep"]["stage_seconds"]
    # the scene stays where it was asked for, as upstream's CLI expects it
    assert os.path.isfile(os.path.join(str(tmp_path), "sparse", "0", "images.bin")) and len(os.listdir(os.path.join(str(tmp_path), "images_4"))) == 8


def test_the_generated_scene_is_what_the_matcher_warps(tmp_path):
    """write_colmap_scene + plan_scene + SyntheticMatcher: the cameras read back from the COLMAP files are the ones the fields are made from, and the
    matcher finds the pair it is asked about by keys or - for a driver that hands over nothing but images (upstream's) - by the images' fingerprints"""
    from lichtfeld_densification_plugin_amd import densify, synthetic
    from lichtfeld_densification_plugin_amd.core.image_io import load_rgb_u8
    synthetic.write_colmap_scene(str(tmp_path), n_cams=6, width=160, height=120, fmt="png")
    args = densify.build_argparser().parse_args(["--scene_root", str(tmp_path), "--images_subdir", "images_4", "--num_refs", "0.5", "--nns_per_ref", "2"])
    records, refs, nn, _ = densify.plan_scene(args)
    assert len(records) == 6 and len(refs) == 3 and all(os.path.isfile(r.image_path) for r in records)
    ring = synthetic.ring_cameras(6, width=160, height=120, seed=0)
    for a, b in zip(records, ring):
        np.testing.assert_allclose(a.P, b.P, rtol=2e-6, atol=2e-4)
    m = synthetic.SyntheticMatcher(records, setting="turbo", noise_px=0.0, outlier_frac=0.0, channels=4)
    assert m.precompute(refs, nn, 2) == 3
    for i, r in enumerate(records):
        m.register_image(i, load_rgb_u8(r.image_path, (m.w_resized, m.h_resized)))
    r0 = refs[0]
    nb = [int(n) for n in nn[r0][:2]]
    by_key = m.match_grids_batch(None, [None, None], keys=(r0, nb))
    by_img = m.match_grids_batch(load_rgb_u8(records[r0].image_path, (320, 320)), [load_rgb_u8(records[n].image_path, (320, 320)) for n in nb])
    for (w0, c0), (w1, c1) in zip(by_key, by_img):
        assert w0.shape == (320, 320, 4) and torch.equal(w0, w1) and torch.equal(c0, c1)
    # noise-free fields triangulate back onto the ground surface: the warp is the exact projection into the neighbour
    assert m.calls == 2 and m.pairs == 4


@pytest.mark.gpu
def test_pipeline_leg_contract_on_the_device(tmp_path):
    import bench_pipeline
    dev = torch.device("cuda", 0)
    leg = bench_pipeline.pipeline_leg(dev, n_cams=8, latency_ms=2.0, scene_root=str(tmp_path), backend="device", roma_setting="turbo", width=320, height=208,
                                      refs_per_launch=4)
    _check_runs(leg, {"host_prep", "device_prep", "device_prep_matcher_2ms_per_pair", "stages"})   # (+ "device_prep_<cores>_pack_workers" where the container has more than 4 cores)
    for mode in ("sampled", "dense"):
        st = leg[mode]["stages"]["seconds_per_stage"]
        assert {"decode", "prepare", "match", "kernel", "d2h"} <= set(st) and ("select" in st) == (mode == "sampled")
    d = leg["dense"]["device_prep"]
    assert d["d2h_bytes"] == 15 * d["points"] and d["d2h_GBps"] > 0 and "write" in d["stage_seconds"]      # records only: 15 B per survivor crossed PCIe
    p = leg["pcie"]
    assert p["d2h_pinned_copy_GBps"] > 1 and 0 < p["frac_of_measured_copy"] <= 1.5 and p["nominal_gen5_x16_GBps"] == 64.0
    # the GUI entry point on the same cameras as scene nodes: previews after every third reference that produced points, the same cloud without them
    g = leg["gui"]
    a, b = g["previews_every_3"], g["no_previews"]
    assert a["references"] == b["references"] == 6 and a["points"] == b["points"] > 0 and a["file_bytes"] == b["file_bytes"]
    assert a["previews"] == a["references"] // 3 and b["previews"] == 0 and a["preview_bytes"] > a["file_bytes"] // 2 and a["progress_callbacks"] > a["references"]
