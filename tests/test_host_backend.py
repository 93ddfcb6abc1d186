"""BASELINE config 1 ("2 COLMAP views, `turbo` (H_lr=320), --no-filter, CPU ... plumbing, no GPU") through the product's EXPLICIT
host backend: run_dense_pipeline(..., backend="host") / DensePipelineConfig(backend="host") = the CPU twin of the C-ABI
(lfd_create_host) + core/sampling.py, chosen by the caller.  Full size (2 cameras 1297x840, f ~ 960, baseline 0.6,
depth ~ 4, grid 320^2, k = 1, 2 references, no_filter, M = 12000, tie-free certainty) against the NumPy oracle.  Runs without a GPU."""
import math
import os

import numpy as np
import pytest
import torch

import lichtfeld_densification_plugin_amd as lfd
from lichtfeld_densification_plugin_amd import densify, synthetic
from lichtfeld_densification_plugin_amd.core import hip_backend as hb
from lichtfeld_densification_plugin_amd.core import pipeline as pl
from helpers import oracle_cam
from oracle import densify_oracle as orc

H = W = WM = HM = 320          # `turbo`
M = 12000                      # CLI default (densify.py:367 upstream)


class TableMatcher:
    """Replays the synthetic RoMa outputs (CPU tensors), one table row per reference in the order they are consumed."""
    sample_thresh = 0.9

    def __init__(self, table):
        self.w_resized, self.h_resized, self.table, self.calls = WM, HM, table, 0

    def match_grids_batch(self, imA, imB_list):
        res = self.table[self.calls]
        self.calls += 1
        assert len(res) == len(imB_list)
        return res

    def close(self):
        pass


def _scene(tmp_path, cams):
    """Config 1's two views with their images on disk and each view's warp / certainty into the other."""
    from PIL import Image
    srefs = []
    for i, cam in enumerate(cams):
        s = synthetic.synth_reference(cams, i, [1 - i], H, W, WM, HM, noise_px=0.3, outlier_frac=0.0, channels=4, seed=0,
                                      cert_mode="tiefree")
        cam.image_path = os.path.join(str(tmp_path), f"view{i}.png")
        Image.fromarray(s.image.numpy()).save(cam.image_path)
        srefs.append(s)
    return srefs


def _two_views():
    arc = 4.0 * math.asin(0.6 / (2.0 * 4.0))          # two cameras of the ring generator 0.6 apart at radius 4
    return synthetic.ring_cameras(2, width=1297, height=840, focal=960.0, radius=4.0, wobble=0.0, seed=0, arc=arc)


def _oracle_points(cams, srefs, order, params):
    xyz, rgb, err, counts = [], [], [], []
    rng = np.random.RandomState(0)
    for i in order:
        s = srefs[i]
        with np.errstate(all="ignore"):
            res, sel = orc.triangulate_reference([s.cert[0].numpy()], [s.warp[0].numpy()], s.image.numpy(), oracle_cam(cams[i]),
                                                 [oracle_cam(cams[1 - i])], WM, HM, params, rng=rng)
        assert sel.size == M
        xyz.append(res.xyz); rgb.append(res.rgb); err.append(res.err); counts.append(res.xyz.shape[0])
    return np.concatenate(xyz), np.concatenate(rgb), np.concatenate(err), counts


def test_config1_full_size_through_the_host_backend_equals_the_oracle(tmp_path):
    cams = _two_views()
    assert abs(np.linalg.norm(cams[0].C - cams[1].C) - 0.6) < 1e-3
    srefs = _scene(tmp_path, cams)
    flat = np.stack([c.flat_pose() for c in cams])
    from lichtfeld_densification_plugin_amd.core.selection import nearest_neighbors, select_cameras_kcenters
    refs_local = select_cameras_kcenters(flat, round(0.75 * 2))                 # CLI default num_refs 0.75 -> both views
    nn = nearest_neighbors(flat, 1)
    assert sorted(refs_local) == [0, 1]
    table = [[(srefs[r].warp[0], srefs[r].cert[0])] for r in refs_local]
    cfg = lfd.DensePipelineConfig(output_path=os.path.join(str(tmp_path), "out", "points3D_dense.ply"), roma_setting="turbo",
                                  num_refs=0.75, nns_per_ref=1, matches_per_ref=M, reproj_thresh=1.5, no_filter=True, seed=0,
                                  viz_interval=0, backend="host")
    progress = []
    res = pl.run_dense_pipeline(cams, refs_local, nn, cfg, progress_callback=lambda p, m: progress.append(p), matcher=TableMatcher(table))
    params = orc.OracleParams(certainty_thresh=0.2, reproj_thresh=1.5, sampson_thresh=5.0, min_parallax_deg=0.5, no_filter=True,
                              matches_per_ref=M)
    ox, oc, oe, counts = _oracle_points(cams, srefs, refs_local, params)
    assert res.xyz.shape[0] == ox.shape[0] == 2 * M                              # no_filter: every selected cell with finite results
    np.testing.assert_array_equal(res.points_per_reference, counts)
    np.testing.assert_allclose(res.xyz, ox, rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(res.rgb, oc)                                   # upstream's f64 blend, bit for bit
    np.testing.assert_allclose(res.err, oe, rtol=1e-4, atol=2e-3)
    assert res.pairs_processed == 2 and res.pairs_matched == 2 and progress[0] == 10.0 and progress[-1] == 90.0
    assert res.device_points is not None and res.device_points[0].device.type == "cpu"
    # the override argument does the same as the configuration field, and dense mode runs on the twin as well
    cfg2 = lfd.DensePipelineConfig(output_path=cfg.output_path, nns_per_ref=1, matches_per_ref=M, reproj_thresh=1.5, no_filter=True,
                                   viz_interval=0)
    again = pl.run_dense_pipeline(cams, refs_local, nn, cfg2, matcher=TableMatcher(table), backend="host")
    np.testing.assert_array_equal(again.xyz, res.xyz)
    cfg3 = lfd.DensePipelineConfig(output_path=cfg.output_path, nns_per_ref=1, reproj_thresh=1.5, viz_interval=0, backend="host",
                                   triangulation_mode="dense", refs_per_launch=2)
    dense = pl.run_dense_pipeline(cams, refs_local, nn, cfg3, matcher=TableMatcher(table))
    with np.errstate(all="ignore"):
        od = orc.triangulate_dense([srefs[refs_local[0]].cert[0].numpy()], [srefs[refs_local[0]].warp[0].numpy()],
                                   srefs[refs_local[0]].image.numpy(), oracle_cam(cams[refs_local[0]]), [oracle_cam(cams[1 - refs_local[0]])],
                                   WM, HM, orc.OracleParams(reproj_thresh=1.5))
    assert abs(int(dense.points_per_reference[0]) - od["xyz"].shape[0]) <= max(4, od["xyz"].shape[0] // 2000)


def test_the_device_backend_never_turns_into_the_host_one(tmp_path):
    """Without a GPU a "device" run raises; nothing selects the twin but the caller."""
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    cams = _two_views()
    cfg = lfd.DensePipelineConfig(output_path=os.path.join(str(tmp_path), "o.ply"), nns_per_ref=1)
    with pytest.raises(hb.HipBackendError, match="no GPU visible"):
        pl.run_dense_pipeline(cams, [0, 1], np.array([[1], [0]]), cfg, matcher=TableMatcher([]))
    with pytest.raises(ValueError):
        lfd.DensePipelineConfig(output_path="", backend="auto")
    with pytest.raises(ValueError, match="lives on"):
        pl.run_dense_pipeline(cams, [0, 1], np.array([[1], [0]]), cfg, matcher=TableMatcher([]), backend="host",
                              densifier=type("D", (), {"device": torch.device("cuda", 0), "upload_cameras": lambda s, c: None})())


class _Node:
    def __init__(self, cam):
        self.has_camera, self.camera_uid = True, cam.uid
        self.camera_width, self.camera_height = cam.width, cam.height
        self.camera_focal_x, self.camera_focal_y = float(cam.K[0, 0]), float(cam.K[1, 1])
        self.camera_R, self.camera_T = cam.R, cam.t.reshape(3)
        self.image_path, self.has_mask, self.mask_path = cam.image_path, False, None


def test_entry_point_writes_the_oracles_cloud_on_the_host_backend(tmp_path):
    """dense_init_from_lfs (the GUI entry point; dense_init differs only in reading COLMAP files through pycolmap, which is not
    installed here) with backend="host": the PLY holds the oracle's points, quantised by the host writer."""
    from lichtfeld_densification_plugin_amd.core import writers
    from lichtfeld_densification_plugin_amd.core.image_io import to_uint8_rgb
    base = _two_views()
    for c in base:
        c.image_path = ""
    recs = densify.extract_cameras_from_lfs([_Node(c) for c in base])          # principal point at the image centre, as the LFS path assumes
    srefs = _scene(tmp_path, recs)
    nodes = [_Node(c) for c in recs]
    out = os.path.join(str(tmp_path), "gui", "dense.ply")
    cfg = lfd.DensePipelineConfig(output_path=out, roma_setting="turbo", num_refs=2, nns_per_ref=4, matches_per_ref=M, no_filter=True,
                                  seed=0, viz_interval=0, backend="host", stream_output=True)
    from lichtfeld_densification_plugin_amd.core.selection import select_cameras_kcenters
    order = select_cameras_kcenters(np.stack([c.flat_pose() for c in recs]), 2)
    table = [[(srefs[r].warp[0], srefs[r].cert[0])] for r in order]
    code, info = densify.dense_init_from_lfs(nodes, cfg, matcher=TableMatcher(table))
    assert code == 0 and info == out
    params = orc.OracleParams(no_filter=True, matches_per_ref=M)
    ox, oc, oe, _ = _oracle_points(recs, srefs, order, params)
    head, body = open(out, "rb").read().split(b"end_header\n", 1)
    assert int(head.split(b"element vertex ")[1].split(b"\n")[0]) == ox.shape[0] == 2 * M
    rec = np.frombuffer(body, dtype=np.dtype([("p", "<f4", 3), ("c", "u1", 3)]))
    np.testing.assert_allclose(rec["p"], ox, rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(rec["c"], to_uint8_rgb(oc))
