"""CPU tests of the host-side mirror of upstream's interface: boundary types, sampling stage,
writers, camera selection, entry-point plumbing (no GPU needed)."""
import dataclasses
import hashlib
import io
import os

import numpy as np
import pytest
import torch

import lichtfeld_densification_plugin_amd as lfd
from lichtfeld_densification_plugin_amd import densify
from lichtfeld_densification_plugin_amd.core import sampling, selection, writers
from lichtfeld_densification_plugin_amd.core.config import DensePipelineConfig
from lichtfeld_densification_plugin_amd.core.camera_models import CameraRecord
from lichtfeld_densification_plugin_amd.core.debug_viz import MatchDebugState, MatchPreview
from lichtfeld_densification_plugin_amd.core.image_io import to_uint8_rgb
from helpers import orc


UPSTREAM_FIELDS = [("output_path", None), ("roma_setting", "fast"), ("roi_only_selected", False), ("num_refs", 0.8),
                   ("nns_per_ref", 3), ("matches_per_ref", 10000), ("certainty_thresh", 0.20), ("reproj_thresh", 0.8),
                   ("sampson_thresh", 5.0), ("min_parallax_deg", 0.5), ("max_points", 0), ("no_filter", False),
                   ("use_masks", True), ("voxel_size", 0.0), ("seed", 0), ("viz_interval", 3), ("prefetch_packages", 8),
                   ("pack_workers", 4)]


def test_config_keeps_upstream_fields_in_order():
    fields = dataclasses.fields(DensePipelineConfig)
    assert [f.name for f in fields[:18]] == [n for n, _ in UPSTREAM_FIELDS]
    for f, (_, default) in zip(fields[1:18], UPSTREAM_FIELDS[1:]):
        assert f.default == default
    cfg = DensePipelineConfig("/tmp/x.ply", "turbo", False, 0.5, 2)       # positional construction still works
    assert cfg.roma_setting == "turbo" and cfg.nns_per_ref == 2 and cfg.triangulation_mode == "sampled"
    with pytest.raises(ValueError):
        DensePipelineConfig("/tmp/x.ply", triangulation_mode="bogus")


def test_camera_record_flat_pose_and_derived():
    K = np.array([[900, 0, 640], [0, 905, 360], [0, 0, 1]], np.float32)
    R = np.eye(3, dtype=np.float32)[[1, 2, 0]]
    t = np.array([0.5, -1.0, 2.0], np.float32)
    c = CameraRecord.from_krt(7, K, R, t, 1280, 720)
    T = c.flat_pose().reshape(4, 4)
    np.testing.assert_array_equal(T[:3, :3], R)
    np.testing.assert_array_equal(T[:3, 3], t)
    np.testing.assert_array_equal(c.P, K @ np.concatenate([R, t.reshape(3, 1)], 1))
    np.testing.assert_allclose(c.C, -R.T @ t, atol=1e-6)
    assert c.P.dtype == np.float32 and c.C.shape == (3,)


def _tiefree(h, w, seed):
    rs = np.random.RandomState(seed)
    perm = rs.permutation(h * w).astype(np.float64)
    return (0.2 + 0.7 * (perm + 0.5) / (h * w)).astype(np.float32).reshape(h, w)


def test_sampling_stage_matches_upstream_golden(g2):
    for ci, (h, w, M, seed, cseed) in enumerate(g2["cases"]):
        cert = _tiefree(int(h), int(w), int(cseed))
        for no_filter in (False, True):
            key = f"c{ci}_{'nf' if no_filter else 'f'}_"
            rng = np.random.RandomState(int(seed))
            sel = sampling.select_samples_with_coverage(torch.from_numpy(cert), int(M), no_filter=no_filter, rng=rng)
            np.testing.assert_array_equal(sel, g2[key + "sel"])
            assert int(rng.get_state()[2]) == int(g2[key + "mt_pos"])
            # the process-global stream gives the same answer (upstream's way of calling it)
            np.random.seed(int(seed))
            sel2 = sampling.select_samples_with_coverage(cert, int(M), no_filter=no_filter)
            np.testing.assert_array_equal(sel2, sel)
    assert sampling.select_samples_with_coverage(np.zeros((16, 16), np.float32), 100).size == 0
    with pytest.raises(ValueError):
        sampling.select_samples_with_coverage(_tiefree(64, 64, 1), 10000, rng=np.random.RandomState(0))


def test_sampling_coverage_pass_equals_the_walk_on_tied_maps(g2):
    """Floor/cap clamps create massive ties; the vectorised first-occurrence pass must equal upstream's
    sequential walk for whatever order argsort yields."""
    cert = g2["ties_cert"]
    sel = sampling.select_samples_with_coverage(cert, 1200, rng=np.random.RandomState(3))
    ref = orc.select_samples(cert, 1200, rng=np.random.RandomState(3))
    np.testing.assert_array_equal(sel, ref)


def test_writers_byte_exact(g5, tmp_path):
    u8 = to_uint8_rgb(g5["rgb"])
    np.testing.assert_array_equal(u8, g5["rgb_u8"])
    p1, p2 = tmp_path / "a.ply", tmp_path / "a.bin"
    writers.write_ply(str(p1), g5["xyz"], u8)
    writers.write_points3D_bin(str(p2), g5["xyz"], u8, g5["err"])
    assert p1.read_bytes() == g5["ply"].tobytes()
    assert p2.read_bytes() == g5["points3d_bin"].tobytes()
    writers.write_points3D_bin(str(p2), g5["xyz"], u8, None)
    assert len(p2.read_bytes()) == 8 + 43 * 5


def test_streamed_ply_writer_is_a_valid_ply_with_the_same_payload(g5, tmp_path):
    u8 = g5["rgb_u8"]
    path = tmp_path / "s.ply"
    with writers.StreamedPlyWriter(str(path)) as w:
        w.append(g5["xyz"][:2], u8[:2])
        w.append(g5["xyz"][2:], u8[2:])
        assert w.count == 5
    raw = path.read_bytes()
    head, body = raw.split(b"end_header\n", 1)
    assert b"element vertex 5\n" in head and head.startswith(b"ply\nformat binary_little_endian 1.0\n")
    assert body == g5["ply"].tobytes().split(b"end_header\n", 1)[1]


def test_selection_tables_match_upstream(g6):
    flat = g6["flat_poses"]
    for k in (3, 4, 8):
        np.testing.assert_array_equal(selection.nearest_neighbors(flat, k), g6[f"nn_k{k}"])
    for k in (1, 56, 148):
        np.testing.assert_array_equal(selection.select_cameras_kcenters(flat, k), g6[f"kcenters_{k}"])
    np.testing.assert_array_equal(selection.nearest_neighbors(flat[:2], 5), g6["nn_two"])
    assert selection.nearest_neighbors(flat[:1], 3).shape == (1, 0)


def test_argparser_has_upstream_flags_and_defaults():
    a = densify.build_argparser().parse_args(["--scene_root", "/x"])
    assert (a.images_subdir, a.out_name, a.roma_setting, a.num_refs, a.nns_per_ref, a.matches_per_ref) == \
        ("images_2", "points3D_dense.ply", "fast", 0.75, 4, 12000)
    assert (a.certainty_thresh, a.reproj_thresh, a.sampson_thresh, a.min_parallax_deg) == (0.20, 1.5, 5.0, 0.5)
    assert (a.no_filter, a.max_points, a.prefetch_packages, a.pack_workers, a.seed) == (False, 0, 8, 4, 0)
    assert a.triangulation_mode == "sampled"
    with pytest.raises(SystemExit):
        densify.build_argparser().parse_args(["--scene_root", "/x", "--roma_setting", "ultra"])


class _Node:
    def __init__(self, uid, R, T, w=640, h=480, has_camera=True):
        self.has_camera, self.camera_uid = has_camera, uid
        self.camera_width, self.camera_height = w, h
        self.camera_focal_x, self.camera_focal_y = 500.0, 505.0
        self.camera_R, self.camera_T = R, T
        self.image_path, self.has_mask, self.mask_path = f"/nowhere/{uid}.png", False, None


def test_dense_init_from_lfs_argument_checks():
    cfg = DensePipelineConfig(output_path="/tmp/out.ply")
    code, msg = densify.dense_init_from_lfs([_Node(1, np.eye(3), np.zeros(3))], cfg)
    assert (code, msg) == (1, "Need at least 2 cameras for dense initialization")
    recs = densify.extract_cameras_from_lfs([_Node(1, np.eye(3), np.zeros(3)), _Node(2, np.eye(3), np.ones(3), has_camera=False)])
    assert len(recs) == 1 and recs[0].K[0, 2] == 320.0 and recs[0].K[1, 2] == 240.0


def test_point_cap_and_voxel_filter():
    rs = np.random.RandomState(0)
    xyz = rs.uniform(0, 1, (500, 3)).astype(np.float32)
    rgb = rs.uniform(0, 1, (500, 3)).astype(np.float32)
    err = rs.uniform(0, 1, 500).astype(np.float32)
    a, b, c = densify._apply_point_cap(xyz, rgb, err, 100, 3)
    keep = np.random.default_rng(3).choice(500, size=100, replace=False)
    np.testing.assert_array_equal(a, xyz[keep])
    assert densify._apply_point_cap(xyz, rgb, err, 0, 3)[0] is xyz
    vx, vc = densify._voxel_downsample(xyz, rgb, 0.25)
    assert 1 <= vx.shape[0] <= 125 and vx.dtype == np.float32 and vc.min() >= 0 and vc.max() <= 1
    np.testing.assert_allclose(vx.mean(0), xyz.mean(0), atol=0.1)


def test_debug_state_blocks_until_stepped():
    """The five calls run_dense_pipeline makes on the host's debug object (core/debug_viz.py here documents them)."""
    import threading
    st = MatchDebugState()
    pv = MatchPreview(1, 2, "a", "b", np.zeros((2, 2, 3), np.uint8), np.zeros((2, 2, 3), np.uint8),
                      np.zeros((1, 4), np.float32), np.zeros(1, np.float32), 1, 1, 1)
    st.submit_preview(pv)
    assert st.latest() is None and not st.is_enabled()      # disabled: dropped
    st.set_enabled(True)
    st.set_total_pairs(7)
    assert st.is_auto_step() and st.total_pairs() == 7
    st.submit_preview(pv)                                      # auto stepping: returns at once
    st.set_auto_step(False)
    done = threading.Event()
    th = threading.Thread(target=lambda: (st.submit_preview(pv), done.set()))
    th.start()
    assert not done.wait(0.2)                                  # manual stepping parks the producer ...
    st.step_once()
    assert done.wait(2.0)                                      # ... until the consumer steps
    th.join()
    assert st.latest() is pv
    done.clear()
    th = threading.Thread(target=lambda: (st.submit_preview(pv), done.set()))
    th.start()
    assert not done.wait(0.2)
    st.release_waiters()                                       # the pipeline's `finally`: nobody stays parked
    assert done.wait(2.0)
    th.join()
    st.submit_preview(pv)                                      # and nobody parks afterwards


def test_host_arrays_slices_the_packed_buffer_like_three_copies():
    """TriangulationOutput.host_arrays: one copy of the packed float buffer must yield what three separate copies of the
    xyz / rgb / err views yield (layout [xyz cap*3][rgb cap*3][err cap])."""
    import numpy as np
    import torch
    from lichtfeld_densification_plugin_amd.core import hip_backend as hb
    cap, n = 37, 21
    packed = torch.arange(cap * 7, dtype=torch.float32)
    xyz, rgb, err = packed[:cap * 3].view(cap, 3), packed[cap * 3:cap * 6].view(cap, 3), packed[cap * 6:]
    out = hb.TriangulationOutput(xyz=xyz[:n], rgb=rgb[:n], err=err[:n], cell=None, slot=None,
                                 ref_offsets=np.array([0, n], np.int64), seg_counts=np.zeros((1, 3), np.int32),
                                 _packed=packed, _cap=cap)
    hx, hc, he = out.host_arrays()
    np.testing.assert_array_equal(hx, xyz[:n].numpy())
    np.testing.assert_array_equal(hc, rgb[:n].numpy())
    np.testing.assert_array_equal(he, err[:n].numpy())
    assert hx.flags["C_CONTIGUOUS"] and hc.flags["C_CONTIGUOUS"] and out.count == n
    # without the packed buffer (or when it is large) the three views are copied one by one
    out2 = hb.TriangulationOutput(xyz=xyz[:n], rgb=rgb[:n], err=err[:n], cell=None, slot=None,
                                  ref_offsets=np.array([0, n], np.int64), seg_counts=np.zeros((1, 3), np.int32))
    for a, b in zip(out2.host_arrays(), (hx, hc, he)):
        np.testing.assert_array_equal(a, b)


def test_host_threads_fit_the_container_quota(monkeypatch, tmp_path):
    """core/hostenv.py: the cgroup quota is read (v2 `cpu.max`), torch's intra-op threads are lowered to it and never raised, and a user's
    OMP_NUM_THREADS wins"""
    import builtins
    import torch
    from lichtfeld_densification_plugin_amd.core import hostenv
    real_open = builtins.open

    def fake_open(path, *a, **kw):
        if path == "/sys/fs/cgroup/cpu.max":
            return real_open(str(tmp_path / "cpu.max"), *a, **kw)
        return real_open(path, *a, **kw)
    monkeypatch.setattr(builtins, "open", fake_open)
    (tmp_path / "cpu.max").write_text("300000 100000\n")
    assert hostenv.cpu_quota() == 3.0 and hostenv.usable_cores() <= 3
    (tmp_path / "cpu.max").write_text("max 100000\n")
    assert hostenv.cpu_quota() is None
    before = torch.get_num_threads()
    try:
        (tmp_path / "cpu.max").write_text("200000 100000\n")
        monkeypatch.delenv("OMP_NUM_THREADS", raising=False)
        monkeypatch.delenv("MKL_NUM_THREADS", raising=False)
        torch.set_num_threads(max(before, 4))
        msgs = []
        assert hostenv.fit_threads_to_quota(log=msgs.append) == min(2, hostenv.usable_cores()) and torch.get_num_threads() <= 2 and msgs
        (tmp_path / "cpu.max").write_text("6400000 100000\n")           # a quota above what torch uses: nothing is raised
        assert hostenv.fit_threads_to_quota() == torch.get_num_threads() <= 2
        torch.set_num_threads(max(before, 4))
        monkeypatch.setenv("OMP_NUM_THREADS", "4")                       # the user's setting wins
        (tmp_path / "cpu.max").write_text("100000 100000\n")
        assert hostenv.fit_threads_to_quota() == max(before, 4)
    finally:
        torch.set_num_threads(before)


def test_intrinsics_follow_upstreams_model_chain_in_its_order():
    """K_from_camera (upstream core/geometry.py:10-30): SIMPLE_RADIAL_FISHEYE is a SIMPLE_RADIAL (f, cx, cy) before it is a FISHEYE (fx, fy, cx, cy) -
    the order of upstream's tests matters (found by tests/golden/check_oracle_fuzz.py against upstream's own function)"""
    import types
    from lichtfeld_densification_plugin_amd import densify

    def cam(name, params):
        return types.SimpleNamespace(model=types.SimpleNamespace(name=name), params=np.asarray(params, np.float64), width=1000, height=800)
    for name, params, want in [("PINHOLE", [900, 910, 500, 400], (900, 910, 500, 400)), ("SIMPLE_PINHOLE", [900, 500, 400], (900, 900, 500, 400)),
                               ("SIMPLE_RADIAL", [900, 500, 400, 0.1], (900, 900, 500, 400)), ("RADIAL", [900, 500, 400, 0.1, 0.0], (900, 900, 500, 400)),
                               ("OPENCV", [900, 910, 500, 400, 0, 0, 0, 0], (900, 910, 500, 400)), ("OPENCV_FISHEYE", [900, 910, 500, 400, 0, 0, 0, 0], (900, 910, 500, 400)),
                               ("SIMPLE_RADIAL_FISHEYE", [900, 500, 400, 0.1], (900, 900, 500, 400)),
                               ("RADIAL_FISHEYE", [900, 500, 400, 0.1, 0.0], (900, 500, 400, 0.1)),       # (upstream reads four parameters here: reproduced, not corrected)
                               ("FOV", [900, 910, 500, 400, 0.9], (900, 900, 910, 500)), ("UNKNOWN", [900], (900, 900, 500, 400))]:
        K = densify.K_from_camera(cam(name, params))
        assert K.dtype == np.float32 and (float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2])) == tuple(float(np.float32(v)) for v in want), name


def test_visibility_selection_is_upstreams_greedy_loop_at_scene_size():
    """core/selection.py::select_cameras_by_visibility computes upstream's picks (core/selection.py:10-33 there: the first by observations incl.
    duplicates, the others by distinct uncovered points, ties to the first image) incrementally; against the plain loop on a scene-sized model -
    60 images, 20 000 points, 0.3 M observations with duplicates, unobserved entries and ties."""
    from lichtfeld_densification_plugin_amd.core import colmap_io as cio, selection
    rs = np.random.RandomState(3)
    n_img, n_pts = 60, 20000

    class Rec:
        points3D = {1: 1}
        images = {}
    rec = Rec()
    for i in range(n_img):
        m = int(rs.randint(2000, 8000))
        p = (int(rs.randint(0, n_pts)) + rs.randint(-4000, 4000, size=m)) % n_pts + 1
        p[rs.rand(m) < 0.1] = -1
        if i % 7 == 3:
            p = rec.images[i].observed_point3D_ids().copy() if i in rec.images else p           # an exact duplicate of the previous image: a tie
        rec.images[i + 1] = cio.Image(i + 1, [1, 0, 0, 0], [0, 0, 0], 1, f"{i}.jpg", np.zeros((p.size, 2)), p)

    def upstream_loop(k):
        lists = {im.image_id: [int(x) for x in im.observed_point3D_ids()] for im in rec.images.values()}
        k = min(k, len(lists))
        sel, cov, scores = [], set(), {i: len(p) for i, p in lists.items()}
        for _ in range(k):
            if not scores:
                break
            b = max(scores, key=scores.get)
            sel.append(b)
            cov.update(set(lists[b]) - cov)
            del scores[b]
            for c, pts in lists.items():
                if c in scores:
                    scores[c] = len(set(pts) - cov)
        return sorted(sel)
    for k in (1, 7, 48, 60, 200):
        assert selection.select_cameras_by_visibility(rec, k) == upstream_loop(k), k
    # objects that only offer pycolmap's interface (points2D / has_point3D) take the same path through the generic extraction
    class Img:
        def __init__(self, im):
            self.image_id, self._im = im.image_id, im
        points2D = property(lambda self: self._im.points2D)
    rec2 = Rec()
    rec2.images = {i: Img(im) for i, im in list(rec.images.items())[:12]}
    rec.images = dict(list(rec.images.items())[:12])
    assert selection.select_cameras_by_visibility(rec2, 9) == upstream_loop(9)


def test_settings_that_moved_to_experimental_are_still_accepted_and_readable():
    """Seven settings were dataclass fields before the configuration surface was collapsed: a caller written against that surface keeps
    running (keyword and attribute both forward to ``experimental``, each with a DeprecationWarning); an unknown keyword is still a TypeError."""
    import dataclasses
    import warnings
    import lichtfeld_densification_plugin_amd as lfd
    with warnings.catch_warnings():
        warnings.simplefilter("error")                         # the plain surface warns about nothing
        cfg = lfd.DensePipelineConfig("a.ply", triangulation_mode="dense", experimental={"dense_tile_segments": True})
        assert dataclasses.replace(cfg, seed=3).exp("dense_tile_segments") is True
    with pytest.warns(DeprecationWarning, match="moved to experimental"):
        old = lfd.DensePipelineConfig("a.ply", triangulation_mode="dense", dense_tile_segments=True, exchange_round=8, experimental={"exchange_round": 4})
    assert old.experimental == {"dense_tile_segments": True, "exchange_round": 4}                      # an explicit experimental entry wins
    with pytest.warns(DeprecationWarning, match=r"experimental\['upstream_fundamental'\]"):
        assert old.upstream_fundamental is True
    assert "dense_tile_segments" not in {f.name for f in dataclasses.fields(old)}
    with pytest.raises(ValueError, match="needs triangulation_mode='dense'"), warnings.catch_warnings():
        warnings.simplefilter("ignore", DeprecationWarning)
        lfd.DensePipelineConfig("a.ply", dense_tile_segments=True)                                      # the moved keyword is validated like the dict entry
    with pytest.raises(TypeError):
        lfd.DensePipelineConfig("a.ply", no_such_setting=1)
