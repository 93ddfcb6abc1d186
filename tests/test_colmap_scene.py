"""BASELINE config 1 from real COLMAP FILES: ``dense_init(args)`` (upstream densify.py:148-212) on a ``--scene_root`` whose ``sparse/0`` holds
cameras.bin / images.bin / points3D.bin - read by core/colmap_io.py because ``pycolmap`` is not installed in the ROCm image - `turbo`,
``--no_filter``, against the NumPy oracle.  The host backend runs here without a GPU; the device backend with ``-m gpu``."""
import os
import shutil

import numpy as np
import pytest
import torch

from lichtfeld_densification_plugin_amd import densify, synthetic
from lichtfeld_densification_plugin_amd.core import colmap_io as cio
from lichtfeld_densification_plugin_amd.core import pipeline as pl
from lichtfeld_densification_plugin_amd.core.image_io import to_uint8_rgb
from test_host_backend import HM, M, WM, TableMatcher, _oracle_points, _two_views
from oracle import densify_oracle as orc

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "colmap_2view")
H = W = 320


def test_reader_round_trips_the_fixture_and_matches_the_generating_cameras(tmp_path):
    rec = cio.Reconstruction(os.path.join(FIXTURE, "sparse", "0"))
    assert sorted(rec.cameras) == [1, 2] and sorted(rec.images) == [1, 2] and len(rec.points3D) == 40
    cams = _two_views()
    for i in range(2):
        open(os.path.join(str(tmp_path), f"view{i}.png"), "wb").close()              # (the records carry the image paths: the files must exist)
    records, ids = densify.camera_records_from_colmap(rec.cameras, rec.images, str(tmp_path))
    assert ids == [1, 2]
    for r, c in zip(records, cams):
        assert (r.width, r.height) == (1297, 840) and r.K.dtype == np.float32
        np.testing.assert_array_equal(r.K, c.K)                                      # PINHOLE: fx fy cx cy as stored (f64 -> f32)
        np.testing.assert_allclose(r.R, c.R, atol=2e-7)                              # through the quaternion and back
        np.testing.assert_allclose(r.t, c.t, atol=1e-6)
        np.testing.assert_allclose(r.P, c.P, rtol=1e-5, atol=1e-3)
    # the observations: 40 tracked points + one without a 3-D point in view 1, 30 + 1 in view 2
    p2 = rec.images[2].points2D
    assert len(p2) == 31 and sum(p.has_point3D() for p in p2) == 30 and len(rec.images[1].points2D) == 41
    from lichtfeld_densification_plugin_amd.core.selection import select_cameras_by_visibility
    assert select_cameras_by_visibility(rec, 1) == [1] and select_cameras_by_visibility(rec, 2) == [1, 2]
    # write -> read: the same bytes
    out = str(tmp_path)
    cio.write_cameras_bin(os.path.join(out, "cameras.bin"), [rec.cameras[i] for i in sorted(rec.cameras)])
    cio.write_images_bin(os.path.join(out, "images.bin"), [rec.images[i] for i in sorted(rec.images)])
    for f in ("cameras.bin", "images.bin"):
        assert open(os.path.join(out, f), "rb").read() == open(os.path.join(FIXTURE, "sparse", "0", f), "rb").read()
    # the text form of the same model
    with open(os.path.join(out, "cameras.txt"), "w") as fh:
        fh.write("# Camera list\n")
        for i in sorted(rec.cameras):
            c = rec.cameras[i]
            fh.write(f"{i} {c.model.name} {c.width} {c.height} " + " ".join(repr(float(v)) for v in c.params) + "\n")
    with open(os.path.join(out, "images.txt"), "w") as fh:
        fh.write("# Image list\n")
        for i in sorted(rec.images):
            im = rec.images[i]
            q, t = im.cam_from_world.rotation.quat, im.cam_from_world.translation
            fh.write(f"{i} " + " ".join(repr(float(v)) for v in list(q) + list(t)) + f" {im.camera_id} {im.name}\n")
            fh.write(" ".join(f"{p.xy[0]!r} {p.xy[1]!r} {p.point3D_id}" for p in im.points2D) + "\n")
    os.remove(os.path.join(out, "cameras.bin")); os.remove(os.path.join(out, "images.bin"))
    txt = cio.Reconstruction(out)
    for i in (1, 2):
        np.testing.assert_array_equal(txt.cameras[i].params, rec.cameras[i].params)
        np.testing.assert_array_equal(txt.images[i].cam_from_world.rotation.matrix(), rec.images[i].cam_from_world.rotation.matrix())
        assert [p.point3D_id for p in txt.images[i].points2D] == [p.point3D_id for p in rec.images[i].points2D]
    with pytest.raises(FileNotFoundError):
        cio.Reconstruction(os.path.join(out, "nothing_here"))


def _scene_root(tmp_path):
    """scene_root/{sparse/0, images_2}: the committed model + the synthetic images of its two views; returns the records the entry point
    will build from the files and each view's synthetic matcher output into the other"""
    from PIL import Image
    root = os.path.join(str(tmp_path), "scene")
    shutil.copytree(os.path.join(FIXTURE, "sparse"), os.path.join(root, "sparse"))
    os.makedirs(os.path.join(root, "images_2"))
    for i in range(2):
        open(os.path.join(root, "images_2", f"view{i}.png"), "wb").close()
    rec = cio.Reconstruction(os.path.join(root, "sparse", "0"))
    records, _ids = densify.camera_records_from_colmap(rec.cameras, rec.images, os.path.join(root, "images_2"))
    srefs = []
    for i in range(2):
        s = synthetic.synth_reference(records, i, [1 - i], H, W, WM, HM, noise_px=0.3, outlier_frac=0.0, channels=4, seed=0, cert_mode="tiefree")
        Image.fromarray(s.image.numpy()).save(os.path.join(root, "images_2", f"view{i}.png"))
        srefs.append(s)
    return root, records, srefs


def _run_dense_init(root, srefs, backend, monkeypatch):
    calls = {}

    def make_matcher(**kw):              # dense_init builds its own matcher: hand it the synthetic RoMa outputs, in consumption order
        calls["setting"] = kw.get("setting")
        return TableMatcher([[(srefs[r].warp[0], srefs[r].cert[0])] for r in (0, 1)])
    monkeypatch.setattr(pl, "RomaMatcher", make_matcher)
    monkeypatch.setattr(pl, "has_cached_romav2_weights", lambda: True)
    args = densify.build_argparser().parse_args(["--scene_root", root, "--roma_setting", "turbo", "--no_filter", "--nns_per_ref", "1",
                                                 "--backend", backend, "--pack_workers", "1"])
    progress = []
    rc = densify.dense_init(args, progress_callback=lambda p, m: progress.append((p, m)))
    assert rc == 0 and calls["setting"] == "turbo" and progress[-1][0] == 100.0
    return os.path.join(root, "sparse", "0", "points3D_dense.ply")


def _check_against_oracle(path, records, srefs):
    params = orc.OracleParams(certainty_thresh=0.2, reproj_thresh=1.5, sampson_thresh=5.0, min_parallax_deg=0.5, no_filter=True, matches_per_ref=M)
    ox, oc, _oe, _counts = _oracle_points(records, srefs, [0, 1], params)
    head, body = open(path, "rb").read().split(b"end_header\n", 1)
    assert int(head.split(b"element vertex ")[1].split(b"\n")[0]) == ox.shape[0] == 2 * M
    rec = np.frombuffer(body, dtype=np.dtype([("p", "<f4", 3), ("c", "u1", 3)]))
    np.testing.assert_allclose(rec["p"], ox, rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(rec["c"], to_uint8_rgb(oc))


def test_dense_init_from_scene_root_on_the_host_backend_equals_the_oracle(tmp_path, monkeypatch):
    root, records, srefs = _scene_root(tmp_path)
    _check_against_oracle(_run_dense_init(root, srefs, "host", monkeypatch), records, srefs)


@pytest.mark.gpu
def test_dense_init_from_scene_root_on_the_device_equals_the_oracle(tmp_path, monkeypatch):
    assert torch.cuda.is_available()
    root, records, srefs = _scene_root(tmp_path)
    _check_against_oracle(_run_dense_init(root, srefs, "device", monkeypatch), records, srefs)
