"""The oracle's RESTATEMENTS of library calls against the libraries themselves (torch on the CPU, here without a GPU): where the oracle does not
call the library upstream calls but re-derives its arithmetic (so that the kernels can be written from it), the re-derivation is fuzzed against
the real call - F.interpolate(nearest) of the masks (upstream core/pipeline.py:372-378), F.grid_sample(nearest, zeros, align_corners=False) of
mask B at the warped position (:419-430), torch.clamp(min=) + torch.max(dim=0) of the certainty stack (:407, :632-634).  (The golden fixtures pin
the same functions on a handful of upstream-captured cases; this pins them on thousands of adversarial ones.)"""
import numpy as np
import torch
import torch.nn.functional as F

from helpers import orc


def test_nearest_mask_resize_is_f_interpolate():
    rs = np.random.RandomState(0)
    sizes = [(1, 1, 1, 1), (5, 7, 5, 7), (64, 64, 320, 320), (840, 1297, 512, 512), (3, 1000, 7, 3), (17, 31, 1, 1), (2, 2, 999, 5)]
    while len(sizes) < 60:
        sizes.append((int(rs.randint(1, 400)), int(rs.randint(1, 400)), int(rs.randint(1, 400)), int(rs.randint(1, 400))))
    for ih, iw, oh, ow in sizes:
        m = (rs.rand(ih, iw) > 0.5).astype(np.float32)
        want = F.interpolate(torch.from_numpy(m).view(1, 1, ih, iw), size=(oh, ow), mode="nearest").view(oh, ow).numpy()
        np.testing.assert_array_equal(orc.nearest_resize_mask(m, (oh, ow)), want, err_msg=f"{ih}x{iw} -> {oh}x{ow}")


def test_warped_mask_is_f_grid_sample_nearest():
    rs = np.random.RandomState(1)
    for h, w in [(1, 1), (2, 3), (64, 64), (37, 53), (320, 320), (5, 400)]:
        mask = (rs.rand(h, w) > 0.4).astype(np.float32)
        n = 20000
        g = rs.uniform(-1.3, 1.3, size=(n, 2)).astype(np.float32)
        # adversarial coordinates: exactly on the half-way points between two pixels (round half to even), on the image edge, just outside
        kx = rs.randint(-2, w + 2, size=4000)
        ky = rs.randint(-2, h + 2, size=4000)
        half = np.stack([((2 * (kx + 0.5) + 1) / w - 1), ((2 * (ky + 0.5) + 1) / h - 1)], 1).astype(np.float32)
        edge = np.array([[-1.0, -1.0], [1.0, 1.0], [-1.0, 1.0], [np.nextafter(np.float32(1), np.float32(2)), 0.0], [np.nextafter(np.float32(-1), np.float32(-2)), 0.0],
                         [np.nan, 0.0], [0.0, np.inf], [-np.inf, np.nan], [1e30, -1e30]], np.float32)
        g = np.concatenate([g, half, half + np.float32(1e-7), half - np.float32(1e-7), edge], 0)
        grid = torch.from_numpy(g).view(1, 1, -1, 2)
        want = F.grid_sample(torch.from_numpy(mask).view(1, 1, h, w), grid, mode="nearest", padding_mode="zeros", align_corners=False).view(-1).numpy()
        with np.errstate(all="ignore"):
            got = orc.warp_mask_nearest(mask, g[:, 0], g[:, 1])
        np.testing.assert_array_equal(got, want, err_msg=f"mask {h}x{w}")


def test_floor_and_first_max_are_torch_clamp_and_max():
    rs = np.random.RandomState(2)
    for k, h, w in [(1, 8, 8), (3, 33, 47), (8, 64, 64), (16, 9, 5)]:
        cert = rs.beta(2, 2, size=(k, h, w)).astype(np.float32)
        cert[rs.rand(k, h, w) < 0.3] = 0.25                              # massive exact ties (floor / cap clamps in real data)
        cert[rs.rand(k, h, w) < 0.01] = np.nan
        cert[rs.rand(k, h, w) < 0.01] = np.inf
        cert[rs.rand(k, h, w) < 0.01] = -np.inf
        th = 0.2
        t = torch.clamp(torch.from_numpy(cert), min=th)
        best_t, idx_t = torch.max(t, dim=0)
        floored = [orc.certainty_prologue(cert[j], np.zeros((h, w, 2), np.float32), th) for j in range(k)]
        for j in range(k):
            np.testing.assert_array_equal(floored[j], t[j].numpy())          # (NaN == NaN under assert_array_equal)
        best, best_k, _agg = orc.aggregate_best(floored, [np.zeros((h, w, 4), np.float32)] * k)
        np.testing.assert_array_equal(best, best_t.numpy())
        np.testing.assert_array_equal(best_k, idx_t.numpy())


def test_the_record_of_the_fuzz_against_upstream_reports_no_mismatch():
    """tests/golden/check_oracle_fuzz.py (development container: it imports upstream's functions from /root/reference) compares every geometry /
    sampling / writer function of the oracle with upstream's own on thousands of seeded adversarial inputs, bit for bit; its record is committed"""
    import json
    import os
    rec = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g9_oracle_fuzz.json")))
    assert rec["mismatches"] == {} and rec["total"] >= 5000
    for fn in ("fundamental_from_world2cam", "sampson_error", "dlt_triangulate_batch", "dlt_triangulate_batch LinAlgError", "reprojection_errors", "cheirality_mask",
               "parallax_mask", "select_samples_with_coverage", "select_samples_with_coverage(no_filter)", "select_samples_with_coverage ValueError",
               "MT19937 position after a selection", "to_uint8_rgb", "write_ply body", "write_points3D_bin",
               "select_cameras_kcenters", "nearest_neighbors", "select_cameras_by_visibility", "K_from_camera", "pose_world2cam R", "pose_world2cam t"):
        assert rec["cases"].get(fn, 0) > 0, fn


def test_the_record_of_the_driver_loop_against_upstreams_reports_no_mismatch():
    """tests/golden/check_pipeline_fuzz.py (development container): upstream's run_dense_pipeline against this package's (CPU twin) on seeded scenes -
    counts, counters, colours, positions, progress sequence, previews, error behaviour; its tally is committed"""
    import json
    import os
    rec = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g10_pipeline_fuzz.json")))
    assert rec["scenes"] >= 20 and rec["points"] > 30000 and rec["max_xyz_rel"] <= 1e-5
    assert rec["lfs_cases"] >= 10             # ... and the GUI entry point dense_init_from_lfs on fake scene nodes (return codes, output files, progress)
    for k in ("count_mismatch", "counter_mismatch", "rgb_mismatch", "xyz_out_of_tol", "err_out_of_tol", "progress_mismatch", "preview_mismatch", "raised_differently",
              "lfs_return_mismatch", "lfs_file_mismatch", "lfs_progress_mismatch"):
        assert rec[k] == 0, k
