#!/usr/bin/env python3
"""Fuzz the oracle (oracle/densify_oracle.py) against the UPSTREAM functions themselves - development container only, like
make_golden.py: /root/reference does not exist on the GPU box, and nothing of it is copied.

    python tests/golden/check_oracle_fuzz.py [--cases 300] [--write]

The golden fixtures pin the oracle on a handful of upstream-captured cases; this calls upstream's own functions
(/root/reference/core/geometry.py, core/sampling.py, core/writers.py, core/image_utils.py through tests/golden/ref_import.py) and the oracle's
restatements on thousands of seeded inputs - clean, noisy, degenerate (coincident pixels, points at infinity, points behind a camera, NaN /
Inf coordinates, tied and masked-out certainties, empty selections) - in the SAME process (same NumPy, LAPACK, torch), and requires BIT
equality (NaN == NaN) of every output, and of the legacy MT19937 stream position after a selection.
``--write`` stores the per-function case counts and a digest of upstream's outputs as tests/golden/g9_oracle_fuzz.json (a record of what
was checked here; ``tests/test_oracle_golden.py`` checks that the record exists and reports no mismatch)."""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

from ref_import import load_reference  # noqa: E402
from lichtfeld_densification_plugin_amd import synthetic  # noqa: E402
from oracle import densify_oracle as orc  # noqa: E402


def same(a, b) -> bool:
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and bool(np.array_equal(a, b, equal_nan=a.dtype.kind == "f"))


class Tally:
    def __init__(self):
        self.cases, self.bad, self.h = {}, {}, hashlib.sha256()

    def check(self, name, ref, got):
        self.cases[name] = self.cases.get(name, 0) + 1
        self.h.update(np.ascontiguousarray(np.asarray(ref)).tobytes())
        if not same(ref, got):
            self.bad[name] = self.bad.get(name, 0) + 1
            if self.bad[name] <= 3:
                r, g = np.asarray(ref), np.asarray(got)
                print(f"MISMATCH {name}: shapes {r.shape} {g.shape} dtypes {r.dtype} {g.dtype}", flush=True)


def correspondences(rs, cams, n):
    """(cam a, cam b, uv1, uv2) float32: projections of random scene points + noise, with a share of adversarial rows"""
    ia, ib = rs.choice(len(cams), 2, replace=False)
    ca, cb = cams[ia], cams[ib]
    X = np.concatenate([rs.uniform(-2, 2, (n, 2)), rs.uniform(-0.5, 0.8, (n, 1)), np.ones((n, 1))], 1)
    pa, pb = X @ np.asarray(ca.P, np.float64).T, X @ np.asarray(cb.P, np.float64).T
    uv1 = (pa[:, :2] / pa[:, 2:3]).astype(np.float32)
    uv2 = (pb[:, :2] / pb[:, 2:3]).astype(np.float32) + rs.normal(0, rs.choice([0.0, 0.3, 3.0]), (n, 2)).astype(np.float32)
    k = max(1, n // 10)
    uv2[:k] = rs.uniform(-500, 2500, (k, 2)).astype(np.float32)              # gross outliers
    uv2[k:k + 2] = uv1[k:k + 2]                                               # coincident pixels
    if n > 8:
        uv1[-1] = [np.nan, 1.0]; uv2[-2] = [np.inf, -np.inf]; uv2[-3] = [1e30, 1e-30]
    return ca, cb, uv1, uv2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--write", action="store_true")
    args = ap.parse_args()
    ns = load_reference()
    G, S = ns.geometry, ns.sampling
    cams = synthetic.ring_cameras(40, seed=3)
    t = Tally()
    rs = np.random.RandomState(2024)
    with np.errstate(all="ignore"):
        for c in range(args.cases):
            n = int(rs.choice([1, 2, 7, 64, 500]))
            ca, cb, uv1, uv2 = correspondences(rs, cams, n)
            P1, P2 = np.asarray(ca.P, np.float32), np.asarray(cb.P, np.float32)
            F_ref = G.fundamental_from_world2cam(ca.K, ca.R, ca.t, cb.K, cb.R, cb.t)
            F_orc = orc.fundamental_matrix(ca.K, ca.R, ca.t, cb.K, cb.R, cb.t)
            t.check("fundamental_from_world2cam", F_ref, F_orc)
            t.check("sampson_error", G.sampson_error(F_ref, uv1, uv2), orc.sampson_error(F_ref, uv1, uv2))
            def guarded(fn):
                try:
                    return fn(P1, P2, uv1, uv2), None
                except np.linalg.LinAlgError as exc:          # (non-finite rows: LAPACK gives up on the whole batch, upstream drops the reference)
                    return None, type(exc).__name__ + ": " + str(exc)
            (X_ref, e_ref), (X_orc, e_orc) = guarded(G.dlt_triangulate_batch), guarded(orc.dlt_triangulate)
            if X_ref is None or X_orc is None:
                t.check("dlt_triangulate_batch LinAlgError", np.frombuffer(str(e_ref).encode(), np.uint8), np.frombuffer(str(e_orc).encode(), np.uint8))
                finite = np.isfinite(uv1).all(1) & np.isfinite(uv2).all(1)
                uv1, uv2 = uv1[finite], uv2[finite]
                X_ref = G.dlt_triangulate_batch(P1, P2, uv1, uv2)
                X_orc = orc.dlt_triangulate(P1, P2, uv1, uv2)
            t.check("dlt_triangulate_batch", X_ref, X_orc)
            for P, uv in ((P1, uv1), (P2, uv2)):
                t.check("reprojection_errors", G.reprojection_errors(P, X_ref, uv), orc.reprojection_error(P, X_ref, uv))
                t.check("cheirality_mask", G.cheirality_mask(P, X_ref), orc.depth_positive(P, X_ref))
            for deg in (0.5, 0.0, 3.0):
                t.check("parallax_mask", G.parallax_mask(ca.C, cb.C, X_ref.copy(), deg), orc.parallax_ok(ca.C, cb.C, X_ref.copy(), deg))
        # ---- coverage sampling: the legacy global stream upstream consumes against a RandomState of the same seed --------------------------
        for c in range(max(args.cases // 3, 20)):
            h, w = [(64, 64), (48, 80), (33, 47), (96, 96), (17, 300), (5, 5), (2, 9)][c % 7]
            M = int(rs.choice([50, 500, 3000, 20000]))
            mode = c % 5
            cert = rs.beta(2, 2, (h, w)).astype(np.float32)
            if mode == 1:
                cert[rs.rand(h, w) < 0.4] = 0.2                                # massive ties at the floor
            if mode == 2:
                cert[rs.rand(h, w) < 0.6] = 0.0                                # masked-out cells
            if mode == 3:
                cert[:] = 0.0                                                  # nothing to draw from
            if mode == 4:
                cert[rs.rand(h, w) < 0.7] = 0.95                               # the cap
            for no_filter in (False, True):
                seed = int(rs.randint(0, 2 ** 31 - 1))
                np.random.seed(seed)
                try:
                    ref, ref_err = S.select_samples_with_coverage(torch.from_numpy(cert), M, cap=0.9, border=2, tiles=24, no_filter=no_filter), None
                except ValueError as exc:
                    ref, ref_err = None, str(exc)
                pos_ref = int(np.random.get_state()[2])
                r2 = np.random.RandomState(seed)
                try:
                    got, got_err = orc.select_samples(cert, M, cap=0.9, border=2, tiles=24, no_filter=no_filter, rng=r2), None
                except ValueError as exc:
                    got, got_err = None, str(exc)
                name = "select_samples_with_coverage" + ("(no_filter)" if no_filter else "")
                if ref is None or got is None:
                    t.check(name + " ValueError", np.frombuffer(str(ref_err).encode(), np.uint8), np.frombuffer(str(got_err).encode(), np.uint8))
                else:
                    t.check(name, ref, got)
                t.check("MT19937 position after a selection", np.int64(pos_ref), np.int64(r2.get_state()[2]))
        # ---- writers ------------------------------------------------------------------------------------------------------------------------------
        import tempfile
        for c in range(20):
            n = int(rs.choice([0, 1, 17, 5000]))
            xyz = rs.normal(0, 5, (n, 3)).astype(np.float32)
            rgb = rs.uniform(-0.2, 1.2, (n, 3)).astype(np.float32)
            err = rs.uniform(0, 2, (n,)).astype(np.float32)
            u8_ref = ns.image_utils.to_uint8_rgb(rgb) if hasattr(ns.image_utils, "to_uint8_rgb") else None
            if u8_ref is not None:
                t.check("to_uint8_rgb", u8_ref, orc.to_uint8_rgb(rgb))
            u8 = orc.to_uint8_rgb(rgb)
            with tempfile.TemporaryDirectory() as d:
                if n:
                    p = os.path.join(d, "a.ply")
                    ns.writers.write_ply(p, xyz, u8)
                    body = open(p, "rb").read().split(b"end_header\n", 1)[1]
                    t.check("write_ply body", np.frombuffer(body, np.uint8), np.frombuffer(orc.ply_bytes(xyz, u8).split(b"end_header\n", 1)[1], np.uint8))
                    p3 = os.path.join(d, "points3D.bin")
                    ns.writers.write_points3D_bin(p3, xyz, u8, err) if "err" in ns.writers.write_points3D_bin.__code__.co_varnames else ns.writers.write_points3D_bin(p3, xyz, u8)
                    t.check("write_points3D_bin", np.frombuffer(open(p3, "rb").read(), np.uint8),
                            np.frombuffer(orc.points3d_bin_bytes(xyz, u8, err if "err" in ns.writers.write_points3D_bin.__code__.co_varnames else None), np.uint8))
        # ---- camera selection (product host code, lichtfeld-densification-plugin_amd/core/selection.py, against upstream core/selection.py) ------
        from lichtfeld_densification_plugin_amd.core import selection as sel_mine
        import types as _types
        for c in range(max(args.cases // 3, 30)):
            n = int(rs.choice([1, 2, 3, 17, 60, 185]))
            poses = rs.normal(0, rs.choice([1.0, 5.0]), (n, 12)).astype(np.float32)
            if c % 4 == 1 and n > 3:
                poses[1] = poses[0]; poses[3] = poses[2]                       # coincident cameras: distance ties
            if c % 4 == 2:
                poses[:, 5] = 1.0                                              # a constant column (sigma -> 1e-8)
            for k in (1, 3, 8, n, n + 5):
                t.check("select_cameras_kcenters", np.asarray(ns.selection.select_cameras_kcenters(poses, k), np.int64),
                        np.asarray(sel_mine.select_cameras_kcenters(poses, k), np.int64))
                t.check("nearest_neighbors", ns.selection.nearest_neighbors(poses, k), sel_mine.nearest_neighbors(poses, k))
            # visibility: a reconstruction as far as the function looks at one (images with points2D that may or may not have a 3-D point)
            n_img, n_pts = int(rs.randint(1, 30)), int(rs.randint(1, 200))

            def p2d(pid):
                return _types.SimpleNamespace(point3D_id=int(pid), has_point3D=lambda pid=pid: pid != -1 and pid != 2 ** 64 - 1)
            images = {int(iid): _types.SimpleNamespace(image_id=int(iid), points2D=[p2d(pid) for pid in rs.choice(np.concatenate([[-1, -1], np.arange(n_pts)]),
                                                                                                     size=int(rs.randint(0, 80)))])
                      for iid in rs.choice(1000, n_img, replace=False)}
            rec = _types.SimpleNamespace(images=images, points3D={i: None for i in range(n_pts)})
            for k in (1, 4, n_img, n_img + 3):
                t.check("select_cameras_by_visibility", np.asarray(ns.selection.select_cameras_by_visibility(rec, k), np.int64),
                        np.asarray(sel_mine.select_cameras_by_visibility(rec, k), np.int64))
        # ---- COLMAP cameras / images -> K, R, t (product host code densify.py against upstream core/geometry.py:10-50): every COLMAP model name
        from lichtfeld_densification_plugin_amd import densify as dens_mine
        models = {"SIMPLE_PINHOLE": 3, "PINHOLE": 4, "SIMPLE_RADIAL": 4, "RADIAL": 5, "OPENCV": 8, "OPENCV_FISHEYE": 8, "FULL_OPENCV": 12, "FOV": 5,
                  "SIMPLE_RADIAL_FISHEYE": 4, "RADIAL_FISHEYE": 5, "THIN_PRISM_FISHEYE": 12, "RAD_TAN_THIN_PRISM_FISHEYE": 16, "SOMETHING_ELSE": 1, "simple_pinhole": 3}
        for name, npar in models.items():
            for rep in range(3):
                cam = _types.SimpleNamespace(model=_types.SimpleNamespace(name=name), params=rs.uniform(100, 2000, npar), width=int(rs.randint(100, 4000)), height=int(rs.randint(100, 4000)))
                t.check("K_from_camera", G.K_from_camera(cam), dens_mine.K_from_camera(cam))
        for rep in range(30):
            q = rs.normal(size=4); q /= np.linalg.norm(q)
            w_, x_, y_, z_ = q
            Rm = np.array([[1 - 2 * (y_ * y_ + z_ * z_), 2 * (x_ * y_ - z_ * w_), 2 * (x_ * z_ + y_ * w_)], [2 * (x_ * y_ + z_ * w_), 1 - 2 * (x_ * x_ + z_ * z_), 2 * (y_ * z_ - x_ * w_)],
                           [2 * (x_ * z_ - y_ * w_), 2 * (y_ * z_ + x_ * w_), 1 - 2 * (x_ * x_ + y_ * y_)]])
            tv = rs.normal(size=3)
            rigid = _types.SimpleNamespace(rotation=_types.SimpleNamespace(matrix=lambda Rm=Rm: Rm), translation=tv)
            for im in (_types.SimpleNamespace(cam_from_world=rigid), _types.SimpleNamespace(cam_from_world=lambda rigid=rigid: rigid),
                       type("OldImage", (), {"qvec": _types.SimpleNamespace(to_rotation_matrix=lambda Rm=Rm: Rm.astype(np.float32)), "tvec": tv})()):
                (R_u, t_u), (R_m, t_m) = G.pose_world2cam(im), dens_mine.pose_world2cam(im)
                t.check("pose_world2cam R", R_u, R_m)
                t.check("pose_world2cam t", t_u, t_m)
                t.check("P_from_KRt / cam_center_world", np.concatenate([G.P_from_KRt(np.eye(3, dtype=np.float32), R_u, t_u).reshape(-1), G.cam_center_world(R_u, t_u)]),
                        np.concatenate([(np.eye(3, dtype=np.float32) @ np.concatenate([R_m, t_m], axis=1)).reshape(-1), (-R_m.T @ t_m).reshape(3)]))
    total, bad = sum(t.cases.values()), sum(t.bad.values())
    for k in sorted(t.cases):
        print(f"{k:48s} {t.cases[k]:6d} cases  {t.bad.get(k, 0)} mismatches")
    print(f"total {total} comparisons, {bad} mismatches; digest of upstream's outputs {t.h.hexdigest()[:16]}")
    if args.write:
        rec = {"comment": "tests/golden/check_oracle_fuzz.py: the oracle's functions against upstream's own (imported from /root/reference in the development "
                          "container), bit for bit on seeded adversarial inputs", "cases": t.cases, "mismatches": t.bad, "total": total,
               "digest_of_upstream_outputs": t.h.hexdigest(), "numpy": np.__version__, "torch": torch.__version__, "seeded_cases": args.cases}
        with open(os.path.join(HERE, "g9_oracle_fuzz.json"), "w") as fh:
            json.dump(rec, fh, indent=1, sort_keys=True)
        print("wrote g9_oracle_fuzz.json")
    return 1 if bad else 0


if __name__ == "__main__":
    raise SystemExit(main())
