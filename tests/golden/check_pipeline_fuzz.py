#!/usr/bin/env python3
"""The driver loop against UPSTREAM'S OWN ``run_dense_pipeline`` on seeded scenes - development container only (it imports /root/reference through
tests/golden/ref_import.py; nothing of it is copied, the GPU box never sees it).

    python tests/golden/check_pipeline_fuzz.py [--scenes 16] [--write]

Golden g4 pins the loop on ONE scene in two configurations.  Here: scenes of 4-9 cameras, 1-3 neighbours, different reference subsets, square and
rectangular grids, filter and no_filter, different matches_per_ref / seeds / viz intervals, mask files (match-sized, other sizes, unreadable), a reference whose matcher call fails - upstream's
``run_dense_pipeline`` (its matcher class replaced by a table of prepared warps, exactly like make_golden.py) against this package's, on the CPU twin
(``backend="host"``: the host build of the kernels' per-cell source + upstream's own host sampling stage).  Compared: per-reference survivor counts
and the processed / matched counters (exact), colours (bit for bit: the same f64 blend), positions (1e-5: the twin's f64 null vector against
LAPACK's f32 SVD) and errors, the sequence of progress percentages and message heads, the intermediate preview files (names, vertex counts, bodies
to the same tolerances).  ``--write`` stores the tally as tests/golden/g10_pipeline_fuzz.json."""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

from ref_import import load_reference  # noqa: E402
import lichtfeld_densification_plugin_amd as lfd  # noqa: E402
from lichtfeld_densification_plugin_amd import synthetic  # noqa: E402
from lichtfeld_densification_plugin_amd.core import pipeline as mine  # noqa: E402


class Table:
    """Duck-typed stand-in for RomaMatcher: the prepared (warp HxWx4, cert HxW) tensors, reference by reference; ``fail_at``: that call raises."""
    sample_thresh = 0.9

    def __init__(self, wm, hm, table, fail_at=-1):
        self.w_resized, self.h_resized, self.table, self.calls, self.fail_at = wm, hm, table, 0, fail_at

    def match_grids_batch(self, imA, imB_list, **_kw):
        i = self.calls
        self.calls += 1
        if i == self.fail_at:
            raise RuntimeError("matcher failed on this reference")
        return [(w.clone(), c.clone()) for (w, c) in self.table[i]]

    def close(self):
        pass


def read_ply(path):
    head, body = open(path, "rb").read().split(b"end_header\n", 1)
    n = int([l for l in head.decode().split("\n") if l.startswith("element vertex")][0].split()[-1])
    rec = np.frombuffer(body, np.dtype([("p", "<f4", 3), ("c", "u1", 3)]), count=n)
    return n, rec["p"].copy(), rec["c"].copy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=16)
    ap.add_argument("--write", action="store_true")
    args = ap.parse_args()
    from PIL import Image
    ns = load_reference()
    P = ns.pipeline
    rs = np.random.RandomState(77)
    tally = {"scenes": 0, "count_mismatch": 0, "counter_mismatch": 0, "rgb_mismatch": 0, "xyz_out_of_tol": 0, "err_out_of_tol": 0, "progress_mismatch": 0,
             "preview_mismatch": 0, "raised_differently": 0, "points": 0, "max_xyz_rel": 0.0}
    for sc in range(args.scenes):
        n_cams = int(rs.randint(4, 10))
        H, W = [(64, 64), (48, 80), (40, 40), (72, 56)][sc % 4]
        wm, hm = W, H
        k = int(rs.randint(1, 4))
        cams = synthetic.ring_cameras(n_cams, seed=100 + sc, arc=0.9)
        refs_local = sorted(int(r) for r in rs.choice(n_cams, size=int(rs.randint(1, min(n_cams, 4) + 1)), replace=False))
        nn_table = ns.selection.nearest_neighbors(np.stack([c.flat_pose() for c in cams]), k)
        no_filter = bool(sc % 3 == 1)
        cfg_kw = dict(nns_per_ref=k, seed=int(rs.randint(0, 1000)), viz_interval=int(rs.choice([0, 1, 2])), pack_workers=int(rs.choice([1, 4])),
                      prefetch_packages=int(rs.choice([1, 2, 8])), no_filter=no_filter,
                      matches_per_ref=int(rs.choice([200, 900, 2500])), reproj_thresh=float(rs.choice([0.8, 1.5])), min_parallax_deg=float(rs.choice([0.5, 0.0])))
        fail_at = 1 if (sc % 5 == 4 and len(refs_local) > 1) else -1
        with tempfile.TemporaryDirectory() as d:
            for i, c in enumerate(cams):
                c.image_path = os.path.join(d, f"im{i:02d}.png")
                Image.fromarray(synthetic.synth_image(hm, wm, 500 + 10 * sc + i).numpy()).save(c.image_path)
                c.mask_path = None
                if sc % 2 == 1 and rs.rand() < 0.6:            # mask files: match size, another size (resized NEAREST), one that cannot be read
                    mh, mw = (hm, wm) if rs.rand() < 0.5 else (int(rs.randint(20, 200)), int(rs.randint(20, 200)))
                    blob = np.full((mh, mw), 255, np.uint8)
                    for _ in range(4):
                        y, x = int(rs.randint(0, mh)), int(rs.randint(0, mw))
                        blob[y:y + mh // 3, x:x + mw // 4] = int(rs.choice([0, 90, 140]))          # grey levels on both sides of the 0.5 threshold
                    c.mask_path = os.path.join(d, f"mask{i:02d}.png")
                    if rs.rand() < 0.12:
                        open(c.mask_path, "wb").write(b"not an image")
                    else:
                        Image.fromarray(blob, mode="L").save(c.mask_path)
            table = []
            for r in refs_local:
                nbrs = [int(n) for n in nn_table[r][:k]]
                s = synthetic.synth_reference(cams, r, nbrs, H, W, wm, hm, noise_px=float(rs.choice([0.2, 0.6])), outlier_frac=0.05, channels=4,
                                              seed=900 + sc, cert_mode=str(rs.choice(["tiefree", "smooth"])))
                table.append([(s.warp[j], s.cert[j]) for j in range(len(nbrs))])
            out = {}
            for who in ("upstream", "mine"):
                fm = Table(wm, hm, table, fail_at)
                progress, viz = [], []
                sub = os.path.join(d, who)
                try:
                    with np.errstate(all="ignore"):
                        if who == "upstream":
                            P.RomaMatcher = lambda device="cpu", mode="outdoor", setting="fast", _fm=fm: _fm
                            P.has_cached_romav2_weights = lambda: True
                            # (upstream's loader hands packages over in COMPLETION order when it has several workers - its run then depends on thread
                            # timing, and a matcher that replays a table cannot follow it: one worker there; this package's prefetcher is ordered, so
                            # its side runs with whatever the case drew and must still give the one-worker result)
                            cfg = ns.config.DensePipelineConfig(output_path=os.path.join(sub, "dense.ply"), roma_setting="fast", **dict(cfg_kw, pack_workers=1))
                            res = P.run_dense_pipeline(cams, refs_local, nn_table, cfg, progress_callback=lambda p, m: progress.append((p, m)),
                                                       on_sequential_viz=lambda path: viz.append(path))
                        else:
                            cfg = lfd.DensePipelineConfig(output_path=os.path.join(sub, "dense.ply"), roma_setting="fast", backend="host", **cfg_kw)
                            res = mine.run_dense_pipeline(cams, refs_local, nn_table, cfg, progress_callback=lambda p, m: progress.append((p, m)),
                                                          on_sequential_viz=lambda path: viz.append(path), matcher=fm)
                    previews = [(os.path.basename(v),) + read_ply(v) for v in viz]
                    out[who] = dict(xyz=res.xyz, rgb=res.rgb, err=res.err, processed=int(res.pairs_processed), matched=int(getattr(res, "pairs_matched", -1)),
                                    progress=[(round(float(p), 6), m.split(" | ")[0]) for p, m in progress], previews=previews, error=None)
                except Exception as exc:                      # noqa: BLE001 - "No points triangulated" and friends: both sides must agree
                    out[who] = dict(error=f"{type(exc).__name__}: {str(exc)[:80]}")
            tally["scenes"] += 1
            u, m = out["upstream"], out["mine"]
            if (u["error"] is None) != (m["error"] is None) or (u["error"] is not None and u["error"] != m["error"]):
                tally["raised_differently"] += 1
                print(f"scene {sc}: raised differently: upstream {u['error']!r}, mine {m['error']!r}")
                continue
            if u["error"] is not None:
                continue
            tally["points"] += int(u["xyz"].shape[0])
            if u["xyz"].shape != m["xyz"].shape:
                tally["count_mismatch"] += 1
                print(f"scene {sc}: {u['xyz'].shape[0]} points upstream, {m['xyz'].shape[0]} here ({cfg_kw})")
                continue
            if (u["processed"], u["matched"]) != (m["processed"], m["matched"]) and u["matched"] >= 0:
                tally["counter_mismatch"] += 1
                print(f"scene {sc}: counters {(u['processed'], u['matched'])} vs {(m['processed'], m['matched'])}")
            if not np.array_equal(u["rgb"], m["rgb"]):
                tally["rgb_mismatch"] += 1
            scale = np.maximum(1.0, np.abs(u["xyz"]).max(axis=1, keepdims=True)) if u["xyz"].size else np.ones((0, 1))
            rel = float((np.abs(u["xyz"] - m["xyz"]) / scale).max()) if u["xyz"].size else 0.0
            tally["max_xyz_rel"] = max(tally["max_xyz_rel"], rel)
            tally["xyz_out_of_tol"] += int(rel > 1e-5)
            tally["err_out_of_tol"] += int(u["err"].size and not np.allclose(m["err"], u["err"], rtol=1e-4, atol=2e-3))      # (no_filter keeps outliers: errors of 1e4 px)
            if u["progress"] != m["progress"]:
                tally["progress_mismatch"] += 1
                print(f"scene {sc}: progress differs\n  upstream {u['progress'][:6]} ...\n  mine     {m['progress'][:6]} ...")
            ok = len(u["previews"]) == len(m["previews"])
            for a, b in zip(u["previews"], m["previews"]):
                ok &= a[0] == b[0] and a[1] == b[1] and np.array_equal(a[3], b[3]) and (a[1] == 0 or float(np.abs(a[2] - b[2]).max()) <= 1e-4)
            tally["preview_mismatch"] += int(not ok)
    # ---- the GUI entry point (densify.py:215-316 upstream): camera nodes -> records -> k-centres references -> neighbours -> pipeline -> point cap ->
    # PLY, against upstream's, on fake scene nodes: nodes without a camera, masks switched off, fractions and counts of references, more neighbours
    # than cameras (clamped), a point cap, a single camera (code 1), a cancellation after a few progress calls (code 2)
    import types as _types
    from lichtfeld_densification_plugin_amd import densify as mine_densify
    D = sys.modules[ns.pipeline.__name__.rsplit(".core.", 1)[0] + ".densify"] if (ns.pipeline.__name__.rsplit(".core.", 1)[0] + ".densify") in sys.modules else None
    if D is None:
        import importlib
        D = importlib.import_module(ns.pipeline.__name__.rsplit(".core.", 1)[0] + ".densify")
    tally.update({"lfs_cases": 0, "lfs_return_mismatch": 0, "lfs_file_mismatch": 0, "lfs_progress_mismatch": 0})
    for sc in range(max(args.scenes // 2, 6)):
        n_cams = 1 if sc == 3 else int(rs.randint(3, 8))
        H = W = 48
        cams = synthetic.ring_cameras(max(n_cams, 2), seed=300 + sc, arc=0.9)[:n_cams]
        cfg_kw = dict(num_refs=float(rs.choice([0.5, 0.8, 1.0, 2.0, 3.0])), nns_per_ref=int(rs.choice([1, 2, 12])), seed=int(rs.randint(0, 99)), viz_interval=0,
                      pack_workers=1, matches_per_ref=int(rs.choice([300, 1500])), max_points=int(rs.choice([0, 0, 700])), use_masks=bool(sc % 2))
        cancel_after = 4 if sc == 5 else -1
        with tempfile.TemporaryDirectory() as d:
            nodes = []
            for i, c in enumerate(cams):
                path = os.path.join(d, f"im{i:02d}.png")
                Image.fromarray(synthetic.synth_image(H, W, 700 + 10 * sc + i).numpy()).save(path)
                mpath = os.path.join(d, f"mask{i:02d}.png")
                blob = np.full((H, W), 255, np.uint8)
                blob[: H // 3, : W // 2] = 0
                Image.fromarray(blob, mode="L").save(mpath)
                nodes.append(_types.SimpleNamespace(has_camera=True, camera_width=c.width, camera_height=c.height, camera_focal_x=float(c.K[0, 0]),
                                                    camera_focal_y=float(c.K[1, 1]), camera_R=np.asarray(c.R), camera_T=np.asarray(c.t).reshape(3),
                                                    camera_uid=int(c.uid), image_path=path, mask_path=mpath, has_mask=bool(i % 2)))
                if i == 1:
                    nodes.append(_types.SimpleNamespace(has_camera=False))
            recs = D.extract_cameras_from_lfs(nodes)
            table = []
            if len(recs) >= 2:
                flat = np.stack([r.flat_pose() for r in recs])
                n_refs = int(round(cfg_kw["num_refs"] * len(recs))) if cfg_kw["num_refs"] <= 1.0 else int(cfg_kw["num_refs"])
                refs_local = ns.selection.select_cameras_kcenters(flat, max(1, n_refs))
                k_eff = max(1, min(cfg_kw["nns_per_ref"], len(recs) - 1))
                nn_table = ns.selection.nearest_neighbors(flat, k_eff)
                by_uid = {int(c.uid): j for j, c in enumerate(cams)}
                for r in refs_local:
                    nbrs = [int(n) for n in nn_table[r][:k_eff]]
                    s = synthetic.synth_reference(cams, by_uid[int(recs[r].uid)], [by_uid[int(recs[n].uid)] for n in nbrs], H, W, W, H, noise_px=0.3, outlier_frac=0.05,
                                                  channels=4, seed=40 + sc, cert_mode="tiefree")
                    table.append([(s.warp[j], s.cert[j]) for j in range(len(nbrs))])
            got = {}
            for who in ("upstream", "mine"):
                fm = Table(W, H, table)
                progress = []
                out_path = os.path.join(d, who, "dense.ply")
                os.makedirs(os.path.dirname(out_path), exist_ok=True)          # (upstream's writer expects the directory to exist)
                calls = {"n": 0}

                def cancel():
                    calls["n"] += 1
                    return cancel_after >= 0 and calls["n"] > cancel_after
                with np.errstate(all="ignore"):
                    if who == "upstream":
                        P.RomaMatcher = lambda device="cpu", mode="outdoor", setting="fast", _fm=fm: _fm
                        P.has_cached_romav2_weights = lambda: True
                        cfg = ns.config.DensePipelineConfig(output_path=out_path, roma_setting="fast", **cfg_kw)
                        ret = D.dense_init_from_lfs(nodes, cfg, progress_callback=lambda p, m: progress.append((round(float(p), 6), m.split(" | ")[0])), cancel_requested=cancel)
                    else:
                        cfg = lfd.DensePipelineConfig(output_path=out_path, roma_setting="fast", backend="host", **cfg_kw)
                        ret = mine_densify.dense_init_from_lfs(nodes, cfg, progress_callback=lambda p, m: progress.append((round(float(p), 6), m.split(" | ")[0])),
                                                               cancel_requested=cancel, matcher=fm)
                ply = read_ply(out_path) if (ret[0] == 0 and os.path.exists(out_path)) else None
                got[who] = (ret[0], (ret[1] or "").replace(os.path.join(d, who), "<out>"), ply, progress)
            tally["lfs_cases"] += 1
            u, m = got["upstream"], got["mine"]
            if u[:2] != m[:2]:
                tally["lfs_return_mismatch"] += 1
                print(f"lfs case {sc}: returned {u[:2]} upstream, {m[:2]} here ({cfg_kw})")
            if (u[2] is None) != (m[2] is None) or (u[2] is not None and not (u[2][0] == m[2][0] and np.array_equal(u[2][2], m[2][2]) and
                                                                                (u[2][0] == 0 or float(np.abs(u[2][1] - m[2][1]).max()) <= 1e-4))):
                tally["lfs_file_mismatch"] += 1
                print(f"lfs case {sc}: output files differ ({None if u[2] is None else u[2][0]} vs {None if m[2] is None else m[2][0]} vertices, {cfg_kw})")
            if u[3] != m[3]:
                tally["lfs_progress_mismatch"] += 1
                print(f"lfs case {sc}: progress differs\n  upstream {u[3]}\n  mine     {m[3]}")
    print(json.dumps(tally, indent=1))
    bad = sum(v for k_, v in tally.items() if k_.endswith(("mismatch", "out_of_tol", "differently")))
    if args.write:
        tally["comment"] = ("tests/golden/check_pipeline_fuzz.py: upstream's run_dense_pipeline (imported from /root/reference in the development container) against "
                            "this package's on the CPU twin, seeded scenes")
        tally["numpy"], tally["torch"] = np.__version__, torch.__version__
        with open(os.path.join(HERE, "g10_pipeline_fuzz.json"), "w") as fh:
            json.dump(tally, fh, indent=1, sort_keys=True)
        print("wrote g10_pipeline_fuzz.json")
    return 1 if bad else 0


if __name__ == "__main__":
    raise SystemExit(main())
