#!/usr/bin/env python3
"""Generate the committed golden vectors by running the UPSTREAM code (development container only).

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

Imports the upstream plugin modules from /root/reference through ``ref_import.load_reference`` (stub
host modules, nothing copied), drives them with seeded synthetic inputs and stores inputs + outputs.
The fixtures are data only.  NumPy / torch versions and the CPU capability used are recorded because
LAPACK rounding, ``argsort`` tie order and torch's f32 sum order may differ between builds.
"""
from __future__ import annotations

import hashlib
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

ns = load_reference()
G = ns.geometry
VERSIONS = json.dumps({"numpy": np.__version__, "torch": torch.__version__,
                       "cpu_capability": torch.backends.cpu.get_cpu_capability()})


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, versions=np.array(VERSIONS), **arrays)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


def cam_arrays(cams):
    return dict(
        cam_K=np.stack([c.K for c in cams]).astype(np.float32), cam_R=np.stack([c.R for c in cams]).astype(np.float32),
        cam_t=np.stack([c.t.reshape(3) for c in cams]).astype(np.float32), cam_P=np.stack([c.P for c in cams]).astype(np.float32),
        cam_C=np.stack([c.C for c in cams]).astype(np.float32), cam_wh=np.array([[c.width, c.height] for c in cams], np.int32),
        cam_uid=np.array([c.uid for c in cams], np.int64))


# ---------------------------------------------------------------------------------------------
# G1: geometry known-answer vectors
# ---------------------------------------------------------------------------------------------
def g1_geometry():
    cams = synthetic.ring_cameras(24, seed=3)
    # near-degenerate partner: almost the same centre as camera 0 (tiny baseline)
    c0 = cams[0]
    tiny = lfd.CameraRecord.from_krt(99, c0.K, c0.R, c0.t.reshape(3) + np.array([1e-3, 0, 0], np.float32),
                                     c0.width, c0.height)
    pairs = [(cams[0], cams[1]), (cams[0], cams[3]), (cams[5], cams[4]), (cams[0], tiny), (cams[2], cams[14])]
    rng = np.random.RandomState(11)
    out = {}
    for pi, (ca, cb) in enumerate(pairs):
        n = 257
        # true 3-D points in front of camera a, projected into both, plus pixel noise
        Xw = np.stack([rng.uniform(-1.5, 1.5, n), rng.uniform(-1.5, 1.5, n), rng.uniform(-0.2, 0.6, n)], 1)
        def proj(c, X):
            p = (c.P.astype(np.float64) @ np.concatenate([X, np.ones((len(X), 1))], 1).T).T
            return (p[:, :2] / p[:, 2:3])
        uv1 = (proj(ca, Xw) + rng.normal(0, 0.4, (n, 2))).astype(np.float32)
        uv2 = (proj(cb, Xw) + rng.normal(0, 0.4, (n, 2))).astype(np.float32)
        uv2[::17] += rng.normal(0, 30.0, uv2[::17].shape).astype(np.float32)     # gross outliers
        F = G.fundamental_from_world2cam(ca.K, ca.R, ca.t, cb.K, cb.R, cb.t)
        se = G.sampson_error(F, uv1, uv2)
        with np.errstate(all="ignore"):
            X = G.dlt_triangulate_batch(ca.P, cb.P, uv1, uv2)
            X1 = G.dlt_triangulate_batch(ca.P, cb.P, uv1[:1], uv2[:1])
            e1 = G.reprojection_errors(ca.P, X, uv1)
            e2 = G.reprojection_errors(cb.P, X, uv2)
            ch1 = G.cheirality_mask(ca.P, X)
            ch2 = G.cheirality_mask(cb.P, X)
            par = G.parallax_mask(ca.C, cb.C, X, min_deg=0.5)
        pre = f"p{pi}_"
        out.update({pre + "camA": np.int64(pi), pre + "uv1": uv1, pre + "uv2": uv2, pre + "F": F, pre + "sampson": se,
                    pre + "X": X, pre + "X_single": X1, pre + "err1": e1, pre + "err2": e2,
                    pre + "cheir1": ch1, pre + "cheir2": ch2, pre + "parallax": par})
        out.update({pre + k: v for k, v in cam_arrays([ca, cb]).items()})
    out["n_pairs"] = np.int64(len(pairs))
    # points behind the camera / w ~ 0
    ca, cb = cams[0], cams[1]
    uv1 = np.array([[100.0, 100.0], [648.5, 420.0], [5000.0, -3000.0]], np.float32)
    uv2 = np.array([[1200.0, 800.0], [648.5, 420.0], [-4000.0, 9000.0]], np.float32)
    with np.errstate(all="ignore"):
        X = G.dlt_triangulate_batch(ca.P, cb.P, uv1, uv2)
        out["odd_uv1"], out["odd_uv2"], out["odd_X"] = uv1, uv2, X
        out["odd_err1"] = G.reprojection_errors(ca.P, X, uv1)
        out["odd_cheir1"] = G.cheirality_mask(ca.P, X)
        out["odd_cheir2"] = G.cheirality_mask(cb.P, X)
    save("g1_geometry.npz", **out)


# ---------------------------------------------------------------------------------------------
# G2: select_samples_with_coverage
# ---------------------------------------------------------------------------------------------
def tiefree_cert(h, w, seed):
    """Distinct f32 values strictly inside (0.2, 0.9): no ties under the floor/cap clamps."""
    rs = np.random.RandomState(seed)
    perm = rs.permutation(h * w).astype(np.float64)
    return (0.2 + 0.7 * (perm + 0.5) / (h * w)).astype(np.float32).reshape(h, w)


def g2_selection():
    out = {}
    cases = [(64, 64, 1000, 0), (64, 64, 3000, 1), (80, 96, 1500, 2), (320, 320, 10000, 0), (512, 512, 10000, 0)]
    meta = []
    for ci, (h, w, M, seed) in enumerate(cases):
        cert = tiefree_cert(h, w, 100 + ci)
        for no_filter in (False, True):
            np.random.seed(seed)
            sel = ns.sampling.select_samples_with_coverage(torch.from_numpy(cert.copy()), M, cap=0.9, border=2,
                                                           tiles=24, no_filter=no_filter)
            pos = int(np.random.get_state()[2])
            nxt = np.random.random_sample(2)          # the next two doubles of the stream
            key = f"c{ci}_{'nf' if no_filter else 'f'}_"
            out[key + "sel"] = np.asarray(sel, np.int64)
            out[key + "mt_pos"] = np.int64(pos)
            out[key + "next_doubles"] = nxt
        if h * w <= 96 * 96:
            out[f"c{ci}_cert"] = cert
        out[f"c{ci}_cert_sha256"] = np.array(hashlib.sha256(cert.tobytes()).hexdigest())
        yy, xx = np.mgrid[0:h, 0:w]
        inside = (xx >= 2) & (xx <= w - 3) & (yy >= 2) & (yy <= h - 3)
        wts = torch.clamp(torch.from_numpy(cert), max=0.9) * torch.from_numpy(inside).float()
        out[f"c{ci}_s"] = np.float32(wts.reshape(-1).sum().item())     # the normaliser upstream's torch sum produced here
        meta.append([h, w, M, seed, 100 + ci])
    out["cases"] = np.array(meta, np.int64)
    # a case with massive ties (floor + cap) - only order-insensitive facts are pinned for it
    rs = np.random.RandomState(5)
    cert = rs.beta(2, 2, size=(64, 64)).astype(np.float32)
    cert = np.maximum(cert, np.float32(0.2))
    np.random.seed(3)
    sel = ns.sampling.select_samples_with_coverage(torch.from_numpy(cert.copy()), 1200, cap=0.9, border=2, tiles=24,
                                                   no_filter=False)
    out["ties_cert"], out["ties_sel"] = cert, np.asarray(sel, np.int64)
    # more draws requested than non-zero weights: upstream's np.random.choice raises ValueError
    # (the pipeline logs it per reference and skips the reference, core/pipeline.py:874-879)
    try:
        np.random.seed(0)
        ns.sampling.select_samples_with_coverage(torch.from_numpy(tiefree_cert(64, 64, 1)), 10000)
        out["too_many_raises"] = np.int64(0)
    except ValueError as exc:
        out["too_many_raises"] = np.int64(1)
        out["too_many_msg"] = np.array(str(exc))
    # all-zero weights -> empty
    z = ns.sampling.select_samples_with_coverage(torch.zeros(16, 16), 100)
    out["zero_sel"] = np.asarray(z, np.int64)
    save("g2_selection.npz", **out)


# ---------------------------------------------------------------------------------------------
# G3: _collect_reference_matches (prologue) + _triangulate_ref
# ---------------------------------------------------------------------------------------------
class FakeMatcher:
    """Duck-typed stand-in for RomaMatcher: returns the prepared (warp HxWx4, cert HxW) tensors."""
    sample_thresh = 0.9

    def __init__(self, w_match, h_match, table):
        self.w_resized, self.h_resized = w_match, h_match
        self.table = table
        self.calls = 0

    def match_grids_batch(self, imA, imB_list):
        res = self.table[self.calls]
        self.calls += 1
        assert len(res) == len(imB_list)
        return [(w.clone(), c.clone()) for (w, c) in res]

    def close(self):
        pass


def run_upstream_reference(cams, ref, nbrs, sref, cfg, mask_a=None, masks_b=None, seed=0):
    """Drive upstream _collect_reference_matches + _triangulate_ref on one reference."""
    P = ns.pipeline
    lookup = P._build_camera_lookup(cams)
    ids = [c.uid for c in cams]
    packed = P._PackedReferenceBatch(
        ref_id=ids[ref], ref_path=cams[ref].image_path, imA_np=sref.image.numpy(), maskA_np=mask_a,
        wA_cam=cams[ref].width, hA_cam=cams[ref].height, nn_ids=[ids[n] for n in nbrs],
        nn_masks=list(masks_b) if masks_b is not None else [None] * len(nbrs),
        nn_arrays=[np.zeros_like(sref.image.numpy()) for _ in nbrs])
    table = [[(sref.warp[j], sref.cert[j]) for j in range(len(nbrs))]]
    fm = FakeMatcher(sref.w_match, sref.h_match, table)
    matched, _ = P._collect_reference_matches(packed, fm, cfg, 0, None)
    ctx = P._TriangulationContext(cameras=lookup, config=cfg, matcher_sample_cap=fm.sample_thresh,
                                  w_match=sref.w_match, h_match=sref.h_match)
    captured = {}
    orig = P.select_samples_with_coverage

    def spy(*a, **kw):
        r = orig(*a, **kw)
        captured["sel"] = np.asarray(r, np.int64).copy()
        return r

    P.select_samples_with_coverage = spy
    try:
        np.random.seed(seed)
        with np.errstate(all="ignore"):
            tri = P._triangulate_ref(matched, ctx, collect_debug_matches=True)
    finally:
        P.select_samples_with_coverage = orig
    post_cert = np.stack([c.numpy() for c in matched.cert_list_cpu])
    return tri, captured.get("sel", np.zeros(0, np.int64)), post_cert


def g3_triangulate():
    cams = synthetic.ring_cameras(40, seed=1)
    Cfg = ns.config.DensePipelineConfig
    cases = [
        # name, H, W, w_match, h_match, ref, k, cfg kwargs, synth kwargs, masks
        ("a_filter_k3", 64, 64, 64, 64, 0, 3, dict(matches_per_ref=1500), dict(noise_px=0.35, outlier_frac=0.08), False),
        ("b_nofilter_k1", 48, 48, 48, 48, 3, 1, dict(matches_per_ref=600, no_filter=True), dict(noise_px=0.3), False),
        ("c_rect_k3", 56, 72, 72, 56, 5, 3, dict(matches_per_ref=1200, reproj_thresh=1.5), dict(noise_px=0.6, outlier_frac=0.05), False),
        ("d_hires_k2", 96, 96, 64, 64, 7, 2, dict(matches_per_ref=2000), dict(noise_px=0.3), False),
        ("e_masks_k3", 64, 64, 64, 64, 9, 3, dict(matches_per_ref=1500), dict(noise_px=0.3, outlier_frac=0.03), True),
        ("f_nosampson_k4", 64, 64, 64, 64, 11, 4, dict(matches_per_ref=1500, sampson_thresh=0.0, min_parallax_deg=0.0),
         dict(noise_px=0.5, outlier_frac=0.05), False),
    ]
    out = cam_arrays(cams)
    names = []
    for (name, H, W, wm, hm, ref, k, ckw, skw, use_masks) in cases:
        nbrs = synthetic.ring_neighbours(len(cams), ref, k)
        if name == "a_filter_k3":
            nbrs = [nbrs[0], nbrs[1], ref + 12]            # one wide-baseline neighbour
        sref = synthetic.synth_reference(cams, ref, nbrs, H, W, wm, hm, channels=4, seed=21, cert_mode="tiefree",
                                         **skw)
        cfg = Cfg(output_path="/tmp/x.ply", **ckw)
        mask_a = masks_b = None
        if use_masks:
            yy, xx = np.mgrid[0:hm, 0:wm]
            mask_a = ((xx - 20) ** 2 + (yy - 30) ** 2 > 12 ** 2).astype(np.uint8)
            masks_b = [((xx + 2 * j) % 16 > 2).astype(np.uint8) if j != 1 else None for j in range(k)]
        tri, sel, post_cert = run_upstream_reference(cams, ref, nbrs, sref, cfg, mask_a, masks_b, seed=7)
        ids = [c.uid for c in cams]
        pre = name + "_"
        out[pre + "warp"] = sref.warp.numpy()
        out[pre + "cert"] = sref.cert.numpy()
        out[pre + "image"] = sref.image.numpy()
        out[pre + "post_cert"] = post_cert
        out[pre + "sel"] = sel
        out[pre + "dims"] = np.array([H, W, wm, hm, ref, k], np.int64)
        out[pre + "nbrs"] = np.array(nbrs, np.int64)
        out[pre + "cfg"] = np.array(json.dumps({f: getattr(cfg, f) for f in (
            "matches_per_ref", "certainty_thresh", "reproj_thresh", "sampson_thresh", "min_parallax_deg", "no_filter")}))
        if mask_a is not None:
            out[pre + "mask_a"] = mask_a
            for j, m in enumerate(masks_b):
                if m is not None:
                    out[pre + f"mask_b{j}"] = m
        if tri is None:
            out[pre + "none"] = np.int64(1)
        else:
            out[pre + "xyz"], out[pre + "rgb"], out[pre + "err"] = tri.xyz, tri.rgb, tri.err
            order = [ids.index(n) for n in tri.debug_matches_by_nbr.keys()]
            out[pre + "seg_nbr_cam"] = np.array(order, np.int64)
            out[pre + "seg_count"] = np.array([v.shape[0] for v in tri.debug_matches_by_nbr.values()], np.int64)
            out[pre + "dbg_matches"] = np.concatenate(list(tri.debug_matches_by_nbr.values()), 0)
            out[pre + "dbg_cert"] = np.concatenate(list(tri.debug_cert_by_nbr.values()), 0)
            print(f"  {name}: sel={sel.size} survivors={tri.xyz.shape[0]} segs={out[pre + 'seg_count'].tolist()}")
        names.append(name)
    out["names"] = np.array(names)
    save("g3_triangulate.npz", **out)


# ---------------------------------------------------------------------------------------------
# G5: writers
# ---------------------------------------------------------------------------------------------
def g5_writers():
    rs = np.random.RandomState(2)
    xyz = rs.normal(0, 3, (5, 3)).astype(np.float32)
    rgb = np.array([[0.0, 1.0, 0.5], [0.0019607844, 0.49803922, 0.5019608], [2.5 / 255, 3.5 / 255, 254.5 / 255],
                    [1.2, -0.3, 0.999], [0.25, 0.75, 0.1]], np.float32)
    err = rs.uniform(0, 1, 5).astype(np.float32)
    u8 = ns.image_utils.to_uint8_rgb(rgb)
    with tempfile.TemporaryDirectory() as d:
        p1, p2 = os.path.join(d, "a.ply"), os.path.join(d, "a.bin")
        ns.writers.write_ply(p1, xyz, u8)
        ns.writers.write_points3D_bin(p2, xyz, u8, err)
        ply = np.frombuffer(open(p1, "rb").read(), np.uint8)
        pbin = np.frombuffer(open(p2, "rb").read(), np.uint8)
    save("g5_writers.npz", xyz=xyz, rgb=rgb, err=err, rgb_u8=u8, ply=ply, points3d_bin=pbin)


# ---------------------------------------------------------------------------------------------
# G4: run_dense_pipeline end to end with a fake matcher (3 references, pack_workers=1)
# ---------------------------------------------------------------------------------------------
def g4_pipeline():
    from PIL import Image
    P = ns.pipeline
    n_cams, wm, hm, H, W = 6, 64, 64, 64, 64
    cams = synthetic.ring_cameras(n_cams, seed=4, arc=0.9)          # partial arc: every pair overlaps
    refs_local = [0, 2, 5]
    nn_table = ns.selection.nearest_neighbors(np.stack([c.flat_pose() for c in cams]), 2)
    out = cam_arrays(cams)
    with tempfile.TemporaryDirectory() as d:
        images = []
        for i, c in enumerate(cams):
            img = synthetic.synth_image(hm, wm, 300 + i).numpy()
            c.image_path = os.path.join(d, f"im{i:02d}.png")
            Image.fromarray(img).save(c.image_path)
            images.append(img)
        table, sref_store = [], {}
        for r in refs_local:
            nbrs = [int(n) for n in nn_table[r][:2]]
            s = synthetic.synth_reference(cams, r, nbrs, H, W, wm, hm, noise_px=0.4, outlier_frac=0.05, channels=4,
                                          seed=31, cert_mode="tiefree")
            table.append([(s.warp[j], s.cert[j]) for j in range(len(nbrs))])
            sref_store[r] = s
        for mode_name, cfg_kw in (("filter", dict(matches_per_ref=1200)), ("nofilter", dict(matches_per_ref=500, no_filter=True))):
            fm = FakeMatcher(wm, hm, table)
            P.RomaMatcher = lambda device="cpu", mode="outdoor", setting="fast", _fm=fm: _fm
            P.has_cached_romav2_weights = lambda: True
            cfg = ns.config.DensePipelineConfig(output_path=os.path.join(d, "out", "dense.ply"), roma_setting="fast",
                                                nns_per_ref=2, seed=5, viz_interval=2, pack_workers=1, **cfg_kw)
            progress, viz = [], []
            with np.errstate(all="ignore"):
                res = P.run_dense_pipeline(cams, refs_local, nn_table, cfg, progress_callback=lambda p, m: progress.append((p, m)),
                                           on_sequential_viz=lambda path: viz.append(os.path.basename(path)))
            pre = mode_name + "_"
            out[pre + "xyz"], out[pre + "rgb"], out[pre + "err"] = res.xyz, res.rgb, res.err
            out[pre + "pairs_processed"] = np.int64(res.pairs_processed)
            out[pre + "progress_pct"] = np.array([p for p, _ in progress], np.float64)
            out[pre + "progress_msg"] = np.array([m.split(" | ")[0] for _, m in progress])
            out[pre + "viz_files"] = np.array(viz)
            out[pre + "cfg"] = np.array(json.dumps(cfg_kw))
            print(f"  g4 {mode_name}: {res.xyz.shape[0]} points, refs={res.pairs_processed}, viz={viz}")
        out["images"] = np.stack(images)
        out["refs_local"] = np.array(refs_local, np.int64)
        out["nn_table"] = np.asarray(nn_table, np.int64)
        for r in refs_local:
            out[f"ref{r}_warp"] = sref_store[r].warp.numpy()
            out[f"ref{r}_cert"] = sref_store[r].cert.numpy()
    save("g4_pipeline.npz", **out)


# ---------------------------------------------------------------------------------------------
# G6: camera selection / neighbour table
# ---------------------------------------------------------------------------------------------
def g6_selection_tables():
    cams = synthetic.ring_cameras(185, seed=0)
    flat = np.stack([c.flat_pose() for c in cams])
    out = {"flat_poses": flat}
    for k in (3, 4, 8):
        out[f"nn_k{k}"] = np.asarray(ns.selection.nearest_neighbors(flat, k), np.int64)
    for k in (1, 56, 148):
        out[f"kcenters_{k}"] = np.asarray(ns.selection.select_cameras_kcenters(flat, k), np.int64)
    out["nn_two"] = np.asarray(ns.selection.nearest_neighbors(flat[:2], 5), np.int64)
    save("g6_selection_tables.npz", **out)


# ---------------------------------------------------------------------------------------------
# G7: image preparation (core/image_utils.py:40-91, core/pipeline.py:163-171): decoded arrays -> what upstream feeds the matcher
# ---------------------------------------------------------------------------------------------
def g7_image_prep():
    import tempfile
    from PIL import Image
    import PIL
    rs = np.random.RandomState(7)
    out = {"pillow_version": np.array(PIL.__version__)}
    cases = [("down", 61, 97, 64, 48), ("downup", 84, 130, 51, 96), ("up", 40, 50, 80, 64), ("same", 48, 64, 64, 48),
             ("wonly", 48, 64, 40, 48), ("honly", 33, 77, 77, 20), ("garden", 210, 324, 128, 128)]
    with tempfile.TemporaryDirectory() as tmp:
        for name, h, w, wo, ho in cases:
            yy, xx = np.mgrid[0:h, 0:w]
            base = np.stack([127 + 120 * np.sin(xx / 7.0 + yy / 11.0), 127 + 120 * np.cos(xx / 5.0 - yy / 13.0), (xx * 3 + yy * 5) % 256], -1)
            img = np.clip(base + rs.randint(-20, 21, size=(h, w, 3)), 0, 255).astype(np.uint8)
            mask = (rs.randint(0, 256, size=(h, w)) * (np.hypot(xx - w / 2, yy - h / 2) < 0.45 * min(h, w))).astype(np.uint8)
            ip, mp = os.path.join(tmp, name + ".png"), os.path.join(tmp, name + "_mask.png")
            Image.fromarray(img).save(ip)
            Image.fromarray(mask, mode="L").save(mp)
            rgb = ns.image_utils.load_rgb_resized(ip, (wo, ho))
            m01 = ns.image_utils.load_mask_resized_np(mp, (wo, ho))
            m01_inv = ns.image_utils.load_mask_resized_np(mp, (wo, ho), invert=True, threshold=0.3)
            blk = ns.image_utils.apply_mask_to_rgb(rgb, m01)
            pre = name + "_"
            out[pre + "image"], out[pre + "mask"] = img, mask
            out[pre + "size"] = np.array([wo, ho], np.int64)
            out[pre + "rgb"] = np.asarray(rgb, dtype=np.uint8)
            out[pre + "mask01"], out[pre + "mask01_inv03"] = m01, m01_inv
            out[pre + "blacked"] = np.asarray(blk, dtype=np.uint8)
            print(f"  g7 {name}: {h}x{w} -> {ho}x{wo}, kept {int(m01.sum())} mask pixels")
    out["names"] = np.array([c[0] for c in cases])
    save("g7_image_prep.npz", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7"]
    for w in which:
        {"g1": g1_geometry, "g2": g2_selection, "g3": g3_triangulate, "g4": g4_pipeline, "g5": g5_writers,
         "g6": g6_selection_tables, "g7": g7_image_prep}[w]()
