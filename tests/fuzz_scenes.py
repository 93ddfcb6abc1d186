"""Seeded scenes of the driver-loop fuzz, shared by tests/test_gpu_pipeline_fuzz.py (device pipeline), tests/test_pipeline_upstream_fixture.py (CPU twin) and
tests/golden/make_pipeline_fixture.py (UPSTREAM'S own run_dense_pipeline on the same scenes, development container -> tests/golden/g12_pipeline_upstream.npz):
cameras, neighbour counts, reference subsets, rectangular grids, filter / no_filter, mask files of other sizes, viz intervals, packing workers."""
import os

import numpy as np
from PIL import Image

import lichtfeld_densification_plugin_amd as lfd
from lichtfeld_densification_plugin_amd import synthetic
from lichtfeld_densification_plugin_amd.core import pipeline as pl
from lichtfeld_densification_plugin_amd.core import selection

N_SCENES = 12


class Table:
    sample_thresh = 0.9

    def __init__(self, wm, hm, table, two_channel=False):
        self.w_resized, self.h_resized, self.table, self.calls, self.two = wm, hm, table, 0, two_channel

    def match_grids_batch(self, imA, imB_list, **_kw):
        res = self.table[self.calls]
        self.calls += 1
        return [((w[..., 2:4] if self.two else w).clone(), c.clone()) for (w, c) in res]

    def reference_axes(self, H, W):                      # two-channel warps: the reference grid is the matcher's own linspace
        from lichtfeld_densification_plugin_amd.core import hip_backend as hb
        return hb.identity_axis(W), hb.identity_axis(H)

    def close(self):
        pass


def scene(sc, d):
    rs = np.random.RandomState(1000 + sc)
    n_cams = int(rs.randint(4, 9))
    H, W = [(64, 64), (48, 80), (40, 40), (72, 56)][sc % 4]
    k = int(rs.randint(1, 4))
    cams = synthetic.ring_cameras(n_cams, seed=200 + sc, arc=0.9)
    refs = sorted(int(r) for r in rs.choice(n_cams, size=int(rs.randint(1, min(n_cams, 4) + 1)), replace=False))
    nn = selection.nearest_neighbors(np.stack([c.flat_pose() for c in cams]), k)
    for i, c in enumerate(cams):
        c.image_path = os.path.join(d, f"im{i:02d}.png")
        Image.fromarray(synthetic.synth_image(H, W, 600 + 10 * sc + i).numpy()).save(c.image_path)
        c.mask_path = None
        if sc % 2 == 1 and rs.rand() < 0.6:
            mh, mw = (H, W) if rs.rand() < 0.5 else (int(rs.randint(20, 200)), int(rs.randint(20, 200)))
            blob = np.full((mh, mw), 255, np.uint8)
            for _ in range(4):
                y, x = int(rs.randint(0, mh)), int(rs.randint(0, mw))
                blob[y:y + mh // 3, x:x + mw // 4] = int(rs.choice([0, 90, 140]))
            c.mask_path = os.path.join(d, f"mask{i:02d}.png")
            Image.fromarray(blob, mode="L").save(c.mask_path)
    table = []
    for r in refs:
        nbrs = [int(n) for n in nn[r][:k]]
        # tie-free certainties: among EQUAL weights (cells on the cap) upstream's coverage pass follows NumPy's unspecified argsort order, the device
        # stage takes the lowest index - with ties the two selections differ in a few cells by design, tests/test_gpu_beta.py covers that
        s = synthetic.synth_reference(cams, r, nbrs, H, W, W, H, noise_px=float(rs.choice([0.2, 0.6])), outlier_frac=0.05, channels=4, seed=900 + sc,
                                      cert_mode="tiefree")
        table.append([(s.warp[j], s.cert[j]) for j in range(len(nbrs))])
    kw = dict(nns_per_ref=k, seed=int(rs.randint(0, 1000)), viz_interval=int(rs.choice([0, 1, 2])), pack_workers=int(rs.choice([1, 4])), no_filter=bool(sc % 3 == 1),
              matches_per_ref=int(rs.choice([200, 900, 2500])), reproj_thresh=float(rs.choice([0.8, 1.5])), min_parallax_deg=float(rs.choice([0.5, 0.0])))
    return cams, refs, nn, table, (W, H), kw


def run(cams, refs, nn, table, size, out, two=False, **cfg_kw):
    progress, viz = [], []
    cfg = lfd.DensePipelineConfig(output_path=out, roma_setting="fast", **cfg_kw)
    try:
        res = pl.run_dense_pipeline(cams, refs, nn, cfg, progress_callback=lambda p, m: progress.append((round(float(p), 6), m.split(" | ")[0])),
                                    on_sequential_viz=lambda p: viz.append((os.path.basename(p), open(p, "rb").read())), matcher=Table(size[0], size[1], table, two))
    except RuntimeError as exc:                            # "No points triangulated" (every reference refused: more draws asked for than weights): both sides must say so
        return str(exc), progress, viz
    return res, progress, viz




def assert_is_upstreams_run(result, progress, viz, g12, sc, rgb_atol=0.0):
    """``run(...)``'s return against what UPSTREAM'S run_dense_pipeline produced on scene ``sc`` (tests/golden/g12_pipeline_upstream.npz, made by
    tests/golden/make_pipeline_fixture.py): the same error, or the same survivor count, counters, progress sequence and previews, colours bit for bit
    (``rgb_atol`` for two-channel warps, whose reference grid is the matcher's linspace), positions within 1e-5 relative and errors within 2e-3 px
    (the f64 null vector against LAPACK's f32 SVD)."""
    import json
    pre = f"s{sc}_"
    want_progress = [tuple(p) for p in json.loads(str(g12[pre + "progress"]))]
    if (pre + "error") in g12.files:
        assert isinstance(result, str) and result in str(g12[pre + "error"]), (result, str(g12[pre + "error"]))
        assert [tuple(p) for p in progress] == want_progress
        return 0
    assert not isinstance(result, str), result
    xyz, rgb, err = g12[pre + "xyz"], g12[pre + "rgb"], g12[pre + "err"]
    assert result.xyz.shape == xyz.shape, (result.xyz.shape, xyz.shape)
    assert result.pairs_processed == int(g12[pre + "processed"]) and result.pairs_matched == int(g12[pre + "matched"])
    scale = np.maximum(1.0, np.abs(xyz).max(axis=1, keepdims=True))
    assert float((np.abs(result.xyz - xyz) / scale).max()) <= 1e-5
    if rgb_atol:
        np.testing.assert_allclose(result.rgb, rgb, rtol=0, atol=rgb_atol)
    else:
        np.testing.assert_array_equal(result.rgb, rgb)
    np.testing.assert_allclose(result.err, err, rtol=1e-4, atol=2e-3)
    assert [tuple(p) for p in progress] == want_progress
    names, counts = json.loads(str(g12[pre + "previews"]))
    assert [n for n, _ in viz] == names
    for (_n, blob), c in zip(viz, counts):
        head = blob.split(b"end_header\n", 1)[0].decode()
        assert int([ln for ln in head.split("\n") if ln.startswith("element vertex")][0].split()[-1]) == c
    return int(xyz.shape[0])
