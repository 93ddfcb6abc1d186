"""The device pipeline against the CPU-twin pipeline (``backend="host"``: what tests/golden/check_pipeline_fuzz.py pins against upstream's own
run_dense_pipeline) on seeded scenes - cameras, neighbour counts, reference subsets, rectangular grids, filter / no_filter, mask files of other
sizes, viz intervals, packing workers, device image preparation, two-channel warps, dense mode.  ``pytest -m gpu``."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

import lichtfeld_densification_plugin_amd as lfd
from lichtfeld_densification_plugin_amd import synthetic
from lichtfeld_densification_plugin_amd.core import pipeline as pl
from lichtfeld_densification_plugin_amd.core import selection

pytestmark = pytest.mark.gpu


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


def _scene(sc, d):
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


def _run(cams, refs, nn, table, size, out, two=False, **cfg_kw):
    progress, viz = [], []
    cfg = lfd.DensePipelineConfig(output_path=out, roma_setting="fast", **cfg_kw)
    try:
        res = pl.run_dense_pipeline(cams, refs, nn, cfg, progress_callback=lambda p, m: progress.append((round(float(p), 6), m.split(" | ")[0])),
                                    on_sequential_viz=lambda p: viz.append((os.path.basename(p), open(p, "rb").read())), matcher=Table(size[0], size[1], table, two))
    except RuntimeError as exc:                            # "No points triangulated" (every reference refused: more draws asked for than weights): both sides must say so
        return str(exc), progress, viz
    return res, progress, viz


def _close_points(a, b, tol=1e-5):
    scale = np.maximum(1.0, np.abs(b).max(axis=1, keepdims=True))
    assert float((np.abs(a - b) / scale).max()) <= tol if a.size else True


@pytest.mark.parametrize("sc", range(12))
def test_device_pipeline_equals_the_cpu_twin_pipeline(sc, tmp_path):
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    d = str(tmp_path)
    cams, refs, nn, table, size, kw = _scene(sc, d)
    # the sampled (upstream-equivalent) mode: the host backend makes upstream's own library calls for the selection; the device backend selects on
    # the GPU with upstream's normaliser (the default) - the same draws, the same MT19937 stream; options of the device side vary by scene
    host, hp, hv = _run(cams, refs, nn, table, size, os.path.join(d, "h", "dense.ply"), backend="host", **kw)
    dev, dp, dv = _run(cams, refs, nn, table, size, os.path.join(d, "g", "dense.ply"), two=bool(sc % 4 == 2), device_image_prep=bool(sc % 2),
                       refs_per_launch=1, **kw)
    if isinstance(host, str) or isinstance(dev, str):
        assert host == dev and "No points triangulated" in host and dp == hp
    else:
        np.testing.assert_array_equal(dev.points_per_reference, host.points_per_reference)
        assert (dev.pairs_processed, dev.pairs_matched) == (host.pairs_processed, host.pairs_matched)
        _close_points(dev.xyz, host.xyz)
        if sc % 4 == 2:     # two-channel warps: the reference positions are the matcher's linspace (reference_axes), not the warp's own first two
            np.testing.assert_allclose(dev.rgb, host.rgb, rtol=0, atol=2e-6)   # channels, which the synthetic scene rounds differently in the last bit
        else:
            np.testing.assert_array_equal(dev.rgb, host.rgb)                # the indexed path blends in f64 on both
        np.testing.assert_allclose(dev.err, host.err, rtol=1e-4, atol=2e-3)
        assert dp == hp and [n for n, _ in dv] == [n for n, _ in hv]
        for (_, a), (_, b) in zip(dv, hv):                                  # previews: the same vertex counts, bodies to the position tolerance
            assert a.split(b"end_header\n")[0] == b.split(b"end_header\n")[0]
    # dense mode: every candidate cell, on the device in one launch per `refs_per_launch` references against the twin
    kd = dict(kw, viz_interval=0)
    hd, _, _ = _run(cams, refs, nn, table, size, os.path.join(d, "hd", "dense.ply"), backend="host", triangulation_mode="dense", exact_colour=True, **kd)
    gd, _, _ = _run(cams, refs, nn, table, size, os.path.join(d, "gd", "dense.ply"), triangulation_mode="dense", refs_per_launch=int(1 + sc % 3), exact_colour=True, **kd)
    diff = np.abs(gd.points_per_reference.astype(np.int64) - hd.points_per_reference.astype(np.int64))
    assert int(diff.sum()) <= max(2, int(2e-3 * hd.xyz.shape[0]))           # a handful of cells at a threshold's last bit (device reciprocals vs IEEE divisions)
    if int(diff.sum()) == 0:
        _close_points(gd.xyz, hd.xyz)
        np.testing.assert_array_equal(gd.rgb, hd.rgb)
