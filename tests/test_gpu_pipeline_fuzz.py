"""The device pipeline against UPSTREAM'S OWN run (fixture g12: upstream's run_dense_pipeline on these scenes, made in the development container) and
against the CPU-twin pipeline (``backend="host"``: what tests/golden/check_pipeline_fuzz.py pins against upstream's own
run_dense_pipeline) on seeded scenes - cameras, neighbour counts, reference subsets, rectangular grids, filter / no_filter, mask files of other
sizes, viz intervals, packing workers, device image preparation, two-channel warps, dense mode.  ``pytest -m gpu``."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from fuzz_scenes import N_SCENES, assert_is_upstreams_run, run as _run, scene as _scene

pytestmark = pytest.mark.gpu


def _close_points(a, b, tol=1e-5):
    scale = np.maximum(1.0, np.abs(b).max(axis=1, keepdims=True))
    assert float((np.abs(a - b) / scale).max()) <= tol if a.size else True


@pytest.mark.parametrize("sc", range(N_SCENES))
def test_device_pipeline_equals_the_cpu_twin_pipeline(sc, tmp_path):
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    d = str(tmp_path)
    cams, refs, nn, table, size, kw = _scene(sc, d)
    # the sampled (upstream-equivalent) mode: the host backend makes upstream's own library calls for the selection; the device backend selects on
    # the GPU with upstream's normaliser (the default) - the same draws, the same MT19937 stream; options of the device side vary by scene
    host, hp, hv = _run(cams, refs, nn, table, size, os.path.join(d, "h", "dense.ply"), backend="host", **kw)
    dev, dp, dv = _run(cams, refs, nn, table, size, os.path.join(d, "g", "dense.ply"), two=bool(sc % 4 == 2), device_image_prep=bool(sc % 2),
                       refs_per_launch=1, **kw)
    # ... and against UPSTREAM'S OWN run on this scene (fixture g12, made in the development container from the imported reference): the chain
    # upstream -> device closes on the GPU box without the CPU twin in between
    assert_is_upstreams_run(dev, dp, dv, load_golden("g12_pipeline_upstream.npz"), sc, rgb_atol=2e-6 if sc % 4 == 2 else 0.0)
    # several references per fused call on upstream's ONE stream (lfd_triangulate_sampled_chain; these grids are below the multi-workgroup kernel's
    # size, so the references follow each other inside the call - tests/test_gpu_selection.py and test_gpu_pipeline.py chain them on large grids)
    ch, cp, cv = _run(cams, refs, nn, table, size, os.path.join(d, "c", "dense.ply"), two=bool(sc % 4 == 2), device_image_prep=bool(sc % 2),
                      refs_per_launch=3, **kw)
    assert_is_upstreams_run(ch, cp, cv, load_golden("g12_pipeline_upstream.npz"), sc, rgb_atol=2e-6 if sc % 4 == 2 else 0.0)
    if not isinstance(dev, str):
        np.testing.assert_array_equal(ch.xyz, dev.xyz)
        np.testing.assert_array_equal(ch.points_per_reference, dev.points_per_reference)
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
