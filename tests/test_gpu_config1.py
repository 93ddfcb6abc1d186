"""BASELINE config 1 at FULL size on the device: 2 views, `turbo` 320x320 grid, k = 1, --no-filter, M = 12000 (the CLI default) -
lfd_select_top_m + the indexed kernels through run_dense_pipeline - against the NumPy oracle, and against the explicit host backend
(the CPU twin) on the same scene.  tests/test_host_backend.py runs the same shape without a GPU."""
import os

import numpy as np
import pytest
import torch

import lichtfeld_densification_plugin_amd as lfd
from lichtfeld_densification_plugin_amd.core import pipeline as pl
from lichtfeld_densification_plugin_amd.core.selection import nearest_neighbors, select_cameras_kcenters
from test_host_backend import HM, M, WM, TableMatcher, _oracle_points, _scene, _two_views
from oracle import densify_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("selection", ["device", "host"])
def test_config1_full_size_on_the_device_equals_the_oracle_and_the_host_backend(tmp_path, selection):
    cams = _two_views()
    srefs = _scene(tmp_path, cams)
    flat = np.stack([c.flat_pose() for c in cams])
    refs_local = select_cameras_kcenters(flat, round(0.75 * 2))
    nn = nearest_neighbors(flat, 1)
    table = [[(srefs[r].warp[0], srefs[r].cert[0])] for r in refs_local]
    kw = dict(output_path=os.path.join(str(tmp_path), "o.ply"), roma_setting="turbo", num_refs=0.75, nns_per_ref=1, matches_per_ref=M,
              reproj_thresh=1.5, no_filter=True, seed=0, viz_interval=0)
    dev = pl.run_dense_pipeline(cams, refs_local, nn, lfd.DensePipelineConfig(selection_backend=selection, **kw), matcher=TableMatcher(table))
    host = pl.run_dense_pipeline(cams, refs_local, nn, lfd.DensePipelineConfig(backend="host", **kw), matcher=TableMatcher(table))
    params = orc.OracleParams(certainty_thresh=0.2, reproj_thresh=1.5, sampson_thresh=5.0, min_parallax_deg=0.5, no_filter=True, matches_per_ref=M)
    ox, oc, oe, counts = _oracle_points(cams, srefs, refs_local, params)
    assert dev.xyz.shape[0] == ox.shape[0] == 2 * M and dev.device_points[0].is_cuda
    np.testing.assert_array_equal(dev.points_per_reference, counts)
    np.testing.assert_allclose(dev.xyz, ox, rtol=1e-5, atol=1e-6)            # same cells, same (descending-certainty) order
    np.testing.assert_array_equal(dev.rgb, oc)                               # upstream's f64 blend, bit for bit
    np.testing.assert_allclose(dev.err, oe, rtol=1e-4, atol=2e-3)
    # device and CPU twin run the same per-cell source: they differ by the 1-ulp reciprocal / square root only
    np.testing.assert_allclose(dev.xyz, host.xyz, rtol=2e-6, atol=1e-6)
    np.testing.assert_array_equal(dev.rgb, host.rgb)
    np.testing.assert_allclose(dev.err, host.err, rtol=1e-5, atol=1e-4)
