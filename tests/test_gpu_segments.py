"""Unordered retirement (lfd_triangulate_dense_segments, opt-in): the tile table + the consumers restore the ORDERED kernel's result
bit for bit - the structure-of-arrays arrays (lfd_order_segments) and the file payloads (lfd_pack_ply_segments /
lfd_pack_points3d_segments) - on the full-size shapes of the BASELINE configurations.  ``pytest -m gpu``."""
import numpy as np
import pytest
import torch

import lichtfeld_densification_plugin_amd as lfd
from lichtfeld_densification_plugin_amd.core import hip_backend as hb
from test_gpu_guardband import SHAPES, _scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def _run(dev, name, exact=False):
    spec = SHAPES[name]
    H, W, wm, hm = spec["grid"]
    cams, _srefs, refs = _scene(spec, dev)
    dens = hb.HipDensifier(dev)
    dens.upload_cameras(cams)
    cfg = lfd.DensePipelineConfig(output_path="", reproj_thresh=spec["reproj"], nns_per_ref=spec["k"])
    params = hb.make_params(cfg, exact_colour=exact)
    batch = hb.PreparedBatch(refs, wm, hm, cameras=cams)
    ordered = dens.triangulate_dense(batch, params)
    seg = dens.triangulate_dense_segments(batch, params)
    return dens, ordered, seg, (H, W)


@pytest.mark.parametrize("name", ["fast_k3_gui", "high_k3_patch", "fast_k8_multi", "precise_k8_roi", "fast_k3_masks_c4"])
def test_ordered_result_from_the_tile_table_is_the_ordered_kernels(dev, name):
    dens, ordered, seg, (H, W) = _run(dev, name)
    # the table itself: every tile's count, offsets that tile the reference's region without gaps or overlaps
    table = seg.table.cpu().numpy().reshape(seg.n_refs, -1, 2)
    counts = seg.ref_counts.cpu().numpy()
    np.testing.assert_array_equal(counts, np.diff(ordered.ref_offsets))
    for r in range(seg.n_refs):
        t = table[r]
        assert int(t[:, 1].sum()) == int(counts[r])
        order = np.argsort(t[:, 0], kind="stable")
        live = order[t[order, 1] > 0]
        np.testing.assert_array_equal(t[live, 0], np.concatenate([[0], np.cumsum(t[live, 1])[:-1]]))
    again = dens.order_segments(seg)
    np.testing.assert_array_equal(again.ref_offsets, ordered.ref_offsets)
    np.testing.assert_array_equal(again.seg_counts, ordered.seg_counts)
    for a, b in ((again.xyz, ordered.xyz), (again.rgb, ordered.rgb), (again.err, ordered.err), (again.cell, ordered.cell), (again.slot, ordered.slot)):
        assert torch.equal(a, b)
    # ... and the file payloads, straight from the unordered buffers
    ply, offs = dens.pack_ply_segments(seg)
    np.testing.assert_array_equal(offs, ordered.ref_offsets)
    assert torch.equal(ply, dens.pack_ply(ordered.xyz, ordered.rgb))
    p3d, offs3 = dens.pack_points3d_segments(seg, id_base=7)
    np.testing.assert_array_equal(offs3, ordered.ref_offsets)
    assert torch.equal(p3d, dens.pack_points3d(ordered.xyz, ordered.rgb, ordered.err, id_base=7))
    dens.close()


def test_exact_colour_and_repeated_launches(dev):
    """the f64-colour form of the kernel, and launches back to back on one context (the cursors are zeroed by the launch itself)"""
    dens, ordered, seg, _ = _run(dev, "fast_k3_gui", exact=True)
    a = dens.order_segments(seg)
    assert torch.equal(a.rgb, ordered.rgb) and torch.equal(a.xyz, ordered.xyz)
    spec = SHAPES["fast_k3_gui"]
    H, W, wm, hm = spec["grid"]
    cams, _s, refs = _scene(spec, dev)
    cfg = lfd.DensePipelineConfig(output_path="", reproj_thresh=spec["reproj"], nns_per_ref=spec["k"])
    params = hb.make_params(cfg, exact_colour=True)
    batch = hb.PreparedBatch(refs, wm, hm, cameras=cams)
    out = hb.OutputBuffers(batch.n_refs * H * W, batch.n_refs, batch.k, dev)
    tpr = (H * W + 1023) // 1024
    table = torch.zeros((batch.n_refs * tpr, 2), dtype=torch.int32, device=dev)
    counts = torch.full((batch.n_refs,), 12345, dtype=torch.int64, device=dev)       # stale values: the launch zeroes them
    for _ in range(5):
        dens.launch_dense_segments(batch, params, out, table, counts)
    dens.check_launches()
    np.testing.assert_array_equal(counts.cpu().numpy(), np.diff(ordered.ref_offsets))
    b = dens.order_segments(hb.SegmentedOutput(out, table, counts, batch.n_refs, H, W, batch.k))
    assert torch.equal(b.xyz, ordered.xyz) and torch.equal(b.err, ordered.err) and torch.equal(b.rgb, ordered.rgb)
    dens.close()


def test_capacity_below_a_region_per_reference_is_refused(dev):
    spec = SHAPES["fast_k3_gui"]
    H, W, wm, hm = spec["grid"]
    cams, _s, refs = _scene(spec, dev)
    dens = hb.HipDensifier(dev)
    dens.upload_cameras(cams)
    cfg = lfd.DensePipelineConfig(output_path="", nns_per_ref=spec["k"])
    batch = hb.PreparedBatch(refs, wm, hm, cameras=cams)
    out = hb.OutputBuffers(batch.n_refs * H * W - 1, batch.n_refs, batch.k, dev)
    table = torch.zeros((batch.n_refs * ((H * W + 1023) // 1024), 2), dtype=torch.int32, device=dev)
    counts = torch.zeros((batch.n_refs,), dtype=torch.int64, device=dev)
    with pytest.raises(hb.HipBackendError, match="capacity"):
        dens.launch_dense_segments(batch, hb.make_params(cfg), out, table, counts)
    dens.close()


@pytest.mark.parametrize("name,exact", [("fast_k3_gui", False), ("fast_k3_gui", True), ("high_k3_patch", False), ("fast_k3_masks_c4", False), ("fast_k8_multi", False)])
def test_dense_ply_kernel_writes_the_packers_bytes(dev, name, exact):
    """lfd_triangulate_dense_ply: the kernel's own 15-byte records = lfd_pack_ply of lfd_triangulate_dense's arrays, byte for byte (the same
    colour arithmetic, quantised like upstream's to_uint8_rgb), with the same per-reference offsets - and, with the f64 colour flag, the bytes of
    upstream's writer for upstream's colours."""
    spec = SHAPES[name]
    H, W, wm, hm = spec["grid"]
    cams, _srefs, refs = _scene(spec, dev)
    dens = hb.HipDensifier(dev)
    dens.upload_cameras(cams)
    cfg = lfd.DensePipelineConfig(output_path="", reproj_thresh=spec["reproj"], nns_per_ref=spec["k"])
    params = hb.make_params(cfg, exact_colour=exact)
    batch = hb.PreparedBatch(refs, wm, hm, cameras=cams)
    ordered = dens.triangulate_dense(batch, params)
    body, offs = dens.triangulate_dense_ply(batch, params)
    np.testing.assert_array_equal(offs, ordered.ref_offsets)
    assert torch.equal(body, dens.pack_ply(ordered.xyz, ordered.rgb))
    # a buffer that is too small: counted, not written beyond
    small = ordered.count // 2
    rec = torch.full((small * 15 + 64,), 0xAB, dtype=torch.uint8, device=dev)
    off2 = torch.zeros((batch.n_refs + 1,), dtype=torch.int64, device=dev)
    dens._check(dens._lib.lfd_triangulate_dense_ply(dens._ctx, __import__("ctypes").byref(batch.c), __import__("ctypes").byref(params), rec.data_ptr(), small,
                                                    off2.data_ptr(), None, None, None), "lfd_triangulate_dense_ply")
    dens.check_launches()
    np.testing.assert_array_equal(off2.cpu().numpy(), ordered.ref_offsets)           # the counts are still the full ones
    assert torch.equal(rec[:small * 15], body[:small * 15]) and bool((rec[small * 15:] == 0xAB).all())
    dens.close()
