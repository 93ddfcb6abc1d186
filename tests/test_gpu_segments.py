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


@pytest.mark.parametrize("fill", ["nothing", "one cell", "non-finite"])
def test_degenerate_inputs_through_the_unordered_and_ply_forms(dev, fill):
    """every cell masked out (zero survivors everywhere), exactly one cell left by the mask, and NaN / Inf in warps and certainties: the three
    forms of the kernel agree (empty tables and payloads, offsets all zero; the one record; non-finite cells rejected, nothing fatal)"""
    from lichtfeld_densification_plugin_amd import synthetic
    H, W, k = 70, 90, 3
    cams = synthetic.ring_cameras(24, seed=1)
    refs = []
    for ref in (2, 9):
        nbrs = synthetic.ring_neighbours(24, ref, k)
        s = synthetic.synth_reference(cams, ref, nbrs, H, W, W, H, noise_px=0.2, outlier_frac=0.0, channels=2, seed=ref, cert_mode="tiefree")
        cert = [c.clone() for c in s.cert]
        warp = [w.clone().contiguous() for w in s.warp]
        mask_a = None
        if fill in ("nothing", "one cell"):
            # (upstream's certainty "floor" LIFTS low certainties to the threshold - every cell stays a candidate; what removes cells is a mask)
            mask_a = torch.zeros((H, W), dtype=torch.uint8)
            if fill == "one cell" and ref == 9:
                mask_a[H // 2, W // 2] = 1
        else:
            cert[0][3, :] = float("nan"); cert[1][:, 5] = float("inf"); warp[2][10:20, 10:20, :] = float("nan"); warp[0][30, 30, 0] = float("-inf")
        refs.append(hb.ReferenceInputs(ref_cam=ref, nbr_cams=nbrs, cert=[c.to(dev) for c in cert], warp=[w.to(dev) for w in warp], image=s.image.to(dev),
                                       mask_a=None if mask_a is None else mask_a.to(dev)))
    dens = hb.HipDensifier(dev)
    dens.upload_cameras(cams)
    params = hb.make_params(lfd.DensePipelineConfig(output_path="", nns_per_ref=k, reproj_thresh=1.5))
    batch = hb.PreparedBatch(refs, W, H, cameras=cams)
    ordered = dens.triangulate_dense(batch, params)
    if fill == "nothing":
        assert ordered.count == 0
    elif fill == "one cell":
        assert ordered.count <= 1 and int(ordered.ref_offsets[1]) == 0
    else:
        assert ordered.count > 1000 and bool(torch.isfinite(ordered.xyz).all()) and bool(torch.isfinite(ordered.err).all())
    seg = dens.triangulate_dense_segments(batch, params)
    np.testing.assert_array_equal(seg.ref_counts.cpu().numpy(), np.diff(ordered.ref_offsets))
    again = dens.order_segments(seg)
    np.testing.assert_array_equal(again.ref_offsets, ordered.ref_offsets)
    assert torch.equal(again.xyz, ordered.xyz) and torch.equal(again.rgb, ordered.rgb) and torch.equal(again.cell, ordered.cell)
    ply_ref = dens.pack_ply(ordered.xyz, ordered.rgb)
    body_s, offs_s = dens.pack_ply_segments(seg)
    body_k, offs_k = dens.triangulate_dense_ply(batch, params)
    np.testing.assert_array_equal(offs_s, ordered.ref_offsets)
    np.testing.assert_array_equal(offs_k, ordered.ref_offsets)
    assert body_s.numel() == body_k.numel() == ply_ref.numel() == 15 * ordered.count
    assert torch.equal(body_s, ply_ref) and torch.equal(body_k, ply_ref)
    dens.close()


@pytest.mark.parametrize("name,exact", [("fast_k3_gui", False), ("fast_k8_multi", False), ("fast_k3_masks_c4", True)])
def test_unordered_ply_records_are_the_ordered_ply_kernels_tile_by_tile(dev, name, exact):
    """lfd_triangulate_dense_ply_segments (round 5: the file payload WITHOUT the look-back): every tile's records, put back in tile order with the
    table, are lfd_triangulate_dense_ply's records byte for byte; a reference's region holds exactly its point set."""
    spec = SHAPES[name]
    H, W, wm, hm = spec["grid"]
    cams, _srefs, refs = _scene(spec, dev)
    dens = hb.HipDensifier(dev)
    dens.upload_cameras(cams)
    cfg = lfd.DensePipelineConfig(output_path="", reproj_thresh=spec["reproj"], nns_per_ref=spec["k"])
    params = hb.make_params(cfg, exact_colour=exact)
    batch = hb.PreparedBatch(refs, wm, hm, cameras=cams)
    ordered, offs = dens.triangulate_dense_ply(batch, params)
    n = len(refs)
    tpr = dens.tiles_per_ref(H, W)
    rec = torch.zeros((n * H * W * 15,), dtype=torch.uint8, device=dev)
    counts = torch.zeros((n,), dtype=torch.int64, device=dev)
    table = torch.zeros((n * tpr, 2), dtype=torch.int32, device=dev)
    for _ in range(2):                   # twice on one context: the cursors are zeroed by the launch itself
        dens.launch_dense_ply_segments(batch, params, rec, counts, table)
        dens.check_launches()
    np.testing.assert_array_equal(counts.cpu().numpy(), np.diff(offs))
    t = table.cpu().numpy().reshape(n, tpr, 2)
    host = rec.cpu().numpy().reshape(n, H * W, 15)
    want = ordered.cpu().numpy().reshape(-1, 15)
    for r in range(n):
        mine = np.concatenate([host[r, o:o + c] for o, c in t[r]]) if int(t[r, :, 1].sum()) else np.zeros((0, 15), np.uint8)
        np.testing.assert_array_equal(mine, want[offs[r]:offs[r + 1]])
    dens.close()
