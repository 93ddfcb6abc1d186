"""CPU twin of the C-ABI (lfd_create_host + the *_host entry points) against the golden vectors and the oracle.
Runs without a GPU: the twin is the host build of the kernels' per-cell source (csrc/lfd_geometry.hpp), so these tests
also pin that source's arithmetic on every machine the CPU suite runs on."""
import numpy as np
import pytest
import torch

import lichtfeld_densification_plugin_amd as lfd
from lichtfeld_densification_plugin_amd import synthetic
from lichtfeld_densification_plugin_amd.core import hip_backend as hb
from helpers import flip_report, g3_case, oracle_cam, oracle_cams, orc

G3_NAMES = ["a_filter_k3", "b_nofilter_k1", "c_rect_k3", "d_hires_k2", "e_masks_k3", "f_nosampson_k4"]


def _records(ocams):
    return [lfd.CameraRecord(uid=i, image_path="", width=c.width, height=c.height, K=c.K, R=c.R, t=c.t, P=c.P, C=c.C)
            for i, c in enumerate(ocams)]


def _config(params):
    return lfd.DensePipelineConfig(output_path="", certainty_thresh=params.certainty_thresh, reproj_thresh=params.reproj_thresh,
                                   sampson_thresh=params.sampson_thresh, min_parallax_deg=params.min_parallax_deg,
                                   no_filter=params.no_filter, matches_per_ref=params.matches_per_ref)


def _inputs(case):
    k = case["k"]
    mb = None
    if case["masks_b"] is not None:
        mb = [torch.from_numpy(m) if m is not None else None for m in case["masks_b"]]
    return hb.ReferenceInputs(ref_cam=case["ref"], nbr_cams=list(case["nbrs"]),
                              cert=[torch.from_numpy(case["cert"][j]) for j in range(k)],
                              warp=[torch.from_numpy(np.ascontiguousarray(case["warp"][j])) for j in range(k)],
                              image=torch.from_numpy(case["image"]),
                              mask_a=torch.from_numpy(case["mask_a"]) if case["mask_a"] is not None else None, mask_b=mb)


@pytest.mark.parametrize("name", G3_NAMES)
def test_indexed_twin_reproduces_upstream_goldens(g3, name):
    """Upstream's captured selection through lfd_triangulate_indexed_host: upstream's survivors, order, segments, values."""
    ocams = oracle_cams(g3)
    case = g3_case(g3, name)
    twin = hb.HostDensifier(2)
    twin.upload_cameras(_records(ocams))
    batch = hb.PreparedBatch([_inputs(case)], case["w_match"], case["h_match"], cameras=_records(ocams))
    sel = torch.from_numpy(case["sel"])
    out = twin.triangulate_indexed(batch, hb.make_params(_config(case["params"])), sel, [0, sel.numel()])
    assert out.count == case["xyz"].shape[0]
    order = [int(s) for s in out.seg_order[0] if s >= 0]
    assert [case["nbrs"][s] for s in order] == [int(v) for v in case["seg_nbr_cam"]]
    assert [int(out.seg_counts[0, s]) for s in order] == [int(v) for v in case["seg_count"]]
    np.testing.assert_allclose(out.xyz.numpy(), case["xyz"], rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(out.rgb.numpy(), case["rgb"])              # f64 colour path: bit-identical
    # err: 1e-3 px + 1e-5 relative + 4x the oracle's per-point bound on upstream's own f32 rounding noise (up to a pixel for
    # points almost in a camera's principal plane), as in tests/test_gpu_parity.py
    cert_list = [case["cert"][j] for j in range(case["k"])]
    warp_list = [case["warp"][j] for j in range(case["k"])]
    with np.errstate(all="ignore"):
        _, bk, agg = orc.prepare_reference(cert_list, warp_list, case["params"], case["mask_a"], case["masks_b"])
        noise = orc.cell_diagnostics(out.cell.numpy().astype(np.int64), bk, agg, ocams[case["ref"]], [ocams[n] for n in case["nbrs"]],
                                     case["w_match"], case["h_match"])["err_noise"]
    tol = 1e-3 + 1e-5 * np.abs(case["err"].astype(np.float64)) + 4.0 * noise.astype(np.float64)
    bad = np.abs(out.err.numpy().astype(np.float64) - case["err"].astype(np.float64)) > tol
    assert not bad.any(), (out.err.numpy()[bad][:5], case["err"][bad][:5])
    best, slot = twin.aggregate(batch, hb.make_params(_config(case["params"])))
    bc, bk, _ = orc.aggregate_best([case["post_cert"][j] for j in range(case["k"])], [case["warp"][j] for j in range(case["k"])])
    np.testing.assert_array_equal(best[0].numpy(), bc)
    np.testing.assert_array_equal(slot[0].numpy().astype(np.int64), bk)
    twin.close()


def test_dense_twin_flips_are_all_in_band_and_threads_do_not_matter():
    """Dense mode on a 192x160 grid, 2 references with 3 and 2 neighbours: every cell decided differently from the oracle
    lies inside the derived rounding band (oracle.classify_flips); 1 and 4 threads return the same bytes; the default f32
    colour stays within 2.5e-7 of the exact one."""
    H, W, wm, hm = 160, 192, 192, 160
    cams = synthetic.ring_cameras(185, seed=0)
    refs, srefs = [], []
    for i, (ref, k) in enumerate(((4, 3), (77, 2))):
        nbrs = synthetic.ring_neighbours(185, ref, k)
        s = synthetic.synth_reference(cams, ref, nbrs, H, W, wm, hm, noise_px=0.6, outlier_frac=0.08, channels=2, seed=40 + i,
                                      low_parallax_patch=(0.2, 0.5, 0.1, 0.9) if i == 0 else None)
        srefs.append(s)
        refs.append(hb.ReferenceInputs(ref_cam=ref, nbr_cams=nbrs, cert=[s.cert[j] for j in range(k)],
                                       warp=[s.warp[j].contiguous() for j in range(k)], image=s.image))
    cfg = lfd.DensePipelineConfig(output_path="")
    outs = []
    for threads in (1, 4):
        twin = hb.HostDensifier(threads)
        twin.upload_cameras(cams)
        batch = hb.PreparedBatch(refs, wm, hm, cameras=cams)
        outs.append(twin.triangulate_dense(batch, hb.make_params(cfg, exact_colour=True)))
        fast = twin.triangulate_dense(batch, hb.make_params(cfg))
        twin.close()
    a, b = outs
    assert a.count == b.count > 10000
    for x, y in ((a.xyz, b.xyz), (a.rgb, b.rgb), (a.err, b.err), (a.cell, b.cell), (a.slot, b.slot)):
        assert torch.equal(x, y)
    np.testing.assert_array_equal(a.ref_offsets, b.ref_offsets)
    np.testing.assert_array_equal(a.seg_counts, b.seg_counts)
    assert torch.equal(fast.xyz, a.xyz) and (fast.rgb - a.rgb).abs().max().item() <= 2.5e-7
    axes = (orc.identity_axis_scalar(W), orc.identity_axis_scalar(H))
    cell = a.cell.numpy().astype(np.int64)
    for r, s in enumerate(srefs):
        lo, hi = int(a.ref_offsets[r]), int(a.ref_offsets[r + 1])
        assert np.all(np.diff(cell[lo:hi]) > 0)
        rep = flip_report(cell[lo:hi], s, cams, wm, hm, orc.OracleParams(), axes)
        assert rep["out_of_band"] == 0, rep
        assert rep["flipped"] <= 3e-4 * rep["cells"] + 4, rep
        res = rep["oracle"]
        common = np.intersect1d(cell[lo:hi], res.cell)
        order_o = np.argsort(res.cell, kind="stable")
        po = order_o[np.searchsorted(res.cell[order_o], common)]
        ph = np.searchsorted(cell[lo:hi], common) + lo
        np.testing.assert_allclose(a.xyz.numpy()[ph], res.xyz[po], rtol=1e-5, atol=1e-6)
        np.testing.assert_array_equal(a.rgb.numpy()[ph], res.rgb[po])
        np.testing.assert_array_equal(a.seg_counts[r, :len(s.nbr_indices)], np.bincount(a.slot.numpy()[lo:hi], minlength=len(s.nbr_indices)))


def test_twin_and_device_contexts_do_not_stand_in_for_each_other():
    """A host context refuses the device entry points and the device API never falls back to the host."""
    import ctypes as C
    lib = hb.load_library()
    twin = hb.HostDensifier(1)
    cams = synthetic.ring_cameras(4, seed=0)
    twin.upload_cameras(cams)
    z = torch.zeros((8, 8))
    ref = hb.ReferenceInputs(ref_cam=0, nbr_cams=[1], cert=[z], warp=[torch.zeros((8, 8, 2))], image=torch.zeros((8, 8, 3), dtype=torch.uint8))
    batch = hb.PreparedBatch([ref], 8, 8)
    p = hb.make_params(lfd.DensePipelineConfig(output_path=""))
    best = torch.empty((1, 8, 8))
    rc = lib.lfd_aggregate(twin._ctx, C.byref(batch.c), C.byref(p), best.data_ptr(), None)
    assert rc == 4 and b"host" in lib.lfd_last_error(twin._ctx)          # LFD_ERR_STATE
    assert lib.lfd_rng_seed(twin._ctx, 1) == 4
    out = twin.triangulate_dense(batch, p)                                  # zero warp: nothing survives, nothing crashes
    assert out.count == 0
    twin.close()
    if not torch.cuda.is_available():
        with pytest.raises(hb.HipBackendError, match="no GPU|no CPU fallback"):
            hb.HipDensifier()


def test_sixteen_neighbours_and_one_more():
    """LFD_MAX_SLOTS = 16 neighbour slots per reference is the most a launch takes: the twin on a reference with all 16 (beside ragged
    ones) against the oracle, cell by cell through the derived bands; a 17th is refused where the batch is built."""
    H, W = 40, 56
    cams = synthetic.ring_cameras(60, seed=3)
    refs, srefs = [], []
    for i, k in enumerate((16, 5, 16)):
        ref = (7 * i + 3) % 60
        nbrs = synthetic.ring_neighbours(60, ref, k)
        s = synthetic.synth_reference(cams, ref, nbrs, H, W, W, H, noise_px=0.4, outlier_frac=0.05, channels=2, seed=11 + i, cert_mode="smooth")
        srefs.append(s)
        refs.append(hb.ReferenceInputs(ref_cam=ref, nbr_cams=nbrs, cert=[s.cert[j] for j in range(k)], warp=[s.warp[j].contiguous() for j in range(k)],
                                       image=s.image))
    twin = hb.HostDensifier(2)
    twin.upload_cameras(cams)
    batch = hb.PreparedBatch(refs, W, H, cameras=cams)
    assert batch.k == 16
    out = twin.triangulate_dense(batch, hb.make_params(lfd.DensePipelineConfig(output_path="")))
    cell, slot = out.cell.numpy().astype(np.int64), out.slot.numpy().astype(np.int64)
    axes = (orc.identity_axis_scalar(W), orc.identity_axis_scalar(H))
    for r, s in enumerate(srefs):
        lo, hi = int(out.ref_offsets[r]), int(out.ref_offsets[r + 1])
        rep = flip_report(cell[lo:hi], s, cams, W, H, orc.OracleParams(), axes)
        assert rep["out_of_band"] == 0 and rep["flipped"] <= 3, rep
        np.testing.assert_array_equal(slot[lo:hi], rep["best_k"].reshape(-1)[cell[lo:hi]])       # the winner among up to 16 slots
        assert slot[lo:hi].max() >= min(8, len(s.nbr_indices) - 1) and hi - lo > 0.5 * H * W
    twin.close()
    nbrs = synthetic.ring_neighbours(60, 0, 17)
    s = srefs[0]
    with pytest.raises(ValueError, match="neighbour slots"):
        hb.PreparedBatch([hb.ReferenceInputs(ref_cam=0, nbr_cams=nbrs, cert=[s.cert[0]] * 17, warp=[s.warp[0].contiguous()] * 17, image=s.image)], W, H)
