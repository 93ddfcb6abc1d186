"""GPU parity tests: the HIP path (through the C-ABI) against the oracle and the committed golden
vectors.  Run on the MI355X box with ``pytest -m gpu``.

Tolerances (stated once, used everywhere):
  * survivor COUNTS, segment order, cell indices, neighbour slots: bit-identical, on every cell whose
    decision variables are farther than a guard band from their thresholds (upstream's f32 LAPACK SVD
    carries ~1e-4 px of noise in the reprojection error; a cell inside the band may legitimately flip).
    The golden fixtures contain no in-band cell, so their counts must match exactly.
  * xyz: relative 1e-5 (+1e-6 abs);  err: 1e-3 px absolute (+1e-5 relative, + 4x the oracle's bound on
    upstream's own f32 rounding noise for that point);  rgb: 1/(255*4) absolute.
"""
import numpy as np
import pytest
import torch

import lichtfeld_densification_plugin_amd as lfd
from lichtfeld_densification_plugin_amd import synthetic
from lichtfeld_densification_plugin_amd.core import hip_backend as hb
from helpers import g3_case, guard_band_ok, oracle_cams, orc

pytestmark = pytest.mark.gpu

XYZ_RTOL, XYZ_ATOL, ERR_ATOL, ERR_RTOL, RGB_ATOL = 1e-5, 1e-6, 1e-3, 1e-5, 1.0 / 255.0 / 4.0


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def _records(ocams):
    return [lfd.CameraRecord(uid=i, image_path="", width=c.width, height=c.height, K=c.K, R=c.R, t=c.t, P=c.P, C=c.C)
            for i, c in enumerate(ocams)]


def _config(params: "orc.OracleParams") -> lfd.DensePipelineConfig:
    return lfd.DensePipelineConfig(output_path="", certainty_thresh=params.certainty_thresh,
                                   reproj_thresh=params.reproj_thresh, sampson_thresh=params.sampson_thresh,
                                   min_parallax_deg=params.min_parallax_deg, no_filter=params.no_filter,
                                   matches_per_ref=params.matches_per_ref)


def _ref_inputs(case, dev, channels=4):
    k = case["k"]
    warp = case["warp"] if channels == 4 else case["warp"][..., 2:4]
    mb = None
    if case["masks_b"] is not None:
        mb = [torch.from_numpy(m).to(dev) if m is not None else None for m in case["masks_b"]]
    return hb.ReferenceInputs(
        ref_cam=case["ref"], nbr_cams=list(case["nbrs"]),
        cert=[torch.from_numpy(case["cert"][j]).to(dev) for j in range(k)],
        warp=[torch.from_numpy(np.ascontiguousarray(warp[j])).to(dev) for j in range(k)],
        image=torch.from_numpy(case["image"]).to(dev),
        mask_a=torch.from_numpy(case["mask_a"]).to(dev) if case["mask_a"] is not None else None, mask_b=mb)


def _assert_values(out_xyz, out_rgb, out_err, xyz, rgb, err, err_noise=None):
    """err_noise: the oracle's per-point bound on upstream's own f32 rounding noise in the error
    (orc._reproj_noise): ~1e-4 px normally, up to a pixel for points almost in a camera's principal
    plane, where the f32 reference value itself is not reproducible by any other evaluation order."""
    np.testing.assert_allclose(out_xyz, xyz, rtol=XYZ_RTOL, atol=XYZ_ATOL)
    tol = ERR_ATOL + ERR_RTOL * np.abs(err.astype(np.float64))      # rtol matters only for no_filter garbage (err ~ 1e14)
    if err_noise is not None:
        tol = tol + 4.0 * np.asarray(err_noise, np.float64)
    bad = np.abs(out_err.astype(np.float64) - err.astype(np.float64)) > tol
    assert not bad.any(), (f"{int(bad.sum())} reprojection errors differ by more than the tolerance: "
                           f"{out_err[bad][:5]} vs {err[bad][:5]}")
    np.testing.assert_allclose(out_rgb, rgb, rtol=0, atol=RGB_ATOL)


def _noise_for(cells, case_or_inputs, ocams_a, ocams_b, wm, hm, cert_list, warp_list, params, axes=None, masks=(None, None)):
    with np.errstate(all="ignore"):
        _, bk, agg = orc.prepare_reference(cert_list, warp_list, params, masks[0], masks[1])
        return orc.cell_diagnostics(cells, bk, agg, ocams_a, ocams_b, wm, hm, axes=axes)["err_noise"]


G3_NAMES = ["a_filter_k3", "b_nofilter_k1", "c_rect_k3", "d_hires_k2", "e_masks_k3", "f_nosampson_k4"]


@pytest.mark.parametrize("name", G3_NAMES)
def test_indexed_matches_upstream_golden(g3, dev, name):
    """Upstream-equivalent mode on the captured selection: same survivors, same order, same values."""
    ocams = oracle_cams(g3)
    case = g3_case(g3, name)
    dens = hb.HipDensifier(dev)
    dens.upload_cameras(_records(ocams))
    batch = hb.PreparedBatch([_ref_inputs(case, dev)], case["w_match"], case["h_match"])
    sel = torch.from_numpy(case["sel"]).to(dev)
    out = dens.triangulate_indexed(batch, hb.make_params(_config(case["params"])), sel, [0, sel.numel()])
    assert out.count == case["xyz"].shape[0]
    order = [int(s) for s in out.seg_order[0] if s >= 0]
    assert [case["nbrs"][s] for s in order] == [int(v) for v in case["seg_nbr_cam"]]
    assert [int(out.seg_counts[0, s]) for s in order] == [int(v) for v in case["seg_count"]]
    cert_list = [case["cert"][j] for j in range(case["k"])]
    warp_list = [case["warp"][j] for j in range(case["k"])]
    noise = _noise_for(out.cell.cpu().numpy().astype(np.int64), case, ocams[case["ref"]], [ocams[n] for n in case["nbrs"]],
                       case["w_match"], case["h_match"], cert_list, warp_list, case["params"],
                       masks=(case["mask_a"], case["masks_b"]))
    _assert_values(out.xyz.cpu().numpy(), out.rgb.cpu().numpy(), out.err.cpu().numpy(), case["xyz"], case["rgb"],
                   case["err"], noise)
    # cell / slot bookkeeping agrees with the oracle's segments
    with np.errstate(all="ignore"):
        res, _ = orc.triangulate_reference(cert_list, warp_list, case["image"], ocams[case["ref"]],
                                           [ocams[n] for n in case["nbrs"]], case["w_match"], case["h_match"],
                                           case["params"], sel_idx=case["sel"], mask_a=case["mask_a"],
                                           mask_b_list=case["masks_b"])
    np.testing.assert_array_equal(out.cell.cpu().numpy(), res.cell)
    np.testing.assert_array_equal(out.slot.cpu().numpy(), res.nbr)
    dens.close()


@pytest.mark.parametrize("name", ["a_filter_k3", "c_rect_k3", "e_masks_k3"])
def test_indexed_two_channel_warp_with_explicit_axes(g3, dev, name):
    """[xB,yB]-only warp + the A-grid axes handed over explicitly == 4-channel result."""
    ocams = oracle_cams(g3)
    case = g3_case(g3, name)
    dens = hb.HipDensifier(dev)
    dens.upload_cameras(_records(ocams))
    ax = torch.from_numpy(np.ascontiguousarray(case["warp"][0][0, :, 0])).to(dev)
    ay = torch.from_numpy(np.ascontiguousarray(case["warp"][0][:, 0, 1])).to(dev)
    batch = hb.PreparedBatch([_ref_inputs(case, dev, channels=2)], case["w_match"], case["h_match"], axes=(ax, ay))
    sel = torch.from_numpy(case["sel"]).to(dev)
    out = dens.triangulate_indexed(batch, hb.make_params(_config(case["params"])), sel, [0, sel.numel()])
    assert out.count == case["xyz"].shape[0]
    cert_list = [case["cert"][j] for j in range(case["k"])]
    warp_list = [case["warp"][j] for j in range(case["k"])]
    noise = _noise_for(out.cell.cpu().numpy().astype(np.int64), case, ocams[case["ref"]], [ocams[n] for n in case["nbrs"]],
                       case["w_match"], case["h_match"], cert_list, warp_list, case["params"],
                       masks=(case["mask_a"], case["masks_b"]))
    _assert_values(out.xyz.cpu().numpy(), out.rgb.cpu().numpy(), out.err.cpu().numpy(), case["xyz"], case["rgb"],
                   case["err"], noise)
    dens.close()


@pytest.mark.parametrize("name", G3_NAMES)
def test_aggregate_bit_exact(g3, dev, name):
    ocams = oracle_cams(g3)
    case = g3_case(g3, name)
    dens = hb.HipDensifier(dev)
    dens.upload_cameras(_records(ocams))
    batch = hb.PreparedBatch([_ref_inputs(case, dev)], case["w_match"], case["h_match"])
    best, slot = dens.aggregate(batch, hb.make_params(_config(case["params"])))
    bc, bk, _ = orc.aggregate_best([case["post_cert"][j] for j in range(case["k"])], [case["warp"][j] for j in range(case["k"])])
    np.testing.assert_array_equal(best[0].cpu().numpy(), bc)
    np.testing.assert_array_equal(slot[0].cpu().numpy().astype(np.int64), bk)
    dens.close()


def _synthetic_batch(dev, n_refs, k_list, H, W, wm, hm, seed, noise=0.4, outliers=0.05, channels=2, cert_mode="smooth",
                     n_cams=60):
    cams = synthetic.ring_cameras(n_cams, seed=seed)
    refs, srefs = [], []
    for i in range(n_refs):
        ref = (7 * i + 3) % n_cams
        k = k_list[i % len(k_list)]
        nbrs = synthetic.ring_neighbours(n_cams, ref, k)
        s = synthetic.synth_reference(cams, ref, nbrs, H, W, wm, hm, noise_px=noise, outlier_frac=outliers,
                                      channels=channels, seed=seed + i, cert_mode=cert_mode, device="cpu")
        srefs.append(s)
        refs.append(hb.ReferenceInputs(ref_cam=ref, nbr_cams=nbrs, cert=[s.cert[j].to(dev) for j in range(k)],
                                       warp=[s.warp[j].contiguous().to(dev) for j in range(k)], image=s.image.to(dev)))
    return cams, refs, srefs


def _oracle_cam(c):
    return orc.OracleCamera(K=c.K, R=c.R, t=c.t, P=c.P, C=c.C, width=c.width, height=c.height)


@pytest.mark.parametrize("H,W,wm,hm,channels,no_filter,slots", [(64, 64, 64, 64, 2, False, [3, 1, 2]), (48, 80, 80, 48, 4, False, [3, 1, 2]),
                                                                (96, 96, 64, 64, 2, False, [3, 1, 2]), (64, 64, 64, 64, 2, True, [3, 1, 2]),
                                                                (50, 50, 50, 50, 4, False, [3, 1, 2]),
                                                                # the most neighbours a launch takes (LFD_MAX_SLOTS = 16), beside ragged counts
                                                                (64, 64, 64, 64, 2, False, [16, 9, 1]), (40, 72, 72, 40, 4, False, [16, 5, 16])])
def test_dense_matches_oracle(dev, H, W, wm, hm, channels, no_filter, slots):
    """Fused kernel over every cell of several references with differing neighbour counts."""
    cams, refs, srefs = _synthetic_batch(dev, 3, slots, H, W, wm, hm, seed=5, channels=channels)
    cfg = lfd.DensePipelineConfig(output_path="", no_filter=no_filter)
    params = orc.OracleParams(no_filter=no_filter)
    dens = hb.HipDensifier(dev)
    dens.upload_cameras(cams)
    axes_np = (orc.identity_axis_scalar(W), orc.identity_axis_scalar(H))
    batch = hb.PreparedBatch(refs, wm, hm)
    out = dens.triangulate_dense(batch, hb.make_params(cfg))
    xyz, rgb, err = out.xyz.cpu().numpy(), out.rgb.cpu().numpy(), out.err.cpu().numpy()
    cell, slot = out.cell.cpu().numpy().astype(np.int64), out.slot.cpu().numpy().astype(np.int64)
    n_band = 0
    for r, s in enumerate(srefs):
        k = len(s.nbr_indices)
        ca, cbs = _oracle_cam(cams[s.ref_index]), [_oracle_cam(cams[n]) for n in s.nbr_indices]
        cert_list = [s.cert[j].numpy() for j in range(k)]
        warp_list = [s.warp[j].numpy() for j in range(k)]
        with np.errstate(all="ignore"):
            d = orc.triangulate_dense(cert_list, warp_list, s.image.numpy(), ca, cbs, wm, hm, params, axes=axes_np)
            _, bk, agg = orc.prepare_reference(cert_list, warp_list, params)
            diag = orc.cell_diagnostics(np.arange(H * W), bk, agg, ca, cbs, wm, hm, axes=axes_np)
        lo, hi = int(out.ref_offsets[r]), int(out.ref_offsets[r + 1])
        my_cell, my_slot = cell[lo:hi], slot[lo:hi]
        assert np.all(np.diff(my_cell) > 0), "survivors must be in raster order"
        np.testing.assert_array_equal(my_slot, bk.reshape(-1)[my_cell])
        np.testing.assert_array_equal(out.seg_counts[r, :k], np.bincount(my_slot, minlength=k))
        keep_hip = np.zeros(H * W, bool); keep_hip[my_cell] = True
        keep_orc = np.zeros(H * W, bool); keep_orc[d["cell"]] = True
        sure = guard_band_ok(diag, params) if not no_filter else np.ones(H * W, bool)
        n_band += int((~sure).sum())
        np.testing.assert_array_equal(keep_hip[sure], keep_orc[sure])
        both = keep_hip & keep_orc
        if no_filter:      # garbage correspondences are ill-conditioned for ANY solver: compare well-posed ones
            both &= diag["sv_ratio"] < 0.2
        pos_h = np.searchsorted(my_cell, np.nonzero(both)[0]) + lo
        pos_o = np.searchsorted(d["cell"], np.nonzero(both)[0])
        _assert_values(xyz[pos_h], rgb[pos_h], err[pos_h], d["xyz"][pos_o], d["rgb"][pos_o], d["err"][pos_o],
                       diag["err_noise"][np.nonzero(both)[0]])
    assert n_band < 0.01 * 3 * H * W
    dens.close()


def test_dense_is_deterministic_and_overflow_safe(dev):
    cams, refs, _ = _synthetic_batch(dev, 4, [3], 128, 128, 128, 128, seed=9)
    dens = hb.HipDensifier(dev)
    dens.upload_cameras(cams)
    batch = hb.PreparedBatch(refs, 128, 128)
    p = hb.make_params(lfd.DensePipelineConfig(output_path=""))
    a = dens.triangulate_dense(batch, p)
    b = dens.triangulate_dense(batch, p)
    assert a.count == b.count and a.count > 1000
    for x, y in ((a.xyz, b.xyz), (a.rgb, b.rgb), (a.err, b.err), (a.cell, b.cell), (a.slot, b.slot)):
        assert torch.equal(x, y)
    # too-small capacity: the count is still reported, nothing is written out of bounds
    out = hb.OutputBuffers(100, batch.n_refs, batch.k, dev)
    guard = out.xyz.clone()
    dens.launch_dense(batch, p, out)
    with pytest.raises(hb.HipBackendError):
        out.collect()
    assert int(out.ref_offsets[-1].item()) == a.count
    assert torch.equal(out.xyz[:100], a.xyz[:100]) and guard.shape == out.xyz.shape
    dens.close()


def test_dense_launch_sequence_of_varying_sizes_on_one_context(dev):
    """One context, launches of 1..9 references x 1..2 tiles in changing order (with and without the optional
    outputs): the ticket sequences, epochs and in-kernel counter zeroing carry over from launch to launch, so
    every launch must reproduce what a fresh context computes for the same batch."""
    cams, refs, _ = _synthetic_batch(dev, 9, [2, 3, 1], 40, 48, 48, 40, seed=21)      # 1920 cells = 2 tiles per reference
    p = hb.make_params(lfd.DensePipelineConfig(output_path=""))
    dens = hb.HipDensifier(dev)
    dens.upload_cameras(cams)
    expected = {}
    for n in (1, 2, 3, 4, 5, 8, 9):
        fresh = hb.HipDensifier(dev)
        fresh.upload_cameras(cams)
        expected[n] = fresh.triangulate_dense(hb.PreparedBatch(refs[:n], 48, 40), p)
        fresh.close()
    for n in (9, 1, 4, 4, 2, 8, 3, 1, 5, 9, 2):
        batch = hb.PreparedBatch(refs[:n], 48, 40)
        got = dens.triangulate_dense(batch, p)
        ref = expected[n]
        assert got.count == ref.count and got.count > 0
        np.testing.assert_array_equal(got.ref_offsets, ref.ref_offsets)
        np.testing.assert_array_equal(got.seg_counts, ref.seg_counts)
        for x, y in ((got.xyz, ref.xyz), (got.rgb, ref.rgb), (got.err, ref.err), (got.cell, ref.cell), (got.slot, ref.slot)):
            assert torch.equal(x, y)
        bare = hb.OutputBuffers(got.count, batch.n_refs, batch.k, dev, with_cell=False, with_segments=False)
        dens.launch_dense(batch, p, bare)
        res = bare.collect()
        assert res.count == ref.count and torch.equal(res.xyz, ref.xyz) and torch.equal(res.err, ref.err)
    dens.close()


def test_dense_full_size_properties(dev):
    """512^2 x k=3 x 8 references (the bench shape, smaller batch): size-independent properties."""
    H = W = 512
    cams, refs, srefs = _synthetic_batch(dev, 8, [3], H, W, 512, 512, seed=2, noise=0.5, outliers=0.05, n_cams=185)
    dens = hb.HipDensifier(dev)
    dens.upload_cameras(cams)
    batch = hb.PreparedBatch(refs, 512, 512)
    cfg = lfd.DensePipelineConfig(output_path="")
    out = dens.triangulate_dense(batch, hb.make_params(cfg))
    offs = out.ref_offsets
    assert offs[0] == 0 and np.all(np.diff(offs) >= 0) and offs[-1] == out.count
    np.testing.assert_array_equal(out.seg_counts.sum(axis=1), np.diff(offs))
    cell = out.cell.cpu().numpy()
    for r in range(8):
        c = cell[offs[r]:offs[r + 1]]
        assert np.all(np.diff(c) > 0) and c.min() >= 0 and c.max() < H * W
    assert 0.3 * 8 * H * W < out.count < 8 * H * W
    # every survivor really satisfies the filters (recomputed in f64 from the emitted point)
    xyz = out.xyz.cpu().numpy().astype(np.float64)
    err = out.err.cpu().numpy()
    assert np.all(err <= np.float32(cfg.reproj_thresh))
    slot = out.slot.cpu().numpy()
    for r in (0, 5):
        s = srefs[r]
        lo, hi = offs[r], offs[r + 1]
        X = np.concatenate([xyz[lo:hi], np.ones((hi - lo, 1))], 1)
        ca = cams[s.ref_index]
        z1 = (ca.P.astype(np.float64) @ X.T)[2]
        assert np.all(z1 > 0)
        # reprojection into the reference view lands on the cell's own pixel
        ax = orc.identity_axis_scalar(W).astype(np.float64)
        px = (ax[cell[lo:hi] % W] + 1) * 0.5 * 511 * (ca.width / 512.0)
        u = (ca.P.astype(np.float64) @ X.T)
        assert np.all(np.abs(u[0] / u[2] - px) <= cfg.reproj_thresh + 1e-3)
        for j, nb in enumerate(s.nbr_indices):
            m = slot[lo:hi] == j
            z2 = (cams[nb].P.astype(np.float64) @ X[m].T)[2]
            assert np.all(z2 > 0)
    # oracle spot check of one whole reference at full size
    r = 3
    s = srefs[r]
    params = orc.OracleParams()
    axes_np = (orc.identity_axis_scalar(W), orc.identity_axis_scalar(H))
    ca, cbs = _oracle_cam(cams[s.ref_index]), [_oracle_cam(cams[n]) for n in s.nbr_indices]
    with np.errstate(all="ignore"):
        d = orc.triangulate_dense([s.cert[j].numpy() for j in range(3)], [s.warp[j].numpy() for j in range(3)],
                                  s.image.numpy(), ca, cbs, 512, 512, params, axes=axes_np)
    lo, hi = offs[r], offs[r + 1]
    sym = np.setxor1d(cell[lo:hi], d["cell"])
    assert sym.size <= 3e-4 * H * W, f"{sym.size} cells differ (measured: 1.5e-4 of the cells, every one inside its band - tests/test_gpu_guardband.py)"
    common = np.intersect1d(cell[lo:hi], d["cell"])
    ph = np.searchsorted(cell[lo:hi], common) + lo
    po = np.searchsorted(d["cell"], common)
    with np.errstate(all="ignore"):
        diag = orc.cell_diagnostics(common, d["best_k"], np.stack([s.warp[j].numpy() for j in range(3)])[
            d["best_k"].reshape(-1), np.arange(H * W) // W, np.arange(H * W) % W], ca, cbs, 512, 512, axes=axes_np)
    _assert_values(out.xyz.cpu().numpy()[ph], out.rgb.cpu().numpy()[ph], err[ph], d["xyz"][po], d["rgb"][po], d["err"][po],
                   diag["err_noise"])
    dens.close()


def test_indexed_batch_of_references_and_empty_selection(dev):
    """Several references in one indexed launch, one of them with nothing selected."""
    cams, refs, srefs = _synthetic_batch(dev, 3, [3, 2, 3], 64, 64, 64, 64, seed=11, channels=4, cert_mode="tiefree")
    dens = hb.HipDensifier(dev)
    dens.upload_cameras(cams)
    batch = hb.PreparedBatch(refs, 64, 64)
    params = orc.OracleParams(matches_per_ref=900)
    rng = np.random.RandomState(4)
    sels, expect = [], []
    for i, s in enumerate(srefs):
        k = len(s.nbr_indices)
        ca, cbs = _oracle_cam(cams[s.ref_index]), [_oracle_cam(cams[n]) for n in s.nbr_indices]
        sel_in = np.zeros(0, np.int64) if i == 1 else None
        with np.errstate(all="ignore"):
            res, sel = orc.triangulate_reference([s.cert[j].numpy() for j in range(k)], [s.warp[j].numpy() for j in range(k)],
                                                 s.image.numpy(), ca, cbs, 64, 64, params, rng=rng, sel_idx=sel_in)
        sels.append(sel)
        expect.append(res)
    offs = np.concatenate([[0], np.cumsum([len(s) for s in sels])])
    sel_t = torch.from_numpy(np.concatenate(sels)).to(dev)
    out = dens.triangulate_indexed(batch, hb.make_params(lfd.DensePipelineConfig(output_path="")), sel_t, offs.tolist())
    np.testing.assert_array_equal(np.diff(out.ref_offsets), [e.count for e in expect])
    assert expect[1].count == 0 and expect[0].count > 100
    np.testing.assert_array_equal(out.cell.cpu().numpy(), np.concatenate([e.cell for e in expect]))
    noises = []
    for s, e in zip(srefs, expect):
        k = len(s.nbr_indices)
        noises.append(_noise_for(e.cell, None, _oracle_cam(cams[s.ref_index]), [_oracle_cam(cams[n]) for n in s.nbr_indices],
                                 64, 64, [s.cert[j].numpy() for j in range(k)], [s.warp[j].numpy() for j in range(k)], params))
    _assert_values(out.xyz.cpu().numpy(), out.rgb.cpu().numpy(), out.err.cpu().numpy(),
                   np.concatenate([e.xyz for e in expect]), np.concatenate([e.rgb for e in expect]),
                   np.concatenate([e.err for e in expect]), np.concatenate(noises))
    dens.close()


@pytest.mark.parametrize("channels", [2, 4])
def test_dense_and_aggregate_with_masks_pick_the_oracle_winners(dev, channels):
    """Masks on the reference and on every neighbour (certainty prologue, core/pipeline.py:405-430): the four-cells-at-a-time
    masked path of the dense and aggregate kernels picks, cell for cell, the neighbour upstream's stack/max picks, and
    the aggregated certainty is bit-identical."""
    H, W, wm, hm = 96, 128, 128, 96
    cams, refs, srefs = _synthetic_batch(dev, 2, [3, 2], H, W, wm, hm, seed=31, channels=channels, cert_mode="tiefree")
    rs = np.random.RandomState(7)

    def blob():
        m = np.ones((hm, wm), np.uint8)
        for _ in range(5):
            y, x = rs.randint(0, hm - 30), rs.randint(0, wm - 30)
            m[y:y + 25, x:x + 28] = 0
        return m
    masks = []
    for r in refs:
        ma = blob()
        mbs = [blob() for _ in r.cert]
        masks.append((ma, mbs))
        r.mask_a = torch.from_numpy(ma).to(dev)
        r.mask_b = [torch.from_numpy(m).to(dev) for m in mbs]
    dens = hb.HipDensifier(dev)
    dens.upload_cameras(cams)
    batch = hb.PreparedBatch(refs, wm, hm)
    params = hb.make_params(lfd.DensePipelineConfig(output_path=""))
    best, slot = dens.aggregate(batch, params)
    out = dens.triangulate_dense(batch, params)
    oparams = orc.OracleParams()
    cells, slots = out.cell.cpu().numpy(), out.slot.cpu().numpy()
    for i, s in enumerate(srefs):
        k = len(s.nbr_indices)
        bc, bk, _ = orc.prepare_reference([s.cert[j].numpy() for j in range(k)], [s.warp[j].numpy() for j in range(k)], oparams,
                                          masks[i][0], masks[i][1])
        np.testing.assert_array_equal(best[i].cpu().numpy(), bc)
        np.testing.assert_array_equal(slot[i].cpu().numpy().astype(np.int64), bk.astype(np.int64))
        lo, hi = int(out.ref_offsets[i]), int(out.ref_offsets[i + 1])
        assert hi - lo > 1000
        np.testing.assert_array_equal(slots[lo:hi].astype(np.int64), bk.reshape(-1)[cells[lo:hi]].astype(np.int64))
    dens.close()


def test_indexed_split_kernels_equal_the_single_kernel(dev, monkeypatch):
    """Indexed mode evaluates the selected cells chip-wide (lfd_indexed_eval_kernel) and orders them per reference;
    with LFD_INDEXED_SPLIT=0 one workgroup per reference does both.  Same bits either way, including an empty and a
    ragged selection and out-of-range indices."""
    cams, refs, srefs = _synthetic_batch(dev, 4, [3, 2, 3, 1], 96, 80, 96, 80, seed=21, channels=2, cert_mode="tiefree")
    dens = hb.HipDensifier(dev)
    dens.upload_cameras(cams)
    batch = hb.PreparedBatch(refs, 96, 80)
    rs = np.random.RandomState(3)
    sels = [np.sort(rs.choice(96 * 80, 3000, replace=False)), np.zeros(0, np.int64), rs.permutation(96 * 80)[:1500],
            np.concatenate([[-5, 96 * 80 + 7], rs.choice(96 * 80, 257, replace=False)])]
    offs = np.concatenate([[0], np.cumsum([len(x) for x in sels])]).tolist()
    sel_t = torch.from_numpy(np.concatenate(sels).astype(np.int64)).to(dev)
    params = hb.make_params(lfd.DensePipelineConfig(output_path=""))
    got = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("LFD_INDEXED_SPLIT", mode)
        dens.reload_env()                     # the switches are read at creation, not per launch
        out = dens.triangulate_indexed(batch, params, sel_t, offs)
        got[mode] = [out.xyz.cpu().numpy().copy(), out.rgb.cpu().numpy().copy(), out.err.cpu().numpy().copy(), out.cell.cpu().numpy().copy(),
                     out.slot.cpu().numpy().copy(), out.ref_offsets.copy(), out.seg_counts.copy(), out.seg_order.copy()]
    assert got["0"][5][-1] > 1000
    for a, b in zip(got["0"], got["1"]):
        np.testing.assert_array_equal(a, b)
    dens.close()


def test_invalid_arguments_are_reported_not_fatal(dev):
    cams, refs, _ = _synthetic_batch(dev, 1, [2], 32, 32, 32, 32, seed=1)
    dens = hb.HipDensifier(dev)
    batch = hb.PreparedBatch(refs, 32, 32)
    p = hb.make_params(lfd.DensePipelineConfig(output_path=""))
    with pytest.raises(hb.HipBackendError, match="lfd_upload_cameras"):
        dens.triangulate_dense(batch, p)
    dens.upload_cameras(cams[:2])      # reference camera index is now out of range
    with pytest.raises(hb.HipBackendError, match="out of range"):
        dens.triangulate_dense(batch, p)
    dens.close()


def test_precise_grid_k8_max_size(dev):
    """`precise` preset shape (1280^2 grid, 800-px match images) with 8 neighbours: the largest
    configuration upstream offers.  Dense mode properties + an oracle check of 30k random cells through
    the indexed kernel (the oracle's per-cell LAPACK path is too slow for all 1.6M cells)."""
    H = W = 1280
    wm = hm = 800
    n_cams = 12                                   # ROI subset: 12 cameras on a partial arc
    cams = synthetic.ring_cameras(n_cams, seed=3, arc=1.2)
    ref, k = 5, 8
    nbrs = synthetic.ring_neighbours(n_cams, ref, k)
    s = synthetic.synth_reference(cams, ref, nbrs, H, W, wm, hm, noise_px=0.5, outlier_frac=0.05, channels=2, seed=8,
                                  cert_mode="smooth", device="cpu")
    inp = hb.ReferenceInputs(ref_cam=ref, nbr_cams=nbrs, cert=[s.cert[j].to(dev) for j in range(k)],
                             warp=[s.warp[j].contiguous().to(dev) for j in range(k)], image=s.image.to(dev))
    dens = hb.HipDensifier(dev)
    dens.upload_cameras(cams)
    batch = hb.PreparedBatch([inp], wm, hm)
    cfg = lfd.DensePipelineConfig(output_path="")
    p = hb.make_params(cfg)
    out = dens.triangulate_dense(batch, p)
    cell = out.cell.cpu().numpy().astype(np.int64)
    assert out.count == int(out.ref_offsets[1]) and 0.2 * H * W < out.count <= H * W
    assert np.all(np.diff(cell) > 0) and cell.max() < H * W
    assert int(out.seg_counts.sum()) == out.count and out.seg_counts.shape == (1, 8)
    # random cells through the upstream-equivalent path vs the oracle
    rs = np.random.RandomState(1)
    sel = np.sort(rs.choice(H * W, size=30000, replace=False)).astype(np.int64)
    idx = dens.triangulate_indexed(batch, p, torch.from_numpy(sel).to(dev), [0, sel.size])
    params = orc.OracleParams()
    axes_np = (orc.identity_axis_scalar(W), orc.identity_axis_scalar(H))
    ca, cbs = _oracle_cam(cams[ref]), [_oracle_cam(cams[n]) for n in nbrs]
    certs, warps = [s.cert[j].numpy() for j in range(k)], [s.warp[j].numpy() for j in range(k)]
    with np.errstate(all="ignore"):
        res, _ = orc.triangulate_reference(certs, warps, s.image.numpy(), ca, cbs, wm, hm, params, sel_idx=sel, axes=axes_np)
        _, bk, agg = orc.prepare_reference(certs, warps, params)
        diag = orc.cell_diagnostics(sel, bk, agg, ca, cbs, wm, hm, axes=axes_np)
    sure = guard_band_ok(diag, params)
    keep_h = np.isin(sel, idx.cell.cpu().numpy())
    keep_o = np.isin(sel, res.cell)
    np.testing.assert_array_equal(keep_h[sure], keep_o[sure])
    assert (~sure).sum() < 0.01 * sel.size
    # the dense kernel and the indexed kernel agree cell by cell
    np.testing.assert_array_equal(np.isin(sel, cell), keep_h)
    both = np.intersect1d(idx.cell.cpu().numpy(), res.cell)
    ih = np.nonzero(np.isin(idx.cell.cpu().numpy(), both))[0]
    io = np.nonzero(np.isin(res.cell, both))[0]
    order_h = ih[np.argsort(idx.cell.cpu().numpy()[ih], kind="stable")]
    order_o = io[np.argsort(res.cell[io], kind="stable")]
    noise = diag["err_noise"][np.searchsorted(sel, np.sort(both))]
    _assert_values(idx.xyz.cpu().numpy()[order_h], idx.rgb.cpu().numpy()[order_h], idx.err.cpu().numpy()[order_h],
                   res.xyz[order_o], res.rgb[order_o], res.err[order_o], noise)
    dens.close()


def test_non_finite_inputs_are_rejected_not_fatal(dev):
    """NaN / Inf in the warp or the certainty must neither crash nor leak into the output in filter
    mode (every comparison with NaN is false, as in NumPy)."""
    cams, refs, srefs = _synthetic_batch(dev, 1, [3], 64, 64, 64, 64, seed=4, channels=2)
    w0 = refs[0].warp[0].clone()
    w0[10:14, 5:9, 0] = float("nan")
    w0[20:22, 30:40, 1] = float("inf")
    refs[0].warp[0] = w0
    c1 = refs[0].cert[1].clone()
    c1[40:44, 40:44] = float("nan")               # a NaN certainty wins the arg-max (torch.max semantics)
    refs[0].cert[1] = c1
    dens = hb.HipDensifier(dev)
    dens.upload_cameras(cams)
    out = dens.triangulate_dense(hb.PreparedBatch(refs, 64, 64), hb.make_params(lfd.DensePipelineConfig(output_path="")))
    assert out.count > 500
    for t in (out.xyz, out.rgb, out.err):
        assert torch.isfinite(t).all()
    cell = out.cell.cpu().numpy()
    slot = out.slot.cpu().numpy()
    nan_cert_cells = [y * 64 + x for y in range(40, 44) for x in range(40, 44)]
    assert np.all(slot[np.isin(cell, nan_cert_cells)] == 1)
    best, bslot = dens.aggregate(hb.PreparedBatch(refs, 64, 64), hb.make_params(lfd.DensePipelineConfig(output_path="")))
    assert torch.isnan(best[0, 41, 41]) and int(bslot[0, 41, 41]) == 1
    dens.close()


@pytest.mark.parametrize("name", ["a_filter_k3", "c_rect_k3", "d_hires_k2", "e_masks_k3", "f_nosampson_k4"])
def test_debug_matches_of_the_product_path_equal_upstream_golden(g3, dev, name):
    """F10 debug output (upstream core/pipeline.py:761-769): per neighbour group the clipped [xA,yA,xB,yB] in match pixels and
    clip(cert/cap, 0, 1) of the survivors.  The pipeline's hot path gathers them on the GPU from the maps the kernels consumed
    (core/pipeline.py::_HotPath.debug_matches); concatenated in upstream's group order they are upstream's arrays, bit for bit."""
    from lichtfeld_densification_plugin_amd.core import pipeline as pl
    ocams = oracle_cams(g3)
    case = g3_case(g3, name)
    cfg = _config(case["params"])
    dens = hb.HipDensifier(dev)
    hot = pl._HotPath(_records(ocams), cfg, 0.9, case["w_match"], case["h_match"], dev, dens)
    ref = _ref_inputs(case, dev)
    batch = hb.PreparedBatch([ref], case["w_match"], case["h_match"], cameras=_records(ocams))
    sel = torch.from_numpy(case["sel"]).to(dev)
    out = dens.triangulate_indexed(batch, hot.params, sel, [0, sel.numel()])
    best, _ = dens.aggregate(batch, hot.params)
    dbg = hot.debug_matches(ref, out.cell, out.slot, None, best[0])
    order = [int(s) for s in out.seg_order[0] if s >= 0]
    got_m = np.concatenate([dbg[s][0] for s in order], 0)
    got_c = np.concatenate([dbg[s][1] for s in order], 0)
    assert got_m.shape == case["dbg_matches"].shape
    np.testing.assert_array_equal(got_m, case["dbg_matches"])
    np.testing.assert_array_equal(got_c, case["dbg_cert"])
    dens.close()
