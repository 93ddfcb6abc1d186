"""On-device coverage sampling (lfd_select_samples) against upstream's captured selections (golden
g2) and the oracle.  Bit-exact: the drawn cells, their order after np.unique, and the position of the
legacy MT19937 stream afterwards.  The one input that is NOT part of the algorithm is upstream's
normaliser s = torch f32 sum (its rounding depends on the host's thread count / ISA): the golden
tests hand the captured s to the device; the oracle tests hand the device's s to the oracle."""
import hashlib

import numpy as np
import pytest
import torch

from lichtfeld_densification_plugin_amd.core import hip_backend as hb
from helpers import orc

pytestmark = pytest.mark.gpu


def _tiefree(h, w, seed):
    rs = np.random.RandomState(seed)
    perm = rs.permutation(h * w).astype(np.float64)
    return (0.2 + 0.7 * (perm + 0.5) / (h * w)).astype(np.float32).reshape(h, w)


def _exact_s(cert, cap=0.9, border=2):
    h, w = cert.shape
    yy, xx = np.mgrid[0:h, 0:w]
    inside = (xx >= border) & (xx <= w - 1 - border) & (yy >= border) & (yy <= h - 1 - border)
    wts = np.minimum(cert, np.float32(cap)) * inside.astype(np.float32)
    return np.float32(wts.astype(np.float64).sum())


@pytest.fixture(scope="module")
def dens():
    d = hb.HipDensifier(torch.device("cuda:0"))
    yield d
    d.close()


def _numpy_state_after(dens):
    key, pos = dens.rng_state()
    rs = np.random.RandomState(0)
    rs.set_state(("MT19937", key, pos, 0, 0.0))
    return rs


def test_device_selection_reproduces_upstream_golden(g2, dens):
    dev = dens.device
    for ci, (h, w, M, seed, cseed) in enumerate(g2["cases"]):
        cert = _tiefree(int(h), int(w), int(cseed))
        assert hashlib.sha256(cert.tobytes()).hexdigest() == str(g2[f"c{ci}_cert_sha256"])
        dens.seed_rng(int(seed))
        sel = dens.select_samples(torch.from_numpy(cert).to(dev), int(M), s_override=float(g2[f"c{ci}_s"]))
        np.testing.assert_array_equal(sel.cpu().numpy(), g2[f"c{ci}_f_sel"])
        key, pos = dens.rng_state()
        assert pos == int(g2[f"c{ci}_f_mt_pos"])
        np.testing.assert_array_equal(_numpy_state_after(dens).random_sample(2), g2[f"c{ci}_f_next_doubles"])


@pytest.mark.parametrize("h,w,M,seed", [(64, 64, 1000, 3), (96, 80, 2500, 11), (320, 320, 10000, 5), (512, 512, 12000, 7),
                                        # one cell per coverage tile (W < 48): 2209 and 2080 tiles, the most a square / near-square grid has
                                        (47, 47, 600, 2), (65, 63, 900, 4), (33, 40, 300, 9)])
def test_device_selection_with_its_own_normaliser_equals_oracle(dens, h, w, M, seed):
    cert = _tiefree(h, w, 1000 + seed)
    dens.seed_rng(seed)
    sel = dens.select_samples(torch.from_numpy(cert).to(dens.device), M)
    rng = np.random.RandomState(seed)
    ref = orc.select_samples(cert, M, rng=rng, s_override=_exact_s(cert))
    np.testing.assert_array_equal(sel.cpu().numpy(), ref)
    assert dens.rng_state()[1] == int(rng.get_state()[2])


def test_stream_continues_across_references(dens):
    """Two references in a row consume ONE stream, exactly like upstream's global np.random."""
    a, b = _tiefree(64, 64, 1), _tiefree(64, 64, 2)
    dens.seed_rng(9)
    s1 = dens.select_samples(torch.from_numpy(a).to(dens.device), 1500)
    s2 = dens.select_samples(torch.from_numpy(b).to(dens.device), 1500)
    rng = np.random.RandomState(9)
    np.testing.assert_array_equal(s1.cpu().numpy(), orc.select_samples(a, 1500, rng=rng, s_override=_exact_s(a)))
    np.testing.assert_array_equal(s2.cpu().numpy(), orc.select_samples(b, 1500, rng=rng, s_override=_exact_s(b)))
    # state round trip
    key, pos = dens.rng_state()
    dens.set_rng_state(key, pos)
    assert dens.rng_state()[1] == pos


def test_checkpoints_of_the_stream_are_taken_and_rolled_back_on_the_device(dens):
    """lfd_rng_checkpoint / lfd_rng_rollback: the stream put aside and taken back in stream order, several places side by side; a place nobody
    checkpointed, or one out of range, is refused."""
    a = torch.from_numpy(_tiefree(96, 96, 5)).to(dens.device)
    dens.seed_rng(21)
    k0, p0 = dens.rng_state()
    dens.checkpoint_rng(0)
    s1 = dens.select_samples(a, 2000)
    k1, p1 = dens.rng_state()
    dens.checkpoint_rng(3)
    s2 = dens.select_samples(a, 2000)
    assert (p1, k1.tobytes()) != (p0, k0.tobytes()) and not torch.equal(s1, s2)
    dens.rollback_rng(3)
    k, p = dens.rng_state()
    assert p == p1 and np.array_equal(k, k1)
    assert torch.equal(dens.select_samples(a, 2000), s2)                     # the stream continues as it did from there
    dens.rollback_rng(0)
    k, p = dens.rng_state()
    assert p == p0 and np.array_equal(k, k0)
    assert torch.equal(dens.select_samples(a, 2000), s1)
    fresh = hb.HipDensifier(dens.device)
    try:
        fresh.seed_rng(1)
        with pytest.raises(hb.HipBackendError, match="no checkpoint"):
            fresh.rollback_rng(1)
        with pytest.raises(hb.HipBackendError, match="out of range"):
            fresh.checkpoint_rng(hb.RNG_CHECKPOINTS)
    finally:
        fresh.close()


def test_edge_cases_follow_upstream(dens, g2):
    dev = dens.device
    dens.seed_rng(0)
    assert dens.select_samples(torch.zeros((16, 16), device=dev), 100).numel() == 0          # s <= 0 -> empty
    with pytest.raises(ValueError, match="Fewer non-zero entries in p than size"):
        dens.select_samples(torch.from_numpy(_tiefree(64, 64, 1)).to(dev), 10000)
    bad = torch.from_numpy(_tiefree(32, 32, 1)).to(dev)
    bad[10, 10] = float("nan")
    with pytest.raises(ValueError, match="NaN"):
        dens.select_samples(bad, 200)
    # massive ties (floor / cap clamps): the random part is pinned, the coverage part only up to ties
    cert = g2["ties_cert"]
    dens.seed_rng(3)
    sel = dens.select_samples(torch.from_numpy(cert).to(dev), 1200).cpu().numpy()
    ref = orc.select_samples(cert, 1200, rng=np.random.RandomState(3), s_override=_exact_s(cert))
    # which of several equally heavy cells represents a tile is a tie (argsort order upstream, lowest
    # index here); a different representative may or may not coincide with a randomly drawn cell, so
    # the union size can differ by a few.  Everything else must agree.
    assert np.all(np.diff(sel) > 0) and abs(sel.size - ref.size) <= 16
    assert np.intersect1d(sel, ref).size >= ref.size - 30
    w = np.minimum(cert, np.float32(0.9)).reshape(-1)
    only_dev, only_ref = np.setdiff1d(sel, ref), np.setdiff1d(ref, sel)
    assert np.all(w[only_dev] == np.float32(0.9)) and np.all(w[only_ref] == np.float32(0.9))   # tied at the cap


@pytest.mark.parametrize("h,w,M", [(64, 64, 600), (80, 96, 5000), (512, 512, 12000), (48, 48, 5000)])
def test_top_m_equals_upstream_argsort_on_tie_free_maps(dens, g2, h, w, M):
    cert = _tiefree(h, w, 77)
    sel = dens.select_top_m(torch.from_numpy(cert).to(dens.device), M).cpu().numpy()
    ref = orc.select_samples(cert, M, no_filter=True)
    np.testing.assert_array_equal(sel, ref)


def test_top_m_matches_golden_and_orders_ties_by_index(dens, g2):
    for ci, (h, w, M, seed, cseed) in enumerate(g2["cases"]):
        if int(M) > 16384:
            continue
        cert = _tiefree(int(h), int(w), int(cseed))
        sel = dens.select_top_m(torch.from_numpy(cert).to(dens.device), int(M)).cpu().numpy()
        np.testing.assert_array_equal(sel, g2[f"c{ci}_nf_sel"])
    cert = g2["ties_cert"].copy()                       # floor / cap clamps: massive ties
    cert[5, 7] = np.nan
    sel = dens.select_top_m(torch.from_numpy(cert).to(dens.device), 1500).cpu().numpy()
    flat = np.minimum(cert, np.float32(0.9)).reshape(-1)
    v = flat[sel]
    assert sel.size == 1500 and np.unique(sel).size == 1500 and not np.isnan(v).any()
    assert np.all(np.diff(v) <= 0)                                   # descending values
    same = np.diff(v) == 0
    assert np.all(np.diff(sel)[same] > 0)                            # equal values: ascending cell index
    kth = v[-1]
    assert np.sum(flat > kth) <= 1500 <= np.sum(flat >= kth)        # exactly the M largest


@pytest.mark.parametrize("h,w,M,seed,n_wg", [(320, 320, 10000, 5, 4), (512, 512, 10000, 7, 16), (512, 384, 12000, 9, 7),
                                             (1280, 1280, 10000, 13, 64), (500, 333, 6000, 17, 5)])
def test_multi_workgroup_kernel_equals_single_workgroup_kernel(dens, monkeypatch, h, w, M, seed, n_wg):
    """The selection shared out over several workgroups (grid barriers, one workgroup on the MT19937 stream) returns
    the same cells and leaves the stream at the same position as the single-workgroup kernel, also on ragged sizes,
    with masked (zero-weight) regions, and when upstream's argument checks refuse the input (stream untouched)."""
    cert = _tiefree(h, w, 2000 + seed)
    cert[h // 3: h // 2, w // 4: w // 2] = 0.0                       # a masked block: zero weights
    t = torch.from_numpy(cert).to(dens.device)
    out = {}
    for mode, val in (("single", "0"), ("multi", str(n_wg))):
        monkeypatch.setenv("LFD_SELECT_WORKGROUPS", val)
        dens.reload_env()
        dens.seed_rng(seed)
        sel = dens.select_samples(t, M)
        sel2 = dens.select_samples(t, M)                              # a second reference continues the stream
        out[mode] = (sel.cpu().numpy(), sel2.cpu().numpy(), dens.rng_state()[1], dens.rng_state()[0].copy())
    np.testing.assert_array_equal(out["single"][0], out["multi"][0])
    np.testing.assert_array_equal(out["single"][1], out["multi"][1])
    assert out["single"][2] == out["multi"][2]
    np.testing.assert_array_equal(out["single"][3], out["multi"][3])
    # refused input (fewer non-zero weights than draws): ValueError like upstream, stream untouched
    monkeypatch.setenv("LFD_SELECT_WORKGROUPS", str(n_wg))
    dens.reload_env()
    sparse = np.zeros((h, w), np.float32)
    sparse[h // 2, : min(w, 50)] = 0.5
    dens.seed_rng(seed)
    before = dens.rng_state()
    with pytest.raises(ValueError):
        dens.select_samples(torch.from_numpy(sparse).to(dens.device), M)
    after = dens.rng_state()
    assert before[1] == after[1] and np.array_equal(before[0], after[0])
    monkeypatch.delenv("LFD_SELECT_WORKGROUPS")
    dens.reload_env()                                                  # the shared context goes back to its defaults


@pytest.mark.parametrize("no_filter", [False, True])
def test_fused_sampled_call_equals_the_three_calls(dens, no_filter):
    """lfd_triangulate_sampled (aggregate -> selection -> indexed in one asynchronous call, the count staying on the
    device) returns exactly what lfd_aggregate + lfd_select_samples / lfd_select_top_m + lfd_triangulate_indexed return, and
    leaves the MT19937 stream where they leave it; a refused input raises upstream's ValueError and emits nothing."""
    import lichtfeld_densification_plugin_amd as lfd
    from lichtfeld_densification_plugin_amd import synthetic
    dev = dens.device
    cams = synthetic.ring_cameras(40, seed=0)
    dens.upload_cameras(cams)
    H = W = 160
    M = 3000
    cfg = lfd.DensePipelineConfig(output_path="", matches_per_ref=M, no_filter=no_filter)
    params = hb.make_params(cfg)
    outs = {}
    for mode in ("three", "fused"):
        dens.seed_rng(5)
        res = []
        for ref in (3, 11):
            nbrs = synthetic.ring_neighbours(40, ref, 3)
            s = synthetic.synth_reference(cams, ref, nbrs, H, W, H, W, noise_px=0.4, outlier_frac=0.05, channels=2, seed=ref,
                                          cert_mode="tiefree", device=dev)
            r = hb.ReferenceInputs(ref_cam=ref, nbr_cams=nbrs, cert=[s.cert[j] for j in range(3)], warp=[s.warp[j] for j in range(3)], image=s.image)
            b = hb.PreparedBatch([r], W, H)
            if mode == "three":
                best, _ = dens.aggregate(b, params)
                sel = dens.select_top_m(best[0], M, cap=0.9) if no_filter else dens.select_samples(best[0], M, cap=0.9, border=2, tiles=24)
                out = dens.triangulate_indexed(b, params, sel, [0, int(sel.numel())])
                n_sel = int(sel.numel())
            else:
                out = dens.triangulate_sampled(b, params, M, cap=0.9, border=2, tiles=24)
                n_sel = out.n_selected
            res.append((n_sel, out.xyz.cpu().numpy(), out.rgb.cpu().numpy(), out.err.cpu().numpy(), out.cell.cpu().numpy(),
                        out.slot.cpu().numpy(), out.ref_offsets, out.seg_counts, out.seg_order))
        outs[mode] = (res, dens.rng_state())
    for a, b in zip(outs["three"][0], outs["fused"][0]):
        assert a[0] == b[0] and a[0] > 1000
        for x, y in zip(a[1:], b[1:]):
            np.testing.assert_array_equal(x, y)
    assert outs["three"][1][1] == outs["fused"][1][1]
    np.testing.assert_array_equal(outs["three"][1][0], outs["fused"][1][0])
    if not no_filter:       # fewer non-zero weights than draws: upstream raises, the stream is untouched, nothing is emitted
        nbrs = synthetic.ring_neighbours(40, 3, 2)
        s = synthetic.synth_reference(cams, 3, nbrs, H, W, H, W, noise_px=0.4, outlier_frac=0.0, channels=2, seed=1, cert_mode="tiefree", device=dev)
        mask = torch.zeros((H, W), dtype=torch.uint8, device=dev)
        mask[80, :40] = 1
        r = hb.ReferenceInputs(ref_cam=3, nbr_cams=nbrs, cert=[s.cert[j] for j in range(2)], warp=[s.warp[j] for j in range(2)], image=s.image, mask_a=mask)
        before = dens.rng_state()
        with pytest.raises(ValueError):
            dens.triangulate_sampled(hb.PreparedBatch([r], W, H), params, M)
        after = dens.rng_state()
        assert before[1] == after[1] and np.array_equal(before[0], after[0])


def test_tiny_weights_fall_back_to_the_host_stage_on_the_same_stream(dens):
    """certainty_thresh = 0 leaves raw certainties like 1e-12 in the map: a normalised weight below 2^-29 makes the device
    selection refuse (LFD_SELECT_INEXACT) WITHOUT consuming its MT19937 stream; upstream handles such maps normally, and so
    does the pipeline's hot path, by running the host stage (core/sampling.py) on the device's stream and handing the
    advanced stream back: same cells as upstream's own selection, stream positioned where upstream's would be."""
    import lichtfeld_densification_plugin_amd as lfd
    from lichtfeld_densification_plugin_amd import synthetic
    from lichtfeld_densification_plugin_amd.core import pipeline as pl
    from lichtfeld_densification_plugin_amd.core.sampling import select_samples_with_coverage
    dev = torch.device("cuda:0")
    H = W = 96
    cams = synthetic.ring_cameras(40, seed=0)
    nbrs = synthetic.ring_neighbours(40, 7, 2)
    s = synthetic.synth_reference(cams, 7, nbrs, H, W, W, H, noise_px=0.3, channels=2, seed=3, cert_mode="tiefree")
    cert = [c.clone() for c in s.cert]
    for c in cert:
        c[10:40, 10:40] = 1e-12                       # far below 2^-29 of the sum once normalised
    ref = hb.ReferenceInputs(ref_cam=7, nbr_cams=nbrs, cert=[c.to(dev) for c in cert], warp=[w.contiguous().to(dev) for w in s.warp],
                             image=s.image.to(dev))
    cfg = lfd.DensePipelineConfig(output_path="", certainty_thresh=0.0, matches_per_ref=1500, seed=11)
    batch = hb.PreparedBatch([ref], W, H, cameras=cams)
    d2 = hb.HipDensifier(dev)
    d2.upload_cameras(cams)
    d2.seed_rng(cfg.seed)
    best, _ = d2.aggregate(batch, hb.make_params(cfg))
    with pytest.raises(hb.SelectionInexact):
        d2.select_samples(best[0], cfg.matches_per_ref)
    key0, pos0 = d2.rng_state()
    rs0 = np.random.RandomState(cfg.seed)
    assert pos0 == rs0.get_state()[2] and np.array_equal(key0, rs0.get_state()[1])        # the refused call consumed nothing
    hot = pl._HotPath(cams, cfg, 0.9, W, H, dev, d2)
    d2.seed_rng(cfg.seed)
    out, _ = hot.sampled(ref, None, None, None)
    expect = select_samples_with_coverage(best[0].cpu(), cfg.matches_per_ref, cap=0.9, border=2, tiles=24, rng=rs0)
    idx = d2.triangulate_indexed(batch, hb.make_params(cfg), torch.from_numpy(expect.astype(np.int64)).to(dev), [0, int(expect.size)])
    assert out is not None and out.count == idx.count > 500
    assert torch.equal(out.xyz, idx.xyz) and torch.equal(out.cell, idx.cell)
    key1, pos1 = d2.rng_state()
    assert pos1 == rs0.get_state()[2] and np.array_equal(key1, rs0.get_state()[1])        # stream handed back advanced
    d2.close()


def test_sampled_mode_cli_defaults_full_size(dens):
    """BASELINE config 2 as the CLI runs it (densify.py:318-415): fast 512^2, k=4, M=12000, reproj 1.5, upstream's own mode.  The
    fused device call selects exactly the cells the oracle selects when given the device's normaliser, and emits the oracle's
    survivors of those cells, group by group, in upstream's order."""
    import lichtfeld_densification_plugin_amd as lfd
    from lichtfeld_densification_plugin_amd import synthetic
    from helpers import oracle_cam
    dev = dens.device
    H = W = 512
    cams = synthetic.ring_cameras(185, seed=0)
    dens.upload_cameras(cams)
    ref, k, M = 42, 4, 12000
    nbrs = synthetic.ring_neighbours(185, ref, k)
    # tie-free certainties: with ties (floor / cap clamps) upstream's coverage pass depends on NumPy's unspecified argsort order
    s = synthetic.synth_reference(cams, ref, nbrs, H, W, W, H, noise_px=1.0, outlier_frac=0.05, channels=2, seed=77, cert_mode="tiefree")
    r = hb.ReferenceInputs(ref_cam=ref, nbr_cams=nbrs, cert=[s.cert[j].to(dev) for j in range(k)], warp=[s.warp[j].contiguous().to(dev) for j in range(k)],
                           image=s.image.to(dev))
    cfg = lfd.DensePipelineConfig(output_path="", matches_per_ref=M, reproj_thresh=1.5, nns_per_ref=k)
    batch = hb.PreparedBatch([r], W, H, cameras=cams)
    dens.seed_rng(0)
    out = dens.triangulate_sampled(batch, hb.make_params(cfg), M, cap=0.9, border=2, tiles=24)
    params = orc.OracleParams(reproj_thresh=1.5, matches_per_ref=M)
    axes = (orc.identity_axis_scalar(W), orc.identity_axis_scalar(H))
    certs, warps = [s.cert[j].numpy() for j in range(k)], [s.warp[j].numpy() for j in range(k)]
    with np.errstate(all="ignore"):
        best, bk, agg = orc.prepare_reference(certs, warps, params)
        sel = orc.select_samples(best, M, rng=np.random.RandomState(0), s_override=float(_exact_s(best)))
        res = orc.triangulate_selected(sel, best, bk, agg, s.image.numpy(), oracle_cam(cams[ref]), [oracle_cam(cams[n]) for n in nbrs], W, H, params, axes=axes)
    assert out.n_selected == sel.size and 0.85 * M < sel.size <= M + 24 * 24
    assert abs(out.count - res.count) <= 3                 # a cell within upstream's rounding noise of a threshold may differ (test_gpu_guardband)
    common, ih, io_ = np.intersect1d(out.cell.cpu().numpy(), res.cell, return_indices=True)
    assert common.size >= res.count - 3
    np.testing.assert_allclose(out.xyz.cpu().numpy()[ih], res.xyz[io_], rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(out.rgb.cpu().numpy()[ih], res.rgb[io_])
    if out.count == res.count and common.size == res.count:
        np.testing.assert_array_equal(out.cell.cpu().numpy(), res.cell)          # upstream's group order


@pytest.mark.parametrize("no_filter", [False, True])
def test_batched_selection_equals_one_reference_at_a_time(dens, no_filter):
    """lfd_triangulate_sampled_multi runs the selections of a batch side by side (one launch per 32 references, every reference
    with its own scratch block, barrier words and MT19937 state): 40 references at 256x256, one of them refusing its input,
    give bit for bit what 40 seeded single-reference calls give, and leave the context's own stream alone."""
    import lichtfeld_densification_plugin_amd as lfd
    from lichtfeld_densification_plugin_amd import synthetic
    dev = dens.device
    n_cams = 48
    cams = synthetic.ring_cameras(n_cams, seed=0)
    dens.upload_cameras(cams)
    H = W = 256
    M = 2500
    R = 40
    cfg = lfd.DensePipelineConfig(output_path="", matches_per_ref=M, no_filter=no_filter)
    params = hb.make_params(cfg)
    refs, keep = [], []
    for ref in range(R):
        nbrs = synthetic.ring_neighbours(n_cams, ref, 3)
        s = synthetic.synth_reference(cams, ref, nbrs, H, W, H, W, noise_px=0.4, outlier_frac=0.05, channels=2, seed=100 + ref,
                                      cert_mode="tiefree", device=dev)
        mask = None
        if ref == 17 and not no_filter:      # fewer non-zero weights than draws: np.random.choice refuses (status 3), nothing is emitted
            mask = torch.zeros((H, W), dtype=torch.uint8, device=dev)
            mask[100, :30] = 1
        keep.append(s)
        refs.append(hb.ReferenceInputs(ref_cam=ref, nbr_cams=nbrs, cert=[s.cert[j] for j in range(3)], warp=[s.warp[j] for j in range(3)],
                                       image=s.image, mask_a=mask))
    seeds = [1000 + 7 * r for r in range(R)]
    dens.seed_rng(99)
    before = dens.rng_state()
    cap = M + 24 * 24 + 64
    out = hb.OutputBuffers(cap * R, R, 3, dev, True)
    dens.launch_sampled_multi(hb.PreparedBatch(refs, W, H), params, M, out, seeds, cap=0.9, border=2, tiles=24)
    with torch.cuda.stream(dens.stream):
        got = out.collect(indexed=True, check_selection=False)
    after = dens.rng_state()
    assert before[1] == after[1] and np.array_equal(before[0], after[0])
    info = out.sel_info.cpu().numpy()
    offs = got.ref_offsets
    total = 0
    for r in range(R):
        if r == 17 and not no_filter:
            assert int(info[2 * r + 1]) == 3 and offs[r + 1] == offs[r]
            continue
        dens.seed_rng(seeds[r])
        one = dens.triangulate_sampled(hb.PreparedBatch([refs[r]], W, H), params, M, cap=0.9, border=2, tiles=24)
        lo, hi = int(offs[r]), int(offs[r + 1])
        assert hi - lo == one.count and int(info[2 * r]) == one.n_selected and int(info[2 * r + 1]) == 0
        np.testing.assert_array_equal(got.xyz[lo:hi].cpu().numpy(), one.xyz.cpu().numpy())
        np.testing.assert_array_equal(got.rgb[lo:hi].cpu().numpy(), one.rgb.cpu().numpy())
        np.testing.assert_array_equal(got.err[lo:hi].cpu().numpy(), one.err.cpu().numpy())
        np.testing.assert_array_equal(got.cell[lo:hi].cpu().numpy(), one.cell.cpu().numpy())
        total += one.count
    assert total > 1000 * (R - 1)


def _chain_scene(dens, R, H, W, refuse=(), empty=()):
    from lichtfeld_densification_plugin_amd import synthetic
    dev = dens.device
    n_cams = 48
    cams = synthetic.ring_cameras(n_cams, seed=0)
    dens.upload_cameras(cams)
    refs, keep = [], []
    for r in range(R):
        ref = r % n_cams
        nbrs = synthetic.ring_neighbours(n_cams, ref, 3)
        s = synthetic.synth_reference(cams, ref, nbrs, H, W, H, W, noise_px=0.4, outlier_frac=0.05, channels=2, seed=300 + r,
                                      cert_mode="tiefree", device=dev)
        mask = None
        if r in refuse:          # fewer non-zero weights than draws: np.random.choice raises before it draws (status 3)
            mask = torch.zeros((H, W), dtype=torch.uint8, device=dev)
            mask[H // 2, :30] = 1
        if r in empty:           # nothing but masked cells: s = 0, upstream returns an empty selection without drawing
            mask = torch.zeros((H, W), dtype=torch.uint8, device=dev)
        keep.append(s)
        refs.append(hb.ReferenceInputs(ref_cam=ref, nbr_cams=nbrs, cert=[s.cert[j] for j in range(3)], warp=[s.warp[j] for j in range(3)],
                                       image=s.image, mask_a=mask))
    return refs, keep


@pytest.mark.parametrize("H,M,R,pos0,with_s", [(256, 2500, 40, None, False), (256, 2500, 7, 311, True), (512, 10000, 6, 623, True),
                                               (512, 12000, 3, 624, False), (64, 900, 5, 17, True), (128, 3000, 33, 0, False),
                                               # four compute workgroups (4 096 threads) for 5 100 draws per first round: the window a reference looks up in
                                               # advance is shorter than what it owns
                                               (192, 6000, 6, 5, True)])
def test_chained_references_equal_successive_calls_on_one_stream(dens, H, M, R, pos0, with_s):
    """lfd_triangulate_sampled_chain: R references in one call on the context's single MT19937 stream give, bit for bit, the cells, the
    points and the final stream of R successive lfd_triangulate_sampled calls - with a refused reference (it consumes nothing) and an empty one
    in the middle, from even and odd stream positions (a double that straddles two keys), with handed-in normalisers, across the 32-reference
    launch boundary, and on a map too small for the multi-workgroup kernel (64 x 64: reference after reference inside the call)."""
    import lichtfeld_densification_plugin_amd as lfd
    dev = dens.device
    W = H
    refuse, empty = ({2, R - 1} if R > 4 else {1}), ({4} if R > 5 else set())
    refs, _keep = _chain_scene(dens, R, H, W, refuse=refuse, empty=empty)
    cfg = lfd.DensePipelineConfig(output_path="", matches_per_ref=M)
    params = hb.make_params(cfg)
    cap = M + 24 * 24 + 64

    def start():
        dens.seed_rng(4242)
        if pos0 is not None:
            key, _ = dens.rng_state()
            dens.set_rng_state(key, pos0)

    # the normalisers: the exact sum for most, one ulp off for every third reference (upstream's torch sum may round either way)
    s_list = None
    if with_s:
        best, _ = dens.aggregate(hb.PreparedBatch(refs, W, H), params)
        s_list = []
        for r in range(R):
            s = float(_exact_s(best[r].cpu().numpy()))
            if r % 3 == 1 and s > 0:
                s = float(np.nextafter(np.float32(s), np.float32(np.inf)))
            s_list.append(s)
    # reference after reference
    start()
    singles = []
    for r in range(R):
        out1 = hb.OutputBuffers(cap, 1, 3, dev, True)
        cells1 = torch.full((cap,), -1, dtype=torch.int64, device=dev)
        dens.launch_sampled(hb.PreparedBatch([refs[r]], W, H), params, M, out1, cap=0.9, border=2, tiles=24,
                            s_override=s_list[r] if s_list else 0.0, sel_cells=cells1)
        with torch.cuda.stream(dens.stream):
            one = out1.collect(indexed=True, check_selection=False)
        singles.append((one, cells1.cpu().numpy(), out1.sel_info.cpu().numpy().copy()))
    key_a, pos_a = dens.rng_state()
    # the same in one call
    start()
    out = hb.OutputBuffers(cap * R, R, 3, dev, True)
    cells = torch.full((cap * R,), -1, dtype=torch.int64, device=dev)
    dens.launch_sampled_chain(hb.PreparedBatch(refs, W, H), params, M, out, s_overrides=s_list, cap=0.9, border=2, tiles=24, sel_cells=cells)
    with torch.cuda.stream(dens.stream):
        got = out.collect(indexed=True, check_selection=False)
    key_b, pos_b = dens.rng_state()
    assert got.launch_status == 0
    assert pos_a == pos_b and np.array_equal(key_a, key_b)
    cells = cells.cpu().numpy()
    offs = got.ref_offsets
    drew = 0
    for r, (one, cells1, info1) in enumerate(singles):
        assert int(got.sel_status[r]) == int(info1[1]), r
        if r in refuse:
            assert int(got.sel_status[r]) == 3
        n_sel = int(info1[0])
        assert int(out.sel_info[2 * r]) == n_sel
        np.testing.assert_array_equal(cells[r * cap:r * cap + n_sel], cells1[:n_sel])
        lo, hi = int(offs[r]), int(offs[r + 1])
        assert hi - lo == one.count
        if r in refuse or r in empty:
            assert hi == lo
            continue
        drew += 1
        np.testing.assert_array_equal(got.xyz[lo:hi].cpu().numpy(), one.xyz.cpu().numpy())
        np.testing.assert_array_equal(got.rgb[lo:hi].cpu().numpy(), one.rgb.cpu().numpy())
        np.testing.assert_array_equal(got.err[lo:hi].cpu().numpy(), one.err.cpu().numpy())
        np.testing.assert_array_equal(got.cell[lo:hi].cpu().numpy(), one.cell.cpu().numpy())
    assert drew == R - len(refuse) - len(empty) and got.count > 0.5 * M * drew


@pytest.mark.parametrize("h,w,M,seed,heavy,light", [(256, 256, 6000, 21, 40, 0.0004), (512, 512, 10000, 22, 200, 0.0005), (320, 256, 4000, 23, 8, 0.00005),
                                                    (128, 128, 3000, 24, 30, 0.0008)])
def test_peaked_weights_take_many_rounds_of_both_kinds(dens, h, w, M, seed, heavy, light):
    """A few cells that carry most of the weight: the first round's draws pile up on them, the second round has THOUSANDS of draws left (the
    multi-workgroup kernel rebuilds its cumulative sum for it), the rounds after that a handful (searched in the weights themselves, a threshold
    per draw instead of a division per cell).  Cells, their number and the stream position against NumPy's own choice() (the oracle), and the
    single-workgroup kernel - which rebuilds every round - the same."""
    rs = np.random.RandomState(seed)
    perm = rs.permutation(h * w).astype(np.float64)
    cert = (light * (1.0 + (perm + 0.5) / (h * w))).astype(np.float32).reshape(h, w)          # tie-free, all within a factor of two
    inner = [(y, x) for y in range(4, h - 4) for x in range(4, w - 4)]
    for n, k in enumerate(rs.choice(len(inner), size=heavy, replace=False)):
        cert[inner[k]] = np.float32(0.5 + 0.3 * n / heavy)                                   # distinct heavy cells
    dev = dens.device
    t = torch.from_numpy(cert).to(dev)
    dens.seed_rng(seed)
    sel = dens.select_samples(t, M).cpu().numpy()
    pos = dens.rng_state()[1]
    rng = np.random.RandomState(seed)
    counting = _CountingRng(rng)
    ref = orc.select_samples(cert, M, rng=counting, s_override=_exact_s(cert))
    np.testing.assert_array_equal(sel, ref)
    assert pos == int(rng.get_state()[2])
    assert len(counting.rounds) >= 3 and counting.rounds[1] > 1024 and counting.rounds[-1] <= 1024, counting.rounds
    import os
    os.environ["LFD_SELECT_WORKGROUPS"] = "0"                  # the single-workgroup kernel
    try:
        dens.reload_env()
        dens.seed_rng(seed)
        np.testing.assert_array_equal(dens.select_samples(t, M).cpu().numpy(), ref)
        assert dens.rng_state()[1] == pos
    finally:
        del os.environ["LFD_SELECT_WORKGROUPS"]
        dens.reload_env()


class _CountingRng:
    """np.random.RandomState that notes how many doubles every random_sample() call of choice() asks for (= the draws of a round)"""

    def __init__(self, rs):
        self._rs, self.rounds = rs, []

    def choice(self, a, size=None, replace=True, p=None):
        # numpy's legacy algorithm (mtrand.pyx, RandomState.choice, replace=False with p), restated to see the rounds
        n_uniq, pp = 0, np.array(p, dtype=np.float64, copy=True)
        found = np.zeros(size, dtype=np.int64)
        flat_found = found.ravel()
        while n_uniq < size:
            self.rounds.append(size - n_uniq)
            x = self._rs.random_sample((size - n_uniq,))
            if n_uniq > 0:
                pp[flat_found[0:n_uniq]] = 0
            cdf = np.cumsum(pp)
            cdf /= cdf[-1]
            new = cdf.searchsorted(x, side="right")
            _, unique_indices = np.unique(new, return_index=True)
            unique_indices.sort()
            new = new.take(unique_indices)
            flat_found[n_uniq:n_uniq + new.size] = new
            n_uniq += new.size
        return found

    def __getattr__(self, name):
        return getattr(self._rs, name)


def test_chained_selections_stress_against_numpy(dens):
    """Thirty chained calls of random shape - 1 to 12 references, grids from 128 x 128 to 512 x 384, M from hundreds to 12 000, tie-free maps and maps
    whose weight sits on a few cells (a second round of thousands of draws in the middle of a chain), one call continuing the stream of the other -
    against NumPy's own choice() on ONE RandomState: every reference's cells, and the stream position after every call."""
    import lichtfeld_densification_plugin_amd as lfd
    from lichtfeld_densification_plugin_amd import synthetic
    dev = dens.device
    n_cams = 24
    cams = synthetic.ring_cameras(n_cams, seed=0)
    dens.upload_cameras(cams)
    import os
    n_calls = int(os.environ.get("LFD_CHAIN_STRESS_CALLS", "30"))          # (a longer one-off: LFD_CHAIN_STRESS_CALLS=250 LFD_CHAIN_STRESS_SEED=1 ... -k stress)
    rs = np.random.RandomState(int(os.environ.get("LFD_CHAIN_STRESS_SEED", "2026")))
    host_rng = np.random.RandomState(777)
    dens.seed_rng(777)
    warps = {}
    total_refs = peaked_refs = 0
    for call in range(n_calls):
        H, W = [(128, 128), (192, 160), (256, 256), (320, 320), (512, 384)][int(rs.randint(0, 5))]
        R = int(rs.randint(1, 13))
        M = int(rs.choice([400, 1500, 3000, 6000, 12000]))
        M = min(M, int(0.5 * H * W))
        if (H, W) not in warps:
            s = synthetic.synth_reference(cams, 0, [1, 2], H, W, W, H, noise_px=0.4, outlier_frac=0.05, channels=2, seed=1, cert_mode="tiefree", device=dev)
            warps[(H, W)] = (s.warp, s.image)
        warp, image = warps[(H, W)]
        maps, refs = [], []
        for r in range(R):
            perm = rs.permutation(H * W).astype(np.float64)
            if rs.rand() < 0.3:          # a few heavy cells over a light floor
                light = float(rs.choice([0.0004, 0.001]))
                cert = (light * (1.0 + (perm + 0.5) / (H * W))).astype(np.float32).reshape(H, W)
                heavy = int(rs.randint(5, 60))
                ys, xs = rs.randint(4, H - 4, size=heavy), rs.randint(4, W - 4, size=heavy)
                cert[ys, xs] = (0.5 + 0.3 * np.arange(heavy) / heavy).astype(np.float32)
                peaked_refs += 1
            else:
                cert = (0.2 + 0.7 * (perm + 0.5) / (H * W)).astype(np.float32).reshape(H, W)
            maps.append(cert)
            t = torch.from_numpy(cert).to(dev)
            refs.append(hb.ReferenceInputs(ref_cam=0, nbr_cams=[1, 2], cert=[t, t], warp=[warp[0], warp[1]], image=image))
        cfg = lfd.DensePipelineConfig(output_path="", matches_per_ref=M, certainty_thresh=1e-5, nns_per_ref=2)
        params = hb.make_params(cfg)
        cap = M + 24 * 24 + 64
        out = hb.OutputBuffers(cap * R, R, 2, dev, True)
        cells = torch.full((cap * R,), -1, dtype=torch.int64, device=dev)
        batch = hb.PreparedBatch(refs, W, H)
        best, _ = dens.aggregate(batch, params)
        dens.launch_sampled_chain(batch, params, M, out, cap=0.9, border=2, tiles=24, sel_cells=cells)
        with torch.cuda.stream(dens.stream):
            got = out.collect(indexed=True, check_selection=False)
        assert got.launch_status == 0
        cells = cells.cpu().numpy()
        info = out.sel_info.cpu().numpy()
        for r in range(R):
            b = best[r].cpu().numpy()
            np.testing.assert_array_equal(b, maps[r])                 # (the aggregated map IS the map handed in: both neighbours carry it)
            assert int(info[2 * r + 1]) == 0, (call, r, int(info[2 * r + 1]))
            ref = orc.select_samples(b, M, rng=host_rng, s_override=_exact_s(b))
            n = int(info[2 * r])
            assert n == ref.size, (call, r, n, ref.size)
            np.testing.assert_array_equal(cells[r * cap:r * cap + n], ref)
        assert dens.rng_state()[1] == int(host_rng.get_state()[2]), call
        total_refs += R
    assert total_refs > 4 * n_calls and peaked_refs > n_calls // 2
