"""Ragged shapes through every form of the dense kernel (``pytest -m gpu``).  The full-size suites run tile-aligned grids (512^2, 960^2, 1280^2:
H*W a multiple of the 1024-cell tile, W a multiple of 4).  Here: grids whose last tile of every reference is partial, rows that do not divide
by four (the one-cell-at-a-time front end), widths that are not powers of two, one to five references per launch, 1-8 neighbours, two- and
four-channel warps, masks - seeded, 20 cases.  Required of each:

  * the four output forms agree BIT FOR BIT: ordered arrays (lfd_triangulate_dense) == unordered retirement restored from the tile table
    (lfd_triangulate_dense_segments + lfd_order_segments / lfd_pack_ply_segments / lfd_pack_points3d_segments) == the kernel-written PLY records
    (lfd_triangulate_dense_ply), with the same per-reference offsets and per-slot counts;
  * the aggregate kernel equals the CPU twin (the host build of the same per-cell source) bit for bit;
  * the dense result equals the twin's: the same cells up to a handful decided at a threshold's last bit (the device's reciprocals are Newton
    refined approximations, the host's are IEEE divisions), coordinates within 1e-5."""
import numpy as np
import pytest
import torch

import lichtfeld_densification_plugin_amd as lfd
from lichtfeld_densification_plugin_amd import synthetic
from lichtfeld_densification_plugin_amd.core import hip_backend as hb

pytestmark = pytest.mark.gpu

GRIDS = [(37, 53), (48, 52), (100, 101), (33, 1024), (65, 63), (31, 33), (129, 68), (7, 150), (1, 77), (90, 2)]     # (H, W)


def _cases():
    rs = np.random.RandomState(20240)
    out = []
    for i in range(20):
        H, W = GRIDS[i % len(GRIDS)]
        out.append(dict(id=i, H=H, W=W, k=int(rs.choice([1, 2, 3, 5, 8])), n_refs=int(rs.randint(1, 6)), channels=int(rs.choice([2, 4])),
                        masks=bool(rs.rand() < 0.4), exact=bool(rs.rand() < 0.3), noise=float(rs.choice([0.3, 1.0])),
                        cert_mode=str(rs.choice(["smooth", "tiefree", "beta"])), match=(int(rs.choice([max(W, 2), 96, 161])), int(rs.choice([max(H, 2), 80, 97])))))
    return out


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def _build(case, dev):
    H, W, k = case["H"], case["W"], case["k"]
    wm, hm = case["match"]
    cams = synthetic.ring_cameras(30, seed=3)
    rs = np.random.RandomState(100 + case["id"])
    refs_dev, refs_host = [], []
    for r in range(case["n_refs"]):
        ref = int(rs.randint(0, 30))
        nbrs = synthetic.ring_neighbours(30, ref, k)
        s = synthetic.synth_reference(cams, ref, nbrs, H, W, wm, hm, noise_px=case["noise"], outlier_frac=0.05, channels=case["channels"],
                                      seed=case["id"] * 31 + r, cert_mode=case["cert_mode"])
        ma = mbs = None
        if case["masks"]:
            ma = (rs.rand(hm, wm) > 0.15).astype(np.uint8)
            mbs = [(rs.rand(hm, wm) > 0.15).astype(np.uint8) if j % 2 == 0 else None for j in range(k)]      # (a neighbour without a mask among masked ones)
        for store, d in ((refs_dev, dev), (refs_host, torch.device("cpu"))):
            store.append(hb.ReferenceInputs(ref_cam=ref, nbr_cams=nbrs, cert=[s.cert[j].to(d) for j in range(k)],
                                            warp=[s.warp[j].contiguous().to(d) for j in range(k)], image=s.image.to(d),
                                            mask_a=None if ma is None else torch.from_numpy(ma).to(d),
                                            mask_b=None if mbs is None else [None if m is None else torch.from_numpy(m).to(d) for m in mbs]))
    return cams, refs_dev, refs_host, wm, hm


@pytest.mark.parametrize("case", _cases(), ids=lambda c: f"{c['id']}-{c['H']}x{c['W']}-k{c['k']}-r{c['n_refs']}-c{c['channels']}{'-m' if c['masks'] else ''}{'-x' if c['exact'] else ''}")
def test_every_output_form_agrees_on_ragged_shapes(dev, case):
    H, W, k = case["H"], case["W"], case["k"]
    cams, refs_dev, refs_host, wm, hm = _build(case, dev)
    cfg = lfd.DensePipelineConfig(output_path="", nns_per_ref=k, reproj_thresh=1.5)
    params = hb.make_params(cfg, exact_colour=case["exact"])
    dens = hb.HipDensifier(dev)
    dens.upload_cameras(cams)
    batch = hb.PreparedBatch(refs_dev, wm, hm, cameras=cams)
    ordered = dens.triangulate_dense(batch, params)
    assert ordered.count > 0 or case["masks"] or case["cert_mode"] == "beta"

    # ---- unordered retirement, restored ----
    seg = dens.triangulate_dense_segments(batch, params)
    tpr = (H * W + 1023) // 1024
    table = seg.table.cpu().numpy().reshape(case["n_refs"], tpr, 2)
    np.testing.assert_array_equal(seg.ref_counts.cpu().numpy(), np.diff(ordered.ref_offsets))
    np.testing.assert_array_equal(table[:, :, 1].sum(axis=1), np.diff(ordered.ref_offsets))
    assert int(table[:, :, 1].max(initial=0)) <= 1024 and (H * W % 1024 == 0 or int(table[:, -1, 1].max(initial=0)) <= H * W % 1024)    # a partial last tile holds at most its cells
    again = dens.order_segments(seg)
    np.testing.assert_array_equal(again.ref_offsets, ordered.ref_offsets)
    np.testing.assert_array_equal(again.seg_counts, ordered.seg_counts)
    for a, b in ((again.xyz, ordered.xyz), (again.rgb, ordered.rgb), (again.err, ordered.err), (again.cell, ordered.cell), (again.slot, ordered.slot)):
        assert torch.equal(a, b)
    ply_ref = dens.pack_ply(ordered.xyz, ordered.rgb)
    ply_seg, offs = dens.pack_ply_segments(seg)
    np.testing.assert_array_equal(offs, ordered.ref_offsets)
    assert torch.equal(ply_seg, ply_ref)
    p3d_seg, offs3 = dens.pack_points3d_segments(seg, id_base=3)
    np.testing.assert_array_equal(offs3, ordered.ref_offsets)
    assert torch.equal(p3d_seg, dens.pack_points3d(ordered.xyz, ordered.rgb, ordered.err, id_base=3))

    # ---- the kernel's own PLY records ----
    body, offs_p = dens.triangulate_dense_ply(batch, params)
    np.testing.assert_array_equal(offs_p, ordered.ref_offsets)
    assert torch.equal(body, ply_ref)

    # ---- against the CPU twin ----
    twin = hb.HostDensifier(2)
    twin.upload_cameras(cams)
    hbatch = hb.PreparedBatch(refs_host, wm, hm, cameras=cams)
    best_d, slot_d = dens.aggregate(batch, params)
    best_h, slot_h = twin.aggregate(hbatch, params)
    assert torch.equal(best_d.cpu(), best_h) and torch.equal(slot_d.cpu(), slot_h)
    host = twin.triangulate_dense(hbatch, params)
    key_d = np.repeat(np.arange(case["n_refs"]), np.diff(ordered.ref_offsets)) * (H * W) + ordered.cell.cpu().numpy().astype(np.int64)
    key_h = np.repeat(np.arange(case["n_refs"]), np.diff(host.ref_offsets)) * (H * W) + host.cell.numpy().astype(np.int64)
    assert np.all(np.diff(key_d) > 0) and np.all(np.diff(key_h) > 0)                    # raster order inside reference order, both
    common, id_, ih_ = np.intersect1d(key_d, key_h, return_indices=True)
    n_diff = key_d.size + key_h.size - 2 * common.size
    assert n_diff <= max(2, int(2e-3 * max(key_d.size, 1))), (n_diff, key_d.size, key_h.size)
    if common.size:
        xd, xh = ordered.xyz.cpu().numpy()[id_], host.xyz.numpy()[ih_]
        scale = np.maximum(1.0, np.abs(xh).max(axis=1, keepdims=True))
        assert np.abs(xd - xh).max() <= 1e-5 * scale.max() and np.quantile(np.abs(xd - xh) / scale, 0.999) <= 2e-6
        assert np.abs(ordered.err.cpu().numpy()[id_] - host.err.numpy()[ih_]).max() <= 2e-3
        if case["exact"]:
            np.testing.assert_array_equal(ordered.rgb.cpu().numpy()[id_], host.rgb.numpy()[ih_])
        else:
            assert np.abs(ordered.rgb.cpu().numpy()[id_] - host.rgb.numpy()[ih_]).max() <= 1e-6
    twin.close()
    dens.close()


def _coverage_tiles(H, W, tiles=24):
    t = max(1, W // tiles)
    return ((W - 1) // t + 1) * ((H - 1) // t + 1)


def test_a_grid_beyond_the_coverage_limit_is_refused_with_a_message_that_says_so(dev):
    """the device selection holds 2304 coverage tiles - every square grid fits (47 x 47 has the most), RoMa's need at most 625; a grid much taller
    than wide (24 x 200: one cell per tile, 4800 of them) is refused, loudly"""
    assert max(_coverage_tiles(n, n) for n in range(1, 1400)) == 2209 and max(_coverage_tiles(n, n) for n in (320, 512, 640, 960, 1280)) == 625
    assert _coverage_tiles(200, 24) == 4800
    dens = hb.HipDensifier(dev)
    dens.seed_rng(0)
    with pytest.raises(hb.HipBackendError, match="4800 coverage tiles.*holds 2304.*host selection"):
        dens.select_samples(torch.rand((200, 24), device=dev), 500, cap=0.9, border=2, tiles=24)
    dens.close()


@pytest.mark.parametrize("case", [c for c in _cases() if c["H"] >= 31 and c["W"] >= 31 and _coverage_tiles(c["H"], c["W"]) <= 2304], ids=lambda c: f"{c['id']}-{c['H']}x{c['W']}-k{c['k']}-c{c['channels']}{'-m' if c['masks'] else ''}")
def test_the_upstream_equivalent_path_on_ragged_shapes(dev, case):
    """Sampled mode (what upstream runs) on the same ragged grids, one reference at a time on ONE MT19937 stream like upstream: the fused call
    (lfd_triangulate_sampled) = aggregate + selection + indexed as three calls, bit for bit, stream position included; and the indexed result =
    the CPU twin's on the same selection (same groups in the same order; survivors up to a threshold's last bit, coordinates within 1e-5)."""
    H, W, k = case["H"], case["W"], case["k"]
    cams, refs_dev, refs_host, wm, hm = _build(case, dev)
    M = max(64, min(2000, H * W // 3))
    cfg = lfd.DensePipelineConfig(output_path="", nns_per_ref=k, reproj_thresh=1.5, matches_per_ref=M)
    params = hb.make_params(cfg)
    dens = hb.HipDensifier(dev)
    dens.upload_cameras(cams)
    twin = hb.HostDensifier(2)
    twin.upload_cameras(cams)
    got = {}
    for mode in ("three", "fused"):
        dens.seed_rng(9)
        res = []
        for r in refs_dev:
            b = hb.PreparedBatch([r], wm, hm, cameras=cams)
            try:
                if mode == "three":
                    best, _ = dens.aggregate(b, params)
                    sel = dens.select_samples(best[0], M, cap=0.9, border=2, tiles=24)
                    out = dens.triangulate_indexed(b, params, sel, [0, int(sel.numel())])
                    res.append(("ok", int(sel.numel()), out, sel))
                else:
                    out = dens.triangulate_sampled(b, params, M, cap=0.9, border=2, tiles=24)
                    res.append(("ok", out.n_selected, out, None))
            except ValueError as exc:                       # upstream's np.random.choice refusals (fewer non-zero weights than draws ...)
                res.append(("refused", str(exc)[:40], None, None))
        got[mode] = (res, dens.rng_state())
    assert got["three"][1][1] == got["fused"][1][1]
    np.testing.assert_array_equal(got["three"][1][0], got["fused"][1][0])
    n_ok = 0
    for ri, (a, b) in enumerate(zip(got["three"][0], got["fused"][0])):
        assert a[0] == b[0], (a[:2], b[:2])
        if a[0] != "ok":
            continue
        n_ok += 1
        assert a[1] == b[1] and a[1] > 0
        for f in ("xyz", "rgb", "err", "cell", "slot"):
            assert torch.equal(getattr(a[2], f), getattr(b[2], f)), f
        np.testing.assert_array_equal(a[2].ref_offsets, b[2].ref_offsets)
        np.testing.assert_array_equal(a[2].seg_counts, b[2].seg_counts)
        np.testing.assert_array_equal(a[2].seg_order, b[2].seg_order)
        # ... and the CPU twin on the same selection
        sel = a[3].cpu()
        hbatch = hb.PreparedBatch([refs_host[ri]], wm, hm, cameras=cams)
        host = twin.triangulate_indexed(hbatch, params, sel, [0, int(sel.numel())])
        cd, ch = a[2].cell.cpu().numpy().astype(np.int64), host.cell.numpy().astype(np.int64)
        common, id_, ih_ = np.intersect1d(cd, ch, return_indices=True)
        assert cd.size + ch.size - 2 * common.size <= max(2, int(2e-3 * max(cd.size, 1)))
        np.testing.assert_array_equal(a[2].slot.cpu().numpy()[id_], host.slot.numpy()[ih_])
        groups = lambda s_: [int(g) for g in s_[np.concatenate([[True], s_[1:] != s_[:-1]])]] if s_.size else []
        gd, gh = groups(a[2].slot.cpu().numpy()), groups(host.slot.numpy())
        assert len(set(gd)) == len(gd) and [g for g in gd if g in gh] == [g for g in gh if g in gd]          # contiguous groups, the same order
        if common.size:
            xd, xh = a[2].xyz.cpu().numpy()[id_], host.xyz.numpy()[ih_]
            assert np.abs(xd - xh).max() <= 1e-5 * max(1.0, float(np.abs(xh).max()))
    assert n_ok >= 1 or case["masks"]
    twin.close()
    dens.close()
