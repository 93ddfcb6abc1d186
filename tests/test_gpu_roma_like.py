"""RoMa-shaped inputs at FULL size: the certainty ``sigmoid(confidence)`` gives - bimodal, coherent regions near 0.02 and near 0.98, so
that after the 0.2 floor and the 0.9 cap the arg-max and the coverage walk are decided by ties nearly everywhere - a third of the grid warped past the
neighbour's edge (upstream does not reject out-of-range B coordinates, core/pipeline.py:697-703: the reprojection test has to), occlusion-style depth
discontinuities, and the `no_filter` branch at 512^2.  Through the aggregate kernel, the fused dense kernel (every cell against the oracle, every flip
classified), the fused sampled call (selection against the oracle's rule, core/sampling.py:13-53) and dense `no_filter`.  ``pytest -m gpu``."""
import numpy as np
import pytest
import torch

import lichtfeld_densification_plugin_amd as lfd
from lichtfeld_densification_plugin_amd import synthetic
from lichtfeld_densification_plugin_amd.core import hip_backend as hb
from helpers import flip_report, oracle_cam, orc

pytestmark = pytest.mark.gpu
H = W = 512
K = 3


@pytest.fixture(scope="module")
def scene():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    dev = torch.device("cuda:0")
    cams = synthetic.ring_cameras(185, seed=0)
    ref = 121
    nbrs = synthetic.ring_neighbours(185, ref, K)
    s = synthetic.synth_reference(cams, ref, nbrs, H, W, W, H, noise_px=0.5, outlier_frac=0.05, channels=2, seed=777, cert_mode="bimodal", out_of_range=0.33,
                                  occlusion_steps=True)
    r = hb.ReferenceInputs(ref_cam=ref, nbr_cams=nbrs, cert=[s.cert[j].to(dev) for j in range(K)], warp=[s.warp[j].contiguous().to(dev) for j in range(K)],
                           image=s.image.to(dev))
    dens = hb.HipDensifier(dev)
    dens.upload_cameras(cams)
    certs, warps = [s.cert[j].numpy() for j in range(K)], [s.warp[j].numpy() for j in range(K)]
    with np.errstate(all="ignore"):
        best, bk, agg = orc.prepare_reference(certs, warps, orc.OracleParams())
    yield dict(dev=dev, cams=cams, ref=ref, nbrs=nbrs, s=s, r=r, dens=dens, best=best, bk=bk, agg=agg, certs=certs, warps=warps)
    dens.close()


def test_the_input_has_the_matchers_pathologies(scene):
    c = np.stack(scene["certs"])
    w = np.stack(scene["warps"])
    assert (c < 0.05).mean() > 0.25 and (c > 0.95).mean() > 0.15 and ((c > 0.2) & (c < 0.9)).mean() < 0.3                      # bimodal
    assert (scene["best"] == np.float32(0.2)).mean() > 0.1                             # an eighth of the cells: EVERY neighbour on the floor - a pure tie
    assert (np.minimum(scene["best"], np.float32(0.9)) == np.float32(0.9)).mean() > 0.6      # ... and two thirds tied on the cap
    out = (np.abs(w[..., 0]) > 1.0) | (np.abs(w[..., 1]) > 1.0)
    assert 0.3 < out.mean() < 0.5                                                       # a third of the warp leaves the neighbour
    near_border = ((np.abs(w[..., 0]) > 0.98) & (np.abs(w[..., 0]) < 1.02)).mean()
    assert near_border > 0.002                                                          # ... and crosses its border on the way
    jumps = np.abs(np.diff(w[0, :, : int(W * 0.6), 0], axis=1)) > 20.0 / W
    assert jumps.mean() > 0.002                                                         # occlusion edges: the warp jumps by many cells


def test_aggregate_of_a_bimodal_field_is_bit_exact(scene):
    dens, r = scene["dens"], scene["r"]
    best, slot = dens.aggregate(hb.PreparedBatch([r], W, H, cameras=scene["cams"]), hb.make_params(lfd.DensePipelineConfig(output_path="")))
    np.testing.assert_array_equal(best[0].cpu().numpy(), scene["best"])
    np.testing.assert_array_equal(slot[0].cpu().numpy().astype(np.int64), np.asarray(scene["bk"]).reshape(H, W))      # first maximum wins


def test_dense_kernel_every_cell_every_flip_in_band(scene):
    dens, r, s, cams = scene["dens"], scene["r"], scene["s"], scene["cams"]
    out = dens.triangulate_dense(hb.PreparedBatch([r], W, H, cameras=cams), hb.make_params(lfd.DensePipelineConfig(output_path="")))
    axes = (orc.identity_axis_scalar(W), orc.identity_axis_scalar(H))
    rep = flip_report(out.cell.cpu().numpy().astype(np.int64), s, cams, W, H, orc.OracleParams(), axes)
    print(f"[guard band] fast_k3_roma_like: cells {rep['cells']} flipped {rep['flipped']} out_of_band {rep['out_of_band']} by reason {rep['by_reason']}")
    assert rep["out_of_band"] == 0, rep["oob_cells"][:8]
    assert rep["flipped"] <= 1.5e-4 * rep["cells"] + 4
    cell = out.cell.cpu().numpy().astype(np.int64)
    np.testing.assert_array_equal(out.slot.cpu().numpy().astype(np.int64), np.asarray(scene["bk"]).reshape(-1)[cell])
    # the third that left the neighbour: upstream converts such coordinates like any other (core/pipeline.py:697-703) and only its geometric tests
    # decide - a shift ALONG the epipolar line is a consistent correspondence at another depth and survives, there as here (the flip report above
    # compared exactly these cells too); what must hold is that the kernel keeps none of them that upstream drops, and the other way round
    w = np.stack(scene["warps"])
    bk = np.asarray(scene["bk"]).reshape(-1)
    wx = np.take_along_axis(w[..., 0].reshape(K, -1), bk[None], 0)[0]
    wy = np.take_along_axis(w[..., 1].reshape(K, -1), bk[None], 0)[0]
    far = (np.abs(wx) > 1.05) | (np.abs(wy) > 1.05)
    keep_hip = np.zeros(H * W, bool); keep_hip[cell] = True
    keep_orc = np.zeros(H * W, bool); keep_orc[rep["oracle"].cell] = True
    assert far.mean() > 0.25 and (keep_hip[far] != keep_orc[far]).sum() <= rep["flipped"]
    print(f"[roma-like] {int(far.sum())} cells warped past the neighbour's edge, {int(keep_orc[far].sum())} of them survive upstream's tests, {int(keep_hip[far].sum())} the kernel's")
    # the common survivors carry upstream's values
    res = rep["oracle"]
    common, ih, io_ = np.intersect1d(cell, res.cell, return_indices=True)
    assert common.size > 50000
    np.testing.assert_allclose(out.xyz.cpu().numpy()[ih], res.xyz[io_], rtol=1e-5, atol=1e-6)


def test_fused_sampled_call_on_a_bimodal_field(scene):
    """the selection is unique(draw, lowest-index maximum of every tile with a positive weight) (tests/test_gpu_beta.py states the rule); with
    a third of the map on the cap the coverage pass is decided by the tie rule in most tiles"""
    dens, r, s, cams, ref, nbrs = (scene[k_] for k_ in ("dens", "r", "s", "cams", "ref", "nbrs"))
    M, cap, border, tiles = 10000, 0.9, 2, 24
    cfg = lfd.DensePipelineConfig(output_path="", matches_per_ref=M)
    batch = hb.PreparedBatch([r], W, H, cameras=cams)
    dens.seed_rng(0)
    cells_t = torch.zeros((M + tiles * tiles + 64,), dtype=torch.int64, device=scene["dev"])
    outb = hb.OutputBuffers(M + tiles * tiles + 64, 1, K, scene["dev"])
    dens.launch_sampled(batch, hb.make_params(cfg), M, outb, cap=cap, border=border, tiles=tiles, sel_cells=cells_t)
    out = outb.collect(indexed=True, check_selection=True)
    sel = cells_t[:out.n_selected].cpu().numpy()
    best = scene["best"]
    cert = np.minimum(best, np.float32(cap))
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    inside = (xx >= border) & (xx <= W - 1 - border) & (yy >= border) & (yy <= H - 1 - border)
    wts = (cert * inside.astype(np.float32)).reshape(-1).astype(np.float32)
    wn = (wts / np.float32(wts.astype(np.float64).sum())).astype(np.float32)          # the device's normaliser: the correctly rounded exact sum
    idx_main = np.random.RandomState(0).choice(wn.size, size=int(M * 0.85), replace=False, p=wn)
    tile = max(1, W // tiles)
    bins = ((xx // tile) * 100000 + (yy // tile)).reshape(-1)
    order = np.lexsort((np.arange(wn.size), -wn.astype(np.float64)))
    _, first = np.unique(bins[order], return_index=True)
    cov = order[first]
    cov = cov[wn[cov] > 0]
    np.testing.assert_array_equal(sel, np.unique(np.concatenate([idx_main, cov])))
    tied_tiles = sum(1 for c in cov if (wn[bins == bins[c]] == wn[c]).sum() > 1)
    assert tied_tiles > 0.5 * cov.size                                                # most tiles are decided by the tie rule
    params = orc.OracleParams(matches_per_ref=M)
    axes = (orc.identity_axis_scalar(W), orc.identity_axis_scalar(H))
    with np.errstate(all="ignore"):
        res = orc.triangulate_selected(sel, best, scene["bk"], scene["agg"], s.image.numpy(), oracle_cam(cams[ref]), [oracle_cam(cams[n]) for n in nbrs], W, H,
                                       params, axes=axes)
    assert abs(out.count - res.count) <= 3
    common, ih, io_ = np.intersect1d(out.cell.cpu().numpy(), res.cell, return_indices=True)
    assert common.size >= res.count - 3
    np.testing.assert_allclose(out.xyz.cpu().numpy()[ih], res.xyz[io_], rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(out.rgb.cpu().numpy()[ih], res.rgb[io_])


def test_dense_no_filter_at_full_size_against_the_oracle(scene):
    """`no_filter` keeps every finite point (core/pipeline.py:739-743): at 512^2, on the RoMa-like input - garbage correspondences included - the kept
    set is the oracle's wherever the oracle's own result is finite with a margin, and on every WELL-POSED common cell (sv_ratio < 0.2: an
    ill-conditioned 4x4 system is ill-conditioned for any solver) positions, colours and errors are upstream's within the stated tolerances."""
    dens, r, s, cams, ref, nbrs = (scene[k_] for k_ in ("dens", "r", "s", "cams", "ref", "nbrs"))
    cfg = lfd.DensePipelineConfig(output_path="", no_filter=True)
    out = dens.triangulate_dense(hb.PreparedBatch([r], W, H, cameras=cams), hb.make_params(cfg))
    params = orc.OracleParams(no_filter=True)
    axes = (orc.identity_axis_scalar(W), orc.identity_axis_scalar(H))
    ca, cbs = oracle_cam(cams[ref]), [oracle_cam(cams[n]) for n in nbrs]
    with np.errstate(all="ignore"):
        d = orc.triangulate_dense(scene["certs"], scene["warps"], s.image.numpy(), ca, cbs, W, H, params, axes=axes)
        diag = orc.cell_diagnostics(np.arange(H * W), scene["bk"], scene["agg"], ca, cbs, W, H, axes=axes)
    cell = out.cell.cpu().numpy().astype(np.int64)
    assert np.all(np.diff(cell) > 0)
    keep_hip = np.zeros(H * W, bool); keep_hip[cell] = True
    keep_orc = np.zeros(H * W, bool); keep_orc[d["cell"]] = True
    # finite on one side, not on the other: only where a coordinate or the error overflows f32 by rounding (|value| next to 3.4e38) - count them
    differ = keep_hip != keep_orc
    assert differ.sum() <= 1e-4 * H * W, int(differ.sum())
    both = keep_hip & keep_orc & (diag["sv_ratio"] < 0.2)
    assert both.sum() > 0.5 * H * W
    idx = np.nonzero(both)[0]
    ph, po = np.searchsorted(cell, idx), np.searchsorted(d["cell"], idx)
    xyz, rgb, err = out.xyz.cpu().numpy(), out.rgb.cpu().numpy(), out.err.cpu().numpy()
    noise = diag["err_noise"][idx].astype(np.float64)
    tol_x = 1e-6 + 1e-5 * np.abs(d["xyz"][po]) + (4.0 * noise / 900.0)[:, None] * np.abs(d["xyz"][po]).max(axis=1, keepdims=True)
    bad = np.abs(xyz[ph] - d["xyz"][po]) > tol_x
    assert bad.any(axis=1).mean() <= 1e-4, (int(bad.any(axis=1).sum()), int(both.sum()))
    np.testing.assert_allclose(rgb[ph], d["rgb"][po], rtol=0, atol=1.0 / 255.0 / 4.0)
    tol_e = 1e-3 + 1e-4 * np.abs(d["err"][po].astype(np.float64)) + 4.0 * noise
    assert (np.abs(err[ph].astype(np.float64) - d["err"][po].astype(np.float64)) > tol_e).mean() <= 1e-4
