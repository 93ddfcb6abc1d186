"""A realistic certainty field at FULL size: iid Beta(2,2) certainties - after the 0.2 floor and the 0.9 cap a tenth of
the cells of every neighbour sit exactly ON a clamp, so the arg-max over the neighbours and the coverage pass of the selection are decided by
TIES "like real data" (synthetic.py, cert_mode="beta") - through the aggregate kernel, the fused dense kernel and the fused sampled call,
against the oracle.  ``pytest -m gpu``."""
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
    ref = 77
    nbrs = synthetic.ring_neighbours(185, ref, K)
    s = synthetic.synth_reference(cams, ref, nbrs, H, W, W, H, noise_px=0.5, outlier_frac=0.05, channels=2, seed=4242, cert_mode="beta")
    r = hb.ReferenceInputs(ref_cam=ref, nbr_cams=nbrs, cert=[s.cert[j].to(dev) for j in range(K)], warp=[s.warp[j].contiguous().to(dev) for j in range(K)],
                           image=s.image.to(dev))
    dens = hb.HipDensifier(dev)
    dens.upload_cameras(cams)
    certs, warps = [s.cert[j].numpy() for j in range(K)], [s.warp[j].numpy() for j in range(K)]
    with np.errstate(all="ignore"):
        best, bk, agg = orc.prepare_reference(certs, warps, orc.OracleParams())
    yield dict(dev=dev, cams=cams, ref=ref, nbrs=nbrs, s=s, r=r, dens=dens, best=best, bk=bk, agg=agg)
    dens.close()


def test_the_field_really_is_full_of_ties(scene):
    c = np.stack([scene["s"].cert[j].numpy() for j in range(K)])
    floored = (c < 0.2).mean()
    assert 0.08 < floored < 0.13                                    # Beta(2,2): P(c < 0.2) = 0.104
    all_floored = (c < 0.2).all(0).mean()                           # cells where EVERY neighbour sits on the floor: the arg-max is a pure tie
    assert all_floored > 5e-4
    capped = (np.minimum(scene["best"], np.float32(0.9)) == np.float32(0.9)).mean()
    assert capped > 0.05                                            # the best certainty of a twentieth of the cells sits on the cap: tied weights


def test_aggregate_with_ties_is_bit_exact(scene):
    dens, r = scene["dens"], scene["r"]
    best, slot = dens.aggregate(hb.PreparedBatch([r], W, H, cameras=scene["cams"]), hb.make_params(lfd.DensePipelineConfig(output_path="")))
    np.testing.assert_array_equal(best[0].cpu().numpy(), scene["best"])
    np.testing.assert_array_equal(slot[0].cpu().numpy().astype(np.int64), np.asarray(scene["bk"]).reshape(H, W))      # first maximum wins


def test_dense_kernel_with_ties_every_flip_in_band(scene):
    dens, r, s, cams = scene["dens"], scene["r"], scene["s"], scene["cams"]
    out = dens.triangulate_dense(hb.PreparedBatch([r], W, H, cameras=cams), hb.make_params(lfd.DensePipelineConfig(output_path="")))
    axes = (orc.identity_axis_scalar(W), orc.identity_axis_scalar(H))
    rep = flip_report(out.cell.cpu().numpy().astype(np.int64), s, cams, W, H, orc.OracleParams(), axes)
    print(f"[guard band] fast_k3_beta: cells {rep['cells']} flipped {rep['flipped']} out_of_band {rep['out_of_band']} by reason {rep['by_reason']}")
    assert rep["out_of_band"] == 0, rep["oob_cells"][:8]
    assert rep["flipped"] <= 1.5e-4 * rep["cells"] + 4
    # the winning slot of every survivor is the oracle's (ties included)
    cell = out.cell.cpu().numpy().astype(np.int64)
    np.testing.assert_array_equal(out.slot.cpu().numpy().astype(np.int64), np.asarray(scene["bk"]).reshape(-1)[cell])


def test_fused_sampled_call_with_ties(scene):
    """The weighted draw does not depend on tie order; the coverage pass does (upstream walks ``np.argsort(-weights)``, whose order among equal
    weights is unspecified).  The device's rule is the LOWEST cell index among the equal maxima of a tile, so its selection is exactly
    unique(draw, lowest-index maximum of every tile with a positive weight) - and what it triangulates from that selection is the oracle's."""
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
    # expected selection
    best = scene["best"]
    cert = np.minimum(best, np.float32(cap))
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    inside = (xx >= border) & (xx <= W - 1 - border) & (yy >= border) & (yy <= H - 1 - border)
    wts = (cert * inside.astype(np.float32)).reshape(-1).astype(np.float32)
    s_exact = np.float32(wts.astype(np.float64).sum())              # the device's normaliser: the correctly rounded exact sum
    wn = (wts / s_exact).astype(np.float32)
    idx_main = np.random.RandomState(0).choice(wn.size, size=int(M * 0.85), replace=False, p=wn)
    tile = max(1, W // tiles)
    bins = ((xx // tile) * 100000 + (yy // tile)).reshape(-1)
    order = np.lexsort((np.arange(wn.size), -wn.astype(np.float64)))    # descending weight, ties by ascending cell index
    _, first = np.unique(bins[order], return_index=True)
    cov = order[first]
    cov = cov[wn[cov] > 0]
    expect = np.unique(np.concatenate([idx_main, cov]))
    np.testing.assert_array_equal(sel, expect)
    assert cov.size == ((W - 1) // tile + 1) * ((H - 1) // tile + 1)
    # ... and the tie rule is observable: some tiles hold several cells at the capped weight
    tied_tiles = sum(1 for c in cov if (wn[bins == bins[c]] == wn[c]).sum() > 1)
    assert tied_tiles > 10
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
