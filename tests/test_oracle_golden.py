"""The oracle must reproduce the vectors captured from the upstream code bit for bit (CPU only).

These are the pins that make the oracle trustworthy; the GPU parity tests then compare the HIP path
against the oracle.  Bit equality here relies on the same NumPy/LAPACK/torch builds that produced
the fixtures (recorded in each file under ``versions``); on a different build the comparison falls
back to tolerances and says so."""
import json

import numpy as np
import pytest
import torch

from helpers import g3_case, oracle_cams, orc


def _same_build(g):
    v = json.loads(str(g["versions"]))
    return (v["numpy"] == np.__version__ and v["torch"] == torch.__version__
            and v["cpu_capability"] == torch.backends.cpu.get_cpu_capability())


def _eq(a, b, exact, rtol=1e-5, atol=1e-6):
    if exact:
        np.testing.assert_array_equal(a, b)
    else:
        np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def test_g1_geometry(g1):
    exact = _same_build(g1)
    for pi in range(int(g1["n_pairs"])):
        pre = f"p{pi}_"
        ca, cb = oracle_cams(g1, pre + "cam_")
        uv1, uv2 = g1[pre + "uv1"], g1[pre + "uv2"]
        F = orc.fundamental_matrix(ca.K, ca.R, ca.t, cb.K, cb.R, cb.t)
        assert F.dtype == np.float32
        _eq(F, g1[pre + "F"], exact)
        se = orc.sampson_error(g1[pre + "F"], uv1, uv2)
        assert se.dtype == np.float64
        _eq(se, g1[pre + "sampson"], exact, rtol=1e-12)
        with np.errstate(all="ignore"):
            X = orc.dlt_triangulate(ca.P, cb.P, uv1, uv2)
            _eq(X, g1[pre + "X"], exact, rtol=1e-3)
            _eq(orc.dlt_triangulate(ca.P, cb.P, uv1[:1], uv2[:1]), g1[pre + "X_single"], exact, rtol=1e-3)
            Xg = g1[pre + "X"]
            _eq(orc.reprojection_error(ca.P, Xg, uv1), g1[pre + "err1"], exact, atol=1e-3)
            _eq(orc.reprojection_error(cb.P, Xg, uv2), g1[pre + "err2"], exact, atol=1e-3)
            np.testing.assert_array_equal(orc.depth_positive(ca.P, Xg), g1[pre + "cheir1"])
            np.testing.assert_array_equal(orc.depth_positive(cb.P, Xg), g1[pre + "cheir2"])
            np.testing.assert_array_equal(orc.parallax_ok(ca.C, cb.C, Xg, 0.5), g1[pre + "parallax"])
    ca, cb = oracle_cams(g1, "p0_cam_")
    with np.errstate(all="ignore"):
        X = orc.dlt_triangulate(ca.P, cb.P, g1["odd_uv1"], g1["odd_uv2"])
        _eq(X, g1["odd_X"], exact, rtol=1e-3)
        np.testing.assert_array_equal(orc.depth_positive(ca.P, g1["odd_X"]), g1["odd_cheir1"])
        np.testing.assert_array_equal(orc.depth_positive(cb.P, g1["odd_X"]), g1["odd_cheir2"])


def _tiefree(h, w, seed):
    rs = np.random.RandomState(seed)
    perm = rs.permutation(h * w).astype(np.float64)
    return (0.2 + 0.7 * (perm + 0.5) / (h * w)).astype(np.float32).reshape(h, w)


def test_g2_selection(g2):
    import hashlib
    for ci, (h, w, M, seed, cseed) in enumerate(g2["cases"]):
        cert = _tiefree(int(h), int(w), int(cseed))
        assert hashlib.sha256(cert.tobytes()).hexdigest() == str(g2[f"c{ci}_cert_sha256"])
        for no_filter in (False, True):
            key = f"c{ci}_{'nf' if no_filter else 'f'}_"
            rng = np.random.RandomState(int(seed))
            sel = orc.select_samples(cert, int(M), cap=0.9, border=2, tiles=24, no_filter=no_filter, rng=rng)
            np.testing.assert_array_equal(sel, g2[key + "sel"])
            assert int(rng.get_state()[2]) == int(g2[key + "mt_pos"])
            np.testing.assert_array_equal(rng.random_sample(2), g2[key + "next_doubles"])
    rng = np.random.RandomState(3)
    sel = orc.select_samples(g2["ties_cert"], 1200, rng=rng)
    # massive ties: argsort tie order is build dependent; the random-draw part and the size are not
    assert sel.size == g2["ties_sel"].size or not _same_build(g2)
    if _same_build(g2):
        np.testing.assert_array_equal(sel, g2["ties_sel"])
    assert orc.select_samples(np.zeros((16, 16), np.float32), 100).size == 0
    assert int(g2["too_many_raises"]) == 1
    with pytest.raises(ValueError):
        orc.select_samples(_tiefree(64, 64, 1), 10000, rng=np.random.RandomState(0))


@pytest.mark.parametrize("name", ["a_filter_k3", "b_nofilter_k1", "c_rect_k3", "d_hires_k2", "e_masks_k3",
                                  "f_nosampson_k4"])
def test_g3_triangulate_reference(g3, name):
    exact = _same_build(g3)
    cams = oracle_cams(g3)
    c = g3_case(g3, name)
    cert_list = [c["cert"][j] for j in range(c["k"])]
    warp_list = [c["warp"][j] for j in range(c["k"])]
    # P1: prologue reproduces the maps upstream handed to _triangulate_ref
    post = [orc.certainty_prologue(cert_list[j], warp_list[j], c["params"].certainty_thresh, c["mask_a"],
                                   None if c["masks_b"] is None else c["masks_b"][j]) for j in range(c["k"])]
    np.testing.assert_array_equal(np.stack(post), c["post_cert"])
    rng = np.random.RandomState(7)
    with np.errstate(all="ignore"):
        res, sel = orc.triangulate_reference(cert_list, warp_list, c["image"], cams[c["ref"]],
                                             [cams[n] for n in c["nbrs"]], c["w_match"], c["h_match"], c["params"],
                                             rng=rng, mask_a=c["mask_a"], mask_b_list=c["masks_b"])
    np.testing.assert_array_equal(sel, c["sel"])
    assert not c["none"]
    assert [c["nbrs"][s.nbr_slot] for s in res.segments] == [int(v) for v in c["seg_nbr_cam"]]
    assert [s.xyz.shape[0] for s in res.segments] == [int(v) for v in c["seg_count"]]
    _eq(res.xyz, c["xyz"], exact, rtol=1e-4)
    _eq(res.rgb, c["rgb"], exact)
    _eq(res.err, c["err"], exact, atol=2e-3)
    assert res.xyz.dtype == np.float32 and res.rgb.dtype == np.float32 and res.err.dtype == np.float32
    _eq(np.concatenate([s.matches_px for s in res.segments]), c["dbg_matches"], exact)
    _eq(np.concatenate([s.cert_norm for s in res.segments]), c["dbg_cert"], exact)


def test_identity_axis_scalar_form_within_one_ulp_of_one_of_torch():
    for n in (1, 2, 3, 48, 64, 320, 512, 640, 960, 1280, 77):
        a = orc.identity_axis_scalar(n)
        b = torch.linspace(-1 + 1 / n, 1 - 1 / n, n).numpy()
        assert np.all(np.abs(a - b) <= 2.0 ** -24)      # one ulp just below 1.0


def test_mask_ops_match_torch():
    import torch.nn.functional as F
    rs = np.random.RandomState(0)
    m = (rs.uniform(size=(40, 56)) > 0.4).astype(np.uint8)
    for hw in ((40, 56), (60, 84), (96, 96), (17, 23)):
        t = F.interpolate(torch.from_numpy(m.astype(np.float32)).view(1, 1, 40, 56), size=hw, mode="nearest")[0, 0]
        np.testing.assert_array_equal(orc.nearest_resize_mask(m, hw), t.numpy())
    g = rs.uniform(-1.2, 1.2, size=(30, 30, 2)).astype(np.float32)
    t = F.grid_sample(torch.from_numpy(m.astype(np.float32)).view(1, 1, 40, 56), torch.from_numpy(g).unsqueeze(0),
                      mode="nearest", padding_mode="zeros", align_corners=False)[0, 0].numpy()
    mine = orc.warp_mask_nearest(m.astype(np.float32), g[..., 0], g[..., 1])
    # torch's CPU kernel un-normalises as (g+1)*(size/2)-0.5, the device kernel as ((g+1)*size-1)/2;
    # they can differ by one ulp at a rounding boundary
    assert (mine != t).mean() < 1e-3


def test_g5_writers(g5):
    u8 = orc.to_uint8_rgb(g5["rgb"])
    np.testing.assert_array_equal(u8, g5["rgb_u8"])
    assert orc.ply_bytes(g5["xyz"], u8) == g5["ply"].tobytes()
    assert orc.points3d_bin_bytes(g5["xyz"], u8, g5["err"]) == g5["points3d_bin"].tobytes()
