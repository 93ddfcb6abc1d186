"""CPU-only checks of the C-ABI library: it loads, exports every declared symbol, and the host build
of the kernels' per-correspondence source agrees with the oracle / golden vectors.  No compute entry
point is called (there is no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import lichtfeld_densification_plugin_amd as lfd
from lichtfeld_densification_plugin_amd.core import hip_backend as hb
from helpers import ROOT, oracle_cams, orc


@pytest.fixture(scope="module")
def lib():
    return hb.load_library()


def test_library_exports_every_declared_symbol(lib):
    header = open(os.path.join(ROOT, "include", "lfd_densify.h")).read()
    names = sorted(set(re.findall(r"\b(lfd_[a-z_0-9]+)\s*\(", header)))
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/lfd_densify.h but not exported"
    assert lib.lfd_abi_version() == hb.LFD_ABI_VERSION == 9
    assert ctypes.sizeof(hb.lfd_params) == 32


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only behaviour")
def test_no_gpu_fails_loudly():
    with pytest.raises(hb.HipBackendError):
        hb.HipDensifier()
    ctx = ctypes.c_void_p()
    rc = hb.load_library().lfd_create(0, None, ctypes.byref(ctx))
    assert rc == 2 and not ctx.value
    assert b"HIP" in hb.load_library().lfd_last_error(None) or b"device" in hb.load_library().lfd_last_error(None)


@pytest.mark.parametrize("n", [1, 2, 3, 48, 77, 320, 512, 640, 960, 1280])
def test_identity_axis(n):
    np.testing.assert_array_equal(hb.identity_axis(n), orc.identity_axis_scalar(n))


def test_parallax_dot_threshold_near_90_degrees():
    # around dot = 0 the f32 grid is far finer than acos' own resolution: libm and NumPy may place the
    # boundary a few 1e-8 apart, orders of magnitude below the rounding noise of a 3-term f32 dot product
    d = hb.parallax_dot_threshold(90.0)
    assert abs(d) < 1e-6


@pytest.mark.parametrize("min_deg", [0.05, 0.5, 1.0, 2.5, 30.0, 179.0])
def test_parallax_dot_threshold_is_the_numpy_boundary(min_deg):
    d = np.float32(hb.parallax_dot_threshold(min_deg))
    up = np.nextafter(d, np.float32(2.0))
    ang = lambda x: np.degrees(np.arccos(np.clip(np.float32(x), -1.0, 1.0)))
    assert ang(d).dtype == np.float32
    assert ang(d) >= np.float32(min_deg)
    assert not (ang(up) >= np.float32(min_deg))


def test_parallax_threshold_extremes():
    assert hb.parallax_dot_threshold(0.0) == 1.0
    assert hb.parallax_dot_threshold(181.0) == -2.0


def test_fundamental_matches_numpy(g1):
    exact = 0
    total = 0
    for pi in range(int(g1["n_pairs"])):
        ca, cb = oracle_cams(g1, f"p{pi}_cam_")
        F = hb.host_fundamental(ca.K, ca.R, ca.t, cb.K, cb.R, cb.t)
        Fg = g1[f"p{pi}_F"]
        scale = np.abs(Fg).max()
        assert np.abs(F - Fg).max() <= 2e-6 * scale      # f32 rounding-order noise only
        exact += int((F == Fg).sum())
        total += 9
    assert exact >= total // 3


def _cam_record(oc, uid=0):
    return lfd.CameraRecord(uid=uid, image_path="", width=oc.width, height=oc.height, K=oc.K, R=oc.R, t=oc.t, P=oc.P, C=oc.C)


def test_python_fundamental_is_upstream_bit_for_bit(g1):
    """core.hip_backend.fundamental_from_world2cam repeats upstream's NumPy calls (core/geometry.py:122-130): on the same
    machine it returns the captured F bit for bit; it is what PreparedBatch(cameras=...) hands to the kernels."""
    for pi in range(int(g1["n_pairs"])):
        ca, cb = oracle_cams(g1, f"p{pi}_cam_")
        F = hb.fundamental_from_world2cam(ca.K, ca.R, ca.t, cb.K, cb.R, cb.t)
        assert F.dtype == np.float32
        np.testing.assert_array_equal(F, g1[f"p{pi}_F"])


def test_host_eval_matches_golden_geometry(g1):
    """Feed camera pixels through the per-cell routine (identity pixel mapping: cameras sized like the
    match grid so that sx = sy = 1) and compare with upstream's X / errors."""
    cfg = lfd.DensePipelineConfig(output_path="", sampson_thresh=5.0, reproj_thresh=0.8, min_parallax_deg=0.5)
    params = hb.make_params(cfg)
    n_checked = 0
    for pi in (0, 1, 2, 4):
        pre = f"p{pi}_"
        ca, cb = oracle_cams(g1, pre + "cam_")
        # the routine converts normalised -> match px -> camera px; invert that so it lands on uv
        wm, hm = ca.width, ca.height
        uv1, uv2 = g1[pre + "uv1"], g1[pre + "uv2"]
        sA = (np.float32(ca.width / float(wm)), np.float32(ca.height / float(hm)))
        assert sA == (1.0, 1.0)
        Xg = g1[pre + "X"]
        err_g = np.maximum(g1[pre + "err1"], g1[pre + "err2"])
        keep_g = ((g1[pre + "sampson"] < 5.0) & (err_g <= np.float32(0.8)) & g1[pre + "cheir1"] & g1[pre + "cheir2"]
                  & g1[pre + "parallax"])
        for i in range(0, uv1.shape[0], 3):
            def to_norm(px, size):
                return np.float32(np.float64(px) / (0.5 * (size - 1)) - 1.0)
            xa, ya = to_norm(uv1[i, 0], wm), to_norm(uv1[i, 1], hm)
            xb, yb = to_norm(uv2[i, 0], wm), to_norm(uv2[i, 1], hm)
            # only use points whose round trip reproduces the golden f32 pixel exactly
            back = [orc.match_pixels(np.float32(v), s) for v, s in ((xa, wm), (ya, hm), (xb, wm), (yb, hm))]
            if not (back[0] == uv1[i, 0] and back[1] == uv1[i, 1] and back[2] == uv2[i, 0] and back[3] == uv2[i, 1]):
                continue
            out = hb.host_eval_correspondence(_cam_record(ca), _cam_record(cb), xa, ya, xb, yb, wm, hm, params)
            se = g1[pre + "sampson"][i]
            if se < 5.0 - 1e-6:
                n_checked += 1
                np.testing.assert_allclose(out[:3], Xg[i, :3], rtol=2e-5, atol=1e-6)
                assert abs(out[6] - err_g[i]) <= 1e-3
                near = (abs(err_g[i] - 0.8) < 2e-3)
                if not near:
                    assert bool(out[7]) == bool(keep_g[i])
            elif se > 5.0 + 1e-6:
                assert out[7] == 0.0
    assert n_checked > 100


# ---- the triangulation solver (csrc/lfd_geometry.hpp::lfd_null_vector) against an f64 SVD -----------------------
def _dlt_matrix(rng, cams, noise_px):
    i = rng.randint(len(cams))
    j = (i + rng.randint(1, 4)) % len(cams)
    P1, P2 = np.asarray(cams[i].P, np.float64), np.asarray(cams[j].P, np.float64)
    X = np.array([rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(-0.3, 0.5), 1.0])
    u1 = (P1 @ X)[:2] / (P1 @ X)[2] + rng.normal(0, noise_px, 2)
    u2 = (P2 @ X)[:2] / (P2 @ X)[2] + rng.normal(0, noise_px, 2)
    u1, u2, P1f, P2f = u1.astype(np.float32), u2.astype(np.float32), P1.astype(np.float32), P2.astype(np.float32)
    return np.stack([u1[0] * P1f[2] - P1f[0], u1[1] * P1f[2] - P1f[1], u2[0] * P2f[2] - P2f[0], u2[1] * P2f[2] - P2f[1]]).astype(np.float32)


@pytest.mark.parametrize("noise_px", [0.0, 0.1, 0.5, 2.0, 10.0])
def test_null_vector_matches_f64_svd(noise_px):
    """DLT matrices of the synthetic ring scene (upstream core/geometry.py:72-75 rows) at several noise levels,
    i.e. from rank-deficient (sigma4 ~ 0) to sigma4/sigma3 close to 1: the direction agrees with numpy's f64 SVD
    far more closely than the f32 LAPACK SVD upstream calls can resolve (~1e-7 x sigma1/sigma3)."""
    from lichtfeld_densification_plugin_amd import synthetic
    cams = synthetic.ring_cameras(60, seed=0)
    rng = np.random.RandomState(int(noise_px * 10) + 3)
    worst, max_it = 0.0, 0
    for _ in range(1500):
        A = _dlt_matrix(rng, cams, noise_px)
        x, it = hb.host_null_vector(A)
        _, S, Vt = np.linalg.svd(A.astype(np.float64))
        v = Vt[-1]
        xn = x / np.linalg.norm(x)
        err = min(np.linalg.norm(xn - v), np.linalg.norm(xn + v))
        ratio = S[3] / S[2]
        # the vector is determined to ~eps / (1 - ratio^2); allow 2e-6 up to ratio 0.9, skip the near-degenerate rest
        if ratio < 0.9:
            worst = max(worst, err)
            assert err < 2e-6, (err, ratio, it)
        if ratio < 0.1:
            assert err < 1e-8, (err, ratio, it)
        max_it = max(max_it, it)
    assert max_it <= 40 and worst < 2e-6


def test_null_vector_degenerate_inputs_terminate():
    """Rank-deficient, zero, huge, tiny and non-finite matrices: the routine returns (no hang), and garbage in
    gives NaN or a finite vector, never an exception (the kernels drop such cells through their finite / error tests)."""
    rng = np.random.RandomState(0)
    cases = [np.zeros((4, 4), np.float32), np.eye(4, dtype=np.float32), np.ones((4, 4), np.float32),
             (rng.randn(4, 4) * 1e18).astype(np.float32), (rng.randn(4, 4) * 1e-18).astype(np.float32)]
    bad = rng.randn(4, 4).astype(np.float32)
    bad[1, 2] = np.nan
    cases.append(bad)
    inf = rng.randn(4, 4).astype(np.float32)
    inf[0, 0] = np.inf
    cases.append(inf)
    rank2 = rng.randn(4, 4).astype(np.float32)
    rank2[2] = rank2[0]
    rank2[3] = rank2[1]
    cases.append(rank2)
    for A in cases:
        x, it = hb.host_null_vector(A)
        assert 3 <= it <= 40 and x.shape == (4,)
    # an exactly singular matrix with a known null vector
    A = rng.randn(4, 4)
    n = np.array([0.3, -0.5, 0.2, 0.7])
    A = (A - np.outer(A @ n, n) / (n @ n)).astype(np.float32)
    x, _ = hb.host_null_vector(A)
    _, _, Vt = np.linalg.svd(A.astype(np.float64))
    xn = x / np.linalg.norm(x)
    assert min(np.linalg.norm(xn - Vt[-1]), np.linalg.norm(xn + Vt[-1])) < 1e-9


def test_markstein_quotient_is_the_correctly_rounded_division():
    """The selection kernels normalise the cumulative sum with q' = RN(q + r (a - b q)), r = RN(1/b), q = RN(a r)
    instead of 262144 IEEE divisions per pass (csrc/lfd_select.hip).  That quotient must BE RN(a / b) - NumPy's
    ``cdf /= cdf[-1]`` - for the operands that occur there: 0 <= a <= b, b = a sum of f32 probabilities near 1 or
    below.  Checked in exact rational arithmetic."""
    import random
    from fractions import Fraction as F
    rnd = random.Random(7)

    def total():
        c = rnd.random()
        if c < 0.3:
            return 1.0 + rnd.randint(-2000, 2000) * 2.0 ** -52 * rnd.randint(1, 1 << 20)
        if c < 0.6:
            return rnd.uniform(0.5, 1.0)
        return rnd.uniform(1e-3, 1.0)
    for _ in range(20000):
        b = total()
        a = min(rnd.random() * b, b)
        if rnd.random() < 0.2:
            a = min(float(F(rnd.randint(0, 1 << 52), 1 << 52)) * b, b)
        r = float(F(1) / F(b))                       # RN(1/b)
        q = a * r                                    # RN(a r)
        rem = float(F(a) - F(b) * F(q))              # fma(-b, q, a): exact
        q2 = float(F(q) + F(rem) * F(r))             # fma(rem, r, q): one rounding
        assert q2 == float(F(a) / F(b)), (a, b)
