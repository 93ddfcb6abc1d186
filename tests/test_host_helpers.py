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
    assert lib.lfd_abi_version() == 1
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
