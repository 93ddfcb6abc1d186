"""Property tests (tier iii) on the CPU twin of the C-ABI - the host build of the kernels' per-cell source - so they run without a GPU:

  * exact geometry comes back: a 3-D point projected exactly into two views triangulates to itself and is kept;
  * the null-vector solver is a minimiser: its residual |A x| / |x| is the smallest singular value (against NumPy's f64 SVD);
  * neighbour order does not matter except for ties: permuting the slots permutes the winners, the survivors are the same points;
  * segment order of the indexed mode: groups in order of first appearance while scanning the selection, members in selection order;
  * emission is idempotent and cell-local: a sub-selection's survivors are the corresponding subset of the full selection's.
"""
import numpy as np
import torch
from hypothesis import HealthCheck, assume, given, settings
from hypothesis import strategies as st

import lichtfeld_densification_plugin_amd as lfd
from lichtfeld_densification_plugin_amd import synthetic
from lichtfeld_densification_plugin_amd.core import hip_backend as hb

CAMS = synthetic.ring_cameras(24, seed=5)
FAST = settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.too_slow])


def _project(cam, X):
    p = np.asarray(cam.P, np.float64) @ np.append(X, 1.0)
    return p[:2] / p[2], p[2]


def _norm(px, size):
    return np.float32(np.float64(px) / (0.5 * (size - 1)) - 1.0)


@FAST
@given(i=st.integers(0, 23), step=st.integers(1, 4), x=st.floats(-1.2, 1.2), y=st.floats(-1.2, 1.2), z=st.floats(-0.2, 0.5))
def test_exact_correspondences_triangulate_back_to_the_point(i, step, x, y, z):
    ca, cb = CAMS[i], CAMS[(i + step) % 24]
    X = np.array([x, y, z])
    (ua, va), za = _project(ca, X)
    (ub, vb), zb = _project(cb, X)
    assume(za > 0.5 and zb > 0.5 and 0 <= ua < ca.width and 0 <= va < ca.height and 0 <= ub < cb.width and 0 <= vb < cb.height)
    wm, hm = ca.width, ca.height                       # match grid = camera size: the pixel mapping is the identity up to f32 rounding
    cfg = lfd.DensePipelineConfig(output_path="", min_parallax_deg=0.0)
    o = hb.host_eval_correspondence(ca, cb, _norm(ua, wm), _norm(va, hm), _norm(ub, wm), _norm(vb, hm), wm, hm, hb.make_params(cfg))
    assert o[7] == 1.0                                   # kept
    # the pixels were rounded to f32 normalised coordinates (~1e-4 px): the point comes back to ~1e-4 of the scene scale
    assert np.abs(o[:3] - X).max() <= 2e-3 * max(1.0, np.abs(X).max()), (o[:3], X)
    assert o[6] <= 0.05                                  # reprojection error, px


@FAST
@given(seed=st.integers(0, 10_000), noise=st.floats(0.0, 3.0), i=st.integers(0, 23), step=st.integers(1, 5))
def test_null_vector_is_the_smallest_singular_vector(seed, noise, i, step):
    rs = np.random.RandomState(seed)
    ca, cb = CAMS[i], CAMS[(i + step) % 24]
    X = np.array([rs.uniform(-1, 1), rs.uniform(-1, 1), rs.uniform(-0.2, 0.4)])
    (ua, va), za = _project(ca, X)
    (ub, vb), zb = _project(cb, X)
    assume(za > 0.5 and zb > 0.5)
    f = np.float32
    ua, va, ub, vb = f(ua), f(va), f(ub + rs.normal(0, noise)), f(vb + rs.normal(0, noise))
    P1, P2 = ca.P.astype(f), cb.P.astype(f)
    A = np.stack([ua * P1[2] - P1[0], va * P1[2] - P1[1], ub * P2[2] - P2[0], vb * P2[2] - P2[1]]).astype(f)     # upstream's rows (core/geometry.py:72-75)
    x, solves = hb.host_null_vector(A)
    assert solves >= 3 and np.all(np.isfinite(x))
    A64 = A.astype(np.float64)
    _, sv, Vt = np.linalg.svd(A64)
    res = np.linalg.norm(A64 @ x) / np.linalg.norm(x)
    assert res <= sv[3] * (1.0 + 1e-6) + 1e-9 * sv[0]                              # a minimiser of |A x| / |x|
    if sv[3] <= 0.5 * sv[2]:                                                      # the direction, where it is well defined
        v = Vt[3] * np.sign(Vt[3] @ x)
        assert np.linalg.norm(x / np.linalg.norm(x) - v) <= 1e-6


def _reference(seed, H=24, W=32, k=3, ref=4):
    nbrs = synthetic.ring_neighbours(24, ref, k)
    s = synthetic.synth_reference(CAMS, ref, nbrs, H, W, W, H, noise_px=0.4, outlier_frac=0.05, channels=2, seed=seed, cert_mode="tiefree")
    return s, nbrs


@settings(max_examples=12, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(seed=st.integers(0, 1000), perm=st.permutations([0, 1, 2]))
def test_neighbour_order_only_matters_for_ties(seed, perm):
    H, W = 24, 32
    s, nbrs = _reference(seed, H, W)
    twin = hb.HostDensifier(1)
    twin.upload_cameras(CAMS)
    params = hb.make_params(lfd.DensePipelineConfig(output_path=""))

    def run(order):
        r = hb.ReferenceInputs(ref_cam=s.ref_index, nbr_cams=[nbrs[j] for j in order], cert=[s.cert[j] for j in order],
                               warp=[s.warp[j].contiguous() for j in order], image=s.image)
        b = hb.PreparedBatch([r], W, H, cameras=CAMS)
        best, slot = twin.aggregate(b, params)
        return best[0].numpy(), slot[0].numpy(), twin.triangulate_dense(b, params)
    best0, slot0, out0 = run([0, 1, 2])
    best1, slot1, out1 = run(list(perm))
    np.testing.assert_array_equal(best0, best1)                                   # tie-free certainties: the maximum does not depend on the order
    np.testing.assert_array_equal(np.asarray(perm)[slot1], slot0)                 # ... and the winner is the same neighbour
    assert out0.count == out1.count
    np.testing.assert_array_equal(out0.cell.numpy(), out1.cell.numpy())
    np.testing.assert_array_equal(out0.xyz.numpy(), out1.xyz.numpy())
    np.testing.assert_array_equal(np.asarray(perm)[out1.slot.numpy()], out0.slot.numpy())
    twin.close()


@settings(max_examples=12, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(seed=st.integers(0, 1000), n_sel=st.integers(1, 300), shuffle=st.booleans())
def test_indexed_mode_groups_by_first_appearance_and_is_cell_local(seed, n_sel, shuffle):
    H, W = 24, 32
    s, nbrs = _reference(seed, H, W)
    twin = hb.HostDensifier(1)
    twin.upload_cameras(CAMS)
    params = hb.make_params(lfd.DensePipelineConfig(output_path=""))
    r = hb.ReferenceInputs(ref_cam=s.ref_index, nbr_cams=nbrs, cert=[s.cert[j] for j in range(3)], warp=[s.warp[j].contiguous() for j in range(3)], image=s.image)
    b = hb.PreparedBatch([r], W, H, cameras=CAMS)
    _best, slot = twin.aggregate(b, params)
    slot = slot[0].numpy().reshape(-1)
    rs = np.random.RandomState(seed)
    sel = rs.choice(H * W, size=min(n_sel, H * W), replace=False)
    if not shuffle:
        sel = np.sort(sel)                                                        # upstream's np.unique order
    out = twin.triangulate_indexed(b, params, torch.from_numpy(sel.astype(np.int64)), [0, int(sel.size)])
    cell, oslot = out.cell.numpy(), out.slot.numpy()
    np.testing.assert_array_equal(oslot, slot[cell])
    # groups: contiguous per slot, in the order the slots first appear while scanning the selection (survivors or not: upstream builds the
    # groups before it filters, core/pipeline.py:685-695) ...
    first_seen = []
    for c in sel:
        if slot[c] not in first_seen:
            first_seen.append(int(slot[c]))
    groups = [int(g) for g in oslot[np.concatenate([[True], oslot[1:] != oslot[:-1]])]] if oslot.size else []
    assert groups == [g for g in first_seen if g in groups] and len(set(groups)) == len(groups)
    # ... members in selection order
    pos = {int(c): i for i, c in enumerate(sel)}
    for g in groups:
        p = [pos[int(c)] for c in cell[oslot == g]]
        assert p == sorted(p)
    # cell-local: the survivors of a sub-selection are exactly the full selection's survivors among those cells
    half = sel[: max(1, sel.size // 2)]
    sub = twin.triangulate_indexed(b, params, torch.from_numpy(half.astype(np.int64)), [0, int(half.size)])
    keep = np.isin(cell, half)
    assert sorted(sub.cell.numpy().tolist()) == sorted(cell[keep].tolist())
    order_full = {int(c): i for i, c in enumerate(cell)}
    idx = [order_full[int(c)] for c in sub.cell.numpy()]
    np.testing.assert_array_equal(sub.xyz.numpy(), out.xyz.numpy()[idx])
    np.testing.assert_array_equal(sub.rgb.numpy(), out.rgb.numpy()[idx])
    twin.close()
