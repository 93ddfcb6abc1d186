#!/usr/bin/env python3
"""CPU study (NumPy, vectorised) for the round-4 solver experiment: can the null vector of the f32 4x4 DLT matrix A come from
an f32 factorisation of A itself (never forming A^T A in f64) plus ONE f64 residual-correction step, at the accuracy the f64
LDL^T + inverse iteration of csrc/lfd_geometry.hpp::lfd_null_vector_rows delivers?

  f32 part : Householder QR of A (f32, no pivoting) -> R; x0 = R^-1 e4 (back-substitution, free of r33); `n32` further solves
             x <- (R^T R)^-1 x in f32 (every division by r33 folded into a scaling of the other terms: a vanishing r33 only
             ever multiplies)
  f64 part : r = A^T (A x) - rho x with rho = |Ax|^2/|x|^2 (products of f32 values: exact in f64), d = (R^T R)^-1 r in f32,
             x <- x - d.  Error after it ~ err(x) * max(q, kappa): q = (s4/s3)^2, kappa = relative error of the f32 factorisation
             on mu3 ~ 2 |dA| / s3.

Truth: f64 SVD of the same f32 matrix.  Error measure: max_i |X_i - X_i^true| / max(|X^true|, 1) on the inhomogeneous point
X = x[:3]/x[3] (what the kernel rounds to f32: 6e-8 relative).  Usage: python profiles/mixed_solver_study.py [shape ...]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "profiles"))
from lichtfeld_densification_plugin_amd import synthetic  # noqa: E402
from solver_stats import build_rows  # noqa: E402

f32 = np.float32


def householder_qr_f32(A):
    """R (n,4,4 upper triangular, f32) of A (n,4,4 f32) by three Householder reflections, f32 throughout."""
    R = A.astype(f32).copy()
    n = R.shape[0]
    for k in range(3):
        x = R[:, k:, k]                                            # (n, 4-k)
        nrm = np.sqrt((x * x).sum(1, dtype=f32)).astype(f32)
        alpha = np.where(x[:, 0] > 0, -nrm, nrm).astype(f32)       # -sign(x0) |x|
        v = x.copy()
        v[:, 0] = (x[:, 0] - alpha).astype(f32)
        vv = (v * v).sum(1, dtype=f32)
        beta = np.where(vv > 0, f32(2.0) / np.where(vv > 0, vv, f32(1)), f32(0)).astype(f32)
        for c in range(k + 1, 4):
            col = R[:, k:, c]
            s = ((v * col).sum(1, dtype=f32) * beta).astype(f32)
            R[:, k:, c] = (col - s[:, None] * v).astype(f32)
        R[:, k, k] = alpha
        R[:, k + 1:, k] = 0
    return R


def solve_pair_f32(R, b, scaled=True):
    """x = r33^2 * (R^T R)^-1 b in f32 with every division by r33 replaced by a multiplication of the other terms."""
    r = [[R[:, i, j] for j in range(4)] for i in range(4)]
    i0, i1, i2 = (f32(1) / r[0][0]).astype(f32), (f32(1) / r[1][1]).astype(f32), (f32(1) / r[2][2]).astype(f32)
    s = r[3][3]
    # R^T y = b (forward); y' = s*y for rows 0..2, y3' = s * y3 * ... keep y3s = s*y3 = b3 - ...
    y0 = b[:, 0] * i0
    y1 = (b[:, 1] - r[0][1] * y0) * i1
    y2 = (b[:, 2] - r[0][2] * y0 - r[1][2] * y1) * i2
    y3s = (b[:, 3] - r[0][3] * y0 - r[1][3] * y1 - r[2][3] * y2)           # = s * y3
    # R z = y ; z'' = s^2 z:  z3'' = s*y3 = y3s ; z_i'' = (s^2 y_i - sum r_ij z_j'') / r_ii
    s2 = (s * s).astype(f32)
    z3 = y3s
    z2 = ((s2 * y2 - r[2][3] * z3) * i2).astype(f32)
    z1 = ((s2 * y1 - r[1][2] * z2 - r[1][3] * z3) * i1).astype(f32)
    z0 = ((s2 * y0 - r[0][1] * z1 - r[0][2] * z2 - r[0][3] * z3) * i0).astype(f32)
    return np.stack([z0, z1, z2, z3], 1).astype(f32)


def start_f32(R):
    """x = r33 * R^-1 e4: back-substitution with x3 = 1."""
    r = [[R[:, i, j] for j in range(4)] for i in range(4)]
    i0, i1, i2 = (f32(1) / r[0][0]).astype(f32), (f32(1) / r[1][1]).astype(f32), (f32(1) / r[2][2]).astype(f32)
    x3 = np.ones(R.shape[0], f32)
    x2 = (-r[2][3] * i2).astype(f32)
    x1 = ((-r[1][2] * x2 - r[1][3]) * i1).astype(f32)
    x0 = ((-r[0][1] * x1 - r[0][2] * x2 - r[0][3]) * i0).astype(f32)
    return np.stack([x0, x1, x2, x3], 1).astype(f32)


def rescale(x):
    m = np.abs(x).max(1, keepdims=True)
    e = np.floor(np.log2(np.where(m > 0, m, 1)))
    return (x * np.exp2(-e)).astype(x.dtype)


def mixed(A, n32=1, ncorr=1):
    R = householder_qr_f32(A)
    x = start_f32(R)
    for _ in range(n32):
        x = rescale(solve_pair_f32(R, rescale(x)))
    xs = [x.astype(np.float64)]
    A64 = A.astype(np.float64)
    x64 = x.astype(np.float64)
    for _ in range(ncorr):
        y = np.einsum("nij,nj->ni", A64, x64)
        g = np.einsum("nij,ni->nj", A64, y)
        rho = (y * y).sum(1) / (x64 * x64).sum(1)
        r = g - rho[:, None] * x64
        # the correction in f32: d = (R^T R)^-1 r; solve_pair returns s^2 * that, so divide by s^2 in f64 (one rcp in the kernel)
        rs = np.abs(r).max(1, keepdims=True)
        rs = np.where(rs > 0, rs, 1.0)
        d = solve_pair_f32(R, (r / rs).astype(f32)).astype(np.float64) * rs
        # x - d / s^2 is, as a direction, s^2 x - d: no division, a vanishing s (rank-deficient A: exact correspondences) only multiplies
        s2 = (R[:, 3, 3].astype(np.float64)) ** 2
        x64 = rescale(s2[:, None] * x64 - d)
        xs.append(x64)
    return xs


def chart(x):
    with np.errstate(all="ignore"):
        return x[:, :3] / x[:, 3:4]


def err_of(x, Xt):
    X = chart(x)
    return np.abs(X - Xt).max(1) / np.maximum(np.abs(Xt).max(1), 1.0)


def current_solver_err(A, Xt, tol=1e-6, maxit=8):
    """the f64 inverse iteration of lfd_null_vector_rows in exact arithmetic from the SVD: returns its error, for the yardstick"""
    A64 = A.astype(np.float64)
    _, sv, Vt = np.linalg.svd(A64)
    mu = sv ** 2
    c = Vt[:, :, 3]
    ratio = mu[:, 3:4] / np.maximum(mu, 1e-300)
    n = A.shape[0]
    done = np.zeros(n, bool)
    out = np.zeros((n, 4))

    def it(k):
        return np.einsum("ni,nij->nj", c * ratio ** k, Vt)
    x = it(1)
    for k in range(1, maxit + 1):
        o = x
        x = it(k + 1)
        if k >= 2:
            ref = np.abs(x[:, 3] * o[:, 3]) * tol
            e = np.abs(x[:, :3] * o[:, 3:4] - x[:, 3:4] * o[:, :3])
            more = (e > ref[:, None]).any(1) | ~(ref > 0)
            newly = ~more & ~done
            out[newly] = x[newly]
            done |= ~more
    out[~done] = x[~done]
    return err_of(out, Xt)


SHAPES = {
    "bench": dict(n_cams=185, k=3, noise=0.5, outl=0.05, preset=(512, 512, 512, 512)),
    "wide": dict(n_cams=40, k=3, noise=1.5, outl=0.0, preset=(512, 512, 512, 512)),
    "high_patch": dict(n_cams=194, k=3, noise=0.5, outl=0.10, preset=(640, 640, 960, 960), patch=(0.3, 0.6, 0.2, 0.8)),
    "exact": dict(n_cams=185, k=3, noise=0.0, outl=0.0, preset=(512, 512, 512, 512)),
}


def matrices(shape, ref=0, sub=4):
    sp = SHAPES[shape]
    hm, wm, H, W = sp["preset"]
    cams = synthetic.ring_cameras(sp["n_cams"], seed=0)
    nbrs = synthetic.ring_neighbours(sp["n_cams"], ref, sp["k"])
    s = synthetic.synth_reference(cams, ref, nbrs, H, W, wm, hm, noise_px=sp["noise"], outlier_frac=sp["outl"], channels=2, seed=1000,
                                  cert_mode="smooth", low_parallax_patch=sp.get("patch"))
    cert = np.maximum(s.cert.numpy(), f32(0.2))
    bj = cert.argmax(0)
    Aw = np.empty((H, W, 4, 4), f32)
    se = np.empty((H, W))
    for j in range(sp["k"]):
        A, sj = build_rows(cams, s, j, H, W, wm, hm)
        m = bj == j
        Aw[m] = A[m]; se[m] = sj[m]
    passed = (se < 5.0)
    A = Aw[::sub, ::sub][passed[::sub, ::sub]]
    return A


def main():
    shapes = sys.argv[1:] or list(SHAPES)
    pct = [50, 90, 99, 99.9, 100]
    for shape in shapes:
        A = matrices(shape)
        A64 = A.astype(np.float64)
        _, sv, Vt = np.linalg.svd(A64)
        Xt = chart(Vt[:, 3, :])
        q = (sv[:, 3] / np.maximum(sv[:, 2], 1e-300)) ** 2
        cond3 = sv[:, 0] / np.maximum(sv[:, 2], 1e-300)
        print(f"== {shape}: {A.shape[0]} matrices past the Sampson gate; q pct {np.percentile(q, pct)}; s1/s3 pct {np.percentile(cond3, pct)}")
        # upstream's own noise: LAPACK f32 SVD
        _, _, Vt32 = np.linalg.svd(A)
        print("   upstream f32 SVD error      pct", np.percentile(err_of(Vt32[:, 3, :].astype(np.float64), Xt), pct))
        print("   current f64 solver (exact arithmetic model) pct", np.percentile(current_solver_err(A, Xt), pct))
        for n32 in (1, 2):
            xs = mixed(A, n32=n32, ncorr=2)
            for i, x in enumerate(xs):
                e = err_of(x, Xt)
                print(f"   mixed: {n32} f32 solve(s) + {i} f64 correction(s): pct {np.percentile(e, pct)}   > 1e-7: {np.mean(e > 1e-7):.5f}  > 1e-6: {np.mean(e > 1e-6):.5f}")


if __name__ == "__main__":
    main()
