#!/usr/bin/env python3
"""Where the dense kernel's solver iterations go (CPU, NumPy emulation of csrc/lfd_geometry.hpp::lfd_null_vector_rows in f64).

For references of the bench workload: per cell that reaches the solver (Sampson gate passed) the convergence ratio
q = (sigma4/sigma3)^2, the number of solves the tested loop makes, and - grouped as the kernel groups cells (a wave = 64 threads
x cell e of 4 consecutive cells) - the solves a WAVE makes (its slowest lane).  Usage: python profiles/solver_stats.py [refs]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lichtfeld_densification_plugin_amd import synthetic  # noqa: E402
from lichtfeld_densification_plugin_amd.core import hip_backend as hb  # noqa: E402


def build_rows(cams, s, j, H, W, wm, hm):
    ca, cb = cams[s.ref_index], cams[s.nbr_indices[j]]
    ax = synthetic.identity_axis_torch(W, "cpu").numpy()
    ay = synthetic.identity_axis_torch(H, "cpu").numpy()
    f = np.float32
    xa = ((ax + f(1)) * f(0.5)) * f(wm - 1)
    ya = ((ay + f(1)) * f(0.5)) * f(hm - 1)
    wp = s.warp[j].numpy()
    xb = ((wp[..., 0] + f(1)) * f(0.5)) * f(wm - 1)
    yb = ((wp[..., 1] + f(1)) * f(0.5)) * f(hm - 1)
    ua = np.broadcast_to((xa * f(ca.width / wm))[None, :], (H, W)).astype(f)
    va = np.broadcast_to((ya * f(ca.height / hm))[:, None], (H, W)).astype(f)
    ub, vb = xb * f(cb.width / wm), yb * f(cb.height / hm)
    P1, P2 = ca.P.astype(f), cb.P.astype(f)
    A = np.empty((H, W, 4, 4), f)
    A[..., 0, :] = ua[..., None] * P1[2] - P1[0]
    A[..., 1, :] = va[..., None] * P1[2] - P1[1]
    A[..., 2, :] = ub[..., None] * P2[2] - P2[0]
    A[..., 3, :] = vb[..., None] * P2[2] - P2[1]
    F = hb.fundamental_from_world2cam(ca.K, ca.R, ca.t, cb.K, cb.R, cb.t).astype(np.float64)
    x1 = np.stack([ua, va, np.ones_like(ua)], -1).astype(np.float64)
    x2 = np.stack([ub, vb, np.ones_like(ub)], -1).astype(np.float64)
    Fx1 = x1 @ F.T
    Ftx2 = x2 @ F
    num = (x2 * Fx1).sum(-1)
    den = Fx1[..., 0] ** 2 + Fx1[..., 1] ** 2 + Ftx2[..., 0] ** 2 + Ftx2[..., 1] ** 2 + 1e-12
    return A, num * num / den


def solves_needed(A, tol=1e-6, maxit=8):
    """solves of the tested loop (first pass only) per matrix, in exact arithmetic from an f64 SVD: the k-th iterate of
    x <- M^-1 x from e4 is sum_i c_i (mu4/mu_i)^k v_i with c = V^T e4; also q = (sigma4/sigma3)^2 and tan(theta_0)"""
    A = A.astype(np.float64)
    _, sv, Vt = np.linalg.svd(A)
    mu = sv ** 2
    q = mu[:, 3] / mu[:, 2]
    c = Vt[:, :, 3]                                    # components of e4 along v_i
    ratio = mu[:, 3:4] / np.maximum(mu, 1e-300)         # (mu4/mu_i)
    n = A.shape[0]
    t0 = np.sqrt((c[:, :3] ** 2).sum(1)) / np.maximum(np.abs(c[:, 3]), 1e-300)

    def iterate(k):
        return np.einsum("ni,nij->nj", c * ratio ** k, Vt)
    it = np.ones(n, int)
    done = np.zeros(n, bool)
    x = iterate(1)
    for k in range(1, maxit + 1):
        o = x
        x = iterate(k + 1)
        it += ~done
        if k >= 2:
            ref = np.abs(x[:, 3] * o[:, 3]) * tol
            e = np.abs(x[:, :3] * o[:, 3:4] - x[:, 3:4] * o[:, :3])
            more = (e > ref[:, None]).any(1) | ~(ref > 0)
            done |= ~more
        if done.all():
            break
    return it, q, t0


def main():
    n_refs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    H = W = wm = hm = 512
    cams = synthetic.ring_cameras(185, seed=0)
    for gi in range(n_refs):
        ref = (gi * 3) % 185
        nbrs = synthetic.ring_neighbours(185, ref, 3)
        s = synthetic.synth_reference(cams, ref, nbrs, H, W, wm, hm, noise_px=0.5, outlier_frac=0.05, channels=2, seed=1000 + gi,
                                      cert_mode="smooth")
        cert = np.maximum(s.cert.numpy(), np.float32(0.2))
        bj = cert.argmax(0)
        Aw = np.empty((H, W, 4, 4), np.float32)
        se = np.empty((H, W))
        for j in range(3):
            A, sj = build_rows(cams, s, j, H, W, wm, hm)
            m = bj == j
            Aw[m] = A[m]; se[m] = sj[m]
        passed = (se < 5.0).reshape(-1)
        it = np.zeros(H * W, int)
        q = np.zeros(H * W)
        idx = np.nonzero(passed)[0]
        it_p, q_p, t0_p = solves_needed(Aw.reshape(-1, 4, 4)[idx])
        it[idx] = it_p; q[idx] = q_p
        print(f"ref {gi}: {passed.mean():.3f} of the cells pass the Sampson gate")
        print("  q percentiles (50/90/99/99.9/max):", np.percentile(q_p, [50, 90, 99, 99.9, 100]))
        print("  tan(theta0) percentiles (50/90/99/max):", np.percentile(t0_p, [50, 90, 99, 100]))
        print("  solves per cell (incl. the free one): mean %.2f  hist %s" % (it_p.mean(), np.bincount(it_p)))
        # waves: tile = 1024 cells, thread t owns cells 4t..4t+3, wave w = threads 64w..64w+63, loop step e
        itw = it.reshape(-1, 4, 64, 4)              # tile, wave, lane, e
        wave_max = itw.max(axis=2)                  # tile, wave, e
        active = (itw > 0).any(axis=2)
        print("  solves per WAVE step: mean %.2f over active steps (%.3f active), hist %s" %
              (wave_max[active].mean(), active.mean(), np.bincount(wave_max[active].reshape(-1))))
        for thr in (1e-4, 1e-3, 1e-2, 3e-2, 1e-1):
            print(f"  cells with q > {thr:g}: {np.mean(q_p > thr):.5f}; waves with such a lane: {((q.reshape(-1, 4, 64, 4) > thr).any(axis=2)[active]).mean():.4f}")


if __name__ == "__main__":
    main()


def predicted_rule(A, E=1e-8, maxit=8):
    """Loop solves (after the free one) under the one-test rule: two solves, then ONE test that predicts the error of the iterate
    in hand from the last two direction changes (d2 = |x3 - x2| ~ err(x2), q_est = d2 / d1) and, from the same two numbers, how
    many more solves bring it under E."""
    A = A.astype(np.float64)
    _, sv, Vt = np.linalg.svd(A)
    mu = sv ** 2
    c = Vt[:, :, 3]
    ratio = mu[:, 3:4] / np.maximum(mu, 1e-300)

    def chart(k):
        x = np.einsum("ni,nij->nj", c * ratio ** k, Vt)
        return x[:, :3] / x[:, 3:4]
    x1, x2, x3 = chart(1), chart(2), chart(3)
    d1 = np.abs(x2 - x1).max(1) / np.maximum(np.abs(x2).max(1), 1e-300)
    d2 = np.abs(x3 - x2).max(1) / np.maximum(np.abs(x3).max(1), 1e-300)
    q_est = np.minimum(d2 / np.maximum(d1, 1e-300), 0.5)
    err3 = d2 * q_est / (1 - q_est)
    n_more = np.zeros(A.shape[0], int)
    e = err3.copy()
    for _ in range(maxit):
        need = e > E
        n_more += need
        e = np.where(need, e * q_est, e)
    true_err = np.abs(chart(20) - x3).max(1) / np.maximum(np.abs(x3).max(1), 1e-300)
    return 2 + n_more, err3, true_err


def main2():
    H = W = wm = hm = 512
    cams = synthetic.ring_cameras(185, seed=0)
    gi = 0
    ref = 0
    nbrs = synthetic.ring_neighbours(185, ref, 3)
    s = synthetic.synth_reference(cams, ref, nbrs, H, W, wm, hm, noise_px=0.5, outlier_frac=0.05, channels=2, seed=1000 + gi, cert_mode="smooth")
    cert = np.maximum(s.cert.numpy(), np.float32(0.2))
    bj = cert.argmax(0)
    Aw = np.empty((H, W, 4, 4), np.float32)
    se = np.empty((H, W))
    for j in range(3):
        A, sj = build_rows(cams, s, j, H, W, wm, hm)
        m = bj == j
        Aw[m] = A[m]; se[m] = sj[m]
    passed = (se < 5.0).reshape(-1)
    idx = np.nonzero(passed)[0]
    for E in (1e-8, 3e-8, 1e-7, 1e-6):
        n, err3, true3 = predicted_rule(Aw.reshape(-1, 4, 4)[idx], E)
        it = np.zeros(H * W, int); it[idx] = n
        wave = it.reshape(-1, 4, 64, 4).max(axis=2)
        act = wave > 0
        ok = true3 <= np.maximum(err3 * 3, 1e-13)
        print(f"E={E:g}: loop solves per cell {n.mean():.2f}, per wave step {wave[act].mean():.2f} hist {np.bincount(wave[act].reshape(-1))}; "
              f"prediction >= true error/3 on {ok.mean():.5f} of the cells")


if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[2] == "predict":
    main2()
