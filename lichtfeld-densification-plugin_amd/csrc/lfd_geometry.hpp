// Per-correspondence two-view geometry shared by the HIP kernels and the host-side helpers.
//
// One source, compiled for gfx950 (device) and x86 (host helpers of the C-ABI) with
// -ffp-contract=off, so every rounding below is explicit: `a*b+c` is two roundings, fmaf()/fma()
// one.  The dtype ladder follows the upstream NumPy code (all file:line are upstream):
//   f32  pixel conversion (core/pipeline.py:655-656,681-683,697-703), DLT rows (core/geometry.py:
//        72-75), reprojection (:91-104), cheirality (:107-110), parallax (:113-119), F (:122-130)
//   f64  Sampson error (:133-141: the homogeneous ones-column is f64), colour weights
//        (core/pipeline.py:671-679: int32 - f32 promotes to f64)
// The one deliberate departure is the 4x4 SVD (core/geometry.py:79,84, LAPACK sgesdd in f32): the
// smallest right singular vector is computed here in f64 from the same f32 matrix (see
// lfd_null_vector) and rounded to f32, which is closer to the exact answer than sgesdd's own result.
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define LFD_HD __host__ __device__ __forceinline__
#else
#define LFD_HD inline
#endif

// ---- device-resident tables ------------------------------------------------------------------
struct LfdCam {       // one row of the uploaded camera table (CameraRecord, f32)
    float K[9];
    float R[9];
    float t[3];
    float P[12];
    float C[3];
    int32_t w, h;
    int32_t pad[2];   // 40 words = 160 B
};

struct LfdRefConst {  // per reference view, staged in LDS
    float P[12];
    float C[3];
    float sx, sy;     // camera px per match px: (float)(w_cam / (double)w_match)
    float pad;
};

struct LfdPairConst { // per (reference, neighbour slot), staged in LDS
    double F[9];      // fundamental matrix: f32 values, widened for the f64 Sampson expression
    float P[12];
    float C[3];
    float sx, sy;
    int32_t cam;
};

struct LfdKernelParams {
    double sampson_thresh;
    float certainty_thresh;
    float reproj_thresh;
    float dot_thresh;     // parallax: keep iff dot(ray1, ray2) <= dot_thresh
    float wm1, hm1;       // (float)(w_match-1), (float)(h_match-1)
    int32_t use_sampson;  // !no_filter && sampson_thresh > 0
    int32_t use_parallax; // !no_filter && min_parallax_deg > 0
    int32_t no_filter;
};

struct LfdCellResult {
    float x, y, z;        // world point (f32)
    float err;            // max reprojection error (px)
    float xa_px, ya_px;   // reference position in match pixels (colour lookup, debug previews)
    int32_t keep;
};

// ---- 3x3 helpers: the accumulation order NumPy/OpenBLAS sgemm uses (forward FMA chain) ------------
LFD_HD float lfd_dot3_chain(float a0, float b0, float a1, float b1, float a2, float b2) {
    return fmaf(a2, b2, fmaf(a1, b1, a0 * b0));
}

// inverse of an intrinsic matrix.  Upstream calls np.linalg.inv (LAPACK sgesv, f32); for the only
// form its entry points ever build - [[fx,0,cx],[0,fy,cy],[0,0,1]] (core/geometry.py:10-30,
// densify.py:224-227) - back-substitution yields exactly 1/fx, -cx/fx, 1/fy, -cy/fy.  Any other
// matrix goes through a f64 adjugate and is rounded to f32.
LFD_HD void lfd_inv3_intrinsics(const float* K, float* Ki) {
    const bool plain = K[1] == 0.0f && K[3] == 0.0f && K[6] == 0.0f && K[7] == 0.0f && K[8] == 1.0f;
    if (plain) {
        Ki[0] = 1.0f / K[0]; Ki[1] = 0.0f; Ki[2] = -K[2] / K[0];
        Ki[3] = 0.0f; Ki[4] = 1.0f / K[4]; Ki[5] = -K[5] / K[4];
        Ki[6] = 0.0f; Ki[7] = 0.0f; Ki[8] = 1.0f;
        if (Ki[2] == 0.0f) Ki[2] = 0.0f;   // -0/fx -> +0 like the LAPACK solve
        if (Ki[5] == 0.0f) Ki[5] = 0.0f;
        return;
    }
    double m[9];
    for (int i = 0; i < 9; ++i) m[i] = (double)K[i];
    const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
    const double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
    const double inv = 1.0 / det;
    Ki[0] = (float)(c00 * inv); Ki[1] = (float)((m[2] * m[7] - m[1] * m[8]) * inv); Ki[2] = (float)((m[1] * m[5] - m[2] * m[4]) * inv);
    Ki[3] = (float)(c01 * inv); Ki[4] = (float)((m[0] * m[8] - m[2] * m[6]) * inv); Ki[5] = (float)((m[2] * m[3] - m[0] * m[5]) * inv);
    Ki[6] = (float)(c02 * inv); Ki[7] = (float)((m[1] * m[6] - m[0] * m[7]) * inv); Ki[8] = (float)((m[0] * m[4] - m[1] * m[3]) * inv);
}

// F = K2^-T [t]x R K1^-1,  R = R2 R1^T,  t = t2 - R t1   (core/geometry.py:53-55,122-130), f32.
LFD_HD void lfd_fundamental(const float* K1, const float* R1, const float* t1, const float* K2,
                            const float* R2, const float* t2, float* F) {
    float R[9], t[3], E[9], T[9], K1i[9], K2i[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            R[i * 3 + j] = lfd_dot3_chain(R2[i * 3 + 0], R1[j * 3 + 0], R2[i * 3 + 1], R1[j * 3 + 1], R2[i * 3 + 2], R1[j * 3 + 2]);
    for (int i = 0; i < 3; ++i) {
        const float rt = (R[i * 3 + 0] * t1[0] + R[i * 3 + 1] * t1[1]) + R[i * 3 + 2] * t1[2];
        t[i] = t2[i] - rt;
    }
    const float S[9] = {0.0f, -t[2], t[1], t[2], 0.0f, -t[0], -t[1], t[0], 0.0f};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            E[i * 3 + j] = lfd_dot3_chain(S[i * 3 + 0], R[0 + j], S[i * 3 + 1], R[3 + j], S[i * 3 + 2], R[6 + j]);
    lfd_inv3_intrinsics(K1, K1i);
    lfd_inv3_intrinsics(K2, K2i);
    for (int i = 0; i < 3; ++i)      // T = K2i^T @ E
        for (int j = 0; j < 3; ++j)
            T[i * 3 + j] = lfd_dot3_chain(K2i[0 + i], E[0 + j], K2i[3 + i], E[3 + j], K2i[6 + i], E[6 + j]);
    for (int i = 0; i < 3; ++i)      // F = T @ K1i
        for (int j = 0; j < 3; ++j)
            F[i * 3 + j] = lfd_dot3_chain(T[i * 3 + 0], K1i[0 + j], T[i * 3 + 1], K1i[3 + j], T[i * 3 + 2], K1i[6 + j]);
}

LFD_HD void lfd_make_ref_const(const LfdCam& c, int w_match, int h_match, LfdRefConst& o) {
    for (int i = 0; i < 12; ++i) o.P[i] = c.P[i];
    for (int i = 0; i < 3; ++i) o.C[i] = c.C[i];
    o.sx = (float)((double)c.w / (double)w_match);
    o.sy = (float)((double)c.h / (double)h_match);
    o.pad = 0.0f;
}

LFD_HD void lfd_make_pair_const(const LfdCam& a, const LfdCam& b, int cam_index, int w_match, int h_match,
                                LfdPairConst& o) {
    float F[9];
    lfd_fundamental(a.K, a.R, a.t, b.K, b.R, b.t, F);
    for (int i = 0; i < 9; ++i) o.F[i] = (double)F[i];
    for (int i = 0; i < 12; ++i) o.P[i] = b.P[i];
    for (int i = 0; i < 3; ++i) o.C[i] = b.C[i];
    o.sx = (float)((double)b.w / (double)w_match);
    o.sy = (float)((double)b.h / (double)h_match);
    o.cam = cam_index;
}

// ---- smallest right singular vector of a 4x4 matrix, f64 ---------------------------------------
// adj(A) = det(A) A^-1, so G = adj(A) adj(A)^T = det(A)^2 (A^T A)^-1 has the right singular vectors
// of A as eigenvectors with eigenvalues prod_{j!=i} sigma_j^2: the wanted vector v4 dominates by the
// factor (sigma_3/sigma_4)^2.  Squaring G squares that factor and every product of the squared matrix
// with its own dominant column multiplies the remaining error by it again; for a PSD matrix
// trace(G)^2 - trace(G^2) ~ 2*(lambda_2/lambda_1)*trace(G)^2 measures the factor, which fixes the number
// of products up front.  No pivoting and no division: G is rescaled once by an exact power of two.  Unlike
// eig(A^T A) the conditioning is that of A, not of A^T A: the entries of A are f32, so every 2x2
// minor is the difference of two EXACT f64 products (fma below changes nothing there).
#ifndef LFD_NULLVEC_TOL
#define LFD_NULLVEC_TOL 2e-8   /* legacy knob (the product count is derived from the trace test) */
#endif
#ifndef LFD_NULLVEC_MAXIT
#define LFD_NULLVEC_MAXIT 6
#endif

LFD_HD double lfd_pow2_inv_scale(double t) {
    // 2^-e with t = m * 2^e, m in [0.5, 1): exact rescaling to keep the squarings inside f64 range.
    // t <= 0, Inf or NaN give 1.0 (the caller's tests then fail and the cell is rejected downstream).
    if (!(t > 0.0) || !(t < 1.7976931348623157e308)) return 1.0;
    int e;
    (void)frexp(t, &e);
    return ldexp(1.0, -e);
}

// c[4]: un-normalised dominant column (a multiple of the null vector); returns squarings used.
LFD_HD int lfd_null_vector(const float* Af, double* c) {
    double a[16];
    for (int i = 0; i < 16; ++i) a[i] = (double)Af[i];
#define A_(i, j) a[(i) * 4 + (j)]
#define MINOR(p, q, r, s) fma(p, q, -((r) * (s)))      /* p*q - r*s, single rounding (products exact) */
    const double s0 = MINOR(A_(0, 0), A_(1, 1), A_(1, 0), A_(0, 1));
    const double s1 = MINOR(A_(0, 0), A_(1, 2), A_(1, 0), A_(0, 2));
    const double s2 = MINOR(A_(0, 0), A_(1, 3), A_(1, 0), A_(0, 3));
    const double s3 = MINOR(A_(0, 1), A_(1, 2), A_(1, 1), A_(0, 2));
    const double s4 = MINOR(A_(0, 1), A_(1, 3), A_(1, 1), A_(0, 3));
    const double s5 = MINOR(A_(0, 2), A_(1, 3), A_(1, 2), A_(0, 3));
    const double c5 = MINOR(A_(2, 2), A_(3, 3), A_(3, 2), A_(2, 3));
    const double c4 = MINOR(A_(2, 1), A_(3, 3), A_(3, 1), A_(2, 3));
    const double c3 = MINOR(A_(2, 1), A_(3, 2), A_(3, 1), A_(2, 2));
    const double c2 = MINOR(A_(2, 0), A_(3, 3), A_(3, 0), A_(2, 3));
    const double c1 = MINOR(A_(2, 0), A_(3, 2), A_(3, 0), A_(2, 2));
    const double c0 = MINOR(A_(2, 0), A_(3, 1), A_(3, 0), A_(2, 1));
#undef MINOR
    // adjugate, each entry x*u - y*v + z*w as fma(z, w, fma(-y, v, x*u))
#define ADJ(x, u, y, v, z, w) fma(z, w, fma(-(y), v, (x) * (u)))
    double J[16];
    J[0] = ADJ(A_(1, 1), c5, A_(1, 2), c4, A_(1, 3), c3);
    J[1] = -ADJ(A_(0, 1), c5, A_(0, 2), c4, A_(0, 3), c3);
    J[2] = ADJ(A_(3, 1), s5, A_(3, 2), s4, A_(3, 3), s3);
    J[3] = -ADJ(A_(2, 1), s5, A_(2, 2), s4, A_(2, 3), s3);
    J[4] = -ADJ(A_(1, 0), c5, A_(1, 2), c2, A_(1, 3), c1);
    J[5] = ADJ(A_(0, 0), c5, A_(0, 2), c2, A_(0, 3), c1);
    J[6] = -ADJ(A_(3, 0), s5, A_(3, 2), s2, A_(3, 3), s1);
    J[7] = ADJ(A_(2, 0), s5, A_(2, 2), s2, A_(2, 3), s1);
    J[8] = ADJ(A_(1, 0), c4, A_(1, 1), c2, A_(1, 3), c0);
    J[9] = -ADJ(A_(0, 0), c4, A_(0, 1), c2, A_(0, 3), c0);
    J[10] = ADJ(A_(3, 0), s4, A_(3, 1), s2, A_(3, 3), s0);
    J[11] = -ADJ(A_(2, 0), s4, A_(2, 1), s2, A_(2, 3), s0);
    J[12] = -ADJ(A_(1, 0), c3, A_(1, 1), c1, A_(1, 2), c0);
    J[13] = ADJ(A_(0, 0), c3, A_(0, 1), c1, A_(0, 2), c0);
    J[14] = -ADJ(A_(3, 0), s3, A_(3, 1), s1, A_(3, 2), s0);
    J[15] = ADJ(A_(2, 0), s3, A_(2, 1), s1, A_(2, 2), s0);
#undef ADJ
#undef A_
    // G = J J^T (symmetric, upper triangle g00 g01 g02 g03 g11 g12 g13 g22 g23 g33)
    double g00, g01, g02, g03, g11, g12, g13, g22, g23, g33;
    {
        const double j0 = J[0], j1 = J[4], j2 = J[8], j3 = J[12];
        g00 = j0 * j0; g01 = j0 * j1; g02 = j0 * j2; g03 = j0 * j3;
        g11 = j1 * j1; g12 = j1 * j2; g13 = j1 * j3; g22 = j2 * j2; g23 = j2 * j3; g33 = j3 * j3;
    }
    for (int r = 1; r < 4; ++r) {
        const double j0 = J[0 + r], j1 = J[4 + r], j2 = J[8 + r], j3 = J[12 + r];
        g00 = fma(j0, j0, g00); g01 = fma(j0, j1, g01); g02 = fma(j0, j2, g02); g03 = fma(j0, j3, g03);
        g11 = fma(j1, j1, g11); g12 = fma(j1, j2, g12); g13 = fma(j1, j3, g13);
        g22 = fma(j2, j2, g22); g23 = fma(j2, j3, g23); g33 = fma(j3, j3, g33);
    }
    double tr = (g00 + g11) + (g22 + g33);
    {   // one exact power-of-two rescale to trace in [0.5, 1): the squarings below then stay far inside
        // the f64 range (trace(G^2) lies between trace(G)^2/4 and trace(G)^2)
        const double s = lfd_pow2_inv_scale(tr);
        g00 *= s; g01 *= s; g02 *= s; g03 *= s; g11 *= s; g12 *= s; g13 *= s; g22 *= s; g23 *= s; g33 *= s;
        tr *= s;
    }
    // q = lambda_2/lambda_1 of G follows from tr(G)^2 - tr(G^2) ~ 2 q tr(G)^2 with tr(G^2) = ||G||_F^2.
    // The dominant column of G is off by q and every product with G multiplies that by q again, so
    // the number of products for a ~1e-8 error is known up front (no per-step test).  Badly
    // conditioned cells (q > 0.05, i.e. sigma_4/sigma_3 > 0.22) are squared first.
    int it = 0;
    double q2t;
    for (;;) {
        const double fro = fma(2.0, fma(g23, g23, fma(g13, g13, fma(g12, g12, fma(g03, g03, fma(g02, g02, g01 * g01))))),
                               fma(g33, g33, fma(g22, g22, fma(g11, g11, g00 * g00))));
        q2t = tr * tr - fro;                 // ~ 2 q tr^2
        if (!(q2t > 0.10 * (tr * tr)) || it >= LFD_NULLVEC_MAXIT) break;
        const double h00 = fma(g03, g03, fma(g02, g02, fma(g01, g01, g00 * g00)));
        const double h01 = fma(g03, g13, fma(g02, g12, fma(g01, g11, g00 * g01)));
        const double h02 = fma(g03, g23, fma(g02, g22, fma(g01, g12, g00 * g02)));
        const double h03 = fma(g03, g33, fma(g02, g23, fma(g01, g13, g00 * g03)));
        const double h11 = fma(g13, g13, fma(g12, g12, fma(g11, g11, g01 * g01)));
        const double h12 = fma(g13, g23, fma(g12, g22, fma(g11, g12, g01 * g02)));
        const double h13 = fma(g13, g33, fma(g12, g23, fma(g11, g13, g01 * g03)));
        const double h22 = fma(g23, g23, fma(g22, g22, fma(g12, g12, g02 * g02)));
        const double h23 = fma(g23, g33, fma(g22, g23, fma(g12, g13, g02 * g03)));
        const double h33 = fma(g33, g33, fma(g23, g23, fma(g13, g13, g03 * g03)));
        g00 = h00; g01 = h01; g02 = h02; g03 = h03; g11 = h11; g12 = h12; g13 = h13; g22 = h22; g23 = h23; g33 = h33;
        tr = (g00 + g11) + (g22 + g33);
        ++it;
    }
    // dominant column of G = column of its largest diagonal entry
    double best = g00;
    c[0] = g00; c[1] = g01; c[2] = g02; c[3] = g03;
    if (g11 > best) { best = g11; c[0] = g01; c[1] = g11; c[2] = g12; c[3] = g13; }
    if (g22 > best) { best = g22; c[0] = g02; c[1] = g12; c[2] = g22; c[3] = g23; }
    if (g33 > best) { best = g33; c[0] = g03; c[1] = g13; c[2] = g23; c[3] = g33; }
    // error of c is q; products needed for q^(m+1) <= ~1e-8:  q <= 1e-4: 1, 2e-3: 2, 1e-2: 3, 2.5e-2: 4, else 5
    const double t2 = tr * tr;
    const int extra = 1 + (q2t > 2e-4 * t2) + (q2t > 4e-3 * t2) + (q2t > 2e-2 * t2) + (q2t > 5e-2 * t2);
    for (int m = 0; m < extra; ++m) {
        const double y0 = fma(g03, c[3], fma(g02, c[2], fma(g01, c[1], g00 * c[0])));
        const double y1 = fma(g13, c[3], fma(g12, c[2], fma(g11, c[1], g01 * c[0])));
        const double y2 = fma(g23, c[3], fma(g22, c[2], fma(g12, c[1], g02 * c[0])));
        const double y3 = fma(g33, c[3], fma(g23, c[2], fma(g13, c[1], g03 * c[0])));
        c[0] = y0; c[1] = y1; c[2] = y2; c[3] = y3;
    }
    return it + extra;
}

// a / b for f64 with ONE division shared by several numerators: r = RN(1/b); q = RN(a*r);
// q' = RN(q + (a - b*q)*r) is the correctly rounded quotient (Markstein), i.e. bit-identical to a/b.
LFD_HD double lfd_div_by_recip(double a, double b, double r) {
    const double q = a * r;
    return fma(fma(-b, q, a), r, q);
}

// ---- one correspondence -------------------------------------------------------------------------
LFD_HD float lfd_match_px(float n, float size_m1) {   // (n + 1.0) * 0.5 * (size - 1), f32
    return ((n + 1.0f) * 0.5f) * size_m1;
}

// a / b in f32 from r = RN(1/b): q = RN(a*r); q' = RN(q + (a - b*q)*r) is the correctly rounded quotient
// (Markstein), i.e. bit-identical to the IEEE division while a*r stays in the normal range - which
// it does for every finite reprojection / ray normalisation here (|b| >= 1e-12, so |r| <= 1e12).
LFD_HD float lfd_div_by_recip_f32(float a, float b, float r) {
    const float q = a * r;
    return fmaf(fmaf(-b, q, a), r, q);
}

LFD_HD float lfd_proj_row(const float* P, int row, float X0, float X1, float X2, float X3) {
    // (X @ P.T)[row] as sgemm accumulates it: forward FMA chain over the 4 terms
    const float* p = P + row * 4;
    return fmaf(X3, p[3], fmaf(X2, p[2], fmaf(X1, p[1], X0 * p[0])));
}

LFD_HD float lfd_reproj(const float* P, float X0, float X1, float X2, float X3, float u, float v, float& pz) {
    const float px = lfd_proj_row(P, 0, X0, X1, X2, X3);
    const float py = lfd_proj_row(P, 1, X0, X1, X2, X3);
    pz = lfd_proj_row(P, 2, X0, X1, X2, X3);
    const float z = (pz < 1e-12f) ? 1e-12f : pz;          // np.maximum(z, 1e-12): NaN stays NaN
    const float rz = 1.0f / z;                             // one division, two correctly rounded quotients
    const float du = lfd_div_by_recip_f32(px, z, rz) - u;
    const float dv = lfd_div_by_recip_f32(py, z, rz) - v;
    return sqrtf(du * du + dv * dv);
}

LFD_HD bool lfd_finite(float x) { return fabsf(x) <= 3.402823466e+38f; }   // false for NaN/Inf

LFD_HD void lfd_eval_correspondence(const LfdRefConst& rc, const LfdPairConst& pc, float xan, float yan,
                                    float xbn, float ybn, const LfdKernelParams& kp, LfdCellResult& o) {
    const float xa = lfd_match_px(xan, kp.wm1), ya = lfd_match_px(yan, kp.hm1);
    const float xb = lfd_match_px(xbn, kp.wm1), yb = lfd_match_px(ybn, kp.hm1);
    const float ua = xa * rc.sx, va = ya * rc.sy;
    const float ub = xb * pc.sx, vb = yb * pc.sy;
    o.xa_px = xa; o.ya_px = ya;
    o.x = o.y = o.z = 0.0f; o.err = 0.0f; o.keep = 0;

    if (kp.use_sampson) {     // f64 throughout (core/geometry.py:133-141)
        const double x1 = (double)ua, y1 = (double)va, x2 = (double)ub, y2 = (double)vb;
        const double* F = pc.F;
        const double fx0 = fma(F[1], y1, F[0] * x1) + F[2];
        const double fx1 = fma(F[4], y1, F[3] * x1) + F[5];
        const double fx2 = fma(F[7], y1, F[6] * x1) + F[8];
        const double ft0 = fma(F[3], y2, F[0] * x2) + F[6];
        const double ft1 = fma(F[4], y2, F[1] * x2) + F[7];
        const double num = (x2 * fx0 + y2 * fx1) + fx2;
        const double den = (((fx0 * fx0 + fx1 * fx1) + ft0 * ft0) + ft1 * ft1) + 1e-12;
        // se = num^2/den < thresh  <=>  num^2 < thresh*den  (den > 0); differs from the division only
        // within one f64 ulp of the threshold, and NaN still rejects
        if (!((num * num) < kp.sampson_thresh * den)) return;
    }

    float A[16];              // DLT rows, f32, multiply then subtract (core/geometry.py:72-75)
    for (int c = 0; c < 4; ++c) {
        A[0 + c] = ua * rc.P[8 + c] - rc.P[0 + c];
        A[4 + c] = va * rc.P[8 + c] - rc.P[4 + c];
        A[8 + c] = ub * pc.P[8 + c] - pc.P[0 + c];
        A[12 + c] = vb * pc.P[8 + c] - pc.P[4 + c];
    }
    double c[4];
#if defined(LFD_ABLATE_SOLVER)
    c[0] = A[0]; c[1] = A[5]; c[2] = A[10]; c[3] = A[15];
#else
    lfd_null_vector(A, c);
#endif
    // upstream: Xh = unit null vector, w = (|Xh[3]| < 1e-12 ? 1e-12 : Xh[3]), X = Xh / w
    // (core/geometry.py:84-87).  |c3|/|c| < 1e-12 is tested on squares; the common branch divides by
    // c3 directly (X3 == 1), the guard branch normalises first (sign is lost on purpose).
    const double n2 = (c[0] * c[0] + c[1] * c[1]) + (c[2] * c[2] + c[3] * c[3]);
    float X0, X1, X2, X3;
    if (c[3] * c[3] < 1e-24 * n2) {
        const double inv = 1.0 / (sqrt(n2) * 1e-12);
        X0 = (float)(c[0] * inv); X1 = (float)(c[1] * inv); X2 = (float)(c[2] * inv); X3 = (float)(c[3] * inv);
    } else {
        const double r = 1.0 / c[3];
        X0 = (float)lfd_div_by_recip(c[0], c[3], r);
        X1 = (float)lfd_div_by_recip(c[1], c[3], r);
        X2 = (float)lfd_div_by_recip(c[2], c[3], r);
        X3 = 1.0f;
    }

    float z1, z2;
    const float e1 = lfd_reproj(rc.P, X0, X1, X2, X3, ua, va, z1);
    const float e2 = lfd_reproj(pc.P, X0, X1, X2, X3, ub, vb, z2);
    const float err = (e1 > e2 || e1 != e1) ? e1 : e2;   // np.maximum: NaN wins
    o.x = X0; o.y = X1; o.z = X2; o.err = err;

    if (kp.no_filter) {       // core/pipeline.py:739-743
        o.keep = (lfd_finite(X0) && lfd_finite(X1) && lfd_finite(X2) && lfd_finite(X3) && lfd_finite(err)) ? 1 : 0;
        return;
    }
    bool keep = (err <= kp.reproj_thresh) && (z1 > 0.0f) && (z2 > 0.0f);
    if (keep && kp.use_parallax) {   // core/geometry.py:113-119, all f32
        float a0 = X0 - rc.C[0], a1 = X1 - rc.C[1], a2 = X2 - rc.C[2];
        float b0 = X0 - pc.C[0], b1 = X1 - pc.C[1], b2 = X2 - pc.C[2];
        const float na = sqrtf((a0 * a0 + a1 * a1) + a2 * a2) + 1e-12f;
        const float nb = sqrtf((b0 * b0 + b1 * b1) + b2 * b2) + 1e-12f;
        const float ra = 1.0f / na, rb = 1.0f / nb;
        a0 = lfd_div_by_recip_f32(a0, na, ra); a1 = lfd_div_by_recip_f32(a1, na, ra); a2 = lfd_div_by_recip_f32(a2, na, ra);
        b0 = lfd_div_by_recip_f32(b0, nb, rb); b1 = lfd_div_by_recip_f32(b1, nb, rb); b2 = lfd_div_by_recip_f32(b2, nb, rb);
        const float dot = (a0 * b0 + a1 * b1) + a2 * b2;
        keep = dot <= kp.dot_thresh;      // == degrees(arccos(clip(dot,-1,1))) >= min_deg
    }
    o.keep = keep ? 1 : 0;
}

// ---- colour (core/pipeline.py:661-679) -------------------------------------------------------
LFD_HD int lfd_floor_to_i32(float x) {   // np.floor(x).astype(np.int32): out-of-range/NaN -> INT_MIN (x86 cvttss2si)
    const float f = floorf(x);
    if (!(f >= -2147483648.0f && f < 2147483648.0f)) return (int)0x80000000;
    return (int)f;
}

LFD_HD int lfd_clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

LFD_HD void lfd_bilinear_rgb(const uint8_t* img, int wi, int hi, float xa_px, float ya_px, float sx_img,
                             float sy_img, float* rgb) {
    const float xi = xa_px * sx_img, yi = ya_px * sy_img;
    // clip(floor(.).astype(int32), 0, dim-1); NaN -> 0 like the x86 conversion upstream runs on
    const int x0 = (int)fminf(fmaxf(floorf(xi), 0.0f), (float)(wi - 1));
    const int y0 = (int)fminf(fmaxf(floorf(yi), 0.0f), (float)(hi - 1));
    const int x1 = lfd_clampi(x0 + 1, 0, wi - 1);
    const int y1 = lfd_clampi(y0 + 1, 0, hi - 1);
    const double xd = (double)xi, yd = (double)yi;
    const double ax = (double)x1 - xd, bx = xd - (double)x0;
    const double ay = (double)y1 - yd, by = yd - (double)y0;
    const double wa = ax * ay, wb = bx * ay, wc = ax * by, wd = bx * by;
    // 32-bit offsets from the (uniform) image base: the compiler keeps the base in SGPRs
    const uint8_t* pa = img + ((unsigned)y0 * (unsigned)wi + (unsigned)x0) * 3u;
    const uint8_t* pb = img + ((unsigned)y0 * (unsigned)wi + (unsigned)x1) * 3u;
    const uint8_t* pcx = img + ((unsigned)y1 * (unsigned)wi + (unsigned)x0) * 3u;
    const uint8_t* pd = img + ((unsigned)y1 * (unsigned)wi + (unsigned)x1) * 3u;
    for (int c = 0; c < 3; ++c) {
        const double s = (((double)(float)pa[c] * wa + (double)(float)pb[c] * wb) + (double)(float)pcx[c] * wc) +
                         (double)(float)pd[c] * wd;
        rgb[c] = (float)lfd_div_by_recip(s, 255.0, 1.0 / 255.0);   // == s / 255.0, correctly rounded
    }
}

// ---- certainty prologue pieces (core/pipeline.py:361-382,405-430) --------------------------------
LFD_HD float lfd_cert_floor(float c, float thresh) { return (c < thresh) ? thresh : c; }   // NaN stays NaN

LFD_HD int lfd_nearest_src(int dst, float scale, int in_size) {   // F.interpolate(mode="nearest")
    const int s = (int)floorf((float)dst * scale);
    return s < in_size - 1 ? s : in_size - 1;
}

// F.grid_sample(..., mode="nearest", padding_mode="zeros", align_corners=False) index, -1 = outside
LFD_HD int lfd_grid_nearest(float g, int size) {
    const float f = ((g + 1.0f) * (float)size - 1.0f) / 2.0f;
    const float r = nearbyintf(f);                 // round half to even
    if (!(r >= 0.0f && r <= (float)(size - 1))) return -1;
    return (int)r;
}
