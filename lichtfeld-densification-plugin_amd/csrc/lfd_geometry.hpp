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
#if defined(__HIP_DEVICE_COMPILE__)
#define LFD_REREAD_CONSTANTS() asm volatile("" ::: "memory")   /* values in memory (LDS) are loaded again after this point */
#define LFD_OPAQUE4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "memory")   /* ... and these are new values to the optimiser */
#else
#define LFD_REREAD_CONSTANTS() do { } while (0)
#define LFD_OPAQUE4(a, b, c, d) do { } while (0)
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

// Projection matrices in the two constant blocks are stored with rows 0 and 1 interleaved: Pi[2c] = P[0][c], Pi[2c+1] = P[1][c],
// Pi[8+c] = P[2][c].  Everything the path computes with P comes in (row 0, row 1) pairs that share row 2's entry - the two DLT
// rows of a view, the x and y of a reprojection - so a pair is one 8-byte operand of a packed f32 instruction.
struct LfdRefConst {  // per reference view, staged in LDS
    float P[12];      // interleaved (see above)
    float C[3];
    float sx, sy;     // camera px per match px: (float)(w_cam / (double)w_match)
    float pad;
};

struct LfdPairConst { // per (reference, neighbour slot), staged in LDS
    double F[9];      // fundamental matrix: f32 values, widened for the f64 Sampson expression
    float P[12];      // interleaved (see above)
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

// A-grid axis of the matcher, torch.linspace(-1 + 1/n, 1 - 1/n, n) (core/matcher.py:132-133), element j:
// start + step*j below the midpoint, end - step*(n-1-j) from it on, all f32 (torch's CPU/GPU kernels).
struct LfdAxis {
    float start, end, step;
    int32_t half, n;
};

LFD_HD LfdAxis lfd_make_axis(int n) {
    LfdAxis a;
    a.start = (float)(-1.0 + 1.0 / (double)n);
    a.end = (float)(1.0 - 1.0 / (double)n);
    a.step = (n > 1) ? (a.end - a.start) / (float)(n - 1) : 0.0f;
    a.half = n / 2;
    a.n = n;
    return a;
}

LFD_HD float lfd_axis_value(const LfdAxis& a, int j) {
    if (a.n == 1) return a.start;
    if (j < a.half) { const float m = a.step * (float)j; return a.start + m; }
    const float m = a.step * (float)(a.n - 1 - j);
    return a.end - m;
}

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

LFD_HD void lfd_interleave_p(const float* P, float* Pi) {
    for (int c = 0; c < 4; ++c) { Pi[2 * c] = P[c]; Pi[2 * c + 1] = P[4 + c]; Pi[8 + c] = P[8 + c]; }
}

LFD_HD void lfd_make_ref_const(const LfdCam& c, int w_match, int h_match, LfdRefConst& o) {
    lfd_interleave_p(c.P, o.P);
    for (int i = 0; i < 3; ++i) o.C[i] = c.C[i];
    o.sx = (float)((double)c.w / (double)w_match);
    o.sy = (float)((double)c.h / (double)h_match);
    o.pad = 0.0f;
}

// `F_given`: the caller's fundamental matrix (f32, row-major; upstream's fundamental_from_world2cam result) or null - then it is derived here
LFD_HD void lfd_make_pair_const(const LfdCam& a, const LfdCam& b, int cam_index, int w_match, int h_match,
                                LfdPairConst& o, const float* F_given = nullptr) {
    float F[9];
    if (F_given) { for (int i = 0; i < 9; ++i) F[i] = F_given[i]; }
    else lfd_fundamental(a.K, a.R, a.t, b.K, b.R, b.t, F);
    for (int i = 0; i < 9; ++i) o.F[i] = (double)F[i];
    lfd_interleave_p(b.P, o.P);
    for (int i = 0; i < 3; ++i) o.C[i] = b.C[i];
    o.sx = (float)((double)b.w / (double)w_match);
    o.sy = (float)((double)b.h / (double)h_match);
    o.cam = cam_index;
}

// ---- smallest right singular vector of a 4x4 matrix, f64 ---------------------------------------
// Inverse iteration on M = A^T A through one LDL^T factorisation (no pivoting: M is positive
// semi-definite):
//   * the entries of A are f32, so every product in M is exact in f64 and M carries only the rounding of
//     three additions per entry; the wanted vector v4 (eigenvalue mu4 = sigma4^2) is then determined to
//     ~1e-16 (sigma1/sigma3)^2, far below what LAPACK's f32 SVD delivers upstream;
//   * x <- d4 * M^-1 x multiplies the v4 component by d4/mu4 and every other one by at most d4/mu3: the error
//     shrinks by q = (sigma4/sigma3)^2 per solve (15 flops).  The first solve from e4 is free:
//     d4 * M^-1 e4 = L^-T e4, the last column of L^-T;
//   * a nearly singular M is the good case (inverse iteration converges in one step); d4 may then come
//     out as rounding noise of either sign or exactly 0 - it only ever multiplies, so nothing blows up;
//   * reciprocals of the pivots are Newton-refined to full precision: an approximate reciprocal would
//     perturb L by its error times sigma1^2, far above sigma4^2.
// Two solves are always made; more follow only while the growth factor has not settled (bad
// conditioning: sigma4/sigma3 not small).
#ifndef LFD_PACK_ROWS
#define LFD_PACK_ROWS 0     /* bit 0: the reference view's DLT rows as packed f32 operations, bit 1: the neighbour's.  Measured: the packed forms need 2-6 registers more than the geometry loop has (scratch in the loop: 0.333-0.384 ms against 0.315), profiles/history.md (r2/ablation.txt) */
#endif
#ifndef LFD_PARALLAX_EXACT
#define LFD_PARALLAX_EXACT 4      /* upstream's own parallax sequence (normalised rays).  1: IEEE sqrtf, one IEEE reciprocal + three Markstein quotients per ray
                                     on the device; 2: six IEEE divisions; 3: one-correction-step root / reciprocal (lfd_sqrt_rn_f32 ...); 4 (default): 3, run only
                                     for cells within a derived rounding band of the threshold - everywhere else the cross-multiplied comparison provably is the
                                     same decision; 0: the cross-multiplied test of rounds 1-3 alone.  Kernel time on the bench workload against 0: 1 +7.4 %,
                                     2 +9.6 %, 3 +4.1 % (profiles/r4/ab_parallax_v1.txt, ab_parallax_v2.txt), 4: ab_parallax_v3.txt */
#endif
#ifndef LFD_NULLVEC_TOL
/* direction change (relative, on x_i/x_3) of the last solve that counts as settled.  The change measures the error of the
 * PREVIOUS iterate; the one returned is q = (sigma4/sigma3)^2 times closer: within 1e-8 of v4 for sigma4/sigma3 <= 0.1.
 * (1e-5 instead would save 0.35 solves per cell - a wave iterates until its slowest lane has settled - for 1 % of the dense
 * kernel's time and ten times the error: measured, not taken, profiles/history.md (r2/ablation.txt.)) */
#define LFD_NULLVEC_TOL 1e-6
#endif
#ifndef LFD_NULLVEC_MAXIT
#define LFD_NULLVEC_MAXIT 8       /* solves per pass */
#endif
#ifndef LFD_NULLVEC_PASSES
#define LFD_NULLVEC_PASSES 4      /* first pass unshifted, the others shifted by the Rayleigh quotient */
#endif

LFD_HD double lfd_pow2_inv_scale(double t) {
    // 2^-e with t = m * 2^e, m in [0.5, 1): exact rescaling.  t <= 0, Inf or NaN give 1.0.
    int e;
    (void)frexp(t, &e);
    int es = ((t > 0.0) && (t < 1.7976931348623157e308)) ? -e : 0;     // the EXPONENT is selected, not the result: no 1.0 literal to keep in a register
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(es));            // (... and the optimiser must not turn it back into a select of the two results)
#endif
    return ldexp(1.0, es);
}

#ifndef LFD_RCP_NEWTON_STEPS
/* v_rcp_f64 is good to 2^-24.4; one Newton step brings it to 2^-48.7 (2.2e-15 relative, profiles/history.md (r1/valu_rate.txt)), two to the last
 * bit.  One is enough here: the factorisation then is that of a matrix 2e-15 (relative) away from M - M itself carries 1e-16 per
 * entry - which turns v4 by at most 2e-15 (sigma1/sigma3)^2.  Measured on three full-size shapes against two steps
 * (profiles/cmp_newton.py): the same survivors, 99.998 % of the f32 coordinates bit-identical, the others 1 ulp apart; -1.1 % of the
 * dense kernel's time. */
#define LFD_RCP_NEWTON_STEPS 1
#endif
// Square root of the RARE paths (the Rayleigh shift of a second solver pass, the w -> 0 guard): on the device the 1-instruction
// v_rsq_f64 with two coupled Newton steps (~1e-15) instead of the compiler's IEEE expansion, whose range-scaling constants would
// otherwise sit in vector registers across the whole geometry loop; neither value is compared with anything.
LFD_HD double lfd_sqrt_rare(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const double r = __builtin_amdgcn_rsq(x);
    double s = x * r, h = 0.5 * r;
    double d = fma(-h, s, 0.5);
    s = fma(s, d, s); h = fma(h, d, h);
    d = fma(-s, s, x);
    s = fma(d, h, s);
    return (x > 0.0) ? s : 0.0;
#else
    return sqrt(x);
#endif
}

LFD_HD double lfd_recip_refined(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(d);       // v_rcp_f64 is an approximation (profiles/microbench/valu_rate.hip measures it); Newton steps bring it to ~1 ulp
    r = fma(fma(-d, r, 1.0), r, r);
#if LFD_RCP_NEWTON_STEPS >= 2
    r = fma(fma(-d, r, 1.0), r, r);
#endif
    return r;
#else
    return 1.0 / d;
#endif
}

// c[4]: un-normalised multiple of the singular vector; returns the number of solves made.
// `rows(Af)` fills the 4x4 matrix (row-major f32).  It is called at the start of EVERY pass, so neither the matrix
// nor M = A^T A stays live across the solves (30 registers on the device); the shifted passes, which are the only
// ones that need M a second time, are rare (sigma4/sigma3 close to 1) and simply rebuild it.
template <class RowFn>
LFD_HD int lfd_null_vector_rows(RowFn rows, double* c) {
    double m00, m01, m02, m03, m11, m12, m13, m22, m23, m33;
#define LFD_BUILD_M()                                                                                                   \
    {                                                                                                                   \
        float Af[16];                                                                                                   \
        rows(Af);                                                                                                       \
        {   /* M = A^T A, upper triangle (products of f32 values are exact in f64) */                                   \
            const double a0 = (double)Af[0], a1 = (double)Af[1], a2 = (double)Af[2], a3 = (double)Af[3];                \
            m00 = a0 * a0; m01 = a0 * a1; m02 = a0 * a2; m03 = a0 * a3;                                                 \
            m11 = a1 * a1; m12 = a1 * a2; m13 = a1 * a3; m22 = a2 * a2; m23 = a2 * a3; m33 = a3 * a3;                   \
        }                                                                                                               \
        _Pragma("unroll") for (int r = 1; r < 4; ++r) {                                                                 \
            const double a0 = (double)Af[4 * r + 0], a1 = (double)Af[4 * r + 1], a2 = (double)Af[4 * r + 2],            \
                         a3 = (double)Af[4 * r + 3];                                                                    \
            m00 = fma(a0, a0, m00); m01 = fma(a0, a1, m01); m02 = fma(a0, a2, m02); m03 = fma(a0, a3, m03);             \
            m11 = fma(a1, a1, m11); m12 = fma(a1, a2, m12); m13 = fma(a1, a3, m13);                                     \
            m22 = fma(a2, a2, m22); m23 = fma(a2, a3, m23); m33 = fma(a3, a3, m33);                                     \
        }                                                                                                               \
    }
    // Convergence monitor: the direction change between successive iterates is the error of the older one (the
    // iteration is linear with ratio q), so once it drops below LFD_NULLVEC_TOL the iterate just computed is within
    // q * TOL of v4.  Ordinary cells (q ~ 1e-4) settle after two solves.
    // A pass that has not settled after LFD_NULLVEC_MAXIT solves (sigma4/sigma3 close to 1) is followed by a
    // pass shifted by the Rayleigh quotient of its result, which separates the two smallest eigenvalues.
    double sh = 0.0;
    double x0 = 0.0, x1 = 0.0, x2 = 0.0, x3 = 1.0;
    int it = 0;
    for (int pass = 0;; ++pass) {
        LFD_BUILD_M();
        // M - sh I = L D L^T
        const double q00 = m00 - sh, q11 = m11 - sh, q22 = m22 - sh, q33 = m33 - sh;
        const double r0 = lfd_recip_refined(q00);
        const double l10 = m01 * r0, l20 = m02 * r0, l30 = m03 * r0;
        const double d1 = fma(-l10, m01, q11);
        const double n12 = fma(-l10, m02, m12), n13 = fma(-l10, m03, m13);
        const double n22 = fma(-l20, m02, q22), n23 = fma(-l20, m03, m23), n33 = fma(-l30, m03, q33);
        const double r1 = lfd_recip_refined(d1);
        const double l21 = n12 * r1, l31 = n13 * r1;
        const double d2 = fma(-l21, n12, n22);
        const double p23 = fma(-l21, n13, n23), p33 = fma(-l31, n13, n33);
        const double r2 = lfd_recip_refined(d2);
        const double l32 = p23 * r2;
        const double d3 = fma(-l32, p23, p33);
        const double s0 = d3 * r0, s1 = d3 * r1, s2 = d3 * r2;     // d3 / d_i
        if (pass == 0) {        // first solve from e4: the last column of L^-T
            x3 = 1.0;
            x2 = -l32;
            x1 = fma(-l21, x2, -l31);
            x0 = fma(-l10, x1, fma(-l20, x2, -l30));
            it = 1;
        }
        bool settled = false;
        for (int k = 1;; ++k) {
            const double o0 = x0, o1 = x1, o2 = x2, o3 = x3;
            // x <- d3 * (M - sh I)^-1 x : forward (L), diagonal, backward (L^T)
            const double y1 = fma(-l10, x0, x1);
            const double y2 = fma(-l21, y1, fma(-l20, x0, x2));
            const double y3 = fma(-l32, y2, fma(-l31, y1, fma(-l30, x0, x3)));
            const double z0 = x0 * s0, z1 = y1 * s1, z2 = y2 * s2;
            x3 = y3;
            x2 = fma(-l32, x3, z2);
            x1 = fma(-l21, x2, fma(-l31, x3, z1));
            x0 = fma(-l10, x1, fma(-l20, x2, fma(-l30, x3, z0)));
            ++it;
            if (k >= 2) {
                // direction change of the last solve, measured on the inhomogeneous coordinates x_i / x_3 (cross-multiplied):
                // it is ~ the error of the PREVIOUS iterate, the one returned is q times closer.  A vanishing x_3 (point at
                // infinity) never passes and runs into the iteration limits.
                const double ref = fabs(x3 * o3) * LFD_NULLVEC_TOL;
                const double e0 = fabs(fma(x0, o3, -(x3 * o0))), e1 = fabs(fma(x1, o3, -(x3 * o1))), e2 = fabs(fma(x2, o3, -(x3 * o2)));
                const bool more = (e0 > ref) || (e1 > ref) || (e2 > ref) || !(ref > 0.0);
                const bool bad = !(e0 == e0) || !(e1 == e1) || !(e2 == e2) || !(ref == ref);     // NaN: leave at once
                if (!more || bad) { settled = true; break; }
                if (k >= LFD_NULLVEC_MAXIT) break;
            }
        }
        if (settled || pass >= LFD_NULLVEC_PASSES - 1) break;
        // Rayleigh quotient of x as the next shift; x rescaled by an exact power of two
        LFD_BUILD_M();
        {
            const double sc = lfd_pow2_inv_scale(fabs(x0) + fabs(x1) + fabs(x2) + fabs(x3));
            x0 *= sc; x1 *= sc; x2 *= sc; x3 *= sc;
            const double t0 = fma(m03, x3, fma(m02, x2, fma(m01, x1, m00 * x0)));
            const double t1 = fma(m13, x3, fma(m12, x2, fma(m11, x1, m01 * x0)));
            const double t2 = fma(m23, x3, fma(m22, x2, fma(m12, x1, m02 * x0)));
            const double t3 = fma(m33, x3, fma(m23, x2, fma(m13, x1, m03 * x0)));
            const double num = fma(x3, t3, fma(x2, t2, fma(x1, t1, x0 * t0)));
            const double den = fma(x3, x3, fma(x2, x2, fma(x1, x1, x0 * x0)));
            const double rden = lfd_recip_refined(den);
            const double rho = num * rden;
            // rho >= mu4 lies between the two smallest eigenvalues and could sit closer to mu3; backing off by the
            // residual norm |Mx - rho x| / |x| ~ eps (mu3 - mu4) puts the shift below mu4, so the shifted matrix
            // stays positive definite and the iteration cannot lock on to v3 (rate ~eps instead of q)
            const double e0 = fma(-rho, x0, t0), e1 = fma(-rho, x1, t1), e2 = fma(-rho, x2, t2), e3 = fma(-rho, x3, t3);
            const double rr = fma(e3, e3, fma(e2, e2, fma(e1, e1, e0 * e0)));
            sh = rho - lfd_sqrt_rare(rr * rden);
        }
    }
    c[0] = x0; c[1] = x1; c[2] = x2; c[3] = x3;
    return it;
#undef LFD_BUILD_M
}

struct LfdCopyRows {
    const float* A;
    LFD_HD void operator()(float* Af) const { for (int i = 0; i < 16; ++i) Af[i] = A[i]; }
};
LFD_HD int lfd_null_vector(const float* Af, double* c) {
    LfdCopyRows rows{Af};
    return lfd_null_vector_rows(rows, c);
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

LFD_HD float lfd_proj_row(const float* Pi, int row, float X0, float X1, float X2, float X3) {
    // (X @ P.T)[row] as sgemm accumulates it: forward FMA chain over the 4 terms (Pi: the interleaved layout of the constant blocks)
    if (row == 2) return fmaf(X3, Pi[11], fmaf(X2, Pi[10], fmaf(X1, Pi[9], X0 * Pi[8])));
    return fmaf(X3, Pi[6 + row], fmaf(X2, Pi[4 + row], fmaf(X1, Pi[2 + row], X0 * Pi[row])));
}

// 1-ulp reciprocal / square root of the vector ALU on the device (v_rcp_f32, v_sqrt_f32: one instruction each
// instead of the ~10-instruction IEEE sequences); the host build keeps the IEEE operations.  The results only
// feed quantities that are compared with a tolerance anyway (X comes from lfd_null_vector, not from sgesdd).
LFD_HD float lfd_rcp_f32(float x) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(LFD_IEEE_FINISH)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}
LFD_HD float lfd_sqrt_f32(float x) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(LFD_IEEE_FINISH)
    return __builtin_amdgcn_sqrtf(x);
#else
    return sqrtf(x);
#endif
}

// Correctly rounded f32 square root / reciprocal from the hardware's 1-ulp instructions and ONE correction step (device only; for finite,
// normal, positive x - the callers' arguments).  s = v_sqrt(x) is within an ulp; the residual x - s^2 is exact in an fma, and
// s + (x - s^2) / (2 s) is the true root to ~2^-46, so its rounding is the correctly rounded root unless the root lies that close to a
// rounding boundary.  The reciprocal likewise: one Newton step on v_rcp.
LFD_HD float lfd_sqrt_rn_f32(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float s = __builtin_amdgcn_sqrtf(x);
    const float h = 0.5f * __builtin_amdgcn_rcpf(s);
    const float r = fmaf(-s, s, x);
    const float c = fmaf(r, h, s);
    return (x > 0.0f && x < 3.0e38f) ? c : s;                   // zero, infinity, NaN, negative: the instruction's own (IEEE) answer
#else
    return sqrtf(x);
#endif
}
LFD_HD float lfd_rcp_rn_f32(float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float y = __builtin_amdgcn_rcpf(b);
    return fmaf(fmaf(-b, y, 1.0f), y, y);
#else
    return 1.0f / b;
#endif
}

// squared reprojection distance (core/geometry.py:91-104 without the final hypot: the caller takes ONE square
// root of the larger of the two views' squares, which equals max(sqrt, sqrt) because sqrt is monotone)
LFD_HD float lfd_reproj_sq(const float* P, float X0, float X1, float X2, float X3, float u, float v, float& pz) {
    const float px = lfd_proj_row(P, 0, X0, X1, X2, X3);
    const float py = lfd_proj_row(P, 1, X0, X1, X2, X3);
    pz = lfd_proj_row(P, 2, X0, X1, X2, X3);
    // np.maximum(z, 1e-12).  (fmaxf returns the bound for a NaN depth where NumPy returns NaN: a NaN depth comes with NaN numerators - X
    // itself is NaN - so the quotients and the error are NaN either way; the maximum takes its bound as an instruction literal, the
    // compare-and-select form kept it in a vector register across the whole geometry loop.)
    const float z = fmaxf(pz, 1e-12f);
    const float rz = lfd_rcp_f32(z);                       // one reciprocal, two Markstein-corrected quotients
    const float du = lfd_div_by_recip_f32(px, z, rz) - u;
    const float dv = lfd_div_by_recip_f32(py, z, rz) - v;
    return du * du + dv * dv;
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
        // fused multiply-adds: upstream's NumPy expression rounds every product and sum separately, which moves the
        // f64 result by ~1e-16 relative; the comparison below already differs from `se < thresh` by that much
        const double fx0 = fma(F[1], y1, fma(F[0], x1, F[2]));
        const double fx1 = fma(F[4], y1, fma(F[3], x1, F[5]));
        const double fx2 = fma(F[7], y1, fma(F[6], x1, F[8]));
        const double ft0 = fma(F[3], y2, fma(F[0], x2, F[6]));
        const double ft1 = fma(F[4], y2, fma(F[1], x2, F[7]));
        const double num = fma(x2, fx0, fma(y2, fx1, fx2));
        const double den = fma(ft1, ft1, fma(ft0, ft0, fma(fx1, fx1, fx0 * fx0))) + 1e-12;      // (the constant last: an operand of the add, not a register to preload)
        // se = num^2/den < thresh  <=>  num^2 < thresh*den  (den > 0); differs from the division only
        // within one f64 ulp of the threshold, and NaN still rejects
        if (!((num * num) < kp.sampson_thresh * den)) return;
    }

    // DLT rows, f32, multiply then subtract (core/geometry.py:72-75), rebuilt from the pixel coordinates and the two
    // projection matrices whenever the solver asks for them (the opaque copy keeps the compiler from hoisting the
    // rows out of the solver's pass loop and pinning them in registers)
    struct Rows {
        const LfdRefConst& rc; const LfdPairConst& pc; float ua, va, ub, vb;
        LFD_HD void operator()(float* A) const {
            float u1 = ua, v1 = va, u2 = ub, v2 = vb;
            LFD_OPAQUE4(u1, v1, u2, v2);
            // the two rows of a view together (one packed multiply and one packed subtract per column on the device; element
            // by element the same f32 multiply-then-subtract as upstream)
            typedef float v2f __attribute__((ext_vector_type(2)));
            const v2f uv1 = {u1, v1}, uv2 = {u2, v2};
            for (int c = 0; c < 4; ++c) {
#if LFD_PACK_ROWS & 1
                const v2f p1 = {rc.P[2 * c], rc.P[2 * c + 1]};
                const v2f m1 = uv1 * rc.P[8 + c];
                const v2f r1 = m1 - p1;
                A[0 + c] = r1.x; A[4 + c] = r1.y;
#else
                A[0 + c] = u1 * rc.P[8 + c] - rc.P[2 * c];
                A[4 + c] = v1 * rc.P[8 + c] - rc.P[2 * c + 1];
#endif
#if LFD_PACK_ROWS & 2
                const v2f p2 = {pc.P[2 * c], pc.P[2 * c + 1]};
                const v2f m2 = uv2 * pc.P[8 + c];
                const v2f r2 = m2 - p2;
                A[8 + c] = r2.x; A[12 + c] = r2.y;
#else
                A[8 + c] = u2 * pc.P[8 + c] - pc.P[2 * c];
                A[12 + c] = v2 * pc.P[8 + c] - pc.P[2 * c + 1];
#endif
            }
        }
    };
    double c[4];
#if defined(LFD_ABLATE_SOLVER)
    c[0] = ua; c[1] = va; c[2] = ub; c[3] = vb;
#else
    {
        const Rows rows{rc, pc, ua, va, ub, vb};
        lfd_null_vector_rows(rows, c);
    }
#endif
    // upstream: Xh = unit null vector, w = (|Xh[3]| < 1e-12 ? 1e-12 : Xh[3]), X = Xh / w
    // (core/geometry.py:84-87).  |c3|/|c| < 1e-12 is tested on squares; the common branch divides by
    // c3 directly (X3 == 1), the guard branch normalises first (sign is lost on purpose).
    const double c33 = c[3] * c[3];
    const double n2 = fma(c[0], c[0], fma(c[1], c[1], fma(c[2], c[2], c33)));
    float X0, X1, X2, X3;
    if (c33 < 1e-24 * n2) {
        const double inv = lfd_recip_refined(lfd_sqrt_rare(n2) * 1e-12);
        X0 = (float)(c[0] * inv); X1 = (float)(c[1] * inv); X2 = (float)(c[2] * inv); X3 = (float)(c[3] * inv);
    } else {
        // r = 1/c3 to ~1 ulp of f64 (Newton-refined v_rcp_f64 on the device); the quotients are rounded to f32
        const double r = lfd_recip_refined(c[3]);
        X0 = (float)(c[0] * r);
        X1 = (float)(c[1] * r);
        X2 = (float)(c[2] * r);
        X3 = 1.0f;
    }

    // The neighbour's P and C were last used for the DLT rows; re-reading them from LDS here (instead of carrying
    // 15 registers across the solver) is what keeps the kernels at six workgroups per CU.
    LFD_REREAD_CONSTANTS();
    float z1, z2;
    const float q1 = lfd_reproj_sq(rc.P, X0, X1, X2, X3, ua, va, z1);
    const float q2 = lfd_reproj_sq(pc.P, X0, X1, X2, X3, ub, vb, z2);
    const float qm = (q1 > q2 || q1 != q1) ? q1 : q2;    // np.maximum of the two distances: NaN wins
    const float err = lfd_sqrt_f32(qm);
    o.x = X0; o.y = X1; o.z = X2; o.err = err;

    if (kp.no_filter) {       // core/pipeline.py:739-743
        o.keep = (lfd_finite(X0) && lfd_finite(X1) && lfd_finite(X2) && lfd_finite(X3) && lfd_finite(err)) ? 1 : 0;
        return;
    }
    bool keep = (err <= kp.reproj_thresh) && (z1 > 0.0f) && (z2 > 0.0f);
    if (keep && kp.use_parallax) {
        const float a0 = X0 - rc.C[0], a1 = X1 - rc.C[1], a2 = X2 - rc.C[2];
        const float b0 = X0 - pc.C[0], b1 = X1 - pc.C[1], b2 = X2 - pc.C[2];
#if LFD_PARALLAX_EXACT
        // Upstream's OWN operation sequence (core/geometry.py:113-119), operation for operation in f32: both rays normalised by
        // (norm + 1e-12) - np.linalg.norm is sqrt((x0^2 + x1^2) + x2^2), every product and sum rounded - the three products of the
        // unit rays summed in order, and the angle test as a comparison of that dot product with the largest f32 whose
        // degrees(arccos(.)) is still >= min_deg (lfd_parallax_dot_threshold: clip / arccos / degrees are monotone).  For the same X
        // the decision then is upstream's, bit for bit.  Round 3 tested  a.b <= thr (|a| + 1e-12)(|b| + 1e-12)  instead (no divisions):
        // the two forms round differently in the last few ulp of the dot product, and with the whole ring scene within 0.1 degrees of
        // the 0.5 degree threshold that decided 82 % of the cells the kernel and upstream disagreed on (profiles/parallax_attribution.py:
        // 60 of 73 flips in 524 288 cells).  sqrtf and the reciprocals are the IEEE operations on both builds.
#if LFD_PARALLAX_EXACT >= 3 && defined(__HIP_DEVICE_COMPILE__)
        // the square root, the reciprocal and the quotients from the 1-ulp hardware instructions + one correction step each (lfd_sqrt_rn_f32,
        // lfd_rcp_rn_f32, Markstein): on all 5.0e8 f32 values in [2^-20, 2^40) the reciprocal and the quotients ARE the IEEE results and the
        // root differs on 30 values (profiles/microbench/rn_check.hip, profiles/r4/rn_check.txt), at a third of the instructions of the
        // compiler's IEEE expansions (scaling for denormals and overflow, which squared ray lengths of a scene never are)
        const float ss_a = (a0 * a0 + a1 * a1) + a2 * a2, ss_b = (b0 * b0 + b1 * b1) + b2 * b2;
#if LFD_PARALLAX_EXACT == 4
        // ... and only where it can matter.  The cross-multiplied form  a.b <= thr |a||b|  (1-ulp roots, no divisions) agrees with upstream's
        // sequence whenever the two sides are farther apart than the rounding of BOTH evaluations can move them: the dot product of two
        // nearly parallel rays has three positive terms (<= 3 x 2^-24 relative), each norm <= 3.5 x 2^-24, their product and the threshold
        // one rounding each; upstream's normalised dot product <= 9 x 2^-24 - together below 1.3e-6 relative.  Outside a band of 2e-6 the
        // cheap comparison IS upstream's decision; inside it (0.9 % of the cells of the ring scene, a lane of one wave step in six) the exact
        // sequence below runs.  Tiny rays (the 1e-12 of upstream's norms would show) and NaN fall through to it as well.
        const float nn = lfd_sqrt_f32(ss_a) * lfd_sqrt_f32(ss_b);
        const float dotc = (a0 * b0 + a1 * b1) + a2 * b2;
        const float rhs = kp.dot_thresh * nn, slack = 2e-6f * nn;
        const bool sure_keep = (dotc < rhs - slack) && (nn > 1e-6f), sure_drop = (dotc > rhs + slack) && (nn > 1e-6f);
        if (sure_keep || sure_drop) {
            keep = sure_keep;
        } else
#endif
        {
        const float na = lfd_sqrt_rn_f32(ss_a) + 1e-12f;
        const float nb = lfd_sqrt_rn_f32(ss_b) + 1e-12f;
        const float ra = lfd_rcp_rn_f32(na), rb = lfd_rcp_rn_f32(nb);
        const float ua0 = lfd_div_by_recip_f32(a0, na, ra), ua1 = lfd_div_by_recip_f32(a1, na, ra), ua2 = lfd_div_by_recip_f32(a2, na, ra);
        const float ub0 = lfd_div_by_recip_f32(b0, nb, rb), ub1 = lfd_div_by_recip_f32(b1, nb, rb), ub2 = lfd_div_by_recip_f32(b2, nb, rb);
        const float dot = (ua0 * ub0 + ua1 * ub1) + ua2 * ub2;
        keep = dot <= kp.dot_thresh;
        }
#else
        const float na = sqrtf((a0 * a0 + a1 * a1) + a2 * a2) + 1e-12f;
        const float nb = sqrtf((b0 * b0 + b1 * b1) + b2 * b2) + 1e-12f;
#if LFD_PARALLAX_EXACT == 2 || !defined(__HIP_DEVICE_COMPILE__)
        const float ua0 = a0 / na, ua1 = a1 / na, ua2 = a2 / na;
        const float ub0 = b0 / nb, ub1 = b1 / nb, ub2 = b2 / nb;
#else
        // one correctly rounded reciprocal per ray, three Markstein quotients from it: q' = RN(q + (a - b q) r) with r = RN(1/b) is the
        // correctly rounded a / b (exact for every b whose significand is not all ones, i.e. all but 2^-23 of the norms)
        const float ra = 1.0f / na, rb = 1.0f / nb;
        const float ua0 = lfd_div_by_recip_f32(a0, na, ra), ua1 = lfd_div_by_recip_f32(a1, na, ra), ua2 = lfd_div_by_recip_f32(a2, na, ra);
        const float ub0 = lfd_div_by_recip_f32(b0, nb, rb), ub1 = lfd_div_by_recip_f32(b1, nb, rb), ub2 = lfd_div_by_recip_f32(b2, nb, rb);
#endif
        const float dot = (ua0 * ub0 + ua1 * ub1) + ua2 * ub2;
        keep = dot <= kp.dot_thresh;                  // == degrees(arccos(clip(dot, -1, 1))) >= min_deg; a NaN rejects in both forms
#endif
#else
        // the same test cross-multiplied (no per-component divisions):  a.b <= dot_thresh * (|a| + 1e-12) * (|b| + 1e-12).  Differs from
        // upstream's form by a few f32 ulp of the dot product.
        const float na = lfd_sqrt_f32((a0 * a0 + a1 * a1) + a2 * a2) + 1e-12f;
        const float nb = lfd_sqrt_f32((b0 * b0 + b1 * b1) + b2 * b2) + 1e-12f;
        const float dot = (a0 * b0 + a1 * b1) + a2 * b2;
        keep = dot <= kp.dot_thresh * (na * nb);
#endif
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

// The four-tap blend in f32 (dense mode's default; LFD_FLAG_EXACT_COLOUR selects the f64 form above): same taps, same
// weights (x1-x)(y1-y) ... and the same association order as upstream, every operation rounded to f32 instead of f64,
// so the result is within 4 x 255 x 2^-24 of upstream's blend before the division, i.e. within 2.5e-7 of upstream's
// rgb - the parity tests allow 1/(255*4) = 9.8e-4 and the writers quantise to 1/255.  A clamped column (x1 == x0)
// reads one texel with both weights upstream; here the weights are folded instead (the sum is the same expression).
// the blend itself, from the four weights' factors (ax = x1 - x, bx = x - x0, ay = y1 - y, by = y - y0; `clamped`: x1 == x0)
LFD_HD void lfd_blend4_weights_f32(const float* a, const float* b, const float* c, const float* d, float ax, float bx, float ay, float by,
                                   bool clamped, float* rgb) {
    float wa = ax * ay, wb = bx * ay, wc = ax * by, wd = bx * by;
    if (clamped) {
        wa = wa + wb; wb = 0.0f;
        wc = wc + wd; wd = 0.0f;
    }
    for (int ch = 0; ch < 3; ++ch) {
        const float s = fmaf(d[ch], wd, fmaf(c[ch], wc, fmaf(b[ch], wb, a[ch] * wa)));
        rgb[ch] = s * 0.00392156862745098f;       // RN(1/255): one more ulp, instead of the ~10-instruction IEEE division
    }
}

LFD_HD void lfd_blend4_f32(const float* a, const float* b, const float* c, const float* d, int wi, int hi, float xa_px,
                           float ya_px, float* rgb) {
    const float x0 = fminf(fmaxf(floorf(xa_px), 0.0f), (float)(wi - 1));
    const float y0 = fminf(fmaxf(floorf(ya_px), 0.0f), (float)(hi - 1));
    const float x1 = fminf(x0 + 1.0f, (float)(wi - 1));
    const float y1 = fminf(y0 + 1.0f, (float)(hi - 1));
    lfd_blend4_weights_f32(a, b, c, d, x1 - xa_px, xa_px - x0, y1 - ya_px, ya_px - y0, x1 == x0, rgb);
}

// Dense mode with the matcher's own A-grid: the reference position of a cell - and with it the first tap's offset and the
// weights' factors - depends on the cell's column and row only.  One entry per column and per row, computed once per grid size
// on the host with the arithmetic above (lfd_api.hip), replaces ~60 vector instructions per survivor by two 16-byte loads.
struct LfdColourCol { uint32_t off; float ax, bx; uint32_t clamped; };      // off = 3 * x0
struct LfdColourRow { uint32_t off0, off1; float ay, by; };                 // off = 3 * w_match * y0, 3 * w_match * y1
LFD_HD LfdColourCol lfd_colour_col(float xa_px, int wi) {
    const float x0 = fminf(fmaxf(floorf(xa_px), 0.0f), (float)(wi - 1));
    const float x1 = fminf(x0 + 1.0f, (float)(wi - 1));
    LfdColourCol c;
    c.off = 3u * (uint32_t)(int)x0; c.ax = x1 - xa_px; c.bx = xa_px - x0; c.clamped = (x1 == x0) ? 1u : 0u;
    // a clamped column (x1 == x0) has ax == -bx exactly, so its folded weights (ax*ay + bx*ay, ax*by + bx*by) are exactly 0:
    // zero factors give the same colour (0) without the fold, and the kernel's table path needs no special case
    if (x1 == x0) { c.ax = 0.0f; c.bx = 0.0f; }
    return c;
}
LFD_HD LfdColourRow lfd_colour_row(float ya_px, int wi, int hi) {
    const float y0 = fminf(fmaxf(floorf(ya_px), 0.0f), (float)(hi - 1));
    const int iy0 = (int)y0;
    int iy1 = iy0 + 1; if (iy1 > hi - 1) iy1 = hi - 1; if (iy1 < 0) iy1 = 0;
    const float y1 = fminf(y0 + 1.0f, (float)(hi - 1));
    LfdColourRow r;
    r.off0 = 3u * (uint32_t)wi * (uint32_t)iy0; r.off1 = 3u * (uint32_t)wi * (uint32_t)iy1; r.ay = y1 - ya_px; r.by = ya_px - y0;
    return r;
}

// byte-addressed form (CPU twin, indexed kernels): the taps upstream reads, blended by lfd_blend4_f32
LFD_HD void lfd_bilinear_rgb_f32(const uint8_t* img, int wi, int hi, float xa_px, float ya_px, float* rgb) {
    const int x0 = (int)fminf(fmaxf(floorf(xa_px), 0.0f), (float)(wi - 1));
    const int y0 = (int)fminf(fmaxf(floorf(ya_px), 0.0f), (float)(hi - 1));
    const int x1 = lfd_clampi(x0 + 1, 0, wi - 1);
    const int y1 = lfd_clampi(y0 + 1, 0, hi - 1);
    const uint8_t* pa = img + ((size_t)y0 * (size_t)wi + (size_t)x0) * 3u;
    const uint8_t* pb = img + ((size_t)y0 * (size_t)wi + (size_t)x1) * 3u;
    const uint8_t* pc = img + ((size_t)y1 * (size_t)wi + (size_t)x0) * 3u;
    const uint8_t* pd = img + ((size_t)y1 * (size_t)wi + (size_t)x1) * 3u;
    const float a[3] = {(float)pa[0], (float)pa[1], (float)pa[2]}, b[3] = {(float)pb[0], (float)pb[1], (float)pb[2]};
    const float c[3] = {(float)pc[0], (float)pc[1], (float)pc[2]}, d[3] = {(float)pd[0], (float)pd[1], (float)pd[2]};
    lfd_blend4_f32(a, b, c, d, wi, hi, xa_px, ya_px, rgb);
}

#if defined(__HIPCC__)
// Same arithmetic as lfd_bilinear_rgb, fewer memory instructions: the two taps of a row are 6
// consecutive bytes, fetched with ONE unaligned 8-byte load per row (2 loads per point instead of 12),
// split into an issue half and an evaluate half so that several points' loads can be in flight.
// The load window is clamped to the image so that it never reads past the buffer; n_bytes = h*w*3 >= 8.
__device__ __forceinline__ unsigned long long lfd_load_u64_unaligned(const uint8_t* p) {
    typedef unsigned long long __attribute__((aligned(1), may_alias)) u64_u;
    return *(const u64_u __attribute__((address_space(1)))*)p;      // the image is in device memory: global_load, not flat
}

struct LfdTapRows { unsigned long long r0, r1; };   // the 8-byte windows of the two image rows, already shifted to the first tap

// issue half: the two loads (nothing waits on them here)
__device__ __forceinline__ LfdTapRows lfd_bilinear_fetch(const uint8_t* img, int wi, int hi, float xa_px, float ya_px,
                                                         unsigned& sh0, unsigned& sh1) {
    const int x0 = (int)fminf(fmaxf(floorf(xa_px), 0.0f), (float)(wi - 1));
    const int y0 = (int)fminf(fmaxf(floorf(ya_px), 0.0f), (float)(hi - 1));
    const int y1 = lfd_clampi(y0 + 1, 0, hi - 1);
    const unsigned last = (unsigned)hi * (unsigned)wi * 3u - 8u;
    // rows and widths are below 2^24 (lfd_batch is validated): 24-bit multiply-adds, full rate (the 32-bit ones run at a quarter)
    const unsigned o0 = (__umul24((unsigned)y0, (unsigned)wi) + (unsigned)x0) * 3u;
    const unsigned o1 = (__umul24((unsigned)y1, (unsigned)wi) + (unsigned)x0) * 3u;
    const unsigned l0 = o0 < last ? o0 : last, l1 = o1 < last ? o1 : last;
    sh0 = (o0 - l0) * 8u; sh1 = (o1 - l1) * 8u;
    LfdTapRows t;
    t.r0 = lfd_load_u64_unaligned(img + l0);
    t.r1 = lfd_load_u64_unaligned(img + l1);
    return t;
}

// evaluate half: upstream's f64 weights and accumulation (core/pipeline.py:661-679), bit-identical to lfd_bilinear_rgb
__device__ __forceinline__ void lfd_bilinear_eval(LfdTapRows t, unsigned sh0, unsigned sh1, int wi, int hi, float xa_px, float ya_px, float* rgb) {
    const int x0 = (int)fminf(fmaxf(floorf(xa_px), 0.0f), (float)(wi - 1));
    const int y0 = (int)fminf(fmaxf(floorf(ya_px), 0.0f), (float)(hi - 1));
    const int x1 = lfd_clampi(x0 + 1, 0, wi - 1);
    const int y1 = lfd_clampi(y0 + 1, 0, hi - 1);
    const double xd = (double)xa_px, yd = (double)ya_px;
    const double ax = (double)x1 - xd, bx = xd - (double)x0;
    const double ay = (double)y1 - yd, by = yd - (double)y0;
    const double wa = ax * ay, wb = bx * ay, wc = ax * by, wd = bx * by;
    const unsigned long long r0 = t.r0 >> sh0, r1 = t.r1 >> sh1;
    const unsigned a3 = (unsigned)r0 & 0xffffffu, c3 = (unsigned)r1 & 0xffffffu;
    const unsigned b3 = (x1 != x0) ? (unsigned)(r0 >> 24) & 0xffffffu : a3;
    const unsigned d3 = (x1 != x0) ? (unsigned)(r1 >> 24) & 0xffffffu : c3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double pa = (double)((a3 >> (8 * c)) & 0xffu), pb = (double)((b3 >> (8 * c)) & 0xffu);
        const double pc = (double)((c3 >> (8 * c)) & 0xffu), pd = (double)((d3 >> (8 * c)) & 0xffu);
        // upstream: ((pa*wa + pb*wb) + pc*wc) + pd*wd with every product rounded.  The products are exact in f64 whenever
        // floor(log2 x) + floor(log2 y) >= 3 (a u8 times two differences of f32 pixel coordinates: <= 8 + 45 bits), i.e.
        // everywhere but a few cells next to the image origin, and then the fused chain below is bit-identical
        const double s = fma(pd, wd, fma(pc, wc, fma(pb, wb, pa * wa)));
        rgb[c] = (float)lfd_div_by_recip(s, 255.0, 1.0 / 255.0);   // == s / 255.0, correctly rounded
    }
}

// the same from a column entry and a row entry of the colour tables: issue half ...
__device__ __forceinline__ LfdTapRows lfd_bilinear_fetch_tab(const uint8_t* img, unsigned n_bytes, const LfdColourCol& cc, const LfdColourRow& cr,
                                                             unsigned& sh0, unsigned& sh1) {
    const unsigned last = n_bytes - 8u;
    const unsigned o0 = cr.off0 + cc.off, o1 = cr.off1 + cc.off;
    const unsigned l0 = o0 < last ? o0 : last, l1 = o1 < last ? o1 : last;
    sh0 = (o0 - l0) * 8u; sh1 = (o1 - l1) * 8u;
    LfdTapRows t;
    t.r0 = lfd_load_u64_unaligned(img + l0);
    t.r1 = lfd_load_u64_unaligned(img + l1);
    return t;
}
// ... and evaluate half (bit-identical to lfd_bilinear_eval_f32 at the same position)
__device__ __forceinline__ void lfd_bilinear_eval_tab(LfdTapRows t, unsigned sh0, unsigned sh1, const LfdColourCol& cc, const LfdColourRow& cr, float* rgb) {
    const unsigned long long r0 = t.r0 >> sh0, r1 = t.r1 >> sh1;
    const unsigned lo0 = (unsigned)r0, hi0 = (unsigned)(r0 >> 32), lo1 = (unsigned)r1, hi1 = (unsigned)(r1 >> 32);
    const float a[3] = {(float)(lo0 & 0xffu), (float)((lo0 >> 8) & 0xffu), (float)((lo0 >> 16) & 0xffu)};
    const float b[3] = {(float)(lo0 >> 24), (float)(hi0 & 0xffu), (float)((hi0 >> 8) & 0xffu)};
    const float c[3] = {(float)(lo1 & 0xffu), (float)((lo1 >> 8) & 0xffu), (float)((lo1 >> 16) & 0xffu)};
    const float d[3] = {(float)(lo1 >> 24), (float)(hi1 & 0xffu), (float)((hi1 >> 8) & 0xffu)};
    lfd_blend4_weights_f32(a, b, c, d, cc.ax, cc.bx, cr.ay, cr.by, false, rgb);      // (clamped columns carry zero factors)
}

// lfd_blend4_f32 on the two 8-byte windows of lfd_bilinear_fetch: the taps come straight out of the windows with
// v_cvt_f32_ubyteN (no mask / shift per tap).  Bit-identical to lfd_bilinear_rgb_f32.
__device__ __forceinline__ void lfd_bilinear_eval_f32(LfdTapRows t, unsigned sh0, unsigned sh1, int wi, int hi, float xa_px, float ya_px, float* rgb) {
    const unsigned long long r0 = t.r0 >> sh0, r1 = t.r1 >> sh1;
    const unsigned lo0 = (unsigned)r0, hi0 = (unsigned)(r0 >> 32), lo1 = (unsigned)r1, hi1 = (unsigned)(r1 >> 32);
    // bytes of a window: a.r a.g a.b b.r | b.g b.b   (with x1 == x0 the b taps are not upstream's, but their weight is 0)
    const float a[3] = {(float)(lo0 & 0xffu), (float)((lo0 >> 8) & 0xffu), (float)((lo0 >> 16) & 0xffu)};
    const float b[3] = {(float)(lo0 >> 24), (float)(hi0 & 0xffu), (float)((hi0 >> 8) & 0xffu)};
    const float c[3] = {(float)(lo1 & 0xffu), (float)((lo1 >> 8) & 0xffu), (float)((lo1 >> 16) & 0xffu)};
    const float d[3] = {(float)(lo1 >> 24), (float)(hi1 & 0xffu), (float)((hi1 >> 8) & 0xffu)};
    lfd_blend4_f32(a, b, c, d, wi, hi, xa_px, ya_px, rgb);
}
#endif

// upstream's colour quantisation (core/image_utils.py:24-26): np.clip(np.round(c * 255), 0, 255).astype(uint8) - f32 multiply, round half to even
LFD_HD unsigned char lfd_quantise_u8(float c) {
    const float v = rintf(c * 255.0f);
    if (!(v == v)) return 0;                      // NaN -> 0 (the x86 float->int conversion upstream runs on)
    return (unsigned char)fminf(fmaxf(v, 0.0f), 255.0f);
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
