// balm_math.h -- per-factor / per-voxel fp64 math of the BALM plane-eigenvalue factor, written for
// one GPU lane (everything lives in registers; no dynamic indexing).  Also compiles as plain C++ so
// tests/host_emul.cpp can run the identical arithmetic on the CPU against the oracle.
//
// Math follows VOX_HESS::acc_evaluate2 (reference include/BALM/bavoxel.hpp:85-169) but is NOT its
// formulation: the voxel's Hessian contribution is factored as
//       H_voxel = blockdiag(E_i)  -  Y Y^T ,      Y = [Y_1; ...; Y_k],  Y_i in R^{6x3}
// with, per observing pose i (A_i = Auk[i] of bavoxel.hpp:133-135, c_m = 2/(lambda_0-lambda_m) < 0):
//       Y_i = [ sqrt(-c_1) A_i^T u_1 | sqrt(-c_2) A_i^T u_2 | (sqrt2/NN) [w_i ; n_i u_0] ]
//       E_i = [ 2/NN (hat(m_i) - hat(a_i) P_i) hat(a_i) - 1/2 hat(g_i[0:3])    (2/NN w_i u_0^T)   ]
//             [ (2/NN w_i u_0^T)^T                                              2 n_i/NN u_0 u_0^T ]
// so every off-diagonal pose-pair block (bavoxel.hpp:159-165) is the rank-3 product -Y_i Y_j^T and the
// diagonal block (bavoxel.hpp:141-146) is E_i - Y_i Y_i^T.  (a_i = R_i^T u_0, w_i = v_i x a_i,
// m_i = P_i a_i + (u_0 . t_i) v_i, t_i = p_i - vbar.)
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define LVBA_HD __host__ __device__ __forceinline__
#else
#define LVBA_HD inline
#endif

namespace lvba {

// cluster c[10] = Pxx Pxy Pxz Pyy Pyz Pzz vx vy vz n (body frame); pose R[9] row-major, p[3].
// T[10] = the same statistics in the world frame (PointCluster::transform, tools.hpp:450-456).
LVBA_HD void transform_cluster(const double *c, const double *R, const double *p, double *T)
{
    const double P00 = c[0], P01 = c[1], P02 = c[2], P11 = c[3], P12 = c[4], P22 = c[5];
    const double v0 = c[6], v1 = c[7], v2 = c[8], n = c[9];
    const double Rv0 = R[0] * v0 + R[1] * v1 + R[2] * v2;
    const double Rv1 = R[3] * v0 + R[4] * v1 + R[5] * v2;
    const double Rv2 = R[6] * v0 + R[7] * v1 + R[8] * v2;
    // RP = R * P
    const double a00 = R[0] * P00 + R[1] * P01 + R[2] * P02, a01 = R[0] * P01 + R[1] * P11 + R[2] * P12,
                 a02 = R[0] * P02 + R[1] * P12 + R[2] * P22;
    const double a10 = R[3] * P00 + R[4] * P01 + R[5] * P02, a11 = R[3] * P01 + R[4] * P11 + R[5] * P12,
                 a12 = R[3] * P02 + R[4] * P12 + R[5] * P22;
    const double a20 = R[6] * P00 + R[7] * P01 + R[8] * P02, a21 = R[6] * P01 + R[7] * P11 + R[8] * P12,
                 a22 = R[6] * P02 + R[7] * P12 + R[8] * P22;
    // (RP) R^T, symmetric
    const double q00 = a00 * R[0] + a01 * R[1] + a02 * R[2];
    const double q01 = a00 * R[3] + a01 * R[4] + a02 * R[5];
    const double q02 = a00 * R[6] + a01 * R[7] + a02 * R[8];
    const double q11 = a10 * R[3] + a11 * R[4] + a12 * R[5];
    const double q12 = a10 * R[6] + a11 * R[7] + a12 * R[8];
    const double q22 = a20 * R[6] + a21 * R[7] + a22 * R[8];
    const double np0 = n * p[0], np1 = n * p[1], np2 = n * p[2];
    T[0] = q00 + 2.0 * Rv0 * p[0] + np0 * p[0];
    T[1] = q01 + Rv0 * p[1] + p[0] * Rv1 + np0 * p[1];
    T[2] = q02 + Rv0 * p[2] + p[0] * Rv2 + np0 * p[2];
    T[3] = q11 + 2.0 * Rv1 * p[1] + np1 * p[1];
    T[4] = q12 + Rv1 * p[2] + p[1] * Rv2 + np1 * p[2];
    T[5] = q22 + 2.0 * Rv2 * p[2] + np2 * p[2];
    T[6] = Rv0 + np0;
    T[7] = Rv1 + np1;
    T[8] = Rv2 + np2;
    T[9] = n;
}

// Reciprocal and reciprocal square root for the eigen-solver.  On the device: the hardware estimate (v_rcp_f64 / v_rsq_f64,
// ~2^-23 relative) + two Newton steps -- 5 / 9 instructions instead of the ~28 / ~22 of an IEEE division / square root, which
// were two thirds of the voxel pass's instruction count.  The results are accurate to an ulp or two, not correctly rounded;
// Jacobi does not care (any rotation that is orthogonal to rounding error preserves the spectrum, and the rotated
// off-diagonal is set to zero explicitly).  On the host (tests/host_emul.cpp): plain IEEE arithmetic.
LVBA_HD double lvba_rcp(double d)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(d);
    r = fma(r, fma(-d, r, 1.0), r);
    r = fma(r, fma(-d, r, 1.0), r);
    return r;
#else
    return 1.0 / d;
#endif
}
LVBA_HD double lvba_rsq(double d)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rsq(d);
    y = fma(0.5 * y, fma(-(d * y), y, 1.0), y);
    y = fma(0.5 * y, fma(-(d * y), y, 1.0), y);
    return y;
#else
    return 1.0 / sqrt(d);
#endif
}

// One Jacobi rotation in the (p,q) plane of a symmetric 3x3; r is the third index.
// Names: app,aqq,apq diagonal/off-diagonal; arp,arq the other two off-diagonals; v?p,v?q columns of V.
// FAST (the LM kernels): with d = aqq - app, h = 2 apq the classical t = sgn(theta) / (|theta| + sqrt(theta^2 + 1)),
// theta = d / h, is t = s |h| / (|d| + sqrt(d^2 + h^2)), s = sgn(d) sgn(h) (s = +1 for d = 0): one reciprocal square root and
// one reciprocal instead of two divisions and a square root; c = rsqrt(t^2 + 1).  Off-diagonals below 1e-150 (d^2 + h^2 could
// underflow) are negligible against any diagonal this code sees and are dropped.  !FAST (the voxel front-end, whose plane test
// lam0 / lam2 > ratio is a discrete decision held bit-for-bit against the C++ restatement): IEEE divisions and square roots.
#define LVBA_JROT(app, aqq, apq, arp, arq, v0p, v0q, v1p, v1q, v2p, v2q)                              \
    do {                                                                                               \
        double g_ = 100.0 * fabs(apq);                                                                 \
        if ((sweep > 3 && fabs(app) + g_ == fabs(app) && fabs(aqq) + g_ == fabs(aqq)) ||               \
            (FAST ? !(fabs(apq) > 1e-150) : false)) {                                                  \
            apq = 0.0;                                                                                 \
        } else if (apq != 0.0) {                                                                       \
            double t_, c_;                                                                             \
            if (FAST) {                                                                                \
                const double d_ = aqq - app, h_ = 2.0 * apq;                                           \
                const double x_0 = fma(d_, d_, h_ * h_);                                               \
                const double den_ = fabs(d_) + x_0 * lvba_rsq(x_0);                                    \
                const double sg_ = (d_ == 0.0 || (d_ > 0.0) == (h_ > 0.0)) ? 1.0 : -1.0;               \
                t_ = sg_ * fabs(h_) * lvba_rcp(den_);                                                  \
                c_ = lvba_rsq(fma(t_, t_, 1.0));                                                       \
            } else {                                                                                   \
                const double th_ = (aqq - app) / (2.0 * apq);                                          \
                t_ = (th_ >= 0.0 ? 1.0 : -1.0) / (fabs(th_) + sqrt(th_ * th_ + 1.0));                  \
                c_ = 1.0 / sqrt(t_ * t_ + 1.0);                                                        \
            }                                                                                          \
            const double s_ = t_ * c_;                                                                 \
            app -= t_ * apq;                                                                           \
            aqq += t_ * apq;                                                                           \
            apq = 0.0;                                                                                 \
            double x_ = arp, y_ = arq;                                                                 \
            arp = c_ * x_ - s_ * y_;                                                                   \
            arq = s_ * x_ + c_ * y_;                                                                   \
            if (WANT_VEC) {                                                                            \
                x_ = v0p; y_ = v0q; v0p = c_ * x_ - s_ * y_; v0q = s_ * x_ + c_ * y_;                  \
                x_ = v1p; y_ = v1q; v1p = c_ * x_ - s_ * y_; v1q = s_ * x_ + c_ * y_;                  \
                x_ = v2p; y_ = v2q; v2p = c_ * x_ - s_ * y_; v2q = s_ * x_ + c_ * y_;                  \
            }                                                                                          \
        }                                                                                              \
    } while (0)

#define LVBA_SWAP(a, b) do { double t__ = a; a = b; b = t__; } while (0)

// Symmetric 3x3 eigen-decomposition by cyclic Jacobi (stand-in for Eigen::SelfAdjointEigenSolver,
// bavoxel.hpp:98).  Input C = [c00 c01 c02 c11 c12 c22].  lam ascending; U[3*r+m] = component r of
// eigenvector m (only if WANT_VEC).  Jacobi gives small eigenvalues of a PSD matrix to high relative
// accuracy, which is what lambda_min (~1e-4 of ~1e-1) needs.
template <bool WANT_VEC, bool FAST = false>
LVBA_HD void eig3(const double *C, double *lam, double *U)
{
    double a00 = C[0], a01 = C[1], a02 = C[2], a11 = C[3], a12 = C[4], a22 = C[5];
    double v00 = 1, v01 = 0, v02 = 0, v10 = 0, v11 = 1, v12 = 0, v20 = 0, v21 = 0, v22 = 1;
    for (int sweep = 0; sweep < 30; ++sweep) {
        if (fabs(a01) + fabs(a02) + fabs(a12) == 0.0) break;
        LVBA_JROT(a00, a11, a01, a02, a12, v00, v01, v10, v11, v20, v21); // (0,1), r=2
        LVBA_JROT(a00, a22, a02, a01, a12, v00, v02, v10, v12, v20, v22); // (0,2), r=1
        LVBA_JROT(a11, a22, a12, a01, a02, v01, v02, v11, v12, v21, v22); // (1,2), r=0
    }
    // sort ascending (3-element network), carrying the eigenvector columns
    if (a00 > a11) { LVBA_SWAP(a00, a11); if (WANT_VEC) { LVBA_SWAP(v00, v01); LVBA_SWAP(v10, v11); LVBA_SWAP(v20, v21); } }
    if (a11 > a22) { LVBA_SWAP(a11, a22); if (WANT_VEC) { LVBA_SWAP(v01, v02); LVBA_SWAP(v11, v12); LVBA_SWAP(v21, v22); } }
    if (a00 > a11) { LVBA_SWAP(a00, a11); if (WANT_VEC) { LVBA_SWAP(v00, v01); LVBA_SWAP(v10, v11); LVBA_SWAP(v20, v21); } }
    lam[0] = a00; lam[1] = a11; lam[2] = a22;
    if (WANT_VEC) {
        U[0] = v00; U[1] = v01; U[2] = v02;
        U[3] = v10; U[4] = v11; U[5] = v12;
        U[6] = v20; U[7] = v21; U[8] = v22;
    }
}

// The same decomposition for the LM kernels, where it was the larger part of the voxel pass's instruction count (cyclic Jacobi
// with eigenvectors: ~1500 instructions per voxel, a dependent chain).  A plane voxel's covariance has one small eigenvalue
// well separated from the other two (the front-end admits lam0 / lam2 <= ~0.1), which a direct method can use:
//   lam0   Newton on p(x) = det(C - x I) from x = 0: left of the smallest root p is positive, decreasing and convex, so the
//          iterates rise monotonically to lam0; p and p' = -(sum of the principal 2 x 2 minors of C - x I) are formed from
//          the shifted matrix itself (not from polynomial coefficients), the first step already lands within lam0 / lam1
//          of the root and the convergence is quadratic -- 4 to 6 steps;
//   u0     the largest of the three cross products of the rows of C - lam0 I (all of them are multiples of u0);
//   u1, u2 the rows of C - lam0 I lie in the plane orthogonal to u0: e1 = the largest row, e2 = u0 x e1, and ONE Jacobi
//          rotation diagonalises the 2 x 2 block e^T C e exactly (also when lam1 ~ lam2, where u1, u2 are not unique but the
//          kernels only use the combination s1 s1^T + s2 s2^T).
// ~330 instructions with eigenvectors, ~120 for lam0 alone.  Accuracy: lam0 to ~eps ||C||^3 / (lam1 lam2) absolute, i.e.
// ~1e-13 relative for the planar voxels this path sees (the parity bar is 1e-8; tests/test_oracle.py holds it against LAPACK
// on the host).  lam ascending; U[3*r+m] = component r of eigenvector m.  !WANT_VEC: lam[0] only.
template <bool WANT_VEC>
LVBA_HD void eig3_planar(const double *C, double *lam, double *U)
{
    const double c00 = C[0], c01 = C[1], c02 = C[2], c11 = C[3], c12 = C[4], c22 = C[5];
    const double q01 = c01 * c01, q02 = c02 * c02, q12 = c12 * c12;
    const double tr = c00 + c11 + c22;
    double x = 0.0;
    // The direct method needs lam0 SEPARATED from lam1: the reference admits a voxel on lam0 / lam2 alone (bavoxel.hpp:351), so
    // an edge- or line-like voxel with lam0 ~ lam1 << lam2 is valid input.  There Newton converges only linearly (a nearly
    // double root) and the cross products that give u0 all vanish like (lam1 - lam0)(lam2 - lam0).  Both are DETECTED and sent
    // to cyclic Jacobi (eig3 with the fast reciprocals), which has no such restriction: (i) Newton that needs more than
    // LVBA_EIG3_NEWTON_OK steps or leaves without meeting its step test -- a plane voxel takes 4 to 6 --, (ii) a largest cross
    // product below 1e-4 tr^2, i.e. (lam1 - lam0)(lam2 - lam0) small against ||C||^2 (also the all-zero case, where rsq(0)
    // would give NaN).  tests/test_oracle.py::test_device_eig3_planar_near_double_root holds lam1 / lam0 in [1, 1.01].
    bool newton_ok = false;
#define LVBA_EIG3_NEWTON_OK 9
    for (int it = 0; it < LVBA_EIG3_NEWTON_OK; ++it) {
        const double a = c00 - x, b = c11 - x, c = c22 - x;
        const double m0 = b * c - q12, m1 = a * c - q02, m2 = a * b - q01; // principal minors of C - x I
        const double p = a * m0 - c01 * (c01 * c - c12 * c02) + c02 * (c01 * c12 - b * c02);
        const double ms = m0 + m1 + m2;
        if (!(ms > 0.0)) break;
        const double dx = p * lvba_rcp(ms);
        x += dx;
        if (!(fabs(dx) > 4e-17 * tr)) { newton_ok = true; break; }
    }
    if (!WANT_VEC) { // the cost kernel wants lam0 only
        if (__builtin_expect(!newton_ok, 0)) { eig3<false, true>(C, lam, nullptr); return; }
        lam[0] = x; lam[1] = lam[2] = 0.0;
        return;
    }
    const double a = c00 - x, b = c11 - x, c = c22 - x;
    // u0: rows r0 = (a, c01, c02), r1 = (c01, b, c12), r2 = (c02, c12, c) of C - lam0 I
    const double n0[3] = {c01 * c12 - c02 * b, c02 * c01 - a * c12, a * b - q01};      // r0 x r1
    const double n1[3] = {c01 * c - c02 * c12, c02 * c02 - a * c, a * c12 - c01 * c02}; // r0 x r2
    const double n2[3] = {b * c - q12, c12 * c02 - c01 * c, c01 * c12 - b * c02};      // r1 x r2
    const double l0 = n0[0] * n0[0] + n0[1] * n0[1] + n0[2] * n0[2], l1 = n1[0] * n1[0] + n1[1] * n1[1] + n1[2] * n1[2],
                 l2 = n2[0] * n2[0] + n2[1] * n2[1] + n2[2] * n2[2];
    double u[3], ll;
    if (l0 >= l1 && l0 >= l2) { u[0] = n0[0]; u[1] = n0[1]; u[2] = n0[2]; ll = l0; }
    else if (l1 >= l2) { u[0] = n1[0]; u[1] = n1[1]; u[2] = n1[2]; ll = l1; }
    else { u[0] = n2[0]; u[1] = n2[1]; u[2] = n2[2]; ll = l2; }
    if (__builtin_expect(!newton_ok || !(ll > 1e-8 * (tr * tr) * (tr * tr)), 0)) { // (i) or (ii): ONE exit to Jacobi
        eig3<true, true>(C, lam, U);
        return;
    }
    lam[0] = x;
    const double iu = lvba_rsq(ll);
    u[0] *= iu; u[1] *= iu; u[2] *= iu;
    // e1: the largest row of C - lam0 I, made exactly orthogonal to u0 (one Gram-Schmidt step against rounding); e2 = u0 x e1
    const double w0 = a * a + q01 + q02, w1 = q01 + b * b + q12, w2 = q02 + q12 + c * c;
    double e1[3];
    if (w0 >= w1 && w0 >= w2) { e1[0] = a; e1[1] = c01; e1[2] = c02; }
    else if (w1 >= w2) { e1[0] = c01; e1[1] = b; e1[2] = c12; }
    else { e1[0] = c02; e1[1] = c12; e1[2] = c; }
    const double pe = e1[0] * u[0] + e1[1] * u[1] + e1[2] * u[2];
    e1[0] -= pe * u[0]; e1[1] -= pe * u[1]; e1[2] -= pe * u[2];
    const double ie = lvba_rsq(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
    e1[0] *= ie; e1[1] *= ie; e1[2] *= ie;
    const double e2[3] = {u[1] * e1[2] - u[2] * e1[1], u[2] * e1[0] - u[0] * e1[2], u[0] * e1[1] - u[1] * e1[0]};
    // B = [e1 e2]^T C [e1 e2]
    const double g1[3] = {c00 * e1[0] + c01 * e1[1] + c02 * e1[2], c01 * e1[0] + c11 * e1[1] + c12 * e1[2], c02 * e1[0] + c12 * e1[1] + c22 * e1[2]};
    const double g2[3] = {c00 * e2[0] + c01 * e2[1] + c02 * e2[2], c01 * e2[0] + c11 * e2[1] + c12 * e2[2], c02 * e2[0] + c12 * e2[1] + c22 * e2[2]};
    double b11 = e1[0] * g1[0] + e1[1] * g1[1] + e1[2] * g1[2];
    double b22 = e2[0] * g2[0] + e2[1] * g2[1] + e2[2] * g2[2];
    const double b12 = e1[0] * g2[0] + e1[1] * g2[1] + e1[2] * g2[2];
    double cs = 1.0, sn = 0.0;
    if (fabs(b12) > 1e-150) { // the rotation of LVBA_JROT, in its division-free form
        const double d_ = b22 - b11, h_ = 2.0 * b12;
        const double x_0 = fma(d_, d_, h_ * h_);
        const double den_ = fabs(d_) + x_0 * lvba_rsq(x_0);
        const double sg_ = (d_ == 0.0 || (d_ > 0.0) == (h_ > 0.0)) ? 1.0 : -1.0;
        const double t_ = sg_ * fabs(h_) * lvba_rcp(den_);
        cs = lvba_rsq(fma(t_, t_, 1.0));
        sn = t_ * cs;
        b11 -= t_ * b12;
        b22 += t_ * b12;
    }
    double ua[3] = {cs * e1[0] - sn * e2[0], cs * e1[1] - sn * e2[1], cs * e1[2] - sn * e2[2]};
    double ub[3] = {sn * e1[0] + cs * e2[0], sn * e1[1] + cs * e2[1], sn * e1[2] + cs * e2[2]};
    if (b11 > b22) {
        LVBA_SWAP(b11, b22);
        LVBA_SWAP(ua[0], ub[0]); LVBA_SWAP(ua[1], ub[1]); LVBA_SWAP(ua[2], ub[2]);
    }
    lam[1] = b11; lam[2] = b22;
    U[0] = u[0]; U[1] = ua[0]; U[2] = ub[0];
    U[3] = u[1]; U[4] = ua[1]; U[5] = ub[1];
    U[6] = u[2]; U[7] = ua[2]; U[8] = ub[2];
}

// Merged-voxel covariance from the summed world-frame statistics S[10] (bavoxel.hpp:97-98).
template <bool FAST = false>
LVBA_HD void voxel_cov(const double *S, double *C, double *vbar)
{
    const double inv = FAST ? lvba_rcp(S[9]) : 1.0 / S[9];
    vbar[0] = S[6] * inv; vbar[1] = S[7] * inv; vbar[2] = S[8] * inv;
    C[0] = S[0] * inv - vbar[0] * vbar[0];
    C[1] = S[1] * inv - vbar[0] * vbar[1];
    C[2] = S[2] * inv - vbar[0] * vbar[2];
    C[3] = S[3] * inv - vbar[1] * vbar[1];
    C[4] = S[4] * inv - vbar[1] * vbar[2];
    C[5] = S[5] * inv - vbar[2] * vbar[2];
}

// Per-voxel record shared by the voxel's factors (13 doubles).
struct VoxRec {
    double NN;        // total point count
    double vb[3];     // merged centroid
    double u0[3];     // eigenvector of lambda_min
    double s1[3];     // sqrt(-c1) * u1,  c1 = 2/(lambda0 - lambda1)
    double s2[3];     // sqrt(-c2) * u2
};
#define LVBA_VOXREC_DOUBLES 13

// From summed statistics to (lambda_min, VoxRec).
LVBA_HD double voxel_finish(const double *S, VoxRec &vr)
{
    double C[6], lam[3], U[9];
    voxel_cov<true>(S, C, vr.vb);
    eig3_planar<true>(C, lam, U);
    vr.NN = S[9];
    const double k1 = lvba_rsq(0.5 * (lam[1] - lam[0])), k2 = lvba_rsq(0.5 * (lam[2] - lam[0])); // sqrt(2 / (lam_m - lam_0))
    vr.u0[0] = U[0]; vr.u0[1] = U[3]; vr.u0[2] = U[6];
    vr.s1[0] = k1 * U[1]; vr.s1[1] = k1 * U[4]; vr.s1[2] = k1 * U[7];
    vr.s2[0] = k2 * U[2]; vr.s2[1] = k2 * U[5]; vr.s2[2] = k2 * U[8];
    return lam[0];
}

LVBA_HD double voxel_lambda_min(const double *S)
{
    double C[6], lam[3], vb[3];
    voxel_cov<true>(S, C, vb);
    eig3_planar<false>(C, lam, nullptr);
    return lam[0];
}

#define LVBA_CROSS(o, a, b)                                                                            \
    do {                                                                                               \
        o[0] = a[1] * b[2] - a[2] * b[1];                                                              \
        o[1] = a[2] * b[0] - a[0] * b[2];                                                              \
        o[2] = a[0] * b[1] - a[1] * b[0];                                                              \
    } while (0)
#define LVBA_DOT(a, b) (a[0] * b[0] + a[1] * b[1] + a[2] * b[2])

// Per-factor derivatives.  Outputs:
//   Y[18]  : Y[6*m + e] = component e (0..5) of column m (0..2) of Y_i
//   D[21]  : lower triangle of the 6x6 diagonal block E_i - Y_i Y_i^T, column-major packed:
//            for c in 0..5, for r in c..5  (index lvba::dlow(r,c))
//   g[6]   : gradient block A_i^T u0 (bavoxel.hpp:137-138)
LVBA_HD void factor_derivs(const double *c, const double *R, const double *p, const VoxRec &vr,
                           double *Y, double *D, double *g)
{
    const double P00 = c[0], P01 = c[1], P02 = c[2], P11 = c[3], P12 = c[4], P22 = c[5];
    const double v[3] = {c[6], c[7], c[8]};
    const double n = c[9];
    const double invN = lvba_rcp(vr.NN);
    const double *u0 = vr.u0;
    // a = R^T u0
    const double a[3] = {R[0] * u0[0] + R[3] * u0[1] + R[6] * u0[2], R[1] * u0[0] + R[4] * u0[1] + R[7] * u0[2],
                         R[2] * u0[0] + R[5] * u0[1] + R[8] * u0[2]};
    const double Pa[3] = {P00 * a[0] + P01 * a[1] + P02 * a[2], P01 * a[0] + P11 * a[1] + P12 * a[2],
                          P02 * a[0] + P12 * a[1] + P22 * a[2]};
    double w[3];
    LVBA_CROSS(w, v, a); // vihat * RiTuk
    const double t[3] = {p[0] - vr.vb[0], p[1] - vr.vb[1], p[2] - vr.vb[2]};
    const double s = LVBA_DOT(u0, t);
    const double m[3] = {Pa[0] + s * v[0], Pa[1] + s * v[1], Pa[2] + s * v[2]}; // combo1 = hat(m)
    const double Rv[3] = {R[0] * v[0] + R[1] * v[1] + R[2] * v[2], R[3] * v[0] + R[4] * v[1] + R[5] * v[2],
                          R[6] * v[0] + R[7] * v[1] + R[8] * v[2]};
    const double c2[3] = {Rv[0] + n * t[0], Rv[1] + n * t[1], Rv[2] + n * t[2]};
    const double c2u = LVBA_DOT(c2, u0);
    // A_left rows:  ((R P + t v^T) hat(a) - R hat(m))_i = M_i x a - R_i x m,   all / NN
    double AL[9];
#define LVBA_ALROW(i)                                                                                  \
    do {                                                                                               \
        const double M_[3] = {R[3 * i] * P00 + R[3 * i + 1] * P01 + R[3 * i + 2] * P02 + t[i] * v[0],  \
                              R[3 * i] * P01 + R[3 * i + 1] * P11 + R[3 * i + 2] * P12 + t[i] * v[1],  \
                              R[3 * i] * P02 + R[3 * i + 1] * P12 + R[3 * i + 2] * P22 + t[i] * v[2]}; \
        const double r_[3] = {R[3 * i], R[3 * i + 1], R[3 * i + 2]};                                   \
        double x_[3], y_[3];                                                                           \
        LVBA_CROSS(x_, M_, a);                                                                         \
        LVBA_CROSS(y_, r_, m);                                                                         \
        AL[3 * i] = (x_[0] - y_[0]) * invN;                                                            \
        AL[3 * i + 1] = (x_[1] - y_[1]) * invN;                                                        \
        AL[3 * i + 2] = (x_[2] - y_[2]) * invN;                                                        \
    } while (0)
    LVBA_ALROW(0);
    LVBA_ALROW(1);
    LVBA_ALROW(2);
#undef LVBA_ALROW
    // gradient
    g[0] = AL[0] * u0[0] + AL[3] * u0[1] + AL[6] * u0[2];
    g[1] = AL[1] * u0[0] + AL[4] * u0[1] + AL[7] * u0[2];
    g[2] = AL[2] * u0[0] + AL[5] * u0[1] + AL[8] * u0[2];
    const double gt = 2.0 * c2u * invN;
    g[3] = gt * u0[0]; g[4] = gt * u0[1]; g[5] = gt * u0[2];
    // Y columns 0,1: A^T (s_m),  A_right^T x = (u0 (c2.x) + c2u x)/NN
    {
        const double *x = vr.s1;
        const double cx = LVBA_DOT(c2, x) * invN, cu = c2u * invN;
        Y[0] = AL[0] * x[0] + AL[3] * x[1] + AL[6] * x[2];
        Y[1] = AL[1] * x[0] + AL[4] * x[1] + AL[7] * x[2];
        Y[2] = AL[2] * x[0] + AL[5] * x[1] + AL[8] * x[2];
        Y[3] = u0[0] * cx + cu * x[0];
        Y[4] = u0[1] * cx + cu * x[1];
        Y[5] = u0[2] * cx + cu * x[2];
    }
    {
        const double *x = vr.s2;
        const double cx = LVBA_DOT(c2, x) * invN, cu = c2u * invN;
        Y[6] = AL[0] * x[0] + AL[3] * x[1] + AL[6] * x[2];
        Y[7] = AL[1] * x[0] + AL[4] * x[1] + AL[7] * x[2];
        Y[8] = AL[2] * x[0] + AL[5] * x[1] + AL[8] * x[2];
        Y[9] = u0[0] * cx + cu * x[0];
        Y[10] = u0[1] * cx + cu * x[1];
        Y[11] = u0[2] * cx + cu * x[2];
    }
    {
        const double k = 1.4142135623730951 * invN;
        Y[12] = k * w[0]; Y[13] = k * w[1]; Y[14] = k * w[2];
        Y[15] = k * n * u0[0]; Y[16] = k * n * u0[1]; Y[17] = k * n * u0[2];
    }
    // E_rr = 2/NN (hat(m) - hat(a) P) hat(a) - 1/2 hat(g_rot); rows: 2/NN (X_i x a) - 1/2 hat(g)_i
    // X = hat(m) - hat(a) P ;  (hat(a) P)_row i = e_i-th row of hat(a) times P
    const double two = 2.0 * invN;
    double Err[9];
    {
        // hat(a) rows: [0,-a2,a1], [a2,0,-a0], [-a1,a0,0]
        const double X0[3] = {0.0 - (-a[2] * P01 + a[1] * P02), -m[2] - (-a[2] * P11 + a[1] * P12), m[1] - (-a[2] * P12 + a[1] * P22)};
        const double X1[3] = {m[2] - (a[2] * P00 - a[0] * P02), 0.0 - (a[2] * P01 - a[0] * P12), -m[0] - (a[2] * P02 - a[0] * P22)};
        const double X2[3] = {-m[1] - (-a[1] * P00 + a[0] * P01), m[0] - (-a[1] * P01 + a[0] * P11), 0.0 - (-a[1] * P02 + a[0] * P12)};
        double z[3];
        LVBA_CROSS(z, X0, a);
        Err[0] = two * z[0];                 Err[1] = two * z[1] + 0.5 * g[2];  Err[2] = two * z[2] - 0.5 * g[1];
        LVBA_CROSS(z, X1, a);
        Err[3] = two * z[0] - 0.5 * g[2];    Err[4] = two * z[1];               Err[5] = two * z[2] + 0.5 * g[0];
        LVBA_CROSS(z, X2, a);
        Err[6] = two * z[0] + 0.5 * g[1];    Err[7] = two * z[1] - 0.5 * g[0];  Err[8] = two * z[2];
    }
    // D lower, column-major packed: col0: r=0..5 -> D[0..5]; col1: r=1..5 -> D[6..10]; col2: r=2..5 -> 11..14;
    // col3: r=3..5 -> 15..17; col4: r=4..5 -> 18..19; col5: r=5 -> 20
#define LVBA_YY(r, c) (Y[r] * Y[c] + Y[6 + r] * Y[6 + c] + Y[12 + r] * Y[12 + c])
    D[0] = Err[0] - LVBA_YY(0, 0);
    D[1] = Err[3] - LVBA_YY(1, 0);
    D[2] = Err[6] - LVBA_YY(2, 0);
    D[3] = two * u0[0] * w[0] - LVBA_YY(3, 0);
    D[4] = two * u0[1] * w[0] - LVBA_YY(4, 0);
    D[5] = two * u0[2] * w[0] - LVBA_YY(5, 0);
    D[6] = Err[4] - LVBA_YY(1, 1);
    D[7] = Err[7] - LVBA_YY(2, 1);
    D[8] = two * u0[0] * w[1] - LVBA_YY(3, 1);
    D[9] = two * u0[1] * w[1] - LVBA_YY(4, 1);
    D[10] = two * u0[2] * w[1] - LVBA_YY(5, 1);
    D[11] = Err[8] - LVBA_YY(2, 2);
    D[12] = two * u0[0] * w[2] - LVBA_YY(3, 2);
    D[13] = two * u0[1] * w[2] - LVBA_YY(4, 2);
    D[14] = two * u0[2] * w[2] - LVBA_YY(5, 2);
    const double tn = two * n;
    D[15] = tn * u0[0] * u0[0] - LVBA_YY(3, 3);
    D[16] = tn * u0[1] * u0[0] - LVBA_YY(4, 3);
    D[17] = tn * u0[2] * u0[0] - LVBA_YY(5, 3);
    D[18] = tn * u0[1] * u0[1] - LVBA_YY(4, 4);
    D[19] = tn * u0[2] * u0[1] - LVBA_YY(5, 4);
    D[20] = tn * u0[2] * u0[2] - LVBA_YY(5, 5);
#undef LVBA_YY
}

// packed index of lower-triangle entry (r >= c) in D[21]
LVBA_HD constexpr int dlow(int r, int c) { return c * 6 - (c * (c - 1)) / 2 + (r - c); }

// Rodrigues, threshold as tools.hpp:62-77.  E row-major.
LVBA_HD void exp_so3(const double *w, double *E)
{
    const double th = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    if (th >= 1e-11) {
        const double k0 = w[0] / th, k1 = w[1] / th, k2 = w[2] / th;
        const double s = sin(th), c = 1.0 - cos(th);
        // K = hat(k); K^2 = k k^T - I
        E[0] = 1.0 + c * (k0 * k0 - 1.0);  E[1] = -s * k2 + c * k0 * k1;     E[2] = s * k1 + c * k0 * k2;
        E[3] = s * k2 + c * k0 * k1;       E[4] = 1.0 + c * (k1 * k1 - 1.0); E[5] = -s * k0 + c * k1 * k2;
        E[6] = -s * k1 + c * k0 * k2;      E[7] = s * k0 + c * k1 * k2;      E[8] = 1.0 + c * (k2 * k2 - 1.0);
    } else {
        E[0] = 1; E[1] = 0; E[2] = 0; E[3] = 0; E[4] = 1; E[5] = 0; E[6] = 0; E[7] = 0; E[8] = 1;
    }
}

// bavoxel.hpp:725-726: R <- R Exp(dtheta), p <- p + dp.  x, out: 12 doubles.
LVBA_HD void retract_pose(const double *x, const double *d, double *out)
{
    double E[9];
    exp_so3(d, E);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            out[3 * i + j] = x[3 * i] * E[j] + x[3 * i + 1] * E[3 + j] + x[3 * i + 2] * E[6 + j];
    out[9] = x[9] + d[3];
    out[10] = x[10] + d[4];
    out[11] = x[11] + d[5];
}

} // namespace lvba
