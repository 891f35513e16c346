// visual_math.h -- per-lane fp64 math of the visual bundle-adjustment factors, hand-derived (the reference lets
// Ceres' Jets differentiate these functors).  Also compiles as plain C++ for tests/host_emul.cpp.
//   reprojection  ReprojErrorWhitenedDistorted::operator()   reference include/utils.hpp:61-111
//   plane prior   PointPlaneErrorWhitened::operator()        reference include/utils.hpp:133-139
//   manifold      ceres::EigenQuaternionManifold applied to the reference's [w,x,y,z] memory
//                 (src/lvba_system.cpp:1516 vs :1579) -- the tangent basis below reproduces that mismatch as is.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define LVBA_HD __host__ __device__ __forceinline__
#else
#ifndef LVBA_HD
#define LVBA_HD inline
#endif
#endif

namespace lvba {

#define LVBA_V_CROSS(o, a, b)                                                                          \
    do {                                                                                               \
        o[0] = a[1] * b[2] - a[2] * b[1];                                                              \
        o[1] = a[2] * b[0] - a[0] * b[2];                                                              \
        o[2] = a[0] * b[1] - a[1] * b[0];                                                              \
    } while (0)

// EigenQuaternionManifold::Plus on the 4-array a (read as Eigen x,y,z,w), tangent d[3]: out = q_d * a,
// q_d = [sin|d|/|d| d, cos|d|].
LVBA_HD void quat_plus(const double *a, const double *d, double *out)
{
    const double nd = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (nd == 0.0) {
        out[0] = a[0]; out[1] = a[1]; out[2] = a[2]; out[3] = a[3];
        return;
    }
    const double s = sin(nd) / nd, cw = cos(nd);
    const double qv[3] = {s * d[0], s * d[1], s * d[2]};
    double cr[3];
    LVBA_V_CROSS(cr, qv, a);
    out[0] = cw * a[0] + a[3] * qv[0] + cr[0];
    out[1] = cw * a[1] + a[3] * qv[1] + cr[1];
    out[2] = cw * a[2] + a[3] * qv[2] + cr[2];
    out[3] = cw * a[3] - (qv[0] * a[0] + qv[1] * a[1] + qv[2] * a[2]);
}

// Residual (2) and, if WANT_JAC, its Jacobians w.r.t. the camera tangent [dq(3), dt(3)] (Jc, 2x6 row-major) and
// the landmark (Jp, 2x3 row-major).  q = [w,x,y,z] (normalised inside, like ceres::QuaternionRotatePoint).
// intr = fx fy cx cy k1 k2 p1 p2; inv_sigma = 1/sigma_px.  Returns false (zero residual, zero Jacobian) when the
// point is behind / on the camera plane (utils.hpp:78).
template <bool WANT_JAC>
LVBA_HD bool reproj_eval(const double *q, const double *t, const double *X, double u, double v, const double *intr,
                         double inv_sigma, double *r, double *Jc, double *Jp)
{
    const double qn = 1.0 / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const double w = q[0] * qn, vx = q[1] * qn, vy = q[2] * qn, vz = q[3] * qn;
    const double vv[3] = {vx, vy, vz};
    double c1[3], c2[3];
    LVBA_V_CROSS(c1, vv, X);                         // v x X
    LVBA_V_CROSS(c2, vv, c1);                        // v x (v x X)
    const double Xc[3] = {X[0] + 2.0 * (w * c1[0] + c2[0]) + t[0], X[1] + 2.0 * (w * c1[1] + c2[1]) + t[1],
                          X[2] + 2.0 * (w * c1[2] + c2[2]) + t[2]};
    const double z = Xc[2];
    if (z <= 1e-8) {
        r[0] = 0.0; r[1] = 0.0;
        if (WANT_JAC) {
            for (int e = 0; e < 12; ++e) Jc[e] = 0.0;
            for (int e = 0; e < 6; ++e) Jp[e] = 0.0;
        }
        return false;
    }
    const double fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3], k1 = intr[4], k2 = intr[5], p1 = intr[6], p2 = intr[7];
    const double iz = 1.0 / z;
    const double xn = Xc[0] * iz, yn = Xc[1] * iz;
    const double r2 = xn * xn + yn * yn, r4 = r2 * r2;
    const double radial = 1.0 + k1 * r2 + k2 * r4;
    const double xd = xn * radial + 2.0 * p1 * xn * yn + p2 * (r2 + 2.0 * xn * xn);
    const double yd = yn * radial + p1 * (r2 + 2.0 * yn * yn) + 2.0 * p2 * xn * yn;
    r[0] = (fx * xd + cx - u) * inv_sigma;
    r[1] = (fy * yd + cy - v) * inv_sigma;
    if (WANT_JAC) {
        const double dr = k1 + 2.0 * k2 * r2;             // d radial / d r2
        const double dxd_dxn = radial + 2.0 * xn * xn * dr + 2.0 * p1 * yn + 6.0 * p2 * xn;
        const double dxd_dyn = 2.0 * xn * yn * dr + 2.0 * p1 * xn + 2.0 * p2 * yn;
        const double dyd_dxn = 2.0 * xn * yn * dr + 2.0 * p1 * xn + 2.0 * p2 * yn;
        const double dyd_dyn = radial + 2.0 * yn * yn * dr + 6.0 * p1 * yn + 2.0 * p2 * xn;
        // d r / d Xc  (2x3)
        const double a00 = fx * inv_sigma * dxd_dxn * iz, a01 = fx * inv_sigma * dxd_dyn * iz;
        const double a10 = fy * inv_sigma * dyd_dxn * iz, a11 = fy * inv_sigma * dyd_dyn * iz;
        const double A[6] = {a00, a01, -(a00 * xn + a01 * yn), a10, a11, -(a10 * xn + a11 * yn)};
        // d Xc / d t = I
        Jc[3] = A[0]; Jc[4] = A[1]; Jc[5] = A[2];
        Jc[9] = A[3]; Jc[10] = A[4]; Jc[11] = A[5];
        // d Xc / d X = R(u): columns R e_k = e_k + 2 (w (v x e_k) + v x (v x e_k))
        // d Xc / d(delta_m): ambient direction dq_m = (dw, dv) tangent to the unit sphere (PlusJacobian columns of
        // the memory [w,x,y,z] read as Eigen [x,y,z,w]):  m=0: (z,-y,x,-w)  m=1: (y,z,-w,-x)  m=2: (-x,w,z,-y)
        // as (dw, dx, dy, dz);   dXc = 2 dw (v x X) + 2 w (dv x X) + 2 (dv x (v x X) + v x (dv x X))
        const double dw[3] = {vz, vy, -vx};
        const double dvm[3][3] = {{-vy, vx, -w}, {vz, -w, -vx}, {w, vz, -vy}};
#define LVBA_DXC(m)                                                                                    \
    do {                                                                                               \
        double dX_[3], e1_[3], e2_[3];                                                                 \
        LVBA_V_CROSS(dX_, dvm[m], X);                                                                  \
        LVBA_V_CROSS(e1_, dvm[m], c1);                                                                 \
        LVBA_V_CROSS(e2_, vv, dX_);                                                                    \
        const double g0_ = 2.0 * (dw[m] * c1[0] + w * dX_[0] + e1_[0] + e2_[0]);                       \
        const double g1_ = 2.0 * (dw[m] * c1[1] + w * dX_[1] + e1_[1] + e2_[1]);                       \
        const double g2_ = 2.0 * (dw[m] * c1[2] + w * dX_[2] + e1_[2] + e2_[2]);                       \
        Jc[m] = A[0] * g0_ + A[1] * g1_ + A[2] * g2_;                                                  \
        Jc[6 + m] = A[3] * g0_ + A[4] * g1_ + A[5] * g2_;                                              \
    } while (0)
        LVBA_DXC(0);
        LVBA_DXC(1);
        LVBA_DXC(2);
#undef LVBA_DXC
        // R columns
#define LVBA_RCOL(k, ex, ey, ez)                                                                       \
    do {                                                                                               \
        const double e_[3] = {ex, ey, ez};                                                             \
        double f1_[3], f2_[3];                                                                         \
        LVBA_V_CROSS(f1_, vv, e_);                                                                     \
        LVBA_V_CROSS(f2_, vv, f1_);                                                                    \
        const double g0_ = e_[0] + 2.0 * (w * f1_[0] + f2_[0]);                                        \
        const double g1_ = e_[1] + 2.0 * (w * f1_[1] + f2_[1]);                                        \
        const double g2_ = e_[2] + 2.0 * (w * f1_[2] + f2_[2]);                                        \
        Jp[k] = A[0] * g0_ + A[1] * g1_ + A[2] * g2_;                                                  \
        Jp[3 + k] = A[3] * g0_ + A[4] * g1_ + A[5] * g2_;                                              \
    } while (0)
        LVBA_RCOL(0, 1.0, 0.0, 0.0);
        LVBA_RCOL(1, 0.0, 1.0, 0.0);
        LVBA_RCOL(2, 0.0, 0.0, 1.0);
#undef LVBA_RCOL
    }
    return true;
}

// Plane prior: r = sqrt(s^2 + 1e-12)/sigma, s = -(n.X + d); J (1x3) = s/sqrt(s^2+1e-12) * (-n)/sigma.
LVBA_HD double plane_eval(const double *X, const double *pl, double inv_sigma, double *J)
{
    const double s = 0.0 - (pl[0] * X[0] + pl[1] * X[1] + pl[2] * X[2] + pl[3]);
    const double rt = sqrt(s * s + 1e-12);
    if (J) {
        const double k = -(s / rt) * inv_sigma;
        J[0] = k * pl[0]; J[1] = k * pl[1]; J[2] = k * pl[2];
    }
    return rt * inv_sigma;
}

// Cholesky of the symmetric positive 3x3 C = [c00 c10 c11 c20 c21 c22] (lower, row-wise) -> L same layout.
LVBA_HD void chol3(const double *C, double *L)
{
    L[0] = sqrt(C[0]);
    L[1] = C[1] / L[0];
    L[2] = sqrt(C[2] - L[1] * L[1]);
    L[3] = C[3] / L[0];
    L[4] = (C[4] - L[3] * L[1]) / L[2];
    L[5] = sqrt(C[5] - L[3] * L[3] - L[4] * L[4]);
}
// y = L^-1 x
LVBA_HD void chol3_fwd(const double *L, const double *x, double *y)
{
    y[0] = x[0] / L[0];
    y[1] = (x[1] - L[1] * y[0]) / L[2];
    y[2] = (x[2] - L[3] * y[0] - L[4] * y[1]) / L[5];
}
// y = L^-T x
LVBA_HD void chol3_bwd(const double *L, const double *x, double *y)
{
    y[2] = x[2] / L[5];
    y[1] = (x[1] - L[4] * y[2]) / L[2];
    y[0] = (x[0] - L[1] * y[1] - L[3] * y[2]) / L[0];
}

} // namespace lvba
