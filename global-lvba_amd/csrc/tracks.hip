// tracks.hip -- per-track landmark initialisation on the device: one lane per feature track.
//
// Replaces the triangulation candidate of LvbaSystem::BuildTracksAndFuse3D (reference src/lvba_system.cpp:1110-1140):
//   TriangulateTrackDLT            src/lvba_system.cpp:50-111   A^T A of the DLT rows (undistorted normalised pixels), its
//                                                               smallest eigenvector, de-homogenisation
//   ComputeMeanReproj              src/lvba_system.cpp:8-48     mean pixel error of the candidate over the track
//   undistortPixelToNormalized, projectWorldToPixel             include/utils.hpp:168-233
// The 4x4 symmetric eigenproblem is solved by cyclic Jacobi rotations in registers (Eigen::SelfAdjointEigenSolver
// upstream); the reference walks an unordered_map, here observations are taken in the caller's order -- the sums differ
// at rounding level only.  The graph part of track building (BFS over matches, view-angle filter) stays with the caller.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "lvba_common.h"
#include "mempool.h"

using namespace lvba;

namespace {

struct Intr { double fx, fy, cx, cy, k1, k2, p1, p2; };

__device__ __forceinline__ bool undistort(const Intr &c, double u, double v, double &x, double &y)
{
    if (!(isfinite(u) && isfinite(v)) || fabs(c.fx) < 1e-12 || fabs(c.fy) < 1e-12) return false;
    const double xd = (u - c.cx) / c.fx, yd = (v - c.cy) / c.fy;
    double xu = xd, yu = yd;
    for (int it = 0; it < 8; ++it) {
        const double r2 = xu * xu + yu * yu, r4 = r2 * r2;
        const double radial = 1.0 + c.k1 * r2 + c.k2 * r4;
        if (fabs(radial) < 1e-12 || !isfinite(radial)) return false;
        const double xt = 2.0 * c.p1 * xu * yu + c.p2 * (r2 + 2.0 * xu * xu);
        const double yt = c.p1 * (r2 + 2.0 * yu * yu) + 2.0 * c.p2 * xu * yu;
        xu = (xd - xt) / radial;
        yu = (yd - yt) / radial;
        if (!(isfinite(xu) && isfinite(yu))) return false;
    }
    x = xu; y = yu;
    return true;
}
__device__ __forceinline__ bool project(const Intr &c, const double *R, const double *t, const double *X, double &u, double &v)
{
    const double X0 = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + t[0];
    const double X1 = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + t[1];
    const double Z = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2];
    if (!(isfinite(X0) && isfinite(X1) && isfinite(Z)) || Z <= 1e-12) return false;
    const double x = X0 / Z, y = X1 / Z;
    const double r2 = x * x + y * y, r4 = r2 * r2;
    const double radial = 1.0 + c.k1 * r2 + c.k2 * r4;
    const double xd = x * radial + 2.0 * c.p1 * x * y + c.p2 * (r2 + 2.0 * x * x);
    const double yd = y * radial + c.p1 * (r2 + 2.0 * y * y) + 2.0 * c.p2 * x * y;
    if (!(isfinite(xd) && isfinite(yd))) return false;
    u = c.fx * xd + c.cx;
    v = c.fy * yd + c.cy;
    return isfinite(u) && isfinite(v);
}

// one Jacobi rotation in the (p, q) plane of the symmetric 4x4 A (full storage) with eigenvector accumulation in V
#define JROT4(p, q)                                                                                      \
    do {                                                                                                 \
        const double apq = A[p][q];                                                                      \
        if (apq != 0.0) {                                                                                \
            const double th = (A[q][q] - A[p][p]) / (2.0 * apq);                                         \
            const double tt = (th >= 0.0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));               \
            const double cs = 1.0 / sqrt(tt * tt + 1.0), sn = tt * cs;                                   \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                              \
                const double arp = A[r][p], arq = A[r][q];                                               \
                A[r][p] = cs * arp - sn * arq;                                                           \
                A[r][q] = sn * arp + cs * arq;                                                           \
            }                                                                                            \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                              \
                const double apr = A[p][r], aqr = A[q][r];                                               \
                A[p][r] = cs * apr - sn * aqr;                                                           \
                A[q][r] = sn * apr + cs * aqr;                                                           \
            }                                                                                            \
            A[p][q] = 0.0; A[q][p] = 0.0;                                                                \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                              \
                const double vrp = V[r][p], vrq = V[r][q];                                               \
                V[r][p] = cs * vrp - sn * vrq;                                                           \
                V[r][q] = sn * vrp + cs * vrq;                                                           \
            }                                                                                            \
        }                                                                                                \
    } while (0)

__global__ void tri_kernel(int64_t n, const int64_t *__restrict__ obs_off, const int32_t *__restrict__ obs_cam,
                           const double *__restrict__ obs_uv, const double *__restrict__ Rcw, const double *__restrict__ tcw,
                           int32_t n_cams, Intr cam, double *__restrict__ Xout, double *__restrict__ err_out,
                           int32_t *__restrict__ cnt_out, uint8_t *__restrict__ ok_out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double *Xo = Xout + 3 * i;
    Xo[0] = Xo[1] = Xo[2] = 0.0;
    err_out[i] = INFINITY;
    cnt_out[i] = 0;
    ok_out[i] = 0;
    const int64_t a = obs_off[i], b = obs_off[i + 1];
    if (b - a < 4) return; // selected_ids.size() < 4
    double A[4][4], V[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) { A[r][c] = 0.0; V[r][c] = r == c ? 1.0 : 0.0; }
    int rows = 0;
    for (int64_t o = a; o < b; ++o) {
        const int32_t cm = obs_cam[o];
        if (cm < 0 || cm >= n_cams) continue;
        double x, y;
        if (!undistort(cam, obs_uv[2 * o], obs_uv[2 * o + 1], x, y)) continue;
        const double *R = Rcw + 9 * (int64_t)cm, *t = tcw + 3 * (int64_t)cm;
        const double P0[4] = {R[0], R[1], R[2], t[0]}, P1[4] = {R[3], R[4], R[5], t[1]}, P2[4] = {R[6], R[7], R[8], t[2]};
        double ru[4], rv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) { ru[c] = x * P2[c] - P0[c]; rv[c] = y * P2[c] - P1[c]; }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) A[r][c] += ru[r] * ru[c] + rv[r] * rv[c];
        rows += 2;
    }
    if (rows < 8) return;
    for (int sweep = 0; sweep < 40; ++sweep) {
        const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[0][3]) + fabs(A[1][2]) + fabs(A[1][3]) + fabs(A[2][3]);
        if (off == 0.0) break;
        JROT4(0, 1); JROT4(0, 2); JROT4(0, 3); JROT4(1, 2); JROT4(1, 3); JROT4(2, 3);
    }
    int m = 0; // column of the smallest eigenvalue
    double lm = A[0][0];
#pragma unroll
    for (int c = 1; c < 4; ++c)
        if (A[c][c] < lm) { lm = A[c][c]; m = c; }
    double Xh[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) Xh[r] = m == 0 ? V[r][0] : (m == 1 ? V[r][1] : (m == 2 ? V[r][2] : V[r][3]));
    if (fabs(Xh[3]) < 1e-12) return;
    const double X[3] = {Xh[0] / Xh[3], Xh[1] / Xh[3], Xh[2] / Xh[3]};
    if (!(isfinite(X[0]) && isfinite(X[1]) && isfinite(X[2]))) return;
    Xo[0] = X[0]; Xo[1] = X[1]; Xo[2] = X[2];
    double sum = 0.0;
    int cnt = 0;
    for (int64_t o = a; o < b; ++o) {
        const int32_t cm = obs_cam[o];
        if (cm < 0 || cm >= n_cams) continue;
        double u, v;
        if (!project(cam, Rcw + 9 * (int64_t)cm, tcw + 3 * (int64_t)cm, X, u, v)) continue;
        const double du = u - obs_uv[2 * o], dv = v - obs_uv[2 * o + 1];
        sum += sqrt(du * du + dv * dv);
        ++cnt;
    }
    cnt_out[i] = cnt;
    if (cnt < 4) return;
    const double mean = sum / (double)cnt;
    err_out[i] = mean;
    ok_out[i] = isfinite(mean) ? 1 : 0;
}

} // namespace

extern "C" int32_t lvba_triangulate_tracks(int32_t device, int32_t n_cams, int64_t n_tracks, const int64_t *obs_off,
                                           const int32_t *obs_cam, const double *obs_uv, const double *Rcw, const double *tcw,
                                           const double intr[8], double *X, double *mean_reproj, int32_t *count, uint8_t *ok)
{
    if (n_cams < 1 || n_tracks < 0 || !obs_off || !Rcw || !tcw || !intr || !X || !mean_reproj || !count || !ok)
        return lvba_fail(LVBA_ERR_ARG, "null argument or n_cams < 1");
    if (n_tracks == 0) return LVBA_OK;
    const int64_t O = obs_off[n_tracks] - obs_off[0];
    if (obs_off[0] != 0 || O < 0 || (O > 0 && (!obs_cam || !obs_uv))) return lvba_fail(LVBA_ERR_ARG, "bad observation arrays");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return lvba_fail(LVBA_ERR_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return lvba_fail(LVBA_ERR_ARG, "device %d out of range [0,%d)", device, ndev);
    HIPCHK(hipSetDevice(device));
    hipStream_t s = nullptr;
    HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    struct SG { hipStream_t s; ~SG() { (void)hipStreamDestroy(s); } } sg{s};
    DevBuf d_off(s), d_cam(s), d_uv(s), d_R(s), d_t(s), d_X(s), d_err(s), d_cnt(s), d_ok(s);
    HIPCHK(d_off.alloc(8 * ((size_t)n_tracks + 1))); HIPCHK(d_cam.alloc(4 * (size_t)O)); HIPCHK(d_uv.alloc(16 * (size_t)O));
    HIPCHK(d_R.alloc(72 * (size_t)n_cams)); HIPCHK(d_t.alloc(24 * (size_t)n_cams));
    HIPCHK(d_X.alloc(24 * (size_t)n_tracks)); HIPCHK(d_err.alloc(8 * (size_t)n_tracks));
    HIPCHK(d_cnt.alloc(4 * (size_t)n_tracks)); HIPCHK(d_ok.alloc((size_t)n_tracks));
    HIPCHK(hipMemcpyAsync(d_off.p, obs_off, 8 * ((size_t)n_tracks + 1), hipMemcpyHostToDevice, s));
    if (O > 0) {
        HIPCHK(hipMemcpyAsync(d_cam.p, obs_cam, 4 * (size_t)O, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(d_uv.p, obs_uv, 16 * (size_t)O, hipMemcpyHostToDevice, s));
    }
    HIPCHK(hipMemcpyAsync(d_R.p, Rcw, 72 * (size_t)n_cams, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d_t.p, tcw, 24 * (size_t)n_cams, hipMemcpyHostToDevice, s));
    Intr c{intr[0], intr[1], intr[2], intr[3], intr[4], intr[5], intr[6], intr[7]};
    tri_kernel<<<(unsigned)((n_tracks + 127) / 128), 128, 0, s>>>(n_tracks, d_off.as<int64_t>(), d_cam.as<int32_t>(), d_uv.as<double>(),
                                                                  d_R.as<double>(), d_t.as<double>(), n_cams, c, d_X.as<double>(),
                                                                  d_err.as<double>(), d_cnt.as<int32_t>(), d_ok.as<uint8_t>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(X, d_X.p, 24 * (size_t)n_tracks, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(mean_reproj, d_err.p, 8 * (size_t)n_tracks, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(count, d_cnt.p, 4 * (size_t)n_tracks, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(ok, d_ok.p, (size_t)n_tracks, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return LVBA_OK;
}
