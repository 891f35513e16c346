// tracks_device.h -- device functions shared by tracks.hip (stand-alone DLT) and fusion.hip (per-track fusion):
//   undistortPixelToNormalized / distortNormalized / projectCameraToPixel   include/utils.hpp:169-233
//   TriangulateTrackDLT                                                     src/lvba_system.cpp:50-111
//   ComputeMeanReproj                                                       src/lvba_system.cpp:8-48
// The functions are host/device-neutral so that tests/host_emul_tracks.cpp can run them on the CPU.
// Observations are visited in the order the caller's index list gives (fusion.hip passes the iteration order the reference's
// std::unordered_map would have, see umap_order below); with no list, in storage order.
#pragma once
#include <math.h>
#include <stdint.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define LVBA_TRK_FN __host__ __device__ __forceinline__
#else // plain C++: tests/host_emul_tracks.cpp runs the same code on the CPU
#define LVBA_TRK_FN inline
#ifndef __restrict__
#define __restrict__
#endif
#endif

namespace lvba {

struct TrkIntr { double fx, fy, cx, cy, k1, k2, p1, p2; };

LVBA_TRK_FN bool trk_undistort(const TrkIntr &c, double u, double v, double &x, double &y)
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
// projectCameraToPixel: camera-frame point -> distorted pixel
LVBA_TRK_FN bool trk_project_cam(const TrkIntr &c, double X0, double X1, double Z, double &u, double &v)
{
    if (!(isfinite(X0) && isfinite(X1) && isfinite(Z)) || Z <= 1e-12) return false;
    const double x = X0 / Z, y = X1 / Z;
    const double r2 = x * x + y * y, r4 = r2 * r2;
    const double radial = 1.0 + c.k1 * r2 + c.k2 * r4;
    const double xd = x * radial + (2.0 * c.p1 * x * y + c.p2 * (r2 + 2.0 * x * x));
    const double yd = y * radial + (c.p1 * (r2 + 2.0 * y * y) + 2.0 * c.p2 * x * y);
    if (!(isfinite(xd) && isfinite(yd))) return false;
    u = c.fx * xd + c.cx;
    v = c.fy * yd + c.cy;
    return isfinite(u) && isfinite(v);
}
LVBA_TRK_FN bool trk_project(const TrkIntr &c, const double *R, const double *t, const double *X, double &u, double &v)
{
    const double X0 = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + t[0];
    const double X1 = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + t[1];
    const double Z = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2];
    return trk_project_cam(c, X0, X1, Z, u, v);
}

// one Jacobi rotation in the (p, q) plane of the symmetric 4x4 A (full storage) with eigenvector accumulation in V
#define LVBA_JROT4(p, q)                                                                                 \
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

// ---- iteration order of the reference's std::unordered_map<int,int> containers (image -> observation) ------------------------
// unique_id / best_id / kept_id_* of BuildTracksAndFuse3D are walked by range-for, and the greedy view-angle filter depends on
// that order.  With libstdc++ it is a function of reserve() and of the insertion sequence: bucket = key % bucket_count, a new
// node goes to the front of its bucket, a bucket that becomes non-empty goes to the front of the list.  bucket_count after
// reserve(n) is the first entry >= n of the rehash policy's prime table (tests/test_ref_system.py re-derives both from the
// real container; tests/test_tracks_host.py runs these functions against it).
LVBA_TRK_FN int umap_bucket_count(int reserve_n)
{
    const int primes[] = {2,     3,     5,     7,     11,    13,    17,    19,    23,    29,    31,    37,    41,    43,    47,    53,
                          59,    61,    67,    71,    73,    79,    83,    89,    97,    103,   109,   113,   127,   137,   139,   149,
                          157,   167,   179,   193,   199,   211,   227,   241,   257,   277,   293,   313,   337,   359,   383,   409,
                          439,   467,   503,   541,   577,   619,   661,   709,   761,   823,   887,   953,   1031,  1109,  1193,  1289,
                          1381,  1493,  1613,  1741,  1879,  2029,  2179,  2357,  2549,  2753,  2971,  3209,  3469,  3739,  4027,  4349,
                          4703,  5087,  5503,  5953,  6427,  6949,  7517,  8123,  8783,  9497,  10273, 11113, 12011, 12983, 14033, 15173,
                          16411, 17749, 19183, 20753, 22447, 24281, 26267, 28411, 30727, 33223, 35933, 38873, 42043, 45481, 49201, 53201,
                          57557, 62233, 67307, 72817, 78779, 85229, 92203, 99733, 107897, 116731, 126271, 136607, 147793, 159871, 172933,
                          187091, 202409};
    const int n = (int)(sizeof(primes) / sizeof(primes[0]));
    for (int i = 0; i < n; ++i)
        if (primes[i] >= reserve_n) return primes[i];
    return primes[n - 1];
}
// ins[0..m): observation offsets in insertion order (distinct images); key(e) = obs_img[base + ins[e]] (>= 0).  Writes the
// iteration order to ord[0..m).  O(m^2), m = images of one track.
LVBA_TRK_FN void umap_order(const int32_t *__restrict__ obs_img, int64_t base, const int32_t *__restrict__ ins, int m, int reserve_n,
                            int32_t *__restrict__ ord)
{
    const int B = umap_bucket_count(reserve_n);
    int k = 0;
    for (int e = m - 1; e >= 0; --e) { // buckets by the time they became non-empty, latest first
        const int be = obs_img[base + ins[e]] % B;
        bool creator = true;
        for (int f = 0; f < e && creator; ++f) creator = (obs_img[base + ins[f]] % B) != be;
        if (!creator) continue;
        for (int f = m - 1; f >= e; --f) // inside a bucket: latest insertion first
            if ((obs_img[base + ins[f]] % B) == be) ord[k++] = ins[f];
    }
}

// ComputeMeanReproj over the observations base + list[i], i < m (list == nullptr: base + i).  false: fewer than min_count
// projectable observations or a non-finite mean.
template <class UV>
LVBA_TRK_FN bool trk_mean_reproj(const TrkIntr &cam, const double *__restrict__ Rcw, const double *__restrict__ tcw, int32_t n_cams,
                                 int64_t base, const int32_t *list, int m, const int32_t *__restrict__ obs_cam,
                                 const UV *__restrict__ obs_uv, const double *X, int min_count, double &mean, int &cnt)
{
    double sum = 0.0;
    cnt = 0;
    for (int i = 0; i < m; ++i) {
        const int64_t o = base + (list ? list[i] : i);
        const int32_t cm = obs_cam[o];
        if (cm < 0 || cm >= n_cams) continue;
        double u, v;
        if (!trk_project(cam, Rcw + 9 * (int64_t)cm, tcw + 3 * (int64_t)cm, X, u, v)) continue;
        const double du = u - (double)obs_uv[2 * o], dv = v - (double)obs_uv[2 * o + 1];
        sum += sqrt(du * du + dv * dv);
        ++cnt;
    }
    if (cnt < min_count) return false;
    mean = sum / (double)cnt;
    return isfinite(mean);
}

// TriangulateTrackDLT over the observations base + list[i], i < m (list == nullptr: base + i).  X, mean, cnt are written only
// as far as the reference gets.
template <class UV>
LVBA_TRK_FN bool trk_dlt(const TrkIntr &cam, const double *__restrict__ Rcw, const double *__restrict__ tcw, int32_t n_cams,
                         int64_t base, const int32_t *list, int m, const int32_t *__restrict__ obs_cam, const UV *__restrict__ obs_uv,
                         double *X, double &mean, int &cnt)
{
    mean = INFINITY;
    cnt = 0;
    if (m < 4) return false; // selected_ids.size() < 4
    double A[4][4], V[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) { A[r][c] = 0.0; V[r][c] = r == c ? 1.0 : 0.0; }
    int rows = 0;
    for (int i = 0; i < m; ++i) {
        const int64_t o = base + (list ? list[i] : i);
        const int32_t cm = obs_cam[o];
        if (cm < 0 || cm >= n_cams) continue;
        double x, y;
        if (!trk_undistort(cam, (double)obs_uv[2 * o], (double)obs_uv[2 * o + 1], x, y)) continue;
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
    if (rows < 8) return false;
    for (int sweep = 0; sweep < 40; ++sweep) {
        const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[0][3]) + fabs(A[1][2]) + fabs(A[1][3]) + fabs(A[2][3]);
        if (off == 0.0) break;
        LVBA_JROT4(0, 1); LVBA_JROT4(0, 2); LVBA_JROT4(0, 3); LVBA_JROT4(1, 2); LVBA_JROT4(1, 3); LVBA_JROT4(2, 3);
    }
    int mn = 0; // column of the smallest eigenvalue
    double lm = A[0][0];
#pragma unroll
    for (int c = 1; c < 4; ++c)
        if (A[c][c] < lm) { lm = A[c][c]; mn = c; }
    double Xh[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) Xh[r] = mn == 0 ? V[r][0] : (mn == 1 ? V[r][1] : (mn == 2 ? V[r][2] : V[r][3]));
    if (fabs(Xh[3]) < 1e-12) return false;
    const double Xc[3] = {Xh[0] / Xh[3], Xh[1] / Xh[3], Xh[2] / Xh[3]};
    if (!(isfinite(Xc[0]) && isfinite(Xc[1]) && isfinite(Xc[2]))) return false;
    X[0] = Xc[0]; X[1] = Xc[1]; X[2] = Xc[2];
    return trk_mean_reproj(cam, Rcw, tcw, n_cams, base, list, m, obs_cam, obs_uv, Xc, 4, mean, cnt);
}

} // namespace lvba
