// fusion_device.h -- the per-track fusion of fusion.hip as host/device-neutral functions (one call = one feature track), so
// that the very code the kernel runs is also exercised on the CPU (tests/host_emul_tracks.cpp, tests/test_tracks_host.py).
// See fusion.hip for what it replaces of the reference (src/lvba_system.cpp:1016-1225, include/utils.hpp:235-284).
#pragma once
#include "tracks_device.h"

namespace lvba {

// un-fused float multiply / add (the reference's float expression, without contraction into FMAs)
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
LVBA_TRK_FN float lvba_fmul(float a, float b) { return __fmul_rn(a, b); }
LVBA_TRK_FN float lvba_fadd(float a, float b) { return __fadd_rn(a, b); }
#else
LVBA_TRK_FN float lvba_fmul(float a, float b) { volatile float r = a * b; return r; }
LVBA_TRK_FN float lvba_fadd(float a, float b) { volatile float r = a + b; return r; }
#endif

// utils.hpp:246-275 for a CV_32FC1 image (all float arithmetic, in the reference's order of operations)
LVBA_TRK_FN bool fetch_depth_bilinear(const float *__restrict__ depth, int w, int h, float u, float v, float &d_out)
{
    if (u < 0.0f || v < 0.0f || u >= (float)(w - 1) || v >= (float)(h - 1)) return false;
    const int x = (int)floorf(u), y = (int)floorf(v);
    const float du = u - (float)x, dv = v - (float)y;
    const float d00 = depth[(int64_t)y * w + x], d10 = depth[(int64_t)y * w + x + 1];
    const float d01 = depth[(int64_t)(y + 1) * w + x], d11 = depth[(int64_t)(y + 1) * w + x + 1];
    if (d00 <= 0 || d10 <= 0 || d01 <= 0 || d11 <= 0) return false;
    float d = lvba_fmul(lvba_fmul(1.0f - du, 1.0f - dv), d00);
    d = lvba_fadd(d, lvba_fmul(lvba_fmul(du, 1.0f - dv), d10));
    d = lvba_fadd(d, lvba_fmul(lvba_fmul(1.0f - du, dv), d01));
    d = lvba_fadd(d, lvba_fmul(lvba_fmul(du, dv), d11));
    d_out = d;
    return d > 0.0f;
}


// The greedy view-angle filter of :1052-1080 / :1124-1150 over the observations base + ord[i], i < m, in that order: an
// observation is kept when its ray makes at least the minimum angle with one ray already kept (or is the first).  point: the
// candidate every ray is drawn to, or nullptr for the observation's own depth point pts[3 o].  The kept observations go to
// kept[0..) in the order they were kept (= insertion order of kept_id_*, order of Track::inlier_indices) and get `bit` set.
LVBA_TRK_FN int view_filter(const int32_t *__restrict__ obs_img, int64_t base, const int32_t *__restrict__ ord, int m,
                            const double *__restrict__ Rcw, const double *__restrict__ tcw, const double *point,
                            const double *__restrict__ pts, double cos_min, double *__restrict__ dirs, uint8_t *__restrict__ flag,
                            uint8_t bit, int32_t *__restrict__ kept)
{
    int n_kept = 0;
    for (int i = 0; i < m; ++i) {
        const int64_t o = base + ord[i];
        const int32_t im = obs_img[o];
        const double *R = Rcw + 9 * (int64_t)im, *tc = tcw + 3 * (int64_t)im;
        const double *P = point ? point : pts + 3 * o;
        double dir[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) dir[r] = P[r] + (R[r] * tc[0] + R[3 + r] * tc[1] + R[6 + r] * tc[2]); // P - Cw, Cw = -Rcw^T tcw
        const double nn = sqrt(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
        if (nn < 1e-6) continue;
        dir[0] /= nn; dir[1] /= nn; dir[2] /= nn;
        double min_dot = 1.0;
        for (int q = 0; q < n_kept; ++q) {
            const double *e = dirs + 3 * (base + q);
            const double dot = dir[0] * e[0] + dir[1] * e[1] + dir[2] * e[2];
            if (dot < min_dot) min_dot = dot;
        }
        if (n_kept == 0 || min_dot <= cos_min) {
            double *e = dirs + 3 * (base + n_kept);
            e[0] = dir[0]; e[1] = dir[1]; e[2] = dir[2];
            kept[n_kept++] = ord[i];
            flag[o] |= bit;
        }
    }
    return n_kept;
}

// One feature track t (observations [obs_off[t], obs_off[t+1]) in BFS order).  pts / dirs / flag / idx are scratch arrays indexed
// like the observations (idx: two int32 per observation).  Wherever the reference walks one of its unordered_maps the
// observations are visited in that container's iteration order (umap_order, tracks_device.h).
LVBA_TRK_FN void fuse_track(int64_t t, const int64_t *__restrict__ obs_off, const int32_t *__restrict__ obs_img,
                            const float *__restrict__ obs_uv, const float *__restrict__ depth, int width, int height,
                            const double *__restrict__ Rcw, const double *__restrict__ tcw, int32_t n_images, TrkIntr cam,
                            int obser_thr, double cos_min, double reproj_thr, double *__restrict__ pts, double *__restrict__ dirs,
                            uint8_t *__restrict__ flag, int32_t *__restrict__ idx, uint8_t *__restrict__ status,
                            double *__restrict__ Xout, double *__restrict__ err_out, uint8_t *__restrict__ kept_out)
{
    const int64_t a = obs_off[t], b = obs_off[t + 1];
    const int n = (int)(b - a);
    status[t] = 0;
    Xout[3 * t] = Xout[3 * t + 1] = Xout[3 * t + 2] = 0.0;
    err_out[t] = INFINITY;
    for (int64_t o = a; o < b; ++o) kept_out[o] = 0;
    if (n < obser_thr) return; // :1000
    int32_t *la = idx + 2 * a, *lb = la + n; // two lists of observation offsets (o - a)
    // flag bits: 2 = depth point valid, 4 = depth inlier chosen for its image (best_id), 8 = kept by the depth candidate's view
    //            filter, 16 = kept by the triangulation candidate's view filter
    int n_unique = 0; // unique_id (:1003-1009): first observation of every image, in BFS order
    for (int64_t o = a; o < b; ++o) {
        const int32_t im = obs_img[o];
        bool first = im >= 0 && im < n_images;
        for (int64_t q = a; q < o && first; ++q) first = obs_img[q] != im;
        flag[o] = 0;
        if (first) la[n_unique++] = (int32_t)(o - a);
    }
    if (n_unique < obser_thr) return; // :1012
    // ------------------------------------------------------------------ triangulation candidate (:1108-1160)
    bool tri_ok = false;
    double Xt[3] = {0, 0, 0}, m_tri = INFINITY;
    if (n_unique >= 4) {
        umap_order(obs_img, a, la, n_unique, n, lb); // unique_id.reserve(component.size())
        double Xs[3], ms;
        int cs;
        if (trk_dlt(cam, Rcw, tcw, n_images, a, lb, n_unique, obs_img, obs_uv, Xs, ms, cs)) {
            const int n_kept = view_filter(obs_img, a, lb, n_unique, Rcw, tcw, Xs, pts, cos_min, dirs, flag, 16, la);
            if (n_kept >= 4) {
                umap_order(obs_img, a, la, n_kept, n_unique, lb); // kept_id_tri.reserve(unique_id.size())
                int ct;
                if (trk_dlt(cam, Rcw, tcw, n_images, a, lb, n_kept, obs_img, obs_uv, Xt, m_tri, ct)) tri_ok = m_tri <= reproj_thr;
            }
        }
    }
    // ------------------------------------------------------------------ depth-fused candidate (:1016-1106)
    bool depth_ok = false;
    double Xd[3] = {0, 0, 0}, m_depth = INFINITY;
    if (depth) {
        int n_valid = 0;
        int64_t first_valid = -1;
        for (int64_t o = a; o < b; ++o) {
            const int32_t im = obs_img[o];
            if (im < 0 || im >= n_images) continue;
            const float u = obs_uv[2 * o], v = obs_uv[2 * o + 1];
            float d;
            if (!fetch_depth_bilinear(depth + (int64_t)im * width * height, width, height, u, v, d)) continue;
            double x, y;
            if (!trk_undistort(cam, (double)u, (double)v, x, y)) continue;
            const double dd = (double)d;
            const double Xc[3] = {x * dd, y * dd, dd};
            if (!(isfinite(Xc[0]) && isfinite(Xc[1]) && isfinite(Xc[2]))) continue;
            const double *R = Rcw + 9 * (int64_t)im, *tc = tcw + 3 * (int64_t)im;
            // camToWorld (utils.hpp:277-284): Rwc Xc + twc with twc = -(Rwc tcw)
            double *p = pts + 3 * o;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const double twc = -(R[r] * tc[0] + R[3 + r] * tc[1] + R[6 + r] * tc[2]);
                p[r] = (R[r] * Xc[0] + R[3 + r] * Xc[1] + R[6 + r] * Xc[2]) + twc;
            }
            flag[o] |= 2;
            if (first_valid < 0) first_valid = o;
            ++n_valid;
        }
        if (n_valid >= obser_thr) {
            const double *anc = pts + 3 * first_valid;
            int n_inl = 0, n_best = 0; // inliers (:1039-1044) and best_id: the first inlier of every image (:1046-1051)
            for (int64_t o = a; o < b; ++o) {
                if (!(flag[o] & 2)) continue;
                const double *p = pts + 3 * o;
                const double dx = p[0] - anc[0], dy = p[1] - anc[1], dz = p[2] - anc[2];
                if (!(sqrt(dx * dx + dy * dy + dz * dz) < 0.12)) continue;
                ++n_inl;
                bool first = true;
                for (int64_t q = a; q < o && first; ++q) first = !((flag[q] & 4) && obs_img[q] == obs_img[o]);
                if (!first) continue;
                flag[o] |= 4;
                la[n_best++] = (int32_t)(o - a);
            }
            if (n_best >= obser_thr) {
                umap_order(obs_img, a, la, n_best, n_inl, lb); // best_id.reserve(inliers.size())
                double sum[3] = {0, 0, 0};
                for (int i = 0; i < n_best; ++i) {
                    const double *p = pts + 3 * (a + lb[i]);
                    sum[0] += p[0]; sum[1] += p[1]; sum[2] += p[2];
                }
                Xd[0] = sum[0] / (double)n_best; Xd[1] = sum[1] / (double)n_best; Xd[2] = sum[2] / (double)n_best;
                const int n_kept = view_filter(obs_img, a, lb, n_best, Rcw, tcw, nullptr, pts, cos_min, dirs, flag, 8, la);
                if (n_kept >= obser_thr) {
                    umap_order(obs_img, a, la, n_kept, n_best, lb); // kept_id_depth.reserve(best_id.size())
                    int cnt;
                    if (trk_mean_reproj(cam, Rcw, tcw, n_images, a, lb, n_kept, obs_img, obs_uv, Xd, obser_thr, m_depth, cnt))
                        depth_ok = m_depth <= reproj_thr;
                }
            }
        }
    }
    // ------------------------------------------------------------------ selection (:1162-1205)
    bool use_tri;
    if (depth_ok && tri_ok) use_tri = m_tri < m_depth;
    else if (tri_ok) use_tri = true;
    else if (depth_ok) use_tri = false;
    else return;
    const double *X = use_tri ? Xt : Xd;
    if (!(isfinite(X[0]) && isfinite(X[1]) && isfinite(X[2]))) return;
    if (fabs(X[0]) <= 1e-12 && fabs(X[1]) <= 1e-12 && fabs(X[2]) <= 1e-12) return; // isZero(1e-12)
    status[t] = use_tri ? 1 : 2;
    Xout[3 * t] = X[0]; Xout[3 * t + 1] = X[1]; Xout[3 * t + 2] = X[2];
    err_out[t] = use_tri ? m_tri : m_depth;
    const uint8_t bit = use_tri ? 16 : 8;
    for (int64_t o = a; o < b; ++o) kept_out[o] = (flag[o] & bit) ? 1 : 0;
}

} // namespace lvba
