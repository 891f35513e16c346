// Host build of the per-track device code (global-lvba_amd/csrc/tracks_device.h, fusion_device.h): the functions the
// kernels call, compiled for the CPU with g++ so that tests/test_tracks_host.py can hold them against the oracle without a GPU.
#include <cstdint>
#include <vector>
#include "../global-lvba_amd/csrc/fusion_device.h"

using namespace lvba;

extern "C" {

// tri_kernel of tracks.hip: one DLT + mean reprojection per track over all its observations (double keypoints)
void emul_triangulate(int64_t n_tracks, const int64_t *obs_off, const int32_t *obs_cam, const double *obs_uv, const double *Rcw,
                      const double *tcw, int32_t n_cams, const double *intr, double *X, double *err, int32_t *cnt, uint8_t *ok)
{
    const TrkIntr cam{intr[0], intr[1], intr[2], intr[3], intr[4], intr[5], intr[6], intr[7]};
    for (int64_t i = 0; i < n_tracks; ++i) {
        double x[3] = {0, 0, 0}, mean;
        int c;
        const bool good = trk_dlt(cam, Rcw, tcw, n_cams, obs_off[i], (const int32_t *)nullptr, (int)(obs_off[i + 1] - obs_off[i]), obs_cam,
                                  obs_uv, x, mean, c);
        X[3 * i] = x[0]; X[3 * i + 1] = x[1]; X[3 * i + 2] = x[2];
        err[i] = mean; cnt[i] = c; ok[i] = good ? 1 : 0;
    }
}

// fuse_kernel of fusion.hip: depth [n_images][height][width] or NULL
void emul_fuse_tracks(int64_t n_tracks, const int64_t *obs_off, const int32_t *obs_img, const float *obs_uv, const float *depth,
                      int width, int height, const double *Rcw, const double *tcw, int32_t n_images, const double *intr,
                      int obser_thr, double cos_min, double reproj_thr, uint8_t *status, double *X, double *err, uint8_t *kept)
{
    const TrkIntr cam{intr[0], intr[1], intr[2], intr[3], intr[4], intr[5], intr[6], intr[7]};
    const int64_t O = obs_off[n_tracks];
    std::vector<double> pts(3 * (size_t)(O + 1)), dirs(3 * (size_t)(O + 1));
    std::vector<uint8_t> flag((size_t)O + 1);
    std::vector<int32_t> idx(2 * (size_t)(O + 1));
    for (int64_t t = 0; t < n_tracks; ++t)
        fuse_track(t, obs_off, obs_img, obs_uv, depth, width, height, Rcw, tcw, n_images, cam, obser_thr, cos_min, reproj_thr,
                   pts.data(), dirs.data(), flag.data(), idx.data(), status, X, err, kept);
}

// umap_order of tracks_device.h: keys in insertion order -> positions in iteration order; returns the bucket count
int emul_umap_order(int reserve_n, int m, const int32_t *keys, int32_t *order)
{
    std::vector<int32_t> ins((size_t)m);
    for (int i = 0; i < m; ++i) ins[i] = i;
    umap_order(keys, 0, ins.data(), m, reserve_n, order);
    return umap_bucket_count(reserve_n);
}

float emul_fetch_depth(const float *depth, int w, int h, float u, float v, int *ok)
{
    float d = 0.0f;
    *ok = fetch_depth_bilinear(depth, w, h, u, v, d) ? 1 : 0;
    return d;
}

} // extern "C"
