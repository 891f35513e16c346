// TEST-ONLY: runs the host loop of include/lvba_adapter.hpp (build_tracks_and_fuse_with: match graph, BFS components, batched
// fusion, the reference's retries, track order) on the CPU, with the per-track device code of fusion_device.h compiled for the
// host standing in for lvba_fuse_tracks -- so that tests/test_ref_system.py can hold the C++ binding a maintainer would use
// against the reference's own BuildTracksAndFuse3D without a GPU.  Never part of liblvba_hip.so; not a fallback.
#include <cmath>
#include <cstdint>
#include <utility>
#include <vector>
#include "../include/lvba_adapter.hpp"
#include "../global-lvba_amd/csrc/fusion_device.h"

namespace {
struct KP { float x, y; };
struct TrackT {
    double Xw_fused[3];
    std::vector<std::pair<int, int>> observations;
    std::vector<int> inlier_indices;
};
struct EmulFuse {
    const std::vector<std::vector<KP>> &kps;
    const float *depth;
    int w, h, n_img;
    const double *Rcw, *tcw, *intr;
    int obser_thr;
    double cos_min, thr;
    lvba::FusedBatch operator()(const std::vector<std::vector<std::pair<int, int>>> &comps) const
    {
        lvba::FusedBatch out;
        std::vector<int32_t> img;
        std::vector<float> uv;
        lvba::pack_components(comps, kps, out.off, img, uv);
        const size_t O = img.size(), C = comps.size();
        out.status.assign(C + 1, 0); out.kept.assign(O + 1, 0); out.X.assign(3 * (C + 1), 0.0);
        std::vector<double> err(C + 1), pts(3 * (O + 1)), dirs(3 * (O + 1));
        std::vector<uint8_t> flag(O + 1);
        std::vector<int32_t> idx(2 * (O + 1));
        const lvba::TrkIntr cam{intr[0], intr[1], intr[2], intr[3], intr[4], intr[5], intr[6], intr[7]};
        for (size_t t = 0; t < C; ++t)
            lvba::fuse_track((int64_t)t, out.off.data(), img.data(), uv.data(), depth, w, h, Rcw, tcw, n_img, cam, obser_thr, cos_min, thr,
                             pts.data(), dirs.data(), flag.data(), idx.data(), out.status.data(), out.X.data(), err.data(), out.kept.data());
        return out;
    }
};
} // namespace

// kp_xy: all key points, images back to back; pairs [n_pairs][2] (i < j), matches of pair q = matches[match_off[q] .. match_off[q+1])
// Outputs (caller-sized by the total number of key points): track_len, X [.][3], obs [.][2], inlier flags per observation.
extern "C" int adapter_build_tracks(int n_img, const int32_t *nk, const float *kp_xy, int n_pairs, const int32_t *pairs,
                                    const int64_t *match_off, const int32_t *matches, const float *depth, int w, int h,
                                    const double *Rcw, const double *tcw, const double *intr, int obser_thr, double angle_deg,
                                    double thr, int32_t *track_len, double *X, int32_t *obs, uint8_t *inlier, int32_t *obs_to_track)
{
    std::vector<std::vector<KP>> kps(n_img);
    size_t k = 0;
    for (int i = 0; i < n_img; ++i) {
        kps[i].resize(nk[i]);
        for (int j = 0; j < nk[i]; ++j, ++k) kps[i][j] = KP{kp_xy[2 * k], kp_xy[2 * k + 1]};
    }
    std::vector<std::vector<std::pair<int, int>>> all_matches((size_t)n_img * (n_img - 1) / 2);
    for (int q = 0; q < n_pairs; ++q) {
        const int i = pairs[2 * q], j = pairs[2 * q + 1];
        auto &m = all_matches[(size_t)(i * (2 * n_img - i - 1) / 2 + (j - i - 1))];
        for (int64_t e = match_off[q]; e < match_off[q + 1]; ++e) m.emplace_back(matches[2 * e], matches[2 * e + 1]);
    }
    const EmulFuse fuse{kps, depth, w, h, n_img, Rcw, tcw, intr, obser_thr, std::cos(angle_deg * M_PI / 180.0), thr};
    std::vector<TrackT> tracks;
    std::vector<std::vector<int>> o2t;
    lvba::build_tracks_and_fuse_with<TrackT>(kps, all_matches, obser_thr, fuse, tracks, &o2t);
    size_t o = 0;
    for (size_t t = 0; t < tracks.size(); ++t) {
        track_len[t] = (int32_t)tracks[t].observations.size();
        for (int r = 0; r < 3; ++r) X[3 * t + r] = tracks[t].Xw_fused[r];
        const size_t o0 = o;
        for (const auto &ob : tracks[t].observations) { obs[2 * o] = ob.first; obs[2 * o + 1] = ob.second; inlier[o] = 0; ++o; }
        for (int i : tracks[t].inlier_indices) inlier[o0 + (size_t)i] = 1;
    }
    k = 0;
    for (int i = 0; i < n_img; ++i)
        for (int j = 0; j < nk[i]; ++j, ++k) obs_to_track[k] = o2t[i][j];
    return (int)tracks.size();
}
