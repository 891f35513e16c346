// lvba_adapter.hpp -- the reference-side binding: what a Global-LVBA maintainer adds to call liblvba_hip.so
// instead of BALM2::damping_iter.  Header-only, duck-typed on the reference's own types so it needs neither
// Eigen nor PCL to compile:
//   VoxHess : has `win_size` and `plvec_voxels`, a sequence of `const std::vector<PointCluster>*`
//             (reference include/BALM/bavoxel.hpp:35-36)
//   PointCluster : `.P(r,c)`, `.v[r]` (or `.v(r)`), `.N`              (include/BALM/tools.hpp:407-412)
//   IMUST   : `.R(r,c)`, `.p[r]`                                       (include/BALM/tools.hpp:147-151)
// Usage at src/lvba_system.cpp:264 and :386 --
//       // opt_lsv->damping_iter(x_win, *voxhess);
//       lvba::damping_iter_hip(x_win, *voxhess);
// or, replacing the whole block src/lvba_system.cpp:247-264 / :365-386 (cut_voxel loop, recut, tras_opt, damping_iter)
// so that the octree is never built on the CPU --
//       lvba::VoxelMap surf_map(pl_win, x_win, stage1_root_voxel_size_, stage1_eigen_ratio_array_.data());
//       if (surf_map.info().n_voxels >= 3 * x_win.size()) surf_map.damping_iter(x_win);
#pragma once
#include <algorithm>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>
#include "lvba_hip.h"

namespace lvba {

// Packs VOX_HESS::plvec_voxels (dense win_size slots per voxel, mostly empty) into the CSR arrays of
// lvba_balm_create: the non-empty slots of every voxel, ascending pose index.
template <class VoxHess>
void pack_voxhess(const VoxHess &vh, std::vector<int64_t> &voxel_off, std::vector<int32_t> &pose_idx,
                  std::vector<double> &clusters)
{
    voxel_off.assign(1, 0);
    pose_idx.clear();
    clusters.clear();
    for (const auto *sig_orig : vh.plvec_voxels) {
        for (int i = 0; i < vh.win_size; ++i) {
            const auto &c = (*sig_orig)[i];
            if (c.N == 0) continue;                       // bavoxel.hpp:90-91
            pose_idx.push_back(i);
            const double rec[10] = {c.P(0, 0), c.P(0, 1), c.P(0, 2), c.P(1, 1), c.P(1, 2), c.P(2, 2),
                                    c.v[0], c.v[1], c.v[2], static_cast<double>(c.N)};
            clusters.insert(clusters.end(), rec, rec + 10);
        }
        voxel_off.push_back(static_cast<int64_t>(pose_idx.size()));
    }
}

// Drop-in for BALM2::damping_iter(x_stats, voxhess)  (bavoxel.hpp:662).  Refines x_stats in place.
// Returns the LM trace (the quantities of the commented printf at bavoxel.hpp:737).
template <class PoseVec, class VoxHess>
std::vector<lvba_lm_trace> damping_iter_hip(PoseVec &x_stats, const VoxHess &voxhess, int device = 0)
{
    std::vector<int64_t> off;
    std::vector<int32_t> idx;
    std::vector<double> clu;
    pack_voxhess(voxhess, off, idx, clu);
    const int32_t N = voxhess.win_size;
    std::vector<double> poses(12 * static_cast<size_t>(N));
    for (int j = 0; j < N; ++j) {
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) poses[12 * j + 3 * r + c] = x_stats[j].R(r, c);
        for (int r = 0; r < 3; ++r) poses[12 * j + 9 + r] = x_stats[j].p[r];
    }
    lvba_balm_t h = nullptr;
    int32_t rc = lvba_balm_create(N, static_cast<int64_t>(off.size()) - 1, off.data(), idx.data(), clu.data(), device, &h);
    if (rc != LVBA_OK) throw std::runtime_error(std::string("lvba_balm_create: ") + lvba_last_error());
    lvba_balm_opts opts;
    lvba_balm_default_opts(&opts);                       // 10 iterations, u=0.01, v=2 (bavoxel.hpp:664,686)
    std::vector<lvba_lm_trace> trace(opts.max_iter);
    int32_t n_trace = 0;
    rc = lvba_balm_refine(h, poses.data(), &opts, trace.data(), &n_trace);
    lvba_balm_destroy(h);
    if (rc < 0) throw std::runtime_error(std::string("lvba_balm_refine: ") + lvba_last_error());
    // rc > 0 (zero pivot / non-finite cost): the reference leaves LDLT failure unchecked (bavoxel.hpp:707-710);
    // here the poses of the last accepted step are returned.
    trace.resize(n_trace);
    for (int j = 0; j < N; ++j) {
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) x_stats[j].R(r, c) = poses[12 * j + 3 * r + c];
        for (int r = 0; r < 3; ++r) x_stats[j].p[r] = poses[12 * j + 9 + r];
    }
    return trace;
}

// RAII wrapper of the device-resident voxel map.  CloudPtrVec: sequence of (smart) pointers to clouds with
// `.points` (contiguous PointT whose first three floats are x, y, z -- pcl::PointXYZINormal and friends).
class VoxelMap {
  public:
    template <class CloudPtrVec, class PoseVec>
    VoxelMap(const CloudPtrVec &clouds, const PoseVec &x_buf, double voxel_size, const float *eigen_ratio_array,
             int device = 0)
    {
        const int32_t n = static_cast<int32_t>(x_buf.size());
        std::vector<const void *> ptr(n);
        std::vector<int64_t> cnt(n);
        int32_t stride = 12;
        for (int32_t j = 0; j < n; ++j) {
            const auto &pts = clouds[j]->points;
            ptr[j] = pts.data();
            cnt[j] = static_cast<int64_t>(pts.size());
            stride = static_cast<int32_t>(sizeof(pts[0]));
        }
        std::vector<double> poses;
        pack_poses(x_buf, poses);
        lvba_voxel_opts o;
        lvba_voxel_default_opts(&o);
        o.voxel_size = voxel_size;
        if (eigen_ratio_array)
            for (int k = 0; k < 4; ++k) o.eigen_ratio[k] = eigen_ratio_array[k];
        if (lvba_voxmap_build(device, n, ptr.data(), cnt.data(), stride, poses.data(), &o, &h_) != LVBA_OK)
            throw std::runtime_error(std::string("lvba_voxmap_build: ") + lvba_last_error());
    }
    ~VoxelMap() { lvba_voxmap_destroy(h_); }
    VoxelMap(const VoxelMap &) = delete;
    VoxelMap &operator=(const VoxelMap &) = delete;

    lvba_voxmap_info_t info() const
    {
        lvba_voxmap_info_t i;
        lvba_voxmap_info(h_, &i);
        return i;
    }
    // tras_opt + BALM2::damping_iter on the map's admitted voxels; refines x_stats in place.
    template <class PoseVec>
    std::vector<lvba_lm_trace> damping_iter(PoseVec &x_stats) const
    {
        lvba_balm_t b = nullptr;
        if (lvba_voxmap_to_balm(h_, &b) != LVBA_OK)
            throw std::runtime_error(std::string("lvba_voxmap_to_balm: ") + lvba_last_error());
        std::vector<double> poses;
        pack_poses(x_stats, poses);
        lvba_balm_opts opts;
        lvba_balm_default_opts(&opts);
        std::vector<lvba_lm_trace> trace(opts.max_iter);
        int32_t n_trace = 0;
        const int32_t rc = lvba_balm_refine(b, poses.data(), &opts, trace.data(), &n_trace);
        lvba_balm_destroy(b);
        if (rc < 0) throw std::runtime_error(std::string("lvba_balm_refine: ") + lvba_last_error());
        trace.resize(n_trace);
        for (size_t j = 0; j < x_stats.size(); ++j) {
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) x_stats[j].R(r, c) = poses[12 * j + 3 * r + c];
            for (int r = 0; r < 3; ++r) x_stats[j].p[r] = poses[12 * j + 9 + r];
        }
        return trace;
    }
    // recompute_local_planes (src/lvba_system.cpp:1531-1565): Xs = sequence of 3-element arrays.
    template <class PointVec>
    void find_planes(const PointVec &Xs, std::vector<double> &plane_nd, std::vector<uint8_t> &valid) const
    {
        std::vector<double> X(3 * Xs.size());
        for (size_t i = 0; i < Xs.size(); ++i)
            for (int r = 0; r < 3; ++r) X[3 * i + r] = Xs[i][r];
        plane_nd.assign(4 * Xs.size(), 0.0);
        valid.assign(Xs.size(), 0);
        if (lvba_voxmap_find_planes(h_, static_cast<int64_t>(Xs.size()), X.data(), plane_nd.data(), valid.data()) != LVBA_OK)
            throw std::runtime_error(std::string("lvba_voxmap_find_planes: ") + lvba_last_error());
    }
    lvba_voxmap_t handle() const { return h_; }

  private:
    template <class PoseVec>
    static void pack_poses(const PoseVec &x, std::vector<double> &poses)
    {
        poses.resize(12 * x.size());
        for (size_t j = 0; j < x.size(); ++j) {
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) poses[12 * j + 3 * r + c] = x[j].R(r, c);
            for (int r = 0; r < 3; ++r) poses[12 * j + 9 + r] = x[j].p[r];
        }
    }
    lvba_voxmap_t h_ = nullptr;
};

// Drop-in for the compute of LvbaSystem::runLidarBA (src/lvba_system.cpp:312-410): window BA -> anchors -> global stage 1 /
// stage 2 -> composed frame poses.  x_buf is refined in place (what the reference stores into dataset_io_->x_buf_, :406).
//       lvba_lidar_ba_opts o; lvba_lidar_ba_default_opts(&o);   // then copy window_ba_size_, anchor_leaf_size_, stage sizes ...
//       lvba::lidar_ba(dataset_io_->pl_fulls_, dataset_io_->x_buf_, o);
template <class CloudPtrVec, class PoseVec>
lvba_lidar_ba_report lidar_ba(const CloudPtrVec &clouds, PoseVec &x_buf, const lvba_lidar_ba_opts &opts, int device = 0)
{
    const int32_t n = static_cast<int32_t>(x_buf.size());
    std::vector<const void *> ptr(n);
    std::vector<int64_t> cnt(n);
    int32_t stride = 12;
    for (int32_t j = 0; j < n; ++j) {
        const auto &pts = clouds[j]->points;
        ptr[j] = pts.data();
        cnt[j] = static_cast<int64_t>(pts.size());
        stride = static_cast<int32_t>(sizeof(pts[0]));
    }
    std::vector<double> poses(12 * static_cast<size_t>(n));
    for (int32_t j = 0; j < n; ++j) {
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) poses[12 * j + 3 * r + c] = x_buf[j].R(r, c);
        for (int r = 0; r < 3; ++r) poses[12 * j + 9 + r] = x_buf[j].p[r];
    }
    lvba_scans_t scans = nullptr;
    if (lvba_scans_create(device, n, ptr.data(), cnt.data(), stride, &scans) != LVBA_OK)
        throw std::runtime_error(std::string("lvba_scans_create: ") + lvba_last_error());
    lvba_lidar_ba_report rep;
    const int32_t rc = lvba_lidar_ba(scans, poses.data(), &opts, poses.data(), &rep);
    lvba_scans_destroy(scans);
    if (rc != LVBA_OK) throw std::runtime_error(std::string("lvba_lidar_ba: ") + lvba_last_error());
    for (int32_t j = 0; j < n; ++j) {
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) x_buf[j].R(r, c) = poses[12 * j + 3 * r + c];
        for (int r = 0; r < 3; ++r) x_buf[j].p[r] = poses[12 * j + 9 + r];
    }
    return rep;
}

// The same over several GPUs of one node: the window stage's windows are dealt out to `devices` in contiguous runs of whole
// windows (lvba_window_split), every device holds only its own frames, the global stages run on devices[0] over the gathered
// anchors (lvba_lidar_ba_multi).  devices.size() == 1 is lidar_ba above.
template <class CloudPtrVec, class PoseVec>
lvba_lidar_ba_report lidar_ba(const CloudPtrVec &clouds, PoseVec &x_buf, const lvba_lidar_ba_opts &opts, const std::vector<int> &devices)
{
    if (devices.empty()) throw std::runtime_error("lvba::lidar_ba: empty device list");
    if (devices.size() == 1 || !opts.window_enable) return lidar_ba(clouds, x_buf, opts, devices[0]);
    const int32_t n = static_cast<int32_t>(x_buf.size()), D = static_cast<int32_t>(devices.size());
    std::vector<int32_t> fb(static_cast<size_t>(D) + 1);
    if (lvba_window_split(n, opts.window.window_size, D, fb.data()) != LVBA_OK)
        throw std::runtime_error(std::string("lvba_window_split: ") + lvba_last_error());
    std::vector<lvba_scans_t> shares;
    auto drop = [&]() { for (lvba_scans_t q : shares) lvba_scans_destroy(q); shares.clear(); };
    for (int32_t k = 0; k < D; ++k) {
        const int32_t a = fb[k], b = fb[k + 1];
        if (b <= a) continue;
        std::vector<const void *> ptr(static_cast<size_t>(b - a));
        std::vector<int64_t> cnt(static_cast<size_t>(b - a));
        int32_t stride = 12;
        for (int32_t j = a; j < b; ++j) {
            const auto &pts = clouds[j]->points;
            ptr[j - a] = pts.data();
            cnt[j - a] = static_cast<int64_t>(pts.size());
            stride = static_cast<int32_t>(sizeof(pts[0]));
        }
        lvba_scans_t sc = nullptr;
        if (lvba_scans_create(devices[static_cast<size_t>(k)], b - a, ptr.data(), cnt.data(), stride, &sc) != LVBA_OK) {
            const std::string msg = lvba_last_error();
            drop();
            throw std::runtime_error("lvba_scans_create: " + msg);
        }
        shares.push_back(sc);
    }
    std::vector<double> poses(12 * static_cast<size_t>(n));
    for (int32_t j = 0; j < n; ++j) {
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) poses[12 * j + 3 * r + c] = x_buf[j].R(r, c);
        for (int r = 0; r < 3; ++r) poses[12 * j + 9 + r] = x_buf[j].p[r];
    }
    lvba_lidar_ba_report rep;
    const int32_t rc = lvba_lidar_ba_multi(static_cast<int32_t>(shares.size()), shares.data(), poses.data(), &opts, poses.data(), &rep);
    const std::string msg = rc != LVBA_OK ? lvba_last_error() : "";
    drop();
    if (rc != LVBA_OK) throw std::runtime_error("lvba_lidar_ba_multi: " + msg);
    for (int32_t j = 0; j < n; ++j) {
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) x_buf[j].R(r, c) = poses[12 * j + 3 * r + c];
        for (int r = 0; r < 3; ++r) x_buf[j].p[r] = poses[12 * j + 9 + r];
    }
    return rep;
}

// ---- LvbaSystem::BuildTracksAndFuse3D (src/lvba_system.cpp:921-1263) ---------------------------------------------------------
// One batch of components (observations = (image, key point) in BFS order) through lvba_fuse_tracks.  status[c] 0 = dropped,
// 1 = triangulated, 2 = depth-fused; X [C][3]; kept: one flag per observation, components back to back.
struct FusedBatch {
    std::vector<uint8_t> status, kept;
    std::vector<double> X;
    std::vector<int64_t> off;
};
template <class Component, class KeypointTable>
void pack_components(const std::vector<Component> &comps, const KeypointTable &all_keypoints, std::vector<int64_t> &off,
                     std::vector<int32_t> &img, std::vector<float> &uv)
{
    off.assign(comps.size() + 1, 0);
    for (size_t c = 0; c < comps.size(); ++c) off[c + 1] = off[c] + static_cast<int64_t>(comps[c].size());
    img.resize(static_cast<size_t>(off.back()));
    uv.resize(2 * static_cast<size_t>(off.back()));
    for (size_t c = 0; c < comps.size(); ++c)
        for (size_t i = 0; i < comps[c].size(); ++i) {
            const size_t o = static_cast<size_t>(off[c]) + i;
            const int im = comps[c][i].first, kp = comps[c][i].second;
            img[o] = im;
            uv[2 * o] = all_keypoints[im][kp].x;
            uv[2 * o + 1] = all_keypoints[im][kp].y;
        }
}
// The GPU batch: a callable (comps) -> FusedBatch bound to the cameras, the depth images and the thresholds.
template <class KeypointTable, class RotVec, class TransVec>
struct GpuFuse {
    const KeypointTable &all_keypoints;
    std::vector<double> R, t;
    int32_t n_img;
    const double *intr;
    lvba_depth_t depth;
    const lvba_fuse_opts *opts;
    int device;
    GpuFuse(const KeypointTable &kps, const RotVec &Rcw_all, const TransVec &tcw_all, const double intr_[8], lvba_depth_t depth_,
            const lvba_fuse_opts *opts_, int device_)
        : all_keypoints(kps), n_img(static_cast<int32_t>(Rcw_all.size())), intr(intr_), depth(depth_), opts(opts_), device(device_)
    {
        R.resize(9 * static_cast<size_t>(n_img));
        t.resize(3 * static_cast<size_t>(n_img));
        for (int32_t m = 0; m < n_img; ++m)
            for (int r = 0; r < 3; ++r) {
                for (int c = 0; c < 3; ++c) R[9 * m + 3 * r + c] = Rcw_all[m](r, c);
                t[3 * m + r] = tcw_all[m][r];
            }
    }
    template <class Component>
    FusedBatch operator()(const std::vector<Component> &comps) const
    {
        FusedBatch out;
        std::vector<int32_t> img;
        std::vector<float> uv;
        pack_components(comps, all_keypoints, out.off, img, uv);
        out.status.assign(comps.size() + 1, 0);
        out.kept.assign(img.size() + 1, 0);
        out.X.assign(3 * (comps.size() + 1), 0.0);
        std::vector<double> err(comps.size() + 1);
        if (lvba_fuse_tracks(device, depth, n_img, R.data(), t.data(), intr, static_cast<int64_t>(comps.size()), out.off.data(),
                             img.data(), uv.data(), opts, out.status.data(), out.X.data(), err.data(), out.kept.data()) != LVBA_OK)
            throw std::runtime_error(std::string("lvba_fuse_tracks: ") + lvba_last_error());
        return out;
    }
};
template <class Track, class Component>
Track make_track(const Component &comp, const FusedBatch &fb, size_t c)
{
    Track tr;
    tr.Xw_fused[0] = fb.X[3 * c]; tr.Xw_fused[1] = fb.X[3 * c + 1]; tr.Xw_fused[2] = fb.X[3 * c + 2];
    tr.observations.assign(comp.begin(), comp.end());
    for (size_t i = 0; i < comp.size(); ++i)
        if (fb.kept[static_cast<size_t>(fb.off[c]) + i]) tr.inlier_indices.push_back(static_cast<int>(i));
    return tr;
}

// Drop-in for the per-component body of LvbaSystem::BuildTracksAndFuse3D (src/lvba_system.cpp:1000-1225) for a caller that
// keeps its own BFS: the components that pass the two size checks, each in ONE BFS order, in one call:
//       lvba::fuse_components<Track>(comps, all_keypoints_, Rcw_all_optimized_, tcw_all_optimized_, intr, depth /*or nullptr*/,
//                                    opts, tracks_, comp_track);       // comp_track[c] = index into tracks_ or -1
// (The reference meets a dropped component again from its next member -- build_tracks_and_fuse below does that too.)
// Track needs the reference's members Xw_fused, observations, inlier_indices (include/utils.hpp:150-156); keypoints .x / .y.
template <class Track, class Component, class KeypointTable, class RotVec, class TransVec>
void fuse_components(const std::vector<Component> &comps, const KeypointTable &all_keypoints, const RotVec &Rcw_all,
                     const TransVec &tcw_all, const double intr[8], lvba_depth_t depth, const lvba_fuse_opts *opts,
                     std::vector<Track> &tracks, std::vector<int> &comp_track, int device = 0)
{
    const GpuFuse<KeypointTable, RotVec, TransVec> fuse(all_keypoints, Rcw_all, tcw_all, intr, depth, opts, device);
    const FusedBatch fb = fuse(comps);
    comp_track.assign(comps.size(), -1);
    for (size_t c = 0; c < comps.size(); ++c) {
        if (!fb.status[c]) continue;
        comp_track[c] = static_cast<int>(tracks.size());
        tracks.push_back(make_track<Track>(comps[c], fb, c));
    }
}

// The whole track loop (:921-1263) with the fusion batched.  all_matches: the reference's table in pairIndex order
// (include/utils.hpp:286-291), entries (key point in image i, key point in image j), i < j.  `fuse` is a callable
// (std::vector<std::vector<std::pair<int,int>>>) -> FusedBatch; see build_tracks_and_fuse for the GPU one.
// What the reference does one component at a time is done in rounds: round r fuses, in one batch, the r-th attempt of every
// component that is still dropped -- the reference releases a dropped component (:1197, :1203) and meets it again at its next
// member in scan order, i.e. in another BFS order.  tracks come out in the reference's order (by the key point the successful
// BFS started from); obs_to_track (optional) as the reference leaves it: track index, or -1.
template <class Track, class KeypointTable, class MatchTable, class Fuse>
void build_tracks_and_fuse_with(const KeypointTable &all_keypoints, const MatchTable &all_matches, int obser_thr, Fuse &&fuse,
                                std::vector<Track> &tracks, std::vector<std::vector<int>> *obs_to_track = nullptr)
{
    typedef std::pair<int, int> Obs;
    typedef std::vector<Obs> Comp;
    const int N = static_cast<int>(all_keypoints.size());
    std::vector<std::vector<std::vector<Obs>>> adj(N);
    for (int i = 0; i < N; ++i) adj[i].resize(all_keypoints[i].size());
    for (int i = 0; i < N - 1; ++i)
        for (int j = i + 1; j < N; ++j) {
            const size_t idx = static_cast<size_t>(i * (2 * N - i - 1) / 2 + (j - i - 1)); // pairIndex
            if (idx >= all_matches.size()) continue;
            for (const auto &m : all_matches[idx]) {
                const int ki = m.first, kj = m.second;
                if (ki < 0 || kj < 0 || ki >= static_cast<int>(adj[i].size()) || kj >= static_cast<int>(adj[j].size())) continue;
                adj[i][ki].push_back(Obs(j, kj));
                adj[j][kj].push_back(Obs(i, ki));
            }
        }
    std::vector<std::vector<int>> stamp(N);
    for (int i = 0; i < N; ++i) stamp[i].assign(all_keypoints[i].size(), 0);
    int epoch = 0;
    auto bfs = [&](const Obs &start) {
        ++epoch;
        Comp comp;
        comp.push_back(start);
        stamp[start.first][start.second] = epoch;
        for (size_t head = 0; head < comp.size(); ++head) {
            const Obs cur = comp[head];
            for (const Obs &nb : adj[cur.first][cur.second])
                if (stamp[nb.first][nb.second] != epoch) {
                    stamp[nb.first][nb.second] = epoch;
                    comp.push_back(nb);
                }
        }
        return comp;
    };
    // components that pass the size checks (:1000, :1012), members in scan order
    std::vector<Comp> members;
    {
        std::vector<std::vector<char>> seen(N);
        for (int i = 0; i < N; ++i) seen[i].assign(all_keypoints[i].size(), 0);
        for (int i = 0; i < N; ++i)
            for (int ki = 0; ki < static_cast<int>(adj[i].size()); ++ki) {
                if (seen[i][ki] || adj[i][ki].empty()) continue;
                Comp comp = bfs(Obs(i, ki));
                std::vector<char> has(N, 0);
                int n_img = 0;
                for (const Obs &o : comp) {
                    seen[o.first][o.second] = 1;
                    if (!has[o.first]) { has[o.first] = 1; ++n_img; }
                }
                if (static_cast<int>(comp.size()) < obser_thr || n_img < obser_thr) continue;
                std::sort(comp.begin(), comp.end());
                members.push_back(std::move(comp));
            }
    }
    struct Done { Obs start; Track track; };
    std::vector<Done> done;
    std::vector<size_t> pending(members.size());
    for (size_t c = 0; c < members.size(); ++c) pending[c] = c;
    for (size_t attempt = 0; !pending.empty(); ++attempt) {
        std::vector<Comp> orders;
        orders.reserve(pending.size());
        for (size_t c : pending) orders.push_back(bfs(members[c][attempt]));
        const FusedBatch fb = fuse(orders);
        std::vector<size_t> next;
        for (size_t n = 0; n < pending.size(); ++n) {
            const size_t c = pending[n];
            if (fb.status[n]) done.push_back(Done{members[c][attempt], make_track<Track>(orders[n], fb, n)});
            else if (attempt + 1 < members[c].size()) next.push_back(c);
        }
        pending.swap(next);
    }
    std::sort(done.begin(), done.end(), [](const Done &x, const Done &y) { return x.start < y.start; });
    if (obs_to_track) {
        obs_to_track->assign(N, std::vector<int>());
        for (int i = 0; i < N; ++i) (*obs_to_track)[i].assign(all_keypoints[i].size(), -1);
    }
    for (Done &d : done) {
        if (obs_to_track)
            for (const Obs &o : d.track.observations) (*obs_to_track)[o.first][o.second] = static_cast<int>(tracks.size());
        tracks.push_back(std::move(d.track));
    }
}
// Replaces the body of LvbaSystem::BuildTracksAndFuse3D from :923 to :1246:
//       tracks_.clear();
//       lvba::build_tracks_and_fuse<Track>(all_keypoints_, all_matches_, Rcw_all_optimized_, tcw_all_optimized_, intr, depth, &opts, tracks_);
//       tracks_before_ = tracks_;
template <class Track, class KeypointTable, class MatchTable, class RotVec, class TransVec>
void build_tracks_and_fuse(const KeypointTable &all_keypoints, const MatchTable &all_matches, const RotVec &Rcw_all,
                           const TransVec &tcw_all, const double intr[8], lvba_depth_t depth, const lvba_fuse_opts *opts,
                           std::vector<Track> &tracks, std::vector<std::vector<int>> *obs_to_track = nullptr, int device = 0)
{
    lvba_fuse_opts o;
    if (opts) o = *opts; else lvba_fuse_default_opts(&o);
    const GpuFuse<KeypointTable, RotVec, TransVec> fuse(all_keypoints, Rcw_all, tcw_all, intr, depth, &o, device);
    build_tracks_and_fuse_with<Track>(all_keypoints, all_matches, o.obser_thr, fuse, tracks, obs_to_track);
}

// ---- the ceres::Problem ... ceres::Solve region of LvbaSystem::optimizeCameraPoses (src/lvba_system.cpp:1571-1649) -------------
// qs [M] (w, x, y, z), ts [M], Xs [P]: the reference's own std::vector<std::array<double, N>> (any contiguous container of
// N-double records works), refined in place like the parameter blocks Ceres is handed; the landmarks are the ones the loop at
// :1593-1640 adds (those with a plane -- `valid` marks them, the others are left untouched with their observations, :1598-1603).
// obs_off [P+1], obs_cam [O], obs_uv [O][2]: the de-duplicated inlier observations per landmark (:1615-1632); plane_nd [P][4].
// Camera 0 is held constant (:1582-1583).  Returns the termination code (LVBA_TERM_*; LVBA_TERM_FAILURE is what the
// reference's `summary.termination_type == ceres::FAILURE` early return at :1646-1649 tests).  Usage:
//       const int term = lvba::optimize_camera_poses_hip(qs, ts, Xs, obs_off, obs_cam, obs_uv, plane_nd, valid, intr, 0.5, 0.01);
//       if (term == LVBA_TERM_FAILURE) { std::cerr << "[optimizeCamPoses] Solver failed!\n"; return; }
// and the write-back block at :1651-1665 runs unchanged afterwards.
template <class QVec, class TVec, class XVec>
int32_t optimize_camera_poses_hip(QVec &qs, TVec &ts, XVec &Xs, const std::vector<int64_t> &obs_off,
                                  const std::vector<int32_t> &obs_cam, const std::vector<double> &obs_uv,
                                  const std::vector<double> &plane_nd, const std::vector<uint8_t> &valid, const double intr[8],
                                  double sigma_px, double sigma_plane, std::vector<lvba_visual_trace> *trace = nullptr,
                                  int device = 0)
{
    const int32_t M = static_cast<int32_t>(qs.size());
    const int64_t P = static_cast<int64_t>(Xs.size());
    if (static_cast<int64_t>(obs_off.size()) != P + 1 || static_cast<int64_t>(valid.size()) != P ||
        static_cast<int64_t>(plane_nd.size()) != 4 * P || ts.size() != qs.size())
        throw std::runtime_error("optimize_camera_poses_hip: array sizes disagree");
    std::vector<double> q(4 * static_cast<size_t>(M)), t(3 * static_cast<size_t>(M)), X(3 * static_cast<size_t>(P));
    for (int32_t m = 0; m < M; ++m) {
        for (int e = 0; e < 4; ++e) q[4 * m + e] = qs[m][e];
        for (int e = 0; e < 3; ++e) t[3 * m + e] = ts[m][e];
    }
    for (int64_t a = 0; a < P; ++a)
        for (int e = 0; e < 3; ++e) X[3 * a + e] = Xs[a][e];
    lvba_visual_t vh = nullptr;
    if (lvba_visual_create(M, P, obs_off.data(), obs_cam.data(), obs_uv.data(), plane_nd.data(), valid.data(), intr, sigma_px,
                           sigma_plane, device, &vh) != LVBA_OK)
        throw std::runtime_error(std::string("lvba_visual_create: ") + lvba_last_error());
    lvba_visual_opts o;
    lvba_visual_default_opts(&o);                        // 50 iterations, Ceres 2.1 defaults (:1572-1576)
    std::vector<lvba_visual_trace> tr(static_cast<size_t>(o.max_iter) + 2);
    int32_t n_trace = 0, term = LVBA_TERM_NO_CONVERGENCE;
    const int32_t rc = lvba_visual_refine(vh, q.data(), t.data(), X.data(), &o, tr.data(), static_cast<int32_t>(tr.size()), &n_trace, &term);
    lvba_visual_destroy(vh);
    if (rc < 0) throw std::runtime_error(std::string("lvba_visual_refine: ") + lvba_last_error());
    if (trace) trace->assign(tr.begin(), tr.begin() + n_trace);
    for (int32_t m = 0; m < M; ++m) {
        for (int e = 0; e < 4; ++e) qs[m][e] = q[4 * m + e];
        for (int e = 0; e < 3; ++e) ts[m][e] = t[3 * m + e];
    }
    for (int64_t a = 0; a < P; ++a)
        if (valid[a])
            for (int e = 0; e < 3; ++e) Xs[a][e] = X[3 * a + e];
    return term;
}

} // namespace lvba
