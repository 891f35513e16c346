// Compile/link check of include/lvba_adapter.hpp against stand-ins for the reference's Eigen-based types.
// Returns 0 if packing is right and (without a GPU) the library refuses loudly, or (with one) refines.
#include <cmath>
#include <cstdio>
#include <array>
#include <cstring>
#include <utility>
#include <vector>
#include "../include/lvba_adapter.hpp"

struct Mat3 { double m[9]; double &operator()(int r, int c) { return m[3 * r + c]; } double operator()(int r, int c) const { return m[3 * r + c]; } };
struct PointCluster { Mat3 P; double v[3]; int N; };
struct IMUST { Mat3 R; double p[3]; };
struct PointXYZINormal { float x, y, z, pad0, nx, ny, nz, pad1, intensity, curvature, pad2, pad3; };   // 48 B, as PCL
struct Cloud { std::vector<PointXYZINormal> points; };
struct VOX_HESS { int win_size; std::vector<const std::vector<PointCluster> *> plvec_voxels; };

int main()
{
    const int N = 3;
    std::vector<PointCluster> a(N), b(N);
    for (auto *vv : {&a, &b})
        for (auto &c : *vv) { std::memset(&c, 0, sizeof c); }
    const double pts[4][3] = {{1, 0, 0.01}, {0, 1, -0.01}, {-1, 0, 0.0}, {0, -1, 0.02}};
    auto push = [&](PointCluster &c, double ox) {
        for (auto &p : pts) {
            double q[3] = {p[0] + ox, p[1], p[2]};
            for (int r = 0; r < 3; ++r) { c.v[r] += q[r]; for (int s = 0; s < 3; ++s) c.P(r, s) += q[r] * q[s]; }
            c.N++;
        }
    };
    push(a[0], 0.0); push(a[2], 0.1); push(b[1], 0.0); push(b[2], 0.2);
    VOX_HESS vh{N, {&a, &b}};
    std::vector<int64_t> off; std::vector<int32_t> idx; std::vector<double> clu;
    lvba::pack_voxhess(vh, off, idx, clu);
    if (off != std::vector<int64_t>{0, 2, 4} || idx != std::vector<int32_t>{0, 2, 1, 2} || clu.size() != 40 || clu[9] != 4) return 1;
    std::vector<IMUST> x(N);
    for (auto &s : x) { std::memset(&s, 0, sizeof s); s.R(0, 0) = s.R(1, 1) = s.R(2, 2) = 1; }
    // front-end adapter: two frames looking at one plane patch inside a single root voxel
    static_assert(sizeof(PointXYZINormal) == 48, "stand-in must have PCL's stride");
    Cloud c0, c1;
    for (int i = 0; i < 40; ++i) {
        PointXYZINormal p{};
        p.x = 0.1f + 0.02f * (i % 8); p.y = 0.1f + 0.1f * (i / 8); p.z = 0.5f + 0.001f * ((i * 7) % 3);
        p.intensity = 100.f;
        (i % 2 ? c1 : c0).points.push_back(p);
    }
    std::vector<const Cloud *> clouds{&c0, &c1};
    std::vector<IMUST> xw(2);
    for (auto &s : xw) { std::memset(&s, 0, sizeof s); s.R(0, 0) = s.R(1, 1) = s.R(2, 2) = 1; }
    const float ratio[4] = {0.3f, 0.1f, 0.06f, 0.03f};
    try {
        lvba::VoxelMap surf_map(clouds, xw, 1.0, ratio);
        const auto info = surf_map.info();
        std::printf("voxel map on the GPU: %lld points, %lld roots, %lld voxels\n", (long long)info.n_points,
                    (long long)info.n_roots, (long long)info.n_voxels);
        if (info.n_points != 40 || info.n_roots != 1 || info.n_voxels != 1 || info.n_factors != 2) return 3;
        std::vector<std::vector<double>> Xs{{0.2, 0.3, 0.9}, {5.0, 5.0, 5.0}};
        std::vector<double> nd; std::vector<uint8_t> ok;
        surf_map.find_planes(Xs, nd, ok);
        if (!ok[0] || ok[1] || std::fabs(std::fabs(nd[2]) - 1.0) > 1e-3 || std::fabs(std::fabs(nd[3]) - 0.5) > 5e-3) return 4;
    } catch (const std::exception &e) {
        std::printf("refused: %s\n", e.what());
        if (!(lvba_device_count() == 0 && std::strstr(e.what(), "no CPU fallback"))) return 5;
    }
    try { // whole LiDAR stage through the adapter: two windows of one frame pair each
        std::vector<const Cloud *> four{&c0, &c1, &c0, &c1};
        std::vector<IMUST> x4(4);
        for (auto &s : x4) { std::memset(&s, 0, sizeof s); s.R(0, 0) = s.R(1, 1) = s.R(2, 2) = 1; }
        lvba_lidar_ba_opts lo;
        lvba_lidar_ba_default_opts(&lo);
        lo.window.window_size = 2;
        lo.window.voxel.voxel_size = 1.0; lo.stage_voxel_size[0] = lo.stage_voxel_size[1] = 1.0;
        const auto rep = lvba::lidar_ba(four, x4, lo);
        std::printf("lidar_ba on the GPU: %d windows, %d skipped, %d anchors\n", rep.n_windows, rep.n_windows_skipped, rep.n_anchors);
        if (rep.n_windows != 2 || rep.n_frames != 4) return 6;
    } catch (const std::exception &e) {
        std::printf("refused: %s\n", e.what());
        if (!(lvba_device_count() == 0 && std::strstr(e.what(), "no CPU fallback"))) return 7;
    }
    try { // the same over a device list (two shares; on a one-GPU box both on device 0): the windows dealt out, anchors gathered
        std::vector<const Cloud *> four{&c0, &c1, &c0, &c1};
        std::vector<IMUST> x4(4);
        for (auto &s : x4) { std::memset(&s, 0, sizeof s); s.R(0, 0) = s.R(1, 1) = s.R(2, 2) = 1; }
        lvba_lidar_ba_opts lo;
        lvba_lidar_ba_default_opts(&lo);
        lo.window.window_size = 2;
        lo.window.voxel.voxel_size = 1.0; lo.stage_voxel_size[0] = lo.stage_voxel_size[1] = 1.0;
        const auto rep = lvba::lidar_ba(four, x4, lo, std::vector<int>{0, 0});
        std::printf("lidar_ba over two shares: %d windows, %d skipped, %d anchors\n", rep.n_windows, rep.n_windows_skipped, rep.n_anchors);
        if (rep.n_windows != 2 || rep.n_frames != 4) return 16;
    } catch (const std::exception &e) {
        std::printf("refused: %s\n", e.what());
        if (!(lvba_device_count() == 0 && std::strstr(e.what(), "no CPU fallback"))) return 17;
    }
    try { // track fusion through the adapter: one landmark seen by five cameras on a line (triangulation candidate)
        struct Vec3 { double v[3]; double &operator[](int i) { return v[i]; } double operator[](int i) const { return v[i]; } };
        struct Kp { float x, y; };
        struct Track { Vec3 Xw_fused{}; std::vector<std::pair<int, int>> observations; std::vector<int> inlier_indices; };
        const double intr[8] = {120, 120, 80, 60, 0, 0, 0, 0};
        const double Xw[3] = {0.3, -0.2, 6.0};
        std::vector<Mat3> Rcw(5); std::vector<Vec3> tcw(5);
        std::vector<std::vector<Kp>> kps(5);
        std::vector<std::pair<int, int>> comp;
        for (int m = 0; m < 5; ++m) {
            std::memset(&Rcw[m], 0, sizeof(Mat3)); Rcw[m](0, 0) = Rcw[m](1, 1) = Rcw[m](2, 2) = 1;
            tcw[m] = Vec3{{-0.5 * m, 0.0, 0.0}};                                 // camera centre at x = 0.5 m
            const double xc = Xw[0] + tcw[m][0], yc = Xw[1], zc = Xw[2];
            kps[m].push_back(Kp{(float)(intr[0] * xc / zc + intr[2]), (float)(intr[1] * yc / zc + intr[3])});
            comp.push_back({m, 0});
        }
        std::vector<std::vector<std::pair<int, int>>> comps{comp, {{0, 0}, {1, 0}}};  // the second one is too short
        std::vector<Track> tracks; std::vector<int> comp_track;
        lvba::fuse_components<Track>(comps, kps, Rcw, tcw, intr, nullptr, nullptr, tracks, comp_track);
        std::printf("fuse_components on the GPU: %zu track(s)\n", tracks.size());
        if (tracks.size() != 1 || comp_track[0] != 0 || comp_track[1] != -1 || tracks[0].observations.size() != 5 ||
            tracks[0].inlier_indices.size() < 4 || std::fabs(tracks[0].Xw_fused[2] - 6.0) > 1e-3)
            return 8;
    } catch (const std::exception &e) {
        std::printf("refused: %s\n", e.what());
        if (!(lvba_device_count() == 0 && std::strstr(e.what(), "no CPU fallback"))) return 9;
    }
    try { // the visual solve through the adapter (src/lvba_system.cpp:1571-1649): 4 cameras on a line, 40 landmarks on z-planes
        std::vector<std::array<double, 4>> qs(4, std::array<double, 4>{1.0, 0.0, 0.0, 0.0});
        std::vector<std::array<double, 3>> ts(4), Xs(40);
        const double intr[8] = {300, 300, 160, 120, 0, 0, 0, 0};
        std::vector<int64_t> obs_off{0};
        std::vector<int32_t> obs_cam;
        std::vector<double> obs_uv, plane;
        std::vector<uint8_t> valid;
        unsigned rs = 12345u;
        auto rnd = [&]() { rs = rs * 1664525u + 1013904223u; return (double)(rs >> 8) / (double)(1u << 24) - 0.5; };
        for (int m = 0; m < 4; ++m) ts[m] = {-0.4 * m, 0.0, 0.0};
        for (int p = 0; p < 40; ++p) {
            const double X[3] = {2.0 * rnd(), 1.5 * rnd(), 6.0 + 2.0 * rnd()};
            for (int m = 0; m < 4; ++m) {
                obs_cam.push_back(m);
                obs_uv.push_back(intr[0] * (X[0] + ts[m][0]) / X[2] + intr[2] + 0.3 * rnd());
                obs_uv.push_back(intr[1] * X[1] / X[2] + intr[3] + 0.3 * rnd());
            }
            obs_off.push_back((int64_t)obs_cam.size());
            plane.insert(plane.end(), {0.0, 0.0, 1.0, -X[2]});
            valid.push_back(p % 7 != 0);                                   // a few landmarks without a plane: left untouched
            Xs[p] = {X[0] + 0.05 * rnd(), X[1] + 0.05 * rnd(), X[2] + 0.05 * rnd()};
        }
        for (int m = 1; m < 4; ++m) ts[m][1] += 0.01 * m;                  // perturbed cameras (camera 0 is held constant)
        const std::array<double, 3> X0 = Xs[0], X1 = Xs[1];
        std::vector<lvba_visual_trace> trace;
        const int32_t term = lvba::optimize_camera_poses_hip(qs, ts, Xs, obs_off, obs_cam, obs_uv, plane, valid, intr, 0.5, 0.01, &trace);
        std::printf("cameras refined on the GPU: %zu rows, cost %.4e -> %.4e, termination %d\n", trace.size(),
                    trace.empty() ? 0.0 : trace.front().cost, trace.empty() ? 0.0 : trace.back().cost, (int)term);
        if (trace.size() < 2 || !(trace.back().cost < 0.2 * trace.front().cost) || term == LVBA_TERM_FAILURE) return 10;
        if (Xs[0] != X0 || Xs[1] == X1) return 11;                        // landmark 0 has no plane; landmark 1 moved
    } catch (const std::exception &e) {
        std::printf("refused: %s\n", e.what());
        if (!(lvba_device_count() == 0 && std::strstr(e.what(), "no CPU fallback"))) return 12;
    }
    try {
        auto trace = lvba::damping_iter_hip(x, vh);
        std::printf("refined on the GPU: %zu LM iterations\n", trace.size());
    } catch (const std::exception &e) {
        std::printf("refused: %s\n", e.what());
        return lvba_device_count() == 0 && std::strstr(e.what(), "no CPU fallback") ? 0 : 2;
    }
    return 0;
}
