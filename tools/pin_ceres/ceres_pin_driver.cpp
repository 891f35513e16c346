// ceres_pin_driver.cpp -- the ONE piece of the visual path that the build image cannot pin: the iterations inside ceres::Solve
// (Ceres Solver 2.1.0 is neither under /root/reference nor installed).  A maintainer whose machine has the
// reference's dependencies builds this with tools/pin_ceres/pin_ceres.sh: it assembles the ceres::Problem exactly as
// LvbaSystem::optimizeCameraPoses does (reference src/lvba_system.cpp:1571-1644: EigenQuaternionManifold on q[w,x,y,z], camera 0
// constant, landmarks without a plane left out with their observations, loss == nullptr, DENSE_SCHUR, 50 iterations) from the
// reference's OWN functors (include/utils.hpp:51-147, included from $LVBA_REFERENCE), runs the REAL ceres::Solve and prints what
// lvba_visual_refine's trace is compared with (tests/test_gpu_ceres_pin.py): per iteration cost, cost change, gradient max norm,
// step norm, trust-region radius, accepted flag -- and the refined cameras and landmarks.
//
// input: a directory written by tests/test_gpu_ceres_pin.py (raw little-endian arrays):
//   meta.txt "M T O"; q.bin [M][4] f64 (w,x,y,z); t.bin [M][3]; X.bin [T][3]; obs_off.bin [T+1] i64; obs_cam.bin [O] i32;
//   obs_uv.bin [O][2] f64; plane.bin [T][4] f64 (n, d); valid.bin [T] i32; intr.bin [8] f64 (fx fy cx cy k1 k2 p1 p2)
#include <ceres/ceres.h>
#include <ceres/rotation.h>
#include <array>
#include <cstdio>
#include <fstream>
#include <string>
#include <thread>
#include <vector>
#include "utils.hpp" // the reference's include/utils.hpp: ReprojErrorWhitenedDistorted, PointPlaneErrorWhitened

template <class T> static std::vector<T> slurp(const std::string &path, size_t n)
{
    std::vector<T> v(n);
    std::ifstream f(path, std::ios::binary);
    if (!f.read(reinterpret_cast<char *>(v.data()), (std::streamsize)(n * sizeof(T)))) { fprintf(stderr, "short read: %s\n", path.c_str()); exit(2); }
    return v;
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: ceres_pin_driver <problem dir>\n"); return 2; }
    const std::string d = std::string(argv[1]) + "/";
    long M = 0, T = 0, O = 0;
    { std::ifstream f(d + "meta.txt"); f >> M >> T >> O; }
    auto qv = slurp<double>(d + "q.bin", 4 * M), tv = slurp<double>(d + "t.bin", 3 * M), Xv = slurp<double>(d + "X.bin", 3 * T);
    auto off = slurp<int64_t>(d + "obs_off.bin", T + 1);
    auto cam = slurp<int32_t>(d + "obs_cam.bin", O);
    auto uv = slurp<double>(d + "obs_uv.bin", 2 * O), plane = slurp<double>(d + "plane.bin", 4 * T), intr = slurp<double>(d + "intr.bin", 8);
    auto valid = slurp<int32_t>(d + "valid.bin", T);
    std::vector<std::array<double, 4>> qs(M);
    std::vector<std::array<double, 3>> ts(M), Xs(T);
    for (long k = 0; k < M; ++k) { for (int e = 0; e < 4; ++e) qs[k][e] = qv[4 * k + e]; for (int e = 0; e < 3; ++e) ts[k][e] = tv[3 * k + e]; }
    for (long p = 0; p < T; ++p) for (int e = 0; e < 3; ++e) Xs[p][e] = Xv[3 * p + e];

    ceres::Problem problem;
    ceres::Solver::Options options;                       // src/lvba_system.cpp:1572-1576
    options.max_num_iterations = 50;
    options.linear_solver_type = ceres::DENSE_SCHUR;
    options.num_threads = std::max(1u, std::thread::hardware_concurrency());
    options.minimizer_progress_to_stdout = false;
    for (long k = 0; k < M; ++k) {                        // :1578-1581
        problem.AddParameterBlock(qs[k].data(), 4, new ceres::EigenQuaternionManifold());
        problem.AddParameterBlock(ts[k].data(), 3);
    }
    problem.SetParameterBlockConstant(qs[0].data());      // :1582-1583
    problem.SetParameterBlockConstant(ts[0].data());
    const double sigma_px = 0.5, sigma_plane = 0.01;      // :1590-1591
    for (long p = 0; p < T; ++p) {
        if (!valid[p]) continue;                          // :1598-1603: no plane, no landmark, no reprojection residuals
        problem.AddParameterBlock(Xs[p].data(), 3);
        for (int64_t o = off[p]; o < off[p + 1]; ++o) {   // :1611-1631 (the caller hands over de-duplicated inlier observations)
            ceres::CostFunction *cost = ReprojErrorWhitenedDistorted::Create(uv[2 * o], uv[2 * o + 1], intr[0], intr[1], intr[2], intr[3],
                                                                             intr[4], intr[5], intr[6], intr[7], sigma_px, sigma_px);
            problem.AddResidualBlock(cost, nullptr, qs[cam[o]].data(), ts[cam[o]].data(), Xs[p].data());
        }
        const Eigen::Vector3d n(plane[4 * p], plane[4 * p + 1], plane[4 * p + 2]);
        problem.AddResidualBlock(PointPlaneErrorWhitened::Create(n, plane[4 * p + 3], sigma_plane), nullptr, Xs[p].data()); // :1638-1639
    }
    ceres::Solver::Summary summary;
    ceres::Solve(options, &problem, &summary);
    printf("{\"ceres_version\": \"%s\", \"termination\": %d, \"iterations\": [", CERES_VERSION_STRING, (int)summary.termination_type);
    for (size_t i = 0; i < summary.iterations.size(); ++i) {
        const auto &it = summary.iterations[i];
        printf("%s{\"iter\": %d, \"cost\": %.17g, \"cost_change\": %.17g, \"gradient_max_norm\": %.17g, \"step_norm\": %.17g, "
               "\"radius\": %.17g, \"accepted\": %d, \"valid\": %d}",
               i ? ", " : "", it.iteration, it.cost, it.cost_change, it.gradient_max_norm, it.step_norm, it.trust_region_radius,
               (int)it.step_is_successful, (int)it.step_is_valid);
    }
    printf("], \"q\": [");
    for (long k = 0; k < M; ++k) printf("%s[%.17g, %.17g, %.17g, %.17g]", k ? ", " : "", qs[k][0], qs[k][1], qs[k][2], qs[k][3]);
    printf("], \"t\": [");
    for (long k = 0; k < M; ++k) printf("%s[%.17g, %.17g, %.17g]", k ? ", " : "", ts[k][0], ts[k][1], ts[k][2]);
    printf("], \"X\": [");
    for (long p = 0; p < T; ++p) printf("%s[%.17g, %.17g, %.17g]", p ? ", " : "", Xs[p][0], Xs[p][1], Xs[p][2]);
    printf("]}\n");
    return 0;
}
