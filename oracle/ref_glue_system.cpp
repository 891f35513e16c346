// ref_glue_system.cpp -- TEST INFRASTRUCTURE ONLY: C entry points around the REFERENCE'S OWN src/lvba_system.cpp and
// src/dataset_io.cpp, compiled as they lie under /root/reference (both files are #included below, nothing is copied) against
// the stand-ins of oracle/shim:
//   ROS            ros/ros.h: NodeHandle::param() served from a table the test fills, publishers that drop their messages
//   OpenCV         a real cv::Mat (depth images); codecs / drawing throw (never reached), previews are no-ops
//   PCL            PointCloud = vector of points, a PCD reader for the encodings the tests write
//   Eigen, Sophus  lvba_eigen_standin.h, sophus/se3.h
//   SiftGPU, GL    every call throws: the SIFT front end is outside the scope contract
//   sqlite3        declarations only; the system's libsqlite3.so.0 is linked, so loadFromColmapDB runs for real
//   Ceres          cost functors differentiated with Jets exactly as AutoDiffCostFunction would, and a ceres::Problem that
//                  RECORDS what optimizeCameraPoses adds to it; ceres::Solve hands the recorded problem to the hook below,
//                  which evaluates it at the initial point and (optionally) installs a solution computed elsewhere.
//                  There is no Ceres solver: the trust-region iterations themselves stay UNPINNED.
// Third translation unit of oracle/_ref/libbalm_ref.so.  tests/test_ref_system.py drives it to pin, against the reference's
// own statements: global-lvba_amd/dataset.py (DatasetIO), oracle/window_oracle.py (runWindowBA / runLidarBA),
// global-lvba_amd/pipeline.py (updateCameraPosesFromLidar, camera_from_imu, build_components, the visual problem set-up),
// oracle/fusion_oracle.py (buildGridMapFromOptimized + generateDepthWithVoxel, BuildTracksAndFuse3D) and
// oracle/track_oracle.py (ComputeMeanReproj / TriangulateTrackDLT).
#include <cstring>
#include <deque>
#include <fstream>
#include <set>
#include <unordered_set>
#ifndef ROOT_DIR
#define ROOT_DIR "" // CMakeLists.txt of the reference defines it as the source directory; the tests pass absolute paths
#endif
#ifdef LVBA_DROPIN
// The DROP-IN build (oracle/_ref/liblvba_system_dropin.so, linked against the product's liblvba_hip.so): the reference's own
// pipeline code with the two hot-path calls going to the GPU library through include/lvba_adapter.hpp -- what a maintainer
// gets after the two-line patch of INTEGRATION.md section 2.  The reference's sources are compiled from where they lie and
// cannot be edited here, so the patch is made by the preprocessor: src/lvba_system.cpp:264 and :386 read
//     opt_lsv->damping_iter(x, *voxhess);
// which becomes   opt_lsv->win_size, lvba_dropin::damping_iter(x, *voxhess);   (a member read, then the adapter call).
// BALM2's own definition is seen first, untouched.  The ceres::Solve of optimizeCameraPoses (:1643) reaches the hook below,
// which hands the problem the reference assembled to lvba::optimize_camera_poses_hip.
#include "BALM/bavoxel.hpp"
#include "../include/lvba_adapter.hpp"
namespace lvba_dropin {
static int n_lidar_calls = 0, n_lidar_iterations = 0, n_visual_calls = 0, visual_termination = -1, visual_iterations = 0;
static double visual_cost0 = 0.0, visual_cost1 = 0.0, lidar_call_diff[16] = {0};
template <class PoseVec, class VoxHess> void damping_iter(PoseVec &x, const VoxHess &vh)
{
    // LVBA_DROPIN_CHECK: the reference's own BALM2::damping_iter (CPU) on a copy of the same inputs, call by call
    PoseVec x_cpu = x;
    const bool check = std::getenv("LVBA_DROPIN_CHECK") != nullptr;
    if (check) { BALM2 ref(vh.win_size); ref.damping_iter(x_cpu, const_cast<VoxHess &>(vh)); }
    const std::vector<lvba_lm_trace> tr = lvba::damping_iter_hip(x, vh);
    if (check && n_lidar_calls < 16) {
        double d = 0.0;
        for (size_t j = 0; j < x.size(); ++j) {
            for (int r = 0; r < 3; ++r) {
                d = std::max(d, std::fabs(x[j].p[r] - x_cpu[j].p[r]));
                for (int c = 0; c < 3; ++c) d = std::max(d, std::fabs(x[j].R(r, c) - x_cpu[j].R(r, c)));
            }
        }
        lidar_call_diff[n_lidar_calls] = d;
        std::fprintf(stderr, "[dropin] damping_iter call %d: %d voxels, %d poses, %d LM iterations, cost %.6e -> %.6e, max |GPU - CPU| = %.3e\n",
                     n_lidar_calls, (int)vh.plvec_voxels.size(), (int)vh.win_size, (int)tr.size(), tr.empty() ? 0.0 : tr.front().residual1,
                     tr.empty() ? 0.0 : (tr.back().accepted ? tr.back().residual2 : tr.back().residual1), d);
    }
    ++n_lidar_calls;
    n_lidar_iterations += (int)tr.size();
}
} // namespace lvba_dropin
#define damping_iter(X, V) win_size, lvba_dropin::damping_iter(X, V)
#endif
#include "dataset_io.cpp"
#include "lvba_system.cpp"
#ifdef LVBA_DROPIN
#undef damping_iter
#endif

namespace {

struct RecordedResidual { int kind; int cam; int point; double r[2]; double loss_a; double uv[2]; };
struct RecordedProblem {
    bool valid = false;
    int n_cams = 0, n_points = 0, max_iter = 0, linear_solver = -1;
    std::vector<double *> cam_q, cam_t, pts;       // parameter block addresses in order of first appearance
    std::vector<int> q_const, t_const, q_manifold; // per camera
    std::vector<double> q0, t0, X0;                // values when Solve was called
    std::vector<RecordedResidual> res;
    std::vector<double> plane;                     // per point: n[3], d  (read back from the functor)
    double cost0 = 0;                              // 0.5 sum rho(|r|^2), Huber where a loss was given
    std::vector<double> sol_q, sol_t, sol_X;       // optional solution to install (same order as q0 / t0 / X0)
};
RecordedProblem g_rec;

int index_of(const std::vector<double *> &v, const double *p)
{
    for (size_t i = 0; i < v.size(); ++i) if (v[i] == p) return (int)i;
    return -1;
}

void solve_hook(const ceres::Solver::Options &opt, ceres::Problem *prob, ceres::Solver::Summary *sum)
{
    RecordedProblem &R = g_rec;
    std::vector<double> sq = R.sol_q, st = R.sol_t, sX = R.sol_X;
    R = RecordedProblem();
    R.valid = true;
    R.max_iter = opt.max_num_iterations;
    R.linear_solver = (int)opt.linear_solver_type;
    // parameter blocks: the reference adds (q, t) per camera first, then one block per landmark it keeps
    for (size_t i = 0; i < prob->parameter_blocks.size(); ++i) {
        const auto &pb = prob->parameter_blocks[i];
        if (pb.size == 4) { R.cam_q.push_back(pb.values); R.q_const.push_back(pb.constant); R.q_manifold.push_back(pb.manifold ? pb.manifold->TangentSize() : 0); }
    }
    for (size_t i = 0; i < prob->parameter_blocks.size(); ++i) {
        const auto &pb = prob->parameter_blocks[i];
        if (pb.size != 3) continue;
        // a 3-block directly following a 4-block in the list is that camera's translation
        if (i > 0 && prob->parameter_blocks[i - 1].size == 4) { R.cam_t.push_back(pb.values); R.t_const.push_back(pb.constant); }
        else R.pts.push_back(pb.values);
    }
    R.n_cams = (int)R.cam_q.size();
    R.n_points = (int)R.pts.size();
    for (double *q : R.cam_q) R.q0.insert(R.q0.end(), q, q + 4);
    for (double *t : R.cam_t) R.t0.insert(R.t0.end(), t, t + 3);
    for (double *X : R.pts) R.X0.insert(R.X0.end(), X, X + 3);
    R.plane.assign((size_t)4 * R.n_points, 0.0);
    for (const auto &rb : prob->residual_blocks) {
        RecordedResidual rr{};
        rr.loss_a = rb.loss ? rb.loss->scale() : 0.0;
        rr.r[0] = rr.r[1] = 0.0;
        if (rb.parameters.size() == 3) {
            rr.kind = 2;
            if (auto *rc = dynamic_cast<const ceres::AutoDiffCostFunction<lvba::ReprojErrorWhitenedDistorted, 2, 4, 3, 3> *>(rb.cost)) {
                rr.uv[0] = rc->functor_->u_; rr.uv[1] = rc->functor_->v_; // the observation the functor was created with (:1624-1628)
            }
            rr.cam = index_of(R.cam_q, rb.parameters[0]);
            if (index_of(R.cam_t, rb.parameters[1]) != rr.cam) rr.cam = -1;
            rr.point = index_of(R.pts, rb.parameters[2]);
        } else {
            rr.kind = 1;
            rr.cam = -1;
            rr.point = index_of(R.pts, rb.parameters[0]);
            // r = (n.X + d) / sigma  ->  n / sigma = gradient, d / sigma = r(0)
            const double zero[3] = {0, 0, 0};
            const double *pz[1] = {zero};
            double r0, J[3];
            double *Jp[1] = {J};
            rb.cost->Evaluate(pz, &r0, Jp);
            const double inv_sigma = std::sqrt(J[0] * J[0] + J[1] * J[1] + J[2] * J[2]); // |n| = 1
            if (rr.point >= 0)
                for (int k = 0; k < 4; ++k) R.plane[4 * rr.point + k] = (k < 3 ? J[k] : r0) / inv_sigma;
        }
        std::vector<const double *> pp(rb.parameters.begin(), rb.parameters.end());
        rb.cost->Evaluate(pp.data(), rr.r, nullptr);
        const double s = rr.r[0] * rr.r[0] + rr.r[1] * rr.r[1];
        if (rb.loss && s > rr.loss_a * rr.loss_a) R.cost0 += 0.5 * (2.0 * rr.loss_a * std::sqrt(s) - rr.loss_a * rr.loss_a);
        else R.cost0 += 0.5 * s;
        R.res.push_back(rr);
    }
    sum->termination_type = ceres::NO_CONVERGENCE;
    if ((int)sq.size() == 4 * R.n_cams && (int)st.size() == 3 * R.n_cams && (int)sX.size() == 3 * R.n_points) {
        for (int k = 0; k < R.n_cams; ++k) {
            std::copy(sq.begin() + 4 * k, sq.begin() + 4 * k + 4, R.cam_q[k]);
            std::copy(st.begin() + 3 * k, st.begin() + 3 * k + 3, R.cam_t[k]);
        }
        for (int p = 0; p < R.n_points; ++p) std::copy(sX.begin() + 3 * p, sX.begin() + 3 * p + 3, R.pts[p]);
        sum->termination_type = ceres::CONVERGENCE;
    }
}

#ifdef LVBA_DROPIN
// ceres::Solve of the drop-in build: the recorded problem -> the arrays of lvba_visual_create (in the order the reference
// added the blocks), solved on the GPU, written back into the parameter blocks as Ceres would.
void dropin_solve_hook(const ceres::Solver::Options &opt, ceres::Problem *prob, ceres::Solver::Summary *sum)
{
    typedef ceres::AutoDiffCostFunction<lvba::ReprojErrorWhitenedDistorted, 2, 4, 3, 3> ReprojCost;
    typedef ceres::AutoDiffCostFunction<lvba::PointPlaneErrorWhitened, 1, 3> PlaneCost;
    std::vector<double *> cam_q, cam_t, pts;
    for (size_t i = 0; i < prob->parameter_blocks.size(); ++i) {
        const auto &pb = prob->parameter_blocks[i];
        if (pb.size == 4) cam_q.push_back(pb.values);
        else if (i > 0 && prob->parameter_blocks[i - 1].size == 4) cam_t.push_back(pb.values);
        else pts.push_back(pb.values);
    }
    const int M = (int)cam_q.size(), P = (int)pts.size();
    std::vector<std::vector<std::pair<int, std::pair<double, double>>>> obs((size_t)P);
    std::vector<double> plane(4 * (size_t)P, 0.0);
    std::vector<uint8_t> valid((size_t)P, 0);
    double intr[8] = {0}, sig_px = 0.5, sig_pl = 0.01;
    for (const auto &rb : prob->residual_blocks) {
        if (rb.loss) throw std::runtime_error("drop-in: a loss function was attached (the reference passes nullptr, :1630,1639)");
        if (const ReprojCost *rc = dynamic_cast<const ReprojCost *>(rb.cost)) {
            const auto &f = *rc->functor_;
            const int cam = index_of(cam_q, rb.parameters[0]), pt = index_of(pts, rb.parameters[2]);
            if (cam < 0 || pt < 0 || index_of(cam_t, rb.parameters[1]) != cam) throw std::runtime_error("drop-in: unexpected block wiring");
            obs[(size_t)pt].push_back({cam, {f.u_, f.v_}});
            const double in[8] = {f.fx_, f.fy_, f.cx_, f.cy_, f.k1_, f.k2_, f.p1_, f.p2_};
            std::copy(in, in + 8, intr);
            sig_px = f.su_;
        } else if (const PlaneCost *pc = dynamic_cast<const PlaneCost *>(rb.cost)) {
            const auto &f = *pc->functor_;
            const int pt = index_of(pts, rb.parameters[0]);
            if (pt < 0) throw std::runtime_error("drop-in: plane residual on an unknown block");
            plane[4 * (size_t)pt] = f.nx_; plane[4 * (size_t)pt + 1] = f.ny_; plane[4 * (size_t)pt + 2] = f.nz_; plane[4 * (size_t)pt + 3] = f.d_;
            valid[(size_t)pt] = 1;
            sig_pl = f.s_;
        } else
            throw std::runtime_error("drop-in: unknown cost function");
    }
    std::vector<int64_t> off(1, 0);
    std::vector<int32_t> ocam;
    std::vector<double> ouv;
    for (int p = 0; p < P; ++p) {
        for (const auto &o : obs[(size_t)p]) { ocam.push_back(o.first); ouv.push_back(o.second.first); ouv.push_back(o.second.second); }
        off.push_back((int64_t)ocam.size());
    }
    std::vector<std::array<double, 4>> qs((size_t)M);
    std::vector<std::array<double, 3>> ts((size_t)M), Xs((size_t)P);
    for (int m = 0; m < M; ++m) { std::copy(cam_q[m], cam_q[m] + 4, qs[m].begin()); std::copy(cam_t[m], cam_t[m] + 3, ts[m].begin()); }
    for (int p = 0; p < P; ++p) std::copy(pts[p], pts[p] + 3, Xs[p].begin());
    std::vector<lvba_visual_trace> trace;
    const int term = lvba::optimize_camera_poses_hip(qs, ts, Xs, off, ocam, ouv, plane, valid, intr, sig_px, sig_pl, &trace);
    for (int m = 0; m < M; ++m) { std::copy(qs[m].begin(), qs[m].end(), cam_q[m]); std::copy(ts[m].begin(), ts[m].end(), cam_t[m]); }
    for (int p = 0; p < P; ++p) std::copy(Xs[p].begin(), Xs[p].end(), pts[p]);
    ++lvba_dropin::n_visual_calls;
    lvba_dropin::visual_termination = term;
    lvba_dropin::visual_iterations = trace.empty() ? 0 : (int)trace.size() - 1;
    lvba_dropin::visual_cost0 = trace.empty() ? 0.0 : trace.front().cost;
    lvba_dropin::visual_cost1 = trace.empty() ? 0.0 : trace.back().cost;
    sum->termination_type = term == LVBA_TERM_FAILURE ? ceres::FAILURE : ceres::CONVERGENCE;
    (void)opt;
}
#endif

ros::NodeHandle g_nh;

template <class F> int guarded(F f)
{
    int rc = 0;
    try { f(); }
    catch (const std::exception &e) { std::fprintf(stderr, "[ref_sys] %s\n", e.what()); rc = -1; }
    std::cout.flush(); std::fflush(stdout); // the reference narrates on stdout: hand it to the test runner's capture now, not at exit
    return rc;
}
void put_R(const Eigen::Matrix3d &R, double *o) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) o[3 * i + j] = R(i, j); }
void put_v(const Eigen::Vector3d &v, double *o) { o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; }

} // namespace

extern "C" {

void ref_sys_clear_params() { ros::lvba_param_table().clear(); }
void ref_sys_set_param(const char *key, const char *value) { ros::lvba_param_table()[key] = value; }

// LvbaSystem::LvbaSystem (:113-134): reads the parameters, DatasetIO loads the dataset directory named by data_config/data_path
void *ref_sys_create()
{
    lvba::LvbaSystem *s = nullptr;
    if (guarded([&] { s = new lvba::LvbaSystem(g_nh); }) != 0) return nullptr;
    return s;
}
void ref_sys_destroy(void *h) { delete static_cast<lvba::LvbaSystem *>(h); }
#define SYS(h) (*static_cast<lvba::LvbaSystem *>(h))

// ---- DatasetIO (src/dataset_io.cpp) -------------------------------------------------------------------------------------
int ref_sys_n_scans(void *h) { return (int)SYS(h).dataset_io_->x_buf_.size(); }
int ref_sys_n_clouds(void *h) { return (int)SYS(h).dataset_io_->pl_fulls_.size(); }
int ref_sys_n_images(void *h) { return (int)SYS(h).dataset_io_->images_ids_.size(); }
// which: 0 = x_buf_ (current), 1 = x_buf_before_.  R [n][9] row-major, p [n][3], t [n]
void ref_sys_scan_poses(void *h, int which, double *R, double *p, double *t)
{
    const std::vector<IMUST> &x = which ? SYS(h).dataset_io_->x_buf_before_ : SYS(h).dataset_io_->x_buf_;
    for (size_t i = 0; i < x.size(); ++i) { put_R(x[i].R, R + 9 * i); put_v(x[i].p, p + 3 * i); t[i] = x[i].t; }
}
void ref_sys_set_scan_poses(void *h, const double *R, const double *p)
{
    std::vector<IMUST> &x = SYS(h).dataset_io_->x_buf_;
    for (size_t i = 0; i < x.size(); ++i) {
        for (int a = 0; a < 3; ++a) { for (int b = 0; b < 3; ++b) x[i].R(a, b) = R[9 * i + 3 * a + b]; x[i].p[a] = p[3 * i + a]; }
    }
}
int ref_sys_cloud_size(void *h, int i) { return (int)SYS(h).dataset_io_->pl_fulls_[i]->size(); }
void ref_sys_cloud(void *h, int i, float *xyzi)
{
    const auto &pl = *SYS(h).dataset_io_->pl_fulls_[i];
    for (size_t k = 0; k < pl.size(); ++k) { xyzi[4 * k] = pl[k].x; xyzi[4 * k + 1] = pl[k].y; xyzi[4 * k + 2] = pl[k].z; xyzi[4 * k + 3] = pl[k].intensity; }
}
void ref_sys_image_ids(void *h, double *ids) { const auto &v = SYS(h).dataset_io_->images_ids_; std::copy(v.begin(), v.end(), ids); }
// which: 0 = DatasetIO::image_poses_ (as loaded), 1 = LvbaSystem::poses_ (after updateCameraPosesFromLidar)
void ref_sys_image_poses(void *h, int which, double *R, double *t)
{
    const std::vector<Sophus::SE3> &v = which ? SYS(h).poses_ : SYS(h).dataset_io_->image_poses_;
    for (size_t i = 0; i < v.size(); ++i) { put_R(v[i].rotation_matrix(), R + 9 * i); put_v(v[i].translation(), t + 3 * i); }
}
// out: width height fx fy cx cy k1 k2 p1 p2 scale  (after the resize scaling of readParameters)
void ref_sys_camera(void *h, double *out)
{
    const lvba::DatasetIO &d = *SYS(h).dataset_io_;
    const double v[11] = {(double)d.width_, (double)d.height_, d.fx_, d.fy_, d.cx_, d.cy_, d.k1_, d.k2_, d.p1_, d.p2_, d.resize_scale_};
    std::copy(v, v + 11, out);
}

// ---- LiDAR stage -------------------------------------------------------------------------------------------------------
int ref_sys_init(void *h) { return guarded([&] { SYS(h).initFromDatasetIO(); }); } // :448-505
// Rci [9], tci [3]: the IMU -> camera extrinsics initFromDatasetIO derives
void ref_sys_extrinsics(void *h, double *Rci, double *tci) { put_R(SYS(h).Rci_, Rci); put_v(SYS(h).tci_, tci); }
// runLidarBA (:312-410).  The reference waits for a '1' on stdin after its preview; the glue answers it.
int ref_sys_run_lidar_ba(void *h)
{
    std::istringstream yes("1\n");
    std::streambuf *old = std::cin.rdbuf(yes.rdbuf());
    const int rc = guarded([&] { SYS(h).runLidarBA(); });
    std::cin.rdbuf(old);
    return rc;
}
void ref_sys_anchor_index(void *h, int *idx) { const auto &v = SYS(h).anchor_index_per_frame_; std::copy(v.begin(), v.end(), idx); }
void ref_sys_rel_poses(void *h, double *R, double *p)
{
    const auto &v = SYS(h).rel_poses_to_anchor_;
    for (size_t i = 0; i < v.size(); ++i) { put_R(v[i].R, R + 9 * i); put_v(v[i].p, p + 3 * i); }
}

// ---- visual stage, step by step (runVisualBAWithLidarAssist :144-154) -------------------------------------------------------
int ref_sys_build_grid_map(void *h) { return guarded([&] { SYS(h).buildGridMapFromOptimized(); }); }      // :1266-1338
int ref_sys_update_camera_poses(void *h) { return guarded([&] { SYS(h).updateCameraPosesFromLidar(); }); } // :412-446
int ref_sys_generate_depth(void *h) { return guarded([&] { SYS(h).generateDepthWithVoxel(); }); }         // :835-919
int64_t ref_sys_grid_points(void *h) { int64_t n = 0; for (const auto &kv : SYS(h).grid_map_) n += (int64_t)kv.second.size(); return n; }
int ref_sys_n_voxel_ids(void *h, int img) { return (int)SYS(h).all_voxel_ids_[img].size(); }
void ref_sys_depth(void *h, int img, float *out)
{
    const cv::Mat &d = SYS(h).all_depths_[img];
    std::memcpy(out, d.data, sizeof(float) * (size_t)d.rows * d.cols);
}
// which: 0 = Rcw_all_ (from the poses as loaded), 1 = Rcw_all_optimized_
void ref_sys_cam_poses(void *h, int which, double *R, double *t)
{
    const auto &Rs = which ? SYS(h).Rcw_all_optimized_ : SYS(h).Rcw_all_;
    const auto &ts = which ? SYS(h).tcw_all_optimized_ : SYS(h).tcw_all_;
    for (size_t i = 0; i < Rs.size(); ++i) { put_R(Rs[i], R + 9 * i); put_v(ts[i], t + 3 * i); }
}
// the SIFT front end is replaced by the test's own key points and matches (all_keypoints_, all_matches_ in pairIndex order)
void ref_sys_set_keypoints(void *h, int img, int n, const float *xy)
{
    auto &all = SYS(h).all_keypoints_;
    if ((int)all.size() <= img) all.resize(img + 1);
    all[img].assign(n, sift::Keypoint{});
    for (int k = 0; k < n; ++k) { all[img][k].x = xy[2 * k]; all[img][k].y = xy[2 * k + 1]; }
}
void ref_sys_set_matches(void *h, int i, int j, int n, const int32_t *pairs)
{
    lvba::LvbaSystem &s = SYS(h);
    const int N = (int)s.all_keypoints_.size();
    s.all_matches_.resize((size_t)N * (N - 1) / 2);
    auto &m = s.all_matches_[lvba::pairIndex(i, j, N)];
    m.clear();
    for (int k = 0; k < n; ++k) m.emplace_back(pairs[2 * k], pairs[2 * k + 1]);
}
void ref_sys_set_fusion_params(void *h, int obser_thr, double min_view_angle_deg, double reproj_mean_thr_px)
{
    SYS(h).obser_thr_ = obser_thr; SYS(h).min_view_angle_deg_ = min_view_angle_deg; SYS(h).reproj_mean_thr_px_ = reproj_mean_thr_px;
}
int ref_sys_build_tracks(void *h) { return guarded([&] { SYS(h).BuildTracksAndFuse3D(); }); } // :921-1263
int ref_sys_n_tracks(void *h) { return (int)SYS(h).tracks_.size(); }
int ref_sys_track_sizes(void *h, int t, int *n_inl) { *n_inl = (int)SYS(h).tracks_[t].inlier_indices.size(); return (int)SYS(h).tracks_[t].observations.size(); }
void ref_sys_track(void *h, int t, double *X, int32_t *obs /*[n][2]*/, int32_t *inl)
{
    const lvba::Track &tr = SYS(h).tracks_[t];
    put_v(tr.Xw_fused, X);
    for (size_t k = 0; k < tr.observations.size(); ++k) { obs[2 * k] = tr.observations[k].first; obs[2 * k + 1] = tr.observations[k].second; }
    std::copy(tr.inlier_indices.begin(), tr.inlier_indices.end(), inl);
}

// optimizeCameraPoses (:1423-1670) with the recording ceres::Problem.  sol_* (may be null): a solution to install in place of
// the Ceres solve, in the order of the recorded blocks (needs a previous call to learn that order).
int ref_sys_optimize(void *h, const double *sol_q, const double *sol_t, const double *sol_X, int n_cams, int n_points)
{
    g_rec.sol_q.clear(); g_rec.sol_t.clear(); g_rec.sol_X.clear();
    if (sol_q && sol_t && sol_X) {
        g_rec.sol_q.assign(sol_q, sol_q + 4 * (size_t)n_cams);
        g_rec.sol_t.assign(sol_t, sol_t + 3 * (size_t)n_cams);
        g_rec.sol_X.assign(sol_X, sol_X + 3 * (size_t)n_points);
    }
    ceres::lvba_solve_hook() = solve_hook;
    const int rc = guarded([&] { SYS(h).optimizeCameraPoses(); });
    ceres::lvba_solve_hook() = nullptr;
    return rc == 0 && !g_rec.valid ? 1 : rc; // 1: the reference returned before building a problem
}
#ifdef LVBA_DROPIN
// optimizeCameraPoses of the drop-in build: ceres::Solve runs lvba_visual_refine on the GPU (dropin_solve_hook)
int ref_sys_optimize_dropin(void *h)
{
    ceres::lvba_solve_hook() = dropin_solve_hook;
    const int rc = guarded([&] { SYS(h).optimizeCameraPoses(); });
    ceres::lvba_solve_hook() = nullptr;
    return rc;
}
// out: GPU damping_iter calls, their LM iterations, GPU visual solves, termination of the last, its LM iterations
void ref_sys_dropin_call_diffs(double *out) { std::copy(lvba_dropin::lidar_call_diff, lvba_dropin::lidar_call_diff + 16, out); }
void ref_sys_dropin_stats(int *out, double *cost)
{
    out[0] = lvba_dropin::n_lidar_calls; out[1] = lvba_dropin::n_lidar_iterations; out[2] = lvba_dropin::n_visual_calls;
    out[3] = lvba_dropin::visual_termination; out[4] = lvba_dropin::visual_iterations;
    cost[0] = lvba_dropin::visual_cost0; cost[1] = lvba_dropin::visual_cost1;
}
#endif
// out: n_cams n_points n_residual_blocks max_num_iterations linear_solver_type(3 = DENSE_SCHUR)
void ref_sys_problem_info(int *out)
{
    out[0] = g_rec.n_cams; out[1] = g_rec.n_points; out[2] = (int)g_rec.res.size(); out[3] = g_rec.max_iter; out[4] = g_rec.linear_solver;
}
double ref_sys_problem_cost() { return g_rec.cost0; }
void ref_sys_problem_blocks(double *q0, double *t0, double *X0, double *plane, int32_t *q_const, int32_t *t_const, int32_t *q_tangent)
{
    std::copy(g_rec.q0.begin(), g_rec.q0.end(), q0);
    std::copy(g_rec.t0.begin(), g_rec.t0.end(), t0);
    std::copy(g_rec.X0.begin(), g_rec.X0.end(), X0);
    std::copy(g_rec.plane.begin(), g_rec.plane.end(), plane);
    std::copy(g_rec.q_const.begin(), g_rec.q_const.end(), q_const);
    std::copy(g_rec.t_const.begin(), g_rec.t_const.end(), t_const);
    std::copy(g_rec.q_manifold.begin(), g_rec.q_manifold.end(), q_tangent);
}
// per residual block: kind (2 = reprojection, 1 = plane), camera, point, r[2], Huber scale (0 = none)
void ref_sys_problem_residuals(int32_t *kind, int32_t *cam, int32_t *point, double *r, double *loss_a)
{
    for (size_t i = 0; i < g_rec.res.size(); ++i) {
        kind[i] = g_rec.res[i].kind; cam[i] = g_rec.res[i].cam; point[i] = g_rec.res[i].point;
        r[2 * i] = g_rec.res[i].r[0]; r[2 * i + 1] = g_rec.res[i].r[1]; loss_a[i] = g_rec.res[i].loss_a;
    }
}

// per residual block: the pixel observation of a reprojection residual (0, 0 for plane residuals)
void ref_sys_problem_uv(double *uv)
{
    for (size_t i = 0; i < g_rec.res.size(); ++i) { uv[2 * i] = g_rec.res[i].uv[0]; uv[2 * i + 1] = g_rec.res[i].uv[1]; }
}

// loadFromColmapDB (:510-685) on <data_path>/<data_config/colmap_db_path>: key points and inlier matches into all_keypoints_ /
// all_matches_ (image_pairs_ = all i < j in pairIndex order, :460-464).  Returns 1 when the reference accepted the database.
int ref_sys_load_colmap_db(void *h)
{
    bool ok = false;
    if (guarded([&] { ok = SYS(h).loadFromColmapDB(); }) != 0) return -1;
    return ok ? 1 : 0;
}
int ref_sys_n_keypoints(void *h, int img) { return (int)SYS(h).all_keypoints_[img].size(); }
void ref_sys_keypoints(void *h, int img, float *xyse /*[n][4]: x y sigma extremum*/)
{
    const auto &v = SYS(h).all_keypoints_[img];
    for (size_t k = 0; k < v.size(); ++k) { xyse[4 * k] = v[k].x; xyse[4 * k + 1] = v[k].y; xyse[4 * k + 2] = v[k].sigma; xyse[4 * k + 3] = v[k].extremum_val; }
}
int ref_sys_n_matches(void *h, int i, int j)
{
    const int N = (int)SYS(h).images_ids_.size();
    return (int)SYS(h).all_matches_[lvba::pairIndex(i, j, N)].size();
}
void ref_sys_matches(void *h, int i, int j, int32_t *out)
{
    const int N = (int)SYS(h).images_ids_.size();
    const auto &m = SYS(h).all_matches_[lvba::pairIndex(i, j, N)];
    for (size_t k = 0; k < m.size(); ++k) { out[2 * k] = m[k].first; out[2 * k + 1] = m[k].second; }
}

// VisualizeOptComparison (:1932-2144) = the COLMAP text export (<data_path>/Colmap/sparse/images.txt, points3D.txt) after
// colourising the LiDAR map from the images.  No image codec here: cv::imread hands out a synthetic w x h pattern (see
// oracle/shim/opencv2/opencv.hpp).  Empties DatasetIO::pl_fulls_ as the reference does (:2143): call it last.
int ref_sys_export_colmap(void *h, int w, int hgt)
{
    cv::lvba_synthetic_image_size()[0] = w; cv::lvba_synthetic_image_size()[1] = hgt;
    const int rc = guarded([&] {
        SYS(h).VisualizeOptComparison(SYS(h).images_ids_, true);
        SYS(h).fout_poses_after.close();
        SYS(h).fout_points_after.close();
    });
    cv::lvba_synthetic_image_size()[0] = cv::lvba_synthetic_image_size()[1] = 0;
    return rc;
}

} // extern "C"
