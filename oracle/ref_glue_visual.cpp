// ref_glue_visual.cpp -- TEST INFRASTRUCTURE ONLY: C entry points around the REFERENCE'S OWN include/utils.hpp (cost
// functors of the visual stage, camera model helpers, track-filter helpers), compiled from where it lies under
// /root/reference against the stand-ins of oracle/shim (OpenCV / PCL / Sophus: declarations only; Eigen: the stand-in of
// lvba_eigen_standin.h; Ceres: a forward-mode Jet so that the functors are differentiated as AutoDiffCostFunction would,
// and ceres::QuaternionRotatePoint RESTATED FROM MEMORY of Ceres 2.1.0 -- that one function is not pinned).  Second
// translation unit of oracle/_ref/libbalm_ref.so; tests/test_ref_pin.py uses it to pin oracle/visual_oracle.py's functors,
// oracle/track_oracle.py's camera model and global-lvba_amd/dataset.py's timestamp parser.
#include "utils.hpp"
#include <omp.h>

extern "C" {

// One residual + Jacobian evaluation pass over a whole visual problem with the reference's own functors, differentiated with
// Jets as ceres::AutoDiffCostFunction does (src/lvba_system.cpp:1615-1639: one ReprojErrorWhitenedDistorted per observation of
// every landmark that has a plane, one PointPlaneErrorWhitened per such landmark), OpenMP over the landmarks like Ceres'
// threaded evaluation (:1575 num_threads).  bench.py times it as the CPU baseline of the visual leg's factor kernels.
// Returns the wall seconds; *cost = 1/2 sum r^2, *jac_sum a checksum of all Jacobian entries (keeps the work alive).
double ref_visual_jacobian_pass(int64_t n_tracks, const int64_t *obs_off, const int32_t *obs_cam, const double *obs_uv,
                                const double *q, const double *t, const double *X, const double *plane, const uint8_t *valid,
                                const double *intr, double sigma_px, double sigma_plane, int nthreads, double *cost, double *jac_sum)
{
    typedef ceres::Jet<double, 10> J10;
    typedef ceres::Jet<double, 3> J3;
    double c = 0.0, js = 0.0;
    const double t0 = omp_get_wtime();
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 256) reduction(+ : c, js)
    for (int64_t a = 0; a < n_tracks; ++a) {
        if (!valid[a]) continue; // landmarks without a plane are left out with their observations (:1598-1603)
        const double *Xa = X + 3 * a;
        for (int64_t o = obs_off[a]; o < obs_off[a + 1]; ++o) {
            const int m = obs_cam[o];
            lvba::ReprojErrorWhitenedDistorted f(obs_uv[2 * o], obs_uv[2 * o + 1], intr[0], intr[1], intr[2], intr[3], intr[4], intr[5],
                                                 intr[6], intr[7], sigma_px, sigma_px);
            J10 jq[4], jt[3], jX[3], jr[2];
            for (int i = 0; i < 4; ++i) jq[i] = J10(q[4 * m + i], i);
            for (int i = 0; i < 3; ++i) jt[i] = J10(t[3 * m + i], 4 + i);
            for (int i = 0; i < 3; ++i) jX[i] = J10(Xa[i], 7 + i);
            f(jq, jt, jX, jr);
            for (int r = 0; r < 2; ++r) {
                c += jr[r].a * jr[r].a;
                for (int i = 0; i < 10; ++i) js += jr[r].v[i];
            }
        }
        lvba::PointPlaneErrorWhitened fp(Eigen::Vector3d(plane[4 * a], plane[4 * a + 1], plane[4 * a + 2]), plane[4 * a + 3], sigma_plane);
        J3 jX[3], jr[1];
        for (int i = 0; i < 3; ++i) jX[i] = J3(Xa[i], i);
        fp(jX, jr);
        c += jr[0].a * jr[0].a;
        for (int i = 0; i < 3; ++i) js += jr[0].v[i];
    }
    const double dt = omp_get_wtime() - t0;
    *cost = 0.5 * c;
    *jac_sum = js;
    return dt;
}

// ReprojErrorWhitenedDistorted (utils.hpp:51-127).  intr = fx fy cx cy k1 k2 p1 p2.  r[2]; J [2][10] = d r / d (q[4], t[3], X[3])
// in the ambient parameters, as the AutoDiffCostFunction<.., 2, 4, 3, 3> of :117 would hand to the solver.
void ref_reproj(const double *q, const double *t, const double *X, const double *uv, const double *intr, double su, double sv,
                double *r, double *J)
{
    lvba::ReprojErrorWhitenedDistorted f(uv[0], uv[1], intr[0], intr[1], intr[2], intr[3], intr[4], intr[5], intr[6], intr[7], su, sv);
    f(q, t, X, r);
    typedef ceres::Jet<double, 10> J10;
    J10 jq[4], jt[3], jX[3], jr[2];
    for (int i = 0; i < 4; ++i) jq[i] = J10(q[i], i);
    for (int i = 0; i < 3; ++i) jt[i] = J10(t[i], 4 + i);
    for (int i = 0; i < 3; ++i) jX[i] = J10(X[i], 7 + i);
    f(jq, jt, jX, jr);
    for (int a = 0; a < 2; ++a)
        for (int i = 0; i < 10; ++i) J[10 * a + i] = jr[a].v[i];
}

// PointPlaneErrorWhitened (utils.hpp:129-147).  r[1]; J[3] = d r / d X
void ref_plane(const double *n, double d, double sigma, const double *X, double *r, double *J)
{
    lvba::PointPlaneErrorWhitened f(Eigen::Vector3d(n[0], n[1], n[2]), d, sigma);
    f(X, r);
    typedef ceres::Jet<double, 3> J3;
    J3 jX[3], jr[1];
    for (int i = 0; i < 3; ++i) jX[i] = J3(X[i], i);
    f(jX, jr);
    for (int i = 0; i < 3; ++i) J[i] = jr[0].v[i];
}

static lvba::CameraIntrinsics cam_of(const double *intr)
{
    lvba::CameraIntrinsics c;
    c.fx = intr[0]; c.fy = intr[1]; c.cx = intr[2]; c.cy = intr[3]; c.k1 = intr[4]; c.k2 = intr[5]; c.p1 = intr[6]; c.p2 = intr[7];
    return c;
}
int ref_distort(const double *intr, double x, double y, double *out) { return lvba::distortNormalized(cam_of(intr), x, y, out, out + 1); }
int ref_undistort(const double *intr, double u, double v, double *out) { return lvba::undistortPixelToNormalized(cam_of(intr), u, v, out, out + 1); }
// out = u, v, Zc
int ref_project_world(const double *intr, const double *Rcw, const double *tcw, const double *Xw, double *out)
{
    Eigen::Matrix3d R;
    R << Rcw[0], Rcw[1], Rcw[2], Rcw[3], Rcw[4], Rcw[5], Rcw[6], Rcw[7], Rcw[8];
    return lvba::projectWorldToPixel(cam_of(intr), R, Eigen::Vector3d(tcw[0], tcw[1], tcw[2]), Eigen::Vector3d(Xw[0], Xw[1], Xw[2]),
                                     out, out + 1, out + 2);
}
int ref_backproject(const double *intr, double u, double v, double depth, double *Xc)
{
    Eigen::Vector3d p;
    const bool ok = lvba::backProjectPixelDepthDistorted(cam_of(intr), u, v, depth, &p);
    if (ok) { Xc[0] = p[0]; Xc[1] = p[1]; Xc[2] = p[2]; }
    return ok;
}
void ref_cam_to_world(const double *Xc, const double *Rcw, const double *tcw, double *Xw)
{
    Eigen::Matrix3d R;
    R << Rcw[0], Rcw[1], Rcw[2], Rcw[3], Rcw[4], Rcw[5], Rcw[6], Rcw[7], Rcw[8];
    const Eigen::Vector3d p = lvba::camToWorld(Eigen::Vector3d(Xc[0], Xc[1], Xc[2]), R, Eigen::Vector3d(tcw[0], tcw[1], tcw[2]));
    Xw[0] = p[0]; Xw[1] = p[1]; Xw[2] = p[2];
}
int64_t ref_pair_index(int i, int j, int N) { return (int64_t)lvba::pairIndex(i, j, N); }
double ref_compute_mad(int64_t n, const double *resid) { return lvba::computeMAD(std::vector<double>(resid, resid + n)); }
// returns the number of inliers written to out
int64_t ref_pick_largest_cluster(int64_t n_pts, const double *pts, int64_t n_valid, const int32_t *idx_valid, int32_t *out)
{
    std::vector<Eigen::Vector3d> P;
    for (int64_t i = 0; i < n_pts; ++i) P.push_back(Eigen::Vector3d(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]));
    std::vector<int> iv(idx_valid, idx_valid + n_valid), inl;
    lvba::pickLargestClusterAsInliers(P, iv, inl);
    for (size_t i = 0; i < inl.size(); ++i) out[i] = inl[i];
    return (int64_t)inl.size();
}
void ref_euler_to_rot(double roll, double pitch, double yaw, double *R)
{
    const Eigen::Matrix3d M = lvba::EulerToRot<double>(roll, pitch, yaw);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) R[3 * r + c] = M(r, c);
}
// fetchDepthBilinear (utils.hpp:246-275) on a CV_32FC1 image [h][w]
int ref_fetch_depth_bilinear(int h, int w, const float *depth, float u, float v, float depth_scale, float *out)
{
    cv::Mat m(h, w, CV_32FC1);
    std::memcpy(m.data, depth, sizeof(float) * (size_t)h * w);
    return lvba::fetchDepthBilinear(m, u, v, *out, depth_scale);
}
int ref_parse_timestamp(const char *name, double *ts) { return parseTimestampFromName(std::string(name), *ts); }

} // extern "C"
