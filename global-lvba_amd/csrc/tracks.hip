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
#include "tracks_device.h"

using namespace lvba;

namespace {

typedef TrkIntr Intr;

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
    double X[3] = {0.0, 0.0, 0.0}, mean;
    int cnt;
    const bool ok = trk_dlt(cam, Rcw, tcw, n_cams, a, (const int32_t *)nullptr, (int)(b - a), obs_cam, obs_uv, X, mean, cnt);
    Xo[0] = X[0]; Xo[1] = X[1]; Xo[2] = X[2];
    cnt_out[i] = cnt;
    err_out[i] = mean;
    ok_out[i] = ok ? 1 : 0;
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
    HIPCHK(lvba::StreamCache::get().acquire(&s));
    struct SG { hipStream_t s; ~SG() { lvba::StreamCache::get().release(s); } } sg{s};
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
