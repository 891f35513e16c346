// fusion.hip -- LiDAR-assisted landmark initialisation on the device (the step before the
// visual solve): depth images rendered from the refined LiDAR map, then one lane per feature track forms the depth-fused and
// the triangulated candidate and picks one.
//
// Replaces, of the reference (paths relative to /root/reference):
//   LvbaSystem::buildGridMapFromOptimized   src/lvba_system.cpp:1266-1338   every scan point in the world frame, hashed into
//                                           0.5 m voxels (float key quotient, -1 if negative); per image: the voxels touched
//                                           by the scans within +-0.5 s of it
//   LvbaSystem::generateDepthWithVoxel      src/lvba_system.cpp:835-919     z-buffer over ALL map points of those voxels:
//                                           pixel (int)u, (int)v; Z < 1e-3 skipped; the smallest (float)Z wins
//   LvbaSystem::BuildTracksAndFuse3D        src/lvba_system.cpp:1016-1225   per BFS component: depth-fused candidate (bilinear
//                                           depth, back-projection through the distortion model, 0.12 m consistency with the
//                                           first valid observation, first observation per image, greedy view-angle filter,
//                                           mean reprojection error), triangulation candidate (DLT seed, the same filter, DLT
//                                           again), selection by mean reprojection error
//   fetchDepthBilinear, backProjectPixelDepthDistorted, camToWorld          include/utils.hpp:235-284
// The reference walks std::unordered_map<int,int> (image -> observation) wherever it iterates over a track's images, an
// unspecified order; here images are visited in the order of their first occurrence in the component.
//
// Device design: the grid map is one radix sort of (voxel key, point) pairs; an image's voxel set is a mark array filled by
// the points of the scans in its time window (a contiguous range of the scan set); marked voxels are expanded to point
// work items by a scan, one thread per map point projects and does one atomicMin on the float bits of the depth image
// (positive floats order like their bit patterns, so the smallest (float)Z wins whatever the order, as in the reference).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "lvba_common.h"
#include "mempool.h"
#include "voxel_internal.h"
#include "tracks_device.h"
#include "fusion_device.h"
#include "../../include/lvba_hip.h"

using namespace lvba;

struct lvba_depth_s {
    int device = 0;
    int n_images = 0, width = 0, height = 0;
    float *d_depth = nullptr; // [n_images][height][width], 0 = no return
};

namespace {

__global__ void gm_world_kernel(int64_t P, const float *__restrict__ pts, const int64_t *__restrict__ frame_off, int n_frames,
                                const double *__restrict__ poses, double vox, double *__restrict__ world,
                                uint64_t *__restrict__ key, uint32_t *__restrict__ idx, int *__restrict__ err)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    int lo = 0, hi = n_frames;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (frame_off[mid] <= i) lo = mid; else hi = mid;
    }
    const double *T = poses + 12 * (int64_t)lo;
    const double p0 = pts[3 * i], p1 = pts[3 * i + 1], p2 = pts[3 * i + 2];
    double pw[3];
    pw[0] = T[0] * p0 + T[1] * p1 + T[2] * p2 + T[9];
    pw[1] = T[3] * p0 + T[4] * p1 + T[5] * p2 + T[10];
    pw[2] = T[6] * p0 + T[7] * p1 + T[8] * p2 + T[11];
    int64_t k[3];
    if (!root_key_of(pw, vox, k)) { *err = 1; k[0] = k[1] = k[2] = 0; }
    world[3 * i] = pw[0]; world[3 * i + 1] = pw[1]; world[3 * i + 2] = pw[2];
    key[i] = pack_key(k);
    idx[i] = (uint32_t)i;
}
__global__ void gm_heads_kernel(int64_t P, const uint64_t *__restrict__ key_s, uint32_t *__restrict__ head)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P) head[i] = (i == 0 || key_s[i] != key_s[i - 1]) ? 1u : 0u;
}
// vidx_s = inclusive scan of head - 1: voxel of every sorted position; voff[v] = first sorted position of voxel v
__global__ void gm_tables_kernel(int64_t P, const uint32_t *__restrict__ head, const uint32_t *__restrict__ incl,
                                 const uint32_t *__restrict__ order, uint32_t *__restrict__ vox_of_point,
                                 int64_t *__restrict__ voff, int64_t V)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const uint32_t v = incl[i] - 1;
    vox_of_point[order[i]] = v;
    if (head[i]) voff[v] = i;
    if (i == P - 1) voff[V] = P;
}
__global__ void gm_mark_kernel(int64_t n, const uint32_t *__restrict__ vox_of_point, uint32_t *__restrict__ mark)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) mark[vox_of_point[i]] = 1u;
}
__global__ void gm_count_kernel(int64_t V, const uint32_t *__restrict__ mark, const int64_t *__restrict__ voff,
                                int64_t *__restrict__ cnt)
{
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v < V) cnt[v] = mark[v] ? voff[v + 1] - voff[v] : 0;
    if (v == V) cnt[V] = 0;
}
// one thread per map point of the image's voxel set (generateDepthWithVoxel's inner loop, :886-905)
__global__ void gm_render_kernel(int64_t M, int64_t V, const int64_t *__restrict__ start /*[V+1] exclusive scan of cnt*/,
                                 const int64_t *__restrict__ voff, const uint32_t *__restrict__ order,
                                 const double *__restrict__ world, const double *__restrict__ Rcw, const double *__restrict__ tcw,
                                 TrkIntr cam, int width, int height, unsigned int *__restrict__ depth_bits)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    int64_t lo = 0, hi = V; // last v with start[v] <= j (empty voxels have start[v] == start[v+1] and are skipped by "last")
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (start[mid] <= j) lo = mid; else hi = mid;
    }
    const int64_t pos = voff[lo] + (j - start[lo]);
    const double *pw = world + 3 * (int64_t)order[pos];
    const double X0 = Rcw[0] * pw[0] + Rcw[1] * pw[1] + Rcw[2] * pw[2] + tcw[0];
    const double X1 = Rcw[3] * pw[0] + Rcw[4] * pw[1] + Rcw[5] * pw[2] + tcw[1];
    const double Z = Rcw[6] * pw[0] + Rcw[7] * pw[1] + Rcw[8] * pw[2] + tcw[2];
    if (!(Z >= 1e-3)) return; // :891 (NaN falls out here as it does through allFinite upstream)
    double uu, vv;
    if (!trk_project_cam(cam, X0, X1, Z, uu, vv)) return;
    if (!(fabs(uu) < 2.0e9 && fabs(vv) < 2.0e9)) return; // static_cast<int> of anything larger is undefined upstream
    const int u = (int)uu, v = (int)vv;
    if (u < 0 || u >= width || v < 0 || v >= height) return;
    atomicMin(depth_bits + (int64_t)v * width + u, __float_as_uint((float)Z));
}
__global__ void gm_finish_kernel(int64_t n, unsigned int *__restrict__ bits)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && bits[i] == 0x7f800000u) bits[i] = 0u; // untouched pixels: +inf -> 0
}

// ---- per-track fusion: fusion_device.h (host/device-neutral, also run on the CPU by the tests) ----------------------
__global__ void fuse_kernel(int64_t n_tracks, const int64_t *__restrict__ obs_off, const int32_t *__restrict__ obs_img,
                            const float *__restrict__ obs_uv, const float *__restrict__ depth, int width, int height,
                            const double *__restrict__ Rcw, const double *__restrict__ tcw, int32_t n_images, TrkIntr cam,
                            int obser_thr, double cos_min, double reproj_thr, double *__restrict__ pts /*[O][3] scratch*/,
                            double *__restrict__ dirs /*[O][3] scratch*/, uint8_t *__restrict__ flag /*[O] scratch*/,
                            int32_t *__restrict__ idx /*[O][2] scratch*/, uint8_t *__restrict__ status, double *__restrict__ Xout, double *__restrict__ err_out,
                            uint8_t *__restrict__ kept_out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tracks) return;
    fuse_track(t, obs_off, obs_img, obs_uv, depth, width, height, Rcw, tcw, n_images, cam, obser_thr, cos_min, reproj_thr, pts, dirs,
               flag, idx, status, Xout, err_out, kept_out);
}

int32_t check_device(int32_t device)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return lvba_fail(LVBA_ERR_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return lvba_fail(LVBA_ERR_ARG, "device %d out of range [0,%d)", device, ndev);
    return LVBA_OK;
}

} // namespace

extern "C" int32_t lvba_depth_render(lvba_scans_t sc, const double *scan_poses, const double *scan_times, int32_t n_images,
                                     const double *image_times, const double *Rcw, const double *tcw, const double intr[8],
                                     int32_t width, int32_t height, double half_window_s, double voxel_size, lvba_depth_t *out)
{
    if (out) *out = nullptr;
    if (!sc || !scan_poses || !scan_times || !image_times || !Rcw || !tcw || !intr || !out || n_images < 0 || width < 2 ||
        height < 2 || !(voxel_size > 0) || !(half_window_s >= 0))
        return lvba_fail(LVBA_ERR_ARG, "null or out-of-range argument");
    HIPCHK(hipSetDevice(sc->device));
    const int nf = sc->n_frames;
    const int64_t P = sc->frame_off[nf];
    lvba_depth_s *h = new (std::nothrow) lvba_depth_s();
    if (!h) return lvba_fail(LVBA_ERR_NOMEM, "host allocation failed");
    h->device = sc->device; h->n_images = n_images; h->width = width; h->height = height;
    const int64_t npix = (int64_t)width * height;
    {
        void *raw = nullptr;
        hipError_t e = hipMalloc(&raw, (size_t)std::max<int64_t>(1, n_images * npix) * 4);
        if (e != hipSuccess) { delete h; return lvba_fail(e == hipErrorOutOfMemory ? LVBA_ERR_NOMEM : LVBA_ERR_DEVICE, "depth images: %s", hipGetErrorString(e)); }
        h->d_depth = (float *)raw;
    }
    struct Guard { lvba_depth_s *h; ~Guard() { if (h) { (void)hipFree(h->d_depth); delete h; } } } guard{h};
    hipStream_t s = nullptr;
    HIPCHK(lvba::StreamCache::get().acquire(&s));
    struct SG { hipStream_t s; ~SG() { lvba::StreamCache::get().release(s); } } sg{s};
    if (n_images == 0) { *out = h; guard.h = nullptr; return LVBA_OK; }
    // every pixel starts at +inf
    HIPCHK(hipMemsetD32Async((hipDeviceptr_t)h->d_depth, 0x7f800000, (size_t)(n_images * npix), s));
    if (P > 0) {
        DevBuf d_poses(s), world(s), key(s), idx(s), key_s(s), order(s), head(s), incl(s), vop(s), voff(s), d_err(s), d_R(s), d_t(s);
        HIPCHK(d_poses.alloc(96 * (size_t)nf)); HIPCHK(world.alloc(24 * (size_t)P)); HIPCHK(key.alloc(8 * (size_t)P));
        HIPCHK(idx.alloc(4 * (size_t)P)); HIPCHK(key_s.alloc(8 * (size_t)P)); HIPCHK(order.alloc(4 * (size_t)P));
        HIPCHK(head.alloc(4 * (size_t)P)); HIPCHK(incl.alloc(4 * (size_t)P)); HIPCHK(vop.alloc(4 * (size_t)P)); HIPCHK(d_err.alloc(4));
        HIPCHK(d_R.alloc(72 * (size_t)n_images)); HIPCHK(d_t.alloc(24 * (size_t)n_images));
        HIPCHK(hipMemcpyAsync(d_poses.p, scan_poses, 96 * (size_t)nf, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(d_R.p, Rcw, 72 * (size_t)n_images, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(d_t.p, tcw, 24 * (size_t)n_images, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemsetAsync(d_err.p, 0, 4, s));
        gm_world_kernel<<<grid_for(P, 256), 256, 0, s>>>(P, sc->d_pts, sc->d_frame_off, nf, d_poses.as<double>(), voxel_size,
                                                         world.as<double>(), key.as<uint64_t>(), idx.as<uint32_t>(), d_err.as<int>());
        HIPCHK(hipGetLastError());
        int err = 0;
        HIPCHK(hipMemcpyAsync(&err, d_err.p, 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (err) return lvba_fail(LVBA_ERR_ARG, "a scan point is non-finite or outside +-2^20 grid voxels");
        TRY(sort_pairs(s, key.as<uint64_t>(), key_s.as<uint64_t>(), idx.as<uint32_t>(), order.as<uint32_t>(), (size_t)P, 63));
        gm_heads_kernel<<<grid_for(P, 256), 256, 0, s>>>(P, key_s.as<uint64_t>(), head.as<uint32_t>());
        TRY(scan_incl<uint32_t>(s, head.as<uint32_t>(), incl.as<uint32_t>(), (size_t)P));
        uint32_t V32 = 0;
        HIPCHK(hipMemcpy(&V32, incl.as<uint32_t>() + (P - 1), 4, hipMemcpyDeviceToHost));
        const int64_t V = V32;
        HIPCHK(voff.alloc(8 * ((size_t)V + 1)));
        gm_tables_kernel<<<grid_for(P, 256), 256, 0, s>>>(P, head.as<uint32_t>(), incl.as<uint32_t>(), order.as<uint32_t>(),
                                                          vop.as<uint32_t>(), voff.as<int64_t>(), V);
        HIPCHK(hipGetLastError());
        DevBuf mark(s), cnt(s), start(s);
        HIPCHK(mark.alloc(4 * (size_t)V)); HIPCHK(cnt.alloc(8 * ((size_t)V + 1))); HIPCHK(start.alloc(8 * ((size_t)V + 1)));
        const TrkIntr cam{intr[0], intr[1], intr[2], intr[3], intr[4], intr[5], intr[6], intr[7]};
        for (int m = 0; m < n_images; ++m) {
            // scans with t_img - half <= t <= t_img + half: std::lower_bound / std::upper_bound on the (sorted) scan times
            const double t0 = image_times[m] - half_window_s, t1 = image_times[m] + half_window_s;
            const int flo = (int)(std::lower_bound(scan_times, scan_times + nf, t0) - scan_times);
            const int fhi = (int)(std::upper_bound(scan_times, scan_times + nf, t1) - scan_times);
            if (fhi <= flo) continue;
            const int64_t p0 = sc->frame_off[flo], np = sc->frame_off[fhi] - p0;
            if (np <= 0) continue;
            HIPCHK(hipMemsetAsync(mark.p, 0, 4 * (size_t)V, s));
            gm_mark_kernel<<<grid_for(np, 256), 256, 0, s>>>(np, vop.as<uint32_t>() + p0, mark.as<uint32_t>());
            gm_count_kernel<<<grid_for(V + 1, 256), 256, 0, s>>>(V, mark.as<uint32_t>(), voff.as<int64_t>(), cnt.as<int64_t>());
            HIPCHK(hipGetLastError());
            TRY(scan_excl<int64_t>(s, cnt.as<int64_t>(), start.as<int64_t>(), (size_t)V + 1));
            int64_t M = 0;
            HIPCHK(hipMemcpy(&M, start.as<int64_t>() + V, 8, hipMemcpyDeviceToHost));
            if (M <= 0) continue;
            gm_render_kernel<<<grid_for(M, 256), 256, 0, s>>>(M, V, start.as<int64_t>(), voff.as<int64_t>(), order.as<uint32_t>(),
                                                              world.as<double>(), d_R.as<double>() + 9 * (int64_t)m,
                                                              d_t.as<double>() + 3 * (int64_t)m, cam, width, height,
                                                              reinterpret_cast<unsigned int *>(h->d_depth) + (int64_t)m * npix);
            HIPCHK(hipGetLastError());
        }
    }
    gm_finish_kernel<<<grid_for(n_images * npix, 256), 256, 0, s>>>(n_images * npix, reinterpret_cast<unsigned int *>(h->d_depth));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));
    *out = h;
    guard.h = nullptr;
    return LVBA_OK;
}

extern "C" int32_t lvba_depth_upload(int32_t device, int32_t n_images, int32_t width, int32_t height, const float *depth,
                                     lvba_depth_t *out)
{
    if (out) *out = nullptr;
    if (!depth || !out || n_images < 1 || width < 2 || height < 2) return lvba_fail(LVBA_ERR_ARG, "null or out-of-range argument");
    TRY(check_device(device));
    HIPCHK(hipSetDevice(device));
    lvba_depth_s *h = new (std::nothrow) lvba_depth_s();
    if (!h) return lvba_fail(LVBA_ERR_NOMEM, "host allocation failed");
    h->device = device; h->n_images = n_images; h->width = width; h->height = height;
    const size_t bytes = (size_t)n_images * width * height * 4;
    hipError_t e = hipMalloc((void **)&h->d_depth, bytes);
    if (e == hipSuccess) e = hipMemcpy(h->d_depth, depth, bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(h->d_depth);
        delete h;
        return lvba_fail(e == hipErrorOutOfMemory ? LVBA_ERR_NOMEM : LVBA_ERR_DEVICE, "depth images: %s", hipGetErrorString(e));
    }
    *out = h;
    return LVBA_OK;
}

extern "C" int32_t lvba_depth_info(lvba_depth_t d, int32_t *n_images, int32_t *width, int32_t *height)
{
    if (!d) return lvba_fail(LVBA_ERR_ARG, "null handle");
    if (n_images) *n_images = d->n_images;
    if (width) *width = d->width;
    if (height) *height = d->height;
    return LVBA_OK;
}

extern "C" int32_t lvba_depth_download(lvba_depth_t d, int32_t image, float *depth)
{
    if (!d || !depth || image < 0 || image >= d->n_images) return lvba_fail(LVBA_ERR_ARG, "null handle / image out of range");
    HIPCHK(hipSetDevice(d->device));
    const size_t npix = (size_t)d->width * d->height;
    HIPCHK(hipMemcpy(depth, d->d_depth + (size_t)image * npix, npix * 4, hipMemcpyDeviceToHost));
    return LVBA_OK;
}

extern "C" void lvba_depth_destroy(lvba_depth_t d)
{
    if (!d) return;
    (void)hipSetDevice(d->device);
    (void)hipFree(d->d_depth);
    delete d;
}

extern "C" void lvba_fuse_default_opts(lvba_fuse_opts *o)
{
    if (!o) return;
    o->obser_thr = 3;              // config: obser_thr
    o->reserved = 0;
    o->min_view_angle_deg = 8.0;   // config: min_view_angle_deg
    o->reproj_mean_thr_px = 3.0;   // config: reproj_mean_thr_px
}

extern "C" int32_t lvba_fuse_tracks(int32_t device, lvba_depth_t depth, int32_t n_images, const double *Rcw, const double *tcw,
                                    const double intr[8], int64_t n_tracks, const int64_t *obs_off, const int32_t *obs_img,
                                    const float *obs_uv, const lvba_fuse_opts *opts, uint8_t *status, double *X,
                                    double *mean_reproj, uint8_t *kept)
{
    if (n_images < 1 || n_tracks < 0 || !Rcw || !tcw || !intr || !obs_off || !status || !X || !mean_reproj)
        return lvba_fail(LVBA_ERR_ARG, "null argument or n_images < 1");
    if (n_tracks == 0) return LVBA_OK;
    const int64_t O = obs_off[n_tracks] - obs_off[0];
    if (obs_off[0] != 0 || O < 0 || (O > 0 && (!obs_img || !obs_uv || !kept))) return lvba_fail(LVBA_ERR_ARG, "bad observation arrays");
    lvba_fuse_opts o;
    lvba_fuse_default_opts(&o);
    if (opts) o = *opts;
    if (depth) {
        if (depth->n_images != n_images) return lvba_fail(LVBA_ERR_ARG, "depth handle holds %d images, %d cameras given", depth->n_images, n_images);
        device = depth->device;
    }
    TRY(check_device(device));
    HIPCHK(hipSetDevice(device));
    hipStream_t s = nullptr;
    HIPCHK(lvba::StreamCache::get().acquire(&s));
    struct SG { hipStream_t s; ~SG() { lvba::StreamCache::get().release(s); } } sg{s};
    DevBuf d_off(s), d_img(s), d_uv(s), d_R(s), d_t(s), d_pts(s), d_dirs(s), d_flag(s), d_idx(s), d_st(s), d_X(s), d_err(s), d_kept(s);
    const size_t O1 = (size_t)std::max<int64_t>(O, 1);
    HIPCHK(d_off.alloc(8 * ((size_t)n_tracks + 1))); HIPCHK(d_img.alloc(4 * O1)); HIPCHK(d_uv.alloc(8 * O1));
    HIPCHK(d_R.alloc(72 * (size_t)n_images)); HIPCHK(d_t.alloc(24 * (size_t)n_images));
    HIPCHK(d_pts.alloc(24 * O1)); HIPCHK(d_dirs.alloc(24 * O1)); HIPCHK(d_flag.alloc(O1)); HIPCHK(d_idx.alloc(8 * O1));
    HIPCHK(d_st.alloc((size_t)n_tracks)); HIPCHK(d_X.alloc(24 * (size_t)n_tracks)); HIPCHK(d_err.alloc(8 * (size_t)n_tracks));
    HIPCHK(d_kept.alloc(O1));
    HIPCHK(hipMemcpyAsync(d_off.p, obs_off, 8 * ((size_t)n_tracks + 1), hipMemcpyHostToDevice, s));
    if (O > 0) {
        HIPCHK(hipMemcpyAsync(d_img.p, obs_img, 4 * (size_t)O, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(d_uv.p, obs_uv, 8 * (size_t)O, hipMemcpyHostToDevice, s));
    }
    HIPCHK(hipMemcpyAsync(d_R.p, Rcw, 72 * (size_t)n_images, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d_t.p, tcw, 24 * (size_t)n_images, hipMemcpyHostToDevice, s));
    const TrkIntr cam{intr[0], intr[1], intr[2], intr[3], intr[4], intr[5], intr[6], intr[7]};
    const double cos_min = cos(o.min_view_angle_deg * M_PI / 180.0);
    fuse_kernel<<<(unsigned)((n_tracks + 63) / 64), 64, 0, s>>>(
        n_tracks, d_off.as<int64_t>(), d_img.as<int32_t>(), d_uv.as<float>(), depth ? depth->d_depth : nullptr,
        depth ? depth->width : 0, depth ? depth->height : 0, d_R.as<double>(), d_t.as<double>(), n_images, cam, o.obser_thr, cos_min,
        o.reproj_mean_thr_px, d_pts.as<double>(), d_dirs.as<double>(), d_flag.as<uint8_t>(), d_idx.as<int32_t>(), d_st.as<uint8_t>(), d_X.as<double>(),
        d_err.as<double>(), d_kept.as<uint8_t>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(status, d_st.p, (size_t)n_tracks, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(X, d_X.p, 24 * (size_t)n_tracks, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(mean_reproj, d_err.p, 8 * (size_t)n_tracks, hipMemcpyDeviceToHost, s));
    if (O > 0) HIPCHK(hipMemcpyAsync(kept, d_kept.p, (size_t)O, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return LVBA_OK;
}
