// visual_api.hip -- visual half of the C-ABI (include/lvba_hip.h): packing of the landmark/observation problem and
// the trust-region Levenberg-Marquardt driver that replaces ceres::Solve in LvbaSystem::optimizeCameraPoses
// (reference src/lvba_system.cpp:1571-1665).  The control flow restates Ceres 2.1.0's TrustRegionMinimizer +
// LevenbergMarquardtStrategy (not in /root/reference; restated from its published sources):
// Jacobi scaling fixed at iteration 0, LM diagonal sqrt(clamp(diag J^T J)/radius),
// parameter/function tolerance checks before the accept test, radius /= max(1/3, 1-(2 rho-1)^3) on success and
// /= 2,4,8.. on failure.  Host logic only; arithmetic runs in visual_kernels.hip, balm_pair_kernel and ldlt.hip.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <new>
#include <vector>

#include "host_arena.h"
#include "block_system.h"

using namespace lvba;
#define fail lvba_fail

struct lvba_visual_s {
    BlockSys bs;
    int32_t M = 0;
    int64_t T = 0, Ta = 0, O = 0;
    lvba::hvec<int64_t> act;      // active landmark -> caller track index
    lvba::hvec<int64_t> h_off;    // CSR of the active landmarks (host, kept until finalize)
    lvba::hvec<int32_t> h_cam;    // caller camera index per kept observation
    bool finalized = false;
    double intr[8] = {}, sig_px = 0.5, sig_pl = 0.01;
    // device
    int64_t *d_off = nullptr;
    int32_t *d_cam = nullptr, *d_track_of_obs = nullptr;
    double *d_uv = nullptr, *d_plane = nullptr, *d_uv_cm = nullptr;
    double *d_Jc = nullptr, *d_Jp = nullptr, *d_r = nullptr, *d_rpl = nullptr, *d_Jpl = nullptr;
    double *d_sc_cam = nullptr, *d_sc_pt = nullptr, *d_Lp = nullptr, *d_zp = nullptr, *d_step_p = nullptr, *d_part = nullptr;
    double *d_q = nullptr, *d_t = nullptr, *d_X = nullptr, *d_q2 = nullptr, *d_t2 = nullptr, *d_X2 = nullptr;
    double *d_blkpart = nullptr;   // per-workgroup partials of the scalar reductions
    double *d_scal = nullptr;      // [0]=cost(x) [1]=cost(cand) [2]=model change [3]=|step|^2 [4]=|x|^2
    unsigned long long *d_gmax = nullptr;
    double *d_camsum = nullptr, *d_colsum = nullptr; // sharded runs: per-camera sums awaiting the other ranks' tracks
    double *d_out = nullptr;       // export staging
    double *h_pin = nullptr;
    double *hq = nullptr, *ht = nullptr, *hX = nullptr; // host staging in solver order: ONE pinned block (h_stage), so the state
    double *h_stage = nullptr;                          // uploads / downloads are DMA transfers, not pageable copies

    VisDev dev() const
    {
        VisDev d;
        d.M = M; d.S = bs.S; d.band_blocks = bs.Bb; d.fixed_cam = bs.iperm.empty() ? 0 : bs.iperm[0];
        d.Ta = Ta; d.O = O; d.off = d_off; d.cam = d_cam; d.track_of_obs = d_track_of_obs; d.uv = d_uv; d.uv_cm = d_uv_cm; d.plane = d_plane;
        for (int e = 0; e < 8; ++e) d.intr[e] = intr[e];
        d.inv_sig_px = 1.0 / sig_px; d.inv_sig_pl = 1.0 / std::max(1e-9, sig_pl); // utils.hpp:131
        d.Jc = d_Jc; d.Jp = d_Jp; d.r = d_r; d.rpl = d_rpl; d.Jpl = d_Jpl; d.sc_cam = d_sc_cam; d.sc_pt = d_sc_pt;
        d.Lp = d_Lp; d.zp = d_zp; d.step_p = d_step_p;
        d.csc_off = bs.d_csc_off; d.csc_f = bs.d_csc_f; d.group_of_pos = bs.d_group_of_pos; d.pos_of = bs.d_pos_of;
        d.Y = bs.d_Y; d.part = d_part;
        d.dist = bs.distributed() ? 1 : 0; d.count_cams = bs.rank == 0 ? 1 : 0; d.camsum = d_camsum; d.colsum = d_colsum;
        return d;
    }
};

extern "C" void lvba_visual_default_opts(lvba_visual_opts *o)
{
    if (!o) return;
    o->max_iter = 50; o->reserved = 0; o->initial_radius = 1e4; o->max_radius = 1e16; o->min_radius = 1e-32;
    o->min_relative_decrease = 1e-3; o->min_lm_diagonal = 1e-6; o->max_lm_diagonal = 1e32;
    o->function_tolerance = 1e-6; o->gradient_tolerance = 1e-10; o->parameter_tolerance = 1e-8;
}

extern "C" int32_t lvba_visual_destroy(lvba_visual_t h)
{
    if (!h) return LVBA_OK;
    hipSetDevice(h->bs.device);
    if (h->bs.stream) hipStreamSynchronize(h->bs.stream);
    void *ptrs[] = {h->d_off, h->d_cam, h->d_track_of_obs, h->d_uv, h->d_uv_cm, h->d_plane, h->d_Jc, h->d_Jp, h->d_r, h->d_rpl, h->d_Jpl,
                    h->d_sc_cam, h->d_sc_pt, h->d_Lp, h->d_zp, h->d_step_p, h->d_part, h->d_q, h->d_t, h->d_X, h->d_q2,
                    h->d_t2, h->d_X2, h->d_blkpart, h->d_scal, h->d_gmax, h->d_out, h->d_camsum, h->d_colsum};
    for (void *p : ptrs)
        if (p) lvba::DevicePool::get().free(p);
    if (h->h_pin) hipHostFree(h->h_pin);
    if (h->h_stage) hipHostFree(h->h_stage);
    bs_destroy(h->bs);
    delete h;
    return LVBA_OK;
}

extern "C" int32_t lvba_visual_create(int32_t n_cams, int64_t n_tracks, const int64_t *obs_off, const int32_t *obs_cam,
                                      const double *obs_uv, const double *plane, const uint8_t *valid, const double intr[8],
                                      double sigma_px, double sigma_plane, int32_t device, lvba_visual_t *out)
{
    if (!out) return fail(LVBA_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (n_cams < 1 || n_tracks < 0 || !obs_off || !plane || !valid || !intr) return fail(LVBA_ERR_ARG, "bad sizes or NULL arrays");
    if (!(sigma_px > 0.0)) return fail(LVBA_ERR_ARG, "sigma_px must be > 0");
    const int64_t base = obs_off[0];
    const int64_t Oall = obs_off[n_tracks] - base;
    if (Oall > 0 && (!obs_cam || !obs_uv)) return fail(LVBA_ERR_ARG, "observation arrays are NULL");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(LVBA_ERR_DEVICE, "no HIP device available (liblvba_hip has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(LVBA_ERR_ARG, "device %d out of range [0,%d)", device, ndev);
    lvba_visual_s *h = new (std::nothrow) lvba_visual_s();
    if (!h) return fail(LVBA_ERR_NOMEM, "host allocation failed");
    h->M = n_cams; h->T = n_tracks; h->sig_px = sigma_px; h->sig_pl = sigma_plane;
    memcpy(h->intr, intr, 8 * sizeof(double));
    // keep only landmarks with a valid plane, together with their observations (src/lvba_system.cpp:1598-1603)
    lvba::hvec<double> uv, pl;
    h->h_off.push_back(0);
    for (int64_t i = 0; i < n_tracks; ++i) {
        if (obs_off[i + 1] < obs_off[i]) { delete h; return fail(LVBA_ERR_ARG, "obs_off is not monotone at %lld", (long long)i); }
        if (!valid[i]) continue;
        for (int64_t o = obs_off[i] - base; o < obs_off[i + 1] - base; ++o) {
            if (obs_cam[o] < 0 || obs_cam[o] >= n_cams) { delete h; return fail(LVBA_ERR_ARG, "obs_cam[%lld] = %d out of range", (long long)o, obs_cam[o]); }
            h->h_cam.push_back(obs_cam[o]);
            uv.push_back(obs_uv[2 * o]);
            uv.push_back(obs_uv[2 * o + 1]);
        }
        h->act.push_back(i);
        h->h_off.push_back((int64_t)h->h_cam.size());
        for (int e = 0; e < 4; ++e) pl.push_back(plane[4 * i + e]);
    }
    h->Ta = (int64_t)h->act.size();
    h->O = (int64_t)h->h_cam.size();
    auto bail = [&](int32_t rc) { lvba_visual_destroy(h); return rc; };
#define CTRY(expr) do { int32_t rc_ = (expr); if (rc_ != LVBA_OK) return bail(rc_); } while (0)
#define CHIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return bail(fail(e_ == hipErrorOutOfMemory ? LVBA_ERR_NOMEM : LVBA_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_))); } while (0)
    CTRY(bs_init(h->bs, device));
    BlockSys &bs = h->bs;
    const int64_t Ta = h->Ta, O = h->O, M = n_cams;
    CTRY(bs_dmalloc(bs, &h->d_off, Ta + 1));
    CTRY(bs_dmalloc(bs, &h->d_cam, O));
    CTRY(bs_dmalloc(bs, &h->d_track_of_obs, O));
    CTRY(bs_dmalloc(bs, &h->d_uv, 2 * O));
    CTRY(bs_dmalloc(bs, &h->d_plane, 4 * Ta));
    CHIP(hipMemcpy(h->d_off, h->h_off.data(), (size_t)(Ta + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    if (O) CHIP(hipMemcpy(h->d_uv, uv.data(), (size_t)(2 * O) * sizeof(double), hipMemcpyHostToDevice));
    if (Ta) CHIP(hipMemcpy(h->d_plane, pl.data(), (size_t)(4 * Ta) * sizeof(double), hipMemcpyHostToDevice));
    {
        lvba::hvec<int32_t> too((size_t)O);
        for (int64_t i = 0; i < Ta; ++i)
            for (int64_t o = h->h_off[i]; o < h->h_off[i + 1]; ++o) too[o] = (int32_t)i;
        if (O) CHIP(hipMemcpy(h->d_track_of_obs, too.data(), (size_t)O * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    CTRY(bs_dmalloc(bs, &h->d_Jc, 12 * O)); CTRY(bs_dmalloc(bs, &h->d_Jp, 6 * O)); CTRY(bs_dmalloc(bs, &h->d_r, 2 * O));
    CTRY(bs_dmalloc(bs, &h->d_rpl, Ta)); CTRY(bs_dmalloc(bs, &h->d_Jpl, 3 * Ta));
    CTRY(bs_dmalloc(bs, &h->d_sc_cam, 6 * M)); CTRY(bs_dmalloc(bs, &h->d_sc_pt, 3 * Ta));
    CTRY(bs_dmalloc(bs, &h->d_Lp, 6 * Ta)); CTRY(bs_dmalloc(bs, &h->d_zp, 3 * Ta)); CTRY(bs_dmalloc(bs, &h->d_step_p, 3 * Ta));
    CTRY(bs_dmalloc(bs, &h->d_q, 4 * M)); CTRY(bs_dmalloc(bs, &h->d_t, 3 * M)); CTRY(bs_dmalloc(bs, &h->d_X, 3 * Ta));
    CTRY(bs_dmalloc(bs, &h->d_q2, 4 * M)); CTRY(bs_dmalloc(bs, &h->d_t2, 3 * M)); CTRY(bs_dmalloc(bs, &h->d_X2, 3 * Ta));
    CTRY(bs_dmalloc(bs, &h->d_blkpart, 2 * ((O + 4 * Ta + M) / 256 + 4))); // vis_back_kernel: four lanes per landmark
    CTRY(bs_dmalloc(bs, &h->d_scal, 16));
    CTRY(bs_dmalloc(bs, &h->d_gmax, 2));
    CTRY(bs_dmalloc(bs, &h->d_out, 36 * (int64_t)M + 16));
    CTRY(bs_dmalloc(bs, &h->d_camsum, 12 * (int64_t)M));
    CTRY(bs_dmalloc(bs, &h->d_colsum, 6 * (int64_t)M));
    CHIP(hipHostMalloc((void **)&h->h_pin, 16 * sizeof(double), hipHostMallocDefault));
#undef CTRY
#undef CHIP
    *out = h;
    return LVBA_OK;
}

static int32_t finalize(lvba_visual_s *h)
{
    if (h->finalized) return LVBA_OK;
    BlockSys &bs = h->bs;
    HIPCHK(hipSetDevice(bs.device));
    bs.spd = true; // S = sum Jc^T Jc + D^2 - sum Y Y^T is the Schur complement of a positive definite matrix
    TRY(bs_build(bs, h->M, h->Ta, h->h_off.data(), h->h_cam.data()));
    {
        lvba::hvec<int32_t> cam((size_t)h->O);
        for (int64_t o = 0; o < h->O; ++o) cam[o] = bs.iperm[h->h_cam[o]];
        if (h->O) HIPCHK(hipMemcpy(h->d_cam, cam.data(), (size_t)h->O * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    TRY(bs_dmalloc(bs, &h->d_part, (int64_t)h->M * bs.S * 40));
    TRY(bs_dmalloc(bs, &h->d_uv_cm, 2 * std::max<int64_t>(1, h->O)));
    vis_launch_gather_uv(h->dev(), h->d_uv_cm, bs.stream);
    lvba::hvec<int32_t>().swap(h->h_cam);
    h->finalized = true;
    return LVBA_OK;
}

// caller arrays -> device state in solver order
static int32_t import_state(lvba_visual_s *h, const double *q, const double *t, const double *X)
{
    BlockSys &bs = h->bs;
    if (!h->h_stage) {
        HIPCHK(hipHostMalloc((void **)&h->h_stage, (7 * (size_t)h->M + 3 * (size_t)h->Ta + 8) * sizeof(double), hipHostMallocDefault));
        h->hq = h->h_stage; h->ht = h->hq + 4 * (size_t)h->M; h->hX = h->ht + 3 * (size_t)h->M;
    }
    for (int c = 0; c < h->M; ++c) {
        const int I = bs.iperm[c];
        for (int e = 0; e < 4; ++e) h->hq[4 * I + e] = q[4 * c + e];
        for (int e = 0; e < 3; ++e) h->ht[3 * I + e] = t[3 * c + e];
    }
    for (int64_t i = 0; i < h->Ta; ++i)
        for (int e = 0; e < 3; ++e) h->hX[3 * i + e] = X[3 * h->act[i] + e];
    HIPCHK(hipMemcpyAsync(h->d_q, h->hq, 4 * (size_t)h->M * sizeof(double), hipMemcpyHostToDevice, bs.stream));
    HIPCHK(hipMemcpyAsync(h->d_t, h->ht, 3 * (size_t)h->M * sizeof(double), hipMemcpyHostToDevice, bs.stream));
    if (h->Ta) HIPCHK(hipMemcpyAsync(h->d_X, h->hX, 3 * (size_t)h->Ta * sizeof(double), hipMemcpyHostToDevice, bs.stream));
    HIPCHK(hipStreamSynchronize(bs.stream)); // the host staging vectors may be reused
    return LVBA_OK;
}

static int32_t export_state(lvba_visual_s *h, double *q, double *t, double *X)
{
    BlockSys &bs = h->bs;
    HIPCHK(hipMemcpyAsync(h->hq, h->d_q, 4 * (size_t)h->M * sizeof(double), hipMemcpyDeviceToHost, bs.stream));
    HIPCHK(hipMemcpyAsync(h->ht, h->d_t, 3 * (size_t)h->M * sizeof(double), hipMemcpyDeviceToHost, bs.stream));
    if (h->Ta) HIPCHK(hipMemcpyAsync(h->hX, h->d_X, 3 * (size_t)h->Ta * sizeof(double), hipMemcpyDeviceToHost, bs.stream));
    HIPCHK(hipStreamSynchronize(bs.stream));
    for (int c = 0; c < h->M; ++c) {
        const int I = bs.iperm[c];
        const double *s = &h->hq[4 * I];
        const double nq = sqrt(s[0] * s[0] + s[1] * s[1] + s[2] * s[2] + s[3] * s[3]); // q_eig.normalize(), :1653
        for (int e = 0; e < 4; ++e) q[4 * c + e] = s[e] / nq;
        for (int e = 0; e < 3; ++e) t[3 * c + e] = h->ht[3 * I + e];
    }
    for (int64_t i = 0; i < h->Ta; ++i)
        for (int e = 0; e < 3; ++e) X[3 * h->act[i] + e] = h->hX[3 * i + e];
    return LVBA_OK;
}

// ---- the steps of a linearisation / an iteration that need the other ranks' track shards (no-ops on one rank) -------------
static int32_t enqueue_colnorms(lvba_visual_s *h)
{
    BlockSys &bs = h->bs;
    const VisDev d = h->dev();
    if (!bs.distributed()) { vis_launch_colnorms(d, bs.stream); return LVBA_OK; }
    vis_launch_colsums(d, bs.stream);
    TRY(bs_allreduce(bs, h->d_colsum, 6 * (size_t)h->M));
    vis_launch_colnorm_finish(d, bs.stream);
    return LVBA_OK;
}
static int32_t enqueue_reduced_system(lvba_visual_s *h, double radius, const lvba_visual_opts &o)
{
    BlockSys &bs = h->bs;
    const VisDev d = h->dev();
    vis_launch_reduced_system(d, bs.pair_dev(), h->d_q, h->d_t, h->d_X, radius, o.min_lm_diagonal, o.max_lm_diagonal, bs.Hblk(), bs.hblk_doubles, bs.g(),
                              h->d_gmax, bs.distributed(), bs.stream);
    if (!bs.distributed()) return LVBA_OK;
    TRY(bs_allreduce_hg(bs));                                   // [S blocks | reduced rhs]: sums over the track shards
    TRY(bs_allreduce(bs, h->d_camsum, 12 * (size_t)h->M));      // diag(Jc^T Jc), Jc^T r
    vis_launch_cam_finish(d, radius, o.min_lm_diagonal, o.max_lm_diagonal, bs.Hblk(), h->d_q, h->d_gmax, bs.stream);
    TRY(bs_comm_allreduce(bs, h->d_gmax, 1, ncclInt64, ncclMax)); // bit patterns of non-negative doubles order like integers
    return LVBA_OK;
}
static int32_t allreduce_scalars(lvba_visual_s *h, int first, int count)
{
    if (!h->bs.distributed()) return LVBA_OK;
    return bs_allreduce(h->bs, h->d_scal + first, (size_t)count);
}

extern "C" int32_t lvba_visual_dist_init(lvba_visual_t h, int32_t n_ranks, int32_t rank, const char uid[128])
{
    if (!h || !uid) return fail(LVBA_ERR_ARG, "NULL argument");
    if (h->finalized) return fail(LVBA_ERR_STATE, "dist_init must precede the first cost/linearize/refine call");
    return bs_dist_init(h->bs, n_ranks, rank, uid, nullptr);
}

extern "C" int32_t lvba_visual_dist_init_external(lvba_visual_t h, int32_t n_ranks, int32_t rank, lvba_allreduce_fn fn, void *ctx)
{
    if (!h || !fn) return fail(LVBA_ERR_ARG, "NULL argument");
    if (h->finalized) return fail(LVBA_ERR_STATE, "dist_init must precede the first cost/linearize/refine call");
    return bs_dist_init_external(h->bs, n_ranks, rank, fn, ctx, nullptr);
}

extern "C" int32_t lvba_visual_cost(lvba_visual_t h, const double *q, const double *t, const double *X, double *cost)
{
    if (!h || !q || !t || !X || !cost) return fail(LVBA_ERR_ARG, "NULL argument");
    TRY(finalize(h));
    BlockSys &bs = h->bs;
    HIPCHK(hipSetDevice(bs.device));
    TRY(import_state(h, q, t, X));
    vis_launch_residuals(h->dev(), false, h->d_q, h->d_t, h->d_X, h->d_blkpart, h->d_scal, bs.stream);
    TRY(allreduce_scalars(h, 0, 1));
    HIPCHK(hipMemcpyAsync(h->h_pin, h->d_scal, sizeof(double), hipMemcpyDeviceToHost, bs.stream));
    HIPCHK(hipStreamSynchronize(bs.stream));
    HIPCHK(hipGetLastError());
    *cost = 0.5 * h->h_pin[0];
    return LVBA_OK;
}

extern "C" int32_t lvba_visual_info(lvba_visual_t h, lvba_balm_info_t *info)
{
    if (!h || !info) return fail(LVBA_ERR_ARG, "NULL argument");
    TRY(finalize(h));
    memset(info, 0, sizeof *info);
    info->n_poses = h->M; info->n_ranks = h->bs.n_ranks; info->n_voxels = h->Ta; info->n_voxels_global = h->Ta;
    info->n_factors = h->O; info->n_pairs = h->bs.Q; info->n_blocks = h->bs.nnzb; info->band_blocks = h->bs.Bb;
    info->use_band = h->bs.use_band ? 1 : 0; info->hess_bytes = h->bs.hblk_doubles * 8; info->device_bytes = h->bs.device_bytes;
    return LVBA_OK;
}

extern "C" int32_t lvba_visual_linearize(lvba_visual_t h, const double *q, const double *t, const double *X, double radius,
                                         double *S, double *rhs, double *cost)
{
    if (!h || !q || !t || !X) return fail(LVBA_ERR_ARG, "NULL argument");
    if (!(radius > 0.0)) return fail(LVBA_ERR_ARG, "radius must be > 0");
    TRY(finalize(h));
    BlockSys &bs = h->bs;
    HIPCHK(hipSetDevice(bs.device));
    TRY(import_state(h, q, t, X));
    lvba_visual_opts o;
    lvba_visual_default_opts(&o);
    const VisDev d = h->dev();
    vis_launch_residuals(d, true, h->d_q, h->d_t, h->d_X, h->d_blkpart, h->d_scal, bs.stream);
    TRY(allreduce_scalars(h, 0, 1));
    TRY(enqueue_colnorms(h));
    TRY(enqueue_reduced_system(h, radius, o));
    const int64_t n = 6 * (int64_t)h->M;
    DevBuf dS(bs.stream); // freed on every path, error returns included
    if (S) {
        HIPCHK(dS.alloc((size_t)(n * n) * sizeof(double)));
        launch_export_dense(bs.Hblk(), bs.Bb, h->M, bs.d_perm, dS.as<double>(), bs.stream);
        HIPCHK(hipMemcpyAsync(S, dS.as<double>(), (size_t)(n * n) * sizeof(double), hipMemcpyDeviceToHost, bs.stream));
    }
    if (rhs) {
        launch_export_vec(bs.g(), bs.d_perm, h->M, h->d_out, bs.stream);
        HIPCHK(hipMemcpyAsync(rhs, h->d_out, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, bs.stream));
    }
    HIPCHK(hipMemcpyAsync(h->h_pin, h->d_scal, sizeof(double), hipMemcpyDeviceToHost, bs.stream));
    HIPCHK(hipStreamSynchronize(bs.stream));
    HIPCHK(hipGetLastError());
    if (cost) *cost = 0.5 * h->h_pin[0];
    return LVBA_OK;
}

extern "C" int32_t lvba_visual_refine(lvba_visual_t h, double *q, double *t, double *X, const lvba_visual_opts *opts,
                                      lvba_visual_trace *trace, int32_t trace_cap, int32_t *n_trace, int32_t *termination)
{
    if (!h || !q || !t || !X) return fail(LVBA_ERR_ARG, "NULL argument");
    if (n_trace) *n_trace = 0;
    TRY(finalize(h));
    BlockSys &bs = h->bs;
    HIPCHK(hipSetDevice(bs.device));
    lvba_visual_opts o;
    if (opts) o = *opts; else lvba_visual_default_opts(&o);
    auto wall_us = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tw0 = wall_us();
    TRY(import_state(h, q, t, X));
    const double tw1 = wall_us();
    int32_t rows = 0, term = LVBA_TERM_NO_CONVERGENCE, rc = LVBA_OK;
    auto push = [&](int it, int acc, int valid, double cost, double dc, double sn, double rad, double rho, double gm) {
        if (trace && rows < trace_cap) {
            lvba_visual_trace &r = trace[rows];
            r.iter = it; r.accepted = acc; r.valid = valid; r.reserved = 0; r.cost = cost; r.cost_change = dc; r.step_norm = sn;
            r.radius = rad; r.rho = rho; r.gradient_max_norm = gm;
        }
        rows++;
    };
    VisDev d = h->dev();
    // iteration 0: evaluate, fix the Jacobi scaling
    vis_launch_residuals(d, true, h->d_q, h->d_t, h->d_X, h->d_blkpart, h->d_scal, bs.stream);
    TRY(allreduce_scalars(h, 0, 1));
    TRY(enqueue_colnorms(h));
    HIPCHK(hipMemcpyAsync(h->h_pin, h->d_scal, sizeof(double), hipMemcpyDeviceToHost, bs.stream));
    HIPCHK(hipStreamSynchronize(bs.stream));
    double cost = 0.5 * h->h_pin[0];
    const double tw2 = wall_us();
    double radius = o.initial_radius, decrease_factor = 2.0;
    bool first = true;
    int invalid_run = 0;
    if (!isfinite(cost)) { term = LVBA_TERM_FAILURE; rc = fail(LVBA_NUM_NONFINITE, "non-finite initial cost"); }
    // LVBA_TIMING=vis: HIP events between the phases of every iteration, averages on stderr at the end (tools/visual_bench.py)
    static const bool prof = timing_on("vis", true); // (a stream synchronisation per phase: only when asked for by name)
    constexpr int NPH = 6, PCAP = 64;
    std::vector<hipEvent_t> pev;
    int pn = 0;
    if (prof) {
        pev.resize((size_t)NPH * PCAP);
        for (auto &e : pev) hipEventCreate(&e);
    }
    auto mark = [&](int ph) { if (prof && pn < PCAP) hipEventRecord(pev[(size_t)pn * NPH + ph], bs.stream); };
    for (int it = 1; rc == LVBA_OK; ++it) {
        // linearised system at the current point (its gradient norm belongs to the row of the previous iteration)
        mark(0);
        TRY(enqueue_reduced_system(h, radius, o));
        mark(1);
        TRY(bs_enqueue_solve(bs, 0.0));
        mark(2);
        if (!bs.distributed()) {
            // one rank: back-substitution, candidate, its cost, and one closing kernel that sums the partial lists and writes what
            // the host reads below into the pinned buffer (no reductions of their own, no device-to-host copies)
            mark(3);
            vis_launch_step_and_trial(d, bs.d_dx, h->d_q, h->d_t, h->d_X, h->d_q2, h->d_t2, h->d_X2, h->d_blkpart, h->d_scal, h->d_gmax,
                                      bs.d_status, h->h_pin, bs.stream);
            mark(4);
        } else {
            vis_launch_back(d, bs.d_dx, h->d_q, h->d_t, h->d_X, h->d_blkpart, h->d_scal + 2, bs.stream);
            mark(3);
            vis_launch_apply(d, bs.d_dx, h->d_q, h->d_t, h->d_X, h->d_q2, h->d_t2, h->d_X2, h->d_blkpart, h->d_scal + 3, bs.stream);
            vis_launch_residuals(d, false, h->d_q2, h->d_t2, h->d_X2, h->d_blkpart, h->d_scal + 1, bs.stream);
            mark(4);
            TRY(allreduce_scalars(h, 1, 4)); // candidate cost, model cost change, |step|^2, |x|^2: sums over the track shards
            HIPCHK(hipMemcpyAsync(h->h_pin, h->d_scal, 5 * sizeof(double), hipMemcpyDeviceToHost, bs.stream));
            HIPCHK(hipMemcpyAsync(h->h_pin + 8, h->d_gmax, sizeof(unsigned long long), hipMemcpyDeviceToHost, bs.stream));
            HIPCHK(hipMemcpyAsync(h->h_pin + 9, bs.d_status, sizeof(int), hipMemcpyDeviceToHost, bs.stream));
        }
        mark(5);
        if (prof && pn < PCAP) ++pn;
        HIPCHK(hipStreamSynchronize(bs.stream));
        HIPCHK(hipGetLastError());
        double gmax;
        memcpy(&gmax, h->h_pin + 8, sizeof(double));
        int st = 0;
        memcpy(&st, h->h_pin + 9, sizeof(int));
        if (first) { push(0, 1, 1, cost, 0.0, 0.0, radius, 0.0, gmax); first = false; }
        // TrustRegionMinimizer::FinalizeIterationAndCheckIfMinimizerCanContinue: iterations, then gradient, then radius
        if (it > o.max_iter) { term = LVBA_TERM_NO_CONVERGENCE; break; }
        if (gmax <= o.gradient_tolerance) { term = LVBA_TERM_GRADIENT; break; }
        const double cand = 0.5 * h->h_pin[1], model = h->h_pin[2];
        const double step_norm = sqrt(h->h_pin[3]), x_norm = sqrt(h->h_pin[4]);
        const bool finite_step = st == 0 && isfinite(model) && isfinite(step_norm);
        if (!finite_step || !(model > 0.0)) { // LevenbergMarquardtStrategy::StepIsInvalid = StepRejected(0): /2, /4, /8 ...
            radius = radius / decrease_factor;
            decrease_factor *= 2.0;
            push(it, 0, 0, cost, 0.0, 0.0, radius, 0.0, gmax);
            if (++invalid_run >= 5) { term = LVBA_TERM_FAILURE; rc = fail(LVBA_NUM_FACTORIZATION, "5 consecutive invalid steps"); }
            if (radius < o.min_radius) { term = LVBA_TERM_RADIUS; break; }
            continue;
        }
        invalid_run = 0;
        if (step_norm <= o.parameter_tolerance * (x_norm + o.parameter_tolerance)) {
            push(it, 0, 1, cost, cost - cand, step_norm, radius, 0.0, gmax);
            term = LVBA_TERM_PARAMETER;
            break;
        }
        const double cost_change = cost - cand;
        if (fabs(cost_change) <= o.function_tolerance * cost) {
            push(it, 0, 1, cost, cost_change, step_norm, radius, 0.0, gmax);
            term = LVBA_TERM_FUNCTION;
            break;
        }
        const double rho = cost_change / model;
        if (isfinite(cand) && rho > o.min_relative_decrease) {
            // the accepted point becomes the current one: nothing to evaluate -- the kernels of the next iteration linearise from
            // the state itself (residuals and Jacobians are re-computed where they are used)
            std::swap(h->d_q, h->d_q2); std::swap(h->d_t, h->d_t2); std::swap(h->d_X, h->d_X2);
            cost = cand;
            radius = std::min(o.max_radius, radius / std::max(1.0 / 3.0, 1.0 - pow(2.0 * rho - 1.0, 3.0)));
            decrease_factor = 2.0;
            push(it, 1, 1, cost, cost_change, step_norm, radius, rho, gmax);
        } else {
            radius = radius / decrease_factor;
            decrease_factor *= 2.0;
            push(it, 0, 1, cand, cost_change, step_norm, radius, rho, gmax);
            if (radius < o.min_radius) { term = LVBA_TERM_RADIUS; break; }
        }
    }
    if (prof) {
        hipStreamSynchronize(bs.stream);
        double acc[NPH] = {0, 0, 0, 0, 0, 0};
        for (int k = 0; k < pn; ++k)
            for (int ph = 0; ph + 1 < NPH; ++ph) {
                float ms = 0.f;
                hipEventElapsedTime(&ms, pev[(size_t)k * NPH + ph], pev[(size_t)k * NPH + ph + 1]);
                acc[ph] += ms;
            }
        for (int k = 0; k + 1 < pn; ++k) { // end of iteration k -> start of iteration k + 1: host turn-around + the accepted point's linearisation
            float ms = 0.f;
            hipEventElapsedTime(&ms, pev[(size_t)k * NPH + NPH - 1], pev[(size_t)(k + 1) * NPH]);
            acc[NPH - 1] += ms;
        }
        if (pn > 0)
            fprintf(stderr, "[lvba visual profile] %d iterations, us each: reduced system %.1f | solve %.1f | back %.1f | step + trial cost %.1f | "
                            "copies %.1f | between iterations %.1f\n", pn, 1e3 * acc[0] / pn, 1e3 * acc[1] / pn, 1e3 * acc[2] / pn,
                    1e3 * acc[3] / pn, 1e3 * acc[4] / pn, pn > 1 ? 1e3 * acc[5] / (pn - 1) : 0.0);
        for (auto &e : pev) hipEventDestroy(e);
    }
    if (n_trace) *n_trace = std::min(rows, trace_cap > 0 ? trace_cap : 0);
    if (termination) *termination = term;
    const double tw3 = wall_us();
    const int32_t rc2 = export_state(h, q, t, X);
    if (prof)
        fprintf(stderr, "[lvba visual profile] host clock, us: state upload %.0f | first evaluation + Jacobi scaling %.0f | iterations %.0f | state download %.0f\n",
                tw1 - tw0, tw2 - tw1, tw3 - tw2, wall_us() - tw3);
    return rc != LVBA_OK ? rc : rc2;
}
