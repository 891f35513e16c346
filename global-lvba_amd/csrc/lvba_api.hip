// lvba_api.hip -- BALM half of the C-ABI (include/lvba_hip.h): problem packing and the Nielsen-LM driver of
// BALM2::damping_iter (reference include/BALM/bavoxel.hpp:662-767).  Host logic only; all arithmetic on problem
// data runs in the kernels of balm_kernels.hip / ldlt.hip.  Shared machinery (ordering, assembly lists, solver,
// RCCL) lives in block_system.hip.  No CPU fallback.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <new>
#include <vector>

#include "host_arena.h"
#include "block_system.h"
#include "host_tables.h"

using namespace lvba;

#define fail lvba_fail

// ------------------------------------------------------------------------------------------ handle
enum { EV_COST = 0, EV_EVAL, EV_SOLVE, EV_REDUCE, EV_COSTK, EV_EVALK, EV_N };

struct lvba_balm_s {
    BlockSys bs;
    int32_t N = 0;
    int64_t V = 0, F = 0, Q = 0, n_chunks = 0, Vglobal = 0;
    // host copies kept until finalize()
    lvba::hvec<int64_t> h_voff;
    lvba::hvec<int32_t> h_pidx;
    bool finalized = false;
    bool voxels_sorted = false; // the voxels were re-laid in the order of the first pose that sees them (balm_create_impl)
    // device data (voxel-major)
    int64_t *d_voff = nullptr, *d_chunk_v0 = nullptr;
    int32_t *d_pidx = nullptr;
    double *d_clu = nullptr, *d_chunk_cost = nullptr;
    // pose-major payload
    double *d_clu_csc = nullptr, *d_vrec = nullptr, *d_part = nullptr;
    double *d_pose_in = nullptr, *d_pose_cur = nullptr, *d_pose_trial = nullptr, *d_out = nullptr;
    double *d_scal2 = nullptr; // [0]=trial cost sum
    double *d_q1part = nullptr; // per-workgroup shares of the q1 numerator (retract_q1_kernel)
    double *h_pin = nullptr;   // pinned host staging, 16 doubles
    // grouped refinement (lvba_balm_set_groups): independent pose / voxel groups, one LM state each
    int32_t n_groups = 0;
    lvba::hvec<int32_t> g_pose_off;   // [n_groups + 1]
    lvba::hvec<int64_t> g_vox_off;    // [n_groups + 1]
    int32_t *d_grp_of_pose = nullptr, *d_gpo = nullptr, *d_gaccept = nullptr;
    int64_t *d_gco = nullptr;          // chunk range of every group
    double *d_gscal = nullptr;         // [3][n_groups]: cost at the current poses, cost at the trial poses, q1 numerator
    double *h_gpin = nullptr;          // pinned, [3][n_groups]
    int32_t *h_gacc = nullptr;         // pinned, [n_groups]
    // LM state (bavoxel.hpp:664-671)
    bool lm_active = false, lm_done = false, is_calc_hess = true;
    lvba_balm_opts lm_opts{};
    double u = 0.01, v = 2.0, residual1 = 0.0;
    int iter = 0;
    bool have_eval = false;
    // the voxel records (d_vrec) and chunk costs belong to the poses in d_pose_cur: the LM loop costs its trial point with the
    // voxel pass of the evaluation, and an accepted trial point is where the next evaluation happens
    bool lin_at_cur = false;
    // profiling
    bool prof_on = false;
    hipEvent_t ev[EV_N][2] = {};
    bool ev_used[EV_N] = {};
    lvba_prof_t prof{};

    hipStream_t stream() const { return bs.stream; }
    BalmDev dev() const
    {
        BalmDev d;
        d.n_poses = N; d.band_blocks = bs.Bb; d.V = V; d.F = F; d.n_chunks = n_chunks;
        d.voff = d_voff; d.pidx = d_pidx; d.clu = d_clu; d.chunk_v0 = d_chunk_v0;
        d.S = bs.S; d.csc_off = bs.d_csc_off; d.clu_csc = d_clu_csc; d.vox_of_pos = bs.d_group_of_pos; d.vrec = d_vrec;
        d.Y = bs.d_Y; d.part = d_part;
        return d;
    }
};

// ------------------------------------------------------------------------------------------ misc API
extern "C" int32_t lvba_version(void) { return 110; }
extern "C" int32_t lvba_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
extern "C" void lvba_balm_default_opts(lvba_balm_opts *o)
{
    if (!o) return;
    o->max_iter = 10; o->reserved = 0; o->u0 = 0.01; o->v0 = 2.0; o->rel_tol = 1e-6;
}
extern "C" void lvba_shard_range(int64_t V, int32_t rank, int32_t G, int64_t *head, int64_t *end)
{
    if (G < 1) G = 1;
    // exact integer floor(V*r/G); the reference's double arithmetic agrees for V < 2^53/G
    if (head) *head = (int64_t)(((__int128)V * rank) / G);
    if (end) *end = (int64_t)(((__int128)V * (rank + 1)) / G);
}

// ------------------------------------------------------------------------------------------ create
static int32_t balm_create_impl(int32_t n_poses, int64_t n_voxels, const int64_t *voxel_off, const int32_t *pose_idx,
                                const double *clusters, const double *d_clusters, int32_t device, lvba_balm_t *out, bool trusted = false);

extern "C" int32_t lvba_balm_create(int32_t n_poses, int64_t n_voxels, const int64_t *voxel_off,
                                    const int32_t *pose_idx, const double *clusters, int32_t device,
                                    lvba_balm_t *out)
{
    return balm_create_impl(n_poses, n_voxels, voxel_off, pose_idx, clusters, nullptr, device, out);
}

// (internal, window_ba.hip) the caller vouches for its arrays -- they come out of this library's own voxel maps: no range checks,
// no voxel re-layout pass (1.1 of the 7.3 ms the joint LM of bench.py's window leg took)
namespace lvba {
int32_t balm_create_dev_trusted(int32_t n_poses, int64_t n_voxels, const int64_t *voxel_off, const int32_t *pose_idx,
                                const double *d_clusters, int32_t device, lvba_balm_t *out)
{
    if (!d_clusters) return fail(LVBA_ERR_ARG, "d_clusters is NULL");
    return balm_create_impl(n_poses, n_voxels, voxel_off, pose_idx, nullptr, d_clusters, device, out, true);
}
} // namespace lvba

extern "C" int32_t lvba_balm_create_dev(int32_t n_poses, int64_t n_voxels, const int64_t *voxel_off, const int32_t *pose_idx,
                                        const double *d_clusters, int32_t device, lvba_balm_t *out)
{
    return balm_create_impl(n_poses, n_voxels, voxel_off, pose_idx, nullptr, d_clusters, device, out);
}

static int32_t balm_create_impl(int32_t n_poses, int64_t n_voxels, const int64_t *voxel_off, const int32_t *pose_idx,
                                const double *clusters, const double *d_clusters, int32_t device, lvba_balm_t *out, bool trusted)
{
    if (!out) return fail(LVBA_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (n_poses < 1 || n_voxels < 1 || !voxel_off || !pose_idx || (!clusters && !d_clusters))
        return fail(LVBA_ERR_ARG, "n_poses/n_voxels must be >= 1 and arrays non-NULL");
    const int64_t base = voxel_off[0];
    const int64_t F = voxel_off[n_voxels] - base;
    if (F < 2 * n_voxels) return fail(LVBA_ERR_ARG, "every voxel needs >= 2 factors (push_voxel, bavoxel.hpp:52)");
    if (F >= (int64_t)1 << 31) return fail(LVBA_ERR_UNSUPPORTED, "more than 2^31 factors per shard");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(LVBA_ERR_DEVICE, "no HIP device available (liblvba_hip has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(LVBA_ERR_ARG, "device %d out of range [0,%d)", device, ndev);

    const bool timing = timing_on("build");
    auto nowc = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tmark = nowc();
    auto mark = [&](const char *what) {
        if (!timing) return;
        const double t = nowc();
        fprintf(stderr, "[balm_create] %-14s %.3f ms\n", what, t - tmark);
        tmark = t;
    };
    // one pass over the voxels: >= 2 factors each, pose indices in range, and the first pose that sees each voxel (for the
    // re-layout decision below; only needed from the size on from which the pair lists are windowed)
    const bool want_key = !trusted && 18 * 8 * F > ((int64_t)24 << 20) && n_voxels > 1;
    lvba::hvec<int32_t> key;
    if (want_key) key.resize((size_t)n_voxels);
    double jump = 0.0;
    for (int64_t a = 0; a < (trusted ? 0 : n_voxels); ++a) { // (trusted: the window driver's own voxel maps, window_ba.hip)
        const int64_t f0 = voxel_off[a] - base, f1 = voxel_off[a + 1] - base;
        if (f1 - f0 < 2) return fail(LVBA_ERR_ARG, "voxel %lld has %lld factors (< 2)", (long long)a, (long long)(f1 - f0));
        int32_t lo = 0x7fffffff, hi = -1;
        for (int64_t f = f0; f < f1; ++f) { lo = std::min(lo, pose_idx[f]); hi = std::max(hi, pose_idx[f]); }
        if (lo < 0 || hi >= n_poses) {
            for (int64_t f = f0; f < f1; ++f)
                if (pose_idx[f] < 0 || pose_idx[f] >= n_poses) return fail(LVBA_ERR_ARG, "pose_idx[%lld] = %d out of range", (long long)f, pose_idx[f]);
        }
        if (want_key) {
            key[(size_t)a] = lo;
            if (a > 0) jump += std::abs((double)lo - (double)key[(size_t)a - 1]);
        }
    }

    // Voxel order.  Every pass over the voxels is laid out for voxels that come roughly in the order of the poses that see
    // them (a voxel map built along a trajectory does; the synthetic problems do): the pose gather of the voxel pass, the
    // voxel records the factor pass gathers and above all the windows of the pair lists draw on a compact slice of memory then.
    // The reference hands its voxels over in the iteration order of an unordered_map (src/lvba_system.cpp:254-262 -> tras_opt),
    // i.e. in no order at all: at C3 that costs 17 % of the evaluation (2.71 vs 2.32 ms, tools/gpu_shuffled.sh).  So a large
    // problem whose voxels jump about is re-laid internally, voxels sorted (stably) by the first pose that sees them.  Nothing
    // the caller gets back is indexed by voxel; sums over voxels change in the last bits only.
    lvba::hvec<int64_t> voff_s;
    lvba::hvec<int32_t> pidx_s, fmap;
    {
        const bool sort_voxels = want_key && jump / (double)(n_voxels - 1) > std::max(64.0, 0.125 * n_poses);
        if (sort_voxels) { // stable counting sort by the first pose
            lvba::hvec<int64_t> start((size_t)n_poses + 1, 0);
            for (int64_t a = 0; a < n_voxels; ++a) ++start[(size_t)key[(size_t)a] + 1];
            for (int32_t i = 0; i < n_poses; ++i) start[(size_t)i + 1] += start[(size_t)i];
            lvba::hvec<int64_t> order((size_t)n_voxels); // order[new] = old
            for (int64_t a = 0; a < n_voxels; ++a) order[(size_t)start[(size_t)key[(size_t)a]]++] = a;
            voff_s.resize((size_t)n_voxels + 1);
            pidx_s.resize((size_t)F);
            fmap.resize((size_t)F);
            voff_s[0] = 0;
            int64_t w = 0;
            for (int64_t a = 0; a < n_voxels; ++a) {
                const int64_t o = order[(size_t)a];
                for (int64_t f = voxel_off[o] - base; f < voxel_off[o + 1] - base; ++f, ++w) {
                    pidx_s[(size_t)w] = pose_idx[f];
                    fmap[(size_t)w] = (int32_t)f;
                }
                voff_s[(size_t)a + 1] = w;
            }
            voxel_off = voff_s.data();
            pose_idx = pidx_s.data();
        }
    }
    const int64_t base2 = voxel_off[0]; // 0 after a re-layout
    mark("checks + order");

    lvba_balm_s *h = new (std::nothrow) lvba_balm_s();
    if (!h) return fail(LVBA_ERR_NOMEM, "host allocation failed");
    h->N = n_poses; h->V = n_voxels; h->F = F; h->Vglobal = n_voxels;
    h->voxels_sorted = !fmap.empty();
    // validate + chunk
    h->h_voff.resize(n_voxels + 1);
    lvba::hvec<int64_t> chunk_v0;
    int64_t Q = 0;
    for (int64_t a = 0; a <= n_voxels; ++a) h->h_voff[a] = voxel_off[a] - base2;
    {
        const int64_t bad = lvba::chunk_voxels(n_voxels, voxel_off, LVBA_CF, LVBA_CV, chunk_v0, Q); // host_tables.h
        if (bad >= 0) {
            const long long k = (long long)(voxel_off[bad + 1] - voxel_off[bad]);
            delete h;
            return fail(LVBA_ERR_ARG, "voxel %lld has %lld factors (< 2)", (long long)bad, k);
        }
    }
    h->n_chunks = (int64_t)chunk_v0.size() - 1;
    h->Q = Q;
    h->h_pidx.assign(pose_idx, pose_idx + F);
    mark("chunks + copies");

    auto bail = [&](int32_t rc) { lvba_balm_destroy(h); return rc; };
#define CTRY(expr) do { int32_t rc_ = (expr); if (rc_ != LVBA_OK) return bail(rc_); } while (0)
#define CHIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return bail(fail(e_ == hipErrorOutOfMemory ? LVBA_ERR_NOMEM : LVBA_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_))); } while (0)
    CTRY(bs_init(h->bs, device));
    BlockSys &bs = h->bs;
    CTRY(bs_dmalloc(bs, &h->d_voff, n_voxels + 1));
    CTRY(bs_dmalloc(bs, &h->d_chunk_v0, h->n_chunks + 1));
    CTRY(bs_dmalloc(bs, &h->d_pidx, F));
    CTRY(bs_dmalloc(bs, &h->d_clu, 10 * F));
    CTRY(bs_dmalloc(bs, &h->d_chunk_cost, h->n_chunks));
    CHIP(lvba::copy_h2d(h->d_voff, h->h_voff.data(), (size_t)(n_voxels + 1) * sizeof(int64_t)));
    CHIP(lvba::copy_h2d(h->d_chunk_v0, chunk_v0.data(), (size_t)(h->n_chunks + 1) * sizeof(int64_t)));
    { // AoS [F][10] -> SoA [10][F] on the device (a host array is staged in a temporary device buffer), through the factor map of
      // a re-layout
        const double *src = d_clusters; // like the host array: indexed relative to voxel_off[0]
        double *d_stage = nullptr;
        int32_t *d_fmap = nullptr;
        if (!d_clusters) {
            CTRY(bs_dmalloc(bs, &d_stage, 10 * F));
            CHIP(lvba::copy_h2d(d_stage, clusters, (size_t)(10 * F) * sizeof(double)));
            src = d_stage;
        }
        if (!fmap.empty()) {
            CTRY(bs_dmalloc(bs, &d_fmap, F));
            CHIP(lvba::copy_h2d(d_fmap, fmap.data(), (size_t)F * sizeof(int32_t)));
        }
        launch_aos_to_soa(src, d_fmap, F, h->d_clu, bs.stream);
        CHIP(hipGetLastError());
        CHIP(hipStreamSynchronize(bs.stream));
        if (d_stage) { lvba::DevicePool::get().free(d_stage); bs.device_bytes -= (int64_t)(10 * F) * (int64_t)sizeof(double); }
        if (d_fmap) { lvba::DevicePool::get().free(d_fmap); bs.device_bytes -= (int64_t)F * (int64_t)sizeof(int32_t); }
    }
    mark("device arrays");
    CHIP(lvba::PinnedCache::get().acquire((void **)&h->h_pin, 4096)); // (16 doubles; one size for the cache's sake)
    for (int e = 0; e < EV_N; ++e)
        for (int s = 0; s < 2; ++s) CHIP(hipEventCreate(&h->ev[e][s]));
    mark("pinned + events");
#undef CTRY
#undef CHIP
    *out = h;
    return LVBA_OK;
}

extern "C" int32_t lvba_balm_destroy(lvba_balm_t h)
{
    if (!h) return LVBA_OK;
    hipSetDevice(h->bs.device);
    if (h->bs.stream) hipStreamSynchronize(h->bs.stream);
    void *ptrs[] = {h->d_voff, h->d_chunk_v0, h->d_pidx, h->d_clu, h->d_chunk_cost, h->d_clu_csc, h->d_vrec, h->d_part,
                    h->d_pose_in, h->d_pose_cur, h->d_pose_trial, h->d_out, h->d_scal2, h->d_q1part, h->d_grp_of_pose, h->d_gpo, h->d_gaccept, h->d_gco,
                    h->d_gscal};
    for (void *p : ptrs)
        if (p) lvba::DevicePool::get().free(p);
    lvba::PinnedCache::get().release(h->h_pin);
    lvba::PinnedCache::get().release(h->h_gpin);
    lvba::PinnedCache::get().release(h->h_gacc);
    for (int e = 0; e < EV_N; ++e)
        for (int s = 0; s < 2; ++s)
            if (h->ev[e][s]) hipEventDestroy(h->ev[e][s]);
    bs_destroy(h->bs);
    delete h;
    return LVBA_OK;
}

extern "C" int32_t lvba_balm_configure(lvba_balm_t h, int32_t ordering, double band_frac)
{
    if (!h) return fail(LVBA_ERR_ARG, "handle is NULL");
    if (h->finalized) return fail(LVBA_ERR_STATE, "configure must precede the first cost/eval/refine call");
    if (ordering != 0 && ordering != 1) return fail(LVBA_ERR_ARG, "ordering must be 0 or 1");
    if (!(band_frac >= 0.0)) return fail(LVBA_ERR_ARG, "band_frac must be >= 0");
    // a grouped problem keeps the caller's pose order: the per-group tables (damping, q1, accept / select) are indexed by it
    if (h->n_groups > 0 && ordering != 0) return fail(LVBA_ERR_STATE, "a grouped problem (lvba_balm_set_groups) keeps ordering 0");
    h->bs.ordering = ordering;
    h->bs.band_frac = band_frac;
    return LVBA_OK;
}

static int32_t finalize(lvba_balm_s *h)
{
    if (h->finalized) return LVBA_OK;
    BlockSys &bs = h->bs;
    HIPCHK(hipSetDevice(bs.device));
    const int N = h->N;
    if (h->n_groups > 0) bs.ordering = 0; // (lvba_balm_configure / dist_init refuse to undo it; kept here as the single point of truth)
    const bool timing = timing_on("build");
    auto nowc = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tmark = nowc();
    auto mark = [&](const char *what) {
        if (!timing) return;
        const double t = nowc();
        fprintf(stderr, "[finalize] %-14s %.3f ms\n", what, t - tmark);
        tmark = t;
    };
    TRY(bs_build(bs, N, h->V, h->h_voff.data(), h->h_pidx.data()));
    { // experiment (round 4): fp32 Y records between the factor and the pair pass -- the column kernel's lists only
        const char *e = getenv("LVBA_Y32");
        bs.y32 = bs.pair_col && e && !strcmp(e, "1");
    }
    mark("bs_build");
    lvba::hvec<int32_t> p((size_t)h->F); // pose indices of the factors in solver order
    for (int64_t f = 0; f < h->F; ++f) p[f] = bs.iperm[h->h_pidx[f]];
    HIPCHK(lvba::copy_h2d(h->d_pidx, p.data(), (size_t)h->F * sizeof(int32_t)));
    mark("pose indices");
    TRY(bs_dmalloc(bs, &h->d_clu_csc, 10 * h->F));
    TRY(bs_dmalloc(bs, &h->d_vrec, 16 * h->V));
    TRY(bs_dmalloc(bs, &h->d_part, (int64_t)N * bs.S * 32));
    launch_gather_csc(h->d_clu, bs.d_csc_f, h->F, h->d_clu_csc, bs.stream);
    TRY(bs_dmalloc(bs, &h->d_pose_in, 12 * (int64_t)N));
    TRY(bs_dmalloc(bs, &h->d_pose_cur, 12 * (int64_t)N));
    TRY(bs_dmalloc(bs, &h->d_pose_trial, 12 * (int64_t)N));
    TRY(bs_dmalloc(bs, &h->d_out, 12 * (int64_t)N));
    TRY(bs_dmalloc(bs, &h->d_scal2, 8));
    TRY(bs_dmalloc(bs, &h->d_q1part, (N + 127) / 128 + 8));
    HIPCHK(hipStreamSynchronize(bs.stream));
    mark("pose-major copy");
    lvba::hvec<int64_t>().swap(h->h_voff);
    lvba::hvec<int32_t>().swap(h->h_pidx);
    h->finalized = true;
    return LVBA_OK;
}

extern "C" int32_t lvba_balm_info(lvba_balm_t h, lvba_balm_info_t *info)
{
    if (!h || !info) return fail(LVBA_ERR_ARG, "NULL argument");
    TRY(finalize(h));
    info->n_poses = h->N; info->n_ranks = h->bs.n_ranks; info->n_voxels = h->V; info->n_voxels_global = h->Vglobal;
    info->n_factors = h->F; info->n_pairs = h->Q; info->n_chunks = h->n_chunks; info->n_blocks = h->bs.nnzb;
    info->band_blocks = h->bs.Bb; info->use_band = h->bs.use_band ? 1 : 0; info->hess_bytes = h->bs.hblk_doubles * 8;
    info->device_bytes = h->bs.device_bytes;
    info->twist_panels = (h->bs.d_bcr || h->bs.nd.active) ? 0 : (int32_t)ldlt_twist_panels(h->bs.A.n, h->bs.A.ld, h->bs.A.bw);
    info->nd_kind = !h->bs.nd.active ? 0 : !strcmp(h->bs.nd.kind, "hubs") ? 1 : 2;
    info->nd_arcs = (int32_t)h->bs.nd.arcs.size(); info->nd_sep_poses = h->bs.nd.Ns; info->nd_sep_band_blocks = h->bs.nd.BbS;
    info->nd_model_band_ms = 1e3 * h->bs.nd.t_band; info->nd_model_nd_ms = 1e3 * h->bs.nd.t_nd;
    info->solve_ranks = !(h->bs.distributed() && h->bs.n_ranks >= 2) ? 1 : h->bs.nd.active ? h->bs.n_ranks : info->twist_panels > 0 ? 2 : 1;
    info->trial_linearised = 1;
    info->y_fp32 = h->bs.y32 ? 1 : 0;
    info->allreduce_bytes = !h->bs.distributed() ? 0 : 8 * ((h->bs.d_ar_slot ? 36 * h->bs.n_ar : h->bs.hblk_doubles) + 6 * (int64_t)h->N + 1);
    return LVBA_OK;
}

extern "C" int32_t lvba_balm_get_ordering(lvba_balm_t h, int32_t *perm)
{
    if (!h || !perm) return fail(LVBA_ERR_ARG, "NULL argument");
    TRY(finalize(h));
    memcpy(perm, h->bs.perm.data(), (size_t)h->N * sizeof(int32_t));
    return LVBA_OK;
}

extern "C" int32_t lvba_balm_nd_model(lvba_balm_t h, int32_t n_ranks, lvba_nd_model_t *out)
{
    if (!h || !out || n_ranks < 1) return fail(LVBA_ERR_ARG, "bad argument");
    TRY(finalize(h));
    memset(out, 0, sizeof *out);
    out->n_ranks = n_ranks;
    double tb = 0.0, tn = 0.0;
    const int32_t rc = bs_nd_model(h->bs, n_ranks, &tb, &tn, &out->arcs, &out->sep_poses, &out->sep_band_blocks, &out->max_arc_poses,
                                   &out->max_arc_band_blocks);
    if (rc != LVBA_OK) return fail(rc, "no co-visibility graph was kept for this problem (fewer than 256 poses, grouped, or solved dense without an ordering)");
    out->band_ms = 1e3 * tb; out->nd_ms = 1e3 * tn;
    return LVBA_OK;
}

// ------------------------------------------------------------------------------------------ profiling
static void ev_begin(lvba_balm_s *h, int which)
{
    if (h->prof_on) hipEventRecord(h->ev[which][0], h->stream());
}
static void ev_end(lvba_balm_s *h, int which)
{
    if (h->prof_on) { hipEventRecord(h->ev[which][1], h->stream()); h->ev_used[which] = true; }
}
static void ev_collect(lvba_balm_s *h) // call after a stream synchronize
{
    if (!h->prof_on) return;
    double *acc[EV_N] = {&h->prof.cost_ms, &h->prof.eval_ms, &h->prof.solve_ms, &h->prof.reduce_ms,
                         &h->prof.cost_kernel_ms, &h->prof.eval_kernel_ms};
    int64_t *cnt[EV_N] = {&h->prof.cost_calls, &h->prof.eval_calls, &h->prof.solve_calls, &h->prof.reduce_calls, nullptr, nullptr};
    for (int e = 0; e < EV_N; ++e)
        if (h->ev_used[e]) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, h->ev[e][0], h->ev[e][1]) == hipSuccess) {
                *acc[e] += ms;
                if (cnt[e]) *cnt[e] += 1;
            }
            h->ev_used[e] = false;
        }
}
extern "C" int32_t lvba_balm_set_profiling(lvba_balm_t h, int32_t enable)
{
    if (!h) return fail(LVBA_ERR_ARG, "handle is NULL");
    h->prof_on = enable != 0;
    return LVBA_OK;
}
extern "C" int32_t lvba_balm_get_profile(lvba_balm_t h, lvba_prof_t *out, int32_t reset)
{
    if (!h || !out) return fail(LVBA_ERR_ARG, "NULL argument");
    *out = h->prof;
    if (reset) memset(&h->prof, 0, sizeof h->prof);
    return LVBA_OK;
}

// ------------------------------------------------------------------------------------------ stages
// enqueue: cost at device poses (solver order) -> dst[0] = (global) sum of lambda_min
// with_lin: by the voxel pass of the evaluation instead of the cost-only kernel -- the same chunk costs, plus the voxel records
// at `poses`, which enqueue_eval can start from
static int32_t enqueue_cost(lvba_balm_s *h, const double *d_poses, double *dst, bool with_lin = false)
{
    ev_begin(h, EV_COST);
    launch_cost(h->dev(), d_poses, h->d_chunk_cost, dst, h->stream(), h->prof_on ? h->ev[EV_COSTK][0] : nullptr,
                h->prof_on ? h->ev[EV_COSTK][1] : nullptr, with_lin);
    if (h->prof_on) h->ev_used[EV_COSTK] = true;
    ev_end(h, EV_COST);
    if (h->bs.distributed()) {
        ev_begin(h, EV_REDUCE);
        TRY(bs_allreduce(h->bs, dst, 1));
        ev_end(h, EV_REDUCE);
    }
    HIPCHK(hipGetLastError());
    return LVBA_OK;
}

// enqueue: H, g, cost at device poses -> bs.d_hg (all-reduced over ranks)
// lin_in_place: the voxel records and chunk costs at d_poses are already there (enqueue_cost with_lin at the same poses)
static int32_t enqueue_eval(lvba_balm_s *h, const double *d_poses, bool lin_in_place = false)
{
    BlockSys &bs = h->bs;
    ev_begin(h, EV_EVAL);
    TRY(bs_clear_reduced(bs));
    launch_eval(h->dev(), bs.pair_dev(), d_poses, bs.Hblk(), bs.hblk_doubles, bs.g(), h->d_chunk_cost, bs.scal(),
                false, bs.stream, h->prof_on ? h->ev[EV_EVALK][0] : nullptr,
                h->prof_on ? h->ev[EV_EVALK][1] : nullptr, lin_in_place);
    if (h->prof_on) h->ev_used[EV_EVALK] = true;
    ev_end(h, EV_EVAL);
    if (bs.distributed()) {
        ev_begin(h, EV_REDUCE);
        TRY(bs_allreduce_hg(bs));
        ev_end(h, EV_REDUCE);
    }
    HIPCHK(hipGetLastError());
    h->have_eval = true;
    return LVBA_OK;
}

static int32_t enqueue_solve(lvba_balm_s *h, double u)
{
    ev_begin(h, EV_SOLVE);
    TRY(bs_enqueue_solve(h->bs, u));
    ev_end(h, EV_SOLVE);
    return LVBA_OK;
}

// poses of the solver's order on the device -> caller's order on the host
static int32_t download_poses(lvba_balm_s *h, const double *d_src, double *poses_out)
{
    const size_t bytes = (size_t)12 * h->N * sizeof(double);
    if (bytes <= lvba::HostStage::kBytes) { // zero-copy, like upload_poses
        if (void *st = lvba::HostStage::get().lock(bytes)) {
            launch_export_poses(d_src, h->bs.d_perm, h->N, static_cast<double *>(st), h->stream());
            const hipError_t e = hipStreamSynchronize(h->stream());
            if (e == hipSuccess) memcpy(poses_out, st, bytes);
            lvba::HostStage::get().unlock(st);
            HIPCHK(e);
            return LVBA_OK;
        }
    }
    launch_export_poses(d_src, h->bs.d_perm, h->N, h->d_out, h->stream());
    HIPCHK(hipStreamSynchronize(h->stream()));
    HIPCHK(lvba::copy_d2h(poses_out, h->d_out, bytes));
    return LVBA_OK;
}

static int32_t upload_poses(lvba_balm_s *h, const double *poses, double *d_dst)
{
    const size_t bytes = (size_t)12 * h->N * sizeof(double);
    if (bytes <= lvba::HostStage::kBytes) { // zero-copy through the process-wide pinned stage (mempool.h)
        if (void *st = lvba::HostStage::get().lock(bytes)) {
            memcpy(st, poses, bytes);
            launch_import_poses(static_cast<const double *>(st), h->bs.d_perm, h->N, d_dst, h->stream());
            const hipError_t e = hipStreamSynchronize(h->stream()); // the kernel has read the stage
            lvba::HostStage::get().unlock(st);
            HIPCHK(e);
            return LVBA_OK;
        }
    }
    HIPCHK(hipStreamSynchronize(h->stream())); // d_pose_in may still be read by an earlier import
    HIPCHK(lvba::copy_h2d(h->d_pose_in, poses, bytes));
    launch_import_poses(h->d_pose_in, h->bs.d_perm, h->N, d_dst, h->stream());
    return LVBA_OK;
}

// ------------------------------------------------------------------------------------------ a7 / a6
extern "C" int32_t lvba_balm_cost(lvba_balm_t h, const double *poses, int32_t is_avg, double *cost)
{
    if (!h || !poses || !cost) return fail(LVBA_ERR_ARG, "NULL argument");
    TRY(finalize(h));
    HIPCHK(hipSetDevice(h->bs.device));
    TRY(upload_poses(h, poses, h->d_pose_trial));
    h->lin_at_cur = false; // the chunk costs are overwritten
    TRY(enqueue_cost(h, h->d_pose_trial, h->d_scal2));
    HIPCHK(hipMemcpyAsync(h->h_pin, h->d_scal2, sizeof(double), hipMemcpyDeviceToHost, h->stream()));
    HIPCHK(hipStreamSynchronize(h->stream()));
    ev_collect(h);
    *cost = is_avg ? h->h_pin[0] / (double)h->Vglobal : h->h_pin[0];
    if (!isfinite(*cost)) return fail(LVBA_NUM_NONFINITE, "non-finite cost");
    return LVBA_OK;
}

extern "C" int32_t lvba_balm_eval(lvba_balm_t h, const double *poses, double *H, double *g, double *cost_avg)
{
    if (!h || !poses) return fail(LVBA_ERR_ARG, "NULL argument");
    TRY(finalize(h));
    BlockSys &bs = h->bs;
    HIPCHK(hipSetDevice(bs.device));
    TRY(upload_poses(h, poses, h->d_pose_cur));
    h->lin_at_cur = false; // (a full evaluation; an LM loop in progress starts its next evaluation from scratch too)
    TRY(enqueue_eval(h, h->d_pose_cur));
    HIPCHK(hipMemcpyAsync(h->h_pin, bs.scal(), sizeof(double), hipMemcpyDeviceToHost, bs.stream));
    const int64_t n = 6 * (int64_t)h->N;
    if (g) {
        launch_export_vec(bs.g(), bs.d_perm, h->N, h->d_out, bs.stream);
        HIPCHK(hipMemcpyAsync(g, h->d_out, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, bs.stream));
    }
    DevBuf dH(bs.stream); // freed on every path, error returns included
    if (H) {
        HIPCHK(dH.alloc((size_t)(n * n) * sizeof(double)));
        launch_export_dense(bs.Hblk(), bs.Bb, h->N, bs.d_perm, dH.as<double>(), bs.stream);
        HIPCHK(hipMemcpyAsync(H, dH.as<double>(), (size_t)(n * n) * sizeof(double), hipMemcpyDeviceToHost, bs.stream));
    }
    HIPCHK(hipStreamSynchronize(bs.stream));
    ev_collect(h);
    if (cost_avg) *cost_avg = h->h_pin[0] / (double)h->Vglobal;
    return LVBA_OK;
}

// H of an evaluation in sparse form (the dense matrix of lvba_balm_eval is 28.8 GB at 10 000 poses): see include/lvba_hip.h
extern "C" int32_t lvba_balm_eval_blocks(lvba_balm_t h, const double *poses, int64_t capacity, int32_t *bi, int32_t *bj,
                                         double *blocks, int64_t *n_blocks, double *g, double *cost_avg)
{
    if (!h || !poses || !n_blocks) return fail(LVBA_ERR_ARG, "NULL argument");
    if (capacity > 0 && (!bi || !bj || !blocks)) return fail(LVBA_ERR_ARG, "NULL block arrays");
    TRY(finalize(h));
    BlockSys &bs = h->bs;
    HIPCHK(hipSetDevice(bs.device));
    // The block set is STRUCTURAL (the pair lists' destinations + the diagonal; the union pattern in a multi-rank job): it is
    // known from the set-up, the same for every call, and a sizing call (capacity 0, no g, no cost) runs no evaluation.
    lvba::hvec<int64_t> slots;
    TRY(bs_pattern_slots(bs, slots));
    const int64_t nb = (int64_t)slots.size();
    *n_blocks = nb;
    if (capacity <= 0 && !g && !cost_avg) return LVBA_OK;
    if (capacity > 0 && nb > capacity) return fail(LVBA_ERR_ARG, "capacity %lld < %lld blocks", (long long)capacity, (long long)nb);
    TRY(upload_poses(h, poses, h->d_pose_cur));
    h->lin_at_cur = false;
    TRY(enqueue_eval(h, h->d_pose_cur));
    HIPCHK(hipMemcpyAsync(h->h_pin, bs.scal(), sizeof(double), hipMemcpyDeviceToHost, bs.stream));
    if (g) {
        launch_export_vec(bs.g(), bs.d_perm, h->N, h->d_out, bs.stream);
        HIPCHK(hipMemcpyAsync(g, h->d_out, (size_t)6 * h->N * sizeof(double), hipMemcpyDeviceToHost, bs.stream));
    }
    HIPCHK(hipStreamSynchronize(bs.stream));
    ev_collect(h);
    if (cost_avg) *cost_avg = h->h_pin[0] / (double)h->Vglobal;
    if (capacity <= 0) return LVBA_OK;
    // one device-side gather of the pattern's blocks, downloaded in slabs; then into the caller's pose order, in place
    TRY(bs_download_blocks(bs, slots.data(), nb, blocks));
    const int64_t Bb1 = (int64_t)bs.Bb + 1;
    for (int64_t q = 0; q < nb; ++q) {
        const int64_t J = slots[(size_t)q] / Bb1, dI = slots[(size_t)q] - J * Bb1;
        double *o = blocks + q * 36, b[36];
        memcpy(b, o, sizeof b); // column-major: b[c * 6 + r] = H(6 I + r, 6 J + c), I = J + dI (diagonal: r >= c only)
        const int32_t pi = bs.perm[(size_t)(J + dI)], pj = bs.perm[(size_t)J];
        if (dI == 0) {
            bi[q] = pi; bj[q] = pj;
            for (int r = 0; r < 6; ++r)
                for (int c = 0; c < 6; ++c) o[6 * r + c] = r >= c ? b[c * 6 + r] : b[r * 6 + c];
        } else if (pi >= pj) {
            bi[q] = pi; bj[q] = pj;
            for (int r = 0; r < 6; ++r)
                for (int c = 0; c < 6; ++c) o[6 * r + c] = b[c * 6 + r];
        } else { // the caller's order flips the pair: the transposed block
            bi[q] = pj; bj[q] = pi;
            for (int r = 0; r < 6; ++r)
                for (int c = 0; c < 6; ++c) o[6 * r + c] = b[r * 6 + c];
        }
    }
    return LVBA_OK;
}

extern "C" int32_t lvba_balm_solve(lvba_balm_t h, double u, double *dx)
{
    if (!h || !dx) return fail(LVBA_ERR_ARG, "NULL argument");
    if (!h->finalized || !h->have_eval) return fail(LVBA_ERR_STATE, "lvba_balm_solve needs a prior lvba_balm_eval");
    BlockSys &bs = h->bs;
    HIPCHK(hipSetDevice(bs.device));
    TRY(enqueue_solve(h, u));
    launch_export_vec(bs.d_dx, bs.d_perm, h->N, h->d_out, bs.stream);
    HIPCHK(hipMemcpyAsync(dx, h->d_out, (size_t)6 * h->N * sizeof(double), hipMemcpyDeviceToHost, bs.stream));
    int st = 0;
    HIPCHK(hipMemcpyAsync(&st, bs.d_status, sizeof(int), hipMemcpyDeviceToHost, bs.stream));
    HIPCHK(hipStreamSynchronize(bs.stream));
    ev_collect(h);
    if (st) return fail(LVBA_NUM_FACTORIZATION, "zero or non-finite pivot in LDL^T");
    return LVBA_OK;
}

// ------------------------------------------------------------------------------------------ a8: LM
extern "C" int32_t lvba_balm_lm_begin(lvba_balm_t h, const double *poses, const lvba_balm_opts *opts)
{
    if (!h || !poses) return fail(LVBA_ERR_ARG, "NULL argument");
    TRY(finalize(h));
    HIPCHK(hipSetDevice(h->bs.device));
    if (opts) h->lm_opts = *opts; else lvba_balm_default_opts(&h->lm_opts);
    if (h->lm_opts.max_iter < 0) return fail(LVBA_ERR_ARG, "max_iter < 0");
    TRY(upload_poses(h, poses, h->d_pose_cur));
    h->u = h->lm_opts.u0; h->v = h->lm_opts.v0;
    h->is_calc_hess = true; h->iter = 0; h->residual1 = 0.0;
    h->lin_at_cur = false;
    h->lm_active = true;
    h->lm_done = h->lm_opts.max_iter == 0;
    return LVBA_OK;
}

// One trip through bavoxel.hpp:686-766.
extern "C" int32_t lvba_balm_lm_step(lvba_balm_t h, lvba_lm_trace *row, int32_t *done)
{
    if (!h) return fail(LVBA_ERR_ARG, "handle is NULL");
    if (!h->lm_active) return fail(LVBA_ERR_STATE, "lm_step without lm_begin");
    if (h->lm_done) { if (done) *done = 1; return fail(LVBA_ERR_STATE, "LM loop already finished"); }
    HIPCHK(hipSetDevice(h->bs.device));
    const bool evaluated = h->is_calc_hess;
    // The trial point is costed by the VOXEL PASS of the evaluation (cost + voxel records): an accepted trial point is where the
    // next evaluation happens, and that evaluation then starts from the records (factor pass, pair pass) instead of reading
    // and eigen-decomposing every voxel again.  A rejected step wastes the difference to the cost-only kernel (C3: 0.04 ms).
    // Also on voxel shards (the records are local).
    const bool with_lin = true;
    if (evaluated) TRY(enqueue_eval(h, h->d_pose_cur, with_lin && h->lin_at_cur));         // :688-689
    TRY(enqueue_solve(h, h->u));                                                           // :692-710
    const int n_q1 = launch_retract_q1(h->d_pose_cur, h->bs.d_dx, h->d_pose_trial, h->N, h->bs.Hblk(), h->bs.Bb, h->bs.g(), h->u, h->d_q1part,
                                       h->stream());                                       // :722-729 (retraction + the q1 numerator's shares)
    TRY(enqueue_cost(h, h->d_pose_trial, h->d_scal2, with_lin));                           // :731 (+ the linearisation at the trial point)
    h->lin_at_cur = false; // it belongs to the trial point now
    launch_lm_report(h->d_scal2, h->d_q1part, n_q1, h->bs.scal(), h->bs.d_status, h->h_pin, h->stream()); // -> pinned host memory, zero-copy
    HIPCHK(hipStreamSynchronize(h->stream()));
    ev_collect(h);
    const double Vg = (double)h->Vglobal;
    if (evaluated) h->residual1 = h->h_pin[2] / Vg;                                        // AVG_THR :634-635
    const double residual1 = h->residual1;
    const double residual2 = h->h_pin[0] / Vg;
    const double q1 = h->h_pin[1] / Vg;                                                    // :732
    int st = 0;
    { long long stl; memcpy(&stl, h->h_pin + 4, sizeof stl); st = (int)stl; }
    double q = residual1 - residual2;                                                      // :736
    int32_t status = LVBA_OK;
    if (st) status = LVBA_NUM_FACTORIZATION;
    else if (!isfinite(residual2) || !isfinite(residual1)) status = LVBA_NUM_NONFINITE;
    // The reference checks neither the LDLT's info() nor the cost (:706-710, :731): a broken factorisation yields a non-finite
    // step, hence a NaN residual2, `q > 0` is false, the step is rejected, u *= v, and the loop goes on with more damping
    // (it can recover).  Same here: a flagged solve counts as a rejected step whatever the kernels left in dx.
    if (st) q = NAN;
    if (row) {
        row->iter = h->iter; row->accepted = q > 0; row->evaluated = evaluated; row->status = status;
        row->residual1 = residual1; row->residual2 = residual2; row->u = h->u; row->v = h->v; row->q = q; row->q1 = q1;
    }
    if (q > 0) {                                                                           // :744-752
        std::swap(h->d_pose_cur, h->d_pose_trial);
        h->lin_at_cur = with_lin; // the trial point is the current point now
        q = q / q1;
        h->v = 2.0;
        q = 1.0 - pow(2.0 * q - 1.0, 3.0);
        h->u *= (q < (1.0 / 3.0) ? (1.0 / 3.0) : q);
        h->is_calc_hess = true;
    } else {                                                                               // :753-758
        h->u = h->u * h->v;
        h->v = 2.0 * h->v;
        h->is_calc_hess = false;
    }
    h->iter += 1;
    if (fabs(residual1 - residual2) / residual1 < h->lm_opts.rel_tol) h->lm_done = true;   // :760
    if (h->iter >= h->lm_opts.max_iter) h->lm_done = true;                                 // :686
    if (done) *done = h->lm_done ? 1 : 0;
    // numerical statuses (> 0) are reported, not fatal: the caller may keep stepping, as BALM2::damping_iter does
    if (status == LVBA_NUM_FACTORIZATION) return fail(status, "zero or non-finite pivot in LDL^T (iteration %d)", h->iter - 1);
    if (status == LVBA_NUM_NONFINITE) return fail(status, "non-finite cost (iteration %d)", h->iter - 1);
    return LVBA_OK;
}

extern "C" int32_t lvba_balm_lm_end(lvba_balm_t h, double *poses_out)
{
    if (!h) return fail(LVBA_ERR_ARG, "handle is NULL");
    if (!h->lm_active) return fail(LVBA_ERR_STATE, "lm_end without lm_begin");
    HIPCHK(hipSetDevice(h->bs.device));
    if (poses_out) {
        TRY(download_poses(h, h->d_pose_cur, poses_out));
    }
    h->lm_active = false;
    return LVBA_OK;
}

extern "C" int32_t lvba_balm_refine(lvba_balm_t h, double *poses_inout, const lvba_balm_opts *opts,
                                    lvba_lm_trace *trace, int32_t *n_trace)
{
    if (!h || !poses_inout) return fail(LVBA_ERR_ARG, "NULL argument");
    if (n_trace) *n_trace = 0;
    TRY(lvba_balm_lm_begin(h, poses_inout, opts));
    int32_t rows = 0, done = h->lm_done ? 1 : 0, rc = LVBA_OK, worst = LVBA_OK;
    char worst_msg[512] = "";
    while (!done) {
        lvba_lm_trace row;
        rc = lvba_balm_lm_step(h, &row, &done);
        if (rc < 0) break;
        if (trace) trace[rows] = row;
        rows++;
        // a numerical failure (zero pivot, non-finite cost) does NOT end the loop: BALM2::damping_iter (bavoxel.hpp:686-766)
        // has no such exit -- the step is rejected, u *= v, and up to 10 iterations run.  The status is kept in the trace row
        // and the worst one is returned at the end.
        if (rc > worst) { worst = rc; snprintf(worst_msg, sizeof worst_msg, "%s", lvba_last_error()); }
    }
    if (n_trace) *n_trace = rows;
    const int32_t rc2 = lvba_balm_lm_end(h, poses_inout);
    if (rc < 0) return rc;
    if (rc2 != LVBA_OK) return rc2;
    if (worst != LVBA_OK) return fail(worst, "%s", worst_msg);
    return LVBA_OK;
}

// ------------------------------------------------------------------------------------------ grouped LM
// Several INDEPENDENT problems in one handle: group k owns the poses [pose_off[k], pose_off[k+1]) and the voxels
// [voxel_off[k], voxel_off[k+1]), and every factor of its voxels is seen from one of its poses -- the windows of
// LvbaSystem::runWindowBA (src/lvba_system.cpp:232-302), which the reference optimises one after the other.  The Hessian is
// block diagonal, so ONE evaluation, ONE band factorisation (damping per group) and ONE cost pass serve all groups per LM
// iteration; every group keeps the LM state of BALM2::damping_iter (bavoxel.hpp:662-767) for itself.
extern "C" int32_t lvba_balm_set_groups(lvba_balm_t h, int32_t n_groups, const int32_t *pose_off, const int64_t *voxel_off)
{
    if (!h || !pose_off || !voxel_off) return fail(LVBA_ERR_ARG, "NULL argument");
    if (h->finalized) return fail(LVBA_ERR_STATE, "set_groups must precede the first cost/eval/refine call");
    if (h->bs.distributed()) return fail(LVBA_ERR_UNSUPPORTED, "groups and voxel shards do not combine");
    if (n_groups < 1) return fail(LVBA_ERR_ARG, "n_groups must be >= 1");
    if (pose_off[0] != 0 || pose_off[n_groups] != h->N || voxel_off[0] != 0 || voxel_off[n_groups] != h->V)
        return fail(LVBA_ERR_ARG, "the groups must cover all poses and all voxels");
    for (int32_t k = 0; k < n_groups; ++k) {
        if (pose_off[k + 1] <= pose_off[k] || voxel_off[k + 1] <= voxel_off[k]) return fail(LVBA_ERR_ARG, "group %d is empty", k);
        for (int64_t f = h->h_voff[voxel_off[k]]; f < h->h_voff[voxel_off[k + 1]]; ++f)
            if (h->h_pidx[f] < pose_off[k] || h->h_pidx[f] >= pose_off[k + 1])
                return fail(LVBA_ERR_ARG, "a voxel of group %d is seen from pose %d, which belongs to another group", k, h->h_pidx[f]);
    }
    HIPCHK(hipSetDevice(h->bs.device));
    BlockSys &bs = h->bs;
    // chunks must not straddle groups: the chunk table is made again with breaks at the group boundaries
    lvba::hvec<int64_t> chunk_v0;
    int64_t Q = 0;
    if (lvba::chunk_voxels(h->V, h->h_voff.data(), LVBA_CF, LVBA_CV, chunk_v0, Q, voxel_off + 1, n_groups - 1) >= 0)
        return fail(LVBA_ERR_STATE, "chunk table");
    lvba::DevicePool::get().free(h->d_chunk_v0); bs.device_bytes -= (h->n_chunks + 1) * (int64_t)sizeof(int64_t); h->d_chunk_v0 = nullptr;
    lvba::DevicePool::get().free(h->d_chunk_cost); bs.device_bytes -= h->n_chunks * (int64_t)sizeof(double); h->d_chunk_cost = nullptr;
    h->n_chunks = (int64_t)chunk_v0.size() - 1;
    TRY(bs_dmalloc(bs, &h->d_chunk_v0, h->n_chunks + 1));
    TRY(bs_dmalloc(bs, &h->d_chunk_cost, h->n_chunks));
    HIPCHK(lvba::copy_h2d(h->d_chunk_v0, chunk_v0.data(), (size_t)(h->n_chunks + 1) * sizeof(int64_t)));
    lvba::hvec<int64_t> gco((size_t)n_groups + 1, 0);
    {
        int64_t c = 0;
        for (int32_t k = 0; k <= n_groups; ++k) {
            while (c < h->n_chunks && chunk_v0[(size_t)c] < voxel_off[k]) ++c;
            if (chunk_v0[(size_t)c] != voxel_off[k]) return fail(LVBA_ERR_STATE, "group %d does not start on a chunk", k);
            gco[(size_t)k] = c;
        }
    }
    lvba::hvec<int32_t> gof((size_t)h->N);
    for (int32_t k = 0; k < n_groups; ++k)
        for (int32_t j = pose_off[k]; j < pose_off[k + 1]; ++j) gof[(size_t)j] = k;
    h->n_groups = n_groups;
    {
        int32_t span = 0; // block half-bandwidth in the caller's pose order: no voxel leaves its group (checked above)
        for (int32_t k = 0; k < n_groups; ++k) span = std::max(span, pose_off[k + 1] - pose_off[k] - 1);
        bs.bb_hint = span;
    }
    h->g_pose_off.assign(pose_off, pose_off + n_groups + 1);
    h->g_vox_off.assign(voxel_off, voxel_off + n_groups + 1);
    TRY(bs_dmalloc(bs, &h->d_grp_of_pose, h->N));
    TRY(bs_dmalloc(bs, &h->d_gpo, n_groups + 1));
    TRY(bs_dmalloc(bs, &h->d_gaccept, n_groups));
    TRY(bs_dmalloc(bs, &h->d_gco, n_groups + 1));
    TRY(bs_dmalloc(bs, &h->d_gscal, 3 * (int64_t)n_groups));
    HIPCHK(lvba::copy_h2d(h->d_grp_of_pose, gof.data(), (size_t)h->N * sizeof(int32_t)));
    HIPCHK(lvba::copy_h2d(h->d_gpo, pose_off, (size_t)(n_groups + 1) * sizeof(int32_t)));
    HIPCHK(lvba::copy_h2d(h->d_gco, gco.data(), (size_t)(n_groups + 1) * sizeof(int64_t)));
    HIPCHK(lvba::PinnedCache::get().acquire((void **)&h->h_gpin, std::max<size_t>(4096, 3 * (size_t)n_groups * sizeof(double))));
    HIPCHK(lvba::PinnedCache::get().acquire((void **)&h->h_gacc, std::max<size_t>(4096, (size_t)n_groups * sizeof(int32_t))));
    // the caller's pose order is the solver's: the block-diagonal structure stays, and pose -> group is a plain table
    bs.ordering = 0;
    bs.n_groups = n_groups;
    bs.d_grp_of_pose = h->d_grp_of_pose;
    return LVBA_OK;
}

extern "C" int32_t lvba_balm_refine_groups(lvba_balm_t h, double *poses_inout, const lvba_balm_opts *opts, int32_t *n_iter,
                                           int32_t *status, double *cost_first, double *cost_last)
{
    if (!h || !poses_inout) return fail(LVBA_ERR_ARG, "NULL argument");
    if (h->n_groups < 1) return fail(LVBA_ERR_STATE, "refine_groups without set_groups");
    TRY(finalize(h));
    if (h->bs.d_bcr) return fail(LVBA_ERR_UNSUPPORTED, "grouped refinement needs the LDL^T solver");
    HIPCHK(hipSetDevice(h->bs.device));
    lvba_balm_opts o;
    if (opts) o = *opts; else lvba_balm_default_opts(&o);
    if (o.max_iter < 0) return fail(LVBA_ERR_ARG, "max_iter < 0");
    const int32_t G = h->n_groups;
    const int64_t n = 6 * (int64_t)h->N;
    (void)n;
    BlockSys &bs = h->bs;
    hipStream_t s = h->stream();
    TRY(upload_poses(h, poses_inout, h->d_pose_cur));
    struct St { double u, v, r1; bool calc, done; int iter; int32_t status; double first, last; };
    lvba::hvec<St> st((size_t)G);
    for (auto &q : st) q = St{o.u0, o.v0, 0.0, true, o.max_iter == 0, 0, LVBA_OK, 0.0, 0.0};
    lvba::hvec<double> u((size_t)G);
    double *c1 = h->d_gscal, *c2 = h->d_gscal + G, *q1d = h->d_gscal + 2 * (int64_t)G;
    int32_t worst = LVBA_OK;
    for (;;) {
        bool any = false, need_eval = false;
        for (const auto &q : st) { any = any || !q.done; need_eval = need_eval || (!q.done && q.calc); }
        if (!any) break;
        if (need_eval) { // :688-689 (groups whose last step was rejected get the same H, g, cost again: the kernels are deterministic)
            TRY(enqueue_eval(h, h->d_pose_cur));

            launch_reduce_chunks_groups(h->d_chunk_cost, h->d_gco, G, c1, s);
        }
        // finished groups stay in the joint system as identity blocks (negative damping: ldlt_prepare_kernel): they cost
        // nothing to factorise and cannot raise the pivot flag for the groups that are still running
        for (int32_t k = 0; k < G; ++k) u[(size_t)k] = st[(size_t)k].done ? -1.0 : st[(size_t)k].u;
        ev_begin(h, EV_SOLVE);
        TRY(bs_enqueue_solve_groups(bs, u.data()));                                            // :692-710
        ev_end(h, EV_SOLVE);
        launch_retract(h->d_pose_cur, bs.d_dx, h->d_pose_trial, h->N, s);                      // :722-727
        launch_predicted_decrease_groups(bs.Hblk(), bs.Bb, bs.g(), bs.d_dx, bs.d_u, h->d_gpo, G, q1d, s); // :729
        launch_cost_chunks(h->dev(), h->d_pose_trial, h->d_chunk_cost, s);                     // :731
        launch_reduce_chunks_groups(h->d_chunk_cost, h->d_gco, G, c2, s);
        HIPCHK(hipMemcpyAsync(h->h_gpin, h->d_gscal, 3 * (size_t)G * sizeof(double), hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(h->h_pin + 4, bs.d_status, sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        ev_collect(h);
        int pivot = 0;
        memcpy(&pivot, h->h_pin + 4, sizeof(int));
        // a zero pivot in one group's block poisons the band below it: the groups are not independent any more
        if (pivot) return fail(LVBA_NUM_FACTORIZATION, "zero or non-finite pivot in LDL^T of the grouped system");
        for (int32_t k = 0; k < G; ++k) {
            St &q = st[(size_t)k];
            h->h_gacc[k] = 0;
            if (q.done) continue;
            const double Vg = (double)(h->g_vox_off[(size_t)k + 1] - h->g_vox_off[(size_t)k]);
            if (need_eval && q.calc) q.r1 = h->h_gpin[k] / Vg;                                 // AVG_THR :634-635
            const double r1 = q.r1, r2 = h->h_gpin[G + k] / Vg, q1 = h->h_gpin[2 * G + k] / Vg; // :732
            double dq = r1 - r2;                                                               // :736
            if (!isfinite(r1) || !isfinite(r2)) q.status = LVBA_NUM_NONFINITE;
            if (q.iter == 0) q.first = r1;
            q.last = dq > 0 ? r2 : r1;
            if (dq > 0) {                                                                      // :744-752
                h->h_gacc[k] = 1;
                dq = dq / q1;
                q.v = 2.0;
                dq = 1.0 - pow(2.0 * dq - 1.0, 3.0);
                q.u *= (dq < (1.0 / 3.0) ? (1.0 / 3.0) : dq);
                q.calc = true;
            } else {                                                                           // :753-758
                q.u = q.u * q.v;
                q.v = 2.0 * q.v;
                q.calc = false;
            }
            q.iter += 1;
            if (fabs(r1 - r2) / r1 < o.rel_tol) q.done = true;                                 // :760
            if (q.iter >= o.max_iter) q.done = true;                                           // :686
            if (q.status > worst) worst = q.status;
        }
        HIPCHK(hipMemcpyAsync(h->d_gaccept, h->h_gacc, (size_t)G * sizeof(int32_t), hipMemcpyHostToDevice, s));
        launch_select_poses(h->d_pose_cur, h->d_pose_trial, h->d_gaccept, h->d_grp_of_pose, h->N, s);
        HIPCHK(hipStreamSynchronize(s)); // h_gacc is rewritten in the next iteration
    }
    TRY(download_poses(h, h->d_pose_cur, poses_inout));
    for (int32_t k = 0; k < G; ++k) {
        if (n_iter) n_iter[k] = st[(size_t)k].iter;
        if (status) status[k] = st[(size_t)k].status;
        if (cost_first) cost_first[k] = st[(size_t)k].first;
        if (cost_last) cost_last[k] = st[(size_t)k].last;
    }
    if (worst != LVBA_OK) return fail(worst, "non-finite cost in a group");
    return LVBA_OK;
}

// ------------------------------------------------------------------------------------------ multi-GPU
extern "C" int32_t lvba_balm_dist_init(lvba_balm_t h, int32_t n_ranks, int32_t rank, const char uid[128])
{
    if (!h || !uid) return fail(LVBA_ERR_ARG, "NULL argument");
    if (h->finalized) return fail(LVBA_ERR_STATE, "dist_init must precede the first cost/eval/refine call");
    if (h->n_groups > 0) return fail(LVBA_ERR_UNSUPPORTED, "groups and voxel shards do not combine");
    int64_t Vg = h->V;
    TRY(bs_dist_init(h->bs, n_ranks, rank, uid, &Vg));
    h->Vglobal = Vg;
    return LVBA_OK;
}

extern "C" int32_t lvba_balm_dist_init_external(lvba_balm_t h, int32_t n_ranks, int32_t rank, lvba_allreduce_fn fn, void *ctx)
{
    if (!h || !fn) return fail(LVBA_ERR_ARG, "NULL argument");
    if (h->finalized) return fail(LVBA_ERR_STATE, "dist_init must precede the first cost/eval/refine call");
    if (h->n_groups > 0) return fail(LVBA_ERR_UNSUPPORTED, "groups and voxel shards do not combine");
    int64_t Vg = h->V;
    TRY(bs_dist_init_external(h->bs, n_ranks, rank, fn, ctx, &Vg));
    h->Vglobal = Vg;
    return LVBA_OK;
}
