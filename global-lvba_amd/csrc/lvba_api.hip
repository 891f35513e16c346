// lvba_api.hip -- C-ABI of liblvba_hip.so (include/lvba_hip.h): problem packing, pose ordering, the
// Nielsen-LM driver of BALM2::damping_iter (reference include/BALM/bavoxel.hpp:662-767) and the RCCL
// reduction that replaces the 16-thread sum of bavoxel.hpp:626-633.  Host logic only; all arithmetic on
// problem data runs in the kernels of balm_kernels.hip / ldlt.hip.  No CPU fallback.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h> // types/enums only; the library is dlopen()ed in lvba_balm_dist_init
#include <dlfcn.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <new>
#include <queue>
#include <vector>

#include "../../include/lvba_hip.h"
#include "lvba_internal.h"

using namespace lvba;

// ------------------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";
static int32_t fail(int32_t code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail(e_ == hipErrorOutOfMemory ? LVBA_ERR_NOMEM : LVBA_ERR_DEVICE, "%s: %s (%s:%d)", \
                        #expr, hipGetErrorString(e_), __FILE__, __LINE__);                             \
    } while (0)

// ------------------------------------------------------------------------------------------ RCCL (lazy)
struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi g_rccl;
static int32_t rccl_load()
{
    if (g_rccl.lib) return LVBA_OK;
    void *lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) return fail(LVBA_ERR_DIST, "dlopen(librccl.so) failed: %s", dlerror());
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(lib, "ncclCommInitRank");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(lib, "ncclAllReduce");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(lib, "ncclCommDestroy");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(lib, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy)
        return fail(LVBA_ERR_DIST, "librccl.so lacks a required symbol");
    g_rccl.lib = lib;
    return LVBA_OK;
}
#define NCCLCHK(expr)                                                                                  \
    do {                                                                                               \
        ncclResult_t r_ = (expr);                                                                      \
        if (r_ != ncclSuccess)                                                                         \
            return fail(LVBA_ERR_DIST, "%s: %s", #expr, g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "rccl error"); \
    } while (0)

// ------------------------------------------------------------------------------------------ handle
enum { EV_COST = 0, EV_EVAL, EV_SOLVE, EV_REDUCE, EV_COSTK, EV_EVALK, EV_N };

struct lvba_balm_s {
    int device = 0;
    int32_t N = 0;
    int64_t V = 0, F = 0, Q = 0, n_chunks = 0, Vglobal = 0;
    hipStream_t stream = nullptr, stream2 = nullptr;
    std::vector<hipEvent_t> evA, evB;      // look-ahead fork/join events of the solver, one pair per panel
    hipGraph_t solve_graph = nullptr;      // the captured launch sequence of one damped solve
    hipGraphExec_t solve_exec = nullptr;
    bool graph_tried = false;
    // host copies kept until finalize()
    std::vector<int64_t> h_voff;
    std::vector<int32_t> h_pidx;
    std::vector<int64_t> h_chunk_v0;
    // configuration
    int ordering = 1;
    double band_frac = 0.6;
    bool finalized = false;
    // ordering / layout
    std::vector<int32_t> perm, iperm; // perm[internal] = caller, iperm[caller] = internal
    int32_t Bb = 0;
    bool use_band = false;
    // device data
    int64_t *d_voff = nullptr, *d_chunk_v0 = nullptr;
    int32_t *d_pidx = nullptr, *d_perm = nullptr;
    double *d_clu = nullptr;
    double *d_hg = nullptr; // [Hblk | g | scal(8)] contiguous: one all-reduce covers all
    int64_t hblk_doubles = 0;
    double *d_chunk_cost = nullptr;
    // pose-major assembly structures (see BalmDev)
    int32_t S = 1;
    int64_t nnzb = 0;
    int64_t *d_csc_off = nullptr, *d_blk_off = nullptr, *d_blk_slot = nullptr;
    int32_t *d_vox_of_pos = nullptr;
    int2 *d_pairs = nullptr;
    double *d_clu_csc = nullptr, *d_vrec = nullptr, *d_Y = nullptr, *d_part = nullptr;
    double *d_pose_in = nullptr, *d_pose_cur = nullptr, *d_pose_trial = nullptr;
    double *d_dx = nullptr, *d_out = nullptr; // d_out: staging for caller-order exports (>= 12N)
    double *d_scal2 = nullptr;                // [0]=trial cost sum, [1]=q1 numerator, [2]=u
    double *d_A = nullptr, *d_work = nullptr;
    int *d_status = nullptr;
    LdltMat A{};
    double *h_pin = nullptr; // pinned host staging, 16 doubles
    int64_t device_bytes = 0;
    // distributed
    int n_ranks = 1, rank = 0;
    ncclComm_t comm = nullptr;
    // LM state (bavoxel.hpp:664-671)
    bool lm_active = false, lm_done = false, is_calc_hess = true;
    lvba_balm_opts lm_opts{};
    double u = 0.01, v = 2.0, residual1 = 0.0;
    int iter = 0;
    bool have_eval = false;
    // profiling
    bool prof_on = false;
    hipEvent_t ev[EV_N][2] = {};
    bool ev_used[EV_N] = {};
    lvba_prof_t prof{};

    BalmDev dev() const
    {
        BalmDev d;
        d.n_poses = N; d.band_blocks = Bb; d.V = V; d.F = F; d.n_chunks = n_chunks;
        d.voff = d_voff; d.pidx = d_pidx; d.clu = d_clu; d.chunk_v0 = d_chunk_v0;
        d.S = S; d.csc_off = d_csc_off; d.clu_csc = d_clu_csc; d.vox_of_pos = d_vox_of_pos; d.vrec = d_vrec;
        d.Y = d_Y; d.part = d_part; d.nnzb = nnzb; d.blk_off = d_blk_off; d.blk_slot = d_blk_slot; d.pairs = d_pairs;
        return d;
    }
    double *Hblk() const { return d_hg; }
    double *g() const { return d_hg + hblk_doubles; }
    double *scal() const { return d_hg + hblk_doubles + 6 * (int64_t)N; } // [0] = eval cost sum
    int64_t hg_doubles() const { return hblk_doubles + 6 * (int64_t)N + 8; }
};

template <typename T>
static int32_t dmalloc(lvba_balm_s *h, T **p, int64_t count)
{
    HIPCHK(hipMalloc((void **)p, (size_t)std::max<int64_t>(count, 1) * sizeof(T)));
    h->device_bytes += count * (int64_t)sizeof(T);
    return LVBA_OK;
}
#define TRY(expr) do { int32_t rc_ = (expr); if (rc_ != LVBA_OK) return rc_; } while (0)

// ------------------------------------------------------------------------------------------ misc API
extern "C" int32_t lvba_version(void) { return 100; }
extern "C" const char *lvba_last_error(void) { return g_err; }
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
extern "C" int32_t lvba_balm_create(int32_t n_poses, int64_t n_voxels, const int64_t *voxel_off,
                                    const int32_t *pose_idx, const double *clusters, int32_t device,
                                    lvba_balm_t *out)
{
    if (!out) return fail(LVBA_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (n_poses < 1 || n_voxels < 1 || !voxel_off || !pose_idx || !clusters)
        return fail(LVBA_ERR_ARG, "n_poses/n_voxels must be >= 1 and arrays non-NULL");
    const int64_t base = voxel_off[0];
    const int64_t F = voxel_off[n_voxels] - base;
    if (F < 2 * n_voxels) return fail(LVBA_ERR_ARG, "every voxel needs >= 2 factors (push_voxel, bavoxel.hpp:52)");
    if (F >= (int64_t)1 << 31) return fail(LVBA_ERR_UNSUPPORTED, "more than 2^31 factors per shard");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(LVBA_ERR_DEVICE, "no HIP device available (liblvba_hip has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(LVBA_ERR_ARG, "device %d out of range [0,%d)", device, ndev);

    lvba_balm_s *h = new (std::nothrow) lvba_balm_s();
    if (!h) return fail(LVBA_ERR_NOMEM, "host allocation failed");
    h->device = device; h->N = n_poses; h->V = n_voxels; h->F = F; h->Vglobal = n_voxels;
    // validate + chunk
    h->h_voff.resize(n_voxels + 1);
    h->h_chunk_v0.clear();
    h->h_chunk_v0.push_back(0);
    int64_t nf = 0, nv = 0, Q = 0;
    for (int64_t a = 0; a < n_voxels; ++a) {
        const int64_t k = voxel_off[a + 1] - voxel_off[a];
        h->h_voff[a] = voxel_off[a] - base;
        if (k < 2) { delete h; return fail(LVBA_ERR_ARG, "voxel %lld has %lld factors (< 2)", (long long)a, (long long)k); }
        if (k > LVBA_CF) { delete h; return fail(LVBA_ERR_UNSUPPORTED, "voxel %lld has %lld observers (> %d per voxel not supported yet)", (long long)a, (long long)k, LVBA_CF); }
        if (nf + k > LVBA_CF || nv == LVBA_CV) { h->h_chunk_v0.push_back(a); nf = 0; nv = 0; }
        nf += k; nv += 1;
        Q += k * (k - 1) / 2;
    }
    h->h_voff[n_voxels] = F;
    h->h_chunk_v0.push_back(n_voxels);
    h->n_chunks = (int64_t)h->h_chunk_v0.size() - 1;
    h->Q = Q;
    h->h_pidx.assign(pose_idx, pose_idx + F);
    for (int64_t f = 0; f < F; ++f)
        if (pose_idx[f] < 0 || pose_idx[f] >= n_poses) { delete h; return fail(LVBA_ERR_ARG, "pose_idx[%lld] = %d out of range", (long long)f, pose_idx[f]); }

    auto bail = [&](int32_t rc) { lvba_balm_destroy(h); return rc; };
#define CTRY(expr) do { int32_t rc_ = (expr); if (rc_ != LVBA_OK) return bail(rc_); } while (0)
#define CHIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return bail(fail(e_ == hipErrorOutOfMemory ? LVBA_ERR_NOMEM : LVBA_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_))); } while (0)
    CHIP(hipSetDevice(device));
    CHIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    CTRY(dmalloc(h, &h->d_voff, n_voxels + 1));
    CTRY(dmalloc(h, &h->d_chunk_v0, h->n_chunks + 1));
    CTRY(dmalloc(h, &h->d_pidx, F));
    CTRY(dmalloc(h, &h->d_clu, 10 * F));
    CTRY(dmalloc(h, &h->d_chunk_cost, h->n_chunks));
    CHIP(hipMemcpy(h->d_voff, h->h_voff.data(), (size_t)(n_voxels + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    CHIP(hipMemcpy(h->d_chunk_v0, h->h_chunk_v0.data(), (size_t)(h->n_chunks + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    { // AoS [F][10] -> SoA [10][F], staged through a bounded host buffer
        const int64_t CH = 1 << 20;
        std::vector<double> tmp((size_t)std::min(F, CH));
        for (int e = 0; e < 10; ++e)
            for (int64_t s = 0; s < F; s += CH) {
                const int64_t m = std::min(CH, F - s);
                for (int64_t f = 0; f < m; ++f) tmp[f] = clusters[10 * (s + f) + e];
                CHIP(hipMemcpy(h->d_clu + (int64_t)e * F + s, tmp.data(), (size_t)m * sizeof(double), hipMemcpyHostToDevice));
            }
    }
    CHIP(hipHostMalloc((void **)&h->h_pin, 16 * sizeof(double), hipHostMallocDefault));
    for (int e = 0; e < EV_N; ++e)
        for (int s = 0; s < 2; ++s) CHIP(hipEventCreate(&h->ev[e][s]));
#undef CTRY
#undef CHIP
    *out = h;
    return LVBA_OK;
}

extern "C" int32_t lvba_balm_destroy(lvba_balm_t h)
{
    if (!h) return LVBA_OK;
    hipSetDevice(h->device);
    if (h->stream) hipStreamSynchronize(h->stream);
    if (h->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(h->comm);
    if (h->solve_exec) hipGraphExecDestroy(h->solve_exec);
    if (h->solve_graph) hipGraphDestroy(h->solve_graph);
    for (hipEvent_t e : h->evA) hipEventDestroy(e);
    for (hipEvent_t e : h->evB) hipEventDestroy(e);
    if (h->stream2) hipStreamDestroy(h->stream2);
    void *ptrs[] = {h->d_csc_off, h->d_blk_off, h->d_blk_slot, h->d_vox_of_pos, h->d_pairs, h->d_clu_csc, h->d_vrec, h->d_Y,
                    h->d_part, h->d_voff, h->d_chunk_v0, h->d_pidx, h->d_perm, h->d_clu, h->d_hg, h->d_chunk_cost, h->d_pose_in,
                    h->d_pose_cur, h->d_pose_trial, h->d_dx, h->d_out, h->d_scal2, h->d_A, h->d_work, h->d_status};
    for (void *p : ptrs)
        if (p) hipFree(p);
    if (h->h_pin) hipHostFree(h->h_pin);
    for (int e = 0; e < EV_N; ++e)
        for (int s = 0; s < 2; ++s)
            if (h->ev[e][s]) hipEventDestroy(h->ev[e][s]);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
    return LVBA_OK;
}

extern "C" int32_t lvba_balm_configure(lvba_balm_t h, int32_t ordering, double band_frac)
{
    if (!h) return fail(LVBA_ERR_ARG, "handle is NULL");
    if (h->finalized) return fail(LVBA_ERR_STATE, "configure must precede the first cost/eval/refine call");
    if (ordering != 0 && ordering != 1) return fail(LVBA_ERR_ARG, "ordering must be 0 or 1");
    if (!(band_frac >= 0.0)) return fail(LVBA_ERR_ARG, "band_frac must be >= 0");
    h->ordering = ordering;
    h->band_frac = band_frac;
    return LVBA_OK;
}

// ------------------------------------------------------------------------------------------ ordering
// Pose ordering for the band solver: reverse Cuthill-McKee on the pose co-visibility graph (byte adjacency
// matrix adj[N*N]) from two start rules (pseudo-peripheral node, minimum-degree node), then a barycenter
// refinement: positions are repeatedly replaced by the mean position of the neighbours and re-ranked, which
// interleaves the two sides of ring-like trajectories (loop closures).  The candidate with the smallest
// pose-block bandwidth wins.  One-off host work at finalize().
static int32_t bandwidth_of(const std::vector<std::vector<int32_t>> &nb, const std::vector<int32_t> &perm)
{
    const int N = (int)perm.size();
    std::vector<int32_t> ip(N);
    for (int i = 0; i < N; ++i) ip[perm[i]] = i;
    int32_t bw = 0;
    for (int i = 0; i < N; ++i)
        for (int j : nb[i]) bw = std::max(bw, std::abs(ip[i] - ip[j]));
    return bw;
}

static void rcm_from(const std::vector<std::vector<int32_t>> &nb, const std::vector<int32_t> &deg, bool peripheral,
                     std::vector<int32_t> &perm)
{
    const int N = (int)nb.size();
    std::vector<char> seen(N, 0), mark(N, 0);
    std::vector<int32_t> order, level(N);
    order.reserve(N);
    auto bfs_far = [&](int start) { // farthest node (minimal degree among the last level) from start
        std::queue<int> q;
        std::vector<int> touched;
        q.push(start); mark[start] = 1; touched.push_back(start); level[start] = 0;
        int last = start;
        while (!q.empty()) {
            int a = q.front(); q.pop();
            if (level[a] > level[last] || (level[a] == level[last] && deg[a] < deg[last])) last = a;
            for (int b : nb[a])
                if (!mark[b] && !seen[b]) { mark[b] = 1; level[b] = level[a] + 1; touched.push_back(b); q.push(b); }
        }
        for (int t : touched) mark[t] = 0;
        return last;
    };
    std::vector<int32_t> by_deg(N);
    for (int i = 0; i < N; ++i) by_deg[i] = i;
    std::stable_sort(by_deg.begin(), by_deg.end(), [&](int a, int b) { return deg[a] < deg[b]; });
    for (int root : by_deg) {
        if (seen[root]) continue;
        int s = root;
        if (peripheral)
            for (int pass = 0; pass < 3; ++pass) s = bfs_far(s);
        std::queue<int> q;
        q.push(s); seen[s] = 1;
        while (!q.empty()) {
            int a = q.front(); q.pop();
            order.push_back(a);
            for (int b : nb[a])
                if (!seen[b]) { seen[b] = 1; q.push(b); }
        }
    }
    perm.assign(order.rbegin(), order.rend());
}

static void rcm_order(const std::vector<uint8_t> &adj, int N, std::vector<int32_t> &perm)
{
    std::vector<std::vector<int32_t>> nb(N);
    std::vector<int32_t> deg(N, 0);
    for (int i = 0; i < N; ++i) {
        const uint8_t *row = adj.data() + (size_t)i * N;
        for (int j = 0; j < N; ++j)
            if (row[j] && j != i) nb[i].push_back(j);
        deg[i] = (int32_t)nb[i].size();
    }
    for (int i = 0; i < N; ++i)
        std::sort(nb[i].begin(), nb[i].end(), [&](int a, int b) { return deg[a] != deg[b] ? deg[a] < deg[b] : a < b; });
    std::vector<int32_t> best, cand;
    int32_t best_bw = INT32_MAX;
    for (int variant = 0; variant < 2; ++variant) {
        rcm_from(nb, deg, variant == 0, cand);
        const int32_t bw = bandwidth_of(nb, cand);
        if (bw < best_bw) { best_bw = bw; best = cand; }
    }
    // barycenter refinement of the best candidate
    std::vector<double> x(N), y(N);
    for (int i = 0; i < N; ++i) x[best[i]] = i;
    std::vector<int32_t> idx(N);
    for (int it = 1; it <= 200; ++it) {
        for (int i = 0; i < N; ++i) {
            if (nb[i].empty()) { y[i] = x[i]; continue; }
            double s = 0.0;
            for (int j : nb[i]) s += x[j];
            y[i] = s / (double)nb[i].size();
        }
        x.swap(y);
        if (it % 5 == 0) {
            for (int i = 0; i < N; ++i) idx[i] = i;
            std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return x[a] < x[b]; });
            const int32_t bw = bandwidth_of(nb, idx);
            if (bw < best_bw) { best_bw = bw; best = idx; }
            for (int i = 0; i < N; ++i) x[idx[i]] = i; // re-rank so the positions do not collapse
        }
    }
    perm = best;
}

static int32_t band_of(const lvba_balm_s *h, const std::vector<int32_t> &iperm)
{
    int32_t Bb = 0;
    for (int64_t a = 0; a < h->V; ++a) {
        int32_t lo = INT32_MAX, hi = -1;
        for (int64_t f = h->h_voff[a]; f < h->h_voff[a + 1]; ++f) {
            const int32_t p = iperm[h->h_pidx[f]];
            lo = std::min(lo, p); hi = std::max(hi, p);
        }
        Bb = std::max(Bb, hi - lo);
    }
    return Bb;
}

static int32_t finalize(lvba_balm_s *h)
{
    if (h->finalized) return LVBA_OK;
    HIPCHK(hipSetDevice(h->device));
    const int N = h->N;
    const int64_t n = 6 * (int64_t)N;
    h->perm.resize(N);
    h->iperm.resize(N);
    for (int i = 0; i < N; ++i) h->perm[i] = h->iperm[i] = i;
    int32_t Bb_nat = band_of(h, h->iperm);
    if (h->comm) { // the Hessian layout must agree on every rank: reduce over the global problem
        int32_t *dtmp = nullptr;
        HIPCHK(hipMalloc((void **)&dtmp, sizeof(int32_t)));
        HIPCHK(hipMemcpy(dtmp, &Bb_nat, sizeof(int32_t), hipMemcpyHostToDevice));
        NCCLCHK(g_rccl.AllReduce(dtmp, dtmp, 1, ncclInt32, ncclMax, h->comm, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipMemcpy(&Bb_nat, dtmp, sizeof(int32_t), hipMemcpyDeviceToHost));
        hipFree(dtmp);
    }
    h->Bb = Bb_nat;
    const bool small = (int64_t)N * N <= ((int64_t)1 << 29); // byte adjacency <= 512 MiB
    if (h->ordering == 1 && N > 2 && small) {
        std::vector<uint8_t> adj((size_t)N * N, 0);
        for (int64_t a = 0; a < h->V; ++a) {
            const int64_t f0 = h->h_voff[a], f1 = h->h_voff[a + 1];
            for (int64_t x = f0; x < f1; ++x)
                for (int64_t y = x + 1; y < f1; ++y) {
                    const int32_t i = h->h_pidx[x], j = h->h_pidx[y];
                    adj[(size_t)i * N + j] = 1; adj[(size_t)j * N + i] = 1;
                }
        }
        if (h->comm) {
            uint8_t *dadj = nullptr;
            HIPCHK(hipMalloc((void **)&dadj, adj.size()));
            HIPCHK(hipMemcpy(dadj, adj.data(), adj.size(), hipMemcpyHostToDevice));
            NCCLCHK(g_rccl.AllReduce(dadj, dadj, adj.size(), ncclUint8, ncclMax, h->comm, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            HIPCHK(hipMemcpy(adj.data(), dadj, adj.size(), hipMemcpyDeviceToHost));
            hipFree(dadj);
        }
        std::vector<int32_t> perm, iperm(N);
        rcm_order(adj, N, perm);
        for (int i = 0; i < N; ++i) iperm[perm[i]] = i;
        int32_t Bb_rcm = 0; // from the (global) adjacency so that all ranks agree
        for (int i = 0; i < N; ++i) {
            const uint8_t *row = adj.data() + (size_t)i * N;
            for (int j = 0; j < N; ++j)
                if (row[j]) Bb_rcm = std::max(Bb_rcm, std::abs(iperm[i] - iperm[j]));
        }
        if (Bb_rcm < Bb_nat) { h->perm = perm; h->iperm = iperm; h->Bb = Bb_rcm; }
    }
    const int64_t bw = 6 * (int64_t)h->Bb + 5;
    h->use_band = (double)(bw + LVBA_NB + 64) < h->band_frac * (double)n;
    if (!h->use_band) h->Bb = N - 1; // full lower block triangle
    const int64_t Bb1 = (int64_t)h->Bb + 1;
    h->hblk_doubles = (int64_t)N * Bb1 * 36;

    { // pose indices of the factors in solver order
        std::vector<int32_t> p((size_t)h->F);
        for (int64_t f = 0; f < h->F; ++f) p[f] = h->iperm[h->h_pidx[f]];
        HIPCHK(hipMemcpy(h->d_pidx, p.data(), (size_t)h->F * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    { // pose-major (CSC) view + per-block voxel lists for the atomic-free Hessian assembly
        const int64_t F = h->F;
        std::vector<int64_t> csc_off((size_t)N + 1, 0);
        for (int64_t f = 0; f < F; ++f) csc_off[(size_t)h->iperm[h->h_pidx[f]] + 1]++;
        for (int i = 0; i < N; ++i) csc_off[i + 1] += csc_off[i];
        std::vector<int32_t> csc_f((size_t)F), vox_of_pos((size_t)F), pos_of((size_t)F);
        {
            std::vector<int64_t> cur(csc_off.begin(), csc_off.end() - 1);
            for (int64_t a = 0; a < h->V; ++a)
                for (int64_t f = h->h_voff[a]; f < h->h_voff[a + 1]; ++f) {
                    const int64_t t = cur[h->iperm[h->h_pidx[f]]]++;
                    csc_f[t] = (int32_t)f; vox_of_pos[t] = (int32_t)a; pos_of[f] = (int32_t)t;
                }
        }
        const int64_t nslots = (int64_t)N * Bb1;
        std::vector<int64_t> start((size_t)nslots + 1, 0);
        auto slot_of = [&](int64_t fx, int64_t fy, int32_t &px, int32_t &py) {
            int32_t I = h->iperm[h->h_pidx[fx]], J = h->iperm[h->h_pidx[fy]];
            px = pos_of[fx]; py = pos_of[fy];
            if (I < J) { std::swap(I, J); std::swap(px, py); }
            return (int64_t)J * Bb1 + (I - J);
        };
        for (int64_t a = 0; a < h->V; ++a)
            for (int64_t x = h->h_voff[a]; x < h->h_voff[a + 1]; ++x)
                for (int64_t y = x + 1; y < h->h_voff[a + 1]; ++y) {
                    int32_t px, py;
                    start[slot_of(x, y, px, py) + 1]++;
                }
        std::vector<int64_t> blk_off, blk_slot;
        blk_off.push_back(0);
        for (int64_t sl = 0; sl < nslots; ++sl) {
            const int64_t c = start[sl + 1];
            start[sl + 1] = start[sl] + c; // exclusive prefix in start[sl]
            if (c > 0) { blk_slot.push_back(sl); blk_off.push_back(start[sl + 1]); }
        }
        std::vector<int2> pairs((size_t)h->Q);
        for (int64_t a = 0; a < h->V; ++a)
            for (int64_t x = h->h_voff[a]; x < h->h_voff[a + 1]; ++x)
                for (int64_t y = x + 1; y < h->h_voff[a + 1]; ++y) {
                    int32_t px, py;
                    const int64_t sl = slot_of(x, y, px, py);
                    pairs[(size_t)start[sl]++] = make_int2(px, py);
                }
        h->nnzb = (int64_t)blk_slot.size();
        // slices per pose: enough workgroups to fill the chip, but >= ~256 factors per slice
        int64_t Ssz = (2048 + N - 1) / N;
        const int64_t avg = F / N;
        Ssz = std::min<int64_t>(Ssz, std::max<int64_t>(1, avg / 256));
        h->S = (int32_t)std::max<int64_t>(1, std::min<int64_t>(Ssz, 64));
        TRY(dmalloc(h, &h->d_csc_off, N + 1));
        TRY(dmalloc(h, &h->d_vox_of_pos, F));
        TRY(dmalloc(h, &h->d_clu_csc, 10 * F));
        TRY(dmalloc(h, &h->d_vrec, 16 * h->V));
        TRY(dmalloc(h, &h->d_Y, 18 * F));
        TRY(dmalloc(h, &h->d_part, (int64_t)N * h->S * 32));
        TRY(dmalloc(h, &h->d_blk_off, h->nnzb + 1));
        TRY(dmalloc(h, &h->d_blk_slot, h->nnzb));
        TRY(dmalloc(h, &h->d_pairs, h->Q));
        HIPCHK(hipMemcpy(h->d_csc_off, csc_off.data(), (size_t)(N + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d_vox_of_pos, vox_of_pos.data(), (size_t)F * sizeof(int32_t), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d_blk_off, blk_off.data(), (size_t)(h->nnzb + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
        if (h->nnzb) HIPCHK(hipMemcpy(h->d_blk_slot, blk_slot.data(), (size_t)h->nnzb * sizeof(int64_t), hipMemcpyHostToDevice));
        if (h->Q) HIPCHK(hipMemcpy(h->d_pairs, pairs.data(), (size_t)h->Q * sizeof(int2), hipMemcpyHostToDevice));
        int32_t *d_csc_f = nullptr;
        HIPCHK(hipMalloc((void **)&d_csc_f, (size_t)F * sizeof(int32_t)));
        HIPCHK(hipMemcpy(d_csc_f, csc_f.data(), (size_t)F * sizeof(int32_t), hipMemcpyHostToDevice));
        launch_gather_csc(h->d_clu, d_csc_f, F, h->d_clu_csc, h->stream);
        HIPCHK(hipStreamSynchronize(h->stream));
        hipFree(d_csc_f);
    }
    TRY(dmalloc(h, &h->d_perm, N));
    HIPCHK(hipMemcpy(h->d_perm, h->perm.data(), (size_t)N * sizeof(int32_t), hipMemcpyHostToDevice));
    TRY(dmalloc(h, &h->d_hg, h->hg_doubles()));
    TRY(dmalloc(h, &h->d_pose_in, 12 * (int64_t)N));
    TRY(dmalloc(h, &h->d_pose_cur, 12 * (int64_t)N));
    TRY(dmalloc(h, &h->d_pose_trial, 12 * (int64_t)N));
    TRY(dmalloc(h, &h->d_dx, n));
    TRY(dmalloc(h, &h->d_out, 12 * (int64_t)N));
    TRY(dmalloc(h, &h->d_scal2, 8));
    TRY(dmalloc(h, &h->d_status, 4));
    h->A.n = n;
    if (h->use_band) {
        const int64_t ldab = bw + LVBA_NB + 64;
        h->A.ld = ldab - 1; h->A.bw = bw;
        TRY(dmalloc(h, &h->d_A, ldab * n + ldab));
    } else {
        h->A.ld = n; h->A.bw = n - 1;
        TRY(dmalloc(h, &h->d_A, n * n));
    }
    h->A.a = h->d_A;
    TRY(dmalloc(h, &h->d_work, ldlt_workspace_doubles(n, h->A.bw)));
    if (!getenv("LVBA_NO_LOOKAHEAD")) {
        HIPCHK(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
        const int64_t np = ldlt_num_panels(n);
        h->evA.resize(np); h->evB.resize(np);
        for (int64_t i = 0; i < np; ++i) {
            HIPCHK(hipEventCreateWithFlags(&h->evA[i], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&h->evB[i], hipEventDisableTiming));
        }
    }
    HIPCHK(hipMemset(h->d_hg, 0, (size_t)h->hg_doubles() * sizeof(double)));
    // host copies are no longer needed
    std::vector<int64_t>().swap(h->h_voff);
    std::vector<int32_t>().swap(h->h_pidx);
    std::vector<int64_t>().swap(h->h_chunk_v0);
    h->finalized = true;
    return LVBA_OK;
}

extern "C" int32_t lvba_balm_info(lvba_balm_t h, lvba_balm_info_t *info)
{
    if (!h || !info) return fail(LVBA_ERR_ARG, "NULL argument");
    TRY(finalize(h));
    info->n_poses = h->N; info->n_ranks = h->n_ranks; info->n_voxels = h->V; info->n_voxels_global = h->Vglobal;
    info->n_factors = h->F; info->n_pairs = h->Q; info->n_chunks = h->n_chunks; info->n_blocks = h->nnzb; info->band_blocks = h->Bb;
    info->use_band = h->use_band ? 1 : 0; info->hess_bytes = h->hblk_doubles * 8; info->device_bytes = h->device_bytes;
    return LVBA_OK;
}

extern "C" int32_t lvba_balm_get_ordering(lvba_balm_t h, int32_t *perm)
{
    if (!h || !perm) return fail(LVBA_ERR_ARG, "NULL argument");
    TRY(finalize(h));
    memcpy(perm, h->perm.data(), (size_t)h->N * sizeof(int32_t));
    return LVBA_OK;
}

// ------------------------------------------------------------------------------------------ profiling
static void ev_begin(lvba_balm_s *h, int which)
{
    if (h->prof_on) hipEventRecord(h->ev[which][0], h->stream);
}
static void ev_end(lvba_balm_s *h, int which)
{
    if (h->prof_on) { hipEventRecord(h->ev[which][1], h->stream); h->ev_used[which] = true; }
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
static int32_t enqueue_cost(lvba_balm_s *h, const double *d_poses, double *dst)
{
    ev_begin(h, EV_COST);
    launch_cost(h->dev(), d_poses, h->d_chunk_cost, dst, h->stream, h->prof_on ? h->ev[EV_COSTK][0] : nullptr,
                h->prof_on ? h->ev[EV_COSTK][1] : nullptr);
    if (h->prof_on) h->ev_used[EV_COSTK] = true;
    ev_end(h, EV_COST);
    if (h->comm) {
        ev_begin(h, EV_REDUCE);
        NCCLCHK(g_rccl.AllReduce(dst, dst, 1, ncclDouble, ncclSum, h->comm, h->stream));
        ev_end(h, EV_REDUCE);
    }
    HIPCHK(hipGetLastError());
    return LVBA_OK;
}

// enqueue: H, g, cost at device poses -> d_hg (all-reduced over ranks)
static int32_t enqueue_eval(lvba_balm_s *h, const double *d_poses)
{
    ev_begin(h, EV_EVAL);
    launch_eval(h->dev(), d_poses, h->Hblk(), h->hblk_doubles, h->g(), h->d_chunk_cost, h->scal(), h->comm != nullptr, h->stream,
                h->prof_on ? h->ev[EV_EVALK][0] : nullptr, h->prof_on ? h->ev[EV_EVALK][1] : nullptr);
    if (h->prof_on) h->ev_used[EV_EVALK] = true;
    ev_end(h, EV_EVAL);
    if (h->comm) {
        ev_begin(h, EV_REDUCE);
        NCCLCHK(g_rccl.AllReduce(h->d_hg, h->d_hg, (size_t)(h->hblk_doubles + 6 * (int64_t)h->N + 1), ncclDouble, ncclSum,
                                 h->comm, h->stream));
        ev_end(h, EV_REDUCE);
    }
    HIPCHK(hipGetLastError());
    h->have_eval = true;
    return LVBA_OK;
}

// enqueue: dx = -(H + u diag H)^-1 g  (u taken from d_scal2[2]).  The launch sequence is static per handle
// (~5 kernels per 64-column panel on two streams), so it is captured once into a hipGraph and replayed.
static void solve_launches(lvba_balm_s *h)
{
    ldlt_solve(h->A, h->Hblk(), h->Bb, h->N, h->g(), h->d_scal2 + 2, h->d_dx, h->d_work, h->d_status, h->stream,
               h->stream2, h->evA.empty() ? nullptr : h->evA.data(), h->evB.empty() ? nullptr : h->evB.data());
}

static int32_t enqueue_solve(lvba_balm_s *h, double u)
{
    h->h_pin[8] = u;
    HIPCHK(hipMemcpyAsync(h->d_scal2 + 2, h->h_pin + 8, sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (!h->graph_tried) {
        h->graph_tried = true;
        if (!getenv("LVBA_NO_GRAPH")) {
            (void)hipGetLastError();
            if (hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                solve_launches(h);
                hipGraph_t gph = nullptr;
                if (hipStreamEndCapture(h->stream, &gph) == hipSuccess && gph &&
                    hipGraphInstantiate(&h->solve_exec, gph, nullptr, nullptr, 0) == hipSuccess) {
                    h->solve_graph = gph;
                } else {
                    if (gph) hipGraphDestroy(gph);
                    h->solve_exec = nullptr;
                }
            }
            (void)hipGetLastError(); // a failed capture falls back to eager launches
        }
    }
    ev_begin(h, EV_SOLVE);
    if (h->solve_exec) HIPCHK(hipGraphLaunch(h->solve_exec, h->stream));
    else solve_launches(h);
    ev_end(h, EV_SOLVE);
    HIPCHK(hipGetLastError());
    return LVBA_OK;
}

static int32_t upload_poses(lvba_balm_s *h, const double *poses, double *d_dst)
{
    HIPCHK(hipMemcpyAsync(h->d_pose_in, poses, (size_t)12 * h->N * sizeof(double), hipMemcpyHostToDevice, h->stream));
    launch_import_poses(h->d_pose_in, h->d_perm, h->N, d_dst, h->stream);
    return LVBA_OK;
}

// ------------------------------------------------------------------------------------------ a7 / a6
extern "C" int32_t lvba_balm_cost(lvba_balm_t h, const double *poses, int32_t is_avg, double *cost)
{
    if (!h || !poses || !cost) return fail(LVBA_ERR_ARG, "NULL argument");
    TRY(finalize(h));
    HIPCHK(hipSetDevice(h->device));
    TRY(upload_poses(h, poses, h->d_pose_trial));
    TRY(enqueue_cost(h, h->d_pose_trial, h->d_scal2));
    HIPCHK(hipMemcpyAsync(h->h_pin, h->d_scal2, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    ev_collect(h);
    *cost = is_avg ? h->h_pin[0] / (double)h->Vglobal : h->h_pin[0];
    if (!isfinite(*cost)) return fail(LVBA_NUM_NONFINITE, "non-finite cost");
    return LVBA_OK;
}

extern "C" int32_t lvba_balm_eval(lvba_balm_t h, const double *poses, double *H, double *g, double *cost_avg)
{
    if (!h || !poses) return fail(LVBA_ERR_ARG, "NULL argument");
    TRY(finalize(h));
    HIPCHK(hipSetDevice(h->device));
    TRY(upload_poses(h, poses, h->d_pose_cur));
    TRY(enqueue_eval(h, h->d_pose_cur));
    HIPCHK(hipMemcpyAsync(h->h_pin, h->scal(), sizeof(double), hipMemcpyDeviceToHost, h->stream));
    const int64_t n = 6 * (int64_t)h->N;
    if (g) {
        launch_export_vec(h->g(), h->d_perm, h->N, h->d_out, h->stream);
        HIPCHK(hipMemcpyAsync(g, h->d_out, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    }
    double *dH = nullptr;
    if (H) {
        HIPCHK(hipMalloc((void **)&dH, (size_t)(n * n) * sizeof(double)));
        launch_export_dense(h->Hblk(), h->Bb, h->N, h->d_perm, dH, h->stream);
        HIPCHK(hipMemcpyAsync(H, dH, (size_t)(n * n) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    if (dH) hipFree(dH);
    ev_collect(h);
    if (cost_avg) *cost_avg = h->h_pin[0] / (double)h->Vglobal;
    return LVBA_OK;
}

extern "C" int32_t lvba_balm_solve(lvba_balm_t h, double u, double *dx)
{
    if (!h || !dx) return fail(LVBA_ERR_ARG, "NULL argument");
    if (!h->finalized || !h->have_eval) return fail(LVBA_ERR_STATE, "lvba_balm_solve needs a prior lvba_balm_eval");
    HIPCHK(hipSetDevice(h->device));
    TRY(enqueue_solve(h, u));
    launch_export_vec(h->d_dx, h->d_perm, h->N, h->d_out, h->stream);
    HIPCHK(hipMemcpyAsync(dx, h->d_out, (size_t)6 * h->N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    int st = 0;
    HIPCHK(hipMemcpyAsync(&st, h->d_status, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    ev_collect(h);
    if (st) return fail(LVBA_NUM_FACTORIZATION, "zero or non-finite pivot in LDL^T");
    return LVBA_OK;
}

// ------------------------------------------------------------------------------------------ a8: LM
extern "C" int32_t lvba_balm_lm_begin(lvba_balm_t h, const double *poses, const lvba_balm_opts *opts)
{
    if (!h || !poses) return fail(LVBA_ERR_ARG, "NULL argument");
    TRY(finalize(h));
    HIPCHK(hipSetDevice(h->device));
    if (opts) h->lm_opts = *opts; else lvba_balm_default_opts(&h->lm_opts);
    if (h->lm_opts.max_iter < 0) return fail(LVBA_ERR_ARG, "max_iter < 0");
    TRY(upload_poses(h, poses, h->d_pose_cur));
    h->u = h->lm_opts.u0; h->v = h->lm_opts.v0;
    h->is_calc_hess = true; h->iter = 0; h->residual1 = 0.0;
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
    HIPCHK(hipSetDevice(h->device));
    const bool evaluated = h->is_calc_hess;
    const int64_t n = 6 * (int64_t)h->N;
    if (evaluated) TRY(enqueue_eval(h, h->d_pose_cur));                                    // :688-689
    TRY(enqueue_solve(h, h->u));                                                           // :692-710
    launch_retract(h->d_pose_cur, h->d_dx, h->d_pose_trial, h->N, h->stream);              // :722-727
    launch_predicted_decrease(h->Hblk(), h->Bb, h->g(), h->d_dx, h->u, n, h->d_scal2 + 1, h->stream); // :729
    TRY(enqueue_cost(h, h->d_pose_trial, h->d_scal2));                                     // :731
    HIPCHK(hipMemcpyAsync(h->h_pin, h->d_scal2, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(h->h_pin + 2, h->scal(), sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(h->h_pin + 4, h->d_status, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    ev_collect(h);
    const double Vg = (double)h->Vglobal;
    if (evaluated) h->residual1 = h->h_pin[2] / Vg;                                        // AVG_THR :634-635
    const double residual1 = h->residual1;
    const double residual2 = h->h_pin[0] / Vg;
    const double q1 = h->h_pin[1] / Vg;                                                    // :732
    int st = 0;
    memcpy(&st, h->h_pin + 4, sizeof(int));
    double q = residual1 - residual2;                                                      // :736
    int32_t status = LVBA_OK;
    if (st) status = LVBA_NUM_FACTORIZATION;
    else if (!isfinite(residual2) || !isfinite(residual1)) status = LVBA_NUM_NONFINITE;
    if (row) {
        row->iter = h->iter; row->accepted = q > 0; row->evaluated = evaluated; row->status = status;
        row->residual1 = residual1; row->residual2 = residual2; row->u = h->u; row->v = h->v; row->q = q; row->q1 = q1;
    }
    if (q > 0) {                                                                           // :744-752
        std::swap(h->d_pose_cur, h->d_pose_trial);
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
    if (status == LVBA_NUM_FACTORIZATION) return fail(status, "zero or non-finite pivot in LDL^T (iteration %d)", h->iter - 1);
    if (status == LVBA_NUM_NONFINITE) return fail(status, "non-finite cost (iteration %d)", h->iter - 1);
    return LVBA_OK;
}

extern "C" int32_t lvba_balm_lm_end(lvba_balm_t h, double *poses_out)
{
    if (!h) return fail(LVBA_ERR_ARG, "handle is NULL");
    if (!h->lm_active) return fail(LVBA_ERR_STATE, "lm_end without lm_begin");
    HIPCHK(hipSetDevice(h->device));
    if (poses_out) {
        launch_export_poses(h->d_pose_cur, h->d_perm, h->N, h->d_out, h->stream);
        HIPCHK(hipMemcpyAsync(poses_out, h->d_out, (size_t)12 * h->N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
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
    int32_t rows = 0, done = h->lm_done ? 1 : 0, rc = LVBA_OK;
    while (!done) {
        lvba_lm_trace row;
        rc = lvba_balm_lm_step(h, &row, &done);
        if (rc < 0) break;
        if (trace) trace[rows] = row;
        rows++;
        if (rc > 0) break; // numerical failure: stop like the reference's FAILURE early-return (lvba_system.cpp:1646)
    }
    if (n_trace) *n_trace = rows;
    const int32_t rc2 = lvba_balm_lm_end(h, poses_inout);
    return rc != LVBA_OK ? rc : rc2;
}

// ------------------------------------------------------------------------------------------ multi-GPU
extern "C" int32_t lvba_dist_unique_id(char uid[128])
{
    if (!uid) return fail(LVBA_ERR_ARG, "uid is NULL");
    TRY(rccl_load());
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
    ncclUniqueId id;
    NCCLCHK(g_rccl.GetUniqueId(&id));
    memcpy(uid, &id, 128);
    return LVBA_OK;
}

extern "C" int32_t lvba_balm_dist_init(lvba_balm_t h, int32_t n_ranks, int32_t rank, const char uid[128])
{
    if (!h || !uid) return fail(LVBA_ERR_ARG, "NULL argument");
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(LVBA_ERR_ARG, "bad rank %d of %d", rank, n_ranks);
    if (h->finalized) return fail(LVBA_ERR_STATE, "dist_init must precede the first cost/eval/refine call");
    // a 1-rank job needs no communicator; LVBA_SINGLE_RANK_COMM=1 builds one anyway so that the whole
    // RCCL path (dlopen, communicator, all-reduces) can be exercised on a 1-GPU box
    if (n_ranks == 1 && !getenv("LVBA_SINGLE_RANK_COMM")) return LVBA_OK;
    TRY(rccl_load());
    HIPCHK(hipSetDevice(h->device));
    ncclUniqueId id;
    memcpy(&id, uid, 128);
    NCCLCHK(g_rccl.CommInitRank(&h->comm, n_ranks, id, rank));
    h->n_ranks = n_ranks; h->rank = rank;
    // global voxel count for the AVG_THR averages
    int64_t *dv = nullptr;
    HIPCHK(hipMalloc((void **)&dv, sizeof(int64_t)));
    HIPCHK(hipMemcpy(dv, &h->V, sizeof(int64_t), hipMemcpyHostToDevice));
    NCCLCHK(g_rccl.AllReduce(dv, dv, 1, ncclInt64, ncclSum, h->comm, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(&h->Vglobal, dv, sizeof(int64_t), hipMemcpyDeviceToHost));
    hipFree(dv);
    return LVBA_OK;
}
