// block_system.hip -- shared host logic of both stages (see block_system.h): error message storage, lazy RCCL
// binding, block ordering (RCM + barycenter refinement), pose-major factor order and per-block pair lists of the
// atomic-free assembly, block-band store, LDL^T solve (captured into a hipGraph), all-reduce.
#include <dlfcn.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <map>
#include <mutex>
#include <queue>
#include <string>
#include <vector>
#include "host_arena.h"
#include "block_system.h"
#include "pair_lists.h"
#include "ordering.h"
#include "nd_plan.h"
#include "host_tables.h"

static std::atomic<int> g_graph_inhibit{0};
namespace lvba { void bs_graph_inhibit(int delta) { g_graph_inhibit.fetch_add(delta); } }

// ------------------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";
int32_t lvba_fail(int32_t code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
extern "C" const char *lvba_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------------------ RCCL (lazy)
RcclApi g_rccl;
int32_t rccl_load()
{
    if (g_rccl.lib) return LVBA_OK;
    void *lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) return lvba_fail(LVBA_ERR_DIST, "dlopen(librccl.so) failed: %s", dlerror());
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(lib, "ncclCommInitRank");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(lib, "ncclAllReduce");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(lib, "ncclCommDestroy");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(lib, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy)
        return lvba_fail(LVBA_ERR_DIST, "librccl.so lacks a required symbol");
    g_rccl.lib = lib;
    return LVBA_OK;
}

extern "C" int32_t lvba_dist_unique_id(char uid[128])
{
    if (!uid) return lvba_fail(LVBA_ERR_ARG, "uid is NULL");
    TRY(rccl_load());
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
    ncclUniqueId id;
    NCCLCHK(g_rccl.GetUniqueId(&id));
    memcpy(uid, &id, 128);
    return LVBA_OK;
}

namespace lvba {

int32_t bs_comm_allreduce(BlockSys &bs, void *dbuf, size_t count, ncclDataType_t dt, ncclRedOp_t op)
{
    if (bs.comm) {
        NCCLCHK(g_rccl.AllReduce(dbuf, dbuf, count, dt, op, bs.comm, bs.stream));
        return LVBA_OK;
    }
    if (!bs.ext_allreduce) return LVBA_OK;
    // the caller's transport (lvba_*_dist_init_external): same contract as ncclAllReduce in place on bs.stream, except that it
    // may return before or after the data has moved -- it is handed the stream and orders itself against it
    const int32_t edt = dt == ncclDouble ? LVBA_DT_F64 : dt == ncclInt64 ? LVBA_DT_I64 : dt == ncclInt32 ? LVBA_DT_I32 : dt == ncclUint8 ? LVBA_DT_U8 : -1;
    const int32_t eop = op == ncclSum ? LVBA_OP_SUM : op == ncclMax ? LVBA_OP_MAX : -1;
    if (edt < 0 || eop < 0) return lvba_fail(LVBA_ERR_UNSUPPORTED, "external transport: unsupported all-reduce type");
    const int32_t rc = bs.ext_allreduce(bs.ext_ctx, dbuf, count, edt, eop, (void *)bs.stream);
    if (rc != 0) return lvba_fail(LVBA_ERR_DIST, "external all-reduce failed with %d", rc);
    return LVBA_OK;
}

int32_t bs_init(BlockSys &bs, int device)
{
    bs.device = device;
    HIPCHK(hipSetDevice(device));
    HIPCHK(lvba::StreamCache::get().acquire(&bs.stream));
    HIPCHK(hipHostMalloc((void **)&bs.h_pin_u, 2 * sizeof(double), hipHostMallocDefault)); // (re-made larger by a grouped problem, bs_build)
    return LVBA_OK;
}

static int32_t dist_global_count(BlockSys &bs, int64_t *group_count_inout)
{
    if (!group_count_inout) return LVBA_OK; // global group count (the AVG_THR averages of the BALM stage)
    int64_t *dv = nullptr;
    HIPCHK(hipMalloc((void **)&dv, sizeof(int64_t)));
    HIPCHK(lvba::copy_h2d(dv, group_count_inout, sizeof(int64_t)));
    const int32_t rc = bs_comm_allreduce(bs, dv, 1, ncclInt64, ncclSum);
    if (rc == LVBA_OK) {
        HIPCHK(hipStreamSynchronize(bs.stream));
        HIPCHK(lvba::copy_d2h(group_count_inout, dv, sizeof(int64_t)));
    }
    hipFree(dv);
    return rc;
}

int32_t bs_dist_init(BlockSys &bs, int32_t n_ranks, int32_t rank, const char uid[128], int64_t *group_count_inout)
{
    if (!uid) return lvba_fail(LVBA_ERR_ARG, "uid is NULL");
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return lvba_fail(LVBA_ERR_ARG, "bad rank %d of %d", rank, n_ranks);
    if (bs.built) return lvba_fail(LVBA_ERR_STATE, "dist_init must precede the first cost/eval/refine call");
    // a 1-rank job needs no communicator; LVBA_SINGLE_RANK_COMM=1 builds one anyway so that the whole
    // RCCL path (dlopen, communicator, all-reduces) can be exercised on a 1-GPU box
    HIPCHK(hipSetDevice(bs.device));
    if (n_ranks == 1 && !getenv("LVBA_SINGLE_RANK_COMM")) return LVBA_OK;
    TRY(rccl_load());
    ncclUniqueId id;
    memcpy(&id, uid, 128);
    NCCLCHK(g_rccl.CommInitRank(&bs.comm, n_ranks, id, rank));
    bs.n_ranks = n_ranks; bs.rank = rank;
    return dist_global_count(bs, group_count_inout);
}

// The same with the caller's all-reduce instead of RCCL (include/lvba_hip.h: lvba_allreduce_fn).
int32_t bs_dist_init_external(BlockSys &bs, int32_t n_ranks, int32_t rank, lvba_allreduce_fn fn, void *ctx, int64_t *group_count_inout)
{
    if (!fn) return lvba_fail(LVBA_ERR_ARG, "all-reduce callback is NULL");
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return lvba_fail(LVBA_ERR_ARG, "bad rank %d of %d", rank, n_ranks);
    if (bs.built) return lvba_fail(LVBA_ERR_STATE, "dist_init must precede the first cost/eval/refine call");
    HIPCHK(hipSetDevice(bs.device));
    bs.ext_allreduce = fn; bs.ext_ctx = ctx;
    bs.graph_tried = true; // a callback cannot be captured into the solve graph (and may drive the device from several threads)
    bs.n_ranks = n_ranks; bs.rank = rank;
    return dist_global_count(bs, group_count_inout);
}

int32_t bs_allreduce(BlockSys &bs, double *buf, size_t count)
{
    return bs_comm_allreduce(bs, buf, count, ncclDouble, ncclSum);
}

// [Hblk | g | cost] <-> [blocks of the union pattern | g | cost]
__global__ void hg_pack_kernel(const double *__restrict__ hg, const int64_t *__restrict__ slot, int64_t n_ar, int64_t hblk_doubles,
                               int64_t tail, double *__restrict__ buf)
{
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t < 36 * n_ar) {
        const int64_t b = t / 36;
        buf[t] = hg[slot[b] * 36 + (t - 36 * b)];
    } else if (t < 36 * n_ar + tail)
        buf[t] = hg[hblk_doubles + (t - 36 * n_ar)];
}
__global__ void hg_unpack_kernel(double *__restrict__ hg, const int64_t *__restrict__ slot, int64_t n_ar, int64_t hblk_doubles,
                                 int64_t tail, const double *__restrict__ buf)
{
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t < 36 * n_ar) {
        const int64_t b = t / 36;
        hg[slot[b] * 36 + (t - 36 * b)] = buf[t];
    } else if (t < 36 * n_ar + tail)
        hg[hblk_doubles + (t - 36 * n_ar)] = buf[t];
}

// The block-band store is mostly structural zeros (C3: 4e5 non-zero blocks of 8.7e5 slots, 270 MB): when the union sparsity
// pattern is known (it is whenever the ordering was computed from the all-reduced adjacency) only its blocks travel --
// xGMI all-reduce time is proportional to bytes, and slots outside the union are zero on every rank.
__global__ void hg_zero_slots_kernel(double *__restrict__ hg, const int64_t *__restrict__ slot, int64_t n_ar)
{
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t < 36 * n_ar) hg[slot[t / 36] * 36 + t % 36] = 0.0;
}
// Before an evaluation of a multi-rank job: the blocks other ranks reduced into the store must read as zero again (this rank's
// kernels only write the blocks of its own shard).  With a union pattern only its slots are cleared -- a dissected system keeps the
// full lower block triangle (N^2 blocks: 28.8 GB at 10 000 poses), of which the pattern is a few per cent.
int32_t bs_clear_reduced(BlockSys &bs)
{
    if (!bs.distributed()) return LVBA_OK;
    if (bs.d_ar_slot)
        hipLaunchKernelGGL(hg_zero_slots_kernel, dim3((unsigned)((36 * bs.n_ar + 255) / 256)), dim3(256), 0, bs.stream, bs.d_hg, bs.d_ar_slot, bs.n_ar);
    else
        HIPCHK(hipMemsetAsync(bs.d_hg, 0, (size_t)bs.hblk_doubles * sizeof(double), bs.stream));
    HIPCHK(hipGetLastError());
    return LVBA_OK;
}

int32_t bs_allreduce_hg(BlockSys &bs)
{
    if (!bs.distributed()) return LVBA_OK;
    const int64_t tail = 6 * (int64_t)bs.N + 1;
    if (!bs.d_ar_slot) return bs_allreduce(bs, bs.d_hg, (size_t)(bs.hblk_doubles + tail));
    const int64_t total = 36 * bs.n_ar + tail;
    const unsigned grid = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(hg_pack_kernel, dim3(grid), dim3(256), 0, bs.stream, bs.d_hg, bs.d_ar_slot, bs.n_ar, bs.hblk_doubles, tail, bs.d_arbuf);
    TRY(bs_allreduce(bs, bs.d_arbuf, (size_t)total));
    hipLaunchKernelGGL(hg_unpack_kernel, dim3(grid), dim3(256), 0, bs.stream, bs.d_hg, bs.d_ar_slot, bs.n_ar, bs.hblk_doubles, tail, bs.d_arbuf);
    HIPCHK(hipGetLastError());
    return LVBA_OK;
}

// Slots (J * (Bb + 1) + dI) of the STRUCTURALLY non-zero blocks of the block-band store, ascending, diagonal blocks included:
// the union pattern of the packed all-reduce when there is one (multi-rank), else this rank's pair-list destinations.  Known
// from the set-up alone -- no evaluation is needed to size a sparse download of H.
int32_t bs_pattern_slots(BlockSys &bs, lvba::hvec<int64_t> &slots)
{
    const int64_t Bb1 = (int64_t)bs.Bb + 1;
    slots.clear();
    if (bs.d_ar_slot) {
        slots.resize((size_t)bs.n_ar);
        HIPCHK(hipMemcpy(slots.data(), bs.d_ar_slot, (size_t)bs.n_ar * sizeof(int64_t), hipMemcpyDeviceToHost));
        return LVBA_OK; // sorted, diagonal included (bs_build)
    }
    if (bs.distributed()) { // no union pattern known: every slot inside the matrix
        for (int64_t J = 0; J < bs.N; ++J)
            for (int64_t dI = 0; dI < Bb1 && J + dI < bs.N; ++dI) slots.push_back(J * Bb1 + dI);
        return LVBA_OK;
    }
    // item destinations: a slot (>= 0) or a partial block (< 0) that balm_pair_reduce_kernel sums into multi_slot[m]
    slots.resize((size_t)(bs.n_items + bs.n_multi));
    if (bs.n_items) HIPCHK(hipMemcpy(slots.data(), bs.d_blk_slot, (size_t)bs.n_items * sizeof(int64_t), hipMemcpyDeviceToHost));
    if (bs.n_multi) HIPCHK(hipMemcpy(slots.data() + bs.n_items, bs.d_multi_slot, (size_t)bs.n_multi * sizeof(int64_t), hipMemcpyDeviceToHost));
    slots.erase(std::remove_if(slots.begin(), slots.end(), [](int64_t v) { return v < 0; }), slots.end());
    for (int64_t J = 0; J < bs.N; ++J) slots.push_back(J * Bb1);
    std::sort(slots.begin(), slots.end());
    slots.erase(std::unique(slots.begin(), slots.end()), slots.end());
    return LVBA_OK;
}

// The blocks of `slots` (host, n of them) gathered on the device and brought to `out` [n][36] (pageable host memory) in slabs of
// <= 64 MB through a pinned buffer on bs.stream: one gather kernel + a handful of copies instead of one blocking copy per
// block column (10 000 of them at the C4 size).
int32_t bs_download_blocks(BlockSys &bs, const int64_t *slots, int64_t n, double *out)
{
    if (n <= 0) return LVBA_OK;
    DevBuf d_slots(bs.stream), d_buf(bs.stream);
    HIPCHK(d_slots.alloc((size_t)n * sizeof(int64_t)));
    HIPCHK(d_buf.alloc((size_t)n * 36 * sizeof(double)));
    HIPCHK(lvba::copy_h2d(d_slots.as<int64_t>(), slots, (size_t)n * sizeof(int64_t)));
    hipLaunchKernelGGL(hg_pack_kernel, dim3((unsigned)((36 * n + 255) / 256)), dim3(256), 0, bs.stream, bs.d_hg, d_slots.as<int64_t>(), n,
                       bs.hblk_doubles, (int64_t)0, d_buf.as<double>());
    HIPCHK(hipGetLastError());
    const size_t total = (size_t)n * 36 * sizeof(double), slab = std::min<size_t>(total, (size_t)64 << 20);
    void *pin = nullptr;
    HIPCHK(hipHostMalloc(&pin, slab, hipHostMallocDefault));
    int32_t rc = LVBA_OK;
    for (size_t o = 0; o < total && rc == LVBA_OK; o += slab) {
        const size_t nb = std::min(slab, total - o);
        if (hipMemcpyAsync(pin, reinterpret_cast<const char *>(d_buf.as<double>()) + o, nb, hipMemcpyDeviceToHost, bs.stream) != hipSuccess ||
            hipStreamSynchronize(bs.stream) != hipSuccess)
            rc = LVBA_ERR_DEVICE;
        else
            memcpy(reinterpret_cast<char *>(out) + o, pin, nb);
    }
    hipHostFree(pin);
    return rc;
}

static double bs_now_ms()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return 1e3 * ts.tv_sec + 1e-6 * ts.tv_nsec;
}
#define BS_MARK(tag) do { if (timing) { const double t_ = bs_now_ms(); fprintf(stderr, "[bs_build] %-12s %.3f ms\n", tag, t_ - tmark); tmark = t_; } } while (0)

// band or dense storage of an n x n system for ldlt_solve (the allocation rules of the whole-system path)
static int32_t nd_alloc_mat(BlockSys &bs, int64_t n, int64_t bw, LdltMat &A, double **d_A, double **work, bool no_twist)
{
    A = LdltMat{};
    A.n = n;
    A.no_twist = no_twist ? 1 : 0;
    if ((double)(bw + LVBA_NB + 64) < bs.band_frac * (double)n) {
        const int64_t ldab = bw + LVBA_NB + 64;
        A.ld = ldab - 1; A.bw = bw;
        const int64_t cnt = 2 * (ldab * (n + 1)) + 65 * ldab;
        TRY(bs_dmalloc(bs, d_A, cnt));
        HIPCHK(hipMemsetAsync(*d_A, 0, (size_t)cnt * sizeof(double), bs.stream)); // zeroed ONCE (ldlt_prepare_band_kernel)
    } else {
        A.ld = n; A.bw = n - 1;
        TRY(bs_dmalloc(bs, d_A, n * n + 64 * n + 128));
    }
    A.a = *d_A;
    TRY(bs_dmalloc(bs, work, ldlt_workspace_doubles(n, A.bw)));
    bs.nd_allocs.push_back(*d_A);
    bs.nd_allocs.push_back(*work);
    return LVBA_OK;
}
// device side of a dissection plan: per arc its matrix, workspace, border blocks and stream; the separator system
static int32_t nd_alloc(BlockSys &bs, const NdPlan &pl)
{
    NdSys &nd = bs.nd;
    nd.ps = pl.ps; nd.Ns = pl.Ns; nd.BbS = pl.BbS; nd.kind = pl.kind; nd.t_band = pl.t_band; nd.t_nd = pl.t_nd;
    const int P = (int)pl.arcs.size();
    int *stat = nullptr;
    TRY(bs_dmalloc(bs, &stat, P + 1));
    bs.nd_allocs.push_back(stat);
    HIPCHK(hipMemsetAsync(stat, 0, (size_t)(P + 1) * sizeof(int), bs.stream));
    nd.d_stat = stat; nd.statusS = stat + P;
    nd.arcs.resize((size_t)P);
    auto dm = [&](double **p, int64_t cnt) -> int32_t {
        TRY(bs_dmalloc(bs, p, cnt));
        bs.nd_allocs.push_back(*p);
        return LVBA_OK;
    };
    for (int a = 0; a < P; ++a) {
        const NdPlanArc &pa = pl.arcs[(size_t)a];
        NdArc &A = nd.arcs[(size_t)a];
        A = NdArc{};
        A.p0 = pa.p0; A.Na = pa.Na; A.nsep = (int32_t)pa.sep.size(); A.owner = pa.owner;
        A.n = 6 * (int64_t)pa.Na; A.ldb = ((6 * (int64_t)A.nsep + 63) / 64) * 64;
        A.status = stat + a;
        if (bs.distributed() && bs.n_ranks >= 2 && A.owner != bs.rank) continue; // another rank's arc: only its ranges are needed here
        if (!(bs.distributed() && bs.n_ranks >= 2)) A.owner = 0;
        TRY(nd_alloc_mat(bs, A.n, 6 * (int64_t)pa.Bb + 5, A.A, &A.d_A, &A.work, true));
        if (A.nsep > 0) {
            TRY(dm(&A.B, A.n * A.ldb)); TRY(dm(&A.Y, A.n * A.ldb)); TRY(dm(&A.Sa, A.ldb * A.ldb));
            TRY(dm(&A.wv, 2 * A.n)); TRY(dm(&A.gpart, ND_GS_SLICES * A.ldb));
            HIPCHK(hipMemsetAsync(A.B, 0, (size_t)(A.n * A.ldb) * sizeof(double), bs.stream)); // (the pad columns stay zero)
            HIPCHK(hipMemsetAsync(A.Y, 0, (size_t)(A.n * A.ldb) * sizeof(double), bs.stream));
            TRY(bs_dmalloc(bs, &A.sep, A.nsep));
            bs.nd_allocs.push_back(A.sep);
            HIPCHK(lvba::copy_h2d(A.sep, pa.sep.data(), (size_t)A.nsep * sizeof(int32_t)));
        }
        HIPCHK(lvba::StreamCache::get().acquire(&A.stream));
        HIPCHK(hipEventCreateWithFlags(&A.done, hipEventDisableTiming));
    }
    TRY(nd_alloc_mat(bs, 6 * (int64_t)nd.Ns, 6 * (int64_t)nd.BbS + 5, nd.AS, &nd.d_AS, &nd.workS, false));
    TRY(dm(&nd.Sblk, (int64_t)nd.Ns * (nd.BbS + 1) * 36 + 6 * (int64_t)nd.Ns + 8));
    TRY(dm(&nd.d_zero, 8));
    HIPCHK(hipMemsetAsync(nd.d_zero, 0, 8 * sizeof(double), bs.stream));
    HIPCHK(hipEventCreateWithFlags(&nd.start, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&nd.mid, hipEventDisableTiming));
    nd.active = true;
    return LVBA_OK;
}

int32_t bs_build(BlockSys &bs, int32_t N, int64_t G, const int64_t *voff, const int32_t *pidx)
{
    if (bs.built) return LVBA_OK;
    const bool timing = timing_on("build");
    double tmark = bs_now_ms();
    HIPCHK(hipSetDevice(bs.device));
    bs.N = N; bs.G = G; bs.F = voff[G];
    const int64_t F = bs.F;
    const int64_t n = 6 * (int64_t)N;
    bs.perm.resize(N);
    bs.iperm.resize(N);
    for (int i = 0; i < N; ++i) bs.perm[i] = bs.iperm[i] = i;
    auto band_of = [&](const lvba::hvec<int32_t> &iperm) {
        int32_t Bb = 0;
        for (int64_t a = 0; a < G; ++a) {
            int32_t lo = INT32_MAX, hi = -1;
            for (int64_t f = voff[a]; f < voff[a + 1]; ++f) {
                const int32_t p = iperm[pidx[f]];
                lo = std::min(lo, p); hi = std::max(hi, p);
            }
            if (hi >= 0) Bb = std::max(Bb, hi - lo);
        }
        return Bb;
    };
    int32_t Bb_nat = (bs.bb_hint >= 0 && !bs.distributed()) ? std::min<int32_t>(bs.bb_hint, std::max(N - 1, 0)) : band_of(bs.iperm);
    if (bs.distributed()) { // the store layout must agree on every rank: reduce over the global problem
        int32_t *dtmp = nullptr;
        HIPCHK(hipMalloc((void **)&dtmp, sizeof(int32_t)));
        HIPCHK(lvba::copy_h2d(dtmp, &Bb_nat, sizeof(int32_t)));
        TRY(bs_comm_allreduce(bs, dtmp, 1, ncclInt32, ncclMax));
        HIPCHK(hipStreamSynchronize(bs.stream));
        HIPCHK(lvba::copy_d2h(&Bb_nat, dtmp, sizeof(int32_t)));
        hipFree(dtmp);
    }
    bs.Bb = Bb_nat;
    int64_t Q = 0;
    for (int64_t a = 0; a < G; ++a) { const int64_t k = voff[a + 1] - voff[a]; Q += k * (k - 1) / 2; }
    bs.Q = Q;
    const bool small = (int64_t)N * N <= ((int64_t)1 << 29); // byte adjacency <= 512 MiB
    // systems of <= 1024 unknowns (window BA: 20 poses) are solved dense whatever the order: skip the graph work
    lvba::hvec<uint8_t> adj; // co-visibility of the GLOBAL problem (all-reduced), when it is computed at all
    NdPlan plan;
    if (bs.ordering == 1 && N > 2 && small && n > 1024) {
        adj.assign((size_t)N * N, 0);
        TRY(adjacency_build(bs.stream, G, voff, F, pidx, N, Q, adj.data())); // one thread per observer pair (pair_lists.hip)
        BS_MARK("adjacency");
        if (bs.distributed()) {
            uint8_t *dadj = nullptr;
            HIPCHK(hipMalloc((void **)&dadj, adj.size()));
            HIPCHK(lvba::copy_h2d(dadj, adj.data(), adj.size()));
            TRY(bs_comm_allreduce(bs, dadj, adj.size(), ncclUint8, ncclMax));
            HIPCHK(hipStreamSynchronize(bs.stream));
            HIPCHK(lvba::copy_d2h(adj.data(), dadj, adj.size()));
            hipFree(dadj);
        }
        lvba::hvec<int32_t> perm, iperm(N);
        rcm_order(adj, N, perm);
        for (int i = 0; i < N; ++i) iperm[perm[i]] = i;
        int32_t Bb_rcm = 0; // from the (global) adjacency so that all ranks agree
        for (int i = 0; i < N; ++i) {
            const uint8_t *row = adj.data() + (size_t)i * N;
            for (int j = 0; j < N; ++j)
                if (row[j]) Bb_rcm = std::max(Bb_rcm, std::abs(iperm[i] - iperm[j]));
        }
        if (Bb_rcm < Bb_nat) { bs.perm = perm; bs.iperm = iperm; bs.Bb = Bb_rcm; }
        // A graph that is not a narrow band (a hub on the ring), or a long band on several ranks: one level of nested dissection
        // (nd_plan.h: the cost model decides; every rank derives the same plan from the same all-reduced graph).  Not for grouped
        // problems (their groups are independent already) nor for the visual stage's SPD systems (bcr.hip).
        if (bs.n_groups == 0 && !bs.spd && N >= 256) { bs.perm_band = bs.perm; bs.Bb_band = bs.Bb; }
        if (bs.n_groups == 0 && !bs.spd && !solver_form("nond")) {
            NdPlan pl = nd_plan(adj.data(), N, bs.perm, bs.Bb, bs.distributed() ? bs.n_ranks : 1, solver_form("nd") ? 1e30 : 0.8);
            if (pl.active) {
                // What the dissected layout allocates on a rank -- the full lower block triangle of H (N^2 blocks), per owned arc its
                // matrix, the border columns B / Y and S_a, the separator's system: the cost model counts time only, and a graph with
                // a hub that fit as a band (N (Bb + 1) blocks) can be far larger this way.  Keep the band when the most loaded rank
                // would need more than half of the device's memory (LVBA_SOLVER=nd insists).  Every rank evaluates the same numbers
                // (all arcs, the device's TOTAL memory), so every rank decides alike.
                const int nr = bs.distributed() ? bs.n_ranks : 1;
                std::vector<double> per_rank((size_t)nr, 0.0);
                for (const NdPlanArc &a : pl.arcs) {
                    const double na = 6.0 * a.Na, bwa = std::min(6.0 * a.Bb + 5.0, na), ldb = 64.0 * std::ceil(6.0 * (double)a.sep.size() / 64.0);
                    per_rank[(size_t)std::min(std::max(a.owner, 0), nr - 1)] += 8.0 * (2.0 * (bwa + 128.0) * (na + 1.0) + 2.0 * na * ldb + ldb * ldb + 4.0 * na);
                }
                const double ns = 6.0 * pl.Ns, bws = std::min(6.0 * pl.BbS + 5.0, ns);
                const double need = 288.0 * (double)N * (double)N + *std::max_element(per_rank.begin(), per_rank.end()) +
                                    8.0 * 2.0 * (bws + 128.0) * (ns + 1.0) + 288.0 * pl.Ns * (double)(pl.BbS + 1);
                size_t free_b = 0, total_b = 0;
                if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && need > 0.5 * (double)total_b && !solver_form("nd")) {
                    if (timing_on("build"))
                        fprintf(stderr, "[bs_build] dissection (%s, %zu arcs) would take %.1f GB of the device's %.1f GB: keeping the band\n", pl.kind,
                                pl.arcs.size(), need / 1e9, (double)total_b / 1e9);
                    pl.active = false;
                }
            }
            if (pl.active) {
                plan = pl;
                bs.perm = plan.perm;
                for (int i = 0; i < N; ++i) bs.iperm[bs.perm[i]] = i;
                bs.Bb = N - 1;
            }
        }
    }
    BS_MARK("ordering");
    const int64_t bw = 6 * (int64_t)bs.Bb + 5;
    bs.use_band = !plan.active && (double)(bw + LVBA_NB + 64) < bs.band_frac * (double)n;
    if (!bs.use_band) bs.Bb = N - 1; // full lower block triangle
    const int64_t Bb1 = (int64_t)bs.Bb + 1;
    bs.hblk_doubles = (int64_t)N * Bb1 * 36;
    {
        if (bs.distributed() && !adj.empty()) {
            lvba::hvec<int64_t> slots;
            for (int32_t J = 0; J < N; ++J) slots.push_back((int64_t)J * Bb1);
            for (int32_t i = 0; i < N; ++i)
                for (int32_t j = 0; j < i; ++j)
                    if (adj[(size_t)i * N + j]) {
                        int32_t I = bs.iperm[i], J = bs.iperm[j];
                        if (I < J) std::swap(I, J);
                        slots.push_back((int64_t)J * Bb1 + (I - J));
                    }
            std::sort(slots.begin(), slots.end());
            if (36 * (int64_t)slots.size() < (bs.hblk_doubles / 4) * 3) {
                bs.n_ar = (int64_t)slots.size();
                TRY(bs_dmalloc(bs, &bs.d_ar_slot, bs.n_ar));
                TRY(bs_dmalloc(bs, &bs.d_arbuf, 36 * bs.n_ar + 6 * (int64_t)N + 8));
                HIPCHK(lvba::copy_h2d(bs.d_ar_slot, slots.data(), (size_t)bs.n_ar * sizeof(int64_t)));
            }
        }
    }

    if (!bs.perm_band.empty()) bs.adj_keep.swap(adj); // (kept for lvba_balm_nd_model)
    { // pose-major (CSC) view + per-block group lists for the atomic-free assembly: built on the device (pair_lists.hip)
        TRY(bs_dmalloc(bs, &bs.d_csc_off, N + 1));
        TRY(bs_dmalloc(bs, &bs.d_group_of_pos, F));
        TRY(bs_dmalloc(bs, &bs.d_csc_f, F));
        TRY(bs_dmalloc(bs, &bs.d_pos_of, F));
        TRY(bs_dmalloc(bs, &bs.d_pairs, Q));
        DevBuf d_blk_of(bs.stream);
        HIPCHK(d_blk_of.alloc(4 * (size_t)std::max<int64_t>(F, 1)));
        TRY(csc_build(bs.stream, G, voff, F, pidx, N, bs.iperm.data(), bs.d_csc_f, bs.d_group_of_pos, bs.d_pos_of, bs.d_csc_off,
                      d_blk_of.as<int32_t>()));
        BS_MARK("csc");
        // Non-empty blocks, visited in 2-D TILES of the block matrix (8 x 8 poses): a tile's pairs touch only a slice of
        // 8 + 8 poses' Y segments (factors are sorted by voxel inside a segment), which stays L2-resident, whereas a
        // column-by-column sweep re-fetches every Y record ~k-1 times from HBM (measured 5.5 GB per pass at C3).
        // The Q-sized grouping itself is a device sort (pair_lists.hip).
        lvba::hvec<int64_t> blk_slot, blk_off;
        // Voxel windows of the pair lists: the pairs processed at about the same time should draw on a slice of Y that the
        // L2s can hold (8 x 4 MB; XCD x sweeps its own eighth of the windows).  A window of w consecutive voxels holds
        // w * F / G records of 144 bytes.  Measured at C3 (PMC, profiles/): 6 MB windows cut the L2 misses of the pair pass from
        // 88 M to 36 M per evaluation; smaller windows miss less still, but every (window, block) item costs a partial block
        // and the pass is latency-bound, not bandwidth-bound, by then.  Problems whose whole Y is small are not windowed.
        // LVBA_PAIR_WINDOW overrides (voxels per window; 0 = no windows, the plain block-major lists and the 16-lane kernel): the
        // tests use it to reach either pair kernel at test sizes.
        int64_t window_groups = 0;
        if (18 * 8 * F > ((int64_t)24 << 20)) window_groups = std::max<int64_t>(256, (((int64_t)6 << 20) / 144) * G / std::max<int64_t>(F, 1));
        // A grouped problem (the windows of the window stage as ONE handle) has few blocks with long lists and lives for a
        // handful of iterations: the windowed, length-sorted lists cost it more set-up (pair lists 8.1 -> 1.0 ms, tables
        // 4.1 -> 0.2 ms for 16 windows of 20 x 100 k points) than they can give back -- this was round 3's regression of the
        // window stage (2.15 -> 2.9 ms per window), measured back to 2.13 with the plain lists.
        if (bs.n_groups > 0) window_groups = 0;
        if (const char *e = getenv("LVBA_PAIR_WINDOW")) window_groups = atoll(e);
        // (the items of a window are laid out by length: a wavefront's ten items finish together)
        TRY(pair_lists_build(bs.stream, G, voff, F, d_blk_of.as<int32_t>(), bs.d_pos_of, N, (int32_t)Bb1, Q, window_groups,
                             LVBA_PAIR_CUT, bs.d_pairs, blk_slot, blk_off));
        BS_MARK("pairs");
        bs.nnzb = (int64_t)blk_slot.size();
        if (window_groups > 0) {
            // windows only pay when consecutive voxels are seen from neighbouring poses (a window then touches a band of the block
            // matrix).  With the voxels in an order unrelated to the poses every window touches every block: one partial block per
            // pair in the limit.  Then the plain block-major lists are the better ones.  (lvba_balm_create re-lays voxels that
            // come in no order by the first pose that sees them, so this is the fall-back for problems without such structure.
            // Forming the windows from voxel RANKS instead of re-laying the voxels was tried: 3.0 ms per C3 evaluation against 2.7 ms
            // for this fall-back and 2.5 ms after a re-layout -- a window's Y records have to be neighbours in memory, not just few.)
            lvba::hvec<int64_t> u(blk_slot);
            std::sort(u.begin(), u.end());
            const int64_t distinct = (int64_t)(std::unique(u.begin(), u.end()) - u.begin());
            if ((int64_t)blk_slot.size() > 24 * std::max<int64_t>(distinct, 1)) {
                window_groups = 0;
                TRY(pair_lists_build(bs.stream, G, voff, F, d_blk_of.as<int32_t>(), bs.d_pos_of, N, (int32_t)Bb1, Q, 0, 0,
                                     bs.d_pairs, blk_slot, blk_off));
            }
        }
        { // distinct blocks, not runs
            lvba::hvec<int64_t> u(blk_slot);
            std::sort(u.begin(), u.end());
            bs.nnzb = (int64_t)(std::unique(u.begin(), u.end()) - u.begin());
        }
        // work items of the pair pass.  One 16-lane group per block is right when there are many blocks (C3: 4e5 blocks of
        // ~60 pairs); with few blocks and long lists (window BA: 190 blocks x 2000 pairs) it leaves the chip empty, so lists
        // longer than `cut` pairs become several items whose partial blocks are summed afterwards.
        lvba::hvec<int64_t> item_off, item_dst, multi_off, multi_slot, multi_idx;
        {
            int64_t n_partial = 0;
            // windowed lists (large problems: many short (window, block) items) go to the column-per-lane kernel, cut at
            // LVBA_PAIR_CUT pairs; few blocks with long lists (window BA: 190 blocks x 2000 pairs) to the 16-lane kernel, cut so
            // that the chip is filled.
            bs.pair_col = window_groups > 0;
            group_pair_items(blk_slot, blk_off, bs.pair_col ? LVBA_PAIR_CUT : pair_cut_length(Q), (int64_t)N * Bb1, item_off, item_dst,
                             multi_off, multi_slot, multi_idx, n_partial); // host_tables.h
            bs.n_items = (int64_t)item_dst.size();
            bs.n_multi = (int64_t)multi_slot.size();
            if (n_partial) TRY(bs_dmalloc(bs, &bs.d_partial, 36 * n_partial));
            if (bs.n_multi) {
                TRY(bs_dmalloc(bs, &bs.d_multi_off, bs.n_multi + 1));
                TRY(bs_dmalloc(bs, &bs.d_multi_slot, bs.n_multi));
                TRY(bs_dmalloc(bs, &bs.d_multi_idx, (int64_t)multi_idx.size()));
                HIPCHK(lvba::copy_h2d(bs.d_multi_idx, multi_idx.data(), multi_idx.size() * sizeof(int64_t)));
                HIPCHK(lvba::copy_h2d(bs.d_multi_off, multi_off.data(), (size_t)(bs.n_multi + 1) * sizeof(int64_t)));
                HIPCHK(lvba::copy_h2d(bs.d_multi_slot, multi_slot.data(), (size_t)bs.n_multi * sizeof(int64_t)));
            }
        }
        // slices per block: enough workgroups to fill the chip, but >= ~256 factors per slice
        int64_t Ssz = (2048 + N - 1) / N;
        const int64_t avg = F / N;
        Ssz = std::min<int64_t>(Ssz, std::max<int64_t>(1, avg / 256));
        bs.S = (int32_t)std::max<int64_t>(1, std::min<int64_t>(Ssz, 64));
        TRY(bs_dmalloc(bs, &bs.d_Y, 18 * F));
        TRY(bs_dmalloc(bs, &bs.d_blk_off, bs.n_items + 1));
        TRY(bs_dmalloc(bs, &bs.d_blk_slot, bs.n_items));
        HIPCHK(lvba::copy_h2d(bs.d_blk_off, item_off.data(), (size_t)(bs.n_items + 1) * sizeof(int64_t)));
        if (bs.n_items) HIPCHK(lvba::copy_h2d(bs.d_blk_slot, item_dst.data(), (size_t)bs.n_items * sizeof(int64_t)));
    }
    BS_MARK("upload");
    TRY(bs_dmalloc(bs, &bs.d_perm, N));
    HIPCHK(lvba::copy_h2d(bs.d_perm, bs.perm.data(), (size_t)N * sizeof(int32_t)));
    TRY(bs_dmalloc(bs, &bs.d_hg, bs.hg_doubles()));
    TRY(bs_dmalloc(bs, &bs.d_dx, n));
    TRY(bs_dmalloc(bs, &bs.d_u, std::max<int64_t>(1, bs.n_groups)));
    if (bs.n_groups > 2) {
        if (bs.h_pin_u) hipHostFree(bs.h_pin_u);
        bs.h_pin_u = nullptr;
        HIPCHK(hipHostMalloc((void **)&bs.h_pin_u, (size_t)bs.n_groups * sizeof(double), hipHostMallocDefault));
    }
    TRY(bs_dmalloc(bs, &bs.d_status, 4));
    bs.A.n = n;
    if (plan.active) {
        TRY(nd_alloc(bs, plan));
    } else if (bs.use_band) {
        const int64_t ldab = bw + LVBA_NB + 64;
        bs.A.ld = ldab - 1; bs.A.bw = bw;
        TRY(bs_dmalloc(bs, &bs.d_A, 2 * (ldab * (n + 1)) + 65 * ldab)); // room for the second matrix of the twisted factorisation (+ slack: edge tiles read past the window, ldlt.hip)
        // zero ONCE: every solve rewrites the columns its two matrices use (ldlt_prepare_band_kernel); the rest must read as zero
        HIPCHK(hipMemsetAsync(bs.d_A, 0, (size_t)(2 * (ldab * (n + 1)) + 65 * ldab) * sizeof(double), bs.stream));
    } else {
        bs.A.ld = n; bs.A.bw = n - 1;
        TRY(bs_dmalloc(bs, &bs.d_A, n * n + 64 * n + 128)); // (+ slack: edge tiles read past the window, ldlt.hip)
    }
    bs.A.a = bs.d_A;
    if (!plan.active) TRY(bs_dmalloc(bs, &bs.d_work, ldlt_workspace_doubles(n, bs.A.bw)));
    {
        const char *e = getenv("LVBA_BCR");
        if (bs.spd && bs.use_band && bcr_applicable(N, bs.Bb) && !(e && !strcmp(e, "0")))
            TRY(bs_dmalloc(bs, &bs.d_bcr, bcr_workspace_doubles(N, bs.Bb)));
    }
    HIPCHK(hipMemset(bs.d_hg, 0, (size_t)bs.hg_doubles() * sizeof(double)));
    BS_MARK("solver");
    bs.built = true;
    return LVBA_OK;
}

// The launch sequence of one solve is static per BlockSys (3 kernels per 64-column panel + 1 for the backward pass), so it
// is captured once into a hipGraph and replayed; u is read from device memory.
static int32_t dist_sum_cb(void *ctx, double *dbuf, size_t count) { return bs_comm_allreduce(*static_cast<BlockSys *>(ctx), dbuf, count, ncclDouble, ncclSum); }
static int32_t dist_max_cb(void *ctx, int *dbuf) { return bs_comm_allreduce(*static_cast<BlockSys *>(ctx), dbuf, 1, ncclInt32, ncclMax); }
static int32_t solve_launches(BlockSys &bs)
{
    if (bs.nd.active) {
        LdltDist dd{bs.rank, bs.n_ranks, &bs, dist_sum_cb, dist_max_cb};
        return nd_solve(bs.nd, bs.Hblk(), bs.N, bs.g(), bs.d_u, bs.d_dx, bs.d_status, bs.stream, bs.distributed() && bs.n_ranks >= 2 ? &dd : nullptr);
    }
    if (bs.d_bcr) { bcr_solve(bs.Hblk(), bs.Bb, bs.N, bs.g(), bs.d_u, bs.d_dx, bs.d_bcr, bs.d_status, bs.stream); return LVBA_OK; }
    // multi-rank: the two ends of the band factorisation on ranks 0 and 1
    LdltDist dd{bs.rank, bs.n_ranks, &bs, dist_sum_cb, dist_max_cb};
    return ldlt_solve(bs.A, bs.Hblk(), bs.Bb, bs.N, bs.g(), bs.d_u, bs.d_dx, bs.d_work, bs.d_status, bs.stream,
                      bs.distributed() && bs.n_ranks >= 2 ? &dd : nullptr, bs.n_groups > 0 ? bs.d_grp_of_pose : nullptr);
}

static int32_t enqueue_solve_launches(BlockSys &bs);
int32_t bs_enqueue_solve(BlockSys &bs, double u)
{
    if (bs.n_groups > 0) { // a grouped problem solved with one damping value for all groups
        for (int32_t k = 0; k < bs.n_groups; ++k) bs.h_pin_u[k] = u;
        HIPCHK(hipMemcpyAsync(bs.d_u, bs.h_pin_u, (size_t)bs.n_groups * sizeof(double), hipMemcpyHostToDevice, bs.stream));
        bs.u_known = false;
        return enqueue_solve_launches(bs);
    }
    if (!(bs.u_known && bs.u_on_device == u)) { // the visual stage solves with u = 0 every time (its damping is in the system)
        bs.u_slot ^= 1; // two staging words: the previous copy may still be reading the other one
        bs.h_pin_u[bs.u_slot] = u;
        HIPCHK(hipMemcpyAsync(bs.d_u, bs.h_pin_u + bs.u_slot, sizeof(double), hipMemcpyHostToDevice, bs.stream));
        bs.u_on_device = u;
        bs.u_known = true;
    }
    return enqueue_solve_launches(bs);
}

int32_t bs_enqueue_solve_groups(BlockSys &bs, const double *u)
{
    if (bs.n_groups <= 0 || bs.d_bcr) return LVBA_ERR_STATE;
    HIPCHK(hipStreamSynchronize(bs.stream)); // the pinned staging buffer may still be read by the previous copy
    for (int32_t k = 0; k < bs.n_groups; ++k) bs.h_pin_u[k] = u[k];
    HIPCHK(hipMemcpyAsync(bs.d_u, bs.h_pin_u, (size_t)bs.n_groups * sizeof(double), hipMemcpyHostToDevice, bs.stream));
    bs.u_known = false;
    return enqueue_solve_launches(bs);
}

static int32_t enqueue_solve_launches(BlockSys &bs)
{
    // capture + instantiate costs about as much as a few eager solves of a small system: wait for the third solve, so
    // handles that live for one short refinement (window BA) never pay for it
    // no capture while several host threads drive the device (bs_graph_inhibit): with HIP 7.0 a capture in one thread is
    // invalidated by allocations / synchronous copies in another even in hipStreamCaptureModeThreadLocal
    if (bs.distributed() && bs.n_ranks >= 2) bs.graph_tried = true; // the solve has exchanges between the ranks in it
    // a dissected solve forks onto one stream per arc: not captured -- the graph executor ran the arcs' branches one after the
    // other (timeline, round 5: 9.6 ms as a graph against 8.3 ms eager for two arcs)
    if (bs.nd.active) bs.graph_tried = true;
    if (!bs.graph_tried && g_graph_inhibit.load() == 0 && ++bs.solve_calls >= 3) {
        bs.graph_tried = true;
        if (!getenv("LVBA_NO_GRAPH")) {
            (void)hipGetLastError();
            if (hipStreamBeginCapture(bs.stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                (void)solve_launches(bs);
                hipGraph_t gph = nullptr;
                if (hipStreamEndCapture(bs.stream, &gph) == hipSuccess && gph &&
                    hipGraphInstantiate(&bs.solve_exec, gph, nullptr, nullptr, 0) == hipSuccess) {
                    bs.solve_graph = gph;
                } else {
                    if (gph) hipGraphDestroy(gph);
                    bs.solve_exec = nullptr;
                }
            }
            (void)hipGetLastError(); // a failed capture falls back to eager launches
        }
    }
    if (bs.solve_exec) HIPCHK(hipGraphLaunch(bs.solve_exec, bs.stream));
    else TRY(solve_launches(bs));
    HIPCHK(hipGetLastError());
    return LVBA_OK;
}

// The dissection plan nd_plan.h would make of this system's graph on n_ranks ranks, and its cost model (seconds per solve)
int32_t bs_nd_model(BlockSys &bs, int32_t n_ranks, double *t_band, double *t_nd, int32_t *arcs, int32_t *sep_poses, int32_t *sep_bb,
                    int32_t *max_arc_poses, int32_t *max_arc_bb)
{
    if (bs.adj_keep.empty()) return LVBA_ERR_STATE;
    const NdPlan pl = nd_plan(bs.adj_keep.data(), bs.N, bs.perm_band, bs.Bb_band, n_ranks, 1e30);
    *t_band = pl.t_band; *t_nd = pl.active ? pl.t_nd : 0.0;
    *arcs = (int32_t)pl.arcs.size(); *sep_poses = pl.Ns; *sep_bb = pl.BbS;
    *max_arc_poses = *max_arc_bb = 0;
    for (const NdPlanArc &a : pl.arcs) { *max_arc_poses = std::max(*max_arc_poses, a.Na); *max_arc_bb = std::max(*max_arc_bb, a.Bb); }
    return LVBA_OK;
}

void bs_destroy(BlockSys &bs)
{
    hipSetDevice(bs.device);
    if (bs.stream) hipStreamSynchronize(bs.stream);
    if (bs.comm && g_rccl.CommDestroy) g_rccl.CommDestroy(bs.comm);
    if (bs.solve_exec) hipGraphExecDestroy(bs.solve_exec);
    if (bs.solve_graph) hipGraphDestroy(bs.solve_graph);
    void *ptrs[] = {bs.d_ar_slot, bs.d_arbuf, bs.d_multi_off, bs.d_multi_slot, bs.d_multi_idx, bs.d_partial, bs.d_perm, bs.d_csc_off, bs.d_blk_off, bs.d_blk_slot, bs.d_group_of_pos, bs.d_csc_f, bs.d_pos_of,
                    bs.d_pairs, bs.d_Y, bs.d_hg, bs.d_A, bs.d_work, bs.d_bcr, bs.d_dx, bs.d_u, bs.d_status};
    for (void *p : ptrs)
        if (p) DevicePool::get().free(p);
    for (void *p : bs.nd_allocs)
        if (p) DevicePool::get().free(p);
    for (NdArc &A : bs.nd.arcs) {
        if (A.stream) lvba::StreamCache::get().release(A.stream);
        if (A.done) hipEventDestroy(A.done);
    }
    if (bs.nd.start) hipEventDestroy(bs.nd.start);
    if (bs.nd.mid) hipEventDestroy(bs.nd.mid);
    if (bs.h_pin_u) hipHostFree(bs.h_pin_u);
    if (bs.stream) lvba::StreamCache::get().release(bs.stream);
    bs = BlockSys();
}

} // namespace lvba
