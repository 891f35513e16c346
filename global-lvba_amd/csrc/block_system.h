// block_system.h -- the part of the solver that both stages share: a set of N 6-dof blocks (LiDAR poses or
// cameras), F factors each attached to one block and grouped into G groups (voxels / landmark tracks), and the
// symmetric block system   H = blockdiag(.) - sum_groups Y Y^T   that couples the blocks of a group.
// BlockSys owns: the fill-reducing block ordering, the pose-major ("CSC") factor order, the per-block pair lists
// of the atomic-free assembly, the block-band store [H | g | scalars], the LDL^T workspace + captured hipGraph,
// and the RCCL communicator.  The factor payload (clusters, observations) stays with the caller.
#pragma once
#include "host_arena.h"
#include <memory>
#include <vector>
#include "lvba_common.h"
#include "lvba_internal.h"
#include "mempool.h"

namespace lvba {

struct BlockSys {
    int device = 0;
    hipStream_t stream = nullptr;
    int solve_calls = 0;
    int32_t N = 0;
    int64_t G = 0, F = 0, Q = 0;
    // configuration
    int ordering = 1;
    double band_frac = 0.6;
    bool spd = false; // the caller promises a symmetric POSITIVE DEFINITE system (visual stage): narrow bands go to bcr.hip
    // ordering / layout
    lvba::hvec<int32_t> perm, iperm; // perm[internal] = caller, iperm[caller] = internal
    int32_t Bb = 0;
    bool use_band = false, built = false;
    int32_t *d_perm = nullptr;
    // pose-major order + pair lists
    int32_t S = 1;        // slices per block for the per-factor reduction
    int64_t nnzb = 0;
    int64_t *d_csc_off = nullptr, *d_blk_off = nullptr, *d_blk_slot = nullptr;
    bool pair_col = false;            // which pair kernel the item lists were cut for
    bool y32 = false;                 // LVBA_Y32=1 on a LiDAR handle whose lists use the column kernel: fp32 Y records (experiment)
    int64_t n_items = 0, n_multi = 0; // work items of the pair pass (>= nnzb), blocks cut into several items
    int64_t *d_multi_off = nullptr, *d_multi_slot = nullptr, *d_multi_idx = nullptr;
    double *d_partial = nullptr;
    int32_t *d_group_of_pos = nullptr, *d_csc_f = nullptr, *d_pos_of = nullptr;
    int2 *d_pairs = nullptr;
    double *d_Y = nullptr; // [F][18]
    // block-band store
    double *d_hg = nullptr; // [Hblk | g | scal(8)] contiguous: one all-reduce covers all
    int64_t hblk_doubles = 0;
    // solver
    LdltMat A{};
    NdSys nd;                // one level of nested dissection, when that is the solver (nd_plan.h / ldlt_nd.h)
    std::vector<void *> nd_allocs;
    // what nd_plan.h was given (kept for lvba_balm_nd_model: "what would n ranks do with this graph"); empty for small systems
    lvba::hvec<uint8_t> adj_keep;
    lvba::hvec<int32_t> perm_band;
    int32_t Bb_band = 0;
    double *d_bcr = nullptr; // workspace of the block cyclic reduction, when that is the solver
    double *d_A = nullptr, *d_work = nullptr, *d_dx = nullptr, *d_u = nullptr, *h_pin_u = nullptr;
    double u_on_device = 0.0;      // the damping value d_u[0] holds (u_known): an unchanged value is not uploaded again
    bool u_known = false;
    int u_slot = 0;
    // grouped refinement: d_u holds one damping value per group, d_grp_of_pose [N] (owned by the caller) maps pose blocks to them
    int32_t n_groups = 0;
    const int32_t *d_grp_of_pose = nullptr;
    int32_t bb_hint = -1; // >= 0: an upper bound of the block half-bandwidth in the caller's pose order (a grouped problem: the
                          // largest group's pose count - 1; the voxels were checked against the groups) -- no host pass over the factors
    int *d_status = nullptr;
    hipGraph_t solve_graph = nullptr;
    hipGraphExec_t solve_exec = nullptr;
    bool graph_tried = false;
    // distributed
    int n_ranks = 1, rank = 0;
    ncclComm_t comm = nullptr;
    lvba_allreduce_fn ext_allreduce = nullptr; // the caller's transport instead of RCCL (bs_dist_init_external)
    void *ext_ctx = nullptr;
    bool distributed() const { return comm != nullptr || ext_allreduce != nullptr; }
    // packed all-reduce: slots of the blocks that are non-zero on ANY rank (+ the diagonal), and the staging buffer
    int64_t n_ar = 0, *d_ar_slot = nullptr;
    double *d_arbuf = nullptr;
    int64_t device_bytes = 0;

    double *Hblk() const { return d_hg; }
    double *g() const { return d_hg + hblk_doubles; }
    double *scal() const { return d_hg + hblk_doubles + 6 * (int64_t)N; }
    int64_t hg_doubles() const { return hblk_doubles + 6 * (int64_t)N + 8; }
    PairDev pair_dev() const
    {
        PairDev p;
        p.nnzb = n_items; p.blk_off = d_blk_off; p.blk_slot = d_blk_slot; p.pairs = d_pairs; p.Y = d_Y;
        p.partial = d_partial; p.n_multi = n_multi; p.multi_off = d_multi_off; p.multi_slot = d_multi_slot; p.multi_idx = d_multi_idx; p.col_form = pair_col ? (y32 ? 2 : 1) : 0;
        return p;
    }
};

template <typename T>
int32_t bs_dmalloc(BlockSys &bs, T **p, int64_t count)
{
    // through the caching pool: short-lived handles (one per window in the window-BA stage) would otherwise spend more
    // time in hipMalloc / hipFree than in kernels
    HIPCHK(DevicePool::get().alloc((void **)p, (size_t)(count > 0 ? count : 1) * sizeof(T)));
    bs.device_bytes += count * (int64_t)sizeof(T);
    return LVBA_OK;
}

// Creates the stream(s).  Call once, before bs_dist_init / bs_build.
int32_t bs_init(BlockSys &bs, int device);
// voff [G+1] (0-based) and pidx [F] (caller block indices) are host arrays; groups may have any size >= 0.
int32_t bs_build(BlockSys &bs, int32_t N, int64_t G, const int64_t *voff, const int32_t *pidx);
// enqueue: dx = -(H + u diag H)^-1 g on bs.stream (u = 0: H is used as assembled)
int32_t bs_enqueue_solve(BlockSys &bs, double u);
// grouped form: u [bs.n_groups] (host); bs.n_groups / bs.d_grp_of_pose are set before the system is built
int32_t bs_enqueue_solve_groups(BlockSys &bs, const double *u);
// all-reduce `count` doubles in place over the ranks (no-op without a communicator)
int32_t bs_allreduce(BlockSys &bs, double *buf, size_t count);
// all-reduce [Hblk | g | cost] over the ranks: only the blocks of the union sparsity pattern travel when that is known
int32_t bs_allreduce_hg(BlockSys &bs);
// clear what the last all-reduce left in the store of a multi-rank job (before the next evaluation writes this rank's blocks)
int32_t bs_clear_reduced(BlockSys &bs);
// all-reduce `count` elements of a device buffer in place (sum or max; double / int64 / int32 / uint8) over the ranks
int32_t bs_comm_allreduce(BlockSys &bs, void *dbuf, size_t count, ncclDataType_t dt, ncclRedOp_t op);
// slots of the structurally non-zero blocks of the block-band store (ascending, diagonal included); their download [n][36]
int32_t bs_pattern_slots(BlockSys &bs, lvba::hvec<int64_t> &slots);
int32_t bs_download_blocks(BlockSys &bs, const int64_t *slots, int64_t n, double *out);
int32_t bs_dist_init(BlockSys &bs, int32_t n_ranks, int32_t rank, const char uid[128], int64_t *group_count_inout);
int32_t bs_dist_init_external(BlockSys &bs, int32_t n_ranks, int32_t rank, lvba_allreduce_fn fn, void *ctx, int64_t *group_count_inout);
int32_t bs_nd_model(BlockSys &bs, int32_t n_ranks, double *t_band, double *t_nd, int32_t *arcs, int32_t *sep_poses, int32_t *sep_bb,
                    int32_t *max_arc_poses, int32_t *max_arc_bb);
void bs_destroy(BlockSys &bs);
// +1 / -1 around sections in which several host threads use the library concurrently: solve graphs are not captured then
void bs_graph_inhibit(int delta);

} // namespace lvba
