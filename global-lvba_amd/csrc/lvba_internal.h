// lvba_internal.h -- shared declarations between the HIP translation units of liblvba_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define LVBA_CF 256   // max factors per chunk == workgroup size of the BALM kernels
#define LVBA_CV 128   // max voxels per chunk (every voxel has >= 2 factors)
#define LVBA_NB 64    // LDL^T panel width
#define LVBA_PAIR_CUT 32 // longest pair list one lane of balm_pair_lane_kernel walks

namespace lvba {

// Device view of one packed BALM problem (a rank's voxel shard).
struct BalmDev {
    int32_t n_poses;
    int32_t band_blocks;       // pose-block half bandwidth Bb of the Hessian store (solver order)
    int64_t V, F, n_chunks;
    const int64_t *voff;       // [V+1] CSR offsets (local, voff[0] == 0)
    const int32_t *pidx;       // [F] pose of each factor, SOLVER order
    const double *clu;         // [10][F] SoA cluster statistics
    const int64_t *chunk_v0;   // [n_chunks+1] first voxel of each chunk
    // pose-major ("CSC") view used by the deterministic Hessian assembly
    int32_t S;                 // each pose's factor segment is split over S workgroups
    const int64_t *csc_off;    // [N+1] factor positions of each pose (solver order), sorted by voxel
    const double *clu_csc;     // [10][F] cluster statistics in pose-major order
    const int32_t *vox_of_pos; // [F] voxel of the factor at each pose-major position
    double *vrec;              // [V][16] per-voxel records (13 used; 128-byte stride)
    double *Y;                 // [F][18] per-factor Y_i, pose-major positions
    double *part;              // [N*S][32] per-workgroup partial sums of (D[21], g[6])
};

// Per-block contributor lists of the atomic-free assembly of  -sum Y_I Y_J^T  (shared by both stages).
struct PairDev {
    int64_t nnzb;              // work items of the pair pass: off-diagonal blocks, long pair lists cut into several items
    const int64_t *blk_off;    // [nnzb+1] offsets into pairs
    const int64_t *blk_slot;   // [nnzb] >= 0: block slot in the block-band store (the item is the whole block);
                               //        < 0: -(1 + index into `partial`) (the block is summed from several items)
    const int2 *pairs;         // [Q] (position of block I's factor, position of block J's factor), I > J
    const double *Y;           // [F][18] per-factor Y, pose-major positions
    double *partial;           // [n_partial][36] partial blocks of the cut lists
    int64_t n_multi;           // blocks assembled from several items
    const int64_t *multi_off;  // [n_multi+1] their ranges in multi_idx
    const int64_t *multi_slot; // [n_multi] their slots in the store
    const int64_t *multi_idx;  // indices into `partial`, in summation order
    int32_t col_form;          // 1: balm_pair_col_kernel (short items of windowed lists), 0: balm_pair_staged_kernel,
                               // 2: balm_pair_col_kernel on fp32 Y records of 80 bytes (LVBA_Y32=1; LiDAR handles only)
};

// Device view of one packed visual problem (cameras in solver order; only landmarks with a valid plane).
struct VisDev {
    int32_t M, S, band_blocks, fixed_cam;
    int64_t Ta, O;
    const int64_t *off;            // [Ta+1] CSR offsets of each landmark's observations
    const int32_t *cam;            // [O] camera (solver order) of each observation
    const int32_t *track_of_obs;   // [O]
    const double *uv;              // [O][2]
    const double *uv_cm;           // [O][2] the same in camera-major order (position t of the CSC lists): vis_cam_kernel
    const double *plane;           // [Ta][4]
    double intr[8];
    double inv_sig_px, inv_sig_pl;
    // linearisation at the current point
    double *Jc, *Jp, *r;           // [O][12], [O][6], [O][2]
    double *rpl, *Jpl;             // [Ta], [Ta][3]
    double *sc_cam, *sc_pt;        // [M][6], [Ta][3] Jacobi column scales
    double *Lp, *zp, *step_p;      // [Ta][6], [Ta][3], [Ta][3]
    // camera-major order (BlockSys)
    const int64_t *csc_off;
    const int32_t *csc_f, *group_of_pos, *pos_of;
    double *Y, *part;              // [O][18], [M*S][40]
    // track shards over several ranks (cameras replicated): per-camera sums that need the other ranks' tracks before they are
    // used -- camsum [12 M] = diag(Jc^T Jc) | Jc^T r (LM diagonal, gradient max), colsum [6 M] (Jacobi scaling)
    int32_t dist, count_cams;      // dist: 1 when sharded; count_cams: this rank counts the (replicated) cameras in the step norms
    double *camsum, *colsum;
};


// Working matrix of the damped system, lower triangle, column-major with leading dimension ld:
// A(r,c) = a[r + c*ld].  Dense: ld = n.  Band: LAPACK lower-band storage with ldab = ld+1, i.e.
// A(r,c) = ab[(r-c) + c*ldab]; valid offsets 0 <= r-c <= ld.  bw = half bandwidth in scalars.
struct LdltMat {
    double *a;
    int64_t n, ld, bw;
    int32_t no_twist; // 1: plain top-down factorisation whatever the shape (the arcs of nd_solve.hip: their factors are reused)
};

// LVBA_SOLVER = comma-separated forms the damped solve is FORCED into (tests reach every form at test sizes that way):
//   notwist  plain top-down band factorisation (what bands too short for two ends take)
//   bulk64   64 x 64 update tiles with 64-bit pointers (what matrices of 4 GB and more take)
//   nd       nested dissection (ldlt_nd.h) whenever a partition exists, whatever the cost model says;  nond: never
//   nodefer  the row roles' full form in the two-ended phases as well (ldlt_schedule.h: the deferred form is their default)
inline bool solver_form(const char *name)
{
    const char *e = getenv("LVBA_SOLVER");
    if (!e) return false;
    const size_t n = strlen(name);
    for (const char *p = e; *p;) {
        const char *q = strchr(p, ',');
        const size_t len = q ? (size_t)(q - p) : strlen(p);
        if (len == n && !strncmp(p, name, n)) return true;
        p += len + (q ? 1 : 0);
    }
    return false;
}

// LVBA_TIMING = comma-separated parts whose stage times go to stderr: build (bs_build, finalize, create), window (window_ba.hip),
// vis (the visual refinement's phases); "1" or "all": every part.
inline bool timing_on(const char *part, bool named_only = false) // named_only: the part must be listed itself ("1" / "all" do not count)
{
    const char *e = getenv("LVBA_TIMING");
    if (!e || !*e) return false;
    const size_t n = strlen(part);
    for (const char *p = e; *p;) {
        const char *q = strchr(p, ',');
        const size_t len = q ? (size_t)(q - p) : strlen(p);
        if (len == n && !strncmp(p, part, n)) return true;
        if (!named_only && ((len == 3 && !strncmp(p, "all", 3)) || (len == 1 && *p == '1'))) return true;
        p += len + (q ? 1 : 0);
    }
    return false;
}

// One level of nested dissection (nd_plan.h decides, ldlt_nd.h solves): device-side pieces of an arc and of the whole system
struct NdArc {
    int32_t p0, Na;     // pose range [p0, p0 + Na) of the solver order
    int32_t nsep;       // separator poses the arc touches
    int32_t owner;      // rank that factorises it
    int64_t n, ldb;     // 6 Na; row stride of B / Y: 6 nsep rounded up to 64
    LdltMat A;          // its band (or dense) matrix, no_twist
    double *d_A;        // (allocation behind A.a)
    double *work;       // ldlt_solve's workspace: G, d, b
    double *B, *Y;      // [n][ldb] row-major: E_a^T as the forward substitution leaves it / L^-1 E_a^T
    double *Sa;         // [ldb][ldb], column j at Sa + j ldb: lower tiles of Y^T D^-1 Y
    double *wv;         // [2][n] G^T b, panel by panel (= D^-1 L^-1 b); 1 / d
    double *gpart;      // [ND_GS_SLICES][ldb] partial sums of Y^T wv
    int32_t *sep;       // [nsep] separator-local pose indices, ascending (device)
    int *status;
    hipStream_t stream;
    hipEvent_t done;
};
struct NdSys {
    bool active = false;
    int32_t ps = 0, Ns = 0, BbS = 0; // separator: solver positions [ps, ps + Ns), half-bandwidth of its system in pose blocks
    std::vector<NdArc> arcs;
    LdltMat AS{};                    // the separator system
    double *d_AS = nullptr, *workS = nullptr;
    double *Sblk = nullptr;          // [Ns (BbS + 1) 36 | 6 Ns]: its block store and gradient (one all-reduce covers both)
    double *d_zero = nullptr;        // a zero on the device: the separator system is damped when it is built
    int *d_stat = nullptr, *statusS = nullptr; // [arcs + 1]
    hipEvent_t start = nullptr, mid = nullptr;
    const char *kind = "band";
    double t_band = 0.0, t_nd = 0.0;
};
#define ND_GS_SLICES 16

// balm_kernels.hip
void launch_cost(const BalmDev &d, const double *poses, double *chunk_cost, double *out, hipStream_t s,
                 hipEvent_t k0, hipEvent_t k1, bool with_records = false);
// zero_first: clear the whole store first (needed when other ranks' blocks were reduced into it)
void launch_eval(const BalmDev &d, const PairDev &pd, const double *poses, double *Hblk, int64_t hblk_doubles, double *g,
                 double *chunk_cost, double *out, bool zero_first, hipStream_t s, hipEvent_t k0, hipEvent_t k1,
                 bool skip_voxel_pass = false);
void launch_pairs(const PairDev &pd, double *Hblk, hipStream_t s);
void launch_aos_to_soa(const double *aos, const int32_t *fmap, int64_t F, double *soa, hipStream_t s);
// grouped refinement (lvba_balm_refine_groups)
void launch_reduce_chunks_groups(const double *chunk_cost, const int64_t *gco, int n_groups, double *out, hipStream_t s);
void launch_cost_chunks(const BalmDev &d, const double *poses, double *chunk_cost, hipStream_t s);
void launch_predicted_decrease_groups(const double *Hblk, int band_blocks, const double *g, const double *dx, const double *u,
                                      const int32_t *gpo, int n_groups, double *out, hipStream_t s);
void launch_select_poses(double *cur, const double *trial, const int32_t *accept, const int32_t *grp_of_pose, int n_poses, hipStream_t s);
void launch_gather_csc(const double *clu, const int32_t *csc_f, int64_t F, double *clu_csc, hipStream_t s);
void launch_retract(const double *poses, const double *dx, double *out, int n_poses, hipStream_t s);
void launch_predicted_decrease(const double *Hblk, int band_blocks, const double *g, const double *dx, double u,
                               int64_t n, double *out, hipStream_t s);
void launch_lm_report(const double *scal2, const double *q1_part, int n_q1, const double *cost_cur, const int *status, double *host_pin,
                      hipStream_t s);
int launch_retract_q1(const double *poses, const double *dx, double *out, int n_poses, const double *Hblk, int band_blocks, const double *g,
                      double u, double *q1_part, hipStream_t s); // returns the number of q1 shares written
void launch_export_dense(const double *Hblk, int band_blocks, int n_poses, const int *perm, double *Hd, hipStream_t s);
void launch_export_vec(const double *v, const int *perm, int n_poses, double *out, hipStream_t s);
void launch_import_poses(const double *in, const int *perm, int n_poses, double *out, hipStream_t s);
void launch_export_poses(const double *in, const int *perm, int n_poses, double *out, hipStream_t s);

// visual_kernels.hip
void vis_launch_residuals(const VisDev &d, bool jac, const double *qc, const double *tc, const double *Xp, double *part,
                          double *cost_out, hipStream_t s);
void vis_launch_colnorms(const VisDev &d, hipStream_t s);          // single rank: sums + scaling in one go
void vis_launch_colsums(const VisDev &d, hipStream_t s);           // sharded: landmark scaling + per-camera column sums -> colsum
void vis_launch_colnorm_finish(const VisDev &d, hipStream_t s);    //          (after the all-reduce of colsum) camera scaling
void vis_launch_cam_finish(const VisDev &d, double radius, double min_diag, double max_diag, double *Hblk, const double *qc, unsigned long long *gmax,
                           hipStream_t s);                         // sharded, after the all-reduces: LM diagonal, gradient max
void vis_launch_gather_uv(const VisDev &d, double *uv_cm, hipStream_t s);
void vis_launch_reduced_system(const VisDev &d, const PairDev &pd, const double *qc, const double *tc, const double *Xp, double radius, double min_diag, double max_diag, double *Hblk,
                               int64_t hblk_doubles, double *g, unsigned long long *gmax, bool zero_first, hipStream_t s);
void vis_launch_step_and_trial(const VisDev &d, const double *step_c, const double *qc, const double *tc, const double *Xp, double *qc2,
                               double *tc2, double *Xp2, double *part, double *scal, const unsigned long long *gmax, const int *status,
                               double *host_pin, hipStream_t s);
void vis_launch_back(const VisDev &d, const double *step_c, const double *qc, const double *tc, const double *Xp, double *part,
                     double *model_out, hipStream_t s);
void vis_launch_apply(const VisDev &d, const double *step_c, const double *qc, const double *tc, const double *Xp, double *qc2,
                      double *tc2, double *Xp2, double *part, double *norms_out, hipStream_t s);

// ldlt.hip
// Workspace doubles needed by ldlt_solve for an n x n system.
int64_t ldlt_workspace_doubles(int64_t n, int64_t bw);
int64_t ldlt_num_panels(int64_t n);
// panels each end of the two-ended ("twisted") band factorisation eliminates; 0: plain top-down
int64_t ldlt_twist_panels(int64_t n, int64_t ld, int64_t bw);
// where ldlt_solve keeps the pieces of a (plain top-down) factorisation in its workspace: G [panels][64 x 64] (G = L11^-T D^-1 of
// every diagonal block, [m][c] row-major), d [n], and the right-hand side b [n] -- after the forward substitution b holds, panel
// by panel, the rows as the earlier panels' updates left them (L11^-1 not yet applied: c_k = D_k G_k^T b_k)
inline double *ldlt_work_G(double *work) { return work; }
inline double *ldlt_work_d(int64_t n, double *work) { return work + ((n + LVBA_NB - 1) / LVBA_NB) * 4096; }
inline double *ldlt_work_b(int64_t n, double *work) { return ldlt_work_d(n, work) + n; }
enum { LDLT_ALL = 0, LDLT_FACTOR = 1, LDLT_BACKWARD = 2 }; // ldlt_solve's `phase`: everything / fill + factorise + forward / backward only
// A <- H + u*diag(H) from the block-band store, b <- -g; then unpivoted blocked LDL^T of the lower
// triangle and the two triangular solves.  x (length n) receives the solution; status[0] != 0 on a
// zero / non-finite pivot.  u is read from device memory (u_dev) so the launch sequence is static and
// can be captured into a hipGraph.
// dist (may be NULL): a multi-rank job -- the two ends of the twisted factorisation are eliminated by rank 0 and rank 1 (the
// callbacks all-reduce device buffers in place over the ranks; they return 0 on success).
// Extra right-hand sides carried through a plain top-down factorisation (the border columns of a dissected system, ldlt_nd.h):
// B [n][ldb] row-major is forward-substituted in place (its rows end as the earlier panels' updates left them: L11^-1 not applied,
// like ldlt_solve's own b), Y [n][ldb] receives L^-1 B.  The work rides in the factorisation's own launches, one panel behind.
struct LdltBorder {
    double *B, *Y;
    int64_t ldb; // a multiple of 64
};
struct LdltDist {
    int rank, n_ranks;
    void *ctx;
    int32_t (*allreduce_sum)(void *ctx, double *dbuf, size_t count);
    int32_t (*allreduce_max_i32)(void *ctx, int *dbuf);
};
struct LdltDist;
int32_t nd_solve(NdSys &nd, const double *Hblk, int32_t N, const double *g, const double *u_dev, double *x, int *status, hipStream_t s,
                 const LdltDist *dist);
int32_t ldlt_solve(const LdltMat &A, const double *Hblk, int band_blocks, int n_poses, const double *g,
                   const double *u_dev, double *x, double *work, int *status, hipStream_t s, const LdltDist *dist = nullptr,
                   const int32_t *grp = nullptr, // grp: pose block -> entry of u_dev (grouped refinement)
                   int phase = LDLT_ALL,         // LDLT_FACTOR / LDLT_BACKWARD: the two halves of a solve (single rank, A.no_twist)
                   const LdltBorder *border = nullptr);

// bcr.hip: block cyclic reduction for narrow-band SPD systems (the visual stage's reduced camera system)
bool bcr_applicable(int n_poses, int band_blocks);
int64_t bcr_workspace_doubles(int n_poses, int band_blocks);
void bcr_solve(const double *Hblk, int band_blocks, int n_poses, const double *g, const double *u_dev, double *x, double *work,
               int *status, hipStream_t s);

} // namespace lvba
