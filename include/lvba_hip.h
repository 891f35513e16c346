/* lvba_hip.h -- C-ABI of liblvba_hip.so: the MI355X (gfx950) replacement for the LM-refinement
 * hot path of xuankuzcr/Global-LVBA.  Plain pointers and sizes only; no C++/torch types.
 *
 * Reference interfaces replaced (paths relative to the reference repo):
 *   lvba_balm_cost     <- BALM2::only_residual            include/BALM/bavoxel.hpp:641-648
 *                         (VOX_HESS::evaluate_only_residual  bavoxel.hpp:176-203)
 *   lvba_balm_eval     <- BALM2::divide_thread            include/BALM/bavoxel.hpp:597-639
 *                         (VOX_HESS::acc_evaluate2           bavoxel.hpp:68-174)
 *   lvba_balm_refine   <- BALM2::damping_iter             include/BALM/bavoxel.hpp:662-767
 *                         call sites src/lvba_system.cpp:264 (window BA) and :386 (global BA)
 *   lvba_balm_create   <- VOX_HESS::push_voxel / plvec_voxels (bavoxel.hpp:35,45-54): the packed
 *                         CSR copy of every admitted voxel's non-empty PointCluster slots
 *   lvba_balm_dist_*   <- the 16-way thread sum of bavoxel.hpp:626-633, as an RCCL all-reduce
 * The reference has no refine() symbol (SURVEY.md "five facts" #1); refine here is the name
 * BASELINE.json uses for damping_iter.
 *
 * Conventions (kept from the reference):
 *   pose      = T_world<-body, 12 doubles: R row-major (9) then p (3)          tools.hpp:147-207
 *   tangent   = [dtheta(3), dp(3)] per pose; retraction R*Exp(dtheta), p+dp   bavoxel.hpp:722-727
 *   cluster   = 10 doubles: Pxx Pxy Pxz Pyy Pyz Pzz vx vy vz n  (P = sum p p^T, v = sum p, n = #points,
 *               body frame)                                                     tools.hpp:407-466
 *   factor    = one non-empty (voxel, pose) cluster slot; a voxel needs >= 2   bavoxel.hpp:45-54
 *   cost      = sum over voxels of lambda_min(cov of the merged, transformed clusters); the *_avg
 *               variants divide by the number of voxels (AVG_THR, bavoxel.hpp:11,634-635)
 *   H, g      = exact Hessian / gradient of the (un-averaged) cost, 6N x 6N / 6N
 *
 * Ownership: create() copies and repacks everything it is given onto the device; caller memory is
 * never referenced after a call returns.  The caller owns pose arrays and output buffers.
 * Threading: one caller thread per handle; calls are synchronous.  Handles are independent.
 * Errors: 0 = ok, < 0 = usage/runtime error, > 0 = numerical condition; never throws.
 * lvba_last_error() returns a thread-local message for the last non-zero status.
 * All arithmetic is IEEE fp64.  There is no CPU fallback: without a HIP device every entry point
 * that needs one returns LVBA_ERR_DEVICE.
 */
#ifndef LVBA_HIP_H
#define LVBA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LVBA_OK 0
#define LVBA_ERR_ARG (-1)          /* bad argument / shape */
#define LVBA_ERR_DEVICE (-2)       /* HIP runtime error or no device */
#define LVBA_ERR_NOMEM (-3)
#define LVBA_ERR_UNSUPPORTED (-4)
#define LVBA_ERR_DIST (-5)         /* RCCL error */
#define LVBA_ERR_STATE (-6)        /* call sequence error (e.g. lm_step before lm_begin) */
#define LVBA_NUM_FACTORIZATION 1   /* zero / non-finite pivot in LDL^T (reference: unchecked, bavoxel.hpp:707) */
#define LVBA_NUM_NONFINITE 2       /* non-finite cost */

typedef struct lvba_balm_s *lvba_balm_t;

/* LM options; lvba_balm_default_opts fills the reference's hard-coded values. */
typedef struct {
    int32_t max_iter;   /* 10   bavoxel.hpp:686 (rejected steps count) */
    int32_t reserved;
    double u0;          /* 0.01 bavoxel.hpp:664 */
    double v0;          /* 2    bavoxel.hpp:664 */
    double rel_tol;     /* 1e-6 bavoxel.hpp:760 */
} lvba_balm_opts;

/* One LM iteration, the quantities of the commented printf at bavoxel.hpp:737. */
typedef struct {
    int32_t iter;
    int32_t accepted;    /* q > 0 */
    int32_t evaluated;   /* H/g recomputed at the start of this iteration (is_calc_hess) */
    int32_t status;      /* LVBA_OK or LVBA_NUM_* for this iteration */
    double residual1;    /* averaged cost at the current poses */
    double residual2;    /* averaged cost at the trial poses */
    double u, v;         /* damping state used for this solve */
    double q;            /* residual1 - residual2 */
    double q1;           /* predicted decrease, averaged */
} lvba_lm_trace;

typedef struct {
    int32_t n_poses;
    int32_t n_ranks;
    int64_t n_voxels;        /* local shard */
    int64_t n_voxels_global;
    int64_t n_factors;       /* local shard */
    int64_t n_pairs;         /* local pose-pair contributions, sum k(k-1)/2 */
    int64_t n_chunks;        /* workgroup work items */
    int64_t n_blocks;        /* off-diagonal pose blocks (I>J) with at least one contributing voxel */
    int32_t band_blocks;     /* pose-block half bandwidth after ordering */
    int32_t use_band;        /* 1 = band LDL^T, 0 = dense */
    int64_t hess_bytes;      /* block-band Hessian storage */
    int64_t device_bytes;    /* total device memory held by the handle */
    int64_t allreduce_bytes; /* bytes all-reduced per evaluation (0 without a communicator): the union-pattern blocks + g + cost */
    int32_t twist_panels;    /* band solver: 64-column panels each END eliminates (0: plain top-down factorisation) */
    int32_t solve_ranks;     /* ranks the factorisation is spread over: 2 when ranks 0 and 1 take one end each, else 1 */
    int32_t trial_linearised; /* 1: the LM loop costs its trial point with the voxel pass of the evaluation (cost + voxel records),
                               * and an accepted step's next evaluation starts from those records (LiDAR handles only: 0 in
                               * lvba_visual_info) */
    int32_t y_fp32;           /* 1: LVBA_Y32=1 took effect -- the per-factor Y records travel as fp32 between the factor and the
                               * pair pass (an experiment: off-diagonal pose blocks then carry ~1e-7 relative rounding) */
    /* The damped solve by one level of nested dissection (csrc/nd_plan.h, csrc/ldlt_nd.h) instead of one band: a co-visibility
     * graph with a hub (a place crossed many times), or a long band on several ranks.  nd_kind 0: no (band / dense as above). */
    int32_t nd_kind;          /* 1: hub separator, 2: chunks of the band ordering */
    int32_t nd_arcs;          /* independent band systems (over all ranks) */
    int32_t nd_sep_poses;     /* poses of the separator system */
    int32_t nd_sep_band_blocks; /* its pose-block half bandwidth */
    double nd_model_band_ms, nd_model_nd_ms; /* the plan's cost model: seconds -> ms per solve, band against dissection */
} lvba_balm_info_t;

/* Accumulated device times (HIP events on the handle's stream) since the last reset, ms. */
typedef struct {
    double cost_ms;     int64_t cost_calls;      /* cost-only kernel (+ reduction) */
    double eval_ms;     int64_t eval_calls;      /* H/g/cost evaluation kernels */
    double solve_ms;    int64_t solve_calls;     /* damped LDL^T + triangular solves */
    double reduce_ms;   int64_t reduce_calls;    /* RCCL all-reduce */
    double cost_kernel_ms;  /* dominant kernel only: balm_cost_kernel */
    double eval_kernel_ms;  /* dominant kernel only: balm_eval_kernel */
} lvba_prof_t;

int32_t lvba_version(void);
const char *lvba_last_error(void);
int32_t lvba_device_count(void);

void lvba_balm_default_opts(lvba_balm_opts *opts);

/* Contiguous voxel range of rank r of G: [floor(V*r/G), floor(V*(r+1)/G)), the formula of
 * bavoxel.hpp:621-624 with thread -> GPU. */
void lvba_shard_range(int64_t n_voxels, int32_t rank, int32_t n_ranks, int64_t *head, int64_t *end);

/* Pack a problem (or one rank's voxel shard of it) onto HIP device `device`.
 *   voxel_off [n_voxels+1]  CSR offsets into pose_idx/clusters (voxel_off[0] may be non-zero: the
 *                           arrays are indexed by voxel_off[v]-voxel_off[0])
 *   pose_idx  [F]           observing pose of each factor, in [0, n_poses), distinct inside a voxel
 *   clusters  [F][10]
 * Every voxel must have >= 2 factors (push_voxel's admission rule). */
int32_t lvba_balm_create(int32_t n_poses, int64_t n_voxels, const int64_t *voxel_off,
                         const int32_t *pose_idx, const double *clusters, int32_t device,
                         lvba_balm_t *out);
/* The same with the clusters [F][10] already on `device` (a hipMalloc'ed / HIP-visible pointer, indexed like the host array
 * relative to voxel_off[0]): what the voxel front-end of this library hands over without a host round trip, and what a caller
 * that builds its clusters on the GPU would use.  voxel_off / pose_idx stay host arrays.  The device array is only read during
 * the call. */
int32_t lvba_balm_create_dev(int32_t n_poses, int64_t n_voxels, const int64_t *voxel_off, const int32_t *pose_idx,
                             const double *d_clusters, int32_t device, lvba_balm_t *out);
int32_t lvba_balm_destroy(lvba_balm_t h);

/* Optional, before the first cost/eval/refine call: pose ordering for the linear solver
 * (1 = reverse Cuthill-McKee on the pose co-visibility graph [default], 0 = caller order) and the
 * band/dense switch: the band LDL^T is used when (half-bandwidth + 128) < band_frac * 6N
 * (default 0.6), otherwise the dense one.  Results do not depend on either beyond rounding. */
int32_t lvba_balm_configure(lvba_balm_t h, int32_t ordering, double band_frac);
int32_t lvba_balm_info(lvba_balm_t h, lvba_balm_info_t *info);

/* only_residual: cost at `poses` [N][12]; is_avg divides by the (global) voxel count. */
int32_t lvba_balm_cost(lvba_balm_t h, const double *poses, int32_t is_avg, double *cost);

/* divide_thread: H [6N*6N] (symmetric, so row/col-major agree), g [6N], averaged cost.
 * H and g may be NULL (kept on the device for lvba_balm_solve). */
int32_t lvba_balm_eval(lvba_balm_t h, const double *poses, double *H, double *g, double *cost_avg);

/* The same evaluation with H in SPARSE form, for problems whose dense matrix does not fit (10 000 poses: 28.8 GB): the
 * STRUCTURALLY non-zero 6x6 pose blocks of the lower triangle in the CALLER's pose order -- bi[k] >= bj[k], every unordered pose
 * pair that shares a voxel once (on a voxel shard without a union pattern: every slot of the band), the diagonal blocks in full
 * --, blocks[k][6 r + c] = H[6 bi[k] + r][6 bj[k] + c].  The set does not depend on the poses: it is the same on every call (the
 * sizing call and the filling call agree), and a block of it may be numerically zero.  *n_blocks receives the number of blocks;
 * call with capacity 0 (arrays may be NULL) to size the arrays.  g [6 n_poses] and cost_avg may be NULL. */
int32_t lvba_balm_eval_blocks(lvba_balm_t h, const double *poses, int64_t capacity, int32_t *bi, int32_t *bj, double *blocks,
                              int64_t *n_blocks, double *g, double *cost_avg);

/* Solve (H + u*diag(H)) dx = -g with the H, g of the last lvba_balm_eval (bavoxel.hpp:692-710). */
int32_t lvba_balm_solve(lvba_balm_t h, double u, double *dx);

/* damping_iter: refines poses in place.  trace may be NULL; at most opts->max_iter rows. */
int32_t lvba_balm_refine(lvba_balm_t h, double *poses_inout, const lvba_balm_opts *opts,
                         lvba_lm_trace *trace, int32_t *n_trace);

/* The same loop, one iteration per call (what bench.py times).  *done is set when the reference
 * loop would exit (max_iter reached or the bavoxel.hpp:760 test). */
int32_t lvba_balm_lm_begin(lvba_balm_t h, const double *poses, const lvba_balm_opts *opts);
int32_t lvba_balm_lm_step(lvba_balm_t h, lvba_lm_trace *row, int32_t *done);
int32_t lvba_balm_lm_end(lvba_balm_t h, double *poses_out);

/* Several INDEPENDENT refinements in one handle, advanced in lock-step -- the windows of LvbaSystem::runWindowBA
 * (src/lvba_system.cpp:232-302: one damping_iter per window, one after the other).  Group k owns the poses
 * [pose_off[k], pose_off[k+1]) and the voxels [voxel_off[k], voxel_off[k+1]); every factor of its voxels must be seen from
 * one of its poses.  The Hessian is block diagonal, so ONE evaluation, ONE band factorisation (damping per group) and ONE
 * cost pass per LM iteration serve all groups, while each group keeps the LM state of damping_iter (u, v, accept / reject,
 * the bavoxel.hpp:760 exit) for itself: group k's poses are what lvba_balm_refine gives for group k alone, up to rounding.
 *   lvba_balm_set_groups     after lvba_balm_create, before the first cost / eval / refine call (single rank only)
 *   lvba_balm_refine_groups  n_iter / status / cost_first / cost_last [n_groups] may be NULL; status[k] = LVBA_OK or
 *                            LVBA_NUM_NONFINITE.  Returns LVBA_NUM_FACTORIZATION (poses untouched) if a pivot of the joint
 *                            factorisation broke down: the groups are not independent then, refine them one by one. */
int32_t lvba_balm_set_groups(lvba_balm_t h, int32_t n_groups, const int32_t *pose_off, const int64_t *voxel_off);
int32_t lvba_balm_refine_groups(lvba_balm_t h, double *poses_inout, const lvba_balm_opts *opts, int32_t *n_iter,
                                int32_t *status, double *cost_first, double *cost_last);

/* Profiling (HIP events around the stages, on the stream the kernels are launched on). */
int32_t lvba_balm_set_profiling(lvba_balm_t h, int32_t enable);
int32_t lvba_balm_get_profile(lvba_balm_t h, lvba_prof_t *out, int32_t reset);

/* Pose ordering used internally: perm[internal] = caller pose index (n_poses entries). */
int32_t lvba_balm_get_ordering(lvba_balm_t h, int32_t *perm);

/* What the damped solve would look like on n_ranks ranks: the nested-dissection plan csrc/nd_plan.h makes of this problem's
 * co-visibility graph (arcs dealt out over the ranks, separator system) and its cost model in ms per solve, next to the band
 * factorisation's (two ranks at best).  A MODEL, in the solver's own measured units (launches on the serial chain x us per
 * launch, flops / sustained rate); nd_ms = 0: no partition exists.  Needs >= 256 poses. */
typedef struct {
    int32_t n_ranks, arcs, sep_poses, sep_band_blocks, max_arc_poses, max_arc_band_blocks;
    double band_ms, nd_ms;
} lvba_nd_model_t;
int32_t lvba_balm_nd_model(lvba_balm_t h, int32_t n_ranks, lvba_nd_model_t *out);

/* Multi-GPU (one process per GPU): factors are sharded by voxel range across ranks; every eval
 * all-reduces {block-band H, g, cost} and every cost pass all-reduces one double, over RCCL.
 * uid is an ncclUniqueId (128 bytes) created on rank 0 and distributed by the caller. */
int32_t lvba_dist_unique_id(char uid[128]);
int32_t lvba_balm_dist_init(lvba_balm_t h, int32_t n_ranks, int32_t rank, const char uid[128]);
/* The same with the CALLER'S transport instead of RCCL (an MPI job that already owns a communicator; a test harness that runs
 * several ranks as host threads on one device, where RCCL refuses a duplicate GPU -- tests/host_transport.cpp).  `fn` must
 * all-reduce `count` elements of type `dtype` of the DEVICE buffer `device_buf` in place over the n_ranks ranks with `op`,
 * ordered after the work already enqueued on `hip_stream` (a hipStream_t) and visible to work enqueued on it afterwards;
 * every rank must obtain bitwise the same result (sum in a fixed rank order), as RCCL guarantees for its own.  It returns 0
 * on success.  The library calls it from the thread that calls the library, never concurrently for one handle. */
#define LVBA_DT_F64 0
#define LVBA_DT_I64 1
#define LVBA_DT_I32 2
#define LVBA_DT_U8 3
#define LVBA_OP_SUM 0
#define LVBA_OP_MAX 1
typedef int32_t (*lvba_allreduce_fn)(void *ctx, void *device_buf, size_t count, int32_t dtype, int32_t op, void *hip_stream);
int32_t lvba_balm_dist_init_external(lvba_balm_t h, int32_t n_ranks, int32_t rank, lvba_allreduce_fn fn, void *ctx);

/* ===================================================================================================
 * Visual stage: replaces the ceres::Problem ... ceres::Solve region of LvbaSystem::optimizeCameraPoses
 * (src/lvba_system.cpp:1571-1665) with the cost functors of include/utils.hpp:51-147.
 *   camera   = T_cam<-world: q [w,x,y,z] (memory order of src/lvba_system.cpp:1516) + t; camera 0 is held
 *              constant (:1582-1583); quaternions move on ceres::EigenQuaternionManifold applied to that memory
 *              as the reference does (:1579) -- see DESIGN.md
 *   landmark = X_w [3]; a landmark whose `valid` flag is 0 (no plane found, :1598-1603) is dropped together with
 *              its reprojection observations and is returned unchanged
 *   residuals: per observation the whitened Brown-Conrady reprojection error (2), per landmark the whitened
 *              point-to-plane distance sqrt(s^2+1e-12)/sigma (1); no loss function (:1630,1639)
 *   solver:    Levenberg-Marquardt trust region with Ceres 2.1 defaults restated (Jacobi scaling, LM diagonal
 *              clamp(diag)/radius, radius schedule, tolerances), landmark blocks eliminated (Schur), reduced
 *              camera system solved by the same LDL^T as the LiDAR stage.
 * =================================================================================================== */
typedef struct lvba_visual_s *lvba_visual_t;

typedef struct {
    int32_t max_iter;               /* 50     src/lvba_system.cpp:1573 */
    int32_t reserved;
    double initial_radius;          /* 1e4    Ceres default initial_trust_region_radius */
    double max_radius;              /* 1e16 */
    double min_radius;              /* 1e-32 */
    double min_relative_decrease;   /* 1e-3 */
    double min_lm_diagonal;         /* 1e-6 */
    double max_lm_diagonal;         /* 1e32 */
    double function_tolerance;      /* 1e-6 */
    double gradient_tolerance;      /* 1e-10 */
    double parameter_tolerance;     /* 1e-8 */
} lvba_visual_opts;

#define LVBA_TERM_NO_CONVERGENCE 0  /* max_iter reached */
#define LVBA_TERM_FUNCTION 1
#define LVBA_TERM_PARAMETER 2
#define LVBA_TERM_GRADIENT 3
#define LVBA_TERM_RADIUS 4
#define LVBA_TERM_FAILURE 5         /* linear solver failed repeatedly / non-finite cost */

/* one row per iteration, iteration 0 = the initial evaluation (like Ceres' progress table) */
typedef struct {
    int32_t iter;
    int32_t accepted;     /* step successful (always 1 for iteration 0) */
    int32_t valid;        /* linear solve produced a finite step with positive model decrease */
    int32_t reserved;
    double cost;          /* 1/2 sum r^2: the new cost if accepted, the rejected candidate's cost otherwise */
    double cost_change;
    double step_norm;
    double radius;        /* trust-region radius after this iteration's update */
    double rho;           /* relative decrease (step quality) */
    double gradient_max_norm;
} lvba_visual_trace;

void lvba_visual_default_opts(lvba_visual_opts *opts);

/* obs_off [n_tracks+1] CSR offsets of each landmark's (de-duplicated, inlier) observations; obs_cam [O] in
 * [0,n_cams); obs_uv [O][2] pixels; plane [n_tracks][4] = (n, d); valid [n_tracks]; intr = fx fy cx cy k1 k2 p1 p2
 * (already scaled, src/dataset_io.cpp:59-62); sigma_px = 0.5, sigma_plane = 0.01 upstream (:1590-1591). */
int32_t lvba_visual_create(int32_t n_cams, int64_t n_tracks, const int64_t *obs_off, const int32_t *obs_cam,
                           const double *obs_uv, const double *plane, const uint8_t *valid, const double intr[8],
                           double sigma_px, double sigma_plane, int32_t device, lvba_visual_t *out);
int32_t lvba_visual_destroy(lvba_visual_t h);

/* Sizes of the packed problem in the fields of lvba_balm_info_t that apply: n_poses = cameras, n_voxels = landmarks with a
 * plane (the active ones), n_factors = their observations, n_pairs, n_blocks (off-diagonal camera blocks of the reduced
 * system), band_blocks, use_band, hess_bytes, device_bytes. */
int32_t lvba_visual_info(lvba_visual_t h, lvba_balm_info_t *info);

/* Multi-GPU: landmark tracks are sharded over the ranks (any partition; contiguous ranges as for the voxels), the cameras are
 * replicated.  Every rank creates its handle from ITS tracks (X, plane, valid and the observation arrays of that shard) and
 * calls this before the first cost / linearize / refine call.  Per LM iteration the ranks all-reduce the reduced camera system
 * [S | rhs] (only the blocks of the union pattern), 12 M per-camera sums (the LM diagonal and the gradient test need the
 * whole diag(Jc^T Jc), Jc^T r) and five scalars; the reduced system is solved on every rank.  All ranks return bitwise the same
 * cameras and trace; X holds the rank's own landmarks.  uid as for lvba_balm_dist_init. */
int32_t lvba_visual_dist_init(lvba_visual_t h, int32_t n_ranks, int32_t rank, const char uid[128]);
int32_t lvba_visual_dist_init_external(lvba_visual_t h, int32_t n_ranks, int32_t rank, lvba_allreduce_fn fn, void *ctx);

/* 1/2 sum r^2 over the residuals of the active landmarks at (q [M][4], t [M][3], X [n_tracks][3]). */
int32_t lvba_visual_cost(lvba_visual_t h, const double *q, const double *t, const double *X, double *cost);

/* Linearise at the given point with trust-region radius `radius` (Jacobi scaling taken from THIS Jacobian, as at
 * Ceres' iteration 0): the reduced camera system S [6M x 6M] (symmetric; camera 0's block is decoupled) and its
 * right-hand side rhs [6M] in the scaled tangent variables, caller camera order.  For tests / inspection. */
int32_t lvba_visual_linearize(lvba_visual_t h, const double *q, const double *t, const double *X, double radius,
                              double *S, double *rhs, double *cost);

/* The solve: refines q, t, X in place (quaternions re-normalised on write-back, :1651-1657). */
int32_t lvba_visual_refine(lvba_visual_t h, double *q, double *t, double *X, const lvba_visual_opts *opts,
                           lvba_visual_trace *trace, int32_t trace_cap, int32_t *n_trace, int32_t *termination);

/* ---- voxel front-end: raw scans -> the packed LiDAR-BA problem, on the device ---------------------------------------
 * Replaces the construction that precedes every damping_iter call (src/lvba_system.cpp:247-258, :365-377, :1498-1506):
 *   lvba_voxmap_build       <- cut_voxel (include/BALM/bavoxel.hpp:799-836) for every frame, then, for every root,
 *                              OCTO_TREE_NODE::recut (:391-464, judge_eigen :335-352)
 *   lvba_voxmap_to_balm     <- OCTO_TREE_NODE::tras_opt (:466-474) into VOX_HESS::push_voxel (:45-54)
 *   lvba_voxmap_find_planes <- recompute_local_planes (src/lvba_system.cpp:1531-1565) +
 *                              OCTO_TREE_NODE::findCorrespondPoint (bavoxel.hpp:320-333)
 * Kept from the reference: fp32 points promoted to double; root key = C truncation of a FLOAT quotient that had 1
 * subtracted when negative; float voxel centres / quarter lengths; octant test `double > float`; a node with fewer
 * than min_points points is dropped, one whose lambda_min/lambda_max exceeds eigen_ratio[layer] is split (dropped at
 * layer 2); PointCluster sums accumulate in cloud order (results are bit-identical to PointCluster::push).
 * Voxels come out sorted by (root key x,y,z, octant path); the reference's unordered_map order is unspecified. */
typedef struct lvba_voxmap_s *lvba_voxmap_t;
typedef struct {
    double voxel_size;      /* root voxel edge (stage1_root_voxel_size_ / stage2_root_voxel_size_) */
    float eigen_ratio[4];   /* eigen_ratio_array (bavoxel.hpp:17, include/dataset_io.h:77,80); [3] unused as upstream */
    int32_t min_points;     /* min_ps = 15 (bavoxel.hpp:24) */
    int32_t layer_limit;    /* must be 2 (bavoxel.hpp:13) */
} lvba_voxel_opts;
typedef struct {
    int64_t n_points, n_roots, n_planes; /* points hashed; root voxels; PLANE nodes (admitted or not) */
    int64_t n_voxels, n_factors;         /* admitted voxels (>= 2 observing frames) and their cluster slots */
    /* host wall clock of the build phases, ms (each ends on a stream synchronise): host->device copy of the clouds
     * (lvba_voxmap_build only), key kernel, root sort + gather + root table, octree walk (count), octree walk (write) */
    double upload_ms, key_ms, sort_ms, count_ms, write_ms;
} lvba_voxmap_info_t;
void lvba_voxel_default_opts(lvba_voxel_opts *opts);

/* frame_points[f] -> host memory of frame f's cloud: frame_count[f] points, x,y,z as the first three floats of every
 * point_stride_bytes (12 for packed xyz, sizeof(pcl::PointXYZINormal) = 48 for the reference's PointType);
 * poses [n_frames][12].  Fails with LVBA_ERR_ARG on a non-finite point. */
int32_t lvba_voxmap_build(int32_t device, int32_t n_frames, const void *const *frame_points,
                          const int64_t *frame_count, int32_t point_stride_bytes, const double *poses,
                          const lvba_voxel_opts *opts, lvba_voxmap_t *out);
int32_t lvba_voxmap_destroy(lvba_voxmap_t h);

/* The same, in two steps, for callers that voxelise the same clouds more than once (the reference re-cuts the anchor
 * clouds for stage 1, stage 2 and the visual stage, src/lvba_system.cpp:365-377,1498-1506, and the raw scans window by
 * window, :232-258): lvba_scans_create copies the clouds to the device once; lvba_voxmap_build_scans builds the map of
 * frames [frame_begin, frame_begin + n_frames) at poses [n_frames][12] (pose / frame indices in the outputs are
 * relative to frame_begin, as cut_voxel's fnum is relative to the window). */
typedef struct lvba_scans_s *lvba_scans_t;
int32_t lvba_scans_create(int32_t device, int32_t n_frames, const void *const *frame_points,
                          const int64_t *frame_count, int32_t point_stride_bytes, lvba_scans_t *out);
int32_t lvba_scans_destroy(lvba_scans_t scans);
int32_t lvba_voxmap_build_scans(lvba_scans_t scans, int32_t frame_begin, int32_t n_frames, const double *poses,
                                const lvba_voxel_opts *opts, lvba_voxmap_t *out);

int32_t lvba_voxmap_info(lvba_voxmap_t h, lvba_voxmap_info_t *info);
/* The front-end keeps its device workspaces in a per-process cache between maps (hipMalloc/hipFree of 100 MB-class
 * buffers would otherwise dominate small maps); this returns the cached bytes to the driver.  Returns bytes freed. */
int64_t lvba_release_cached_memory(void);
/* Host copies of the admitted voxels in lvba_balm_create's layout (any pointer may be NULL): voxel_off [V+1],
 * pose_idx [F], clusters [F][10], voxel_key [V][4] = root key x, y, z and layer | o1 << 4 | o2 << 8. */
int32_t lvba_voxmap_export(lvba_voxmap_t h, int64_t *voxel_off, int32_t *pose_idx, double *clusters,
                           int64_t *voxel_key);
/* The VOX_HESS of this map (poses = the n_frames of the build), ready for lvba_balm_refine. */
int32_t lvba_voxmap_to_balm(lvba_voxmap_t h, lvba_balm_t *out);
/* plane [n][4] = (unit normal, d = -n.centre) and valid [n] for n world points X [n][3]; invalid -> zeros. */
int32_t lvba_voxmap_find_planes(lvba_voxmap_t h, int64_t n, const double *X, double *plane, uint8_t *valid);

/* ---- track -> landmark initialisation (the step before the visual solve) ------------------------------------------------
 *   lvba_triangulate_tracks <- TriangulateTrackDLT (src/lvba_system.cpp:50-111) + ComputeMeanReproj (:8-48) for every track:
 *   DLT on the undistorted normalised pixels (undistortPixelToNormalized, include/utils.hpp:207-233: 8 fixed-point
 *   iterations of the Brown-Conrady model), smallest eigenvector of the 4x4 A^T A, X = Xh.xyz / Xh.w, then the mean pixel
 *   error of X over the track.  Tracks are CSR (obs_off [n+1] from 0, obs_cam [O], obs_uv [O][2]), one observation per
 *   image as the reference's selected_ids holds them; Rcw [M][9] row-major and tcw [M][3] are T_cam<-world.
 *   ok[i] = 1 iff the track has >= 4 observations, >= 8 DLT rows, |Xh.w| >= 1e-12, finite X and >= 4 valid reprojections
 *   (mean_reproj = +inf otherwise; the caller applies its own reproj_mean_thr_px_ as at :1143-1146). */
int32_t lvba_triangulate_tracks(int32_t device, int32_t n_cams, int64_t n_tracks, const int64_t *obs_off,
                                const int32_t *obs_cam, const double *obs_uv, const double *Rcw, const double *tcw,
                                const double intr[8], double *X, double *mean_reproj, int32_t *count, uint8_t *ok);

/* ---- LiDAR-assisted landmark initialisation: depth images + per-track fusion ---------------------------------------------
 *   lvba_depth_render <- LvbaSystem::buildGridMapFromOptimized (src/lvba_system.cpp:1266-1338) + generateDepthWithVoxel
 *   (:835-919): every scan point in the world frame (scan_poses [n_frames][12], T_world<-body of the REFINED trajectory) is
 *   hashed into voxel_size (reference: 0.5 m) voxels; image m sees the voxels touched by the scans with
 *   |scan_time - image_time| <= half_window_s (reference: 0.5 s; scan_times ascending; the reference passes the image id
 *   through std::to_string, i.e. rounds it to 1e-6 s -- the caller does that) and ALL map points of those voxels are
 *   z-buffered through camera m (Rcw [n][9], tcw [n][3], intr = fx fy cx cy k1 k2 p1 p2): pixel (int)u, (int)v, Z < 1e-3
 *   skipped, smallest (float)Z wins, 0 = no return.  Images stay on the device.
 *   lvba_depth_upload makes a handle from host images [n][height][width] (depth from another source).
 *   lvba_fuse_tracks <- the per-component part of LvbaSystem::BuildTracksAndFuse3D (:1000-1225): tracks are the BFS
 *   components in CSR form (obs_off [n+1] from 0, obs_img [O], obs_uv [O][2] float keypoint coordinates, observations in BFS
 *   order, several per image allowed).  Per track: the depth-fused candidate (fetchDepthBilinear, back-projection through the
 *   distortion model, 0.12 m consistency with the first valid observation, first observation per image, greedy view-angle
 *   filter, mean reprojection error), the triangulation candidate (DLT over the first observation of every image, the same
 *   filter around the seed, DLT again over >= 4 kept observations), then the reference's selection.  depth may be NULL
 *   (triangulation candidate only).  status[t] = 0 dropped / 1 triangulated / 2 depth-fused; X [n][3]; mean_reproj [n]
 *   (+inf when dropped); kept [O] = the track's inlier_indices as a mask.  Where the reference iterates a
 *   std::unordered_map<int,int> of images (unique_id :994, best_id :1051, kept_id_depth :1064), images are visited in the order libstdc++ gives that
 *   container for the reference's reserve() and insertion sequence (bucket = key mod the rehash policy's prime, a new node
 *   goes to the front of its bucket and a new bucket to the front of the list: csrc/tracks_device.h umap_order), because
 *   the greedy view-angle filter's survivors depend on it. */
typedef struct lvba_depth_s *lvba_depth_t;
int32_t lvba_depth_render(lvba_scans_t scans, const double *scan_poses, const double *scan_times, int32_t n_images,
                          const double *image_times, const double *Rcw, const double *tcw, const double intr[8],
                          int32_t width, int32_t height, double half_window_s, double voxel_size, lvba_depth_t *out);
int32_t lvba_depth_upload(int32_t device, int32_t n_images, int32_t width, int32_t height, const float *depth,
                          lvba_depth_t *out);
int32_t lvba_depth_info(lvba_depth_t depth, int32_t *n_images, int32_t *width, int32_t *height);
int32_t lvba_depth_download(lvba_depth_t depth, int32_t image, float *out);
void lvba_depth_destroy(lvba_depth_t depth);

typedef struct lvba_fuse_opts {
    int32_t obser_thr;          /* minimum observations / images per track (config obser_thr, default 3) */
    int32_t reserved;
    double min_view_angle_deg;  /* greedy view-angle filter (config min_view_angle_deg, default 8) */
    double reproj_mean_thr_px;  /* acceptance threshold of a candidate's mean reprojection error (default 3) */
} lvba_fuse_opts;
void lvba_fuse_default_opts(lvba_fuse_opts *opts);
int32_t lvba_fuse_tracks(int32_t device, lvba_depth_t depth, int32_t n_images, const double *Rcw, const double *tcw,
                         const double intr[8], int64_t n_tracks, const int64_t *obs_off, const int32_t *obs_img,
                         const float *obs_uv, const lvba_fuse_opts *opts, uint8_t *status, double *X, double *mean_reproj,
                         uint8_t *kept);

/* ---- window BA: raw scans + odometry -> anchor frames --------------------------------------------------------------
 *   lvba_window_ba <- LvbaSystem::runWindowBA  src/lvba_system.cpp:204-310 : for every window of window_size frames the voxel
 *   map at the odometry poses (:247-257), the "fewer than 3 plane voxels per frame -> skip" rule (:258-262), damping_iter
 *   (:264), re-alignment of the optimised window to its first odometry pose (:268-279), relative poses to the anchor and the
 *   merged cloud in the anchor frame with fp32 write-back (:284-299, pl_transform include/BALM/tools.hpp:385-395) and
 *   down_sampling_voxel2 (tools.hpp:300-359; survivors come out sorted by voxel key, upstream order is unspecified).
 * Outputs follow the reference's members: rel_poses [n][12] = rel_poses_to_anchor_ (identity for frames of skipped windows),
 * anchor_index [n] = anchor_index_per_frame_ (-1 when skipped), anchor_poses [<= ceil(n/w)][12] = odometry pose of each
 * window's first frame, anchor_scans = the merged, down-sampled clouds as a device-resident scan set (destroy with
 * lvba_scans_destroy), window_poses [n][12] (may be NULL) = the optimised poses before re-alignment, win_info
 * [ceil(n/w)] (may be NULL). */
typedef struct {
    int32_t window_size;    /* window_ba_size_ (10; config.yaml 20) */
    int32_t use_rel;        /* use_window_ba_rel_ */
    double anchor_leaf;     /* anchor_leaf_size_ (0.1; config.yaml 0.01); < 0.001 disables the down-sampling */
    lvba_voxel_opts voxel;  /* stage1_root_voxel_size_ and the eigen_ratio_array in effect (bavoxel.hpp:17 until a stage sets it) */
    lvba_balm_opts lm;
    int32_t merge_only;     /* 1: no map, no LM, no skip rule -- every window is merged at the given poses (rel = anchor^-1 o pose) and
                               down-sampled: the anchor clouds optimizeCameraPoses rebuilds from the refined poses (:1464-1487) */
    int32_t lm_mode;        /* 0: the damping_iter of all windows in lock-step as one grouped problem (lvba_balm_refine_groups;
                               default), 1: one window at a time.  The windows are independent either way; results agree to
                               rounding */
} lvba_window_opts;
typedef struct {
    int32_t start, n_frames, skipped, anchor; /* anchor = index into anchor_poses / anchor_scans, -1 if skipped */
    int32_t n_iter, lm_status;
    int64_t n_voxels, n_factors, n_anchor_points;
    double cost_first, cost_last;             /* averaged LiDAR cost before / after the window's damping_iter */
    double map_ms, solve_ms, merge_ms;        /* host wall clock: voxel map, problem set-up + LM (lm_mode 0: the joint problem's
                                                 time shared out evenly over its windows), anchor merge + down-sampling */
    double setup_ms;                          /* the part of solve_ms before the first LM iteration */
} lvba_window_info;
void lvba_window_default_opts(lvba_window_opts *opts);
int32_t lvba_window_ba(lvba_scans_t scans, const double *poses, const lvba_window_opts *opts, double *window_poses,
                       double *rel_poses, int32_t *anchor_index, double *anchor_poses, int32_t *n_anchors,
                       lvba_scans_t *anchor_scans, lvba_window_info *win_info);
/* The window stage over several GPUs of one node (or several shares on one GPU): the windows are independent problems
 * (src/lvba_system.cpp:232 solves them one after the other), so share k -- scans[k], on whatever device it was created on --
 * holds a contiguous run of whole windows of the sequence (every share but the last a multiple of window_size frames;
 * lvba_window_split gives the frame ranges: frame_begin [n_shares + 1], the thread split of bavoxel.hpp:621-624 applied to
 * windows).  One host thread per share runs lvba_window_ba on it; outputs are those of lvba_window_ba for the concatenated
 * sequence, in window order (poses [n][12] covers all frames); the anchor scan set lives on scans[0]'s device. */
int32_t lvba_window_split(int32_t n_frames, int32_t window_size, int32_t n_shares, int32_t *frame_begin);
int32_t lvba_window_ba_multi(int32_t n_shares, const lvba_scans_t *scans, const double *poses, const lvba_window_opts *opts,
                             double *window_poses, double *rel_poses, int32_t *anchor_index, double *anchor_poses, int32_t *n_anchors,
                             lvba_scans_t *anchor_scans, lvba_window_info *win_info);
/* ---- the whole LiDAR stage -------------------------------------------------------------------------------------------
 *   lvba_lidar_ba <- LvbaSystem::runLidarBA  src/lvba_system.cpp:312-410 (compute only): window BA (or, with
 *   window_enable = 0, every frame its own anchor, :221-229), then for stage 1 (optional) and stage 2 the voxel map of the
 *   anchor clouds at the current anchor poses with that stage's root voxel size and eigen_ratio_array (:356-377) and
 *   damping_iter over all anchors (:386), finally pose_i = anchor(anchor_index_i) o rel_i for every frame (:393-404; frames of
 *   skipped windows keep their input pose).  poses_in / poses_out [n][12] may alias. */
typedef struct {
    lvba_window_opts window;
    int32_t window_enable, stage1_enable;
    double stage_voxel_size[2];
    float stage_eigen_ratio[2][4];
    lvba_balm_opts lm;           /* damping_iter options of the global stages */
} lvba_lidar_ba_opts;
typedef struct {
    int32_t n_frames, n_windows, n_windows_skipped, n_anchors;
    int32_t stage_ran[2], stage_iters[2], stage_status[2], reserved;
    int64_t stage_voxels[2], stage_factors[2];
    double stage_cost_first[2], stage_cost_last[2];
    double window_ms, stage_ms[2];  /* host wall clock */
} lvba_lidar_ba_report;
void lvba_lidar_ba_default_opts(lvba_lidar_ba_opts *opts);
int32_t lvba_lidar_ba(lvba_scans_t scans, const double *poses_in, const lvba_lidar_ba_opts *opts, double *poses_out,
                      lvba_lidar_ba_report *report);
/* The same with the window stage over several GPUs (lvba_window_ba_multi: scans[k] = share k's frames, whole windows, in
 * order; poses_in / poses_out cover all frames): the global stages are single problems over all anchors and run on scans[0]'s
 * device, where the anchor clouds are gathered.  Needs window_enable = 1. */
int32_t lvba_lidar_ba_multi(int32_t n_shares, const lvba_scans_t *scans, const double *poses_in, const lvba_lidar_ba_opts *opts,
                            double *poses_out, lvba_lidar_ba_report *report);

/* Frame count and per-frame point counts of a scan set; host copy of one frame's xyz [count][3]. */
int32_t lvba_scans_info(lvba_scans_t scans, int32_t *n_frames, int64_t *frame_count);
int32_t lvba_scans_download(lvba_scans_t scans, int32_t frame, float *xyz);

#ifdef __cplusplus
}
#endif
#endif /* LVBA_HIP_H */
