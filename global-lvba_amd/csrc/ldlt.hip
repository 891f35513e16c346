// ldlt.hip -- damped normal-equation solve (H + u*diag(H)) dx = -g on gfx950, fp64.
//
// Replaces the dense->triplet scan + Eigen::SimplicialLDLT of BALM2::damping_iter (reference
// include/BALM/bavoxel.hpp:692-710).  Like SimplicialLDLT it is an UNPIVOTED LDL^T of the lower
// triangle (the exact second-order Hessian may be indefinite; Cholesky would fail where the reference
// succeeds).  Right-looking blocked algorithm, panel width NB = 64, on a column-major lower matrix that
// is either dense (ld = n) or LAPACK lower-band storage (ld = ldab-1) -- the same kernels serve both,
// the band only limits the row window [k+NB, k+NB+bw) each panel touches.
//
//   ldlt_prepare.h    the damped matrix from the block-band Hessian store; the kernels that join the two ends
//   ldlt_diag.h       LDL^T of a 64x64 diagonal block with G = L11^-T D^-1 (every later triangular solve with the block is a product)
//   ldlt_tiles.h      the trailing update A22 -= L21 Z^T as fp64-MFMA tiles (v_mfma_f64_16x16x4_f64)
//   ldlt_lookahead.h  ONE launch per panel: chain role (the next diagonal block), row roles (L21, Z, the next block column),
//                     bulk tiles of earlier panels' rank-128 updates; which launch carries what: ldlt_schedule.h (host)
//   ldlt_back.h       the backward substitution as one chained launch
//   ldlt_nd.h         graphs that are not a narrow band: one level of nested dissection over arcs solved by this very file
//   this file         the launch sequence of one solve (captured once into a hipGraph by block_system.hip)
//
// MFMA operand layout used (v_mfma_f64_16x16x4_f64): lane l supplies A[i=l&15][k=l>>4] and
// B[k=l>>4][j=l&15]; result register r of lane l is D[(l>>4)+4r][l&15].  Products are arranged so that
// l&15 indexes ROWS of the column-major target, i.e. 16 lanes touch 128 contiguous bytes.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <math.h>
#include <utility>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <thread>
#include "lvba_internal.h"
#include "ldlt_schedule.h"
#include "../../include/lvba_hip.h" // status codes

namespace lvba {

typedef double d4 __attribute__((ext_vector_type(4)));
#define LVBA_TS 80 // LDS tile stride (doubles): 80 = 16 mod 32 -> MFMA operand reads are bank-conflict free

#include "ldlt_prepare.h"
#include "ldlt_diag.h"
#include "ldlt_tiles.h"
#include "ldlt_lookahead.h"
#include "ldlt_back.h"

// ------------------------------------------------------------------------------------------- driver
static inline int64_t ldz_for(int64_t n, int64_t bw)
{
    int64_t w = bw + LVBA_NB + 64;
    return w < n ? w : n;
}

static inline int64_t ldlt_ws_one(int64_t n, int64_t bw)
{
    const int64_t nsteps = (n + LVBA_NB - 1) / LVBA_NB;
    return nsteps * 4096 /*G*/ + 3 * n /*d, b, bacc*/ + 4 * ldz_for(n, bw) * LVBA_NB /*Z, four buffers (st % 4)*/ + 64 +
           2 * 4096 /*side copies of A(p+1, p), look-ahead schedule*/ + 2 * 4096 /*panel q's share of the next diagonal block, ditto*/;
}
// two problems' workspaces + matrix 2's solution vector (twisted factorisation)
// + the exchange buffer of the multi-rank form: |S| <= bw + 2 * 64 columns of bw + 1 entries, and the S part of the rhs
static inline int64_t ldlt_exchange_doubles(int64_t n, int64_t bw) { return std::min<int64_t>(n, bw + 2 * LVBA_NB) * (bw + 2) + 64; }
int64_t ldlt_workspace_doubles(int64_t n, int64_t bw) { return 2 * ldlt_ws_one(n, bw) + n + 64 + ldlt_exchange_doubles(n, bw) + 8 /* LVBA_CHECK_BAND's counter */; }

int64_t ldlt_num_panels(int64_t n) { return (n + LVBA_NB - 1) / LVBA_NB; }

// Panels each end eliminates in the twisted form (0: plain top-down).  Both ends run the same number of panels, so the two
// problems have identical geometry and share every launch (blockIdx.y); the middle block S = n - 128 P keeps >= bw columns.
static int64_t twist_panels_of(const LdltMat &A) { return A.no_twist ? 0 : ldlt_twist_panels(A.n, A.ld, A.bw); }
int64_t ldlt_twist_panels(int64_t n, int64_t ld, int64_t bw)
{
    static const bool off = solver_form("notwist");
    if (off || ld == n) return 0; // dense storage: every column is coupled to every other
    const int64_t P = (n - bw) / (2 * LVBA_NB);
    return P >= 4 ? P : 0;
}


// Launch sequence of one solve on one stream (captured once into a hipGraph by the caller): ONE launch per 64-column panel
// (ldlt_lookahead.h), which launch carries which bulk job decided by ldlt_schedule.h (checked on the CPU against a tile-level
// model of the factorisation, tests/ldlt_schedule_check.cpp).
// Band systems are factorised from BOTH ENDS at once (LdltTwist, ldlt_prepare.h; LVBA_SOLVER=notwist turns it off): the serial chain of
// panels -- the latency that bounds this solver -- is P + |S| / 64 long instead of n / 64, with the same flops and no fill.
// With `dist` (>= 2 ranks) the two ends run on two GPUs: rank 0 eliminates T, rank 1 eliminates B (as the second problem,
// alone in its launches), the S block + right-hand side are all-reduced (ranks >= 2 contribute zeros), every rank factorises S,
// rank 0 back-substitutes T and rank 1 B, and the solution is all-reduced.  Per rank the end phase is as long as the
// single-GPU one but moves one window per launch instead of two; what is replicated is the S phase only.
int32_t ldlt_solve(const LdltMat &A, const double *Hblk, int band_blocks, int n_poses, const double *g,
                   const double *u_dev, double *x, double *work, int *status, hipStream_t s, const LdltDist *dist, const int32_t *grp,
                   int phase, const LdltBorder *border)
{
    const int64_t n = A.n, bw = A.bw;
    const int64_t P1 = twist_panels_of(A);
    if (phase != LDLT_ALL && (P1 > 0 || dist)) return LVBA_ERR_STATE; // the halves exist for plain single-rank factorisations
    LdltTwist tw;
    tw.m = P1 * LVBA_NB; tw.n1 = n - tw.m;
    tw.sA = (A.ld + 1) * (n + 1); tw.sW = ldlt_ws_one(n, bw);
    tw.x2 = reinterpret_cast<unsigned long long *>(work + 2 * tw.sW);
    double *Ebuf = work + 2 * tw.sW + n + 64;
    // side: -1 = both ends here (one rank); 0 / 1 = this rank eliminates T / B; 2 = neither (it only takes part in the exchanges)
    const int side = (dist && dist->n_ranks >= 2 && P1 > 0) ? (dist->rank < 2 ? dist->rank : 2) : -1;
    const int64_t nf = tw.n1; // columns the top-down problem factorises (all of them without the twist)
    const int64_t nsteps = (nf + LVBA_NB - 1) / LVBA_NB;
    double *Gall = work;
    double *dvec = Gall + ((n + LVBA_NB - 1) / LVBA_NB) * 4096;
    double *b = dvec + n;
    double *bacc = b + n;
    const int64_t ldz = ldz_for(n, bw);
    double *Zbuf[4] = {bacc + n, bacc + n + ldz * LVBA_NB, bacc + n + 2 * ldz * LVBA_NB, bacc + n + 3 * ldz * LVBA_NB};
    LdltMat M = A; // the problem the launches see: matrix 1 (and matrix 2 through the block index)
    M.n = nf;
    const size_t abytes = (size_t)((A.ld == n) ? n * n : (A.ld + 1) * n) * sizeof(double);
    const bool fill = A.ld != n && n == 6 * (int64_t)n_poses; // band storage: destination-major fill, no memset
    // grid y of the fill: block offsets d0 = 28 y must cover every stored offset d in [0, ldab) of a column, for every element
    // row / column e in [0, 6): d = 6 d0 + t - e (matrix 1) or 6 d0 + t + e - 5 (matrix 2), t in [0, 168) -- i.e. up to ldab + 5
    static_assert(LVBA_PB_ROWS == 6 * LVBA_PB_BLOCKS, "a workgroup of the band fill covers LVBA_PB_BLOCKS block offsets");
    LdltMat M2 = M; // the second problem alone (its pointers are the launch's base pointers): what rank 1 of a multi-rank job runs
    M2.a += tw.sA;
    if (phase != LDLT_BACKWARD) { // ---- fill, factorisation, forward substitution
    if (fill) {
        static const bool check_band = [] { const char *e = getenv("LVBA_CHECK_BAND"); return e && !strcmp(e, "1"); }();
        const int64_t ldab = A.ld + 1, total = 2 * (ldab * (n + 1)) + 65 * ldab; // = block_system.hip's allocation
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(s, &cap);
        if (check_band && cap == hipStreamCaptureStatusNone) {
            unsigned long long *cnt = reinterpret_cast<unsigned long long *>(work + ldlt_workspace_doubles(n, bw) - 8), hc = 0;
            hipMemsetAsync(cnt, 0, sizeof hc, s);
            hipLaunchKernelGGL(ldlt_check_untouched_kernel, dim3(2048), dim3(256), 0, s, (const double *)A.a, ldab, n, tw.n1, tw.sA, total,
                               P1 > 0 ? 1 : 0, cnt);
            hipMemcpyAsync(&hc, cnt, sizeof hc, hipMemcpyDeviceToHost, s);
            hipStreamSynchronize(s);
            if (hc) {
                fprintf(stderr, "lvba: LVBA_CHECK_BAND: %llu non-zero entries in the never-rewritten part of the band store\n", hc);
                return LVBA_ERR_STATE;
            }
        }
        const int64_t cols1 = (tw.n1 + 5) / 6, rows2 = P1 > 0 ? n_poses - tw.m / 6 : 0;
        hipLaunchKernelGGL(ldlt_prepare_band_kernel, dim3((unsigned)std::max(cols1, rows2), (unsigned)((A.ld + 1 + 5 + LVBA_PB_CHUNKS * LVBA_PB_ROWS - 1) / (LVBA_PB_CHUNKS * LVBA_PB_ROWS)), P1 > 0 ? 2 : 1),
                           dim3(256), 0, s, A, Hblk, band_blocks, n_poses, u_dev, tw, grp);
    } else
        hipMemsetAsync(A.a, 0, P1 > 0 ? (size_t)tw.sA * sizeof(double) + abytes : abytes, s);
    hipLaunchKernelGGL(ldlt_prepare_kernel, dim3(fill ? 64 : 2048), dim3(256), 0, s, A, Hblk, band_blocks, n_poses, g, u_dev, b,
                       reinterpret_cast<unsigned long long *>(x), (unsigned long long)LVBA_X_SENTINEL, tw, grp, fill ? 1 : 0, status);
    struct Geo { int64_t k, w0, rend, T; int nbe; };
    auto geom = [&](int64_t st) {
        Geo q;
        q.k = st * LVBA_NB;
        q.nbe = (int)((nf - q.k) < LVBA_NB ? (nf - q.k) : LVBA_NB);
        q.w0 = q.k + q.nbe;
        q.rend = q.k + q.nbe + bw;
        if (q.rend > nf) q.rend = nf;
        q.T = q.w0 < q.rend ? (q.rend - q.w0 + 63) / 64 : 0;
        return q;
    };
    double *side_buf[2] = {Zbuf[3] + ldz * LVBA_NB + 64, Zbuf[3] + ldz * LVBA_NB + 64 + 4096};
    double *dq_buf[2] = {side_buf[1] + 4096, side_buf[1] + 2 * 4096};
    static const int n_cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
        (void)hipGetLastError();
        return n > 0 ? n : 256;
    }();
    // LVBA_SOLVER=bulk64 forces the 64 x 64 bulk tiles (64-bit pointers) that matrices of 4 GB and more take anyway -- the 128 x 64
    // tiles address one problem's storage with 32-bit byte offsets (buffer instructions) -- so that the tests reach that path at test sizes
    static const bool big_env = !solver_form("bulk64");
    const bool big = big_env && ((uint64_t)A.ld * (uint64_t)(n + 128) + (uint64_t)n + 256) * 8 < 0xFFFF0000ull;
    // Panels [sa, sb) of one problem (or of both, ny = 2).  Consecutive panels are PAIRED (e, o = e + 1): e leaves the bulk of its
    // trailing update to its partner's launches, where every C tile is read and written once for both (rank 128).
    // the border's forward substitution (LdltBorder), two panels behind the roles: stage q = [Y role of panel q - 1, U role of panel
    // q - 2] rides in launch X_q (ldlt_lookahead.h: FwdPassenger); the stages no launch is left for run on their own at the end
    int64_t fwd_next = 1;
    auto fwd_stage = [&](int64_t q, int64_t &nwg_out) {
        FwdPassenger f{};
        f.on = 1; f.ldb = border->ldb; f.B = border->B; f.Y = border->Y;
        const int64_t nct = border->ldb / 64;
        nwg_out = 0;
        if (q - 1 < nsteps) {
            const Geo ga = geom(q - 1);
            f.y_on = 1; f.a_k = ga.k; f.a_nbe = ga.nbe; f.Ga = Gall + (q - 1) * 4096;
            if (q - 2 >= 0) { const Geo gm = geom(q - 2); f.has_prev = 1; f.am_k = gm.k; f.am_nbe = gm.nbe; f.am_rend = gm.rend; }
            nwg_out += nct;
        }
        if (q - 2 >= 0) {
            const Geo gc = geom(q - 2);
            if (gc.T >= 2) { f.u_on = 1; f.c_k = gc.k; f.c_nbe = gc.nbe; f.c_w0 = gc.w0; f.c_rend = gc.rend; f.c_T = gc.T; nwg_out += (gc.T - 1) * nct; }
        }
        return f;
    };
    static const bool defer_env = !solver_form("nodefer"); // (A / B and tests: the row roles' full form in every phase)
    auto run_phase = [&](int64_t sa, int64_t sb, unsigned ny, bool second, bool close) {
        const int64_t wo = second ? tw.sW : 0;
        std::vector<SchedLaunch> sched;
        ldlt_schedule_phase(sa, sb, close, true, [&](int64_t st) { return st < nsteps ? geom(st).T : (int64_t)0; }, sched, close && defer_env);
        auto pg = [&](int64_t st) { const Geo g = geom(st); return PanelGeo{g.k, g.w0, g.rend, g.nbe, (int)g.T}; };
        bool dq_prev_written = false; // did the role launch before this one leave panel q's share of the diagonal block?
        for (const SchedLaunch &L : sched) {
            if (L.kind == 0) {
                const Geo g = geom(L.p);
                hipLaunchKernelGGL(ldlt_diag_blocked_kernel, dim3(g.T > 0 ? 2 : 1, ny), dim3(256), 0, s, second ? M2 : M, g.k, g.nbe,
                                   Gall + wo + L.p * 4096, dvec + wo, status, tw.sA, tw.sW, side_buf[L.p % 2] + wo, g.rend);
                continue;
            }
            Step2Args a{};
            a.M = second ? M2 : M; a.sA = tw.sA; a.sW = tw.sW; a.ldz = ldz; a.nprob = (int)ny;
            a.roles = L.roles; a.has_q = L.has_q; a.do_diag = L.do_diag; a.q_extra = L.q_extra; a.defer = L.defer; a.status = status;
            a.dvec = dvec + wo; a.b = b + wo;
            a.stamp_id = L.roles ? (int)L.p : -1; a.stamp_prob = second ? 1 : 0;
            int64_t nwg = 0;
            if (L.roles) {
                a.p = pg(L.p);
                if (L.has_q) a.q = pg(L.p - 1);
                const Geo gn = geom(L.p + 1);
                a.nbe_next = gn.nbe; a.rend_next = gn.rend;
                a.Gp = Gall + wo + L.p * 4096; a.Gn = Gall + wo + (L.p + 1) * 4096;
                a.Zp = Zbuf[L.p % 4] + wo; a.Zq = L.has_q ? Zbuf[(L.p - 1) % 4] + wo : nullptr;
                a.side_r = side_buf[L.p % 2] + wo; a.side_w = side_buf[(L.p + 1) % 2] + wo;
                // Panel p's share of the NEXT launch's diagonal block is formed by row 1 of this launch (which holds L(p+2, p)) and
                // handed to that launch's chain workgroup ready-made (ldlt_lookahead.h: dq_w / dq_r); in the launches where the rows
                // also carry the q_extra product, row 1's q_extra tile goes to a role workgroup of its own (qx_helper)
                a.dq_w = dq_buf[L.p % 2] + wo;
                a.dq_r = (L.has_q && geom(L.p - 1).T >= 2 && dq_prev_written) ? dq_buf[(L.p - 1) % 2] + wo : nullptr;
                dq_prev_written = true;
                a.qx_helper = (L.q_extra && a.p.T >= 2) ? 1 : 0;
                nwg += a.p.T + a.qx_helper;
            }
            for (int j = 0; j < L.njobs; ++j) {
                BulkJob &J = a.job[a.njobs];
                const SchedJob &sj = L.job[j];
                J.o = pg(sj.o); J.Zo = Zbuf[sj.o % 4] + wo; J.pair = sj.pair;
                if (sj.pair) { J.e = pg(sj.o - 1); J.Ze = Zbuf[(sj.o - 1) % 4] + wo; }
                const int64_t Tb = J.o.T - 1;
                if (big) {
                    J.ca = sj.ca; J.cb = sj.cb; J.nwg = 0;
                    for (int64_t c = sj.ca; c < sj.cb; ++c) J.nwg += pair_col_items(c, Tb);
                } else {
                    J.ca = col_start(sj.ca, Tb); J.cb = col_start(sj.cb, Tb); J.nwg = J.cb - J.ca;
                }
                if (J.nwg <= 0) continue;
                nwg += J.nwg;
                ++a.njobs;
            }
            if (border && L.roles && ny == 1 && fwd_next <= L.p) { // (<=: a stage no launch carried rides the next one that can)
                int64_t nf = 0;
                a.fwd = fwd_stage(fwd_next++, nf);
                nwg += nf;
            }
            int64_t grid = nwg * ny;
            if (L.roles && grid > n_cus) {
                // the seats next to the chain workgroups and next to row 1 (one product more than the other rows) stay empty
                // (ldlt_lookahead.h: resv_at)
                a.resv_at = n_cus; a.resv_n = (int)ny * (a.p.T >= 2 ? 2 : 1);
                grid += a.resv_n;
            }
            if (nwg > 0) {
                if (a.defer) {
                    if (big) hipLaunchKernelGGL((ldlt_step2_kernel<true, true>), dim3((unsigned)grid), dim3(256), 0, s, a);
                    else hipLaunchKernelGGL((ldlt_step2_kernel<false, true>), dim3((unsigned)grid), dim3(256), 0, s, a);
                } else {
                    if (big) hipLaunchKernelGGL((ldlt_step2_kernel<true, false>), dim3((unsigned)grid), dim3(256), 0, s, a);
                    else hipLaunchKernelGGL((ldlt_step2_kernel<false, false>), dim3((unsigned)grid), dim3(256), 0, s, a);
                }
            }
        }
    };
    int64_t st0 = 0;
    if (P1 > 0) { // both ends, panels 0 .. P1-1 of the two problems in the same launches (or this rank's end alone)
        if (side != 2) run_phase(0, P1, side < 0 ? 2 : 1, side == 1, true);
        if (side < 0) {
            hipLaunchKernelGGL(ldlt_twist_merge_kernel, dim3(1024), dim3(256), 0, s, A, tw, b); // A: the reversal needs the full n
        } else { // exchange the S block and the S part of the right-hand side
            const int64_t ne = (tw.n1 - tw.m) * (bw + 2);
            hipLaunchKernelGGL(ldlt_twist_pack_kernel, dim3(1024), dim3(256), 0, s, A, tw, (const double *)b, side, Ebuf);
            if (dist->allreduce_sum(dist->ctx, Ebuf, (size_t)ne)) return LVBA_ERR_DIST;
            hipLaunchKernelGGL(ldlt_twist_unpack_kernel, dim3(1024), dim3(256), 0, s, A, tw, b, (const double *)Ebuf);
        }
        st0 = P1;
    }
    run_phase(st0, nsteps, 1, false, false);
    if (border) // the stages no step launch was left to carry
        for (; fwd_next <= nsteps; ++fwd_next) {
            int64_t nf = 0;
            const FwdPassenger f = fwd_stage(fwd_next, nf);
            if (nf > 0) hipLaunchKernelGGL(ldlt_fwd_kernel, dim3((unsigned)nf), dim3(256), 0, s, M, f, (const double *)dvec);
        }
    }
    if (phase == LDLT_FACTOR) return LVBA_OK;
    // backward: the whole substitution as chained launches (ldlt_back.h).  At most 256 panels per launch: one workgroup per CU is
    // then resident whatever else shares the device, so the chain cannot starve even if workgroups were not dispatched in index
    // order; later launches only read finished x (a multi-rank job stops matrix 1's chain after the S panels unless this rank owns T)
    if (side < 0 && P1 > 0) {
        // one rank, both ends: S on matrix 1, then T (matrix 1) and B (matrix 2, from x of S reversed) side by side
        for (int64_t top = nsteps - 1; top >= P1; top -= 256) {
            const int64_t cnt = std::min<int64_t>(256, top - P1 + 1);
            hipLaunchKernelGGL(ldlt_back_chain_kernel, dim3((unsigned)cnt), dim3(256), 0, s, M, (int)top, Gall, dvec, b, x, (int64_t)0, (int64_t)0, (double *)nullptr);
        }
        const int64_t ns = tw.n1 - tw.m;
        hipLaunchKernelGGL(ldlt_twist_xs_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, s, tw, n, (const double *)x);
        for (int64_t top = P1 - 1; top >= 0; top -= 128) { // 2 x 128 workgroups: one per CU, the chains cannot starve
            const int64_t cnt = std::min<int64_t>(128, top + 1);
            hipLaunchKernelGGL(ldlt_back_chain_kernel, dim3((unsigned)cnt, 2), dim3(256), 0, s, M, (int)top, Gall, dvec, b, x, tw.sA, tw.sW,
                               reinterpret_cast<double *>(tw.x2));
        }
        hipLaunchKernelGGL(ldlt_twist_xb_kernel, dim3((unsigned)((tw.m + 255) / 256)), dim3(256), 0, s, tw, n, x);
        return LVBA_OK;
    }
    const int64_t low = side >= 1 ? P1 : 0;
    for (int64_t top = nsteps - 1; top >= low; top -= 256) {
        const int64_t cnt = std::min<int64_t>(256, top - low + 1);
        hipLaunchKernelGGL(ldlt_back_chain_kernel, dim3((unsigned)cnt), dim3(256), 0, s, M, (int)top, Gall, dvec, b, x, (int64_t)0, (int64_t)0, (double *)nullptr);
    }
    if (P1 > 0 && side == 1) { // matrix 2's B part: its chain starts from x of S (reversed)
        const int64_t ns = tw.n1 - tw.m;
        hipLaunchKernelGGL(ldlt_twist_xs_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, s, tw, n, (const double *)x);
        for (int64_t top = P1 - 1; top >= 0; top -= 256) {
            const int64_t cnt = std::min<int64_t>(256, top + 1);
            hipLaunchKernelGGL(ldlt_back_chain_kernel, dim3((unsigned)cnt), dim3(256), 0, s, M2, (int)top, Gall + tw.sW, dvec + tw.sW,
                               b + tw.sW, reinterpret_cast<double *>(tw.x2), (int64_t)0, (int64_t)0, (double *)nullptr);
        }
        hipLaunchKernelGGL(ldlt_twist_xb_kernel, dim3((unsigned)((tw.m + 255) / 256)), dim3(256), 0, s, tw, n, x);
    }
    if (side >= 0) { // everybody gets the whole solution and the worst status
        hipLaunchKernelGGL(ldlt_twist_xmask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, tw, n, side, x);
        if (dist->allreduce_sum(dist->ctx, x, (size_t)n)) return LVBA_ERR_DIST;
        if (dist->allreduce_max_i32(dist->ctx, status)) return LVBA_ERR_DIST;
    }
    return LVBA_OK;
}

#include "ldlt_nd.h"

#ifdef LVBA_STAMPS
} // namespace lvba
// (debug builds only) the stamp arrays of ldlt_lookahead.h, roles then bulk, to host memory; returns the number of words
extern "C" int64_t lvba_debug_stamps(unsigned long long *out, int64_t cap)
{
    const int64_t n1 = (int64_t)LVBA_ST_LAUNCHES * LVBA_ST_ROLES * LVBA_ST_MARKS, n2 = (int64_t)LVBA_ST_LAUNCHES * LVBA_ST_BULK * 2;
    if (!out || cap < n1 + n2) return n1 + n2;
    hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(lvba::g_lvba_stamps), (size_t)n1 * 8) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out + n1, HIP_SYMBOL(lvba::g_lvba_bulk_stamps), (size_t)n2 * 8) != hipSuccess) return -2;
    return n1 + n2;
}
extern "C" int64_t lvba_debug_bulk_marks(unsigned long long *out, int64_t cap)
{
    const int64_t n = (int64_t)LVBA_ST_LAUNCHES * LVBA_ST_BTILES * 12;
    if (!out || cap < n) return n;
    hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(lvba::g_lvba_bulk_marks), (size_t)n * 8) != hipSuccess) return -1;
    return n;
}
namespace lvba { __global__ void debug_tick_kernel(unsigned long long *o) { o[0] = __builtin_amdgcn_s_memrealtime(); } }
// the s_memrealtime counter now (its rate differs from box to box: the reader calibrates it against the host's clock)
extern "C" unsigned long long lvba_debug_tick(void)
{
    unsigned long long *d = nullptr, h = 0;
    if (hipMalloc(&d, 8) != hipSuccess) return 0;
    hipLaunchKernelGGL(lvba::debug_tick_kernel, dim3(1), dim3(1), 0, 0, d);
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    hipFree(d);
    return h;
}
namespace lvba {
#endif

} // namespace lvba
