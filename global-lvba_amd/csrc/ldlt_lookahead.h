// ldlt_lookahead.h -- the look-ahead form of the band LDL^T's panel loop: ONE launch per 64-column panel (included by ldlt.hip,
// after the tile kernels it builds on; no other file includes it).
//
// Until round 3 every panel cost two launches on the critical path: [factorise panel p: every tile-row workgroup repeats the
// 64x64 diagonal factorisation, then forms its tile of L21] -> [apply panel p to block column p+1: "first column"] -> [factorise
// p+1] ...  = 22 us + 7 us per panel, for a chain whose real dependency is only
//      diag(p) -> L(p+1,p) -> A(p+1,p+1) -= L(p+1,p) D L(p+1,p)^T -> diag(p+1).
// Here that chain is the work of ONE workgroup per problem (the "chain" role), which ends launch X_p with the factorisation of
// the NEXT diagonal block; everything else is off the chain and runs beside it in the same launch:
//
//   launch X_p (G_p = L_pp^-T D_p^-1 and d_p come from launch X_{p-1}; q = p - 1):
//     chain  (1 workgroup)        L(p+1,p) = A(p+1,p) G_p, Z = L D;  A(p+1,p+1) -= L(p+1,q) Z(p+1,q)^T + L(p+1,p) Z(p+1,p)^T;
//                                 LDL^T of that block -> G_{p+1}, d_{p+1}   (3 products + the 64-pivot chain)
//     row i  (T_p - 1 workgroups) L(i,p) = A(i,p) G_p, Z(i,p); the tile of block column p+1 in its row:
//                                 A(i,p+1) -= L(i,q) Z(p+1,q)^T + L(i,p) Z(p+1,p)^T, with Z(p+1,p) recomputed from A(p+1,p) and G_p
//                                 (one product more, nothing exchanged inside the launch)          (4 products)
//                                 -- the FULL form.  The two ends of a twisted factorisation run the DEFERRED form
//                                 (ldlt_schedule.h; row_role<true>): panel p's update of block column p+1 waits for the start of
//                                 X_{p+1}, where its operands are stored ones, and only row 1 recomputes Z(p+1,p) (3 products).
//                                 A(p+1,p) is read from a SIDE COPY: the chain workgroup turns that tile into L(p+1,p) in place
//                                 during this very launch.  The copy is written by whoever finished the tile -- row p+1 of
//                                 X_{p-1} (its block-column-p tile), or the phase's first launch -- into one of two buffers.
//     bulk                        the trailing update of panel q on block columns >= p+2 (128 x 64 tiles, bulk_tile_128)
//   Every tile has exactly one writer per launch: block column p is turned into L by the roles, block column p+1 receives the
//   last two panels' contributions from the roles (panel q's could not come earlier: L(.,q) is made in X_q), block columns
//   >= p+2 belong to the bulk.  No workgroup waits for another inside a launch; the only ordering is between launches.
//
// The forward substitution rides along as before: every role workgroup forms y_p = D G^T b_p itself and updates its rows of b.
#pragma once

struct PanelGeo { int64_t k, w0, rend; int nbe, T; };

// -DLVBA_STAMPS (tools/build_variant.py stamps -DLVBA_STAMPS; never in the shipped library): wall-clock marks (s_memrealtime) of
// the role workgroups and the first / last instruction of every bulk workgroup, per launch, read back through
// lvba_debug_stamps (ldlt.hip) by tools/step_stamps.py.
#ifdef LVBA_STAMPS
#define LVBA_ST_LAUNCHES 256
#define LVBA_ST_ROLES 96   // problem * 48 + role index
#define LVBA_ST_MARKS 12
#define LVBA_ST_BULK 640
__device__ unsigned long long g_lvba_stamps[LVBA_ST_LAUNCHES][LVBA_ST_ROLES][LVBA_ST_MARKS];
__device__ unsigned long long g_lvba_bulk_stamps[LVBA_ST_LAUNCHES][LVBA_ST_BULK][2];
#define LVBA_ST_BTILES 640
__device__ unsigned long long g_lvba_bulk_marks[LVBA_ST_LAUNCHES][LVBA_ST_BTILES][12]; // phases inside the first bulk tiles (ldlt_tiles.h)
#define LVBA_STAMP(A_, prob_, role_, m_)                                                                                          \
    do {                                                                                                                          \
        if (threadIdx.x == 0 && (A_).stamp_id >= 0 && (A_).stamp_id < LVBA_ST_LAUNCHES && (role_) < 48)                            \
            g_lvba_stamps[(A_).stamp_id][(prob_) * 48 + (role_)][m_] = __builtin_amdgcn_s_memrealtime();                          \
    } while (0)
#else
#define LVBA_STAMP(A_, prob_, role_, m_) do { } while (0)
#endif

// one trailing-update job of a launch: panel o alone (rank 64) or together with its partner e = o - 1 (rank 128), the 128 x 64
// tiles of the tile columns [ca, cb) in bulk coordinates (tile column tj' <-> block column o + 2 + tj'); 64 x 64 form: the tiles
// [ca, cb) of the column-major enumeration
struct BulkJob {
    PanelGeo o, e;
    const double *Zo, *Ze;
    int pair;
    int64_t ca, cb, nwg;
};
// Passengers of the launch: the forward substitution of a block of extra right-hand sides (the border columns of a dissected
// system, ldlt_nd.h), two panels behind the roles.  B, Y: [n][ldb] row-major.  Stage q (it rides in launch X_q) is
//   Y role  (panel a = q - 1; ldb / 64 workgroups)  B_a -= L(a, a - 1) Y_{a-1}  -- the one update its rows still miss --, then
//                                                   Y_a = D_a G_a^T B_a
//   U role  (panel c = q - 2; (T_c - 1) ldb / 64)   B[tile t >= 2 of c's window] -= L(tile, c) Y_c
// Y_{a-1} and Y_c come from the stages before (launches before): nothing is exchanged inside a launch, and no product is done
// twice (a first form let every tile recompute its panel's Y: 16 TFLOP/s on twice the flops).
struct FwdPassenger {
    int on;
    int y_on, a_nbe, am_nbe, has_prev; // Y role: panel a (and a - 1, if there is one)
    int u_on, c_nbe;                   // U role: panel c
    int64_t ldb;
    int64_t a_k, am_k, am_rend;
    int64_t c_k, c_w0, c_rend, c_T;
    const double *Ga;
    double *B, *Y;
};
struct Step2Args {
    LdltMat M;
    int64_t sA, sW, ldz;
    int nprob, roles, has_q, do_diag, nbe_next, njobs;
    int q_extra;         // row roles: also A(i, p+2) -= L(i, q) Z(p+2, q)^T (ldlt_schedule.h)
    int defer;           // the deferred form of the row roles (ldlt_schedule.h; row_role)
    int64_t rend_next;   // row limit of panel p + 1's window
    PanelGeo p, q;
    const double *side_r; // A(p+1, p) as [m][row] (64 x 64, masked like load_panel_tile), read by the row roles
    double *side_w;       // A(p+2, p+1) for the next launch, written by row 1
    // Panel q's contribution to the diagonal block p + 1, L(p+1, q) Z(p+1, q)^T, formed OFF the chain: row 1 of launch X_q holds
    // L(p+1, q) in its registers anyway and leaves the 64 x 64 product in dq_w ([column][row]); the chain workgroup of X_p starts
    // its accumulators from dq_r instead of fetching L and Z and multiplying itself (22.7 instead of 26.1 us alone).  nullptr:
    // the chain multiplies itself / row 1 leaves nothing.
    const double *dq_r;
    double *dq_w;
    int qx_helper; // q_extra launches with dq_w: row 1's q_extra tile -- the diagonal tile (p+2, p+2) -- is done by one more role
                   // workgroup (role index T), so that row 1 (which forms the dq product) does as many products as the other rows
    const double *Gp; // G of panel p
    double *Gn;       // G of panel p + 1 (written by the chain role)
    double *dvec, *b, *Zp;
    const double *Zq;
    int *status;
    BulkJob job[2];
    // The chain workgroups ALONE on their CUs.  The step kernel's 80 KB of LDS admit two workgroups per CU, and the dispatcher
    // fills the CUs in block order, round after round (observed, round 4: block b and block b + #CUs share a CU):
    // the blocks [resv_at, resv_at + resv_n) -- the would-be partners of the chain workgroups, blocks 0 .. resv_n - 1 -- return
    // at once and the later blocks count from resv_n less.  Measured on the two-problem launch of config C3: the chain workgroup
    // 79 900 -> 59 600 cycles (what it takes alone), the launch 37.9 -> 29.6 us.  Placement is a matter of speed only: wherever
    // the blocks land, every tile is still done exactly once.  resv_n = 0: off.
    // (Round 5, measured and withdrawn: the second half of the ROW workgroups on the seats next to the first half -- rows next to
    // rows, bulk tiles next to bulk tiles, where today every row shares its CU with a bulk tile: C3 solve 4.11 -> 4.17 ms.)
    int resv_at, resv_n;
    int stamp_id, stamp_prob; // (LVBA_STAMPS builds: slot of this launch in the stamp arrays; problem of a one-problem launch)
    FwdPassenger fwd;
};

// Scratch doubles in the PAD of the first [m][row] tile (LVBA_TS = 80 doubles per column of 64 rows: 16 spare behind each of the
// 64 columns, which stage_tile / put_acc never touch): b_k, y_k, d_p and the partial sums of the role workgroups.
__device__ __forceinline__ double &pad_at(double *lds, int idx) { return lds[(idx >> 4) * LVBA_TS + 64 + (idx & 15)]; }
#define LVBA_FETCH(...) asm volatile("" ::__VA_ARGS__) // the values in scalar registers HERE: their kernel-argument loads leave in one batch (ldlt_step2_kernel)
#define LVBA_PAD_BK 0
#define LVBA_PAD_YS 64
#define LVBA_PAD_DP 128
#define LVBA_PAD_RED 192 // [4][64]

// acc[t][reg] += sum_m Zs[m][16 w + kk + 4 reg] * Ls[m][16 t + i]   (one 64 x 64 x 64 product out of LDS, 16 MFMAs per wave and
// tile row block; the same operand pattern serves  C -= L Z^T  (Ls = L, Zs = Z)  and  L = A G  (Ls = A, Zs = G[m][j]))
// The operands of step k0 + 4 are read before the MFMAs of step k0 are issued (a role workgroup is ALONE on its SIMDs: a
// wavefront issues in order, and an LDS read placed behind the MFMAs that need the previous one only starts when the matrix
// pipe is draining -- measured with the loop in its plain form: 127 cycles per MFMA instead of 64).
__device__ __forceinline__ void tile_product(const double *Ls, const double *Zs, int w, int i, int kk, d4 (&acc)[4])
{
    const double *zp = Zs + kk * LVBA_TS + 16 * w + i, *lp = Ls + kk * LVBA_TS + i;
    double a[2], bv[2][4];
    a[0] = zp[0];
#pragma unroll
    for (int t = 0; t < 4; ++t) bv[0][t] = lp[16 * t];
#pragma unroll
    for (int st = 0; st < 16; ++st) {
        const int q = st & 1;
        if (st + 1 < 16) {
            a[q ^ 1] = zp[4 * (st + 1) * LVBA_TS];
#pragma unroll
            for (int t = 0; t < 4; ++t) bv[q ^ 1][t] = lp[4 * (st + 1) * LVBA_TS + 16 * t];
        }
        __builtin_amdgcn_sched_barrier(0); // (LLVM's scheduler otherwise sinks the reads to just before their first use)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], bv[q][t], acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}
// registers (thread (row, m = w + 4 it)) -> T[m][row]
__device__ __forceinline__ void stage_tile(double *T, const double (&v)[16], int w, int row)
{
#pragma unroll
    for (int it = 0; it < 16; ++it) T[(w + 4 * it) * LVBA_TS + row] = v[it];
}
// a product's result (acc[t][reg] <-> row 16 t + i of the Ls operand, row 16 w + kk + 4 reg of the Zs operand) as an operand
// tile T[col = Zs row][row = Ls row], every column scaled by sc[col] (nullptr: unscaled)
__device__ __forceinline__ void put_acc(double *T, const d4 (&acc)[4], int w, int i, int kk, double *lds_scale)
{
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int c = 16 * w + kk + 4 * reg;
        const double sc = lds_scale ? pad_at(lds_scale, LVBA_PAD_DP + c) : 1.0;
#pragma unroll
        for (int t = 0; t < 4; ++t) T[c * LVBA_TS + 16 * t + i] = acc[t][reg] * sc;
    }
}
// tile rows [r0, r0 + 64) x the 64 columns of a panel (column kc + m), rows >= rlim and columns >= nbe read as zero.
// The loads are UNCONDITIONAL -- what lies past the window or the panel's last column is inside the allocation (block_system.hip
// leaves the slack; ldlt.hip's workspace likewise) -- and edge tiles are masked afterwards: written as `cond ? load : 0` every one of
// a role's 64 first loads sat behind its own exec-mask branch and 64-bit multiply, and a row workgroup spent 4.8 us REQUESTING them
// (stamps, round 6: start 0.6 us, last request 5.4 us, data there 5.6 us).
__device__ __forceinline__ void mask_tile(double (&v)[16], bool whole, bool rok, int nbe, int w)
{
    if (whole) return; // block-uniform
#pragma unroll
    for (int it = 0; it < 16; ++it) v[it] = (rok && w + 4 * it < nbe) ? v[it] : 0.0;
}
__device__ __forceinline__ void load_panel_tile(const LdltMat &M, int64_t r0, int64_t kc, int64_t rlim, int nbe, int w, int row,
                                                double (&v)[16])
{
    const double *__restrict__ base = M.a + (r0 + row) + (kc + w) * M.ld;
    const int64_t step = 4 * M.ld;
#pragma unroll
    for (int it = 0; it < 16; ++it) v[it] = base[it * step];
    mask_tile(v, r0 + 64 <= rlim && nbe == 64, r0 + row < rlim, nbe, w);
}
// the same rows of a panel's Z buffer (row r at r - w0)
__device__ __forceinline__ void load_z_tile(const double *__restrict__ Z, int64_t ldz, int64_t r0, int64_t w0, int64_t rlim, int nbe,
                                            int w, int row, double (&v)[16])
{
    const double *__restrict__ base = Z + (r0 + row - w0) + w * ldz;
    const int64_t step = 4 * ldz;
#pragma unroll
    for (int it = 0; it < 16; ++it) v[it] = base[it * step];
    mask_tile(v, r0 + 64 <= rlim && nbe == 64, r0 + row < rlim, nbe, w);
}
// a product's result layout (acc[t][reg] <-> row R0 + 16 t + i, column C0 + 16 w + kk + 4 reg) <-> global memory: the C tile's
// entries, loaded unconditionally (the caller masks what it must), and the stores of an inner tile without a branch per entry
__device__ __forceinline__ void load_c_tile(const LdltMat &M, int64_t R0, int64_t C0, int w, int i, int kk, double (&cv)[16])
{
    const double *__restrict__ base = M.a + (R0 + i) + (C0 + 16 * w + kk) * M.ld;
    const int64_t step = 4 * M.ld;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) cv[4 * t + reg] = base[16 * t + reg * step];
}
// y_k and the rows' share of the forward substitution, common to both roles.  Before: Zs = G[m][j], pad BK = b_k, DP = d_p.
__device__ __forceinline__ void fwd_partial_y(double *lds, const double *Zs, int w, int row)
{
    double z = 0.0; // z_j = sum_m G[m][j] b_m : thread (j = row, w) sums m in [16 w, 16 w + 16)
#pragma unroll
    for (int m = 0; m < 16; ++m) z += Zs[(16 * w + m) * LVBA_TS + row] * pad_at(lds, LVBA_PAD_BK + 16 * w + m);
    pad_at(lds, LVBA_PAD_RED + 64 * w + row) = z;
}
__device__ __forceinline__ double red4(double *lds, int j)
{
    return pad_at(lds, LVBA_PAD_RED + j) + pad_at(lds, LVBA_PAD_RED + 64 + j) + pad_at(lds, LVBA_PAD_RED + 128 + j) +
           pad_at(lds, LVBA_PAD_RED + 192 + j);
}

// L(tile, p) and Z(tile, p) = L D to global memory straight from the product's registers (16 lanes = 128 contiguous bytes of a
// column); a tile wholly inside the window without a branch per entry
__device__ __forceinline__ void store_lz_tile(const LdltMat &M, const PanelGeo &p, int64_t R0, const d4 (&accL)[4], double *lds,
                                              double *__restrict__ Zp, int64_t ldz, int w, int i, int kk)
{
    double *__restrict__ lb = M.a + (R0 + i) + (p.k + 16 * w + kk) * M.ld;
    double *__restrict__ zb = Zp + (R0 + i - p.w0) + (16 * w + kk) * ldz;
    const int64_t ls = 4 * M.ld, zs = 4 * ldz;
    if (R0 + 64 <= p.rend && p.nbe == 64) { // block-uniform
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const double dc = pad_at(lds, LVBA_PAD_DP + 16 * w + kk + 4 * reg);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                lb[16 * t + reg * ls] = accL[t][reg];
                zb[16 * t + reg * zs] = accL[t][reg] * dc;
            }
        }
        return;
    }
    const int rl_lim = (int)(p.rend - R0 < 64 ? p.rend - R0 : 64);
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int c = 16 * w + kk + 4 * reg;
        const double dc = pad_at(lds, LVBA_PAD_DP + c);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (16 * t + i < rl_lim && c < p.nbe) {
                lb[16 * t + reg * ls] = accL[t][reg];
                zb[16 * t + reg * zs] = accL[t][reg] * dc;
            }
        }
    }
}
// C tile (rows R0.., columns C0..) = cv - acc, entries with row < rlim, column < clim (and row >= column if `lower`); an inner
// tile without a branch per entry
__device__ __forceinline__ void store_c_tile(const LdltMat &M, int64_t R0, int64_t C0, int64_t rlim, int64_t clim, bool lower,
                                             const double (&cv)[16], const d4 (&acc)[4], int w, int i, int kk)
{
    double *__restrict__ base = M.a + (R0 + i) + (C0 + 16 * w + kk) * M.ld;
    const int64_t step = 4 * M.ld;
    if (R0 + 64 <= rlim && C0 + 64 <= clim && (!lower || R0 >= C0 + 63)) { // block-uniform
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) base[16 * t + reg * step] = cv[4 * t + reg] - acc[t][reg];
        return;
    }
    const int rl_lim = (int)(rlim - R0 < 64 ? rlim - R0 : 64), cl_lim = (int)(clim - C0 < 64 ? clim - C0 : 64); // tile-local, 32 bits
    const int dd = lower ? (int)(C0 - R0 < -64 ? -64 : C0 - R0) : -64; // lower triangle: local row >= local column + dd
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int rl = 16 * t + i, cl = 16 * w + kk + 4 * reg;
            if (rl < rl_lim && cl < cl_lim && rl >= cl + dd) base[16 * t + reg * step] = cv[4 * t + reg] - acc[t][reg];
        }
}

// ---------------------------------------------------------------------------------------------- the chain role
__device__ __forceinline__ void chain_role(double *lds, const LdltMat &M, const Step2Args &A, const double *__restrict__ Gp,
                                           double *__restrict__ Gn, double *__restrict__ dvec, double *__restrict__ b,
                                           double *__restrict__ Zp, const double *__restrict__ Zq, const double *__restrict__ dq_r, int st_prob)
{
    LVBA_FETCH("s"(A.p.k), "s"(A.p.w0), "s"(A.p.rend), "s"(A.p.nbe), "s"(A.q.k), "s"(A.q.w0), "s"(A.q.rend), "s"(A.q.nbe), "s"(A.has_q), "s"(A.do_diag),
               "s"(A.nbe_next), "s"(A.ldz), "s"(Gp), "s"(Gn), "s"(dvec), "s"(b), "s"(Zp), "s"(Zq), "s"(dq_r));
    LVBA_STAMP(A, st_prob, 0, 0);
    double *Ls = lds, *Zs = lds + 64 * LVBA_TS;
    const PanelGeo &p = A.p, &q = A.q;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, row = tid & 63, i = lane & 15, kk = lane >> 4;
    const int64_t r0 = p.w0, r = r0 + row; // rows of tile p + 1 = columns of panel p + 1
    __builtin_amdgcn_s_setprio(3);        // the launch is as long as this workgroup: first call on the issue slots it shares
    const bool has_q = A.has_q && r0 < q.rend; // (a band narrower than two tiles: panel q does not reach tile row p + 1)
    const bool use_dq = has_q && dq_r;         // panel q's contribution comes ready-made from row 1 of the launch before
    const bool use_q = has_q && !use_dq;
    double va[16], vb[16], a1[16], gp[16];
    if (use_q) {
        load_panel_tile(M, r0, q.k, q.rend, q.nbe, w, row, va);    // L(p+1, q)
        load_z_tile(Zq, A.ldz, r0, q.w0, q.rend, q.nbe, w, row, vb); // Z(p+1, q)
    }
    load_panel_tile(M, r0, p.k, p.rend, p.nbe, w, row, a1);        // A(p+1, p)
#pragma unroll
    for (int it = 0; it < 16; ++it) gp[it] = Gp[tid + 256 * it];   // G[m][c], row-major: thread (c = row, m = w + 4 it)
    const double bk = (tid < p.nbe) ? b[p.k + tid] : 0.0;
    const double dk = (tid < p.nbe) ? dvec[p.k + tid] : 0.0;
    d4 acc[4], accL[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = accL[t] = (d4){0.0, 0.0, 0.0, 0.0};
    if (use_dq) { // requested behind the operands of the first product: needed only by the last one
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) acc[t][reg] = dq_r[(16 * w + kk + 4 * reg) * 64 + 16 * t + i];
    }
    if (use_q) { // block-uniform
        stage_tile(Ls, va, w, row);
        stage_tile(Zs, vb, w, row);
        __syncthreads();
        tile_product(Ls, Zs, w, i, kk, acc);
        __syncthreads();
    }
    const int nbn = A.nbe_next;
    LVBA_STAMP(A, st_prob, 0, 8); // (loads requested)
    stage_tile(Ls, a1, w, row);
    stage_tile(Zs, gp, w, row);
    if (tid < 64) {
        pad_at(lds, LVBA_PAD_BK + tid) = bk;
        pad_at(lds, LVBA_PAD_DP + tid) = dk;
    }
    __syncthreads();
    LVBA_STAMP(A, st_prob, 0, 1);
    fwd_partial_y(lds, Zs, w, row);
    tile_product(Ls, Zs, w, i, kk, accL); // accL[t][reg] = L[row 16 t + i][column 16 w + kk + 4 reg]
    __syncthreads();
    LVBA_STAMP(A, st_prob, 0, 2);
    // the block itself, as the products' result layout has it: cv[4 t + reg] <-> (r0 + 16 t + i, r0 + 16 w + kk + 4 reg).  ALL of
    // it (a band narrower than a tile leaves rows of the block outside panel p's window; they still belong to the block).
    // Requested here: it is needed after the last product, whose 2 us cover the way from L2 / HBM.
    double cv[16];
    load_c_tile(M, r0, r0, w, i, kk, cv);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int cl = 16 * w + kk + 4 * reg, rl = 16 * t + i;
            cv[4 * t + reg] = (rl < nbn && cl <= rl) ? cv[4 * t + reg] : 0.0;
        }
    put_acc(Ls, accL, w, i, kk, nullptr); // L(p+1,p) as [m][row]
    put_acc(Zs, accL, w, i, kk, lds);     // Z = L D
    if (tid < 64) pad_at(lds, LVBA_PAD_YS + tid) = (tid < p.nbe) ? red4(lds, tid) * dk : 0.0;
    store_lz_tile(M, p, r0, accL, lds, Zp, A.ldz, w, i, kk); // L and Z leave for global memory
    __syncthreads();
    LVBA_STAMP(A, st_prob, 0, 3);
    { // b[r] -= L[row] . y_p, four lanes per row
        double sacc = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) sacc += Ls[(16 * w + j) * LVBA_TS + row] * pad_at(lds, LVBA_PAD_YS + 16 * w + j);
        pad_at(lds, LVBA_PAD_RED + 64 * w + row) = sacc;
    }
    tile_product(Ls, Zs, w, i, kk, acc);
    __syncthreads();
    LVBA_STAMP(A, st_prob, 0, 4);
    if (tid < 64 && r < p.rend) b[r] -= red4(lds, tid);
    if (!A.do_diag) { // the phase ends here: the updated block goes back to the matrix
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int cl = 16 * w + kk + 4 * reg, rl = 16 * t + i;
                if (rl < nbn && cl <= rl) M.a[(r0 + rl) + (r0 + cl) * M.ld] = cv[4 * t + reg] - acc[t][reg];
            }
        return;
    }
    __syncthreads(); // (the red4 reads above: W overwrites the pads)
    // W of the blocked factorisation, straight from the registers: lower triangle of the block, identity below its last row
    // (a short last panel), the identity appended as rows 64..127
    double *W = lds;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int c = 16 * w + kk + 4 * reg, rr = 16 * t + i;
            double v;
            if (rr < nbn) v = (c <= rr) ? cv[4 * t + reg] - acc[t][reg] : 0.0;
            else v = (c == rr) ? 1.0 : 0.0;
            W[c * LVBA_W1S + rr] = v;
            W[c * LVBA_W1S + 64 + rr] = (c == rr) ? 1.0 : 0.0;
        }
    // (no barrier here: the first diag step reads columns 0 .. 15 only, and those are wavefront 0's own -- a product's result
    // layout gives wavefront w the columns 16 w .. 16 w + 15 --, so it starts while the others are still writing theirs; the
    // barrier behind that step inside diag_blocked_factor is the first one anybody needs)
    LVBA_STAMP(A, st_prob, 0, 5);
    diag_blocked_factor(lds, nbn, A.status);
    LVBA_STAMP(A, st_prob, 0, 6);
    const double *dvs = lds + LVBA_K1B_DVS;
    if (tid < nbn) dvec[p.w0 + tid] = dvs[tid];
    for (int e = tid; e < 4096; e += 256) Gn[e] = W[(e & 63) * LVBA_W1S + 64 + (e >> 6)];
    LVBA_STAMP(A, st_prob, 0, 7);
}

// ---------------------------------------------------------------------------------------------- the row role
// tile row t >= 1 of panel p's window (global tile row i = p + 1 + t).
// A.defer (ldlt_schedule.h, the two ends of a twisted factorisation): panel q's update of the row's OWN tile, A(i,p) -= L(i,q)
// Z(p,q)^T, is still to come when the launch starts (`pre`); rows t >= 2 of a launch that is not the phase's last are `lean`: they
// stop after L(i,p) and panel q's update of block column p + 1 -- panel p's follows at the start of the next launch -- and never
// form L(p+1,p).
template <bool dfr> // = A.defer, known when the kernel is compiled: either form alone fits the registers, both side by side spill
__device__ __forceinline__ void row_role(double *lds, const LdltMat &M, const Step2Args &A, int64_t t_row, const double *__restrict__ Gp,
                                         const double *__restrict__ dvec, double *__restrict__ b, double *__restrict__ Zp,
                                         const double *__restrict__ Zq, const double *__restrict__ side_r, double *__restrict__ side_w,
                                         double *__restrict__ dq_w, int st_prob)
{
    LVBA_FETCH("s"(A.p.k), "s"(A.p.w0), "s"(A.p.rend), "s"(A.p.nbe), "s"(A.q.k), "s"(A.q.w0), "s"(A.q.rend), "s"(A.q.nbe), "s"(A.has_q), "s"(A.q_extra),
               "s"(A.qx_helper), "s"(A.nbe_next), "s"(A.rend_next), "s"(A.ldz), "s"(A.do_diag), "s"(Gp), "s"(dvec), "s"(b), "s"(Zp), "s"(Zq),
               "s"(side_r), "s"(side_w), "s"(dq_w));
    LVBA_STAMP(A, st_prob, (int)t_row, 0);
    double *Ls = lds, *Zs = lds + 64 * LVBA_TS;
    const PanelGeo &p = A.p, &q = A.q;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, row = tid & 63, i = lane & 15, kk = lane >> 4;
    const int64_t r0 = p.w0 + 64 * t_row, r = r0 + row; // this tile row
    const int64_t s0 = p.w0;                             // rows of tile p + 1 = columns of the updated tile
    const bool use_q = A.has_q && r0 < q.rend;           // the row lies inside panel q's window (block-uniform)
    const bool pre = dfr && use_q;                       // panel q's update of A(i, p) first
    const bool lean = dfr && A.do_diag && t_row >= 2;
    double va[16], vb[16], ai[16], gp[16];
    double bk = 0.0, dk = 0.0;
    // panel p's operands: requested at once if there is no panel q to apply first, else behind the staging of panel q's tiles
    // (they arrive under a product; requested together with panel q's, 128 registers of operands were in flight at once and the
    // C tile of the q_extra product was spilled as it arrived, one memory round trip after the other)
    auto load_g = [&]() {
#pragma unroll
        for (int it = 0; it < 16; ++it) gp[it] = Gp[tid + 256 * it];
        if (tid < p.nbe) { bk = b[p.k + tid]; dk = dvec[p.k + tid]; }
    };
    if (use_q) {
        load_panel_tile(M, r0, q.k, q.rend, q.nbe, w, row, va);      // L(i, q)
        load_z_tile(Zq, A.ldz, s0, q.w0, q.rend, q.nbe, w, row, vb); // Z(p+1, q)
    } else { // no panel q to apply first: A(i, p) as the operand it is
        load_panel_tile(M, r0, p.k, p.rend, p.nbe, w, row, ai);
        load_g();
    }
    d4 acc[4], accI[4], acc0[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = accI[t] = acc0[t] = (d4){0.0, 0.0, 0.0, 0.0};
    LVBA_STAMP(A, st_prob, (int)t_row, 8); // (loads requested)
    // the tile of block column p + 1 in this row: cv[4 t + reg] <-> (r0 + 16 t + i, s0 + 16 w + kk + 4 reg).  Row 1 reads it
    // as far as the NEXT panel's window reaches (rows the band gains there still hold their original entries): its result is
    // also the side copy of A(p+2, p+1).
    const bool mk_side = t_row == 1; // block-uniform
    const int64_t rlim = mk_side ? A.rend_next : p.rend;
    double cv[16];
    auto load_cv = [&]() {
        load_c_tile(M, r0, s0, w, i, kk, cv);
        if (!(r0 + 64 <= rlim && s0 + 64 <= rlim)) { // block-uniform: edge tiles
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    cv[4 * t + reg] = (r0 + 16 * t + i < rlim && s0 + 16 * w + kk + 4 * reg < rlim) ? cv[4 * t + reg] : 0.0;
        }
    };
    if (use_q) {
        double ca[16]; // A(i, p) in the products' layout: (r0 + 16 t + i, p.k + 16 w + kk + 4 reg)
        auto load_p = [&]() { // panel p's operands: requested so that they arrive under the last product before they are needed
            if constexpr (dfr) load_c_tile(M, r0, p.k, w, i, kk, ca);
            else load_panel_tile(M, r0, p.k, p.rend, p.nbe, w, row, ai); // (the full form: A(i, p) is complete, an operand as it is)
            load_g();
        };
        stage_tile(Ls, va, w, row);
        stage_tile(Zs, vb, w, row);
        LVBA_STAMP(A, st_prob, (int)t_row, 9); // (first tiles there and staged)
        const bool qx = A.q_extra && !(A.qx_helper && t_row == 1); // (row 1's tile of block column p + 2: qx_diag_role, if there is one)
        if (qx) load_z_tile(Zq, A.ldz, s0 + 64, q.w0, q.rend, q.nbe, w, row, vb); // Z(p+2, q)
        if (pre) load_z_tile(Zq, A.ldz, q.w0, q.w0, q.rend, q.nbe, w, row, va);  // Z(p, q): rows of tile p = columns of panel p
        __syncthreads();
        if (!qx && !pre) load_p();
        if (lean) load_cv();
        tile_product(Ls, Zs, w, i, kk, acc);
        __syncthreads();
        if (lean) { // block column p + 1 is done for this launch: panel p's share follows at the start of the next one
            store_c_tile(M, r0, s0, p.rend, p.rend, false, cv, acc, w, i, kk);
        }
        if (qx) { // block column p + 2 from panel q alone: this row's tile, with L(i, q) still in LDS
            stage_tile(Zs, vb, w, row);
            double c2[16];
            load_c_tile(M, r0, s0 + 64, w, i, kk, c2); // (unmasked: what is not stored below is not used)
            __syncthreads();
            d4 accx[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) accx[t] = (d4){0.0, 0.0, 0.0, 0.0};
            if (!pre) load_p();
            tile_product(Ls, Zs, w, i, kk, accx);
            store_c_tile(M, r0, s0 + 64, q.rend, q.rend, true, c2, accx, w, i, kk);
            __syncthreads();
        }
        if constexpr (dfr) { // A(i, p) -= L(i, q) Z(p, q)^T in the products' layout, then as the next product's operand
            d4 accA[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) accA[t] = (d4){0.0, 0.0, 0.0, 0.0};
            stage_tile(Zs, va, w, row);
            load_p();
            __syncthreads();
            tile_product(Ls, Zs, w, i, kk, accA);
            const bool whole = r0 + 64 <= p.rend && p.nbe == 64; // block-uniform
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const bool in = whole || (r0 + 16 * t + i < p.rend && 16 * w + kk + 4 * reg < p.nbe);
                    accA[t][reg] = in ? ca[4 * t + reg] - accA[t][reg] : 0.0;
                }
            __syncthreads(); // everybody is done with L(i, q) in Ls
            put_acc(Ls, accA, w, i, kk, nullptr); // A(i, p) as [m][row]
        } else
            stage_tile(Ls, ai, w, row);
    } else
        stage_tile(Ls, ai, w, row);
    LVBA_STAMP(A, st_prob, (int)t_row, 1);
    if (!lean) {
#pragma unroll
        for (int it = 0; it < 16; ++it) va[it] = side_r[(w + 4 * it) * 64 + row]; // A(p+1, p), for Z(p+1, p): the side copy
    }
    stage_tile(Zs, gp, w, row);
    if (tid < 64) {
        pad_at(lds, LVBA_PAD_BK + tid) = bk;
        pad_at(lds, LVBA_PAD_DP + tid) = dk;
    }
    __syncthreads();
    LVBA_STAMP(A, st_prob, (int)t_row, 2);
    fwd_partial_y(lds, Zs, w, row);
    tile_product(Ls, Zs, w, i, kk, accI); // L(i, p)
    __syncthreads();
    LVBA_STAMP(A, st_prob, (int)t_row, 3);
    if (lean) {
        if (tid < 64) pad_at(lds, LVBA_PAD_YS + tid) = (tid < p.nbe) ? red4(lds, tid) * dk : 0.0;
        put_acc(Ls, accI, w, i, kk, nullptr); // L(i, p) as [m][row]
        store_lz_tile(M, p, r0, accI, lds, Zp, A.ldz, w, i, kk); // L(i, p) and Z(i, p) to global memory
        __syncthreads();
        LVBA_STAMP(A, st_prob, (int)t_row, 5);
        double sacc = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) sacc += Ls[(16 * w + j) * LVBA_TS + row] * pad_at(lds, LVBA_PAD_YS + 16 * w + j);
        pad_at(lds, LVBA_PAD_RED + 64 * w + row) = sacc; // (red4 above read these slots before the barrier)
        __syncthreads();
        if (tid < 64 && r < p.rend) b[r] -= red4(lds, tid);
        LVBA_STAMP(A, st_prob, (int)t_row, 7);
        return;
    }
    stage_tile(Ls, va, w, row);           // G stays in Zs
    if (tid < 64) pad_at(lds, LVBA_PAD_YS + tid) = (tid < p.nbe) ? red4(lds, tid) * dk : 0.0;
    __syncthreads();
    load_cv();
    tile_product(Ls, Zs, w, i, kk, acc0); // L(p+1, p)
    __syncthreads();
    LVBA_STAMP(A, st_prob, (int)t_row, 4);
    put_acc(Ls, accI, w, i, kk, nullptr); // L(i, p) as [m][row]
    put_acc(Zs, acc0, w, i, kk, lds);     // Z(p+1, p) = L(p+1, p) D
    store_lz_tile(M, p, r0, accI, lds, Zp, A.ldz, w, i, kk); // L(i, p) and Z(i, p) to global memory
    __syncthreads();
    LVBA_STAMP(A, st_prob, (int)t_row, 5);
    {
        double sacc = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) sacc += Ls[(16 * w + j) * LVBA_TS + row] * pad_at(lds, LVBA_PAD_YS + 16 * w + j);
        pad_at(lds, LVBA_PAD_RED + 64 * w + row) = sacc;
    }
    tile_product(Ls, Zs, w, i, kk, acc);
    __syncthreads();
    LVBA_STAMP(A, st_prob, (int)t_row, 6);
    if (tid < 64 && r < p.rend) b[r] -= red4(lds, tid);
    store_c_tile(M, r0, s0, p.rend, p.rend, false, cv, acc, w, i, kk);
    if (mk_side) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int cl = 16 * w + kk + 4 * reg, rl = 16 * t + i;
                side_w[cl * 64 + rl] = (r0 + rl < A.rend_next && cl < A.nbe_next) ? cv[4 * t + reg] - acc[t][reg] : 0.0;
            }
    }
    if (mk_side && dq_w) { // this panel's contribution to the NEXT launch's diagonal block (p+2, p+2), off that launch's chain:
                             // L(i, p) is in Ls; Z(i, p) = L(i, p) D takes Z(p+1, p)'s place
        put_acc(Zs, accI, w, i, kk, lds);
        __syncthreads();
        d4 accP[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) accP[t] = (d4){0.0, 0.0, 0.0, 0.0};
        tile_product(Ls, Zs, w, i, kk, accP);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) dq_w[(16 * w + kk + 4 * reg) * 64 + 16 * t + i] = accP[t][reg];
    }
    LVBA_STAMP(A, st_prob, (int)t_row, 7);
}

// ---------------------------------------------------------------------------------------------- row 1's q_extra tile, on its own
// A(p+2, p+2) -= L(p+2, q) Z(p+2, q)^T (lower triangle): inputs of the launch before, nobody else in this launch touches the tile
__device__ __forceinline__ void qx_diag_role(double *lds, const LdltMat &M, const Step2Args &A, const double *__restrict__ Zq)
{
    double *Ls = lds, *Zs = lds + 64 * LVBA_TS;
    const PanelGeo &p = A.p, &q = A.q;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, row = tid & 63, i = lane & 15, kk = lane >> 4;
    const int64_t r0 = p.w0 + 64; // tile row p + 2 = its tile column
    if (!(A.has_q && r0 < q.rend)) return;
    double va[16], vb[16];
    load_panel_tile(M, r0, q.k, q.rend, q.nbe, w, row, va);      // L(p+2, q)
    load_z_tile(Zq, A.ldz, r0, q.w0, q.rend, q.nbe, w, row, vb); // Z(p+2, q)
    double c2[16];
    load_c_tile(M, r0, r0, w, i, kk, c2); // (unmasked: what is not stored below is not used)
    stage_tile(Ls, va, w, row);
    stage_tile(Zs, vb, w, row);
    __syncthreads();
    d4 accx[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) accx[t] = (d4){0.0, 0.0, 0.0, 0.0};
    tile_product(Ls, Zs, w, i, kk, accx);
    store_c_tile(M, r0, r0, q.rend, q.rend, true, c2, accx, w, i, kk);
}

// ---------------------------------------------------------------------------------------------- the forward-substitution passengers
__device__ __forceinline__ void fwd_passenger(double *lds, const LdltMat &M, const FwdPassenger &F, const double *__restrict__ dvec,
                                              int64_t idx)
{
    double *Ls = lds, *Zs = lds + 64 * LVBA_TS;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, row = tid & 63, i = lane & 15, kk = lane >> 4;
    const int64_t nct = F.ldb / 64, ldb = F.ldb;
    const int64_t ny = F.y_on ? nct : 0;
    double *__restrict__ B = F.B;
    double *__restrict__ Y = F.Y;
    double va[16], vb[16];
    d4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
    if (idx < ny) { // ---- Y role, column block idx
        const int64_t j0 = 64 * idx, k = F.a_k;
        const int nbe = F.a_nbe;
        if (F.has_prev) { // B_a -= L(a, a - 1) Y_{a-1}
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int m = w + 4 * it;
                va[it] = m < F.am_nbe ? Y[(F.am_k + m) * ldb + j0 + row] : 0.0; // Y_{a-1} as [m = c][x = jj]
            }
            load_panel_tile(M, k, F.am_k, F.am_rend, F.am_nbe, w, row, vb);      // L(a, a - 1) as [m = c][x = r]
            stage_tile(Ls, va, w, row);
            stage_tile(Zs, vb, w, row);
        }
#pragma unroll
        for (int it = 0; it < 16; ++it) vb[it] = F.Ga[tid + 256 * it]; // G_a[m][c]: thread (c = row, m = w + 4 it)
        const double dk = tid < nbe ? dvec[k + tid] : 0.0;
        // B_a in the products' result layout: (jj = 16 t + i, r = 16 w + kk + 4 reg)
        d4 bn[4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int r = 16 * w + kk + 4 * reg;
#pragma unroll
            for (int t = 0; t < 4; ++t) bn[t][reg] = r < nbe ? B[(k + r) * ldb + j0 + 16 * t + i] : 0.0;
        }
        if (F.has_prev) {
            __syncthreads();
            tile_product(Ls, Zs, w, i, kk, acc); // (jj = 16 t + i, r = 16 w + kk + 4 reg): sum_c L[r][c] Y[c][jj]
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int r = 16 * w + kk + 4 * reg;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    bn[t][reg] -= acc[t][reg];
                    if (r < nbe) B[(k + r) * ldb + j0 + 16 * t + i] = bn[t][reg];
                    acc[t][reg] = 0.0;
                }
            }
            __syncthreads();
        }
        put_acc(Ls, bn, w, i, kk, nullptr); // Ls[m = r][x = jj] = B_a
        stage_tile(Zs, vb, w, row);         // Zs[m][c] = G_a
        if (tid < 64) pad_at(lds, LVBA_PAD_DP + tid) = dk;
        __syncthreads();
        tile_product(Ls, Zs, w, i, kk, acc); // (jj = 16 t + i, c = 16 w + kk + 4 reg): (G^T B)[c][jj]
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int c = 16 * w + kk + 4 * reg;
            const double dc = pad_at(lds, LVBA_PAD_DP + c);
            if (c < nbe)
#pragma unroll
                for (int t = 0; t < 4; ++t) Y[(k + c) * ldb + j0 + 16 * t + i] = acc[t][reg] * dc;
        }
        return;
    }
    // ---- U role: tile row ti >= 2 of panel c's window, column block
    if (!F.u_on) return;
    idx -= ny;
    const int64_t ti = 2 + idx % (F.c_T - 1), j0 = 64 * (idx / (F.c_T - 1));
    if (j0 >= ldb) return;
    const int64_t r0 = F.c_w0 + 64 * (ti - 1);
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int m = w + 4 * it;
        va[it] = m < F.c_nbe ? Y[(F.c_k + m) * ldb + j0 + row] : 0.0; // Y_c as [m = c][x = jj]
    }
    load_panel_tile(M, r0, F.c_k, F.c_rend, F.c_nbe, w, row, vb);       // L(tile, c) as [m = c][x = r]
    stage_tile(Ls, va, w, row);
    stage_tile(Zs, vb, w, row);
    __syncthreads();
    tile_product(Ls, Zs, w, i, kk, acc); // (jj = 16 t + i, r = 16 w + kk + 4 reg)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int64_t r = r0 + 16 * w + kk + 4 * reg;
        if (r < F.c_rend)
#pragma unroll
            for (int t = 0; t < 4; ++t) B[r * ldb + j0 + 16 * t + i] -= acc[t][reg];
    }
}

// the same on its own: the panels no step launch is left to carry (the last ones of a factorisation)
__global__ __launch_bounds__(256, 2) void ldlt_fwd_kernel(LdltMat M, FwdPassenger F, const double *__restrict__ dvec)
{
    __shared__ double lds[LVBA_K3_LDS];
    fwd_passenger(lds, M, F, dvec, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------- one launch
// Block order: the chain workgroups of all problems first, then the row workgroups, then the bulk jobs' workgroups alternating
// between the problems.  big: 128 x 64 bulk tiles (bulk_tile_128, 32-bit buffer offsets) / 64 x 64 tiles (update_tile[2], 64-bit
// pointers: matrices of 4 GB and more).  80 KB of LDS: two workgroups per CU.
template <bool big, bool dfr>
__global__ __launch_bounds__(256, 2) void ldlt_step2_kernel(const Step2Args A)
{
    __shared__ double lds[LVBA_K3_LDS];
    static_assert(LVBA_K1B_LDS <= LVBA_K3_LDS && LVBA_K3B_LDS <= LVBA_K3_LDS && LVBA_PAD_RED + 256 <= 1024, "LDS budget of the roles");
    // Kernel arguments are fetched where they are first used, one dependent scalar round trip after the other (the launch's
    // arguments are cold in every XCD's caches): read as the source was until round 6, a bulk workgroup went through five of them
    // and a 64-bit software division before its first load left (stamps: 2.2 us from its first instruction to "first fetches
    // out").  The words every workgroup's dispatch needs are requested here in ONE batch (LVBA_FETCH pins the values in scalar
    // registers at this point), the problem index is a bit of the block index (nprob is 1 or 2), and each branch below fetches
    // what it needs the same way.
    const int nprob = A.nprob, roles = A.roles, Tp = A.p.T, qxh = A.qx_helper, resv_at = A.resv_at, resv_n = A.resv_n, njobs = A.njobs;
    const int nwg0 = (int)A.job[0].nwg;
    const int64_t sW = A.sW, sA = A.sA, ldz = A.ldz;
    LdltMat M = A.M;
    LVBA_FETCH("s"(nprob), "s"(roles), "s"(Tp), "s"(qxh), "s"(resv_at), "s"(resv_n), "s"(njobs), "s"(nwg0), "s"(sW), "s"(sA), "s"(ldz), "s"(M.a),
               "s"(M.ld));
    const int nrole = roles ? Tp + (qxh ? 1 : 0) : 0, nfac = nrole * nprob;
    int bid = (int)blockIdx.x;
    if (resv_n > 0 && bid >= resv_at) {
        if (bid < resv_at + resv_n) return; // the seat next to a chain workgroup stays empty
        bid -= resv_n;
    }
    const int bb = bid < nfac ? bid : bid - nfac;
    const int prob = nprob == 2 ? (bb & 1) : 0;
    int bx = nprob == 2 ? (bb >> 1) : bb;
    const int64_t wo = prob ? sW : 0;
    if (prob) M.a += sA;
    if (bid < nfac) {
        if (bx == 0)
            chain_role(lds, M, A, A.Gp + wo, A.Gn + wo, A.dvec + wo, A.b + wo, A.Zp + wo, A.Zq ? A.Zq + wo : nullptr,
                       A.dq_r ? A.dq_r + wo : nullptr, prob + A.stamp_prob);
        else if (bx == Tp) qx_diag_role(lds, M, A, A.Zq ? A.Zq + wo : nullptr);
        else row_role<dfr>(lds, M, A, bx, A.Gp + wo, A.dvec + wo, A.b + wo, A.Zp + wo, A.Zq ? A.Zq + wo : nullptr, A.side_r + wo, A.side_w + wo,
                      A.dq_w ? A.dq_w + wo : nullptr, prob + A.stamp_prob);
        return;
    }
#ifdef LVBA_STAMPS
    const int st_bulk = bid - nfac;
    if (threadIdx.x == 0 && A.stamp_id >= 0 && A.stamp_id < LVBA_ST_LAUNCHES && st_bulk < LVBA_ST_BULK)
        g_lvba_bulk_stamps[A.stamp_id][st_bulk][0] = __builtin_amdgcn_s_memrealtime();
    struct StampEnd {
        const Step2Args &A; int b;
        __device__ ~StampEnd() {
            if (threadIdx.x == 0 && A.stamp_id >= 0 && A.stamp_id < LVBA_ST_LAUNCHES && b < LVBA_ST_BULK)
                g_lvba_bulk_stamps[A.stamp_id][b][1] = __builtin_amdgcn_s_memrealtime();
        }
    } st_end{A, st_bulk};
#endif
    int jsel = 0;
    if (njobs > 0 && bx >= nwg0) { bx -= nwg0; jsel = 1; }
    if (jsel < njobs) {
        const BulkJob &J = A.job[jsel];
        // this job's words, one batch
        const int64_t ok_ = J.o.k, ow0 = J.o.w0, orend = J.o.rend, ek_ = J.e.k, ew0 = J.e.w0, erend = J.e.rend, ca = J.ca, cb = J.cb, nwg = J.nwg;
        const int onbe = J.o.nbe, oT = J.o.T, enbe = J.e.nbe, pair = J.pair;
        const double *Zo_ = J.Zo, *Ze_ = J.Ze;
        LVBA_FETCH("s"(ok_), "s"(ow0), "s"(orend), "s"(ek_), "s"(ew0), "s"(erend), "s"(ca), "s"(cb), "s"(nwg), "s"(onbe), "s"(oT), "s"(enbe), "s"(pair),
                   "s"(Zo_), "s"(Ze_));
        if (bx < nwg) {
            const double *Zo = Zo_ + wo, *Ze = pair ? Ze_ + wo : nullptr;
            if constexpr (big) {
                int64_t R0, tj;
                const PanelRef po{ok_, ow0, orend, onbe, Zo}, pe{ek_, ew0, erend, enbe, Ze};
                if (!pair_decode(bx, ca, cb, (int64_t)oT - 1, R0, tj)) return;
#ifdef LVBA_STAMPS
                unsigned long long *bst = (A.stamp_id >= 0 && A.stamp_id < LVBA_ST_LAUNCHES && st_bulk < LVBA_ST_BTILES) ? g_lvba_bulk_marks[A.stamp_id][st_bulk] : nullptr;
                if (pair) bulk_tile_128<4>(lds, M, po, pe, ldz, R0, tj, bst);
                else bulk_tile_128<2>(lds, M, po, pe, ldz, R0, tj, bst);
#else
                if (pair) bulk_tile_128<4>(lds, M, po, pe, ldz, R0, tj);
                else bulk_tile_128<2>(lds, M, po, pe, ldz, R0, tj);
#endif
            } else {
                int64_t ti, tj;
                col_decode(ca + bx, (int64_t)oT - 1, ti, tj);
                if (pair) update_tile2(lds, M, ok_, onbe, ow0, orend, Zo, ek_, enbe, ew0, erend, Ze, ldz, ti + 1, tj + 1);
                else update_tile(lds, M, ok_, onbe, ow0, orend, Zo, ldz, ti + 1, tj + 1);
            }
            return;
        }
        bx -= (int)nwg;
    }
    if (A.fwd.on && prob == 0) fwd_passenger(lds, M, A.fwd, A.dvec, bx); // (bx: what the jobs left of the block index)
}
