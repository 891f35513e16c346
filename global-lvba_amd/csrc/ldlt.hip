// ldlt.hip -- damped normal-equation solve (H + u*diag(H)) dx = -g on gfx950, fp64.
//
// Replaces the dense->triplet scan + Eigen::SimplicialLDLT of BALM2::damping_iter (reference
// include/BALM/bavoxel.hpp:692-710).  Like SimplicialLDLT it is an UNPIVOTED LDL^T of the lower
// triangle (the exact second-order Hessian may be indefinite; Cholesky would fail where the reference
// succeeds).  Right-looking blocked algorithm, panel width NB = 64, on a column-major lower matrix that
// is either dense (ld = n) or LAPACK lower-band storage (ld = ldab-1) -- the same kernels serve both,
// the band only limits the row window [k+NB, k+NB+bw) each panel touches.
//
//   per panel k:
//     diag + panel    every one of the panel's T workgroups factorises the 64x64 diagonal block itself (16-column blocked
//                     LDL^T with an identity appended as extra ROWS, so the same elimination yields G = L11^-T D^-1, which
//                     turns every later triangular solve with this block into a product), then forms its 64-row tile of
//                     L21 = A21 * G (fp64 MFMA 16x16x4), Z = L21*D, y_k = D G^T b_k, b -= L21*y_k.
//     update          per 64x64 lower tile of the window: A22 -= L21 * Z^T (fp64 MFMA 16x16x4); the first tile column as a
//                     launch of its own, the rest in the same launch as the NEXT panel's diag + panel work.
//   backward, per panel from the last: x_k = G D (G^T b_k - acc_k), then acc[c] += A(k:k+64, c)^T x_k
//   for the <= bw columns left of the panel (right-looking, one launch per panel).
//   The whole static launch sequence is captured once into a hipGraph (block_system.hip).
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
#include "lvba_internal.h"
#include "ldlt_schedule.h"
#include "../../include/lvba_hip.h" // status codes

namespace lvba {

typedef double d4 __attribute__((ext_vector_type(4)));
#define LVBA_TS 80 // LDS tile stride (doubles): 80 = 16 mod 32 -> MFMA operand reads are bank-conflict free

// Twisted ("burn at both ends") form, tw.m > 0: the band matrix is split into T = [0, m), S = [m, n - m), B = [n - m, n) with
// |S| >= bw, so that T and B are not coupled.  Matrix 1 is the leading block [0, n - m) in natural order, matrix 2 the
// trailing block [m, n) in REVERSED order (index i' = n - 1 - i), both of size n1 = n - m in their own band storage (a,
// a + sA; workspace, workspace + sW).  The S x S entries go to matrix 1 only: matrix 2 collects the Schur complement of B
// there, starting from zero.
struct LdltTwist {
    int64_t m, n1, sA, sW; // m = 0: plain top-down factorisation
    unsigned long long *x2;
};
__global__ void ldlt_prepare_kernel(LdltMat M, const double *__restrict__ Hblk, int band_blocks, int n_poses,
                                    const double *__restrict__ g, const double *__restrict__ u_dev,
                                    double *__restrict__ b, unsigned long long *__restrict__ x, unsigned long long x_fill,
                                    LdltTwist tw, const int32_t *__restrict__ grp, int vectors_only, int *__restrict__ status)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) status[0] = 0; // (no memset node of its own in the solve's graph)
    // grp != nullptr: the damping of pose block J is u_dev[grp[J]] (independent groups of poses, each with its own LM state)
    const double u0 = u_dev[0];
    const int64_t Bb1 = band_blocks + 1;
    const int64_t total = vectors_only ? 0 : (int64_t)n_poses * Bb1 * 36; // band storage: ldlt_prepare_band_kernel fills the matrix
    const int64_t gid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, gsz = (int64_t)gridDim.x * blockDim.x;
    const int64_t n = M.n, n1 = tw.m > 0 ? tw.n1 : n;
    for (int64_t e = gid; e < total; e += gsz) {
        const int64_t slot = e / 36;
        const int el = (int)(e - slot * 36);
        const int64_t J = slot / Bb1, dI = slot - J * Bb1, I = J + dI;
        if (I >= n_poses) continue;
        const int c = el / 6, r = el - c * 6;
        if (dI == 0 && r < c) continue;
        double v = Hblk[e];
        const double uj = grp ? u_dev[grp[J]] : u0;
        // a group whose damping is negative is out of the game (its refinement has ended): identity block, zero right-hand
        // side -- it cannot produce a zero pivot any more, and its step is zero (groups are block diagonal: I and J share it)
        if (uj < 0.0) v = (dI == 0 && r == c) ? 1.0 : 0.0;
        else if (dI == 0 && r == c) v += uj * v;
        const int64_t R = 6 * I + r, C = 6 * J + c;
        if (R < n1) M.a[R + C * M.ld] = v;
        else M.a[tw.sA + (n - 1 - C) + (n - 1 - R) * M.ld] = v; // row in B: reversed and transposed into the lower triangle
    }
    for (int64_t a = gid; a < n; a += gsz) {
        if (a < n1) b[a] = (grp && u_dev[grp[a / 6]] < 0.0) ? 0.0 : -g[a];
        x[a] = x_fill; // the backward chain kernel's "not yet written" mark
    }
    if (tw.m > 0)
        for (int64_t a = gid; a < n1; a += gsz) {
            const int64_t i = n - 1 - a;
            b[tw.sW + a] = (i >= n1 && !(grp && u_dev[grp[i / 6]] < 0.0)) ? -g[i] : 0.0; // the S part of matrix 2's right-hand side only collects updates
            tw.x2[a] = x_fill;
        }
}

// Band storage, destination-major: one thread per STORED entry of the columns the two matrices use, the zeros included (slack
// rows below the band, the S x S block of matrix 2) -- every store is part of a contiguous column segment, and no memset of the
// 0.5 GB store runs first (the columns nobody uses are zeroed once, at allocation).  A workgroup stages 28 blocks of ONE block
// column (matrix 1) or block row (matrix 2: its columns are the original's rows, reversed) through LDS and writes the six
// columns they belong to: every block of the Hessian store is read once per matrix.  (One workgroup per column read every block
// six times, from six XCDs; the element-major kernel above scatters the B part's entries one 8-byte store per column on top of a
// memset: 0.064 + 0.164 ms at C3.)
#define LVBA_PB_BLOCKS 28
#define LVBA_PB_ROWS (6 * LVBA_PB_BLOCKS)
__global__ void __launch_bounds__(256)
ldlt_prepare_band_kernel(LdltMat M, const double *__restrict__ Hblk, int band_blocks, int n_poses,
                         const double *__restrict__ u_dev, LdltTwist tw, const int32_t *__restrict__ grp)
{
    __shared__ double blk[LVBA_PB_BLOCKS * 36];
    const int64_t Bb1 = band_blocks + 1, n = M.n, n1 = tw.m > 0 ? tw.n1 : n;
    const int ldab = (int)(M.ld + 1);
    const bool second = blockIdx.z != 0;
    const int64_t P = second ? tw.m / 6 + (int64_t)blockIdx.x : (int64_t)blockIdx.x; // block column J (matrix 1) / block row I (matrix 2)
    if (P >= n_poses || (!second && 6 * P >= n1)) return;
    const int d0 = (int)blockIdx.y * LVBA_PB_BLOCKS; // first block offset dI (matrix 1: I = P + dI) / dJ (matrix 2: J = P - dJ)
    for (int i = threadIdx.x; i < LVBA_PB_BLOCKS * 36; i += 256) {
        const int bq = i / 36, el = i - 36 * bq;
        const int64_t dd = d0 + bq;
        double v = 0.0;
        if (dd <= band_blocks) {
            if (!second) { if (P + dd < n_poses) v = Hblk[(P * Bb1 + dd) * 36 + el]; }
            else if (P - dd >= 0) v = Hblk[((P - dd) * Bb1 + dd) * 36 + el];
        }
        blk[i] = v;
    }
    __syncthreads();
    const double uj = grp ? u_dev[grp[P]] : u_dev[0];
    const bool dead = uj < 0.0; // a finished group: identity block, see ldlt_prepare_kernel
    for (int idx = threadIdx.x; idx < 6 * LVBA_PB_ROWS; idx += 256) {
        const int e = idx / LVBA_PB_ROWS, t = idx - e * LVBA_PB_ROWS, bq = t / 6, w = t - 6 * bq;
        const int64_t X = 6 * P + e; // the column of matrix 1 / the ORIGINAL row whose entries form a column of matrix 2
        int d;
        double v;
        if (!second) { // element (r = w, c = e) of block (P + dI, P): offset d = 6 dI + r - c
            d = 6 * d0 + t - e;
            if (X >= n1 || d < 0 || d >= ldab) continue;
            v = X + d < n1 ? blk[bq * 36 + e * 6 + w] : 0.0; // rows of the B part belong to matrix 2
        } else {       // element (r = e, c = 5 - w) of block (P, P - dJ): offset d = 6 dJ + r - c, ascending in t
            d = 6 * d0 + t + e - 5;
            if (X < tw.m || X >= n || d < 0 || d >= ldab) continue;
            v = X >= n1 ? blk[bq * 36 + (5 - w) * 6 + e] : 0.0; // the S x S block of matrix 2 starts from zero
        }
        if (dead) v = d == 0 && (second ? X >= n1 : true) ? 1.0 : 0.0;
        else if (d == 0) v += uj * v;
        if (!second) M.a[X * (int64_t)ldab + d] = v;
        else M.a[tw.sA + (n - 1 - X) * (int64_t)ldab + d] = v;
    }
}

// LVBA_CHECK_BAND=1 (debugging; use with LVBA_NO_GRAPH=1): the band store is zeroed ONCE, at allocation (block_system.hip), and
// every solve rewrites only the columns its two matrices use -- [0, n1) of each.  That holds as long as no factorisation or
// update kernel ever stores outside those columns: this kernel counts the non-zero entries of the rest (columns [n1, n] of both
// matrices and the slack behind them) before a solve starts; ldlt_solve reports a non-zero count as LVBA_ERR_STATE.
__global__ void ldlt_check_untouched_kernel(const double *__restrict__ a, int64_t ldab, int64_t n, int64_t n1, int64_t sA,
                                            int64_t total, int two, unsigned long long *__restrict__ cnt)
{
    const int64_t gid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, gsz = (int64_t)gridDim.x * blockDim.x;
    unsigned long long bad = 0;
    for (int64_t e = gid; e < total; e += gsz) {
        bool untouched;
        if (e < sA) untouched = e / ldab >= n1;                       // matrix 1: columns n1 .. n
        else if (two && e < 2 * sA) untouched = (e - sA) / ldab >= n1; // matrix 2: columns n1 .. n
        else untouched = true;                                        // slack (and matrix 2's room when it is not used)
        if (untouched && a[e] != 0.0) ++bad;
    }
    if (bad) atomicAdd(cnt, bad);
}

// After both ends have been eliminated: the Schur complement that matrix 2 (reversed) collected on S is added to matrix 1's
// S block, likewise the right-hand side.
// (32 x 32 tiles through LDS: matrix 2 holds the block transposed and reversed, so reading it in matrix 1's order is one 8-byte
// load per column -- 47 us for 3.5 M entries; a tile is read along ITS columns and added along matrix 1's)
__global__ __launch_bounds__(256) void ldlt_twist_merge_kernel(LdltMat M, LdltTwist tw, double *__restrict__ b)
{
    __shared__ double tile[32][33];
    const int64_t n = M.n, s0 = tw.m, s1 = tw.n1, ns = s1 - s0;
    const int64_t T = (ns + 31) / 32, ntiles = T * (T + 1) / 2;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; // 32 x 8
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        // lower tile (tr >= tc) number t of the column-major enumeration
        int64_t tc = 0, rem = t;
        while (rem >= T - tc) { rem -= T - tc; ++tc; }
        const int64_t tr = tc + rem;
        const int64_t R0 = s0 + 32 * tr, C0 = s0 + 32 * tc;
        if (R0 - (C0 + 31) > M.bw) continue; // wholly below the band (uniform over the workgroup)
#pragma unroll
        for (int k = 0; k < 4; ++k) { // matrix 2: entry (R, C) at (n-1-C) + (n-1-R) ld -- C fastest
            const int64_t R = R0 + ty + 8 * k, C = C0 + tx;
            double v = 0.0;
            if (R < s1 && C < s1 && R >= C && R - C <= M.bw) v = M.a[tw.sA + (n - 1 - C) + (n - 1 - R) * M.ld];
            tile[ty + 8 * k][tx] = v;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) { // matrix 1: R fastest
            const int64_t R = R0 + tx, C = C0 + ty + 8 * k;
            if (R < s1 && C < s1 && R >= C && R - C <= M.bw) M.a[R + C * M.ld] += tile[tx][ty + 8 * k];
        }
        __syncthreads();
    }
    const int64_t gid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, gsz = (int64_t)gridDim.x * blockDim.x;
    for (int64_t a = s0 + gid; a < s1; a += gsz) b[a] += b[tw.sW + (n - 1 - a)];
}
// x of the S part, reversed, into matrix 2's solution vector (its backward chain starts from there)
__global__ void ldlt_twist_xs_kernel(LdltTwist tw, int64_t n, const double *__restrict__ x)
{
    const int64_t a = tw.m + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; // index in matrix 2
    if (a < tw.n1) tw.x2[a] = (unsigned long long)__double_as_longlong(x[n - 1 - a]);
}
// matrix 2's B part back into the caller's order
__global__ void ldlt_twist_xb_kernel(LdltTwist tw, int64_t n, double *__restrict__ x)
{
    const int64_t a = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (a < tw.m) x[n - 1 - a] = __longlong_as_double((long long)tw.x2[a]);
}

// Multi-rank form: rank 0 eliminates T, rank 1 eliminates B, each on its own GPU.  What they exchange is the S block of the
// band storage and the S part of the right-hand side: E = [S x (bw + 1) entries | |S| entries].  side 0 packs matrix 1's
// (original entries + T's Schur complement), side 1 matrix 2's reversed (B's Schur complement), side 2 zeros; after the
// all-reduce (a sum of two non-zero operands: the same a + b the merge kernel forms) every rank unpacks E into matrix 1.
__global__ void ldlt_twist_pack_kernel(LdltMat M, LdltTwist tw, const double *__restrict__ b, int side, double *__restrict__ E)
{
    const int64_t n = M.n, s0 = tw.m, s1 = tw.n1, ns = s1 - s0, bw1 = M.bw + 1;
    const int64_t total = ns * bw1;
    const int64_t gid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, gsz = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = gid; e < total + ns; e += gsz) {
        double v = 0.0;
        if (e < total) {
            const int64_t cc = e / bw1, d = e - cc * bw1;
            const int64_t C = s0 + cc, R = C + d;
            if (R < s1) v = side == 0 ? M.a[R + C * M.ld] : side == 1 ? M.a[tw.sA + (n - 1 - C) + (n - 1 - R) * M.ld] : 0.0;
        } else {
            const int64_t a = s0 + (e - total);
            v = side == 0 ? b[a] : side == 1 ? b[tw.sW + (n - 1 - a)] : 0.0;
        }
        E[e] = v;
    }
}
__global__ void ldlt_twist_unpack_kernel(LdltMat M, LdltTwist tw, double *__restrict__ b, const double *__restrict__ E)
{
    const int64_t s0 = tw.m, s1 = tw.n1, ns = s1 - s0, bw1 = M.bw + 1;
    const int64_t total = ns * bw1;
    const int64_t gid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, gsz = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = gid; e < total + ns; e += gsz) {
        if (e < total) {
            const int64_t cc = e / bw1, d = e - cc * bw1;
            const int64_t C = s0 + cc, R = C + d;
            if (R < s1) M.a[R + C * M.ld] = E[e];
        } else
            b[s0 + (e - total)] = E[e];
    }
}
// x parts that this rank did not compute are zeroed before the ranks' solutions are summed (side 0 keeps [0, n1), side 1
// keeps [n1, n), side 2 nothing)
__global__ void ldlt_twist_xmask_kernel(LdltTwist tw, int64_t n, int side, double *__restrict__ x)
{
    const int64_t a = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (a >= n) return;
    const bool keep = side == 0 ? a < tw.n1 : side == 1 ? a >= tw.n1 : false;
    if (!keep) x[a] = 0.0;
}

// ---------------------------------------------------------------------------------------------- K1
// The pivot reciprocal is v_rcp_f64 + 2 Newton steps instead of an IEEE division.
__device__ __forceinline__ double fast_rcp(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    r = fma(r, fma(-d, r, 1.0), r);
    r = fma(r, fma(-d, r, 1.0), r);
    return r;
}

#ifdef LVBA_K1_TIMING
__device__ unsigned long long g_k1b_clk[16];
#define LVBA_K1B_STAMP(i) do { if (threadIdx.x == 0) g_k1b_clk[i] = __builtin_readcyclecounter(); } while (0)
#else
#define LVBA_K1B_STAMP(i)
#endif
#ifdef LVBA_K1_TIMING
__device__ unsigned long long g_bulk_clk[16];
__device__ int g_bulk_stamp_block = -1;
#define LVBA_BULK_STAMP(i) do { if (threadIdx.x == 0 && (int)blockIdx.x == g_bulk_stamp_block) g_bulk_clk[i] = __builtin_readcyclecounter(); } while (0)
#else
#define LVBA_BULK_STAMP(i)
#endif
#define LVBA_PIN(x) asm volatile("" : "+v"(x)) // keep the value computed HERE (LLVM otherwise sinks it to its first use)

// ------------------------------------------------------------------------------------------ K1, blocked
// LDL^T of the 64x64 diagonal block together with G = L11^-T D^-1 (an identity appended as 64 extra ROWS and carried
// through the same elimination), organised so that the serial chain only ever spans a 16x16 block held in ONE wavefront's
// registers (an earlier row-per-lane form published every column through LDS and cost 22 us per block; this one 14 us): the 64x64 block and the 64 appended identity rows live in LDS
// (W[128][64]); per 16-column block step
//   diag   wave 0: lanes 0..15 hold the block's rows, lanes 16..31 the matching identity rows; 16 compile-time steps, the
//          pivot row reaches the other lanes through v_readlane (no LDS round trip, no barrier); yields d, the block's
//          G rows and G11 = L11^-T D11^-1;
//   panel  3 waves: the 48 rows below / left over (block rows still to come + identity rows of finished blocks) times G11,
//          fp64 MFMA 16x16x4 -- a triangular solve turned into a product, as everywhere else in this file;
//   update 4 waves: trailing 64 x (48 - 16 s) block -= X (X D)^T, fp64 MFMA.
// 64 pivots still follow one another, but each costs ~(16 - j) readlane+FMA pairs instead of an LDS publish / flag /
// read-back of a 64-entry column.
// (Round 3 tried the opposite extreme -- the whole block by the symmetric sweep operator, every thread 16 entries of its column
// in registers, ONE barrier and one 64-double pivot row per pivot, which also yields A11^-1 and turns the panel into a block
// LDL^T step.  Measured with tools/solver_microbench: 22.0 us against 13.3 us for this form -- a workgroup barrier + LDS round
// trip per pivot is ~700 cycles, the in-wavefront chain here ~290 per pivot.  Withdrawn.)
#define LVBA_W1S 130 // column stride of W (doubles)
#define LVBA_Z1S 50  // column stride of the Z^T tile (doubles)
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
#ifndef LVBA_K1B_V2
template <int J>
__device__ __forceinline__ void k1b_step(double (&a)[16], int lane, double &rd)
{
    const bool done = lane < 16 && lane <= J; // finished block rows: l = 0 leaves them untouched
    const double u = a[J];
    const double l = done ? 0.0 : u * rd;
    a[J] = done ? u : l;
    if constexpr (J + 1 < 16) {
        a[J + 1] = fma(-l, readlane_f64(u, J + 1), a[J + 1]);
        // next pivot: start its reciprocal now, refine it after the rest of the row (the FMAs below do not depend on it
        // and fill the latency of v_rcp_f64 and of the readlanes)
        const double pn = readlane_f64(a[J + 1], J + 1);
        double r = __builtin_amdgcn_rcp(pn);
        LVBA_PIN(r);
#pragma unroll
        for (int c = J + 2; c < 16; ++c) {
            a[c] = fma(-l, readlane_f64(u, c), a[c]);
            LVBA_PIN(a[c]);
        }
        r = fma(r, fma(-pn, r, 1.0), r);
        r = fma(r, fma(-pn, r, 1.0), r);
        rd = r;
    }
}
template <int... Js>
__device__ __forceinline__ void k1b_steps(std::integer_sequence<int, Js...>, double (&a)[16], int lane, double rd)
{
    (k1b_step<Js>(a, lane, rd), ...);
}
#else
// Round 4 experiment (LVBA_K1B_V2; measured: no gain, 4696 vs 4672 cycles per 16-column block, and ~70 more scalar registers --
// the pivot chain is bound by the ISSUE of its ~34 instructions per column at ~8.5 cycles each, not by the readlane -> FMA latency):
// the column's broadcasts BATCHED.  Column J's update needs A[c][J] for every later column c in every lane: lane c holds
// it (a[J] of lane c), so it travels through a v_readlane pair into scalar registers.  The form above read it where it was used
// -- readlane, readlane, FMA per entry, all through ONE scalar register pair, i.e. a dependent triple whose readlane -> FMA latency
// was paid 120 times per 16-column block (~22 cycles per entry: the larger part of the ~297 cycles a pivot cost).  Here the
// broadcasts of column J + 1 are issued as soon as its entries are final -- each lane's a[J+1] after the FIRST FMA of column J --
// into scalar registers of their own, back to back, while the rest of column J's FMAs and the next pivot's reciprocal run.
template <int J>
__device__ __forceinline__ void k1b_step(double (&a)[16], int lane, double &rd, double (&uc)[16])
{
    const bool done = lane < 16 && lane <= J; // finished block rows: l = 0 leaves them untouched
    const double u = a[J];
    const double l = done ? 0.0 : u * rd;
    a[J] = done ? u : l;
    if constexpr (J + 1 < 16) {
        a[J + 1] = fma(-l, uc[J + 1], a[J + 1]);
        const double pn = readlane_f64(a[J + 1], J + 1);
        double r = __builtin_amdgcn_rcp(pn);
        LVBA_PIN(r);
        double un[16];
#pragma unroll
        for (int c = J + 2; c < 16; ++c) un[c] = readlane_f64(a[J + 1], c); // column J + 1's broadcasts
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = J + 2; c < 16; ++c) {
            a[c] = fma(-l, uc[c], a[c]);
            LVBA_PIN(a[c]);
        }
        r = fma(r, fma(-pn, r, 1.0), r);
        r = fma(r, fma(-pn, r, 1.0), r);
        rd = r;
#pragma unroll
        for (int c = J + 2; c < 16; ++c) uc[c] = un[c];
    }
}
template <int... Js>
__device__ __forceinline__ void k1b_steps(std::integer_sequence<int, Js...>, double (&a)[16], int lane, double rd)
{
    double uc[16];
#pragma unroll
    for (int c = 1; c < 16; ++c) uc[c] = readlane_f64(a[0], c);
    (k1b_step<Js>(a, lane, rd, uc), ...);
}
#endif

#define LVBA_K1B_LDS (64 * LVBA_W1S + 256 + 16 * LVBA_Z1S + 64) // doubles
// diag_blocked_load: the 64x64 block at (k, k) into W (lower triangle; identity below row nbe) with the identity appended.
// diag_blocked_factor: the factorisation of what W holds (the look-ahead kernel fills W itself, from the registers its updates
// of the block end in).  Leaves d in dvs[64] and G[m][c] in W[c * LVBA_W1S + 64 + m]; ends on a __syncthreads().
__device__ __forceinline__ void diag_blocked_load(double *lds, LdltMat M, int64_t k, int nbe)
{
    double *W = lds;                      // (row, col) at col * LVBA_W1S + row; rows 64..127 = the appended identity
    const int tid = threadIdx.x;
    LVBA_K1B_STAMP(0);
    {
        // all 16 loads of a lane are issued before the first one is waited for (one memory latency instead of a chain of
        // load -> LDS store pairs)
        double vv[16];
        const int row = tid & 63;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int col = (tid >> 6) + 4 * it;
            double v = 0.0;
            if (row < nbe) {
                if (col <= row) v = M.a[(k + row) + (k + col) * M.ld];
            } else if (col == row)
                v = 1.0;
            vv[it] = v;
        }
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int col = (tid >> 6) + 4 * it;
            W[col * LVBA_W1S + row] = vv[it];
            W[col * LVBA_W1S + 64 + row] = (row == col) ? 1.0 : 0.0;
        }
    }
    __syncthreads();
}
__device__ __forceinline__ void diag_blocked_factor(double *lds, int nbe, int *__restrict__ status)
{
    double *W = lds;
    double *G11s = W + 64 * LVBA_W1S;     // [m][c]
    double *Zt = G11s + 256;              // [j][block row relative to c0 + 16] = X * d
    double *dvs = Zt + 16 * LVBA_Z1S;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    LVBA_K1B_STAMP(1);
    const int i15 = lane & 15, kk = lane >> 4;
    // ---- diag step of the 16 columns at c0: wavefront 0 only, no barrier inside
    auto diag_step = [&](int c0) {
        const int r = lane < 16 ? c0 + lane : 64 + c0 + (lane & 15);
        double a[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) a[c] = (lane < 32) ? W[(c0 + c) * LVBA_W1S + r] : 0.0;
        k1b_steps(std::make_integer_sequence<int, 16>{}, a, lane, fast_rcp(readlane_f64(a[0], 0)));
        if (lane < 16) {
            double dl = 0.0;
#pragma unroll
            for (int c = 0; c < 16; ++c) dl = (c == lane) ? a[c] : dl;
            dvs[c0 + lane] = dl;
            if (c0 + lane < nbe && (!(dl != 0.0) || !isfinite(dl))) status[0] = 1;
        } else if (lane < 32) {
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                W[(c0 + c) * LVBA_W1S + r] = a[c];
                G11s[(lane - 16) * 16 + c] = a[c];
            }
        }
    };
    if (w == 0) diag_step(0);
    __syncthreads();
    for (int s = 0; s < 4; ++s) {
        const int c0 = 16 * s;
        const int nb_rows = 48 - c0; // block rows still to come
        LVBA_K1B_STAMP(2 + 3 * s);
        // row tile of this wave in the panel / update steps: waves 0..2 -> the 48 panel rows, wave 3 -> the identity
        // rows of this block (their X is what the diag step wrote)
        const int base = (w < 3) ? ((16 * w < nb_rows) ? c0 + 16 + 16 * w : 64 + 16 * w - nb_rows) : 64 + c0;
        if (w < 3) { // ---- panel: X = A * G11
            d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double av = W[(c0 + 4 * q + kk) * LVBA_W1S + base + i15];
                const double bv = G11s[(4 * q + kk) * 16 + i15];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
            }
            // acc[r] = X[base + kk + 4r][c0 + i15]
            const double dj = dvs[c0 + i15];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                W[(c0 + i15) * LVBA_W1S + base + kk + 4 * r] = acc[r];
                if (16 * w < nb_rows) Zt[i15 * LVBA_Z1S + 16 * w + kk + 4 * r] = acc[r] * dj;
            }
        }
        __syncthreads();
        LVBA_K1B_STAMP(3 + 3 * s);
        // ---- update: C[base + i][c0 + 16 + 16 ct + n] -= sum_j X[base + i][c0 + j] * Z[16 ct + n][j].  A block-row tile
        // only needs its lower part (ct <= its own index); identity-row tiles need every column tile.
        const int ct_end = (w < 3 && 16 * w < nb_rows) ? w + 1 : nb_rows / 16;
        for (int ct = 0; ct < ct_end; ++ct) {
            d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double av = W[(c0 + 4 * q + kk) * LVBA_W1S + base + i15];
                const double bv = Zt[(4 * q + kk) * LVBA_Z1S + 16 * ct + i15];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) W[(c0 + 16 + 16 * ct + i15) * LVBA_W1S + base + kk + 4 * r] -= acc[r];
        }
        // look-ahead: wavefront 0's tile was the next diagonal block (rows c0+16.., column tile 0), which nobody else
        // touches -- its pivot chain runs while the other wavefronts finish their update tiles
        if (w == 0 && nb_rows > 0) diag_step(c0 + 16);
        __syncthreads();
        LVBA_K1B_STAMP(4 + 3 * s);
    }
}
__device__ __forceinline__ void diag_blocked_body(double *lds, LdltMat M, int64_t k, int nbe, int *__restrict__ status)
{
    diag_blocked_load(lds, M, k, nbe);
    diag_blocked_factor(lds, nbe, status);
}

// blockIdx.y: the problem of a two-ended factorisation.  blockIdx.x = 1 (look-ahead schedule, first launch of a phase): the
// side copy of the tile below the diagonal block, A(rows w0 .., columns k ..) as [m][row], masked like load_panel_tile.
__global__ __launch_bounds__(256) void ldlt_diag_blocked_kernel(LdltMat M, int64_t k, int nbe, double *__restrict__ G,
                                                               double *__restrict__ dvec, int *__restrict__ status,
                                                               int64_t sA, int64_t sW, double *__restrict__ side, int64_t rend)
{
    __shared__ double lds[LVBA_K1B_LDS];
    if (blockIdx.y) { M.a += sA; G += sW; dvec += sW; if (side) side += sW; } // the second problem of a two-ended factorisation
    if (blockIdx.x == 1) {
        const int row = threadIdx.x & 63, w = threadIdx.x >> 6;
        const int64_t r = k + nbe + row;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int m = w + 4 * it;
            side[m * 64 + row] = (r < rend && m < nbe) ? M.a[r + (k + m) * M.ld] : 0.0;
        }
        return;
    }
    diag_blocked_body(lds, M, k, nbe, status);
    const double *W = lds, *dvs = lds + 64 * LVBA_W1S + 256 + 16 * LVBA_Z1S;
    const int tid = threadIdx.x;
    if (tid < nbe) dvec[k + tid] = dvs[tid];
    for (int e = tid; e < 4096; e += 256) { // G[m][c], row-major
        const int c = e & 63, m = e >> 6;
        G[e] = W[c * LVBA_W1S + 64 + m];
    }
    LVBA_K1B_STAMP(14);
}

// ---------------------------------------------------------------------------------------------- K1 + K2
// K1 + K2 in one launch: every panel workgroup repeats the (cheap, 1-workgroup) diagonal factorisation itself instead of
// waiting for a separate kernel to publish G -- one kernel boundary and the G / d round trip through global memory less per
// panel, and the workgroup's A21 tile is already in registers when the factorisation ends.  Workgroup 0 also writes G and d
// (the backward pass needs them).
#define LVBA_K12_LDS (LVBA_K1B_LDS + 128 + 256) // doubles: the blocked factorisation's tables + b_k, y_k + partial sums
__device__ __forceinline__ void diagpanel_tile(double *lds, LdltMat M, int64_t k, int nbe, int64_t w0, int64_t rend,
                                               double *__restrict__ G, double *__restrict__ dvec, double *__restrict__ Zws,
                                               int64_t ldz, double *__restrict__ b, int *__restrict__ status, int64_t tile)
{
    double *bks = lds + LVBA_K1B_LDS, *ys = bks + 64;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int row = tid & 63;
    const int64_t r0 = w0 + 64 * tile;
    const int64_t r = r0 + row;
    double av[16];
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int m = w + 4 * it;
        av[it] = (r < rend && m < nbe) ? M.a[r + (k + m) * M.ld] : 0.0;
    }
    const double bk = (tid < nbe) ? b[k + tid] : 0.0;
    diag_blocked_body(lds, M, k, nbe, status);
    double *W = lds;
    const double *dvs = lds + 64 * LVBA_W1S + 256 + 16 * LVBA_Z1S;
    if (tile == 0) {
        if (tid < nbe) dvec[k + tid] = dvs[tid];
        for (int e = tid; e < 4096; e += 256) G[e] = W[(e & 63) * LVBA_W1S + 64 + (e >> 6)];
    }
    // rows 0..63 of W (the factored block itself) are dead now: the A21 tile is staged there as As[m][row] = W[m][row],
    // next to G[m][j] = W[j * LVBA_W1S + 64 + m] in rows 64..127
    double *As = W;
    if (tid < 64) bks[tid] = bk;
#pragma unroll
    for (int it = 0; it < 16; ++it) As[(w + 4 * it) * LVBA_W1S + row] = av[it];
    __syncthreads();
    double *red = bks + 128; // [4][64] partial sums (inside the b_k / y_k scratch area)
    { // y_k = L11^-1 b_k = D G^T b_k: lane (j, q) sums m in [16q, 16q+16), combined after the MFMA loop
        const int q = tid >> 6;
        double z = 0.0;
#pragma unroll
        for (int m = 0; m < 16; ++m) z += W[row * LVBA_W1S + 64 + 16 * q + m] * bks[16 * q + m];
        red[q * 64 + row] = z;
    }
    d4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
    const int i = lane & 15, kk = lane >> 4;
#pragma unroll 4
    for (int k0 = 0; k0 < 64; k0 += 4) {
        const double a = W[(16 * w + i) * LVBA_W1S + 64 + k0 + kk]; // G[m = k0+kk][j = 16w+i]
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const double bv = As[(k0 + kk) * LVBA_W1S + 16 * t + i]; // A21[row=16t+i][m]
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv, acc[t], 0, 0, 0);
        }
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) As[(16 * w + kk + 4 * reg) * LVBA_W1S + 16 * t + i] = acc[t][reg];
    if (tid < 64) ys[tid] = (tid < nbe) ? (red[tid] + red[64 + tid] + red[128 + tid] + red[192 + tid]) * dvs[tid] : 0.0;
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int j = w + 4 * it;
        if (r < rend && j < nbe) {
            const double v = As[j * LVBA_W1S + row];
            M.a[r + (k + j) * M.ld] = v;
            Zws[(r - w0) + j * ldz] = v * dvs[j];
        }
    }
    { // b[r] -= L21[row] . y_k, again four lanes per row
        const int q = tid >> 6;
        double sacc = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) sacc += As[(16 * q + j) * LVBA_W1S + row] * ys[16 * q + j];
        __syncthreads(); // red is being reused
        red[q * 64 + row] = sacc;
    }
    __syncthreads();
    if (tid < 64 && r < rend) b[r] -= red[tid] + red[64 + tid] + red[128 + tid] + red[192 + tid];
}
// blockIdx.y selects the problem of a twisted factorisation (0: matrix 1, 1: matrix 2 at a + sA / workspace + sW)
__global__ __launch_bounds__(256) void ldlt_diagpanel_kernel(LdltMat M, int64_t k, int nbe, int64_t w0, int64_t rend,
                                                             double *__restrict__ G, double *__restrict__ dvec,
                                                             double *__restrict__ Zws, int64_t ldz, double *__restrict__ b,
                                                             int *__restrict__ status, int64_t sA, int64_t sW)
{
    __shared__ double lds[LVBA_K12_LDS];
    if (blockIdx.y) { M.a += sA; G += sW; dvec += sW; Zws += sW; b += sW; }
    diagpanel_tile(lds, M, k, nbe, w0, rend, G, dvec, Zws, ldz, b, status, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------- K3
// every lower tile (ti >= tj) of the window.  At one 64x64 tile per workgroup the kernel moves 128 KB (C in and out, L, Z)
// per 0.52 MFLOP = 4 flop/B: it runs at the HBM bound (~5 TB/s -> ~20 TFLOP/s), not at the MFMA bound.
#define LVBA_K3_LDS (2 * 64 * LVBA_TS) // doubles
__device__ __forceinline__ void tri_decode(int64_t bidx, int64_t &ti, int64_t &tj) // bidx -> (ti >= tj)
{
    ti = (int64_t)((sqrt(8.0 * (double)bidx + 1.0) - 1.0) * 0.5);
    while (ti * (ti + 1) / 2 > bidx) --ti;
    while ((ti + 1) * (ti + 2) / 2 <= bidx) ++ti;
    tj = bidx - ti * (ti + 1) / 2;
}
// column-major enumeration of the lower tiles of a Tb x Tb triangle: column tj holds the Tb - tj tiles ti = tj .. Tb-1, columns
// one after the other (bidx -> (ti >= tj)).  A launch can then take a RANGE of tile columns -- the ones the next
// factorisations need first.
__host__ __device__ __forceinline__ int64_t col_start(int64_t tj, int64_t Tb) { return tj * Tb - tj * (tj - 1) / 2; }
__device__ __forceinline__ void col_decode(int64_t bidx, int64_t Tb, int64_t &ti, int64_t &tj)
{
    const double bq = (double)(2 * Tb + 1);
    tj = (int64_t)((bq - sqrt(bq * bq - 8.0 * (double)bidx)) * 0.5);
    if (tj < 0) tj = 0;
    if (tj > Tb - 1) tj = Tb - 1;
    while (tj > 0 && col_start(tj, Tb) > bidx) --tj;
    while (tj + 1 < Tb && col_start(tj + 1, Tb) <= bidx) ++tj;
    ti = tj + (bidx - col_start(tj, Tb));
}
__device__ __forceinline__ void update_tile(double *lds, LdltMat M, int64_t k, int nbe, int64_t w0, int64_t rend,
                                            const double *__restrict__ Zws, int64_t ldz, int64_t ti, int64_t tj)
{
    double *Ls = lds;                // [m][row of tile ti]
    double *Zs = lds + 64 * LVBA_TS; // [m][row of tile tj] (= column of the updated tile)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t r0 = w0 + 64 * ti, c0 = w0 + 64 * tj;
    const int row = tid & 63;
    const int i = lane & 15, kk = lane >> 4;
    double lv[16], zv[16];
    {
        const int64_t r = r0 + row, c = c0 + row;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int m = w + 4 * it;
            lv[it] = (r < rend && m < nbe) ? M.a[r + (k + m) * M.ld] : 0.0;
            zv[it] = (c < rend && m < nbe) ? Zws[(c - w0) + m * ldz] : 0.0;
        }
    }
    // prefetch the C tile entries this lane updates: c = c0+16w+kk+4reg, r = r0+16t+i
    double cv[16];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int64_t c = c0 + 16 * w + kk + 4 * reg, r = r0 + 16 * t + i;
            cv[4 * t + reg] = (r < rend && c < rend && r >= c) ? M.a[r + c * M.ld] : 0.0;
        }
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int m = w + 4 * it;
        Ls[m * LVBA_TS + row] = lv[it];
        Zs[m * LVBA_TS + row] = zv[it];
    }
    __syncthreads();
    d4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int k0 = 0; k0 < 64; k0 += 4) {
        const double a = Zs[(k0 + kk) * LVBA_TS + 16 * w + i];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const double bv = Ls[(k0 + kk) * LVBA_TS + 16 * t + i];
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv, acc[t], 0, 0, 0);
        }
    }
    // acc[t][reg] = sum_m Z[c = 16w+kk+4reg][m] * L[r = 16t+i][m]
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int64_t c = c0 + 16 * w + kk + 4 * reg, r = r0 + 16 * t + i;
            if (r < rend && c < rend && r >= c) M.a[r + c * M.ld] = cv[4 * t + reg] - acc[t][reg];
        }
}
// The same tile with the contributions of TWO consecutive panels (e, then o = e + 1) in one pass: C is read and written once
// per 128 columns instead of once per 64 (rank-128 update; the end phase of the two-ended factorisation is bound by exactly that
// traffic).  (ti, tj) are tile coordinates in panel o's window; panel e's window starts one tile earlier and ends one tile
// earlier, so its rows / columns >= rend_e contribute nothing.  L and Z of panel o are prefetched while panel e's products run.
__device__ __forceinline__ void update_tile2(double *lds, LdltMat M, int64_t k, int nbe, int64_t w0, int64_t rend,
                                             const double *__restrict__ Zws, int64_t ke, int nbe_e, int64_t w0e, int64_t rend_e,
                                             const double *__restrict__ Zwe, int64_t ldz, int64_t ti, int64_t tj)
{
    double *Ls = lds;                // [m][row of tile ti]
    double *Zs = lds + 64 * LVBA_TS; // [m][row of tile tj] (= column of the updated tile)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t r0 = w0 + 64 * ti, c0 = w0 + 64 * tj;
    const int row = tid & 63;
    const int i = lane & 15, kk = lane >> 4;
    const int64_t rr = r0 + row, cc = c0 + row;
    double lv[16], zv[16], lv2[16], zv2[16];
#pragma unroll
    for (int it = 0; it < 16; ++it) { // panel e first
        const int m = w + 4 * it;
        lv[it] = (rr < rend_e && m < nbe_e) ? M.a[rr + (ke + m) * M.ld] : 0.0;
        zv[it] = (cc < rend_e && m < nbe_e) ? Zwe[(cc - w0e) + m * ldz] : 0.0;
    }
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int m = w + 4 * it;
        lv2[it] = (rr < rend && m < nbe) ? M.a[rr + (k + m) * M.ld] : 0.0;
        zv2[it] = (cc < rend && m < nbe) ? Zws[(cc - w0) + m * ldz] : 0.0;
    }
    double cv[16];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int64_t c = c0 + 16 * w + kk + 4 * reg, r = r0 + 16 * t + i;
            cv[4 * t + reg] = (r < rend && c < rend && r >= c) ? M.a[r + c * M.ld] : 0.0;
        }
    d4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        if (pass) __syncthreads(); // everybody is done reading panel e's tiles
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int m = w + 4 * it;
            Ls[m * LVBA_TS + row] = pass ? lv2[it] : lv[it];
            Zs[m * LVBA_TS + row] = pass ? zv2[it] : zv[it];
        }
        __syncthreads();
#pragma unroll 4
        for (int k0 = 0; k0 < 64; k0 += 4) {
            const double a = Zs[(k0 + kk) * LVBA_TS + 16 * w + i];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const double bv = Ls[(k0 + kk) * LVBA_TS + 16 * t + i];
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv, acc[t], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int64_t c = c0 + 16 * w + kk + 4 * reg, r = r0 + 16 * t + i;
            if (r < rend && c < rend && r >= c) M.a[r + c * M.ld] = cv[4 * t + reg] - acc[t][reg];
        }
}
// A quarter of update_tile: the 16 columns [16 qc, 16 qc + 16) of tile (ti, tj), one 16 x 16 output per wavefront.  Used for
// the first tile column, which is a latency chain on ~40 workgroups with the rest of the chip idle: four workgroups per
// tile cut the MFMA part of the chain from 64 to 16 instructions per wavefront.
__device__ __forceinline__ void update_tile_quarter(double *lds, LdltMat M, int64_t k, int nbe, int64_t w0, int64_t rend,
                                                    const double *__restrict__ Zws, int64_t ldz, int64_t ti, int64_t tj, int qc)
{
    double *Ls = lds;                // [m][row of tile ti]
    double *Zs = lds + 64 * LVBA_TS; // [m][16 columns of tile tj]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t r0 = w0 + 64 * ti, c0 = w0 + 64 * tj + 16 * qc;
    const int row = tid & 63;
    const int i = lane & 15, kk = lane >> 4;
    double lv[16], zv[4];
    {
        const int64_t r = r0 + row;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int m = w + 4 * it;
            lv[it] = (r < rend && m < nbe) ? M.a[r + (k + m) * M.ld] : 0.0;
        }
        const int64_t c = c0 + (tid & 15);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int m = (tid >> 4) + 16 * it;
            zv[it] = (c < rend && m < nbe) ? Zws[(c - w0) + m * ldz] : 0.0;
        }
    }
    double cv[4]; // c = c0 + kk + 4 reg, r = r0 + 16 w + i
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int64_t c = c0 + kk + 4 * reg, r = r0 + 16 * w + i;
        cv[reg] = (r < rend && c < rend && r >= c) ? M.a[r + c * M.ld] : 0.0;
    }
#pragma unroll
    for (int it = 0; it < 16; ++it) Ls[(w + 4 * it) * LVBA_TS + row] = lv[it];
#pragma unroll
    for (int it = 0; it < 4; ++it) Zs[((tid >> 4) + 16 * it) * LVBA_TS + (tid & 15)] = zv[it];
    __syncthreads();
    d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int k0 = 0; k0 < 64; k0 += 4) {
        const double a = Zs[(k0 + kk) * LVBA_TS + i];
        const double bv = Ls[(k0 + kk) * LVBA_TS + 16 * w + i];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv, acc, 0, 0, 0);
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int64_t c = c0 + kk + 4 * reg, r = r0 + 16 * w + i;
        if (r < rend && c < rend && r >= c) M.a[r + c * M.ld] = cv[reg] - acc[reg];
    }
}
// mode 0: every lower tile (ti >= tj) of the window; mode 1: only the first tile column (tj = 0), four workgroups per tile
__global__ __launch_bounds__(256) void ldlt_update_kernel(LdltMat M, int64_t k, int nbe, int64_t w0, int64_t rend,
                                                          const double *__restrict__ Zws, int64_t ldz, int mode, int64_t sA,
                                                          int64_t sW)
{
    __shared__ double lds[LVBA_K3_LDS];
    if (blockIdx.y) { M.a += sA; Zws += sW; }
    int64_t ti, tj;
    if (mode == 1 || mode == 2) { // mode 2: the first TWO tile columns (a panel whose bulk update is deferred to its partner's)
        const int64_t T = (rend - w0 + 63) / 64;
        int64_t b = blockIdx.x;
        if (b < 4 * T) update_tile_quarter(lds, M, k, nbe, w0, rend, Zws, ldz, b >> 2, 0, (int)(b & 3));
        else { b -= 4 * T; update_tile_quarter(lds, M, k, nbe, w0, rend, Zws, ldz, 1 + (b >> 2), 1, (int)(b & 3)); }
        return;
    }
    tri_decode(blockIdx.x, ti, tj);
    update_tile(lds, M, k, nbe, w0, rend, Zws, ldz, ti, tj);
}

// ------------------------------------------------------------------------------------ K3, 128 x 64 tiles
// The bulk of the trailing update as 128 x 64 tiles (two tile rows of one tile column), K in chunks of 32 columns, for one
// panel (K = 64) or a pair of panels (K = 128, panel e's columns first).  Measured on the 64 x 64 form (2 problems x 414 tiles
// of a paired update, tools/step_microbench): L / Z operand loads 6 us (L2-bound), C load + store 8 us (HBM-bound), MFMA + LDS
// 15 us -- and 34 us in total, because every workgroup did them one after the other and the two workgroups of a CU in step.  Here
//   * while the products of a chunk run from LDS, the next chunk's operands are on their way into registers, and the C entries
//     are fetched beside the last chunk's products;
//   * a wavefront owns 32 rows x 64 columns: 6 LDS operand reads per 8 MFMAs (5 per 4 before), and the Z rows are fetched once
//     per 128 rows of L;
//   * operands move 16 bytes per lane (two rows of a column), C entries as the MFMA layout has them;
//   * a paired update of both problems is ~410 workgroups: one round of the 2 x 256 slots the factorisation's LDS leaves.
struct PanelRef { int64_t k, w0, rend; int nbe; const double *Z; };
typedef __attribute__((address_space(1))) double gdouble;
typedef __attribute__((address_space(1))) char gchar;
#define LVBA_TL 144 // LDS stride of an L chunk column (128 rows), 16 mod 32 like LVBA_TS
#define LVBA_K3B_LDS (32 * LVBA_TL + 32 * LVBA_TS) // doubles
// column-major enumeration of the 128 x 64 tiles of tile columns [ca, cb) of a Tb x Tb lower triangle: column tj holds the row
// pairs (tj + 2u, tj + 2u + 1), u < (Tb - tj + 1) / 2 (the last pair of a column may be a single tile row)
__host__ __device__ __forceinline__ int64_t pair_col_items(int64_t tj, int64_t Tb) { return (Tb - tj + 1) / 2; }
__device__ __forceinline__ bool pair_decode(int64_t j, int64_t ca, int64_t cb, int64_t Tb, int64_t &R0, int64_t &tj)
{
    for (int64_t c = ca; c < cb; ++c) {
        const int64_t n = pair_col_items(c, Tb);
        if (j < n) { tj = c; R0 = c + 2 * j; return true; }
        j -= n;
    }
    return false;
}
// Buffer addressing (resource + 32-bit lane offset + 32-bit scalar offset): the 60 loads / 32 stores of a tile then need three
// lane-offset registers between them; as flat 64-bit pointers their addresses alone filled > 100 VGPRs and spilled.
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
#define LVBA_BUF_WORD3 0x00020000 // raw buffer, 32-bit data format (gfx90a / gfx94x / gfx950)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_of(const double *p)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(p), 0, 0xFFFFFFF0u, LVBA_BUF_WORD3); // no range to check against
}
__device__ __forceinline__ double buf_ld(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    const v2u a = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return __hiloint2double(a.y, a.x);
}
__device__ __forceinline__ double2 buf_ld2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    const v4u a = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_double2(__hiloint2double(a.y, a.x), __hiloint2double(a.w, a.z));
}
__device__ __forceinline__ void buf_st(__amdgpu_buffer_rsrc_t r, double v, unsigned voff, unsigned soff)
{
    const v2u a = {(unsigned)__double2loint(v), (unsigned)__double2hiint(v)};
    __builtin_amdgcn_raw_buffer_store_b64(a, r, voff, soff, 0);
}
// DB (round 4, the look-ahead launch): TWO chunk buffers in LDS.  Stamps inside the single-buffer form (tools/solver_microbench,
// a tile alone on its CU): 41 k cycles for 16 k of MFMA issue -- 12 k in the four "stage" steps (registers -> LDS between two
// barriers, with the matrix pipe idle), 5.5 k waiting for C and storing it, 3.8 k before the first product.  With two buffers a
// chunk is staged into the OTHER buffer in front of the products of the one before it and there is one barrier per chunk:
//     stage(ch + 1) -> fetch(ch + 3) -> products(ch) -> barrier
// 114 KB of LDS: one workgroup per CU -- which also leaves the chain workgroup of the look-ahead launch alone on its CU.
#define LVBA_K3DB_LDS (2 * LVBA_K3B_LDS) // doubles
template <int nch, bool DB = false> // K chunks of 32: 2 = one panel, 4 = a pair (pe, then po)
__device__ __forceinline__ void bulk_tile_128(double *lds, LdltMat M, const PanelRef po, const PanelRef pe, int64_t ldz64, int64_t R0,
                                              int64_t tj)
{
    // a chunk buffer: Ls[m][row 0..127] at its start, Zs[m][row 0..63] behind it (+ 32 * LVBA_TL), m = column of the chunk
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned w = (unsigned)__builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, kk = lane >> 4;
    const int64_t r0 = po.w0 + 64 * (R0 + 1), c0 = po.w0 + 64 * (tj + 1);
    const bool two = r0 + 64 < po.rend; // the second tile row exists (else wavefronts 2, 3 have nothing to multiply)
    // all offsets below are BYTES in 32 bits: one problem's band storage is < 2^32 bytes (ld * n * 8 = 256 MB at C3)
    const unsigned ld = (unsigned)M.ld, ldz = (unsigned)ldz64;
    const __amdgpu_buffer_rsrc_t rA = buf_of(M.a), rZo = buf_of(po.Z), rZe = buf_of(nch == 4 ? pe.Z : po.Z);
    // operand fetch: L chunk = 128 rows x 32 columns, lane -> rows 2 lane, 2 lane + 1 of column w + 4 it (it < 8);
    //                Z chunk =  64 rows x 32 columns, lane -> rows 2 (lane & 31), + 1 of column 2 (w + 4 it) + (lane >> 5) (it < 4)
    // Two register sets: the chunk being staged and the next one in flight (L: [0..15], Z: [16..23]); set A also takes the C
    // entries ([0..31]) once the last even chunk has left it -- as arrays of their own the compiler gives them registers of
    // their own and spills.
    double xa[32], xb[24];
    const int lrow = 2 * lane, zrow = 2 * (lane & 31), zc = lane >> 5;
    const unsigned lvoff = 8u * (unsigned)lrow, zvoff = 8u * ((unsigned)zrow + (unsigned)zc * ldz);
    auto fetch = [&](int ch, double *xs) {
        const bool use_e = nch == 4 && ch < 2;
        const unsigned qk = (unsigned)(use_e ? pe.k : po.k), zr = (unsigned)(c0 - (use_e ? pe.w0 : po.w0));
        const unsigned m0 = 32u * (unsigned)(ch & 1);
        const unsigned lsoff = 8u * ((unsigned)r0 + (qk + m0 + w) * ld), zsoff = 8u * (zr + (m0 + 2u * w) * ldz);
        // unconditional: what lies outside the window is masked when it is stored to LDS (the band storage's columns overlap
        // their neighbours', and block_system.hip leaves slack behind the last one)
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const double2 v = buf_ld2(rA, lvoff, lsoff + (unsigned)it * (32u * ld));
            xs[2 * it] = v.x; xs[2 * it + 1] = v.y;
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const double2 v = buf_ld2(use_e ? rZe : rZo, zvoff, zsoff + (unsigned)it * (64u * ldz));
            xs[16 + 2 * it] = v.x; xs[17 + 2 * it] = v.y;
        }
    };
    auto stage = [&](int ch, double *xs) { // registers -> LDS, masking rows / columns outside the panel's window
        double *Ls = lds + (DB ? (ch & 1) * LVBA_K3B_LDS : 0), *Zs = Ls + 32 * LVBA_TL;
        const bool use_e = nch == 4 && ch < 2;
        const int64_t qrend = use_e ? pe.rend : po.rend;
        const int qnbe = use_e ? pe.nbe : po.nbe;
        const int m0 = 32 * (ch & 1);
        if (!(r0 + 128 <= qrend && c0 + 64 <= qrend && qnbe == 64)) { // edge tiles only (wave-uniform)
            const bool l0 = r0 + lrow < qrend, l1 = r0 + lrow + 1 < qrend;
            const bool z0 = c0 + zrow < qrend, z1 = c0 + zrow + 1 < qrend;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const bool mok = m0 + (int)w + 4 * it < qnbe;
                xs[2 * it] = (l0 && mok) ? xs[2 * it] : 0.0;
                xs[2 * it + 1] = (l1 && mok) ? xs[2 * it + 1] : 0.0;
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const bool mok = m0 + 2 * ((int)w + 4 * it) + zc < qnbe;
                xs[16 + 2 * it] = (z0 && mok) ? xs[16 + 2 * it] : 0.0;
                xs[17 + 2 * it] = (z1 && mok) ? xs[17 + 2 * it] : 0.0;
            }
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) *reinterpret_cast<double2 *>(Ls + (w + 4 * it) * LVBA_TL + lrow) = make_double2(xs[2 * it], xs[2 * it + 1]);
#pragma unroll
        for (int it = 0; it < 4; ++it) *reinterpret_cast<double2 *>(Zs + (2 * (w + 4 * it) + zc) * LVBA_TS + zrow) = make_double2(xs[16 + 2 * it], xs[17 + 2 * it]);
    };
    // wavefront w: rows 32 w .. 32 w + 31 (two 16-row blocks tl) x 64 columns (four 16-column blocks cq);
    // acc[tl][cq][reg] <-> row r0 + 32 w + 16 tl + i, column c0 + 16 cq + kk + 4 reg
    d4 acc[2][4];
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
        for (int cq = 0; cq < 4; ++cq) acc[tl][cq] = (d4){0.0, 0.0, 0.0, 0.0};
    const bool busy = two || w < 2;
    auto products = [&](int ch) { // the chunk in LDS.  The operands of step k0 + 4 are read before the MFMAs of step k0 are issued: a
                                  // wavefront issues in order, and reads placed after them only start when the matrix pipe is draining
        if (!busy) return;
        const double *Ls = lds + (DB ? (ch & 1) * LVBA_K3B_LDS : 0), *Zs = Ls + 32 * LVBA_TL;
        double a[2][4], bv[2][2];
        auto rd = [&](int k0, int q) {
#pragma unroll
            for (int cq = 0; cq < 4; ++cq) a[q][cq] = Zs[(k0 + kk) * LVBA_TS + 16 * cq + i];
#pragma unroll
            for (int tl = 0; tl < 2; ++tl) bv[q][tl] = Ls[(k0 + kk) * LVBA_TL + 32 * w + 16 * tl + i];
        };
        rd(0, 0);
#pragma unroll
        for (int k0 = 0; k0 < 32; k0 += 4) {
            const int q = (k0 >> 2) & 1;
            if (k0 + 4 < 32) rd(k0 + 4, q ^ 1);
#ifndef LVBA_BULK_NOSCHED
            __builtin_amdgcn_sched_barrier(0); // keep the reads AHEAD of the MFMAs (LLVM's scheduler sinks them to their first use otherwise)
#endif
#pragma unroll
            for (int tl = 0; tl < 2; ++tl)
#pragma unroll
                for (int cq = 0; cq < 4; ++cq) acc[tl][cq] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q][cq], bv[q][tl], acc[tl][cq], 0, 0, 0);
#ifndef LVBA_BULK_NOSCHED
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
    };
    const unsigned cvoff = 8u * ((unsigned)i + (unsigned)kk * ld);                 // lane part of a C entry's offset
    const unsigned csoff = 8u * ((unsigned)r0 + 32u * w + (unsigned)c0 * ld);      // + 128 tl, + 8 (16 cq + 4 reg) ld
    // Chunk c is staged from its register set (even chunks: A, odd: B) and the set is refilled at once with chunk c + 2, which
    // then has the products of two chunks to arrive in.  After the last even chunk, set A takes the C entries instead.
    // (fully unrolled: inside a loop the compiler's wait counts at the back edge drain every load in flight)
    auto load_c = [&]() { // Unmasked: entries outside the window or above the diagonal are read (inside the allocation, see
                          // block_system.hip) but never stored
#pragma unroll
        for (int cq = 0; cq < 4; ++cq)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const unsigned so = csoff + 8u * (unsigned)(16 * cq + 4 * reg) * ld;
#pragma unroll
                for (int tl = 0; tl < 2; ++tl) xa[16 * tl + 4 * cq + reg] = buf_ld(rA, cvoff + 128u * tl, so);
            }
    };
    LVBA_BULK_STAMP(0);
    fetch(0, xa);
    fetch(1, xb);
    if constexpr (DB) {
        stage(0, xa);
        if (nch > 2) fetch(2, xa);
        else if (busy) load_c();
        __syncthreads();
        LVBA_BULK_STAMP(1);
#pragma unroll
        for (int ch = 0; ch < nch; ++ch) {
            double *xs = ((ch + 1) & 1) ? xb : xa; // the register set of chunk ch + 1
            if (ch + 1 < nch) {
                stage(ch + 1, xs);                 // into the other buffer: everybody left it at the last barrier
                if (ch + 3 < nch) fetch(ch + 3, xs);
                else if (ch + 3 == nch && nch > 2 && busy) load_c(); // chunk nch - 2 has just left set A: C takes its place
            }
            products(ch);
            LVBA_BULK_STAMP(2 + ch);
            if (ch + 1 < nch) __syncthreads();
        }
        LVBA_BULK_STAMP(8);
    } else {
#pragma unroll
    for (int ch = 0; ch < nch; ch += 2) {
        if (ch > 0) __syncthreads(); // everybody is done with chunk ch - 1 in LDS
        stage(ch, xa);
        __syncthreads();
        LVBA_BULK_STAMP(1 + 2 * ch);
        if (ch + 2 < nch) fetch(ch + 2, xa);
        else if (busy) load_c();
        products(ch);
        LVBA_BULK_STAMP(2 + 2 * ch);
        __syncthreads();
        stage(ch + 1, xb);
        __syncthreads();
        if (ch + 3 < nch) fetch(ch + 3, xb);
        LVBA_BULK_STAMP(3 + 2 * ch);
        products(ch + 1);
        LVBA_BULK_STAMP(4 + 2 * ch);
    }
    }
    if (busy) {
        const bool inner = r0 + 128 <= po.rend && r0 > c0; // whole tile inside the window and below the diagonal
#pragma unroll
        for (int cq = 0; cq < 4; ++cq)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const unsigned so = csoff + 8u * (unsigned)(16 * cq + 4 * reg) * ld;
                const int64_t c = c0 + 16 * cq + kk + 4 * reg;
#pragma unroll
                for (int tl = 0; tl < 2; ++tl) {
                    const int64_t r = r0 + 32 * w + 16 * tl + i;
                    if (inner || (r < po.rend && c < po.rend && r >= c))
                        buf_st(rA, xa[16 * tl + 4 * cq + reg] - acc[tl][cq][reg], cvoff + 128u * tl, so);
                }
            }
    }
    LVBA_BULK_STAMP(10);
}

// ------------------------------------------------------------------------------------ K3, 128 x 64 tiles, K in chunks of 16
// Round 4.  Stamps inside bulk_tile_128 (tools/solver_microbench; a tile alone on its CU): 41 k cycles for 16 k of MFMA issue --
// 12 k of it in the four "stage" steps (registers -> LDS between two barriers with the matrix pipe idle), 5.5 k waiting for C and
// storing it, 3.8 k before the first product; and the two workgroups of a CU run those phases in step (408 tiles on 2 x 256 seats:
// 30 us against 21 us for one round).  Two chunk buffers of K = 32 take the stage steps off the path (33 k cycles per tile) but
// need 114 KB of LDS, one workgroup per CU, and lose more than they gain (36 us).  This form keeps two workgroups per CU: chunks
// of SIXTEEN columns, two chunk buffers (57 KB), a ring of three register sets -- per chunk
//     stage(ch + 1) into the other buffer -> fetch(ch + 4) into the set just emptied -> products(ch) -> ONE barrier
// so that a chunk has three products' time to arrive and the LDS writes sit in front of 32 MFMAs instead of between barriers.
#define LVBA_KC 16
#define LVBA_K16_BUF (LVBA_KC * LVBA_TL + LVBA_KC * LVBA_TS) // doubles per chunk buffer
#define LVBA_K16_LDS (2 * LVBA_K16_BUF)
template <int npan> // 1: panel o alone (K = 64); 2: panel e, then its partner o (K = 128)
__device__ __forceinline__ void bulk_tile_k16(double *lds, LdltMat M, const PanelRef po, const PanelRef pe, int64_t ldz64, int64_t R0,
                                              int64_t tj)
{
    constexpr int nch = 4 * npan;
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned w = (unsigned)__builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, kk = lane >> 4;
    const int64_t r0 = po.w0 + 64 * (R0 + 1), c0 = po.w0 + 64 * (tj + 1);
    const bool two = r0 + 64 < po.rend; // the second tile row exists (else wavefronts 2, 3 have nothing to multiply)
    const unsigned ld = (unsigned)M.ld, ldz = (unsigned)ldz64; // byte offsets in 32 bits (see bulk_tile_128)
    const __amdgpu_buffer_rsrc_t rA = buf_of(M.a), rZo = buf_of(po.Z), rZe = buf_of(npan == 2 ? pe.Z : po.Z);
    // L chunk = 128 rows x 16 columns: lane -> rows 2 lane, 2 lane + 1 of column w + 4 it (it < 4)
    // Z chunk =  64 rows x 16 columns: lane -> rows 2 (lane & 31), + 1 of column 2 (w + 4 it) + (lane >> 5) (it < 2)
    double xs[3][12]; // three register sets: L [0..7], Z [8..11]
    double cv[32];
    const int lrow = 2 * lane, zrow = 2 * (lane & 31), zc = lane >> 5;
    const unsigned lvoff = 8u * (unsigned)lrow, zvoff = 8u * ((unsigned)zrow + (unsigned)zc * ldz);
    auto fetch = [&](int ch, double *x) {
        const bool use_e = npan == 2 && ch < 4;
        const unsigned qk = (unsigned)(use_e ? pe.k : po.k), zr = (unsigned)(c0 - (use_e ? pe.w0 : po.w0));
        const unsigned m0 = 16u * (unsigned)(ch & 3);
        const unsigned lsoff = 8u * ((unsigned)r0 + (qk + m0 + w) * ld), zsoff = 8u * (zr + (m0 + 2u * w) * ldz);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const double2 v = buf_ld2(rA, lvoff, lsoff + (unsigned)it * (32u * ld));
            x[2 * it] = v.x; x[2 * it + 1] = v.y;
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const double2 v = buf_ld2(use_e ? rZe : rZo, zvoff, zsoff + (unsigned)it * (64u * ldz));
            x[8 + 2 * it] = v.x; x[9 + 2 * it] = v.y;
        }
    };
    auto stage = [&](int ch, double *x) { // registers -> chunk buffer ch & 1, masking rows / columns outside the panel's window
        double *Ls = lds + (ch & 1) * LVBA_K16_BUF, *Zs = Ls + LVBA_KC * LVBA_TL;
        const bool use_e = npan == 2 && ch < 4;
        const int64_t qrend = use_e ? pe.rend : po.rend;
        const int qnbe = use_e ? pe.nbe : po.nbe;
        const int m0 = 16 * (ch & 3);
        if (!(r0 + 128 <= qrend && c0 + 64 <= qrend && qnbe == 64)) { // edge tiles only (wave-uniform)
            const bool l0 = r0 + lrow < qrend, l1 = r0 + lrow + 1 < qrend;
            const bool z0 = c0 + zrow < qrend, z1 = c0 + zrow + 1 < qrend;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const bool mok = m0 + (int)w + 4 * it < qnbe;
                x[2 * it] = (l0 && mok) ? x[2 * it] : 0.0;
                x[2 * it + 1] = (l1 && mok) ? x[2 * it + 1] : 0.0;
            }
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const bool mok = m0 + 2 * ((int)w + 4 * it) + zc < qnbe;
                x[8 + 2 * it] = (z0 && mok) ? x[8 + 2 * it] : 0.0;
                x[9 + 2 * it] = (z1 && mok) ? x[9 + 2 * it] : 0.0;
            }
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) *reinterpret_cast<double2 *>(Ls + (w + 4 * it) * LVBA_TL + lrow) = make_double2(x[2 * it], x[2 * it + 1]);
#pragma unroll
        for (int it = 0; it < 2; ++it) *reinterpret_cast<double2 *>(Zs + (2 * (w + 4 * it) + zc) * LVBA_TS + zrow) = make_double2(x[8 + 2 * it], x[9 + 2 * it]);
    };
    // wavefront w: rows 32 w .. 32 w + 31 (two 16-row blocks tl) x 64 columns (four 16-column blocks cq);
    // acc[tl][cq][reg] <-> row r0 + 32 w + 16 tl + i, column c0 + 16 cq + kk + 4 reg
    d4 acc[2][4];
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
        for (int cq = 0; cq < 4; ++cq) acc[tl][cq] = (d4){0.0, 0.0, 0.0, 0.0};
    const bool busy = two || w < 2;
    auto products = [&](int ch) {
        if (!busy) return;
        const double *Ls = lds + (ch & 1) * LVBA_K16_BUF, *Zs = Ls + LVBA_KC * LVBA_TL;
        double a[2][4], bv[2][2];
        auto rd = [&](int k0, int q) {
#pragma unroll
            for (int cq = 0; cq < 4; ++cq) a[q][cq] = Zs[(k0 + kk) * LVBA_TS + 16 * cq + i];
#pragma unroll
            for (int tl = 0; tl < 2; ++tl) bv[q][tl] = Ls[(k0 + kk) * LVBA_TL + 32 * w + 16 * tl + i];
        };
        rd(0, 0);
#pragma unroll
        for (int k0 = 0; k0 < LVBA_KC; k0 += 4) {
            const int q = (k0 >> 2) & 1;
            if (k0 + 4 < LVBA_KC) rd(k0 + 4, q ^ 1);
            __builtin_amdgcn_sched_barrier(0); // the reads stay AHEAD of the MFMAs
#pragma unroll
            for (int tl = 0; tl < 2; ++tl)
#pragma unroll
                for (int cq = 0; cq < 4; ++cq) acc[tl][cq] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q][cq], bv[q][tl], acc[tl][cq], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    const unsigned cvoff = 8u * ((unsigned)i + (unsigned)kk * ld);                 // lane part of a C entry's offset
    const unsigned csoff = 8u * ((unsigned)r0 + 32u * w + (unsigned)c0 * ld);      // + 128 tl, + 8 (16 cq + 4 reg) ld
    LVBA_BULK_STAMP(0);
    fetch(0, xs[0]);
    fetch(1, xs[1]);
    fetch(2, xs[2]);
    stage(0, xs[0]);
    if (3 < nch) fetch(3, xs[0]);
    __syncthreads();
    LVBA_BULK_STAMP(1);
#pragma unroll
    for (int ch = 0; ch < nch; ++ch) {
        if (ch + 1 < nch) {
            stage(ch + 1, xs[(ch + 1) % 3]); // into the other buffer: everybody left it at the last barrier
            if (ch + 4 < nch) fetch(ch + 4, xs[(ch + 1) % 3]);
        }
        if (ch == (nch > 3 ? nch - 3 : 0) && busy) { // the C entries: three chunks' products ahead of their use.  Unmasked: entries
                                                     // outside the window or above the diagonal are read (inside the allocation,
                                                     // block_system.hip) but never stored
#pragma unroll
            for (int cq = 0; cq < 4; ++cq)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const unsigned so = csoff + 8u * (unsigned)(16 * cq + 4 * reg) * ld;
#pragma unroll
                    for (int tl = 0; tl < 2; ++tl) cv[16 * tl + 4 * cq + reg] = buf_ld(rA, cvoff + 128u * tl, so);
                }
        }
        products(ch);
        if (ch + 1 < nch) __syncthreads();
    }
    LVBA_BULK_STAMP(8);
    if (busy) {
        const bool inner = r0 + 128 <= po.rend && r0 > c0; // whole tile inside the window and below the diagonal
#pragma unroll
        for (int cq = 0; cq < 4; ++cq)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const unsigned so = csoff + 8u * (unsigned)(16 * cq + 4 * reg) * ld;
                const int64_t c = c0 + 16 * cq + kk + 4 * reg;
#pragma unroll
                for (int tl = 0; tl < 2; ++tl) {
                    const int64_t r = r0 + 32 * w + 16 * tl + i;
                    if (inner || (r < po.rend && c < po.rend && r >= c))
                        buf_st(rA, cv[16 * tl + 4 * cq + reg] - acc[tl][cq][reg], cvoff + 128u * tl, so);
                }
            }
    }
    LVBA_BULK_STAMP(10);
}

// ------------------------------------------------------------------------------------ K3, 128 x 128 tiles, operands straight into LDS
// Round 4 (late).  What the three forms above have in common: two workgroups per CU whose eight wavefronts share four matrix pipes
// and run their phases in step, a register file half full of operands in flight, and prologue + epilogue once per 128 x 64 of C
// (408 tiles of a two-problem launch alone: 28.9 - 30.3 us = 52 % of the issue rate).  This form is built the other way round:
//   * 128 x 128 of C per workgroup, 64 x 64 per wavefront (wavefront w: rows 64 (w >> 1), columns 64 (w & 1)): 16 independent MFMAs
//     per step of four columns behind 8 LDS reads, half the operand bytes per flop, half as many tiles -- ONE per CU and launch
//     (204 + 82 role workgroups on 256 CUs), alone on its matrix pipes;
//   * the operand chunks go from global memory STRAIGHT INTO LDS (buffer_load_dwordx4 ... lds: a wavefront's 64 x 16 bytes are one
//     128-row column of a chunk, M0 = where it goes), no register sets, no stage step; a ring of LVBA_SQ_NBUF chunk buffers of
//     LVBA_SQ_KC columns, chunk ch + NBUF - 1 is requested in front of the products of chunk ch, ONE barrier per chunk;
//   * C is loaded into the accumulators (negated; the result is stored negated): no C registers besides them.
// Loads into LDS cannot be masked on the way: an edge tile zeroes what lies outside the panel's window in LDS (one more barrier
// per chunk, edge tiles only).
#ifndef LVBA_SQ_KC
#define LVBA_SQ_KC 16
#endif
#define LVBA_SQ_NBUF (32 / LVBA_SQ_KC)
#define LVBA_SQ_BUF (2 * LVBA_SQ_KC * LVBA_TL)      // doubles: Ls[KC][TL] (rows of the tile), Zs[KC][TL] (its columns)
#define LVBA_SQ_LDS (LVBA_SQ_NBUF * LVBA_SQ_BUF) // 64 * 144 doubles = 73.7 KB
typedef __attribute__((address_space(3))) void lds_void_t;
__device__ __forceinline__ void wait_vmcnt_le(int n) // s_waitcnt vmcnt(n) alone (n a constant after unrolling)
{
#define LVBA_VMCNT(N) __builtin_amdgcn_s_waitcnt(((N) & 15) | (7 << 4) | (15 << 8) | (((N) >> 4) << 14))
    switch (n) {
    case 0: LVBA_VMCNT(0); break;
    case 2: LVBA_VMCNT(2); break;
    case 4: LVBA_VMCNT(4); break;
    case 6: LVBA_VMCNT(6); break;
    case 8: LVBA_VMCNT(8); break;
    case 12: LVBA_VMCNT(12); break;
    case 16: LVBA_VMCNT(16); break;
    case 24: LVBA_VMCNT(24); break;
    default: LVBA_VMCNT(0); break;
    }
#undef LVBA_VMCNT
}
// A barrier that leaves the loads in flight alone (__syncthreads is a fence: it waits for vmcnt(0), i.e. for the chunks just
// requested): this wavefront's LDS reads / writes are complete, then s_barrier.
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0) alone
    __builtin_amdgcn_s_barrier();
}
// tile columns [ca, cb) in pairs (c, c + 1), c = ca, ca + 2, ...; a pair's tiles are the row pairs (c + 2u, c + 2u + 1) as in
// pair_decode; ncol = 1 for the last column of an odd range
__host__ __device__ __forceinline__ int64_t sq_job_items(int64_t ca, int64_t cb, int64_t Tb)
{
    int64_t n = 0;
    for (int64_t c = ca; c < cb; c += 2) n += pair_col_items(c, Tb);
    return n;
}
__device__ __forceinline__ bool sq_decode(int64_t j, int64_t ca, int64_t cb, int64_t Tb, int64_t &R0, int64_t &tj, int &ncol)
{
    for (int64_t c = ca; c < cb; c += 2) {
        const int64_t n = pair_col_items(c, Tb);
        if (j < n) { tj = c; R0 = c + 2 * j; ncol = c + 1 < cb ? 2 : 1; return true; }
        j -= n;
    }
    return false;
}
template <int npan> // 1: panel o alone (K = 64); 2: panel e, then its partner o (K = 128)
__device__ __forceinline__ void bulk_tile_sq(double *lds, LdltMat M, const PanelRef po, const PanelRef pe, int64_t ldz64, int64_t R0,
                                             int64_t tj, int ncol)
{
    constexpr int KC = LVBA_SQ_KC, NBUF = LVBA_SQ_NBUF, CPP = 64 / KC, nch = CPP * npan, LPC = KC / 2; // LPC: loads per chunk and wavefront
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned w = (unsigned)__builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned wr = w >> 1, wc = w & 1;
    const int i = lane & 15, kk = lane >> 4;
    const int64_t r0 = po.w0 + 64 * (R0 + 1), c0 = po.w0 + 64 * (tj + 1);
    const bool two = r0 + 64 < po.rend;
    const bool busy = (wr == 0 || two) && (wc == 0 || ncol == 2) && !(R0 == tj && wr == 0 && wc == 1); // (0, 1) of a diagonal tile lies above the diagonal
    const unsigned ld = (unsigned)M.ld, ldz = (unsigned)ldz64; // byte offsets in 32 bits (see bulk_tile_128)
    const __amdgpu_buffer_rsrc_t rA = buf_of(M.a), rZo = buf_of(po.Z), rZe = buf_of(npan == 2 ? pe.Z : po.Z);
    const unsigned lvoff = 16u * (unsigned)lane; // rows 2 lane, 2 lane + 1 of a chunk column: 16 bytes per lane, 1 KB per wavefront
    auto fetch = [&](int ch) { // chunk ch -> buffer ch % NBUF; wavefront w: columns w + 4 it of the L part and of the Z part
        const bool use_e = npan == 2 && ch < CPP;
        const unsigned qk = (unsigned)(use_e ? pe.k : po.k), zr = (unsigned)(c0 - (use_e ? pe.w0 : po.w0));
        const unsigned m0 = (unsigned)(KC * (ch % CPP));
        double *Ls = lds + (ch % NBUF) * LVBA_SQ_BUF, *Zs = Ls + KC * LVBA_TL;
        const unsigned lsoff = 8u * ((unsigned)r0 + (qk + m0 + w) * ld), zsoff = 8u * (zr + (m0 + w) * ldz);
#pragma unroll
        for (int it = 0; it < KC / 4; ++it)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_void_t *)(Ls + (w + 4 * it) * LVBA_TL), 16, lvoff, lsoff + (unsigned)it * (32u * ld), 0, 0);
#pragma unroll
        for (int it = 0; it < KC / 4; ++it)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(use_e ? rZe : rZo, (lds_void_t *)(Zs + (w + 4 * it) * LVBA_TL), 16, lvoff,
                                                     zsoff + (unsigned)it * (32u * ldz), 0, 0);
    };
    auto mask_edge = [&](int ch) { // (workgroup-uniform) zero what lies outside the panel's window: rows, columns of the chunk
        const bool use_e = npan == 2 && ch < CPP;
        const int64_t qrend = use_e ? pe.rend : po.rend;
        const int qnbe = use_e ? pe.nbe : po.nbe;
        if (r0 + 128 <= qrend && c0 + 128 <= qrend && qnbe == 64) return;
        double *Ls = lds + (ch % NBUF) * LVBA_SQ_BUF, *Zs = Ls + KC * LVBA_TL;
        const int m0 = KC * (ch % CPP);
#pragma unroll
        for (int idx = tid; idx < KC * 128; idx += 256) {
            const int m = idx >> 7, row = idx & 127;
            const bool mok = m0 + m < qnbe;
            if (!(mok && r0 + row < qrend)) Ls[m * LVBA_TL + row] = 0.0;
            if (!(mok && c0 + row < qrend)) Zs[m * LVBA_TL + row] = 0.0;
        }
        lds_barrier();
    };
    // acc[tr][tc][reg] <-> row r0 + 64 wr + 16 tr + i, column c0 + 64 wc + 16 tc + kk + 4 reg
    d4 acc[4][4];
    const unsigned cvoff = 8u * ((unsigned)i + (unsigned)kk * ld);
    const unsigned csoff = 8u * ((unsigned)r0 + 64u * wr + ((unsigned)c0 + 64u * wc) * ld);
    LVBA_BULK_STAMP(0);
    if (busy) { // -C: unmasked -- entries outside the window or above the diagonal are read (inside the allocation) but never stored
#pragma unroll
        for (int tc = 0; tc < 4; ++tc)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const unsigned so = csoff + 8u * (unsigned)(16 * tc + 4 * reg) * ld;
#pragma unroll
                for (int tr = 0; tr < 4; ++tr) acc[tr][tc][reg] = buf_ld(rA, cvoff + 128u * tr, so);
            }
    } else {
#pragma unroll
        for (int tr = 0; tr < 4; ++tr)
#pragma unroll
            for (int tc = 0; tc < 4; ++tc) acc[tr][tc] = (d4){0.0, 0.0, 0.0, 0.0};
    }
#pragma unroll
    for (int ch = 0; ch < NBUF - 1 && ch < nch; ++ch) fetch(ch);
    wait_vmcnt_le(LPC * ((NBUF - 1 < nch ? NBUF - 1 : nch) - 1)); // chunk 0 (and C, requested before it) has arrived
    lds_barrier();
    if (busy) {
#pragma unroll
        for (int tr = 0; tr < 4; ++tr)
#pragma unroll
            for (int tc = 0; tc < 4; ++tc)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) acc[tr][tc][reg] = -acc[tr][tc][reg];
    }
    LVBA_BULK_STAMP(1);
    auto products = [&](int ch) {
        if (!busy) return;
        const double *Ls = lds + (ch % NBUF) * LVBA_SQ_BUF, *Zs = Ls + KC * LVBA_TL;
        double a[2][4], bv[2][4];
        auto rd = [&](int k0, int q) {
#pragma unroll
            for (int tc = 0; tc < 4; ++tc) a[q][tc] = Zs[(k0 + kk) * LVBA_TL + 64 * wc + 16 * tc + i];
#pragma unroll
            for (int tr = 0; tr < 4; ++tr) bv[q][tr] = Ls[(k0 + kk) * LVBA_TL + 64 * wr + 16 * tr + i];
        };
        rd(0, 0);
#pragma unroll
        for (int k0 = 0; k0 < KC; k0 += 4) {
            const int q = (k0 >> 2) & 1;
            if (k0 + 4 < KC) rd(k0 + 4, q ^ 1);
            __builtin_amdgcn_sched_barrier(0); // the reads stay AHEAD of the MFMAs
#pragma unroll
            for (int tr = 0; tr < 4; ++tr)
#pragma unroll
                for (int tc = 0; tc < 4; ++tc) acc[tr][tc] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q][tc], bv[q][tr], acc[tr][tc], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
#pragma unroll
    for (int ch = 0; ch < nch; ++ch) {
        // here: chunk ch is in LDS and visible to everybody; buffer (ch - 1) % NBUF has been left by everybody
        if (ch + NBUF - 1 < nch) fetch(ch + NBUF - 1);
        mask_edge(ch);
        products(ch);
        if (ch < 5) LVBA_BULK_STAMP(2 + ch);
        if (ch + 1 < nch) {
            const int last = ch + NBUF - 1 < nch - 1 ? ch + NBUF - 1 : nch - 1; // last chunk requested so far
            wait_vmcnt_le(LPC * (last - (ch + 1)));                             // chunk ch + 1 has arrived (requests complete in order)
            lds_barrier();
        }
    }
    LVBA_BULK_STAMP(8);
    if (busy) {
        const bool inner = r0 + 128 <= po.rend && r0 >= c0 + 128; // whole tile inside the window and below the diagonal
#pragma unroll
        for (int tc = 0; tc < 4; ++tc)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const unsigned so = csoff + 8u * (unsigned)(16 * tc + 4 * reg) * ld;
                const int64_t c = c0 + 64 * wc + 16 * tc + kk + 4 * reg;
#pragma unroll
                for (int tr = 0; tr < 4; ++tr) {
                    const int64_t r = r0 + 64 * wr + 16 * tr + i;
                    if (inner || (r < po.rend && c < po.rend && r >= c)) buf_st(rA, -acc[tr][tc][reg], cvoff + 128u * tr, so);
                }
            }
    }
    LVBA_BULK_STAMP(10);
}

// One launch for two independent pieces of work: the factorisation of panel p+1 (diag + panel tiles) and the bulk of the
// trailing update of panel p (tiles ti >= tj >= 1).  The former touches block column p+1 only, the latter block columns >= p+2,
// and neither waits for the other inside the launch -- the only ordering is between launches: update(first column of p) -> this
// -> update(first column of p+1).  The 20 us serial factorisation is thereby hidden behind the update instead of preceding it.
// Block order: the factorisation workgroups of ALL problems first (nprob = 2: both ends of a twisted factorisation; a
// blockIdx.y per problem put the second problem's factorisation behind the first problem's ~800 update tiles in dispatch order
// and made the launch ~10 us longer than its critical path), then the update workgroups, alternating between the problems.
// big = true: 128 x 64 update tiles (bulk_tile_128) of the tile columns [ca, cb); false: the 64 x 64 tiles [ca, cb) of the
// column-major enumeration (update_tile / update_tile2; LVBA_BULK=64, A/B).
template <bool big>
__global__ __launch_bounds__(256, 2) void ldlt_step_kernel(LdltMat M, int64_t k2, int nbe2, int64_t w02, int64_t rend2, int T2,
                                                           double *__restrict__ G2, double *__restrict__ dvec,
                                                           double *__restrict__ Zws2, double *__restrict__ b,
                                                           int *__restrict__ status, int64_t k, int nbe, int64_t w0, int64_t rend,
                                                           const double *__restrict__ Zws, int64_t ldz, int64_t sA, int64_t sW,
                                                           int64_t ke, int nbe_e, int64_t w0e, int64_t rend_e,
                                                           const double *__restrict__ Zwe, int64_t ca, int64_t cb, int nprob)
{
    __shared__ double lds[LVBA_K3_LDS];
    static_assert(LVBA_K12_LDS <= LVBA_K3_LDS && LVBA_K3B_LDS <= LVBA_K3_LDS,
                  "factorisation tables / 128 x 64 chunks must fit the update's LDS");
    const int64_t nfac = (int64_t)T2 * nprob;
    int prob;
    int64_t bx;
    if ((int64_t)blockIdx.x < nfac) { prob = (int)(blockIdx.x / T2); bx = blockIdx.x - (int64_t)prob * T2; }
    else {
        const int64_t bb = blockIdx.x - nfac;
        prob = (int)(bb % nprob); bx = bb / nprob;
    }
    if (prob) { M.a += sA; G2 += sW; dvec += sW; Zws2 += sW; b += sW; Zws += sW; if (Zwe) Zwe += sW; }
    if ((int64_t)blockIdx.x < nfac) {
        diagpanel_tile(lds, M, k2, nbe2, w02, rend2, G2, dvec, Zws2, ldz, b, status, bx);
    } else if constexpr (big) {
        int64_t R0, tj;
        if (!pair_decode(bx, ca, cb, (rend - w0 + 63) / 64 - 1, R0, tj)) return;
        const PanelRef po{k, w0, rend, nbe, Zws}, pe{ke, w0e, rend_e, nbe_e, Zwe};
        if (Zwe) bulk_tile_128<4>(lds, M, po, pe, ldz, R0, tj);
        else bulk_tile_128<2>(lds, M, po, pe, ldz, R0, tj);
    } else {
        int64_t ti, tj; // bulk tiles (ti >= tj >= 1 of the window) in column-major order
        col_decode(ca + bx, (rend - w0 + 63) / 64 - 1, ti, tj);
        if (Zwe) update_tile2(lds, M, k, nbe, w0, rend, Zws, ke, nbe_e, w0e, rend_e, Zwe, ldz, ti + 1, tj + 1); // with the partner panel
        else update_tile(lds, M, k, nbe, w0, rend, Zws, ldz, ti + 1, tj + 1);
    }
}

#include "ldlt_lookahead.h"

// ---------------------------------------------------------------------------------------- backward
// x_k = L11^-T (z_k - sum_{i>k} L_ik^T x_i),  z_k = D^-1 L11^-1 b_k = G^T b_k,  L11^-T = G D.
// bacc accumulates sum_{i>k} L_ik^T x_i (right-looking: after x_k is known every column c left of the panel
// inside the band receives A(k:k+64, c)^T x_k).
__global__ __launch_bounds__(256) void ldlt_back_kernel(LdltMat M, int64_t k, int nbe, const double *__restrict__ G,
                                                        const double *__restrict__ dvec, const double *__restrict__ b,
                                                        double *__restrict__ bacc, double *__restrict__ x, int64_t cmin)
{
    constexpr int LS = 65;
    __shared__ double Gs[64 * LS]; // [c][m] = G[m][c]
    __shared__ double bs[64], sd[64], xs[64], red[4 * 64];
    const int tid = threadIdx.x;
    const int i = tid & 63, q = tid >> 6;
    const int64_t c = cmin + 256 * (int64_t)blockIdx.x + tid;
    // this thread's column segment A(k..k+63, c), issued before anything waits (16-byte loads: k, ld are even)
    double colv[64];
    int64_t rmax = k + nbe - 1;
    if (c < k) {
        if (c + M.bw < rmax) rmax = c + M.bw;
        const double2 *col = reinterpret_cast<const double2 *>(M.a + c * M.ld + k);
#pragma unroll
        for (int e = 0; e < 32; ++e) {
            const double2 v = col[e];
            colv[2 * e] = (k + 2 * e <= rmax) ? v.x : 0.0;
            colv[2 * e + 1] = (k + 2 * e + 1 <= rmax) ? v.y : 0.0;
        }
    }
    double gl[16];
#pragma unroll
    for (int it = 0; it < 16; ++it) gl[it] = G[tid + 256 * it];
    if (tid < 64) {
        bs[tid] = (tid < nbe) ? b[k + tid] : 0.0;
        sd[tid] = (tid < nbe) ? bacc[k + tid] : 0.0; // temporarily: the accumulated right-hand side
    }
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int e = tid + 256 * it; // G[m][c']: m = e>>6, c' = e&63
        Gs[(e & 63) * LS + (e >> 6)] = gl[it];
    }
    __syncthreads();
    { // z_i = sum_m G[m][i] b_m : thread (i, q) sums m in [16q, 16q+16)
        double z = 0.0;
#pragma unroll
        for (int m = 0; m < 16; ++m) z += Gs[i * LS + 16 * q + m] * bs[16 * q + m];
        red[q * 64 + i] = z;
    }
    __syncthreads();
    if (tid < 64) {
        const double z = red[tid] + red[64 + tid] + red[128 + tid] + red[192 + tid];
        sd[tid] = (tid < nbe) ? (z - sd[tid]) * dvec[k + tid] : 0.0; // D t
    }
    __syncthreads();
    { // x_i = sum_c G[i][c] (D t)_c : thread (i, q) sums c in [16q, 16q+16)
        double v = 0.0;
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) v += Gs[(16 * q + cc) * LS + i] * sd[16 * q + cc];
        red[q * 64 + i] = v;
    }
    __syncthreads();
    if (tid < 64) {
        const double v = red[tid] + red[64 + tid] + red[128 + tid] + red[192 + tid];
        xs[tid] = v;
        if (blockIdx.x == 0 && tid < nbe) x[k + tid] = v;
    }
    __syncthreads();
    if (c < k) {
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int e = 0; e < 64; e += 2) {
            s0 += colv[e] * xs[e];
            s1 += colv[e + 1] * xs[e + 1];
        }
        bacc[c] += s0 + s1;
    }
}

// Whole backward substitution as ONE launch: workgroup b owns panel j = P-1-b and walks the chain
//   acc_j = sum_{i>j} L_ij^T x_i   (left-looking: its own 64 columns, 16 rows per wave per step, next tile prefetched)
//   x_j   = G D (G^T b_j - acc_j)
// The x vector is the only inter-workgroup channel: it is pre-filled with a NaN sentinel (ldlt_prepare_kernel), written
// with agent-scope atomic stores and polled with agent-scope atomic loads, 8 bytes carrying data and flag at once -- no
// fences, no L2 write-back.  A workgroup waits only for workgroups with a smaller blockIdx (dispatched earlier) or of an
// earlier launch, and a launch holds at most 256 workgroups (one per CU), so the chain cannot starve.  Critical path per panel: poll round trip + one 64x64 tile-vector
// product + two 64x64 mat-vecs out of LDS, ~3 us, against ~9 us for a kernel boundary per panel.
#define LVBA_X_SENTINEL 0x7ff4dead5eed0001ULL
// gridDim.y = 2: two independent chains of the same geometry in one launch -- matrix 1's T part and matrix 2's B part of a
// twisted factorisation both start from x of S (blockIdx.y = 1: matrix 2 at a + sA / workspace + sW, solution vector x2).
__global__ __launch_bounds__(256) void ldlt_back_chain_kernel(LdltMat M, int j_top, const double *__restrict__ Gall,
                                                              const double *__restrict__ dvec, const double *__restrict__ b,
                                                              double *__restrict__ x, int64_t sA, int64_t sW,
                                                              double *__restrict__ x2)
{
    if (blockIdx.y) { M.a += sA; Gall += sW; dvec += sW; b += sW; x = x2; }
    constexpr int LS = 65;
    __shared__ double Gs[64 * LS]; // [c][m] = G[m][c]
    __shared__ double bs[64], sd[64], red[4 * 64];
    const int tid = threadIdx.x, c = tid & 63, q = tid >> 6;
    const int j = j_top - (int)blockIdx.x; // this launch covers the panels j_top, j_top - 1, ...
    const int64_t n = M.n, k = (int64_t)j * 64;
    const int nbe = (int)((n - k) < 64 ? (n - k) : 64);
    const double *G = Gall + (int64_t)j * 4096;
    double gl[16];
#pragma unroll
    for (int it = 0; it < 16; ++it) gl[it] = G[tid + 256 * it];
    if (tid < 64) bs[tid] = (tid < nbe) ? b[k + tid] : 0.0;
    // this lane's column and the rows of it that lie inside the band
    const int64_t col = k + c;
    int64_t rmaxc = col + M.bw;
    if (rmaxc > n - 1) rmaxc = n - 1;
    if (c >= nbe) rmaxc = -1;
    const double *colp = M.a + col * M.ld;
    int64_t rl = k + 63 + M.bw;
    if (rl > n - 1) rl = n - 1;
    const int ihi = (int)(rl >> 6);
    auto load_tile = [&](int i, double(&t)[16]) {
        const int64_t r0 = (int64_t)i * 64 + 16 * q;
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
            double v0 = 0.0, v1 = 0.0;
            if (r0 + e + 1 <= rmaxc) {
                const double2 v = *reinterpret_cast<const double2 *>(colp + r0 + e);
                v0 = v.x; v1 = v.y;
            } else if (r0 + e <= rmaxc) {
                v0 = colp[r0 + e];
            }
            t[e] = v0; t[e + 1] = v1;
        }
    };
    double t[16], tn[16];
    if (ihi > j) load_tile(ihi, t);
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int e = tid + 256 * it; // G[m][c']: m = e>>6, c' = e&63
        Gs[(e & 63) * LS + (e >> 6)] = gl[it];
    }
    __syncthreads();
    { // z_c = sum_m G[m][c] b_m : thread (c, q) sums m in [16q, 16q+16)
        double z = 0.0;
#pragma unroll
        for (int m = 0; m < 16; ++m) z += Gs[c * LS + 16 * q + m] * bs[16 * q + m];
        red[q * 64 + c] = z;
    }
    __syncthreads();
    double zc = 0.0;
    if (tid < 64) zc = red[tid] + red[64 + tid] + red[128 + tid] + red[192 + tid];
    __syncthreads(); // red is reused below
    double acc = 0.0;
    for (int i = ihi; i > j; --i) {
        if (i - 1 > j) load_tile(i - 1, tn);
        const int64_t r = (int64_t)i * 64 + 16 * q + (c & 15);
        double xv = 0.0;
        if (r < n) {
            const unsigned long long *px = reinterpret_cast<const unsigned long long *>(x + r);
            unsigned long long v;
            while ((v = __hip_atomic_load(px, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == LVBA_X_SENTINEL)
                __builtin_amdgcn_s_sleep(1);
            xv = __longlong_as_double((long long)v);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) acc = fma(t[e], readlane_f64(xv, e), acc);
#pragma unroll
        for (int e = 0; e < 16; ++e) t[e] = tn[e];
    }
    red[q * 64 + c] = acc;
    __syncthreads();
    if (tid < 64) {
        const double a = red[tid] + red[64 + tid] + red[128 + tid] + red[192 + tid];
        sd[tid] = (tid < nbe) ? (zc - a) * dvec[k + tid] : 0.0; // D t
    }
    __syncthreads();
    { // x_i = sum_c G[i][c] (D t)_c : thread (i, q) sums c in [16q, 16q+16)
        double v = 0.0;
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) v += Gs[(16 * q + cc) * LS + c] * sd[16 * q + cc];
        red[q * 64 + c] = v;
    }
    __syncthreads();
    if (tid < nbe) {
        const double v = red[tid] + red[64 + tid] + red[128 + tid] + red[192 + tid];
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(x + k + tid), (unsigned long long)__double_as_longlong(v),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

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
int64_t ldlt_twist_panels(int64_t n, int64_t ld, int64_t bw)
{
    static const bool off = [] { const char *e = getenv("LVBA_TWIST"); return e && !strcmp(e, "0"); }();
    if (off || ld == n) return 0; // dense storage: every column is coupled to every other
    const int64_t P = (n - bw) / (2 * LVBA_NB);
    return P >= 4 ? P : 0;
}

// Launch sequence of one solve on one stream (captured once into a hipGraph by the caller).
//   LVBA_SCHEDULE = overlap (default): per panel  [factorise panel p+1 || bulk update of panel p]  ->  first-column update of p+1
//                   serial            : factorise -> update, one after the other (A/B reference).
// A two-STREAM look-ahead was measured slower than the serial sequence (every cross-stream edge costs ~10 us); the
// overlap form gets the same concurrency from one heterogeneous launch.
// Band systems are factorised from BOTH ENDS at once (LdltTwist above; LVBA_TWIST=0 turns it off): the serial chain of
// panels -- the latency that bounds this solver -- is P + |S| / 64 long instead of n / 64, with the same flops and no fill.
// With `dist` (>= 2 ranks) the two ends run on two GPUs: rank 0 eliminates T, rank 1 eliminates B (as the second problem,
// alone in its launches), the S block + right-hand side are all-reduced (ranks >= 2 contribute zeros), every rank factorises S,
// rank 0 back-substitutes T and rank 1 B, and the solution is all-reduced.  Per rank the end phase is as long as the
// single-GPU one but moves one window per launch instead of two; what is replicated is the S phase only.
int32_t ldlt_solve(const LdltMat &A, const double *Hblk, int band_blocks, int n_poses, const double *g,
                   const double *u_dev, double *x, double *work, int *status, hipStream_t s, const LdltDist *dist, const int32_t *grp)
{
    static const bool overlap = [] { const char *e = getenv("LVBA_SCHEDULE"); return !(e && !strcmp(e, "serial")); }();
    const int64_t n = A.n, bw = A.bw;
    const int64_t P1 = overlap ? ldlt_twist_panels(n, A.ld, bw) : 0;
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
    LdltMat M = A; // the problem the launches see: matrix 1 (and matrix 2 through blockIdx.y)
    M.n = nf;
    const size_t abytes = (size_t)((A.ld == n) ? n * n : (A.ld + 1) * n) * sizeof(double);
    const bool fill = A.ld != n && n == 6 * (int64_t)n_poses; // band storage: destination-major fill, no memset
    // grid y of the fill: block offsets d0 = 28 y must cover every stored offset d in [0, ldab) of a column, for every element
    // row / column e in [0, 6): d = 6 d0 + t - e (matrix 1) or 6 d0 + t + e - 5 (matrix 2), t in [0, 168) -- i.e. up to ldab + 5
    static_assert(LVBA_PB_ROWS == 6 * LVBA_PB_BLOCKS, "a workgroup of the band fill covers LVBA_PB_BLOCKS block offsets");
    if (fill) {
        static const bool check_band = [] { const char *e = getenv("LVBA_CHECK_BAND"); return e && !strcmp(e, "1"); }();
        static const bool band_memset = [] { const char *e = getenv("LVBA_BAND_MEMSET"); return e && !strcmp(e, "1"); }();
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
        if (band_memset) hipMemsetAsync(A.a, 0, (size_t)total * sizeof(double), s); // A/B: the former per-solve memset
    }
    if (fill)
    {
        const int64_t cols1 = (tw.n1 + 5) / 6, rows2 = P1 > 0 ? n_poses - tw.m / 6 : 0;
        hipLaunchKernelGGL(ldlt_prepare_band_kernel, dim3((unsigned)std::max(cols1, rows2), (unsigned)((A.ld + 1 + 5 + LVBA_PB_ROWS - 1) / LVBA_PB_ROWS), P1 > 0 ? 2 : 1),
                           dim3(256), 0, s, A, Hblk, band_blocks, n_poses, u_dev, tw, grp);
    }
    else
        hipMemsetAsync(A.a, 0, P1 > 0 ? (size_t)tw.sA * sizeof(double) + abytes : abytes, s);
    static const bool back_panel_env = [] { const char *e = getenv("LVBA_BACK"); return e && !strcmp(e, "panel"); }();
    if (back_panel_env && P1 == 0) hipMemsetAsync(bacc, 0, (size_t)n * sizeof(double), s); // only ldlt_back_kernel accumulates there
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
    // ny = 2: both problems in one launch (blockIdx.y); ny = 1 with second = true: the second problem alone (its pointers are
    // passed as the launch's base pointers) -- what rank 1 of a multi-rank job runs
    LdltMat M2 = M;
    M2.a += tw.sA;
    auto factor_panel = [&](int64_t st, const Geo &q, unsigned ny, bool second = false) { // diag (+ panel) of one panel
        const int64_t wo = second ? tw.sW : 0;
        double *G = Gall + wo + st * 4096, *Zws = Zbuf[st % 4] + wo;
        if (q.T > 0)
            hipLaunchKernelGGL(ldlt_diagpanel_kernel, dim3((unsigned)q.T, ny), dim3(256), 0, s, second ? M2 : M, q.k, q.nbe, q.w0, q.rend, G,
                               dvec + wo, Zws, ldz, b + wo, status, tw.sA, tw.sW);
        else
            hipLaunchKernelGGL(ldlt_diag_blocked_kernel, dim3(1), dim3(256), 0, s, second ? M2 : M, q.k, q.nbe, G, dvec + wo, status,
                               (int64_t)0, (int64_t)0, (double *)nullptr, (int64_t)0);
    };
    // ncols = 2: a panel whose bulk update is deferred to its partner's launch applies itself to the first TWO tile columns
    auto first_column = [&](int64_t st, const Geo &q, unsigned ny, bool second, int ncols) {
        const int64_t wo = second ? tw.sW : 0;
        if (q.T > 0)
            hipLaunchKernelGGL(ldlt_update_kernel, dim3((unsigned)(ncols == 2 ? 4 * q.T + 4 * (q.T - 1) : 4 * q.T), ny), dim3(256), 0, s,
                               second ? M2 : M, q.k, q.nbe, q.w0, q.rend, Zbuf[st % 4] + wo, ldz, ncols == 2 ? 2 : 1, tw.sA, tw.sW);
    };
    // One launch: [factorise panel st+1 (fac) || bulk tiles [t0, t1) of panel sb's trailing update], the tiles in column-major
    // order.  qe != NULL: panel sb together with its partner sb-1 (rank 128: one pass over C for both).
    auto step = [&](int64_t st, const Geo &q2, bool fac, unsigned ny, bool second, int64_t sb_, const Geo *qb, const Geo *qe,
                    int64_t t0, int64_t t1) {
        const int64_t wo = second ? tw.sW : 0;
        const int64_t nb3 = qb ? t1 - t0 : 0;
        const int64_t T2 = fac ? q2.T : 0;
        const Geo z{0, 0, 0, 0, 0};
        const Geo &B = qb ? *qb : z;
        // LVBA_BULK = 128 (default: 128 x 64 update tiles) | 64 (one 64 x 64 tile per workgroup; A/B)
        static const bool big_env = [] { const char *e = getenv("LVBA_BULK"); return !(e && !strcmp(e, "64")); }();
        // the 128 x 64 tiles address one problem's storage with 32-bit byte offsets (buffer instructions): a matrix of 4 GB or
        // more (dense n >= 23 170, or a very long band) keeps the 64 x 64 tiles and their 64-bit pointers
        const bool big = big_env && ((uint64_t)A.ld * (uint64_t)(n + 128) + (uint64_t)n + 256) * 8 < 0xFFFF0000ull;
        int64_t ca = t0, cb = t1, nbu = nb3; // 64 x 64: the tile range itself
        if (big && nb3 > 0) { // t0, t1 are column starts: tile columns [ca, cb), 128 x 64 tiles
            const int64_t Tb = B.T - 1;
            ca = 0; cb = Tb;
            while (ca < Tb && col_start(ca, Tb) < t0) ++ca;
            while (cb > ca && col_start(cb, Tb) > t1) --cb;
            nbu = 0;
            for (int64_t c = ca; c < cb; ++c) nbu += pair_col_items(c, Tb);
        }
        if (T2 + nbu > 0)
            hipLaunchKernelGGL(big ? ldlt_step_kernel<true> : ldlt_step_kernel<false>, dim3((unsigned)((T2 + nbu) * ny)), dim3(256), 0, s, second ? M2 : M, q2.k, q2.nbe, q2.w0, q2.rend,
                               (int)T2, Gall + wo + (st + 1) * 4096, dvec + wo, Zbuf[(st + 1) % 4] + wo, b + wo, status, B.k, B.nbe, B.w0,
                               B.rend, (const double *)(Zbuf[(sb_ % 4 + 4) % 4] + wo), ldz, tw.sA, tw.sW, qe ? qe->k : 0, qe ? qe->nbe : 0,
                               qe ? qe->w0 : 0, qe ? qe->rend : 0,
                               qe ? (const double *)(Zbuf[((sb_ - 1) % 4 + 4) % 4] + wo) : (const double *)nullptr, ca, cb, (int)ny);
    };
    // LVBA_RANK128=0: every panel applies its own bulk update (A/B)
    static const bool rank128 = [] { const char *e = getenv("LVBA_RANK128"); return !(e && !strcmp(e, "0")); }();
    // Panels [sa, sb) of one problem (or of both, ny = 2).  Consecutive panels are PAIRED (e, o = e + 1): e applies itself to
    // the two tile columns the next two factorisations need (first_column, ncols = 2) and leaves the rest of its trailing update
    // to its partner's, where every C tile is read and written once for both.  That rank-128 update is spread over the two
    // launches that follow o's factorisation -- the tile columns in order, the first half beside the factorisation of e + 2,
    // the second beside that of e + 3 --, so every launch carries about the same share.  close: also finish the last panel's
    // update (end phase).
    auto run_phase = [&](int64_t sa, int64_t sb, unsigned ny, bool second, bool close) {
        auto pe = [&](int64_t st) { return rank128 && ((st - sa) % 2 == 0) && st + 1 < sb && geom(st).T >= 3 && geom(st + 1).T >= 2; };
        auto po = [&](int64_t st) { return st > sa && pe(st - 1); };
        struct Pending { bool on; int64_t o; Geo qo, qe; int64_t t0, t1; } pend{false, 0, {}, {}, 0, 0};
        auto flush = [&]() { // the deferred second half as a launch of its own
            if (pend.on) step(pend.o, pend.qo, false, ny, second, pend.o, &pend.qo, &pend.qe, pend.t0, pend.t1);
            pend.on = false;
        };
        auto launch = [&](int64_t st, const Geo &q, const Geo &q2, bool fac, bool all) {
            const int64_t Tb = q.T - 1, total = Tb > 0 ? Tb * (Tb + 1) / 2 : 0;
            if (pe(st)) { // its own update waits for the partner; this slot carries the previous pair's second half
                if (pend.on) step(st, q2, fac, ny, second, pend.o, &pend.qo, &pend.qe, pend.t0, pend.t1);
                else step(st, q2, fac, ny, second, st, nullptr, nullptr, 0, 0);
                pend.on = false;
                return;
            }
            flush();
            if (po(st)) {
                const Geo qe = geom(st - 1);
                int64_t cs = 1; // first tile column of the second half: about half of the tiles each
                while (cs < Tb && 2 * col_start(cs, Tb) < total) ++cs;
                const int64_t t_half = all ? total : col_start(cs, Tb);
                step(st, q2, fac, ny, second, st, &q, &qe, 0, t_half);
                if (t_half < total) pend = Pending{true, st, q, qe, t_half, total};
            } else
                step(st, q2, fac, ny, second, st, total > 0 ? &q : nullptr, nullptr, 0, total);
        };
        // A panel's first tile column(s) are a launch of their own between two factorisations.  (Folding them into the
        // factorisation of the next panel -- two 64 x 64 x 64 products in front of it -- was measured in round 2: the products
        // and their LDS round trips cost as much as the launch they replace.)
        Geo q = geom(sa);
        factor_panel(sa, q, ny, second);
        first_column(sa, q, ny, second, pe(sa) ? 2 : 1);
        for (int64_t st = sa; st + 1 < sb; ++st) {
            const Geo q2 = geom(st + 1);
            if (q2.T > 0) {
                launch(st, q, q2, true, false);
                first_column(st + 1, q2, ny, second, pe(st + 1) ? 2 : 1);
            } else { // the last panel has no rows below it: nothing to overlap with
                launch(st, q, q2, false, true);
                flush();
                factor_panel(st + 1, q2, ny, second);
            }
            q = q2;
        }
        if (close) {
            launch(sb - 1, q, geom(sb), false, true);
        }
        flush();
    };
    // ---- the look-ahead schedule (default; LVBA_SOLVER=r3 keeps the two-launches-per-panel form above for A/B): one launch per
    // panel, ldlt_lookahead.h; which launch carries which bulk job is decided by ldlt_schedule.h (checked on the CPU against a
    // tile-level model of the factorisation, tests/ldlt_schedule_check.cpp)
    static const bool lookahead = [] { const char *e = getenv("LVBA_SOLVER"); return !(e && !strcmp(e, "r3")); }();
    double *side_buf[2] = {Zbuf[3] + ldz * LVBA_NB + 64, Zbuf[3] + ldz * LVBA_NB + 64 + 4096};
    double *dq_buf[2] = {side_buf[1] + 4096, side_buf[1] + 2 * 4096};
    // Panel q's share of the next diagonal block, L(p+1, q) Z(p+1, q)^T, is formed by row 1 of launch X_q (which holds L(p+1, q))
    // and handed to X_p's chain workgroup ready-made: the chain role alone 26.1 -> 23.2 us.  Row 1 then does one product more
    // than the other rows; in the launches where the rows also carry the q_extra product its q_extra tile goes to a role
    // workgroup of its own (qx_helper), and the seat next to row 1 stays empty like the chain's (LVBA_ROW1_ALONE).  rocprof,
    // S phase: launches 26.5 -> 23 us; C3 solve -0.08 .. -0.13 ms (ABAB on one box: 4.18, 4.12 against 4.05, 4.04).  Without
    // the helper and the empty seat the S phase alternates 27 / 22.5 us and the two-ended launches grow: no gain at all (modes 1, 2).
    // LVBA_CHAIN_DQ=0: the chain workgroup multiplies itself (A/B).
    static const int chain_dq = [] { const char *e = getenv("LVBA_CHAIN_DQ"); return e ? atoi(e) : 3; }(); // 0: off, 1: always, 2: only from launches without q_extra, 3 (default): always + qx_helper
    // LVBA_BULK_TILE = k32 (default: round 2's tile, K chunks of 32, one chunk buffer) | k16 (chunks of 16, two buffers, three
    // register sets: 28.9 against 30.3 us for 408 tiles alone, but the same solve time, 4.10 / 4.06 ms) | k32db (two K = 32
    // buffers, one workgroup per CU: 4.87 ms)
    static const int bulk_tile = [] {
        const char *e = getenv("LVBA_BULK_TILE");
        return !e ? 0 : !strcmp(e, "k16") ? 2 : !strcmp(e, "k32db") ? 1 : !strcmp(e, "sq") ? 3 : 0;
    }();
    static const bool chain_alone = [] { const char *e = getenv("LVBA_CHAIN_ALONE"); return !(e && !strcmp(e, "0")); }();
    static const int bulk_prio = [] { const char *e = getenv("LVBA_BULK_PRIO"); return e ? atoi(e) : 0; }();
    static const int row_prio = [] { const char *e = getenv("LVBA_ROW_PRIO"); return e ? atoi(e) : 0; }();
    static const bool row1_alone = [] { const char *e = getenv("LVBA_ROW1_ALONE"); return !(e && !strcmp(e, "0")); }();
    static const int n_cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
        (void)hipGetLastError();
        return n > 0 ? n : 256;
    }();
    auto run_phase2 = [&](int64_t sa, int64_t sb, unsigned ny, bool second, bool close) {
        static const bool big_env = [] { const char *e = getenv("LVBA_BULK"); return !(e && !strcmp(e, "64")); }();
        const bool big = big_env && ((uint64_t)A.ld * (uint64_t)(n + 128) + (uint64_t)n + 256) * 8 < 0xFFFF0000ull;
        const int64_t wo = second ? tw.sW : 0;
        std::vector<SchedLaunch> sched;
        ldlt_schedule_phase(sa, sb, close, rank128, [&](int64_t st) { return st < nsteps ? geom(st).T : (int64_t)0; }, sched);
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
            a.skip_a = a.skip_b = -1;
            a.M = second ? M2 : M; a.sA = tw.sA; a.sW = tw.sW; a.ldz = ldz; a.nprob = (int)ny;
            a.roles = L.roles; a.has_q = L.has_q; a.do_diag = L.do_diag; a.q_extra = L.q_extra; a.status = status;
            a.dvec = dvec + wo; a.b = b + wo;
            int64_t nwg = 0;
            if (L.roles) {
                a.p = pg(L.p);
                if (L.has_q) a.q = pg(L.p - 1);
                const Geo gn = geom(L.p + 1);
                a.nbe_next = gn.nbe; a.rend_next = gn.rend;
                a.Gp = Gall + wo + L.p * 4096; a.Gn = Gall + wo + (L.p + 1) * 4096;
                a.Zp = Zbuf[L.p % 4] + wo; a.Zq = L.has_q ? Zbuf[(L.p - 1) % 4] + wo : nullptr;
                a.side_r = side_buf[L.p % 2] + wo; a.side_w = side_buf[(L.p + 1) % 2] + wo;
                if (chain_dq) { // written by row 1 of this launch for the next one; read by the chain if the launch before had a row 1
                    const bool wr = chain_dq == 1 || chain_dq == 3 || !L.q_extra;
                    a.dq_w = wr ? dq_buf[L.p % 2] + wo : nullptr;
                    a.dq_r = (L.has_q && geom(L.p - 1).T >= 2 && dq_prev_written) ? dq_buf[(L.p - 1) % 2] + wo : nullptr;
                    dq_prev_written = wr;
                    a.qx_helper = (wr && L.q_extra && a.p.T >= 2 && chain_dq == 3) ? 1 : 0; // (3: + row 1's q_extra tile on a workgroup of its own)
                }
                nwg += a.p.T + a.qx_helper;
            }
            for (int j = 0; j < L.njobs; ++j) {
                BulkJob &J = a.job[a.njobs];
                const SchedJob &sj = L.job[j];
                J.o = pg(sj.o); J.Zo = Zbuf[sj.o % 4] + wo; J.pair = sj.pair;
                if (sj.pair) { J.e = pg(sj.o - 1); J.Ze = Zbuf[(sj.o - 1) % 4] + wo; }
                const int64_t Tb = J.o.T - 1;
                if (big && bulk_tile == 3) {
                    J.ca = sj.ca; J.cb = sj.cb; J.nwg = sq_job_items(sj.ca, sj.cb, Tb);
                } else if (big) {
                    J.ca = sj.ca; J.cb = sj.cb; J.nwg = 0;
                    for (int64_t c = sj.ca; c < sj.cb; ++c) J.nwg += pair_col_items(c, Tb);
                } else {
                    J.ca = col_start(sj.ca, Tb); J.cb = col_start(sj.cb, Tb); J.nwg = J.cb - J.ca;
                }
                if (J.nwg <= 0) continue;
                nwg += J.nwg;
                ++a.njobs;
            }
            a.bulk_prio = bulk_prio; a.row_prio = row_prio;
            int64_t grid = nwg * ny;
            const int bt = big ? bulk_tile : 0; // (ldlt_lookahead.h: BT)
            if (chain_alone && bt != 1 && L.roles && grid > n_cus) { // (ldlt_lookahead.h: resv_at; bt 1 has one workgroup per CU anyway)
                // (row 1 -- the blocks ny .. 2 ny - 1 -- does one product more than the other rows when it forms panel p's share
                // of the next diagonal block: LVBA_ROW1_ALONE=1 keeps the seats next to it empty as well)
                a.resv_at = n_cus; a.resv_n = (int)ny * (row1_alone && a.p.T >= 2 ? 2 : 1);
                grid += a.resv_n;
            }
            if (nwg > 0) {
                if (bt == 3) hipLaunchKernelGGL((ldlt_step2_kernel<true, 3>), dim3((unsigned)grid), dim3(256), 0, s, a);
                else if (bt == 2) hipLaunchKernelGGL((ldlt_step2_kernel<true, 2>), dim3((unsigned)grid), dim3(256), 0, s, a);
                else if (bt == 1) hipLaunchKernelGGL((ldlt_step2_kernel<true, 1>), dim3((unsigned)grid), dim3(256), 0, s, a);
                else if (big) hipLaunchKernelGGL((ldlt_step2_kernel<true, 0>), dim3((unsigned)grid), dim3(256), 0, s, a);
                else hipLaunchKernelGGL((ldlt_step2_kernel<false, 0>), dim3((unsigned)grid), dim3(256), 0, s, a);
            }
        }
    };
    if (overlap && lookahead) {
        int64_t st0 = 0;
        if (P1 > 0) {
            if (side != 2) run_phase2(0, P1, side < 0 ? 2 : 1, side == 1, true);
            if (side < 0) {
                hipLaunchKernelGGL(ldlt_twist_merge_kernel, dim3(1024), dim3(256), 0, s, A, tw, b);
            } else {
                const int64_t ne = (tw.n1 - tw.m) * (bw + 2);
                hipLaunchKernelGGL(ldlt_twist_pack_kernel, dim3(1024), dim3(256), 0, s, A, tw, (const double *)b, side, Ebuf);
                if (dist->allreduce_sum(dist->ctx, Ebuf, (size_t)ne)) return LVBA_ERR_DIST;
                hipLaunchKernelGGL(ldlt_twist_unpack_kernel, dim3(1024), dim3(256), 0, s, A, tw, b, (const double *)Ebuf);
            }
            st0 = P1;
        }
        run_phase2(st0, nsteps, 1, false, false);
    } else if (!overlap) {
        for (int64_t st = 0; st < nsteps; ++st) {
            const Geo q = geom(st);
            factor_panel(st, q, 1);
            if (q.T > 0)
                hipLaunchKernelGGL(ldlt_update_kernel, dim3((unsigned)(q.T * (q.T + 1) / 2)), dim3(256), 0, s, M, q.k, q.nbe, q.w0, q.rend, Zbuf[st % 4], ldz, 0, tw.sA, tw.sW);
        }
    } else {
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
    }
    // backward.  Default: the whole substitution as one chained launch (ldlt_back_chain_kernel).  LVBA_BACK=panel keeps
    // the former one-launch-per-panel form for A/B (1.7 ms of a C3 solve, ~9 us per kernel boundary).
    static const bool back_panel = [] { const char *e = getenv("LVBA_BACK"); return e && !strcmp(e, "panel"); }();
    if (!back_panel || P1 > 0) {
        // at most 256 panels per launch: one workgroup per CU is then resident whatever else shares the device, so the
        // chain cannot starve even if workgroups were not dispatched in index order; later launches only read finished x
        // (a multi-rank job stops matrix 1's chain after the S panels unless this rank owns T)
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
    for (int64_t st = nsteps - 1; st >= 0; --st) {
        const int64_t k = st * LVBA_NB;
        const int nbe = (int)((n - k) < LVBA_NB ? (n - k) : LVBA_NB);
        int64_t cmin = k - bw;
        if (cmin < 0) cmin = 0;
        const int64_t ncols = k - cmin;
        const unsigned nwg = (unsigned)(ncols > 0 ? (ncols + 255) / 256 : 1);
        hipLaunchKernelGGL(ldlt_back_kernel, dim3(nwg), dim3(256), 0, s, A, k, nbe, Gall + st * 4096, dvec, b, bacc, x, cmin);
    }
    return LVBA_OK;
}

} // namespace lvba
