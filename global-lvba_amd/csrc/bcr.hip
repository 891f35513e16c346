// bcr.hip -- block cyclic reduction for NARROW-band symmetric positive definite systems: the reduced camera system of the
// visual stage (replaces the dense Cholesky inside Ceres' DENSE_SCHUR, reference src/lvba_system.cpp:1574,1643).
//
// With feature tracks that span a few consecutive cameras the reduced camera system S (6M x 6M, S = sum_cams Jc^T Jc + D^2
// - sum_landmarks Y Y^T) couples camera I only to I +- Bb, Bb = 3 at the BASELINE.json sizes: a band of 23 scalars in
// n = 12 000.  The blocked band LDL^T of ldlt.hip spends its time in the SERIAL chain of n / 64 panel factorisations there
// (188 x ~24 us; 95 with the two-ended form) for 0.006 GFLOP of arithmetic.  Grouping k >= Bb cameras into one block row makes
// S block-tridiagonal, and cyclic reduction eliminates every second block row of the remaining ones per level, all of them
// independently: ceil(log2(M / k)) levels of small dense work instead of n / 64 dependent steps.
//
//   level l, stride s = 2^l, active rows = multiples of s; odd ones i = (2m+1) s are eliminated:
//     A  (one workgroup per odd row)   Dinv_i by Gauss-Jordan in registers (SPD: no pivoting; a pivot <= 0 raises `status`),
//                                      T1_i = Dinv_i L_i, T2_i = Dinv_i L_{i+s}^T, t_i = Dinv_i rhs_i      (L_r = S[r, r - s])
//     B  (one workgroup per even row)  D_r  -= L_r T2_{r-s} + L_{r+s}^T T1_{r+s}
//                                      rhs_r -= L_r t_{r-s}  + L_{r+s}^T t_{r+s}
//                                      L_r  <- -L_r T1_{r-s}                          (its coupling to r - 2s)
//   last: x_0 = D_0^-1 rhs_0; back, level by level: x_i = t_i - T1_i x_{i-s} - T2_i x_{i+s}.
// Block rows are padded to BP = 32 or 64 scalars with identity rows, so k <= 5 or k <= 10 cameras per block; wider bands stay
// with ldlt.hip.  Everything is deterministic (no atomics); every arithmetic operation runs here, on the device.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <utility>
#include "lvba_internal.h"

namespace lvba {

namespace {

#define BCR_X_SENTINEL 0x7ff4dead5eed0002ULL // a NaN payload no arithmetic produces

struct BcrDev {
    int nb, k, M, Bb;      // block rows, cameras per block row, cameras, camera half-bandwidth
    double *D, *L, *T1, *T2; // [nb][BP*BP] row-major
    double *L2;              // second coupling-block array of the one-launch levels (bcr_level_kernel)
    double *rhs, *t, *x;   // [nb][BP]
};

template <int BP>
__device__ __forceinline__ void load_tile(double *lds, const double *__restrict__ g, bool present)
{
    constexpr int LD = BP + 1;
    for (int e = threadIdx.x; e < BP * BP; e += 256) lds[(e / BP) * LD + (e % BP)] = present ? g[e] : 0.0;
}

// C (registers of the calling threads: entry idx = tid + 256 j -> (idx / BP, idx % BP)) = sum_m opA(r, m) * opB(m, c) out of LDS
template <int BP, bool TA, bool TB>
__device__ __forceinline__ void tile_mul(const double *A, const double *B, double (&C)[BP * BP / 256])
{
    constexpr int LD = BP + 1;
#pragma unroll
    for (int j = 0; j < BP * BP / 256; ++j) {
        const int idx = threadIdx.x + 256 * j, r = idx / BP, c = idx % BP;
        double s = 0.0;
#pragma unroll 8
        for (int m = 0; m < BP; ++m) s += (TA ? A[m * LD + r] : A[r * LD + m]) * (TB ? B[c * LD + m] : B[m * LD + c]);
        C[j] = s;
    }
}
// y (thread r < BP) = sum_m opA(r, m) v[m]
template <int BP, bool TA>
__device__ __forceinline__ double tile_vec(const double *A, const double *v)
{
    constexpr int LD = BP + 1;
    const int r = threadIdx.x;
    double s = 0.0;
    if (r < BP)
        for (int m = 0; m < BP; ++m) s += (TA ? A[m * LD + r] : A[r * LD + m]) * v[m];
    return s;
}

// S(gr, gc), gr >= gc, of the block-band store (6x6 column-major blocks, lower; diagonal blocks hold their lower triangle)
__device__ __forceinline__ double band_entry(const double *__restrict__ Hblk, int Bb, int64_t gr, int64_t gc)
{
    const int64_t I = gr / 6, J = gc / 6;
    if (I - J > Bb) return 0.0;
    const int r = (int)(gr - 6 * I), c = (int)(gc - 6 * J);
    return Hblk[(J * (int64_t)(Bb + 1) + (I - J)) * 36 + c * 6 + r];
}

template <int BP>
__global__ __launch_bounds__(256) void bcr_assemble_kernel(BcrDev p, const double *__restrict__ Hblk, const double *__restrict__ g,
                                                           const double *__restrict__ u_dev)
{
    const int R = blockIdx.x;
    const double u = u_dev[0];
    const int real = 6 * p.k; // scalars of a block row that belong to cameras (the rest is identity padding)
    auto gidx = [&](int Rb, int l) -> int64_t { // global scalar of local index l of block row Rb, or -1 (padding)
        if (l >= real) return -1;
        const int64_t cam = (int64_t)Rb * p.k + l / 6;
        return cam < p.M ? 6 * cam + (l % 6) : -1;
    };
    double *D = p.D + (int64_t)R * BP * BP, *L = p.L + (int64_t)R * BP * BP;
    for (int e = threadIdx.x; e < BP * BP; e += 256) {
        const int l1 = e / BP, l2 = e % BP;
        const int64_t g1 = gidx(R, l1), g2 = gidx(R, l2);
        double d = (l1 == l2) ? 1.0 : 0.0;
        if (g1 >= 0 && g2 >= 0) {
            d = g1 >= g2 ? band_entry(Hblk, p.Bb, g1, g2) : band_entry(Hblk, p.Bb, g2, g1);
            if (l1 == l2) d += u * d;
        } else if (g1 >= 0 || g2 >= 0)
            d = 0.0;
        D[e] = d;
        double lv = 0.0;
        if (R > 0 && g1 >= 0) {
            const int64_t h2 = gidx(R - 1, l2);
            if (h2 >= 0) lv = band_entry(Hblk, p.Bb, g1, h2);
        }
        L[e] = lv;
    }
    if (threadIdx.x < BP) {
        const int64_t g1 = gidx(R, threadIdx.x);
        p.rhs[(int64_t)R * BP + threadIdx.x] = g1 >= 0 ? -g[g1] : 0.0;
        // "not yet written" for the one-launch back substitution (bcr_back_all_kernel)
        reinterpret_cast<unsigned long long *>(p.x)[(int64_t)R * BP + threadIdx.x] = BCR_X_SENTINEL;
    }
}

// A: the odd rows of level `s` (s = stride); s <= 0: the last remaining row 0 alone
template <int BP>
__global__ __launch_bounds__(256) void bcr_A_kernel(BcrDev p, int s, int *__restrict__ status)
{
    constexpr int LD = BP + 1, PER = BP * BP / 256;
    __shared__ double Inv[BP * LD], X[BP * LD];
    __shared__ double v[BP];
    const int tid = threadIdx.x;
    const int i = s > 0 ? (2 * (int)blockIdx.x + 1) * s : 0;
    const int q = s > 0 ? i + s : p.nb; // right neighbour, if any
    // Gauss-Jordan on [A | Inv] with the entries in REGISTERS: thread t owns column c = t % BP of rows r_j = t / BP + (256 / BP) j
    // of both halves.  Per pivot only the pivot row and the pivot column travel through LDS (double-buffered: one barrier per
    // pivot); A is symmetric positive definite, so the pivots are taken in order.
    constexpr int RPT = BP * BP / 256, RS = 256 / BP; // rows per thread and their stride
    __shared__ double colb[2][BP], rowa[2][BP], rowi[2][BP];
    const int c = tid % BP, rb = tid / BP;
    double a[RPT], v_[RPT];
    {
        const double *Dg = p.D + (int64_t)i * BP * BP;
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            const int r = rb + RS * j;
            a[j] = Dg[r * BP + c];
            v_[j] = (r == c) ? 1.0 : 0.0;
        }
    }
    if (tid < BP) v[tid] = p.rhs[(int64_t)i * BP + tid];
#pragma unroll 1
    for (int kk = 0; kk < BP; ++kk) {
        const int pb = kk & 1;
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            const int r = rb + RS * j;
            if (c == kk) colb[pb][r] = a[j];
            if (r == kk) { rowa[pb][c] = a[j]; rowi[pb][c] = v_[j]; }
        }
        __syncthreads();
        const double piv = colb[pb][kk];
        if (tid == 0 && !(piv > 0.0)) status[0] = 1;
        const double rp = 1.0 / piv;
        const double ra = rowa[pb][c] * rp, ri = rowi[pb][c] * rp;
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            const int r = rb + RS * j;
            if (r == kk) { a[j] = ra; v_[j] = ri; }
            else { const double f = colb[pb][r]; a[j] -= f * ra; v_[j] -= f * ri; }
        }
    }
#pragma unroll
    for (int j = 0; j < RPT; ++j) Inv[(rb + RS * j) * LD + c] = v_[j];
    __syncthreads();
    // T1 = Inv L_i, T2 = Inv L_q^T, t = Inv rhs_i
    double C[PER];
    load_tile<BP>(X, p.L + (int64_t)i * BP * BP, s > 0);
    __syncthreads();
    tile_mul<BP, false, false>(Inv, X, C);
    {
        double *o = p.T1 + (int64_t)i * BP * BP;
#pragma unroll
        for (int j = 0; j < PER; ++j) o[tid + 256 * j] = C[j];
    }
    __syncthreads();
    load_tile<BP>(X, p.L + (int64_t)(q < p.nb ? q : 0) * BP * BP, q < p.nb);
    __syncthreads();
    tile_mul<BP, false, true>(Inv, X, C);
    {
        double *o = p.T2 + (int64_t)i * BP * BP;
#pragma unroll
        for (int j = 0; j < PER; ++j) o[tid + 256 * j] = C[j];
    }
    const double ti = tile_vec<BP, false>(Inv, v);
    if (tid < BP) p.t[(int64_t)i * BP + tid] = ti;
}

// B: the even rows r = 2 m s of level s
template <int BP>
__global__ __launch_bounds__(256) void bcr_B_kernel(BcrDev p, int s)
{
    constexpr int LD = BP + 1, PER = BP * BP / 256;
    __shared__ double X[BP * LD], Y[BP * LD];
    __shared__ double v[BP];
    const int tid = threadIdx.x;
    const int r = 2 * (int)blockIdx.x * s;
    const int il = r - s, ir = r + s; // the odd neighbours eliminated at this level
    const bool hasl = il >= 0, hasr = ir < p.nb;
    double dD[PER], dL[PER], drhs = 0.0;
#pragma unroll
    for (int j = 0; j < PER; ++j) dD[j] = dL[j] = 0.0;
    double C[PER];
    if (hasl) {
        load_tile<BP>(X, p.L + (int64_t)r * BP * BP, true);
        load_tile<BP>(Y, p.T2 + (int64_t)il * BP * BP, true);
        if (tid < BP) v[tid] = p.t[(int64_t)il * BP + tid];
        __syncthreads();
        tile_mul<BP, false, false>(X, Y, C); // L_r T2_il
#pragma unroll
        for (int j = 0; j < PER; ++j) dD[j] += C[j];
        drhs += tile_vec<BP, false>(X, v);
        __syncthreads();
        load_tile<BP>(Y, p.T1 + (int64_t)il * BP * BP, true);
        __syncthreads();
        tile_mul<BP, false, false>(X, Y, C); // L_r T1_il -> the new coupling to r - 2 s
#pragma unroll
        for (int j = 0; j < PER; ++j) dL[j] = C[j];
        __syncthreads();
    }
    if (hasr) {
        load_tile<BP>(X, p.L + (int64_t)ir * BP * BP, true);
        load_tile<BP>(Y, p.T1 + (int64_t)ir * BP * BP, true);
        if (tid < BP) v[tid] = p.t[(int64_t)ir * BP + tid];
        __syncthreads();
        tile_mul<BP, true, false>(X, Y, C); // L_ir^T T1_ir
#pragma unroll
        for (int j = 0; j < PER; ++j) dD[j] += C[j];
        drhs += tile_vec<BP, true>(X, v);
    }
    double *D = p.D + (int64_t)r * BP * BP, *L = p.L + (int64_t)r * BP * BP;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        D[tid + 256 * j] -= dD[j];
        L[tid + 256 * j] = (hasl && r - 2 * s >= 0) ? -dL[j] : 0.0;
    }
    if (tid < BP) p.rhs[(int64_t)r * BP + tid] -= drhs;
}

// One level in ONE launch (block rows of 32 scalars): the workgroup of the even row r inverts BOTH odd neighbours itself (two
// halves of a 512-thread workgroup, side by side), so no workgroup waits for another one inside a level -- the A / B pair above
// costs two dependent launches per level, and at ~25 + 14 us each (single-workgroup latency, not work) the nine levels of a
// 2 000-camera system were two thirds of the visual stage's LM iteration.  An odd row is inverted twice (by its left and its
// right even neighbour); T1 / T2 / t of an odd row are stored by the LEFT one (every odd row has it), for the back substitution.
// The new coupling blocks L_r go to the OTHER of two L arrays (Ldst): the right neighbour's workgroup still reads the old
// L_{r} of this level (as its L_q) while this one finishes.  `last`: row 0 is the only row left after this level -- the
// workgroup also solves x_0 = D_0^-1 rhs_0.
typedef double bcr_d4 __attribute__((ext_vector_type(4)));

#ifdef LVBA_BCR_TIMING
__device__ unsigned long long g_bcr_clk[16];
#define BCR_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_bcr_clk[k] = __builtin_readcyclecounter(); } while (0)
#define BCR_PH(k) do { const unsigned long long t_ = __builtin_readcyclecounter(); ph[k] += t_ - tph; tph = t_; } while (0)
#else
#define BCR_STAMP(k) do { } while (0)
#define BCR_PH(k) do { } while (0)
#endif
// ---- 32 x 32 inverse by ONE wavefront, on the matrix pipe ---------------------------------------------------------------
// [A | I] (32 x 64) lives in the accumulators of eight 16 x 16 MFMA tiles (register e of lane l of tile (ti, tj) is the entry
// [16 ti + (l >> 4) + 4 e][16 tj + (l & 15)]).  Gauss-Jordan with 4 x 4 BLOCK pivots: per pivot block k the four pivot rows and
// the four pivot columns go through LDS once (no workgroup barrier: one wavefront), every lane inverts the 4 x 4 pivot block for
// itself (LDL^T with scalar pivots: positive definite, no pivoting -- the same elimination order as scalar Gauss-Jordan), forms
// its component of P^-1 * (pivot rows), and the rank-4 update of all other rows is ONE MFMA per tile (K = 4); the pivot rows
// themselves are the B operand the lane already holds.  Measured (tools/bcr_microbench.hip): 8 block steps of ~2 400 cycles -- ~1 400 of
// them the 4 x 4 inverse and the operands, dependent fp64 chains of one wavefront at 4 cycles per instruction -- = 19-20 k cycles
// against 22 k for the 128-thread scalar-pivot form it replaces (32 steps of ~700 through LDS and a workgroup barrier each).
#define BCR_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#define BCR_ROWS 68 // stride of the pivot-row buffer [4][64]
#define BCR_COLS 5  // stride of the pivot-column buffer [32][4]
#define BCR_XCH (4 * BCR_ROWS + 32 * BCR_COLS)
// inverse of a symmetric positive definite 4 x 4 block (lower part of P read), all in registers; false if a pivot is <= 0
__device__ __forceinline__ bool inv4_spd(const double (&P)[4][4], double (&Pi)[4][4])
{
    double L[4][4], d[4], rd[4];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double dj = P[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) dj -= L[j][k] * L[j][k] * d[k];
        d[j] = dj;
        ok = ok && (dj > 0.0);
        double r = __builtin_amdgcn_rcp(dj);
        r = fma(r, fma(-dj, r, 1.0), r);
        r = fma(r, fma(-dj, r, 1.0), r);
        rd[j] = r;
#pragma unroll
        for (int i = j + 1; i < 4; ++i) {
            double v = P[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) v -= L[i][k] * L[j][k] * d[k];
            L[i][j] = v * r;
        }
    }
    double M[4][4]; // M = L^-1 (unit lower)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < i; ++j) {
            double v = -L[i][j];
#pragma unroll
            for (int k = j + 1; k < i; ++k) v -= L[i][k] * M[k][j];
            M[i][j] = v;
        }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b <= a; ++b) { // P^-1 = M^T D^-1 M
            double v = 0.0;
#pragma unroll
            for (int c = a; c < 4; ++c) v += (c == a ? 1.0 : M[c][a]) * rd[c] * (c == b ? 1.0 : M[c][b]);
            Pi[a][b] = v;
            Pi[b][a] = v;
        }
    return ok;
}
__device__ __forceinline__ bool gj32_wave(bcr_d4 (&acc)[2][4], double *xch)
{
    double *rowbuf = xch, *colbuf = xch + 4 * BCR_ROWS;
    const int l = threadIdx.x & 63, lo = l & 15, hi = l >> 4;
    bool ok = true;
#ifdef LVBA_BCR_TIMING
    unsigned long long ph[4] = {0, 0, 0, 0}, tph = __builtin_readcyclecounter();
#endif
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int tip = k >> 2, rk = k & 3; // the pivot rows 4k .. 4k+3 are register rk of tile row tip; the pivot columns lie in tile column tip
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) rowbuf[hi * BCR_ROWS + 16 * tj + lo] = acc[tip][tj][rk];
        if ((lo >> 2) == rk) {
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int e = 0; e < 4; ++e) colbuf[(16 * ti + hi + 4 * e) * BCR_COLS + (lo & 3)] = acc[ti][tip][e];
        }
        BCR_WAVE_SYNC();
        BCR_PH(0);
        double P[4][4], Pi[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b <= a; ++b) P[a][b] = rowbuf[a * BCR_ROWS + 4 * k + b];
        ok = inv4_spd(P, Pi) && ok;
        BCR_PH(1);
        double pin[4]; // row `hi` of P^-1
#pragma unroll
        for (int q = 0; q < 4; ++q) pin[q] = hi == 0 ? Pi[0][q] : hi == 1 ? Pi[1][q] : hi == 2 ? Pi[2][q] : Pi[3][q];
        double rowp[4]; // (P^-1 * pivot rows)[hi][16 tj + lo]: the B operand of the update AND the new pivot rows
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) {
            double v = 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) v += pin[q] * rowbuf[q * BCR_ROWS + 16 * tj + lo];
            rowp[tj] = v;
        }
        double av[2]; // -A[16 ti + lo][4 k + hi], zero on the pivot rows (they are overwritten below)
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
            const int R = 16 * ti + lo;
            av[ti] = (R >> 2) == k ? 0.0 : -colbuf[R * BCR_COLS + hi];
        }
        BCR_WAVE_SYNC();
        BCR_PH(2);
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ti], rowp[tj], acc[ti][tj], 0, 0, 0);
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) acc[tip][tj][rk] = rowp[tj];
        BCR_PH(3);
    }
#ifdef LVBA_BCR_TIMING
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int q = 0; q < 4; ++q) g_bcr_clk[8 + q] = ph[q];
#endif
    return ok;
}

// one 16 x 16 tile (ti, tj) of opA * opB out of LDS on the matrix pipe (v_mfma_f64_16x16x4_f64: lane l supplies A[l & 15][l >> 4]
// and B[l >> 4][l & 15]; register e of lane l is the result's [(l >> 4) + 4 e][l & 15]); tiles are row-major with stride 33
template <bool TA, bool TB>
__device__ __forceinline__ void mfma_tile32(const double *A, const double *B, int ti, int tj, bcr_d4 &acc)
{
    constexpr int LD = 33;
    const int l = threadIdx.x & 63, lo = l & 15, hi = l >> 4;
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
        const int k = 4 * kb + hi;
        const double av = TA ? A[k * LD + 16 * ti + lo] : A[(16 * ti + lo) * LD + k];
        const double bv = TB ? B[(16 * tj + lo) * LD + k] : B[k * LD + 16 * tj + lo];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
    }
}

// One level in ONE launch (block rows of 32 scalars): the workgroup of the even row r inverts BOTH odd neighbours itself (two
// wavefronts, one each, side by side), so no workgroup waits for another one inside a level -- the A / B pair above costs two
// dependent launches per level, and each is the latency of ONE workgroup (26 + 11 us measured: a 256-thread Gauss-Jordan at
// ~1 500 cycles per pivot, LDS-bandwidth-bound scalar tile products), nine times over for 2 000 cameras.  Here the seven
// 32 x 32 x 32 products of a level run on the matrix pipe (one wavefront per product).  An odd row is inverted twice (by its
// left and its right even neighbour); T1 / T2 / t of an odd row are stored by the LEFT one (every odd row has it), for the back
// substitution.  The new coupling blocks L_r go to the OTHER of two L arrays (Ldst): the right neighbour's workgroup still
// reads the old L_r of this level (as its L_q) while this one finishes.  `last`: row 0 is the only row left after this
// level -- the workgroup also solves x_0 = D_0^-1 rhs_0.
__global__ __launch_bounds__(256) void bcr_level_kernel(BcrDev p, int s, const double *__restrict__ Lsrc, double *__restrict__ Ldst,
                                                        int last, int *__restrict__ status, double *__restrict__ out)
{
    constexpr int BP = 32, LD = BP + 1, TS = BP * LD;
    __shared__ double buf[6][TS]; // 0: Inv_il -> T1_il | 1: Inv_ir -> T1_ir | 2: L_il -> T2_il | 3: L_r | 4: L_ir | 5: L_q
    __shared__ double xch[2][BCR_XCH]; // pivot rows / columns of the two inversions (gj32_wave)
    __shared__ double vrhs[2][BP], vt[2][BP];
    const int tid = threadIdx.x, gp = tid >> 7, lt = tid & 127, wv = tid >> 6, l = tid & 63;
    const int r = 2 * (int)blockIdx.x * s, il = r - s, ir = r + s, q = ir + s;
    const bool hasl = il >= 0, hasr = ir < p.nb, hasq = hasr && q < p.nb;
    const int64_t BB = BP * BP;
    BCR_STAMP(0);
    // every global read of the level up front: the two diagonal blocks (Gauss-Jordan layout, registers), four coupling blocks
    const int io = gp == 0 ? il : ir;
    const bool has = gp == 0 ? hasl : hasr;
    const int lo = l & 15, hi = l >> 4;
    const bool gjw = (wv & 1) == 0; // wavefronts 0 and 2 invert the left / right neighbour's diagonal block
    bcr_d4 acc[2][4];
    if (gjw) {
        const double *Dg = p.D + (int64_t)(has ? io : 0) * BB;
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int R = 16 * ti + hi + 4 * e, C = 16 * tj + lo;
                    acc[ti][tj][e] = has ? Dg[R * BP + C] : (R == C ? 1.0 : 0.0); // a missing neighbour: identity (its coupling blocks are zero)
                    acc[ti][2 + tj][e] = (R == C) ? 1.0 : 0.0;
                }
    }
    {
        const double *g2 = Lsrc + (int64_t)(hasl ? il : 0) * BB, *g3 = Lsrc + (int64_t)r * BB;
        const double *g4 = Lsrc + (int64_t)(hasr ? ir : 0) * BB, *g5 = Lsrc + (int64_t)(hasq ? q : 0) * BB;
#pragma unroll
        for (int e0 = 0; e0 < BP * BP; e0 += 256) {
            const int e = e0 + tid, o = (e / BP) * LD + (e % BP);
            buf[2][o] = hasl ? g2[e] : 0.0;
            buf[3][o] = hasl ? g3[e] : 0.0; // L_r couples r to r - s = il
            buf[4][o] = hasr ? g4[e] : 0.0;
            buf[5][o] = hasq ? g5[e] : 0.0;
        }
    }
    if (lt < BP) vrhs[gp][lt] = has ? p.rhs[(int64_t)io * BP + lt] : 0.0;
    // this row's own blocks: needed at the very end, asked for now
    double dold[4], rold = 0.0;
    const int ti = wv >> 1, tj = wv & 1; // wavefront wv owns tile (ti, tj) of the results for row r
#pragma unroll
    for (int e = 0; e < 4; ++e) dold[e] = p.D[(int64_t)r * BB + (16 * ti + hi + 4 * e) * BP + 16 * tj + lo];
    if (tid < BP) rold = p.rhs[(int64_t)r * BP + tid];
    BCR_STAMP(1);
    if (gjw) {
        if (!gj32_wave(acc, xch[gp]) && l == 0) status[0] = 1;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int u2 = 0; u2 < 2; ++u2)
#pragma unroll
                for (int e = 0; e < 4; ++e) buf[gp][(16 * t2 + hi + 4 * e) * LD + 16 * u2 + lo] = acc[t2][2 + u2][e];
    }
    BCR_STAMP(2);
    __syncthreads();
    // wavefront 0: T1_il = Inv_il L_il | 1: T2_il = Inv_il L_r^T | 2: T1_ir = Inv_ir L_ir | 3: T2_ir = Inv_ir L_q^T ; t = Inv rhs
    bcr_d4 P[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        P[t] = bcr_d4{0.0, 0.0, 0.0, 0.0};
        if (wv == 0) mfma_tile32<false, false>(buf[0], buf[2], t >> 1, t & 1, P[t]);
        else if (wv == 1) mfma_tile32<false, true>(buf[0], buf[3], t >> 1, t & 1, P[t]);
        else if (wv == 2) mfma_tile32<false, false>(buf[1], buf[4], t >> 1, t & 1, P[t]);
        else mfma_tile32<false, true>(buf[1], buf[5], t >> 1, t & 1, P[t]);
    }
    double tsave = 0.0;
    if (lt < BP) {
        double sacc = 0.0;
#pragma unroll 8
        for (int m = 0; m < BP; ++m) sacc += buf[gp][lt * LD + m] * vrhs[gp][m];
        vt[gp][lt] = sacc;
        tsave = sacc;
    }
    __syncthreads();
    if (wv < 3) {
        double *dstb = wv == 0 ? buf[0] : wv == 1 ? buf[2] : buf[1];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) dstb[(16 * (t >> 1) + hi + 4 * e) * LD + 16 * (t & 1) + lo] = P[t][e];
    }
    __syncthreads();
    BCR_STAMP(3);
    // D_r -= L_r T2_il + L_ir^T T1_ir ; L_r <- -L_r T1_il ; rhs_r -= L_r t_il + L_ir^T t_ir : tile (ti, tj) per wavefront
    bcr_d4 dD = bcr_d4{0.0, 0.0, 0.0, 0.0}, dL = bcr_d4{0.0, 0.0, 0.0, 0.0};
    mfma_tile32<false, false>(buf[3], buf[2], ti, tj, dD);
    mfma_tile32<true, false>(buf[4], buf[1], ti, tj, dD);
    mfma_tile32<false, false>(buf[3], buf[0], ti, tj, dL);
    double *D = p.D + (int64_t)r * BB, *Lo = Ldst + (int64_t)r * BB;
    double dnew[4];
    const bool keepl = hasl && r - 2 * s >= 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int o = (16 * ti + hi + 4 * e) * BP + 16 * tj + lo;
        dnew[e] = dold[e] - dD[e];
        D[o] = dnew[e];
        Lo[o] = keepl ? -dL[e] : 0.0;
    }
    double rnew = 0.0;
    if (tid < BP) {
        double d1 = 0.0, d2 = 0.0;
#pragma unroll 8
        for (int m = 0; m < BP; ++m) { d1 += buf[3][tid * LD + m] * vt[0][m]; d2 += buf[4][m * LD + tid] * vt[1][m]; }
        rnew = rold - (d1 + d2);
        p.rhs[(int64_t)r * BP + tid] = rnew;
    }
    // T1 / T2 / t of the right neighbour, for the back substitution: global stores last (a barrier would wait for them)
    if (hasr && gp == 1 && lt < BP) p.t[(int64_t)ir * BP + lt] = tsave;
    if (hasr && wv >= 2) {
        double *o = (wv == 2 ? p.T1 : p.T2) + (int64_t)ir * BB;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) o[(16 * (t >> 1) + hi + 4 * e) * BP + 16 * (t & 1) + lo] = P[t][e];
    }
    BCR_STAMP(4);
    if (!last) return;
    // row 0 alone is left: x_0 = D_0^-1 rhs_0 (the first half inverts, the second one keeps step with an identity)
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) buf[5][(16 * ti + hi + 4 * e) * LD + 16 * tj + lo] = dnew[e];
    if (tid < BP) vrhs[0][tid] = rnew;
    __syncthreads();
    if (wv == 0) {
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int u2 = 0; u2 < 2; ++u2)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int R = 16 * t2 + hi + 4 * e, C = 16 * u2 + lo;
                    acc[t2][u2][e] = buf[5][R * LD + C];
                    acc[t2][2 + u2][e] = (R == C) ? 1.0 : 0.0;
                }
        if (!gj32_wave(acc, xch[0]) && l == 0) status[0] = 1;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int u2 = 0; u2 < 2; ++u2)
#pragma unroll
                for (int e = 0; e < 4; ++e) buf[0][(16 * t2 + hi + 4 * e) * LD + 16 * u2 + lo] = acc[t2][2 + u2][e];
    }
    __syncthreads();
    if (tid < BP) {
        double sacc = 0.0;
        for (int m = 0; m < BP; ++m) sacc += buf[0][tid * LD + m] * vrhs[0][m];
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(p.x) + tid, (unsigned long long)__double_as_longlong(sacc),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // what the back substitution's workgroups poll for
        p.t[tid] = sacc;
        if (tid < 6 * p.k && tid / 6 < p.M) out[tid] = sacc;
    }
}

// back substitution of level s: x_i = t_i - T1_i x_{i-s} - T2_i x_{i+s} for the odd rows; s <= 0: x_0 = t_0
template <int BP>
__global__ __launch_bounds__(256) void bcr_back_kernel(BcrDev p, int s)
{
    constexpr int LD = BP + 1;
    __shared__ double X[BP * LD];
    __shared__ double v[BP];
    const int tid = threadIdx.x;
    if (s <= 0) {
        if (tid < BP) p.x[tid] = p.t[tid];
        return;
    }
    const int i = (2 * (int)blockIdx.x + 1) * s, q = i + s;
    double acc = tid < BP ? p.t[(int64_t)i * BP + tid] : 0.0;
    load_tile<BP>(X, p.T1 + (int64_t)i * BP * BP, true);
    if (tid < BP) v[tid] = p.x[(int64_t)(i - s) * BP + tid];
    __syncthreads();
    acc -= tile_vec<BP, false>(X, v);
    __syncthreads();
    if (q < p.nb) {
        load_tile<BP>(X, p.T2 + (int64_t)i * BP * BP, true);
        if (tid < BP) v[tid] = p.x[(int64_t)q * BP + tid];
        __syncthreads();
        acc -= tile_vec<BP, false>(X, v);
    }
    if (tid < BP) p.x[(int64_t)i * BP + tid] = acc;
}

// The whole back substitution of the 32-scalar form as ONE launch (nine launches of ~4 us each before): workgroup b owns one
// odd row of one level, the levels from the top (largest stride) down, so a workgroup only ever waits for workgroups with a
// smaller blockIdx (dispatched earlier) or for x_0 of the last level kernel.  x is the only channel between workgroups:
// pre-filled with a NaN sentinel (bcr_assemble_kernel), written with agent-scope atomic stores, polled with agent-scope atomic
// loads -- 8 bytes carry data and flag at once (the scheme of ldlt_back_chain_kernel).  T1 / T2 / t of the row are fetched before
// the wait.  Every workgroup also writes its cameras' part of the solution in the caller's layout (30 consecutive scalars: no
// scatter kernel).  At most nb - 1 <= ~400 small workgroups: all resident at once on any device this library runs on.
__global__ __launch_bounds__(256) void bcr_back_all_kernel(BcrDev p, int top, double *__restrict__ out)
{
    constexpr int BP = 32, LD = BP + 1;
    __shared__ double X1[BP * LD], X2[BP * LD];
    __shared__ double v1[BP], v2[BP];
    const int tid = threadIdx.x;
    int b = (int)blockIdx.x, s = top;
    for (;; s >>= 1) { // level of this workgroup
        const int n_odd = (p.nb - s + 2 * s - 1) / (2 * s);
        if (b < n_odd || s == 1) break;
        b -= n_odd;
    }
    const int i = (2 * b + 1) * s, q = i + s, left = i - s;
    if (i >= p.nb) return;
    const bool hasq = q < p.nb;
    load_tile<BP>(X1, p.T1 + (int64_t)i * BP * BP, true);
    load_tile<BP>(X2, p.T2 + (int64_t)(hasq ? i : 0) * BP * BP, hasq);
    double acc = tid < BP ? p.t[(int64_t)i * BP + tid] : 0.0;
    if (tid < 2 * BP) {
        const int m = tid & (BP - 1);
        const bool second = tid >= BP;
        double xv = 0.0;
        if (!second || hasq) {
            const unsigned long long *px = reinterpret_cast<const unsigned long long *>(p.x) + (int64_t)(second ? q : left) * BP + m;
            unsigned long long v;
            while ((v = __hip_atomic_load(px, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == BCR_X_SENTINEL) __builtin_amdgcn_s_sleep(1);
            xv = __longlong_as_double((long long)v);
        }
        (second ? v2 : v1)[m] = xv;
    }
    __syncthreads();
    if (tid < BP) {
        double a1 = 0.0, a2 = 0.0;
#pragma unroll 8
        for (int m = 0; m < BP; ++m) { a1 += X1[tid * LD + m] * v1[m]; a2 += X2[tid * LD + m] * v2[m]; }
        acc -= a1 + a2;
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(p.x) + (int64_t)i * BP + tid, (unsigned long long)__double_as_longlong(acc),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int64_t cam = (int64_t)i * p.k + tid / 6;
        if (tid < 6 * p.k && cam < p.M) out[6 * (int64_t)i * p.k + tid] = acc;
    }
}

template <int BP>
__global__ void bcr_scatter_kernel(BcrDev p, double *__restrict__ out)
{
    const int64_t a = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; // global scalar
    if (a >= 6 * (int64_t)p.M) return;
    const int64_t cam = a / 6, R = cam / p.k;
    const int l = (int)(6 * (cam - R * p.k) + (a - 6 * cam));
    out[a] = p.x[R * BP + l];
}

template <int BP>
void bcr_run(const BcrDev &p, const double *Hblk, const double *g, const double *u_dev, double *x, int *status, hipStream_t s)
{
    hipMemsetAsync(status, 0, sizeof(int), s);
    hipLaunchKernelGGL(bcr_assemble_kernel<BP>, dim3((unsigned)p.nb), dim3(256), 0, s, p, Hblk, g, u_dev);
    int top = 0; // strides 1, 2, 4, ... while an odd row exists (stride < nb)
    if constexpr (BP == 32) {
        if (p.nb > 1) {
            double *Lbuf[2] = {p.L, p.L2};
            int cur = 0;
            for (int st = 1; st < p.nb; st *= 2) {
                const int n_even = (p.nb + 2 * st - 1) / (2 * st); // r = 2 m st < nb
                hipLaunchKernelGGL(bcr_level_kernel, dim3((unsigned)n_even), dim3(256), 0, s, p, st, Lbuf[cur], Lbuf[cur ^ 1], 2 * st >= p.nb ? 1 : 0, status, x);
                cur ^= 1;
                top = st;
            }
            hipLaunchKernelGGL(bcr_back_all_kernel, dim3((unsigned)(p.nb - 1)), dim3(256), 0, s, p, top, x);
            return;
        }
    }
    for (int st = 1; st < p.nb; st *= 2) {
        const int n_odd = (p.nb - st + 2 * st - 1) / (2 * st);  // i = (2m+1) st < nb
        const int n_even = (p.nb + 2 * st - 1) / (2 * st);      // r = 2 m st < nb
        hipLaunchKernelGGL(bcr_A_kernel<BP>, dim3((unsigned)n_odd), dim3(256), 0, s, p, st, status);
        hipLaunchKernelGGL(bcr_B_kernel<BP>, dim3((unsigned)n_even), dim3(256), 0, s, p, st);
        top = st;
    }
    hipLaunchKernelGGL(bcr_A_kernel<BP>, dim3(1), dim3(256), 0, s, p, 0, status); // row 0 alone: t_0 = D_0^-1 rhs_0
    hipLaunchKernelGGL(bcr_back_kernel<BP>, dim3(1), dim3(256), 0, s, p, 0);
    for (int st = top; st >= 1; st /= 2) {
        const int n_odd = (p.nb - st + 2 * st - 1) / (2 * st);
        hipLaunchKernelGGL(bcr_back_kernel<BP>, dim3((unsigned)n_odd), dim3(256), 0, s, p, st);
    }
    hipLaunchKernelGGL(bcr_scatter_kernel<BP>, dim3((unsigned)((6 * (int64_t)p.M + 255) / 256)), dim3(256), 0, s, p, x);
}

inline int bcr_block_cams(int band_blocks) { return band_blocks < 5 ? 5 : band_blocks; } // k >= Bb cameras per block row
inline int bcr_pad(int k) { return 6 * k <= 32 ? 32 : 64; }

} // namespace

// Applicable to camera half-bandwidths up to 10 (block rows of <= 64 scalars) and systems with at least a few block rows.
bool bcr_applicable(int n_poses, int band_blocks)
{
    if (band_blocks < 1 || band_blocks > 10) return false;
    return n_poses >= 8 * bcr_block_cams(band_blocks);
}

int64_t bcr_workspace_doubles(int n_poses, int band_blocks)
{
    if (!bcr_applicable(n_poses, band_blocks)) return 0;
    const int k = bcr_block_cams(band_blocks), BP = bcr_pad(k);
    const int64_t nb = (n_poses + k - 1) / k;
    return nb * (5 * (int64_t)BP * BP + 3 * BP) + 64;
}

// x = -(S + u diag S)^-1 g from the block-band store (u is read from device memory; the visual stage passes 0).
void bcr_solve(const double *Hblk, int band_blocks, int n_poses, const double *g, const double *u_dev, double *x, double *work,
               int *status, hipStream_t s)
{
    BcrDev p;
    p.k = bcr_block_cams(band_blocks);
    const int BP = bcr_pad(p.k);
    p.nb = (n_poses + p.k - 1) / p.k;
    p.M = n_poses; p.Bb = band_blocks;
    const int64_t m2 = (int64_t)p.nb * BP * BP, m1 = (int64_t)p.nb * BP;
    p.D = work; p.L = p.D + m2; p.T1 = p.L + m2; p.T2 = p.T1 + m2;
    p.L2 = p.T2 + m2;
    p.rhs = p.L2 + m2; p.t = p.rhs + m1; p.x = p.t + m1;
    if (BP == 32) bcr_run<32>(p, Hblk, g, u_dev, x, status, s);
    else bcr_run<64>(p, Hblk, g, u_dev, x, status, s);
}

} // namespace lvba
