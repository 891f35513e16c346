// ldlt_prepare.h -- filling the solver's storage from the block-band Hessian store, and the kernels that join the two ends of a
// two-ended ("twisted") factorisation (included by ldlt.hip only, inside namespace lvba).
#pragma once

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
#define LVBA_PB_CHUNKS 4
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
    const double uj = grp ? u_dev[grp[P]] : u_dev[0];
    const bool dead = uj < 0.0; // a finished group: identity block, see ldlt_prepare_kernel
    // LVBA_PB_CHUNKS chunks of 28 block offsets per workgroup, one after the other: with one chunk each, C3 launched 41 k workgroups
    // that wrote 8 KB apiece (0.13 ms for 320 MB: their set-up, not the stores, was the time)
    for (int yc = 0; yc < LVBA_PB_CHUNKS; ++yc) {
    const int d0 = ((int)blockIdx.y * LVBA_PB_CHUNKS + yc) * LVBA_PB_BLOCKS; // first block offset dI (matrix 1: I = P + dI) / dJ (matrix 2: J = P - dJ)
    if (6 * d0 - 5 >= ldab) break; // (block-uniform: nothing of this chunk lies inside the stored offsets)
    if (yc) __syncthreads();       // everybody is done with the chunk before in LDS
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
