// ldlt_back.h -- the backward substitution of the band LDL^T as ONE chained launch (included by ldlt.hip only, inside namespace lvba).
#pragma once

// Whole backward substitution as ONE launch: a workgroup owns one panel j and walks the chain
//   x_j = G D (G^T b_j - sum_{i>j+1} L_ij^T x_i)  -  (G D L_{j+1,j}^T) x_{j+1}
// Only the second term waits for the neighbour: M_j = G D L_{j+1,j}^T is formed in the prologue (all workgroups at once, off the
// chain) and kept as one row per lane in wavefront 0's registers; the first term is ready a step earlier (wavefronts 1..3 walk the
// far tiles, a whole 64 x 64 tile per wavefront: lane = column, x_i broadcast with v_readlane).  What is left on the chain per
// panel is the hand-off of 64 doubles and ONE 64 x 64 mat-vec in one wavefront: no LDS, no barrier.
// The x vector is the only inter-workgroup channel: it is pre-filled with a NaN sentinel (ldlt_prepare_kernel), written
// with agent-scope atomic stores and polled with agent-scope atomic loads, 8 bytes carrying data and flag at once -- no
// fences, no L2 write-back.  A workgroup waits only for workgroups of its own launch or of an earlier one, and a launch holds
// at most 256 workgroups of one per CU (256 registers, 105 KB of LDS): all of them become resident, the chain cannot starve.
#define LVBA_X_SENTINEL 0x7ff4dead5eed0001ULL
// gridDim.y = 2: two independent chains of the same geometry in one launch -- matrix 1's T part and matrix 2's B part of a
// twisted factorisation both start from x of S (blockIdx.y = 1: matrix 2 at a + sA / workspace + sW, solution vector x2).
__global__ __launch_bounds__(256) void ldlt_back_chain_kernel(LdltMat M, int j_top, const double *__restrict__ Gall,
                                                              const double *__restrict__ dvec, const double *__restrict__ b,
                                                              double *__restrict__ x, int64_t sA, int64_t sW,
                                                              double *__restrict__ x2)
{
    if (blockIdx.y) { M.a += sA; Gall += sW; dvec += sW; b += sW; x = x2; }
    constexpr int LS = 65, LT = 66;
    __shared__ double Gs[64 * LS]; // [c][m] = G[m][c]
    __shared__ double Lt[64 * LT]; // [c][e] = d_c L(j+1,j)[e][c]
    __shared__ double Ms[64 * LS]; // [e][i] = (G D L(j+1,j)^T)[i][e]
    __shared__ double bs[64], zred[4 * 64], red[4 * 64];
    const int tid = threadIdx.x, c = tid & 63, q = tid >> 6;
    // this launch covers the panels j_top, j_top - 1, ...  Workgroups go to the XCDs round-robin in dispatch order: chain
    // position t is dealt out so that neighbours on the chain sit on the same XCD (one L2) except at seven seams
    int t;
    {
        const int cnt = (int)gridDim.x, lin = (int)blockIdx.x + (int)blockIdx.y * cnt, xcd = lin & 7;
        t = 0;
        for (int y = 0; y < xcd; ++y) {
            const int first = (y - (int)blockIdx.y * cnt) & 7; // smallest blockIdx.x of this chain on XCD y
            t += first < cnt ? (cnt - first + 7) >> 3 : 0;
        }
        t += (int)blockIdx.x >> 3;
    }
    const int j = j_top - t;
    const int64_t n = M.n, k = (int64_t)j * 64;
    const int nbe = (int)((n - k) < 64 ? (n - k) : 64);
    const double *G = Gall + (int64_t)j * 4096;
    double gl[16];
#pragma unroll
    for (int it = 0; it < 16; ++it) gl[it] = G[tid + 256 * it];
    if (tid < 64) bs[tid] = (tid < nbe) ? b[k + tid] : 0.0;
    const double dc = (c < nbe) ? dvec[k + c] : 0.0;
    // this lane's column and the rows of it that lie inside the band (a lane past the end of the matrix reads column n-1 and
    // keeps nothing)
    const int64_t col = k + c, colc = col < n ? col : n - 1;
    int64_t rmaxc = col + M.bw;
    if (rmaxc > n - 1) rmaxc = n - 1;
    if (c >= nbe) rmaxc = -1;
    int64_t rvalid = colc + M.bw;
    if (rvalid > n - 1) rvalid = n - 1;
    const double *colp = M.a + colc * M.ld;
    int64_t rl = k + 63 + M.bw;
    if (rl > n - 1) rl = n - 1;
    const int ihi = (int)(rl >> 6);
    auto poll = [&](int i) { // x_i, one entry per lane
        const int64_t r = (int64_t)i * 64 + c;
        double xv = 0.0;
        if (r < n) {
            const unsigned long long *px = reinterpret_cast<const unsigned long long *>(x + r);
            unsigned long long v;
            while ((v = __hip_atomic_load(px, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == LVBA_X_SENTINEL)
                __builtin_amdgcn_s_sleep(1);
            xv = __longlong_as_double((long long)v);
        }
        return xv;
    };
    { // prologue: G and the neighbour tile (thread (c, q): rows 16q .. 16q+15 of column c, scaled by d_c) to LDS
        double tq[16];
        const int64_t r0 = (int64_t)(j + 1) * 64 + 16 * q;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            int64_t r = r0 + e;
            const bool in = r <= rmaxc;
            if (r > rvalid) r = rvalid;
            const double v = colp[r];
            tq[e] = in ? v * dc : 0.0;
        }
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int e = tid + 256 * it; // G[m][c']: m = e>>6, c' = e&63
            Gs[(e & 63) * LS + (e >> 6)] = gl[it];
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) Lt[c * LT + 16 * q + e] = tq[e];
    }
    __syncthreads();
    { // z_c = sum_m G[m][c] b_m : thread (c, q) sums m in [16q, 16q+16)
        double z = 0.0;
#pragma unroll
        for (int m = 0; m < 16; ++m) z += Gs[c * LS + 16 * q + m] * bs[16 * q + m];
        zred[q * 64 + c] = z;
    }
    if (ihi > j) { // M[i][e] = sum_cc G[i][cc] d_cc L[e][cc] : thread (i = c, q) forms e in [16q, 16q+16)
        double m[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) m[e] = 0.0;
        for (int cc = 0; cc < 64; ++cc) {
            const double g = Gs[cc * LS + c];
            const double *lt = Lt + cc * LT + 16 * q;
#pragma unroll
            for (int e = 0; e < 16; ++e) m[e] = fma(g, lt[e], m[e]);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) Ms[(16 * q + e) * LS + c] = m[e];
    }
    __syncthreads();
    double mrow[64], zc = 0.0;
    if (q == 0) { // wavefront 0: its row of M and z, then it waits for the others
        if (ihi > j) {
#pragma unroll
            for (int e = 0; e < 64; ++e) mrow[e] = Ms[e * LS + c];
        }
        zc = zred[c] + zred[64 + c] + zred[128 + c] + zred[192 + c];
    } else { // wavefronts 1..3: the far tiles i = j+1+q, j+4+q, ... from the far end; lane = column c, all 64 rows of the tile
        double acc0 = 0.0, acc1 = 0.0;
        const int i0 = j + 1 + q;
        if (ihi >= i0)
            for (int i = i0 + 3 * ((ihi - i0) / 3); i >= i0; i -= 3) {
                double t[64];
                const int64_t r0 = (int64_t)i * 64;
                if (__all(r0 + 63 <= rmaxc)) {
#pragma unroll
                    for (int e = 0; e < 64; e += 2) {
                        const double2 v = *reinterpret_cast<const double2 *>(colp + r0 + e);
                        t[e] = v.x; t[e + 1] = v.y;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 64; ++e) {
                        int64_t r = r0 + e;
                        const bool in = r <= rmaxc;
                        if (r > rvalid) r = rvalid;
                        const double v = colp[r];
                        t[e] = in ? v : 0.0;
                    }
                }
                const double xv = poll(i);
#pragma unroll
                for (int e = 0; e < 64; e += 2) {
                    acc0 = fma(t[e], readlane_f64(xv, e), acc0);
                    acc1 = fma(t[e + 1], readlane_f64(xv, e + 1), acc1);
                }
            }
        red[q * 64 + c] = acc0 + acc1;
    }
    __syncthreads();
    if (q == 0) {
        const double u = (c < nbe) ? (zc - (red[64 + c] + red[128 + c] + red[192 + c])) * dc : 0.0; // D (G^T b - far sum)
        double p0 = 0.0, p1 = 0.0; // x_i = sum_cc G[i][cc] u_cc, lane = row i
#pragma unroll
        for (int cc = 0; cc < 64; cc += 2) {
            p0 = fma(Gs[cc * LS + c], readlane_f64(u, cc), p0);
            p1 = fma(Gs[(cc + 1) * LS + c], readlane_f64(u, cc + 1), p1);
        }
        if (ihi > j) { // the only step on the chain: x_j -= M x_{j+1}
            const double xv = poll(j + 1);
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
            for (int e = 0; e < 64; e += 4) {
                s0 = fma(mrow[e], readlane_f64(xv, e), s0);
                s1 = fma(mrow[e + 1], readlane_f64(xv, e + 1), s1);
                s2 = fma(mrow[e + 2], readlane_f64(xv, e + 2), s2);
                s3 = fma(mrow[e + 3], readlane_f64(xv, e + 3), s3);
            }
            p0 -= (s0 + s1) + (s2 + s3);
        }
        if (c < nbe)
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(x + k + c), (unsigned long long)__double_as_longlong(p0 + p1),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
