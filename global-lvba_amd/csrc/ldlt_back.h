// ldlt_back.h -- the backward substitution of the band LDL^T as ONE chained launch (included by ldlt.hip only, inside namespace lvba).
#pragma once

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
