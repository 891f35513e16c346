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
//     K1 ldlt_diag    1 workgroup: LDL^T of the 64x64 diagonal block, one register-resident row per
//                     thread.  The right-hand side and an identity are appended as extra ROWS, so the same elimination yields
//                     z_k = D^-1 L11^-1 b_k, y_k = L11^-1 b_k and G = L11^-T D^-1 for free.
//     K2 ldlt_panel   per 64-row tile: L21 = A21 * G (fp64 MFMA 16x16x4), Z = L21*D, b -= L21*y_k.
//     K3 ldlt_update  per 64x64 lower tile of the window: A22 -= L21 * Z^T (fp64 MFMA 16x16x4).
//   backward, per panel from the last: x_k = G D s_k (s = updated z), then b[c] -= A(k:k+64, c)^T x_k
//   for the <= bw columns left of the panel (right-looking, one launch per panel).
//
// MFMA operand layout used (v_mfma_f64_16x16x4_f64): lane l supplies A[i=l&15][k=l>>4] and
// B[k=l>>4][j=l&15]; result register r of lane l is D[(l>>4)+4r][l&15].  Products are arranged so that
// l&15 indexes ROWS of the column-major target, i.e. 16 lanes touch 128 contiguous bytes.
#include <hip/hip_runtime.h>
#include <math.h>
#include "lvba_internal.h"

namespace lvba {

typedef double d4 __attribute__((ext_vector_type(4)));
#define LVBA_TS 80 // LDS tile stride (doubles): 80 = 16 mod 32 -> MFMA operand reads are bank-conflict free

__global__ void ldlt_prepare_kernel(LdltMat M, const double *__restrict__ Hblk, int band_blocks, int n_poses,
                                    const double *__restrict__ g, const double *__restrict__ u_dev,
                                    double *__restrict__ b)
{
    const double u = u_dev[0];
    const int64_t Bb1 = band_blocks + 1;
    const int64_t total = (int64_t)n_poses * Bb1 * 36;
    const int64_t gid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, gsz = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = gid; e < total; e += gsz) {
        const int64_t slot = e / 36;
        const int el = (int)(e - slot * 36);
        const int64_t J = slot / Bb1, dI = slot - J * Bb1, I = J + dI;
        if (I >= n_poses) continue;
        const int c = el / 6, r = el - c * 6;
        if (dI == 0 && r < c) continue;
        double v = Hblk[e];
        if (dI == 0 && r == c) v += u * v;
        M.a[(6 * I + r) + (6 * J + c) * M.ld] = v;
    }
    for (int64_t a = gid; a < M.n; a += gsz) b[a] = -g[a];
}

// ---------------------------------------------------------------------------------------------- K1
// One thread per ROW, the row lives in registers (64 doubles, fully unrolled static indexing).  Rows 0..63
// are the diagonal block, row 64 the right-hand side, rows 65..128 an identity; eliminating column j needs
// only the (unscaled) column j broadcast through LDS -> ONE barrier per column (double-buffered).
// Outputs: L11 (strict lower) and D (diagonal) in place, z_k in b, y_k, d_k, and Gt[c][m] = G[m][c] with
// G = L11^-T D^-1.
__global__ __launch_bounds__(192) void ldlt_diag_kernel(LdltMat M, int64_t k, int nbe, double *__restrict__ Gt,
                                                        double *__restrict__ dvec, double *__restrict__ yvec,
                                                        double *__restrict__ b, int *__restrict__ status)
{
    __shared__ double wv[2][136];
    const int tid = threadIdx.x;
    double a[64];
    if (tid < 64) {
#pragma unroll
        for (int c = 0; c < 64; ++c) {
            double v = 0.0;
            if (tid < nbe) {
                if (c <= tid) v = M.a[(k + tid) + (k + c) * M.ld];
            } else if (c == tid)
                v = 1.0;
            a[c] = v;
        }
    } else if (tid == 64) {
#pragma unroll
        for (int c = 0; c < 64; ++c) a[c] = (c < nbe) ? b[k + c] : 0.0;
    } else {
#pragma unroll
        for (int c = 0; c < 64; ++c) a[c] = (c == tid - 65) ? 1.0 : 0.0;
    }
#pragma unroll
    for (int j = 0; j < 64; ++j) {
        double *w = wv[j & 1];
        if (tid <= 128) w[tid] = a[j];
        __syncthreads();
        double d = w[j];
        if (!(d != 0.0) || !isfinite(d)) {
            if (tid == 0) status[0] = 1;
            d = 1.0;
        }
        if (tid == j && j < nbe) dvec[k + j] = d;
        if (tid == 64 && j < nbe) yvec[k + j] = a[j];
        if (tid > j) {
            const double l = a[j] / d;
            a[j] = l;
#pragma unroll
            for (int c = j + 1; c < 64; ++c) a[c] -= l * w[c];
        }
    }
    if (tid < 64) {
        if (tid < nbe) {
#pragma unroll
            for (int c = 0; c < 64; ++c)
                if (c <= tid) M.a[(k + tid) + (k + c) * M.ld] = a[c];
        }
    } else if (tid == 64) {
#pragma unroll
        for (int c = 0; c < 64; ++c)
            if (c < nbe) b[k + c] = a[c];
    } else if (tid <= 128) {
        const int m = tid - 65;
#pragma unroll
        for (int c = 0; c < 64; ++c) Gt[c * 64 + m] = a[c];
    }
}

// ---------------------------------------------------------------------------------------------- K2
#define LVBA_GS 66 // stride of the [j][m] G tile: 66 = 2 mod 32 -> conflict-free A-operand reads
__global__ __launch_bounds__(256) void ldlt_panel_kernel(LdltMat M, int64_t k, int nbe, int64_t w0, int64_t rend,
                                                         const double *__restrict__ Gt,
                                                         const double *__restrict__ dvec,
                                                         const double *__restrict__ yvec, double *__restrict__ Zws,
                                                         int64_t ldz, double *__restrict__ b)
{
    __shared__ double As[64 * LVBA_TS]; // [m][row]; later the L tile as [j][row]
    __shared__ double Gs[64 * LVBA_GS]; // [j][m]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int row = tid & 63;
    const int64_t r0 = w0 + 64 * (int64_t)blockIdx.x;
    const int64_t r = r0 + row;
    double av[16], gv[16];
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int m = w + 4 * it;
        av[it] = (r < rend && m < nbe) ? M.a[r + (k + m) * M.ld] : 0.0;
        gv[it] = Gt[m * 64 + row]; // Gt[j = m][m' = row]
    }
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int m = w + 4 * it;
        As[m * LVBA_TS + row] = av[it];
        Gs[m * LVBA_GS + row] = gv[it];
    }
    __syncthreads();
    d4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
    const int i = lane & 15, kk = lane >> 4;
#pragma unroll 4
    for (int k0 = 0; k0 < 64; k0 += 4) {
        const double a = Gs[(16 * w + i) * LVBA_GS + k0 + kk]; // G[m = k0+kk][j = 16w+i]
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const double bv = As[(k0 + kk) * LVBA_TS + 16 * t + i]; // A21[row=16t+i][m]
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv, acc[t], 0, 0, 0);
        }
    }
    __syncthreads();
    // acc[t][reg] = L[row = 16t + i][j = 16w + kk + 4reg]
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) As[(16 * w + kk + 4 * reg) * LVBA_TS + 16 * t + i] = acc[t][reg];
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int j = w + 4 * it;
        if (r < rend && j < nbe) {
            const double v = As[j * LVBA_TS + row];
            M.a[r + (k + j) * M.ld] = v;
            Zws[(r - w0) + j * ldz] = v * dvec[k + j];
        }
    }
    if (tid < 64) {
        if (r < rend) {
            double s0 = 0.0, s1 = 0.0;
#pragma unroll 8
            for (int j = 0; j < 64; j += 2) {
                s0 += As[j * LVBA_TS + tid] * yvec[k + j];       // yvec/As are 0 beyond nbe? (As yes, yvec guarded)
                s1 += As[(j + 1) * LVBA_TS + tid] * yvec[k + j + 1];
            }
            b[r] -= s0 + s1;
        }
    }
}

// ---------------------------------------------------------------------------------------------- K3
__global__ __launch_bounds__(256) void ldlt_update_kernel(LdltMat M, int64_t k, int nbe, int64_t w0, int64_t rend,
                                                          const double *__restrict__ Zws, int64_t ldz)
{
    __shared__ double Ls[64 * LVBA_TS]; // [m][row of tile ti]
    __shared__ double Zs[64 * LVBA_TS]; // [m][row of tile tj] (= column of the updated tile)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // triangular decode blockIdx.x -> (ti >= tj)
    const int64_t bidx = blockIdx.x;
    int64_t ti = (int64_t)((sqrt(8.0 * (double)bidx + 1.0) - 1.0) * 0.5);
    while (ti * (ti + 1) / 2 > bidx) --ti;
    while ((ti + 1) * (ti + 2) / 2 <= bidx) ++ti;
    const int64_t tj = bidx - ti * (ti + 1) / 2;
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

// ---------------------------------------------------------------------------------------- backward
__global__ __launch_bounds__(256) void ldlt_back_kernel(LdltMat M, int64_t k, int nbe, const double *__restrict__ Gt,
                                                        const double *__restrict__ dvec, double *__restrict__ b,
                                                        double *__restrict__ x, int64_t cmin)
{
    constexpr int LS = 65;
    __shared__ double Gs[64 * LS]; // [i][c] = G[i][c]
    __shared__ double sd[64], xs[64];
    const int tid = threadIdx.x;
    const int64_t c = cmin + 256 * (int64_t)blockIdx.x + tid;
    // this thread's column segment A(k..k+63, c), issued before anything waits
    double colv[64];
    int64_t rmax = k + nbe - 1;
    if (c < k) {
        if (c + M.bw < rmax) rmax = c + M.bw;
        const double *col = M.a + c * M.ld + k;
#pragma unroll
        for (int q = 0; q < 64; ++q) colv[q] = (k + q <= rmax) ? col[q] : 0.0;
    }
    double gl[16];
#pragma unroll
    for (int it = 0; it < 16; ++it) gl[it] = Gt[tid + 256 * it];
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int e = tid + 256 * it; // Gt[c'][i]: c' = e>>6, i = e&63
        Gs[(e & 63) * LS + (e >> 6)] = gl[it];
    }
    if (tid < 64) sd[tid] = (tid < nbe) ? b[k + tid] * dvec[k + tid] : 0.0;
    __syncthreads();
    if (tid < 64) {
        double acc = 0.0;
        for (int cc = tid; cc < 64; ++cc) acc += Gs[tid * LS + cc] * sd[cc]; // x_k = (L11^-T D^-1) (D s)
        xs[tid] = acc;
        if (blockIdx.x == 0 && tid < nbe) x[k + tid] = acc;
    }
    __syncthreads();
    if (c < k) {
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int q = 0; q < 64; q += 2) {
            s0 += colv[q] * xs[q];
            s1 += colv[q + 1] * xs[q + 1];
        }
        b[c] -= s0 + s1;
    }
}

// ------------------------------------------------------------------------------------------- driver
static inline int64_t ldz_for(int64_t n, int64_t bw)
{
    int64_t w = bw + LVBA_NB + 64;
    return w < n ? w : n;
}

int64_t ldlt_workspace_doubles(int64_t n, int64_t bw)
{
    const int64_t nsteps = (n + LVBA_NB - 1) / LVBA_NB;
    return nsteps * 4096 /*G*/ + 3 * n /*d, y, b*/ + ldz_for(n, bw) * LVBA_NB /*Z*/ + 64;
}

void ldlt_solve(const LdltMat &A, const double *Hblk, int band_blocks, int n_poses, const double *g,
                const double *u_dev, double *x, double *work, int *status, hipStream_t s)
{
    const int64_t n = A.n, bw = A.bw;
    const int64_t nsteps = (n + LVBA_NB - 1) / LVBA_NB;
    double *Gall = work;
    double *dvec = Gall + nsteps * 4096;
    double *yvec = dvec + n;
    double *b = yvec + n;
    double *Zws = b + n;
    const int64_t ldz = ldz_for(n, bw);
    const size_t abytes = (size_t)((A.ld == n) ? n * n : (A.ld + 1) * n) * sizeof(double);
    hipMemsetAsync(A.a, 0, abytes, s);
    hipMemsetAsync(status, 0, sizeof(int), s);
    hipLaunchKernelGGL(ldlt_prepare_kernel, dim3(2048), dim3(256), 0, s, A, Hblk, band_blocks, n_poses, g, u_dev, b);
    for (int64_t st = 0; st < nsteps; ++st) {
        const int64_t k = st * LVBA_NB;
        const int nbe = (int)((n - k) < LVBA_NB ? (n - k) : LVBA_NB);
        const int64_t w0 = k + nbe;
        int64_t rend = k + nbe + bw;
        if (rend > n) rend = n;
        double *G = Gall + st * 4096;
        hipLaunchKernelGGL(ldlt_diag_kernel, dim3(1), dim3(192), 0, s, A, k, nbe, G, dvec, yvec, b, status);
        if (w0 < rend) {
            const int64_t T = (rend - w0 + 63) / 64;
            hipLaunchKernelGGL(ldlt_panel_kernel, dim3((unsigned)T), dim3(256), 0, s, A, k, nbe, w0, rend, G, dvec,
                               yvec, Zws, ldz, b);
            hipLaunchKernelGGL(ldlt_update_kernel, dim3((unsigned)(T * (T + 1) / 2)), dim3(256), 0, s, A, k, nbe, w0,
                               rend, Zws, ldz);
        }
    }
    for (int64_t st = nsteps - 1; st >= 0; --st) {
        const int64_t k = st * LVBA_NB;
        const int nbe = (int)((n - k) < LVBA_NB ? (n - k) : LVBA_NB);
        int64_t cmin = k - bw;
        if (cmin < 0) cmin = 0;
        const int64_t ncols = k - cmin;
        const unsigned nwg = (unsigned)(ncols > 0 ? (ncols + 255) / 256 : 1);
        hipLaunchKernelGGL(ldlt_back_kernel, dim3(nwg), dim3(256), 0, s, A, k, nbe, Gall + st * 4096, dvec, b, x, cmin);
    }
}

} // namespace lvba
