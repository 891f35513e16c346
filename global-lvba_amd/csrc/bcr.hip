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
#include "lvba_internal.h"

namespace lvba {

namespace {

struct BcrDev {
    int nb, k, M, Bb;      // block rows, cameras per block row, cameras, camera half-bandwidth
    double *D, *L, *T1, *T2; // [nb][BP*BP] row-major
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
    return nb * (4 * (int64_t)BP * BP + 3 * BP) + 64;
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
    p.rhs = p.T2 + m2; p.t = p.rhs + m1; p.x = p.t + m1;
    if (BP == 32) bcr_run<32>(p, Hblk, g, u_dev, x, status, s);
    else bcr_run<64>(p, Hblk, g, u_dev, x, status, s);
}

} // namespace lvba
