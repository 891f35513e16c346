// balm_kernels.hip -- gfx950 kernels for the BALM plane-eigenvalue factors.
//
// Work decomposition: the host packs consecutive voxels into CHUNKS of <= LVBA_CF factors and
// <= LVBA_CV voxels; one 256-thread workgroup (4 wavefronts) owns one chunk.  Inside a chunk
//   lane = factor   for everything per (voxel,pose) cluster (coalesced SoA loads, 8 B/lane/array),
//   lane = voxel    for the merged covariance + 3x3 eigen-decomposition (reads the transformed
//                   statistics of its factors from LDS),
//   lane = (pair, block column) for the rank-3 pose-pair blocks -Y_i Y_j^T.
// The pose-block Hessian is accumulated with hardware fp64 atomics (global_atomic_add_f64) into a
// block-band lower-triangular store: block (I,J), J <= I <= J+Bb, at ((J*(Bb+1) + I-J)*36), 6x6
// column-major inside, pose indices already in the solver's (RCM) order.
//
// Replaces VOX_HESS::evaluate_only_residual (bavoxel.hpp:176-203) and VOX_HESS::acc_evaluate2
// (bavoxel.hpp:68-174) of the reference; math in balm_math.h.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "balm_math.h"
#include "lvba_internal.h"

namespace lvba {

__device__ __forceinline__ double wave_sum(double x)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    return x;
}

// Sum over the 256 threads of a workgroup; result valid in thread 0.  red: >= 4 doubles of LDS.
__device__ __forceinline__ double block_sum_256(double x, double *red)
{
    x = wave_sum(x);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) red[wv] = x;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__device__ __forceinline__ void atomic_add_f64(double *p, double v)
{
    // lowers to global_atomic_add_f64 (no return) with -munsafe-fp-atomics
    unsafeAtomicAdd(p, v);
}

// ------------------------------------------------------------------------------------------------
// cost only: sum of lambda_min per chunk.  Algorithmic traffic 84 B/factor (80 B cluster + 4 B pose
// index) -> HBM-bound.  LDS: transformed statistics SoA T[10][CF].
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(LVBA_CF) void balm_cost_kernel(BalmDev d, const double *__restrict__ poses,
                                                           double *__restrict__ chunk_cost)
{
    __shared__ double T[10 * LVBA_CF];
    __shared__ int lvoff[LVBA_CV + 1];
    __shared__ double red[4];
    const int tid = threadIdx.x;
    const int ch = blockIdx.x;
    const int64_t v0 = d.chunk_v0[ch], v1 = d.chunk_v0[ch + 1];
    const int64_t f0 = d.voff[v0];
    const int nf = (int)(d.voff[v1] - f0), nv = (int)(v1 - v0);
    if (tid <= nv) lvoff[tid] = (int)(d.voff[v0 + tid] - f0);
    if (tid < nf) {
        const int64_t f = f0 + tid;
        double c[10], x[12], t[10];
#pragma unroll
        for (int e = 0; e < 10; ++e) c[e] = d.clu[(int64_t)e * d.F + f];
        const double *xp = poses + 12 * (int64_t)d.pidx[f];
#pragma unroll
        for (int e = 0; e < 12; ++e) x[e] = xp[e];
        transform_cluster(c, x, x + 9, t);
#pragma unroll
        for (int e = 0; e < 10; ++e) T[e * LVBA_CF + tid] = t[e];
    }
    __syncthreads();
    double lam0 = 0.0;
    if (tid < nv) {
        double S[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int f = lvoff[tid]; f < lvoff[tid + 1]; ++f) {
#pragma unroll
            for (int e = 0; e < 10; ++e) S[e] += T[e * LVBA_CF + f];
        }
        lam0 = voxel_lambda_min(S);
    }
    const double tot = block_sum_256(lam0, red);
    if (tid == 0) chunk_cost[ch] = tot;
}

// Deterministic final sum of per-chunk partials (single workgroup); out[0] = sum.
__global__ __launch_bounds__(1024) void reduce_chunks_kernel(const double *__restrict__ part, int64_t n,
                                                             double *__restrict__ out)
{
    __shared__ double red[16];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) s += part[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < 16; ++i) t += red[i];
        out[0] = t;
    }
}

// ------------------------------------------------------------------------------------------------
// full evaluation: cost + gradient + block-band Hessian.
// LDS (~52 KB -> 3 workgroups/CU): YT = union{ T[10][CF] (phases 1-2), Y[18][CF] (phases 3-4) },
// VR[13][CV] voxel records, small index arrays.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(LVBA_CF) void balm_eval_kernel(BalmDev d, const double *__restrict__ poses,
                                                           double *__restrict__ Hblk, double *__restrict__ g,
                                                           double *__restrict__ chunk_cost)
{
    __shared__ double YT[18 * LVBA_CF];
    __shared__ double VR[LVBA_VOXREC_DOUBLES * LVBA_CV];
    __shared__ int lvoff[LVBA_CV + 1];
    __shared__ int pair_off[LVBA_CV + 1];
    __shared__ int hpose[LVBA_CF];
    __shared__ unsigned char fvox[LVBA_CF];
    __shared__ double red[4];
    const int tid = threadIdx.x;
    const int ch = blockIdx.x;
    const int64_t v0 = d.chunk_v0[ch], v1 = d.chunk_v0[ch + 1];
    const int64_t f0 = d.voff[v0];
    const int nf = (int)(d.voff[v1] - f0), nv = (int)(v1 - v0);
    const int Bb1 = d.band_blocks + 1;

    if (tid <= nv) lvoff[tid] = (int)(d.voff[v0 + tid] - f0);
    // ---- phase 1: lane = factor: load, transform -------------------------------------------------
    double c[10], x[12];
    int myp = 0;
    if (tid < nf) {
        const int64_t f = f0 + tid;
#pragma unroll
        for (int e = 0; e < 10; ++e) c[e] = d.clu[(int64_t)e * d.F + f];
        myp = d.pidx[f];
        const double *xp = poses + 12 * (int64_t)myp;
#pragma unroll
        for (int e = 0; e < 12; ++e) x[e] = xp[e];
        double t[10];
        transform_cluster(c, x, x + 9, t);
#pragma unroll
        for (int e = 0; e < 10; ++e) YT[e * LVBA_CF + tid] = t[e];
        hpose[tid] = myp;
    }
    __syncthreads();
    // ---- phase 2: lane = voxel: merge, eigen-decompose, publish the voxel record ---------------
    double lam0 = 0.0;
    if (tid < nv) {
        double S[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const int b = lvoff[tid], e1 = lvoff[tid + 1];
        for (int f = b; f < e1; ++f) {
#pragma unroll
            for (int e = 0; e < 10; ++e) S[e] += YT[e * LVBA_CF + f];
            fvox[f] = (unsigned char)tid;
        }
        VoxRec vr;
        lam0 = voxel_finish(S, vr);
        VR[0 * LVBA_CV + tid] = vr.NN;
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            VR[(1 + e) * LVBA_CV + tid] = vr.vb[e];
            VR[(4 + e) * LVBA_CV + tid] = vr.u0[e];
            VR[(7 + e) * LVBA_CV + tid] = vr.s1[e];
            VR[(10 + e) * LVBA_CV + tid] = vr.s2[e];
        }
        const int k = e1 - b;
        pair_off[tid + 1] = (k * (k - 1)) / 2;
    }
    if (tid == 0) pair_off[0] = 0;
    const double tot = block_sum_256(lam0, red); // contains a __syncthreads after the LDS writes above
    if (tid == 0) chunk_cost[ch] = tot;
    __syncthreads();
    // inclusive scan of the per-voxel pair counts (nv <= 128: serial by one lane is cheap enough)
    if (tid == 0) {
        int acc = 0;
        for (int i = 1; i <= nv; ++i) { acc += pair_off[i]; pair_off[i] = acc; }
    }
    // ---- phase 3: lane = factor: Y_i, diagonal block, gradient ------------------------------------
    if (tid < nf) {
        const int vx = fvox[tid];
        VoxRec vr;
        vr.NN = VR[0 * LVBA_CV + vx];
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            vr.vb[e] = VR[(1 + e) * LVBA_CV + vx];
            vr.u0[e] = VR[(4 + e) * LVBA_CV + vx];
            vr.s1[e] = VR[(7 + e) * LVBA_CV + vx];
            vr.s2[e] = VR[(10 + e) * LVBA_CV + vx];
        }
        double Y[18], D[21], gi[6];
        factor_derivs(c, x, x + 9, vr, Y, D, gi);
#pragma unroll
        for (int e = 0; e < 18; ++e) YT[e * LVBA_CF + tid] = Y[e]; // T is dead after phase 2 (barrier above)
        double *gp = g + 6 * (int64_t)myp;
#pragma unroll
        for (int e = 0; e < 6; ++e) atomic_add_f64(gp + e, gi[e]);
        double *hp = Hblk + (int64_t)myp * Bb1 * 36; // diagonal block (I == J)
#pragma unroll
        for (int cc = 0; cc < 6; ++cc)
#pragma unroll
            for (int r = cc; r < 6; ++r) atomic_add_f64(hp + cc * 6 + r, D[dlow(r, cc)]);
    }
    __syncthreads();
    // ---- phase 4: lane = (pair, block column): -Y_I Y_J^T into the lower block (I > J) ------------
    const int npairs = pair_off[nv];
    for (int e = tid; e < npairs * 6; e += LVBA_CF) {
        const int pr = e / 6, col = e - pr * 6;
        // voxel of this pair: largest vx with pair_off[vx] <= pr
        int lo = 0, hi = nv;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (pair_off[mid] <= pr) lo = mid; else hi = mid;
        }
        const int fb = lvoff[lo], k = lvoff[lo + 1] - fb;
        int q = pr - pair_off[lo];
        // q-th pair (i<j) in row-major order over the strict upper triangle of k x k
        int i = 0, rowlen = k - 1;
        while (q >= rowlen) { q -= rowlen; ++i; --rowlen; }
        const int j = i + 1 + q;
        int fi = fb + i, fj = fb + j;
        int I = hpose[fi], J = hpose[fj];
        if (I < J) { int t = I; I = J; J = t; t = fi; fi = fj; fj = t; } // now I > J, fi <-> I
        const double b0 = YT[(0 + col) * LVBA_CF + fj], b1 = YT[(6 + col) * LVBA_CF + fj],
                     b2 = YT[(12 + col) * LVBA_CF + fj];
        double *hp = Hblk + ((int64_t)J * Bb1 + (I - J)) * 36 + col * 6;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const double val = YT[r * LVBA_CF + fi] * b0 + YT[(6 + r) * LVBA_CF + fi] * b1 +
                               YT[(12 + r) * LVBA_CF + fi] * b2;
            atomic_add_f64(hp + r, -val);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// small utility kernels
// ------------------------------------------------------------------------------------------------

// trial poses: out_j = retract(poses_j, dx_j)   (bavoxel.hpp:722-727)
__global__ void retract_kernel(const double *__restrict__ poses, const double *__restrict__ dx,
                               double *__restrict__ out, int n_poses)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_poses) return;
    double x[12], dd[6], o[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) x[e] = poses[12 * (int64_t)j + e];
#pragma unroll
    for (int e = 0; e < 6; ++e) dd[e] = dx[6 * (int64_t)j + e];
    retract_pose(x, dd, o);
#pragma unroll
    for (int e = 0; e < 12; ++e) out[12 * (int64_t)j + e] = o[e];
}

// q1 numerator: 0.5 * dx . (u * diag(H) .* dx - g)   (bavoxel.hpp:729); out[0] = value
__global__ __launch_bounds__(1024) void predicted_decrease_kernel(const double *__restrict__ Hblk, int band_blocks,
                                                                  const double *__restrict__ g,
                                                                  const double *__restrict__ dx, double u,
                                                                  int64_t n, double *__restrict__ out)
{
    __shared__ double red[16];
    double s = 0.0;
    const int64_t Bb1 = band_blocks + 1;
    for (int64_t a = threadIdx.x; a < n; a += 1024) {
        const int64_t blk = a / 6, r = a - blk * 6;
        const double dgl = Hblk[blk * Bb1 * 36 + r * 6 + r];
        s += dx[a] * (u * dgl * dx[a] - g[a]);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < 16; ++i) t += red[i];
        out[0] = 0.5 * t;
    }
}

// Export the block-band Hessian (solver pose order) as a full symmetric dense matrix in the CALLER's
// pose order: Hd[(6*pi+r) + (6*pj+c)*n].  One thread per scalar of the lower block-band.
__global__ void export_dense_kernel(const double *__restrict__ Hblk, int band_blocks, int n_poses,
                                    const int *__restrict__ perm /* internal -> caller */, double *__restrict__ Hd)
{
    const int64_t Bb1 = band_blocks + 1;
    const int64_t total = (int64_t)n_poses * Bb1 * 36;
    const int64_t n = 6 * (int64_t)n_poses;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t slot = e / 36;
        const int el = (int)(e - slot * 36);
        const int64_t J = slot / Bb1, dI = slot - J * Bb1, I = J + dI;
        if (I >= n_poses) continue;
        const int c = el / 6, r = el - c * 6;
        if (dI == 0 && r < c) continue; // diagonal block: only the lower triangle is stored
        const double val = Hblk[e];
        const int64_t rr = 6 * (int64_t)perm[I] + r, cc = 6 * (int64_t)perm[J] + c;
        Hd[rr + cc * n] = val;
        Hd[cc + rr * n] = val;
    }
}

// g (solver order) -> caller order
__global__ void export_vec_kernel(const double *__restrict__ v, const int *__restrict__ perm, int n_poses,
                                  double *__restrict__ out)
{
    const int64_t a = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (a >= 6 * (int64_t)n_poses) return;
    const int64_t blk = a / 6, r = a - blk * 6;
    out[6 * (int64_t)perm[blk] + r] = v[a];
}

// caller-order poses -> solver order
__global__ void import_poses_kernel(const double *__restrict__ in, const int *__restrict__ perm, int n_poses,
                                    double *__restrict__ out)
{
    const int64_t a = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (a >= 12 * (int64_t)n_poses) return;
    const int64_t blk = a / 12, r = a - blk * 12;
    out[a] = in[12 * (int64_t)perm[blk] + r];
}
__global__ void export_poses_kernel(const double *__restrict__ in, const int *__restrict__ perm, int n_poses,
                                    double *__restrict__ out)
{
    const int64_t a = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (a >= 12 * (int64_t)n_poses) return;
    const int64_t blk = a / 12, r = a - blk * 12;
    out[12 * (int64_t)perm[blk] + r] = in[a];
}

// ------------------------------------------------------------------------------------------------
// launchers (called from lvba_api.hip)
// ------------------------------------------------------------------------------------------------
void launch_cost(const BalmDev &d, const double *poses, double *chunk_cost, double *out, hipStream_t s,
                 hipEvent_t k0, hipEvent_t k1)
{
    if (k0) hipEventRecord(k0, s);
    hipLaunchKernelGGL(balm_cost_kernel, dim3((unsigned)d.n_chunks), dim3(LVBA_CF), 0, s, d, poses, chunk_cost);
    if (k1) hipEventRecord(k1, s);
    hipLaunchKernelGGL(reduce_chunks_kernel, dim3(1), dim3(1024), 0, s, chunk_cost, d.n_chunks, out);
}

void launch_eval(const BalmDev &d, const double *poses, double *Hblk, int64_t hblk_doubles, double *g,
                 double *chunk_cost, double *out, hipStream_t s, hipEvent_t k0, hipEvent_t k1)
{
    hipMemsetAsync(Hblk, 0, (size_t)hblk_doubles * sizeof(double), s);
    hipMemsetAsync(g, 0, (size_t)6 * d.n_poses * sizeof(double), s);
    if (k0) hipEventRecord(k0, s);
    hipLaunchKernelGGL(balm_eval_kernel, dim3((unsigned)d.n_chunks), dim3(LVBA_CF), 0, s, d, poses, Hblk, g, chunk_cost);
    if (k1) hipEventRecord(k1, s);
    hipLaunchKernelGGL(reduce_chunks_kernel, dim3(1), dim3(1024), 0, s, chunk_cost, d.n_chunks, out);
}

void launch_retract(const double *poses, const double *dx, double *out, int n_poses, hipStream_t s)
{
    hipLaunchKernelGGL(retract_kernel, dim3((n_poses + 127) / 128), dim3(128), 0, s, poses, dx, out, n_poses);
}

void launch_predicted_decrease(const double *Hblk, int band_blocks, const double *g, const double *dx, double u,
                               int64_t n, double *out, hipStream_t s)
{
    hipLaunchKernelGGL(predicted_decrease_kernel, dim3(1), dim3(1024), 0, s, Hblk, band_blocks, g, dx, u, n, out);
}

void launch_export_dense(const double *Hblk, int band_blocks, int n_poses, const int *perm, double *Hd, hipStream_t s)
{
    const int64_t n = 6 * (int64_t)n_poses;
    hipMemsetAsync(Hd, 0, (size_t)(n * n) * sizeof(double), s);
    hipLaunchKernelGGL(export_dense_kernel, dim3(2048), dim3(256), 0, s, Hblk, band_blocks, n_poses, perm, Hd);
}

void launch_export_vec(const double *v, const int *perm, int n_poses, double *out, hipStream_t s)
{
    hipLaunchKernelGGL(export_vec_kernel, dim3((6 * n_poses + 255) / 256), dim3(256), 0, s, v, perm, n_poses, out);
}

void launch_import_poses(const double *in, const int *perm, int n_poses, double *out, hipStream_t s)
{
    hipLaunchKernelGGL(import_poses_kernel, dim3((12 * n_poses + 255) / 256), dim3(256), 0, s, in, perm, n_poses, out);
}

void launch_export_poses(const double *in, const int *perm, int n_poses, double *out, hipStream_t s)
{
    hipLaunchKernelGGL(export_poses_kernel, dim3((12 * n_poses + 255) / 256), dim3(256), 0, s, in, perm, n_poses, out);
}

} // namespace lvba
