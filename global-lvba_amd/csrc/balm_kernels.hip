// balm_kernels.hip -- gfx950 kernels for the BALM plane-eigenvalue factors.
//
// Work decomposition: the host packs consecutive voxels into CHUNKS of <= LVBA_CF factors and
// <= LVBA_CV voxels; one 256-thread workgroup (4 wavefronts) owns one chunk.  Inside a chunk
//   lane = factor   for everything per (voxel,pose) cluster (coalesced SoA loads, 8 B/lane/array),
//   lane = voxel    for the merged covariance + 3x3 eigen-decomposition (reads the transformed
//                   statistics of its factors from LDS).
// The pose-block Hessian lives in a block-band lower-triangular store: block (I,J), J <= I <= J+Bb, at
// ((J*(Bb+1) + I-J)*36), 6x6 column-major inside, pose indices already in the solver's (RCM) order.
// It is assembled WITHOUT atomics (fp64 global atomics measured 18 G/s on MI355X: 12 ms for C2), in
// three deterministic passes over data laid out for each:
//   balm_voxel_kernel   voxel-major  : merged covariance, eigen-decomposition -> cost + voxel records
//   balm_factor_kernel  pose-major   : per factor Y_i (stored), diagonal block E_i - Y_i Y_i^T and gradient
//                                      reduced in registers over the pose's factors (one plain store)
//   balm_pair_*_kernel  block-major  : every off-diagonal block -sum Y_I Y_J^T over its voxel list (column-per-lane form for
//                                      large windowed problems, 16 lanes per item otherwise), plain stores
// so run-to-run results are bitwise identical.
//
// Replaces VOX_HESS::evaluate_only_residual (bavoxel.hpp:176-203) and VOX_HESS::acc_evaluate2
// (bavoxel.hpp:68-174) of the reference; math in balm_math.h.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
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

// A voxel observed by more poses than a workgroup has lanes sits alone in its chunk and is merged in tiles of LVBA_CF
// factors: per-lane partial sums of the transformed statistics, then lane e < 10 adds up column e of the LDS table in lane
// order (deterministic).  Returns the merged statistics in S on every lane.
__device__ __forceinline__ void merge_big_voxel(const BalmDev &d, const double *__restrict__ poses, int64_t f0, int nf,
                                                double *T, double *S)
{
    const int tid = threadIdx.x;
    double p[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int t0 = 0; t0 < nf; t0 += LVBA_CF) {
        if (t0 + tid < nf) {
            const int64_t f = f0 + t0 + tid;
            double c[10], x[12], t[10];
#pragma unroll
            for (int e = 0; e < 10; ++e) c[e] = d.clu[(int64_t)e * d.F + f];
            // the pose is a gather (one 96-byte record per lane): six 16-byte loads, not twelve 8-byte ones
            const double2 *xp = reinterpret_cast<const double2 *>(poses + 12 * (int64_t)d.pidx[f]);
#pragma unroll
            for (int e = 0; e < 6; ++e) {
                const double2 v2 = xp[e];
                x[2 * e] = v2.x; x[2 * e + 1] = v2.y;
            }
            transform_cluster(c, x, x + 9, t);
#pragma unroll
            for (int e = 0; e < 10; ++e) p[e] += t[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 10; ++e) T[e * LVBA_CF + tid] = p[e];
    __syncthreads();
    if (tid < 10) {
        double s = 0.0;
        for (int j = 0; j < LVBA_CF; ++j) s += T[tid * LVBA_CF + j];
        T[tid * LVBA_CF] = s;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 10; ++e) S[e] = T[e * LVBA_CF];
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
    if (nf > LVBA_CF) { // one voxel with more observers than lanes (uniform branch)
        double S[10];
        merge_big_voxel(d, poses, f0, nf, T, S);
        if (tid == 0) chunk_cost[ch] = voxel_lambda_min(S);
        return;
    }
    if (tid <= nv) lvoff[tid] = (int)(d.voff[v0 + tid] - f0);
    if (tid < nf) {
        const int64_t f = f0 + tid;
        double c[10], x[12], t[10];
#pragma unroll
        for (int e = 0; e < 10; ++e) c[e] = d.clu[(int64_t)e * d.F + f];
        // the pose is a gather (one 96-byte record per lane): six 16-byte loads, not twelve 8-byte ones
        const double2 *xp = reinterpret_cast<const double2 *>(poses + 12 * (int64_t)d.pidx[f]);
#pragma unroll
        for (int e = 0; e < 6; ++e) {
            const double2 v2 = xp[e];
            x[2 * e] = v2.x; x[2 * e + 1] = v2.y;
        }
        transform_cluster(c, x, x + 9, t);
#pragma unroll
        for (int e = 0; e < 10; ++e) T[e * LVBA_CF + tid] = t[e];
    }
    __syncthreads();
    double lam0 = 0.0;
    // voxel of this thread: the wavefront that takes the chunk's eigen-decompositions (a serial stretch of one or two
    // wavefronts) rotates with the chunk index, so that the workgroups a CU holds do not pile it on one SIMD
    const int vt = (tid + 256 - 64 * (ch & 3)) & 255;
    if (vt < nv) {
        double S[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int f = lvoff[vt]; f < lvoff[vt + 1]; ++f) {
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
    // 32-byte loads, four of them in flight per thread: one workgroup walking 40 k partials eight bytes at a time, one dependent
    // load after the other, took 18 us -- twice per LM iteration, between two kernels that wait for it
    const double4 *p4 = reinterpret_cast<const double4 *>(part);
    const int64_t n4 = n / 4;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll 4
    for (int64_t i = threadIdx.x; i < n4; i += 1024) {
        const double4 v = p4[i];
        s0 += v.x; s1 += v.y; s2 += v.z; s3 += v.w;
    }
    for (int64_t i = 4 * n4 + threadIdx.x; i < n; i += 1024) s0 += part[i];
    double s = (s0 + s1) + (s2 + s3);
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
// full evaluation, pass 1 (voxel-major): cost + per-voxel records.  Same loads as balm_cost_kernel.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(LVBA_CF, 6) void balm_voxel_kernel(BalmDev d, const double *__restrict__ poses,
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
    if (nf > LVBA_CF) { // one voxel with more observers than lanes (uniform branch)
        double S[10];
        merge_big_voxel(d, poses, f0, nf, T, S);
        if (tid == 0) {
            VoxRec vr;
            chunk_cost[ch] = voxel_finish(S, vr);
            double *o = d.vrec + 16 * v0;
            o[0] = vr.NN;
            for (int e = 0; e < 3; ++e) {
                o[1 + e] = vr.vb[e];
                o[4 + e] = vr.u0[e];
                o[7 + e] = vr.s1[e];
                o[10 + e] = vr.s2[e];
            }
        }
        return;
    }
    if (tid <= nv) lvoff[tid] = (int)(d.voff[v0 + tid] - f0);
    if (tid < nf) {
        const int64_t f = f0 + tid;
        double c[10], x[12], t[10];
#pragma unroll
        for (int e = 0; e < 10; ++e) c[e] = d.clu[(int64_t)e * d.F + f];
        // the pose is a gather (one 96-byte record per lane): six 16-byte loads, not twelve 8-byte ones
        const double2 *xp = reinterpret_cast<const double2 *>(poses + 12 * (int64_t)d.pidx[f]);
#pragma unroll
        for (int e = 0; e < 6; ++e) {
            const double2 v2 = xp[e];
            x[2 * e] = v2.x; x[2 * e + 1] = v2.y;
        }
        transform_cluster(c, x, x + 9, t);
#pragma unroll
        for (int e = 0; e < 10; ++e) T[e * LVBA_CF + tid] = t[e];
    }
    __syncthreads();
    double lam0 = 0.0;
    VoxRec vr;
    const int vt = (tid + 256 - 64 * (ch & 3)) & 255; // as in balm_cost_kernel (same voxel -> lane map: the same cost, bit for bit)
    if (vt < nv) {
        double S[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int f = lvoff[vt]; f < lvoff[vt + 1]; ++f) {
#pragma unroll
            for (int e = 0; e < 10; ++e) S[e] += T[e * LVBA_CF + f];
        }
        lam0 = voxel_finish(S, vr);
    }
    // the records leave through LDS (T is dead once every lane has its sums): lane-per-record stores would put 8 bytes into
    // each of up to 128 cache lines per instruction; staged, the chunk's records go out as contiguous 2-KB stores
    __syncthreads();
    if (vt < nv) {
        double *o = T + 17 * vt;
        o[0] = vr.NN;
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            o[1 + e] = vr.vb[e];
            o[4 + e] = vr.u0[e];
            o[7 + e] = vr.s1[e];
            o[10 + e] = vr.s2[e];
        }
        o[13] = o[14] = o[15] = 0.0;
    }
    __syncthreads();
    {
        double *o = d.vrec + 16 * v0;
        for (int f = tid; f < 16 * nv; f += LVBA_CF) o[f] = T[17 * (f >> 4) + (f & 15)];
    }
    const double tot = block_sum_256(lam0, red);
    if (tid == 0) chunk_cost[ch] = tot;
}

// ------------------------------------------------------------------------------------------------
// pass 2 (pose-major): workgroup (I, s) handles the s-th slice of pose I's factor segment.  The pose is
// uniform per workgroup; cluster loads are coalesced (pose-major copy); the voxel record is one 128-byte
// line per lane.  Y_i goes to global memory for pass 3; (D, g) are summed in registers.
// ------------------------------------------------------------------------------------------------
// Y32 (LVBA_Y32=1, an experiment of round 4): the Y records are stored as fp32, 20 floats = 80 bytes per factor (18 used) instead
// of 144 -- half of the write traffic here and of the pair pass's gathers; the diagonal blocks and the gradient are formed from
// the fp64 values in registers either way, the off-diagonal blocks then carry ~1e-7 relative rounding (fp64 accumulation).
template <bool Y32>
__global__ __launch_bounds__(256) void balm_factor_kernel(BalmDev d, const double *__restrict__ poses)
{
    __shared__ double red[4 * 27];
    __shared__ double Ys[4 * 64 * 19];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // (An XCD-aware order -- every XCD a contiguous eighth of the (pose, slice) list, so that neighbouring poses' gathers of
    // the same voxel records meet in one L2 -- was measured in round 3: FETCH_SIZE unchanged at 2.1 GB, 0.63 vs 0.61 ms.  A
    // voxel's ~5 observers are spread over ~100 poses; the ~16 poses an XCD has in flight rarely hold two of them.)
    const int I = blockIdx.x / d.S, s = blockIdx.x - I * d.S;
    const int64_t seg0 = d.csc_off[I], len = d.csc_off[I + 1] - seg0;
    const int64_t a = seg0 + (len * s) / d.S, b = seg0 + (len * (s + 1)) / d.S;
    double x[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) x[e] = poses[12 * (int64_t)I + e];
    double acc[27];
#pragma unroll
    for (int e = 0; e < 27; ++e) acc[e] = 0.0;
    for (int64_t t0 = a + 64 * wv; t0 < b; t0 += 256) { // wavefront-uniform trip count: every lane takes part in the stores
        const int64_t t = t0 + lane;
        double Y[18];
        if (t < b) {
            double c[10];
#pragma unroll
            for (int e = 0; e < 10; ++e) c[e] = d.clu_csc[(int64_t)e * d.F + t];
            const double2 *vq = reinterpret_cast<const double2 *>(d.vrec + 16 * (int64_t)d.vox_of_pos[t]); // one 128-B line
            double vp[14];
#pragma unroll
            for (int e = 0; e < 7; ++e) {
                const double2 v2 = vq[e];
                vp[2 * e] = v2.x; vp[2 * e + 1] = v2.y;
            }
            VoxRec vr;
            vr.NN = vp[0];
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                vr.vb[e] = vp[1 + e];
                vr.u0[e] = vp[4 + e];
                vr.s1[e] = vp[7 + e];
                vr.s2[e] = vp[10 + e];
            }
            double D[21], gi[6];
            factor_derivs(c, x, x + 9, vr, Y, D, gi);
#pragma unroll
            for (int e = 0; e < 21; ++e) acc[e] += D[e];
#pragma unroll
            for (int e = 0; e < 6; ++e) acc[21 + e] += gi[e];
        }
        // Y records leave through LDS: a lane-per-record store puts 16 bytes into each of 72 cache lines per instruction
        // (stride 144 B); transposed, the wavefront's 64 records go out as 9 contiguous 1-KB stores -- NON-TEMPORAL ones: 1.44 GB
        // written once and read again only by the next kernel do not belong in L2 (round 6, same box, A / B twice: factor + pair
        // passes 1.481 / 1.492 -> 1.455 / 1.461 ms; most of it in the pair pass, which no longer starts behind the write-back)
        double *ys = Ys + wv * (64 * 19);
        if (t < b) {
#pragma unroll
            for (int e = 0; e < 18; ++e) ys[lane * 19 + e] = Y[e];
        }
        const int nrec = (int)((b - t0) < 64 ? (b - t0) : 64);
        if constexpr (Y32) { // 64 records x 5 chunks of four floats: five contiguous 1-KB stores
            float4 *yo = reinterpret_cast<float4 *>(reinterpret_cast<float *>(d.Y) + 20 * t0);
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                const int ch = lane + 64 * r, rec = ch / 5, el = 4 * (ch - 5 * rec);
                if (rec < nrec) {
                    const double *yr = ys + rec * 19 + el;
                    typedef float f4v __attribute__((ext_vector_type(4)));
                    const f4v vv = {(float)yr[0], (float)yr[1], el + 2 < 18 ? (float)yr[2] : 0.f, el + 3 < 18 ? (float)yr[3] : 0.f};
                    __builtin_nontemporal_store(vv, reinterpret_cast<f4v *>(yo) + ch);
                }
            }
        } else {
            double2 *yo = reinterpret_cast<double2 *>(d.Y + 18 * t0);
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                const int f = 2 * (lane + 64 * r); // flat double index inside the batch
                const int rec = f / 18, el = f - 18 * rec;
                if (rec < nrec) {
                    typedef double d2v __attribute__((ext_vector_type(2)));
                    const d2v vv = {ys[rec * 19 + el], ys[rec * 19 + el + 1]};
                    __builtin_nontemporal_store(vv, reinterpret_cast<d2v *>(yo) + lane + 64 * r);
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 27; ++e) {
        const double v = wave_sum(acc[e]);
        if (lane == 0) red[wv * 27 + e] = v;
    }
    __syncthreads();
    if (tid < 27) d.part[(int64_t)blockIdx.x * 32 + tid] = red[tid] + red[27 + tid] + red[54 + tid] + red[81 + tid];
}

// sum the S slice partials of every pose -> diagonal block (lower triangle) and gradient
__global__ void balm_diag_reduce_kernel(BalmDev d, double *__restrict__ Hblk, double *__restrict__ g)
{
    const int64_t gid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t I = gid >> 5;
    const int e = (int)(gid & 31);
    if (I >= d.n_poses || e >= 27) return;
    double s = 0.0;
    for (int q = 0; q < d.S; ++q) s += d.part[(I * d.S + q) * 32 + e];
    if (e >= 21) {
        g[6 * I + (e - 21)] = s;
    } else {
        // packed lower index e -> (r, c): columns hold 6,5,4,3,2,1 entries
        int c = 0, base = 0;
        while (e >= base + (6 - c)) { base += 6 - c; ++c; }
        const int r = c + (e - base);
        Hblk[I * (int64_t)(d.band_blocks + 1) * 36 + c * 6 + r] = s;
    }
}

// sum over the 16 lanes of a DPP row; every lane of the row gets the total
__device__ __forceinline__ double row16_sum(double x)
{
#define LVBA_ROR_ADD(n)                                                                                \
    do {                                                                                               \
        const int lo_ = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x120 + (n), 0xF, 0xF, false); \
        const int hi_ = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x120 + (n), 0xF, 0xF, false); \
        x += __hiloint2double(hi_, lo_);                                                               \
    } while (0)
    LVBA_ROR_ADD(1);
    LVBA_ROR_ADD(2);
    LVBA_ROR_ADD(4);
    LVBA_ROR_ADD(8);
#undef LVBA_ROR_ADD
    return x;
}

// ------------------------------------------------------------------------------------------------
// pass 3 (block-major), 16-lane form: 16 lanes (one DPP row) own one off-diagonal block (or one item of a long list); each lane
// takes one contributing voxel (a pair of Y records) per iteration and forms the 6 x 6 rank-3 product in registers.  The records
// are fetched COOPERATIVELY: if every lane gathered its own two 144-byte records, each of the 18 load instructions of an
// iteration would touch 64+ different cache lines per wavefront (the texture-address path serialises on lines, not bytes).
// Here the 16 lanes of a group fetch the group's 32 records of an iteration as 288 consecutive 16-byte chunks (a lane run of 9
// covers one record), x side then y side, park them in LDS and each lane reads its own records back.  The default for problems
// with few blocks and long lists (window BA); large problems use the column-per-lane form below.
// ------------------------------------------------------------------------------------------------
#define LVBA_PAIR_GROUP_DOUBLES (16 * 18)
__global__ __launch_bounds__(256) void balm_pair_staged_kernel(PairDev d, double *__restrict__ Hblk)
{
    __shared__ double stage[16 * LVBA_PAIR_GROUP_DOUBLES]; // 36.9 KB: 16 groups x 16 records x 18 doubles (x side, then y side)
    __shared__ int2 prs[16 * 16];
    const int64_t per_xcd = (gridDim.x + 7) / 8;
    const int64_t wg = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    const int grp = threadIdx.x >> 4, l16 = threadIdx.x & 15;
    const int64_t blk = wg * 16 + grp;
    const bool live = blk < d.nnzb;
    int64_t o0 = 0, o1 = 0;
    if (live) { o0 = d.blk_off[blk]; o1 = d.blk_off[blk + 1]; }
    double *st = stage + grp * LVBA_PAIR_GROUP_DOUBLES;
    int2 *pg = prs + grp * 16;
    double acc[36];
#pragma unroll
    for (int e = 0; e < 36; ++e) acc[e] = 0.0;
    for (int64_t q = o0; q < o1; q += 16) {
        const int64_t qi = q + l16;
        const int npair = (int)((o1 - q) < 16 ? (o1 - q) : 16);
        pg[l16] = (qi < o1) ? d.pairs[qi] : make_int2(0, 0);
        double2 vx[9], vy[9];
#pragma unroll
        for (int s2 = 0; s2 < 9; ++s2) {
            const int ci = s2 * 16 + l16;       // chunk of the group's 16 x 9
            const int rc = ci / 9, cc = ci - 9 * rc;
            const int2 pp = pg[rc];
            const bool ok = rc < npair;
            vx[s2] = ok ? reinterpret_cast<const double2 *>(d.Y + 18 * (int64_t)pp.x)[cc] : make_double2(0.0, 0.0);
            vy[s2] = ok ? reinterpret_cast<const double2 *>(d.Y + 18 * (int64_t)pp.y)[cc] : make_double2(0.0, 0.0);
        }
        double Yi[18], Yj[18];
#pragma unroll
        for (int s2 = 0; s2 < 9; ++s2) reinterpret_cast<double2 *>(st)[s2 * 16 + l16] = vx[s2]; // record rc at st + 18 rc
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int e = 0; e < 9; ++e) {
            const double2 a = reinterpret_cast<const double2 *>(st + 18 * l16)[e];
            Yi[2 * e] = a.x; Yi[2 * e + 1] = a.y;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s2 = 0; s2 < 9; ++s2) reinterpret_cast<double2 *>(st)[s2 * 16 + l16] = vy[s2];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int e = 0; e < 9; ++e) {
            const double2 a = reinterpret_cast<const double2 *>(st + 18 * l16)[e];
            Yj[2 * e] = a.x; Yj[2 * e + 1] = a.y;
        }
        __builtin_amdgcn_wave_barrier();
        if (qi < o1) {
#pragma unroll
            for (int c = 0; c < 6; ++c)
#pragma unroll
                for (int r = 0; r < 6; ++r)
                    acc[c * 6 + r] += Yi[r] * Yj[c] + Yi[6 + r] * Yj[6 + c] + Yi[12 + r] * Yj[12 + c];
        }
    }
#pragma unroll
    for (int e = 0; e < 36; ++e) acc[e] = row16_sum(acc[e]);
    if (live && l16 == 0) {
        const int64_t dst = d.blk_slot[blk];
        double2 *hp = reinterpret_cast<double2 *>(dst >= 0 ? Hblk + dst * 36 : d.partial + (-dst - 1) * 36);
#pragma unroll
        for (int e = 0; e < 18; ++e) hp[e] = make_double2(-acc[2 * e], -acc[2 * e + 1]);
    }
}

// ------------------------------------------------------------------------------------------------
// pass 3, column-per-lane form (the default).  The pair lists are grouped by (voxel window, block): the pairs the chip is
// working on at any moment then draw on one window's Y records -- an L2-sized slice of every pose's segment -- instead of
// the whole 1.4 GB array (PMC at C3: L2 misses 64 M -> 29 M per pass).  That makes the items short (C3: ~9 pairs per
// (window, block)), and the 16-lane forms above then spend their time on per-item overhead (36 DPP row reductions and a
// half-empty gather round per item).  Here SIX LANES own one item, lane c holding column c of the 6 x 6 block: per pair a
// lane needs Y_I (18 doubles, the same for the six lanes: an LDS broadcast) and the three entries Y_J[c], Y_J[6+c],
// Y_J[12+c], and does 18 FMAs -- no cross-lane reduction at all, 80 VGPRs, 6 waves per SIMD to hide the gathers.  A
// wavefront walks 10 items in lock-step, one pair of each per round: the 20 records of a round are fetched cooperatively
// (180 consecutive 16-byte chunks, a lane run of nine covers one 144-byte record), parked in LDS, and the next round's pair
// indices are already on their way.  Every lane ends with one 48-byte store; a group's 288 bytes are contiguous (a block
// with a single item goes straight into the store, the others into partial blocks in item order).
// ------------------------------------------------------------------------------------------------
#define LVBA_PC_ITEMS 10 // items per wavefront (6 lanes each; lanes 60..63 only help fetching)
#ifndef LVBA_PC_DEPTH
#define LVBA_PC_DEPTH 1  // pairs of every item per round.  With the next round's gathers in flight during a round (below) one pair
                         // per round is best: 90 VGPRs, five wavefronts per SIMD (C3, factor + pair passes: 1.69 ms; two pairs
                         // 1.75, three 1.79, four 1.94)
#endif
#ifndef LVBA_PC_PF
#define LVBA_PC_PF 1     // rounds whose gathers are in flight ahead of the one being multiplied (1 or 2)
#endif
// Y32: fp32 records of 80 bytes (balm_factor_kernel<true>): five 16-byte chunks per record instead of nine
template <bool Y32>
__global__ __launch_bounds__(256) void balm_pair_col_kernel(PairDev d, double *__restrict__ Hblk)
{
    constexpr int RCH = Y32 ? 5 : 9;                               // 16-byte chunks per record
    constexpr int NCH = LVBA_PC_ITEMS * LVBA_PC_DEPTH * 2 * RCH;   // 16-byte chunks per round
    constexpr int NLD = (NCH + 63) / 64;                    // loads per lane per round
    __shared__ double2 recs[4][NCH]; // per wavefront: DEPTH x 10 pairs x (x record, y record) x 9 chunks of 16 bytes
    __shared__ int2 plist[4][LVBA_PC_ITEMS * LVBA_PAIR_CUT]; // the pair indices of the wavefront's items, fetched once
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t per_xcd = (gridDim.x + 7) / 8; // XCD x sweeps a contiguous eighth of the items = a range of voxel windows
    const int64_t wg = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    const int64_t i0 = (wg * 4 + wv) * LVBA_PC_ITEMS;
    if (i0 >= d.nnzb) return; // wavefront-uniform; no workgroup barrier below
    const int g = lane / 6, cc = lane - 6 * g; // item of the wavefront, column of its block (g = 10: idle lanes 60..63)
    // A wavefront lives for ~ ten rounds: what happens before the first one counts.  The ranges of its items come in ONE load (lane l
    // takes blk_off[i0 + l], the others learn what they need through shuffles), the pair lists in one batch of loads -- as a
    // loop of load / wait / store per 64 pairs and with the ranges fetched twice, a wavefront spent four to five memory round trips
    // one after the other before its first gather left (a quarter of its life).
    const int nit = (int)((i0 + LVBA_PC_ITEMS < d.nnzb ? i0 + LVBA_PC_ITEMS : d.nnzb) - i0); // items of this wavefront, 1 .. 10
    long long bo = 0;
    if (lane <= LVBA_PC_ITEMS) bo = d.blk_off[i0 + (lane < nit ? lane : nit)];
    const int64_t q0 = __shfl(bo, 0, 64), q1 = __shfl(bo, nit, 64);
    const long long bo_next = __shfl_down(bo, 1, 64);
    // the items' pair lists are one contiguous range of the sorted pair array: coalesced copy into LDS
    int2 *pl = plist[wv];
    {
        constexpr int NPL = (LVBA_PC_ITEMS * LVBA_PAIR_CUT + 63) / 64;
        int2 pv[NPL];
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            const int64_t q = q0 + lane + 64 * k;
            pv[k] = d.pairs[q < q1 ? q : q1 - 1]; // (unconditional: every list has a pair)
        }
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            const int64_t q = q0 + lane + 64 * k;
            if (q < q1) pl[q - q0] = pv[k];
        }
    }
    // lane l < 10 keeps the range of item l (relative to q0); everybody learns the ranges it needs through shuffles
    const int fa = lane < nit ? (int)(bo - q0) : 0, flen = lane < nit ? (int)(bo_next - bo) : 0;
    int rounds = flen;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) { const int t = __shfl_xor(rounds, o, 16); rounds = t > rounds ? t : rounds; }
    rounds = (__shfl(rounds, 0, 64) + LVBA_PC_DEPTH - 1) / LVBA_PC_DEPTH; // max over lanes 0..15
    const int mylen = __shfl(flen, g < LVBA_PC_ITEMS ? g : 0, 64);
    const bool owner = g < LVBA_PC_ITEMS && i0 + g < d.nnzb;
    // per-lane constants of the cooperative fetch: chunk c = lane + 64 s of a round is piece (c % 9) of record (c / 9);
    // record rc belongs to slot rc >> 1 = depth * 10 + item, side rc & 1
    int c_piece[NLD], c_side[NLD], c_dep[NLD], c_fa[NLD], c_len[NLD];
#pragma unroll
    for (int s2 = 0; s2 < NLD; ++s2) {
        const int c = lane + 64 * s2, rc = c / RCH, slot = rc >> 1;
        const int dep = slot / LVBA_PC_ITEMS, gi = slot - dep * LVBA_PC_ITEMS;
        c_piece[s2] = c - RCH * rc; c_side[s2] = rc & 1; c_dep[s2] = dep;
        c_fa[s2] = __shfl(fa, gi, 64);
        c_len[s2] = (c < NCH) ? __shfl(flen, gi, 64) : 0;
    }
    __builtin_amdgcn_wave_barrier(); // pl is complete
    double2 *rw = recs[wv];
    double acc[6];
#pragma unroll
    for (int e = 0; e < 6; ++e) acc[e] = 0.0;
    // Software pipeline: the records of round r + LVBA_PC_PF are requested as soon as round r's have been parked in LDS, so their
    // way through L2 / HBM overlaps the LDS reads and FMAs of the rounds before (round 2: gather -> LDS -> FMA, one dependent
    // chain per round).
    double2 v[LVBA_PC_PF][NLD];
    // (buffer addressing for these gathers -- 32-bit offsets instead of 64-bit flat addresses -- was measured in round 3: no change)
    auto fetch = [&](int r, double2 (&vv)[NLD]) {
        // the pair indices of the round's chunks: unconditional LDS reads (a clamped pair of the same item), all of them ahead of the
        // gathers -- read inside the condition, each gather waited for its own LDS read in turn
        int2 pr[NLD];
#pragma unroll
        for (int s2 = 0; s2 < NLD; ++s2) {
            const int pi = LVBA_PC_DEPTH * r + c_dep[s2]; // this chunk's pair of its item
            const int lm1 = c_len[s2] > 0 ? c_len[s2] - 1 : 0;
            pr[s2] = pl[c_fa[s2] + (pi < lm1 ? pi : lm1)];
        }
#pragma unroll
        for (int s2 = 0; s2 < NLD; ++s2) {
            const int pi = LVBA_PC_DEPTH * r + c_dep[s2];
            vv[s2] = make_double2(0.0, 0.0);
            if (pi < c_len[s2])
                vv[s2] = reinterpret_cast<const double2 *>(d.Y + (Y32 ? 10 : 18) * (int64_t)(c_side[s2] ? pr[s2].y : pr[s2].x))[c_piece[s2]];
        }
    };
    auto round_body = [&](int r, double2 (&vv)[NLD]) { // vv holds round r; it is refilled with round r + LVBA_PC_PF
#pragma unroll
        for (int s2 = 0; s2 < NLD; ++s2) {
            const int c = lane + 64 * s2;
            if (c < NCH) rw[c] = vv[s2];
        }
        __builtin_amdgcn_wave_barrier();
        if (r + LVBA_PC_PF < rounds) fetch(r + LVBA_PC_PF, vv);
#pragma unroll
        for (int dep = 0; dep < LVBA_PC_DEPTH; ++dep) {
            if (owner && LVBA_PC_DEPTH * r + dep < mylen) {
                const double2 *ri = rw + 2 * RCH * (dep * LVBA_PC_ITEMS + g);
                double Yi[18], j0, j1, j2;
                if constexpr (Y32) {
                    const float4 *rf = reinterpret_cast<const float4 *>(ri);
                    const float *yj = reinterpret_cast<const float *>(ri + RCH);
#pragma unroll
                    for (int e = 0; e < 5; ++e) {
                        const float4 x = rf[e];
                        Yi[4 * e] = x.x; Yi[4 * e + 1] = x.y;
                        if (4 * e + 2 < 18) { Yi[4 * e + 2] = x.z; Yi[4 * e + 3] = x.w; }
                    }
                    j0 = yj[cc]; j1 = yj[6 + cc]; j2 = yj[12 + cc];
                } else {
                    const double *yj = reinterpret_cast<const double *>(ri + RCH);
#pragma unroll
                    for (int e = 0; e < 9; ++e) {
                        const double2 x = ri[e];
                        Yi[2 * e] = x.x; Yi[2 * e + 1] = x.y;
                    }
                    j0 = yj[cc]; j1 = yj[6 + cc]; j2 = yj[12 + cc];
                }
#pragma unroll
                for (int e = 0; e < 6; ++e) acc[e] = fma(Yi[12 + e], j2, fma(Yi[6 + e], j1, fma(Yi[e], j0, acc[e])));
            }
        }
        __builtin_amdgcn_wave_barrier();
    };
#pragma unroll
    for (int k = 0; k < LVBA_PC_PF; ++k)
        if (k < rounds) fetch(k, v[k]);
    for (int r = 0; r < rounds; r += LVBA_PC_PF) {
#pragma unroll
        for (int k = 0; k < LVBA_PC_PF; ++k)
            if (r + k < rounds) round_body(r + k, v[k]);
    }
    if (owner) {
        const int64_t dst = d.blk_slot[i0 + g];
        double2 *hp = reinterpret_cast<double2 *>((dst >= 0 ? Hblk + dst * 36 : d.partial + (-dst - 1) * 36) + 6 * cc);
#pragma unroll
        for (int e = 0; e < 3; ++e) hp[e] = make_double2(-acc[2 * e], -acc[2 * e + 1]);
    }
}

// blocks assembled from several work items (one per voxel window, and lists cut at LVBA_PAIR_CUT pairs): the partial blocks
// are added up in item order -- still no atomics, still bitwise reproducible
__global__ void balm_pair_reduce_kernel(PairDev d, double *__restrict__ Hblk)
{
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t m = t / 18;
    const int e = (int)(t - 18 * m); // 16-byte piece of the 288-byte block
    if (m >= d.n_multi) return;
    const int64_t q0 = d.multi_off[m], q1 = d.multi_off[m + 1];
    double2 s = make_double2(0.0, 0.0);
    int64_t q = q0;
    for (; q + 4 <= q1; q += 4) { // four independent loads in flight; summed in list order
        const double2 a0 = reinterpret_cast<const double2 *>(d.partial + 36 * d.multi_idx[q])[e];
        const double2 a1 = reinterpret_cast<const double2 *>(d.partial + 36 * d.multi_idx[q + 1])[e];
        const double2 a2 = reinterpret_cast<const double2 *>(d.partial + 36 * d.multi_idx[q + 2])[e];
        const double2 a3 = reinterpret_cast<const double2 *>(d.partial + 36 * d.multi_idx[q + 3])[e];
        s.x = (((s.x + a0.x) + a1.x) + a2.x) + a3.x;
        s.y = (((s.y + a0.y) + a1.y) + a2.y) + a3.y;
    }
    for (; q < q1; ++q) {
        const double2 a0 = reinterpret_cast<const double2 *>(d.partial + 36 * d.multi_idx[q])[e];
        s.x += a0.x; s.y += a0.y;
    }
    reinterpret_cast<double2 *>(Hblk + d.multi_slot[m] * 36)[e] = s;
}

// pose-major copy of the cluster statistics (one-off, at finalize)
__global__ void gather_csc_kernel(const double *__restrict__ clu, const int32_t *__restrict__ csc_f, int64_t F,
                                  double *__restrict__ clu_csc)
{
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= F) return;
    const int64_t f = csc_f[t];
#pragma unroll
    for (int e = 0; e < 10; ++e) clu_csc[(int64_t)e * F + t] = clu[(int64_t)e * F + f];
}

// caller layout [F][10] -> device layout [10][F]
// fmap != nullptr: factor f of the device layout is factor fmap[f] of the caller's (voxels reordered at create time)
__global__ void aos_to_soa_kernel(const double *__restrict__ aos, const int32_t *__restrict__ fmap, int64_t F,
                                  double *__restrict__ soa)
{
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= 10 * F) return;
    const int64_t f = t / 10;
    const int e = (int)(t - 10 * f);
    soa[(int64_t)e * F + f] = fmap ? aos[10 * (int64_t)fmap[f] + e] : aos[t];
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

// The LM loop's form of the two kernels around this comment: the retraction and, from the same dx, each workgroup's share of the q1
// numerator 0.5 dx . (u diag(H) .* dx - g) (bavoxel.hpp:722-729), summed in workgroup order by lm_report_kernel.  As a kernel of
// its own (one workgroup of 1024 lanes walking 6 N entries) the numerator took 18 us between the solve and the cost pass.
__global__ __launch_bounds__(128) void retract_q1_kernel(const double *__restrict__ poses, const double *__restrict__ dx, double *__restrict__ out,
                                                         int n_poses, const double *__restrict__ Hblk, int band_blocks,
                                                         const double *__restrict__ g, double u, double *__restrict__ q1_part)
{
    __shared__ double red[2];
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    double s = 0.0;
    if (j < n_poses) {
        double x[12], dd[6], o[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) x[e] = poses[12 * (int64_t)j + e];
#pragma unroll
        for (int e = 0; e < 6; ++e) dd[e] = dx[6 * (int64_t)j + e];
        retract_pose(x, dd, o);
#pragma unroll
        for (int e = 0; e < 12; ++e) out[12 * (int64_t)j + e] = o[e];
        const double *hd = Hblk + (int64_t)j * (band_blocks + 1) * 36;
#pragma unroll
        for (int r = 0; r < 6; ++r) s += dd[r] * (u * hd[r * 6 + r] * dd[r] - g[6 * (int64_t)j + r]);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) q1_part[blockIdx.x] = red[0] + red[1];
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
#pragma unroll 4
    for (int64_t a = threadIdx.x; a < n; a += 1024) { // (unrolled: the three loads of four rounds in flight together)
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

// The five numbers the LM driver reads after an iteration, written straight into pinned host memory (zero-copy): trial cost,
// q1 numerator, cost at the current poses, pivot status -- one tiny kernel instead of three device-to-host copies at the tail
// of every iteration.
__global__ void lm_report_kernel(const double *__restrict__ scal2, const double *__restrict__ q1_part, int n_q1,
                                 const double *__restrict__ cost_cur, const int *__restrict__ status, double *__restrict__ host_pin)
{
    if (threadIdx.x == 0) {
        host_pin[0] = scal2[0];
        double q1 = 0.0; // the q1 numerator from retract_q1_kernel's per-workgroup shares, in workgroup order
        for (int e = 0; e < n_q1; ++e) q1 += q1_part[e];
        host_pin[1] = 0.5 * q1;
        host_pin[2] = cost_cur[0];
        host_pin[4] = __longlong_as_double((long long)(unsigned)status[0]);
        __threadfence_system();
    }
}

// ---- grouped refinement (lvba_balm_refine_groups): independent pose / voxel groups advance through one LM loop in lock-step,
// each with its own cost, damping and accept / reject decision.  One workgroup per group, fixed summation order.
__global__ __launch_bounds__(256) void reduce_chunks_groups_kernel(const double *__restrict__ part, const int64_t *__restrict__ gco,
                                                                  double *__restrict__ out)
{
    __shared__ double red[4];
    const int64_t c0 = gco[blockIdx.x], c1 = gco[blockIdx.x + 1];
    double s = 0.0;
    for (int64_t i = c0 + threadIdx.x; i < c1; i += 256) s += part[i];
    const double t = block_sum_256(s, red);
    if (threadIdx.x == 0) out[blockIdx.x] = t;
}
// per group: 0.5 * dx . (u_g * diag(H) .* dx - g) over the group's poses (bavoxel.hpp:729)
__global__ __launch_bounds__(256) void predicted_decrease_groups_kernel(const double *__restrict__ Hblk, int band_blocks,
                                                                       const double *__restrict__ g, const double *__restrict__ dx,
                                                                       const double *__restrict__ u, const int32_t *__restrict__ gpo,
                                                                       double *__restrict__ out)
{
    __shared__ double red[4];
    const int64_t Bb1 = band_blocks + 1;
    const int64_t a0 = 6 * (int64_t)gpo[blockIdx.x], a1 = 6 * (int64_t)gpo[blockIdx.x + 1];
    const double ug = u[blockIdx.x];
    double s = 0.0;
    for (int64_t a = a0 + threadIdx.x; a < a1; a += 256) {
        const int64_t blk = a / 6, r = a - blk * 6;
        const double dgl = Hblk[blk * Bb1 * 36 + r * 6 + r];
        s += dx[a] * (ug * dgl * dx[a] - g[a]);
    }
    const double t = block_sum_256(s, red);
    if (threadIdx.x == 0) out[blockIdx.x] = 0.5 * t;
}
// cur[j] <- trial[j] for the poses of groups whose step was accepted
__global__ void select_poses_kernel(double *__restrict__ cur, const double *__restrict__ trial, const int32_t *__restrict__ accept,
                                    const int32_t *__restrict__ grp_of_pose, int n_poses)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 12 * n_poses) return;
    if (accept[grp_of_pose[t / 12]]) cur[t] = trial[t];
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
// with_records: the voxel pass of the full evaluation instead of the cost-only kernel -- the same per-chunk cost sums, plus the
// voxel records at `poses`.  The LM loop costs its trial point like that: if the step is accepted, the next evaluation is AT that
// point and starts from the records already there (launch_eval, skip_voxel_pass) instead of reading every cluster again.
void launch_cost(const BalmDev &d, const double *poses, double *chunk_cost, double *out, hipStream_t s,
                 hipEvent_t k0, hipEvent_t k1, bool with_records)
{
    if (k0) hipEventRecord(k0, s);
    if (with_records) hipLaunchKernelGGL(balm_voxel_kernel, dim3((unsigned)d.n_chunks), dim3(LVBA_CF), 0, s, d, poses, chunk_cost);
    else hipLaunchKernelGGL(balm_cost_kernel, dim3((unsigned)d.n_chunks), dim3(LVBA_CF), 0, s, d, poses, chunk_cost);
    if (k1) hipEventRecord(k1, s);
    hipLaunchKernelGGL(reduce_chunks_kernel, dim3(1), dim3(1024), 0, s, chunk_cost, d.n_chunks, out);
}

void launch_pairs(const PairDev &pd, double *Hblk, hipStream_t s)
{
    if (pd.nnzb > 0) {
        if (pd.col_form) {
            const dim3 grid((unsigned)((((pd.nnzb + 4 * LVBA_PC_ITEMS - 1) / (4 * LVBA_PC_ITEMS)) + 7) / 8 * 8));
            if (pd.col_form == 2) hipLaunchKernelGGL(balm_pair_col_kernel<true>, grid, dim3(256), 0, s, pd, Hblk);
            else hipLaunchKernelGGL(balm_pair_col_kernel<false>, grid, dim3(256), 0, s, pd, Hblk);
        } else {
            const dim3 grid((unsigned)((((pd.nnzb + 15) / 16) + 7) / 8 * 8));
            hipLaunchKernelGGL(balm_pair_staged_kernel, grid, dim3(256), 0, s, pd, Hblk);
        }
    }
    if (pd.n_multi > 0)
        hipLaunchKernelGGL(balm_pair_reduce_kernel, dim3((unsigned)((pd.n_multi * 18 + 255) / 256)), dim3(256), 0, s, pd, Hblk);
}

// skip_voxel_pass: the voxel records and chunk costs at `poses` are already in place (launch_cost with_records at the same poses)
void launch_eval(const BalmDev &d, const PairDev &pd, const double *poses, double *Hblk, int64_t hblk_doubles, double *g,
                 double *chunk_cost, double *out, bool zero_first, hipStream_t s, hipEvent_t k0, hipEvent_t k1, bool skip_voxel_pass)
{
    if (zero_first) hipMemsetAsync(Hblk, 0, (size_t)hblk_doubles * sizeof(double), s);
    if (k0) hipEventRecord(k0, s);
    if (!skip_voxel_pass) hipLaunchKernelGGL(balm_voxel_kernel, dim3((unsigned)d.n_chunks), dim3(LVBA_CF), 0, s, d, poses, chunk_cost);
    if (pd.col_form == 2) hipLaunchKernelGGL(balm_factor_kernel<true>, dim3((unsigned)(d.n_poses * d.S)), dim3(256), 0, s, d, poses);
    else hipLaunchKernelGGL(balm_factor_kernel<false>, dim3((unsigned)(d.n_poses * d.S)), dim3(256), 0, s, d, poses);
    hipLaunchKernelGGL(balm_diag_reduce_kernel, dim3((unsigned)((32 * (int64_t)d.n_poses + 255) / 256)), dim3(256), 0, s, d, Hblk, g);
    launch_pairs(pd, Hblk, s);
    if (k1) hipEventRecord(k1, s);
    hipLaunchKernelGGL(reduce_chunks_kernel, dim3(1), dim3(1024), 0, s, chunk_cost, d.n_chunks, out);
}

void launch_aos_to_soa(const double *aos, const int32_t *fmap, int64_t F, double *soa, hipStream_t s)
{
    hipLaunchKernelGGL(aos_to_soa_kernel, dim3((unsigned)((10 * F + 255) / 256)), dim3(256), 0, s, aos, fmap, F, soa);
}

void launch_gather_csc(const double *clu, const int32_t *csc_f, int64_t F, double *clu_csc, hipStream_t s)
{
    hipLaunchKernelGGL(gather_csc_kernel, dim3((unsigned)((F + 255) / 256)), dim3(256), 0, s, clu, csc_f, F, clu_csc);
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

void launch_lm_report(const double *scal2, const double *q1_part, int n_q1, const double *cost_cur, const int *status, double *host_pin,
                      hipStream_t s)
{
    hipLaunchKernelGGL(lm_report_kernel, dim3(1), dim3(64), 0, s, scal2, q1_part, n_q1, cost_cur, status, host_pin);
}

int launch_retract_q1(const double *poses, const double *dx, double *out, int n_poses, const double *Hblk, int band_blocks, const double *g,
                      double u, double *q1_part, hipStream_t s)
{
    const int nb = (n_poses + 127) / 128;
    hipLaunchKernelGGL(retract_q1_kernel, dim3(nb), dim3(128), 0, s, poses, dx, out, n_poses, Hblk, band_blocks, g, u, q1_part);
    return nb;
}

void launch_reduce_chunks_groups(const double *chunk_cost, const int64_t *gco, int n_groups, double *out, hipStream_t s)
{
    hipLaunchKernelGGL(reduce_chunks_groups_kernel, dim3((unsigned)n_groups), dim3(256), 0, s, chunk_cost, gco, out);
}

void launch_cost_chunks(const BalmDev &d, const double *poses, double *chunk_cost, hipStream_t s)
{
    hipLaunchKernelGGL(balm_cost_kernel, dim3((unsigned)d.n_chunks), dim3(LVBA_CF), 0, s, d, poses, chunk_cost);
}

void launch_predicted_decrease_groups(const double *Hblk, int band_blocks, const double *g, const double *dx, const double *u,
                                      const int32_t *gpo, int n_groups, double *out, hipStream_t s)
{
    hipLaunchKernelGGL(predicted_decrease_groups_kernel, dim3((unsigned)n_groups), dim3(256), 0, s, Hblk, band_blocks, g, dx, u, gpo, out);
}

void launch_select_poses(double *cur, const double *trial, const int32_t *accept, const int32_t *grp_of_pose, int n_poses, hipStream_t s)
{
    hipLaunchKernelGGL(select_poses_kernel, dim3((unsigned)((12 * n_poses + 255) / 256)), dim3(256), 0, s, cur, trial, accept, grp_of_pose, n_poses);
}

} // namespace lvba
