// visual_kernels.hip -- gfx950 kernels of the visual bundle-adjustment stage (reference
// LvbaSystem::optimizeCameraPoses, src/lvba_system.cpp:1571-1665, solved there by Ceres DENSE_SCHUR).
//
// One LM iteration = linearise (residuals + hand-derived Jacobians per observation), eliminate the 3x3 landmark
// blocks, assemble the reduced camera system, solve it, back-substitute.  With C_i = L_i L_i^T the Cholesky factor
// of a landmark's (damped) block and E = Jc^T Jp, every camera-pair block of the Schur complement is
//        S_ab = [a==b] (Jc^T Jc + D^2)  -  sum_landmarks Y_a Y_b^T ,   Y = E L^-T   (6x3 per observation)
// the same rank-3 structure as the LiDAR Hessian, so the per-block pair pass (balm_pair_kernel), the camera
// ordering and the LDL^T solver are shared with the BALM stage (block_system.hip).
//   vis_residual_kernel   lane = observation / landmark : residuals (+ Jacobians), cost partials
//   vis_colnorm_*         Jacobi column scaling 1/(1+||col||), fixed at iteration 0 (Ceres jacobi_scaling)
//   vis_point_kernel      lane = landmark : C_i, LM diagonal, Cholesky, z = L^-1 g
//   vis_cam_kernel        workgroup = (camera, slice) of the camera-major order: Y per observation, diagonal block
//                         and reduced right-hand side summed in registers
//   vis_back_kernel       lane = landmark : landmark step by back-substitution + model cost change
//   vis_apply_kernel      candidate point x (+) step on the manifold, step / parameter norms
// Camera order everywhere is the solver's (RCM) order; camera `fixed_cam` is constant (zero Jacobian columns).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "lvba_internal.h"
#include "visual_math.h"

namespace lvba {

__device__ __forceinline__ double v_wave_sum(double x)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    return x;
}
__device__ __forceinline__ double v_block_sum(double x, double *red) // 256 threads; valid in every thread
{
    x = v_wave_sum(x);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
    __syncthreads();
    const double t = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return t;
}

// sum over the four lanes of a DPP quad; every lane of the quad gets the total (quad_perm [1,0,3,2], then [2,3,0,1])
__device__ __forceinline__ double v_quad_sum(double x)
{
#define LVBA_QUAD_ADD(ctrl)                                                                          \
    do {                                                                                             \
        const int lo_ = __builtin_amdgcn_update_dpp(0, __double2loint(x), (ctrl), 0xF, 0xF, false);  \
        const int hi_ = __builtin_amdgcn_update_dpp(0, __double2hiint(x), (ctrl), 0xF, 0xF, false);  \
        x += __hiloint2double(hi_, lo_);                                                             \
    } while (0)
    LVBA_QUAD_ADD(0xB1);
    LVBA_QUAD_ADD(0x4E);
#undef LVBA_QUAD_ADD
    return x;
}

// residuals at (qc, tc, Xp); threads [0,O) observations, [O, O+Ta) plane priors.  part[blockIdx] = sum r^2.
template <bool JAC>
__global__ __launch_bounds__(256) void vis_residual_kernel(VisDev d, const double *__restrict__ qc, const double *__restrict__ tc,
                                                           const double *__restrict__ Xp, double *__restrict__ part)
{
    __shared__ double red[4];
    const int64_t gid = blockIdx.x * (int64_t)256 + threadIdx.x;
    double ss = 0.0;
    if (gid < d.O) {
        const int cam = d.cam[gid];
        const int64_t i = d.track_of_obs[gid];
        double q[4], t[3], X[3], r[2], Jc[12], Jp[6];
#pragma unroll
        for (int e = 0; e < 4; ++e) q[e] = qc[4 * (int64_t)cam + e];
#pragma unroll
        for (int e = 0; e < 3; ++e) { t[e] = tc[3 * (int64_t)cam + e]; X[e] = Xp[3 * i + e]; }
        reproj_eval<JAC>(q, t, X, d.uv[2 * gid], d.uv[2 * gid + 1], d.intr, d.inv_sig_px, r, Jc, Jp);
        if (JAC) {
            // r, Jc, Jp describe the linearisation point: a cost-only evaluation (a trial point, which may be rejected) must not
            // touch them -- the next iteration linearises at the SAME point with a smaller radius
            d.r[2 * gid] = r[0];
            d.r[2 * gid + 1] = r[1];
            const bool fixed = cam == d.fixed_cam;
#pragma unroll
            for (int e = 0; e < 12; ++e) d.Jc[12 * gid + e] = fixed ? 0.0 : Jc[e];
#pragma unroll
            for (int e = 0; e < 6; ++e) d.Jp[6 * gid + e] = Jp[e];
        }
        ss = r[0] * r[0] + r[1] * r[1];
    } else if (gid < d.O + d.Ta) {
        const int64_t i = gid - d.O;
        double X[3], pl[4], J[3];
#pragma unroll
        for (int e = 0; e < 3; ++e) X[e] = Xp[3 * i + e];
#pragma unroll
        for (int e = 0; e < 4; ++e) pl[e] = d.plane[4 * i + e];
        const double rp = plane_eval(X, pl, d.inv_sig_pl, JAC ? J : nullptr);
        if (JAC) {
            d.rpl[i] = rp;
#pragma unroll
            for (int e = 0; e < 3; ++e) d.Jpl[3 * i + e] = J[e];
        }
        ss = rp * rp;
    }
    const double tot = v_block_sum(ss, red);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

// Jacobi scaling of the landmark columns: sc = 1 / (1 + sqrt(sum of squares of the column))
__global__ void vis_colnorm_pt_kernel(VisDev d)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= d.Ta) return;
    double s[3] = {d.Jpl[3 * i] * d.Jpl[3 * i], d.Jpl[3 * i + 1] * d.Jpl[3 * i + 1], d.Jpl[3 * i + 2] * d.Jpl[3 * i + 2]};
    for (int64_t o = d.off[i]; o < d.off[i + 1]; ++o)
#pragma unroll
        for (int e = 0; e < 3; ++e) s[e] += d.Jp[6 * o + e] * d.Jp[6 * o + e] + d.Jp[6 * o + 3 + e] * d.Jp[6 * o + 3 + e];
#pragma unroll
    for (int e = 0; e < 3; ++e) d.sc_pt[3 * i + e] = 1.0 / (1.0 + sqrt(s[e]));
}

// ... and of the camera columns: one wavefront per camera over its camera-major observation list (one THREAD per camera walked
// 250 dependent gathers: 180 us, a tenth of a whole refinement)
__device__ __forceinline__ void colsq_cam(const VisDev &d, int64_t I, double (&s)[6])
{
#pragma unroll
    for (int e = 0; e < 6; ++e) s[e] = 0.0;
    for (int64_t t = d.csc_off[I] + (threadIdx.x & 63); t < d.csc_off[I + 1]; t += 64) {
        const double2 *jc = reinterpret_cast<const double2 *>(d.Jc + 12 * (int64_t)d.csc_f[t]);
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const double2 a = jc[e], b = jc[3 + e];
            s[2 * e] += a.x * a.x + b.x * b.x;
            s[2 * e + 1] += a.y * a.y + b.y * b.y;
        }
    }
#pragma unroll
    for (int e = 0; e < 6; ++e) s[e] = v_wave_sum(s[e]);
}
__global__ __launch_bounds__(64) void vis_colnorm_cam_kernel(VisDev d)
{
    double s[6];
    colsq_cam(d, blockIdx.x, s);
    if (threadIdx.x == 0)
#pragma unroll
        for (int e = 0; e < 6; ++e) d.sc_cam[6 * (int64_t)blockIdx.x + e] = 1.0 / (1.0 + sqrt(s[e]));
}

// sharded form: the sums of squares alone (all-reduced over the ranks' track shards before the scaling is taken)
__global__ __launch_bounds__(64) void vis_colsum_cam_kernel(VisDev d)
{
    double s[6];
    colsq_cam(d, blockIdx.x, s);
    if (threadIdx.x == 0)
#pragma unroll
        for (int e = 0; e < 6; ++e) d.colsum[6 * (int64_t)blockIdx.x + e] = s[e];
}
__global__ void vis_colnorm_cam_finish_kernel(VisDev d)
{
    const int64_t a = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (a < 6 * (int64_t)d.M) d.sc_cam[a] = 1.0 / (1.0 + sqrt(d.colsum[a]));
}

// four lanes (one DPP quad) per landmark, lane `sub` takes the observations off[i] + sub, + 4, ...: C = sum Jp^T Jp (+ plane) in
// the scaled variables, LM diagonal D^2 = clamp(diag C)/radius, Cholesky of C + D^2, z = L^-1 g.  gmax: max |unscaled gradient
// entry| (bit pattern of a non-negative double).  (One lane per landmark left 125 k lanes walking four dependent gathers each:
// 31 us of latency for 12 MB.)
__global__ __launch_bounds__(256) void vis_point_kernel(VisDev d, const double *__restrict__ qc, const double *__restrict__ tc,
                                                        const double *__restrict__ Xp, double radius, double min_diag, double max_diag,
                                                        unsigned long long *gmax)
{
    __shared__ double redm[4];
    const int64_t gid = blockIdx.x * (int64_t)256 + threadIdx.x;
    const int sub = (int)(gid & 3);
    double gm = 0.0;
    // a workgroup walks several groups of 64 landmarks (vis_quad_grid): with one group each, the launch of ~1 900 workgroups that
    // live for 2 us took longer than their work
    for (int64_t i = gid >> 2; i < d.Ta; i += (int64_t)gridDim.x * 64) { // whole quads
        const double sp[3] = {d.sc_pt[3 * i], d.sc_pt[3 * i + 1], d.sc_pt[3 * i + 2]};
        double C[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
        // residuals and landmark Jacobians are re-computed from the state (vis_cam_kernel does the same): nothing an LM iteration
        // reads was stored by an earlier kernel, so an accepted step needs no linearisation pass of its own
        const double X[3] = {Xp[3 * i], Xp[3 * i + 1], Xp[3 * i + 2]};
        if (sub == 0) {
            double pl[4], Jl[3];
#pragma unroll
            for (int e = 0; e < 4; ++e) pl[e] = d.plane[4 * i + e];
            const double rp = plane_eval(X, pl, d.inv_sig_pl, Jl);
            const double j[3] = {Jl[0] * sp[0], Jl[1] * sp[1], Jl[2] * sp[2]};
            C[0] += j[0] * j[0]; C[1] += j[1] * j[0]; C[2] += j[1] * j[1]; C[3] += j[2] * j[0]; C[4] += j[2] * j[1]; C[5] += j[2] * j[2];
            g[0] += j[0] * rp; g[1] += j[1] * rp; g[2] += j[2] * rp;
        }
        const int64_t o1 = d.off[i + 1];
        for (int64_t o = d.off[i] + sub; o < o1; o += 4) {
            const int64_t cam = d.cam[o];
            const double2 qa = *reinterpret_cast<const double2 *>(qc + 4 * cam), qb = *reinterpret_cast<const double2 *>(qc + 4 * cam + 2);
            const double q[4] = {qa.x, qa.y, qb.x, qb.y}, t[3] = {tc[3 * cam], tc[3 * cam + 1], tc[3 * cam + 2]};
            const double2 uv = *reinterpret_cast<const double2 *>(d.uv + 2 * o);
            double rr[2], Jc[12], P[6];
            reproj_eval<true>(q, t, X, uv.x, uv.y, d.intr, d.inv_sig_px, rr, Jc, P);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const double j[3] = {P[3 * k] * sp[0], P[3 * k + 1] * sp[1], P[3 * k + 2] * sp[2]};
                const double rk = rr[k];
                C[0] += j[0] * j[0]; C[1] += j[1] * j[0]; C[2] += j[1] * j[1]; C[3] += j[2] * j[0]; C[4] += j[2] * j[1]; C[5] += j[2] * j[2];
                g[0] += j[0] * rk; g[1] += j[1] * rk; g[2] += j[2] * rk;
            }
        }
#pragma unroll
        for (int e = 0; e < 6; ++e) C[e] = v_quad_sum(C[e]);
#pragma unroll
        for (int e = 0; e < 3; ++e) g[e] = v_quad_sum(g[e]);
        if (sub == 0) {
            gm = fmax(gm, fmax(fabs(g[0] / sp[0]), fmax(fabs(g[1] / sp[1]), fabs(g[2] / sp[2]))));
            C[0] += fmin(fmax(C[0], min_diag), max_diag) / radius;
            C[2] += fmin(fmax(C[2], min_diag), max_diag) / radius;
            C[5] += fmin(fmax(C[5], min_diag), max_diag) / radius;
            double L[6], z[3];
            chol3(C, L);
            chol3_fwd(L, g, z);
#pragma unroll
            for (int e = 0; e < 6; ++e) d.Lp[6 * i + e] = L[e];
#pragma unroll
            for (int e = 0; e < 3; ++e) d.zp[3 * i + e] = z[e];
        }
    }
    // one atomic per workgroup: atomics on ONE address are serialised (one per wavefront of 16 landmarks cost 60 us here)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) gm = fmax(gm, __shfl_down(gm, off, 64));
    if ((threadIdx.x & 63) == 0) redm[threadIdx.x >> 6] = gm;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicMax(gmax, (unsigned long long)__double_as_longlong(fmax(fmax(redm[0], redm[1]), fmax(redm[2], redm[3]))));
}

// workgroup (I, s): slice s of camera I's observations in camera-major order.  Per observation Y = (Jc^T Jp) L^-T
// (stored for the pair pass and the back-substitution); per camera the sums of
//   D = Jc^T Jc - Y Y^T (lower 21) | reduced rhs Jc^T r - Y z (6) | diag(Jc^T Jc) (6) | Jc^T r (6)   -> part[.][40]
__global__ __launch_bounds__(256) void vis_cam_kernel(VisDev d, const double *__restrict__ qc, const double *__restrict__ tc,
                                                      const double *__restrict__ Xp, int wave_units)
{
    __shared__ double red[4 * 39];
    // wave_units = 1: every WAVEFRONT of the workgroup owns a (camera, slice) unit of its own (short slices: four units per
    // workgroup -- a quarter of the workgroups to dispatch); 0: the workgroup's four wavefronts share one unit
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nth = wave_units ? 64 : (int)blockDim.x, tid = wave_units ? lane : (int)threadIdx.x;
    const int unit = wave_units ? (int)blockIdx.x * 4 + wv : (int)blockIdx.x;
    if (unit >= d.M * d.S) return; // (wave_units only; no workgroup barrier below in that mode)
    const int I = unit / d.S, s = unit - I * d.S;
    const int64_t seg0 = d.csc_off[I], len = d.csc_off[I + 1] - seg0;
    const int64_t a = seg0 + (len * s) / d.S, b = seg0 + (len * (s + 1)) / d.S;
    double sc[6];
#pragma unroll
    for (int e = 0; e < 6; ++e) sc[e] = d.sc_cam[6 * (int64_t)I + e];
    double acc[39];
#pragma unroll
    for (int e = 0; e < 39; ++e) acc[e] = 0.0;
    // The observation's residual and Jacobians are RE-COMPUTED here from the camera (uniform over the workgroup), the landmark
    // and the pixel (camera-major copy: coalesced) -- ~300 flops -- instead of gathered from what vis_residual_kernel stored in
    // landmark-major order: 160 bytes per observation in 96 / 48 / 16-byte pieces of other cameras' cache lines were most of this
    // kernel's time (85 us for 80 MB).
    double qI[4], tI[3];
#pragma unroll
    for (int e = 0; e < 4; ++e) qI[e] = qc[4 * (int64_t)I + e];
#pragma unroll
    for (int e = 0; e < 3; ++e) tI[e] = tc[3 * (int64_t)I + e];
    const bool fixed = I == d.fixed_cam;
    for (int64_t t = a + tid; t < b; t += nth) {
        const int64_t i = d.group_of_pos[t];
        double J[12], P[6], L[6], z[3], rr[2];
        {
            const double2 uv = *reinterpret_cast<const double2 *>(d.uv_cm + 2 * t);
            const double X[3] = {Xp[3 * i], Xp[3 * i + 1], Xp[3 * i + 2]};
            reproj_eval<true>(qI, tI, X, uv.x, uv.y, d.intr, d.inv_sig_px, rr, J, P);
            const double sp[3] = {d.sc_pt[3 * i], d.sc_pt[3 * i + 1], d.sc_pt[3 * i + 2]};
#pragma unroll
            for (int e = 0; e < 12; ++e) J[e] = fixed ? 0.0 : J[e] * sc[e % 6];
#pragma unroll
            for (int e = 0; e < 6; ++e) P[e] *= sp[e % 3];
            const double2 *lp = reinterpret_cast<const double2 *>(d.Lp + 6 * i);
#pragma unroll
            for (int e = 0; e < 3; ++e) { const double2 v = lp[e]; L[2 * e] = v.x; L[2 * e + 1] = v.y; }
        }
#pragma unroll
        for (int e = 0; e < 3; ++e) z[e] = d.zp[3 * i + e];
        const double r0 = rr[0], r1 = rr[1];
        double Y[18];
#pragma unroll
        for (int e = 0; e < 6; ++e) {
            // row e of E = Jc^T Jp, then row e of Y: Y L^T = E
            const double E0 = J[e] * P[0] + J[6 + e] * P[3], E1 = J[e] * P[1] + J[6 + e] * P[4], E2 = J[e] * P[2] + J[6 + e] * P[5];
            const double y0 = E0 / L[0];
            const double y1 = (E1 - y0 * L[1]) / L[2];
            const double y2 = (E2 - y0 * L[3] - y1 * L[4]) / L[5];
            Y[e] = y0; Y[6 + e] = y1; Y[12 + e] = y2;
        }
        double2 *yo = reinterpret_cast<double2 *>(d.Y + 18 * t);
#pragma unroll
        for (int e = 0; e < 9; ++e) yo[e] = double2{Y[2 * e], Y[2 * e + 1]};
        int p = 0;
#pragma unroll
        for (int c = 0; c < 6; ++c)
#pragma unroll
            for (int rr = c; rr < 6; ++rr) {
                acc[p] += J[rr] * J[c] + J[6 + rr] * J[6 + c] - (Y[rr] * Y[c] + Y[6 + rr] * Y[6 + c] + Y[12 + rr] * Y[12 + c]);
                ++p;
            }
#pragma unroll
        for (int e = 0; e < 6; ++e) {
            const double jr = J[e] * r0 + J[6 + e] * r1;
            acc[21 + e] += jr - (Y[e] * z[0] + Y[6 + e] * z[1] + Y[12 + e] * z[2]);
            acc[27 + e] += J[e] * J[e] + J[6 + e] * J[6 + e];
            acc[33 + e] += jr;
        }
    }
#pragma unroll
    for (int e = 0; e < 39; ++e) {
        const double v = v_wave_sum(acc[e]);
        if (lane == 0) red[wv * 39 + e] = v;
    }
    if (wave_units) {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane < 39) d.part[(int64_t)unit * 40 + lane] = red[wv * 39 + lane];
        return;
    }
    __syncthreads();
    if (tid < 39) {
        double v = red[tid];
        for (int w = 1; w < (nth >> 6); ++w) v += red[w * 39 + tid];
        d.part[(int64_t)unit * 40 + tid] = v;
    }
}

// A camera's share of Ceres' gradient_max_norm = || x - Plus(x, -g) ||_inf over the ambient parameters (the max norm of the
// PROJECTED gradient step: trust_region_minimizer.cc of 2.1): |g| itself on the Euclidean block (translation; likewise the
// landmarks in vis_point_kernel), the four components of q - Plus(q, -g_rot) on the quaternion block.  gu: unscaled gradient.
__device__ __forceinline__ double cam_projected_gmax(const double *__restrict__ qc, int64_t I, const double (&gu)[6])
{
    const double q[4] = {qc[4 * I], qc[4 * I + 1], qc[4 * I + 2], qc[4 * I + 3]}, ng[3] = {-gu[0], -gu[1], -gu[2]};
    double q2[4], gm = 0.0;
    quat_plus(q, ng, q2);
#pragma unroll
    for (int e = 0; e < 4; ++e) gm = fmax(gm, fabs(q[e] - q2[e]));
#pragma unroll
    for (int c = 3; c < 6; ++c) gm = fmax(gm, fabs(gu[c]));
    return gm;
}

// one thread per camera: slice sums -> diagonal block (lower) + LM diagonal, reduced right-hand side, gradient max
__global__ void vis_cam_reduce_kernel(VisDev d, double radius, double min_diag, double max_diag, double *__restrict__ Hblk,
                                      double *__restrict__ g, const double *__restrict__ qc, unsigned long long *gmax)
{
    const int64_t I = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (I >= d.M) return;
    double acc[39];
#pragma unroll
    for (int e = 0; e < 39; ++e) acc[e] = 0.0;
    for (int q = 0; q < d.S; ++q)
#pragma unroll
        for (int e = 0; e < 39; ++e) acc[e] += d.part[(I * d.S + q) * 40 + e];
    double *hp = Hblk + I * (int64_t)(d.band_blocks + 1) * 36;
    int p = 0;
    double gu[6];
#pragma unroll
    for (int c = 0; c < 6; ++c)
#pragma unroll
        for (int rr = c; rr < 6; ++rr) {
            double v = acc[p++];
            // sharded: diag(Jc^T Jc) is only this rank's part -- the LM diagonal is added after the all-reduce (vis_cam_finish_kernel)
            if (rr == c && !d.dist) v += fmin(fmax(acc[27 + c], min_diag), max_diag) / radius;
            hp[c * 6 + rr] = v;
        }
#pragma unroll
    for (int e = 0; e < 6; ++e) {
        g[6 * I + e] = acc[21 + e];
        gu[e] = acc[33 + e] / d.sc_cam[6 * I + e];
    }
    if (d.dist) {
#pragma unroll
        for (int e = 0; e < 6; ++e) {
            d.camsum[6 * I + e] = acc[27 + e];
            d.camsum[6 * (int64_t)d.M + 6 * I + e] = acc[33 + e];
        }
        return;
    }
    atomicMax(gmax, (unsigned long long)__double_as_longlong(cam_projected_gmax(qc, I, gu)));
}

// sharded, after [H | g] and camsum have been all-reduced: the LM diagonal on the diagonal blocks and the cameras' part of
// the gradient max (the landmarks' part is local to their rank; gmax itself is max-reduced afterwards)
__global__ void vis_cam_finish_kernel(VisDev d, double radius, double min_diag, double max_diag, double *__restrict__ Hblk,
                                      const double *__restrict__ qc, unsigned long long *gmax)
{
    const int64_t I = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (I >= d.M) return;
    double *hp = Hblk + I * (int64_t)(d.band_blocks + 1) * 36;
    double gu[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        hp[c * 6 + c] += fmin(fmax(d.camsum[6 * I + c], min_diag), max_diag) / radius;
        gu[c] = d.camsum[6 * (int64_t)d.M + 6 * I + c] / d.sc_cam[6 * I + c];
    }
    atomicMax(gmax, (unsigned long long)__double_as_longlong(cam_projected_gmax(qc, I, gu)));
}

// four lanes (one DPP quad) per landmark, as in vis_point_kernel: step_p = -L^-T (z + sum_obs Y^T step_c), and the model cost
// change -sum_rows m (r + m/2), m = J step (scaled variables).  part[blockIdx] = partial of the model cost change.
__global__ __launch_bounds__(256) void vis_back_kernel(VisDev d, const double *__restrict__ step_c, const double *__restrict__ qc,
                                                       const double *__restrict__ tc, const double *__restrict__ Xp, double *__restrict__ part)
{
    __shared__ double red[4];
    const int64_t gid = blockIdx.x * (int64_t)256 + threadIdx.x;
    const int sub = (int)(gid & 3);
    double mc = 0.0;
    for (int64_t i = gid >> 2; i < d.Ta; i += (int64_t)gridDim.x * 64) { // whole quads; several groups of 64 landmarks per workgroup
        const int64_t o0 = d.off[i] + sub, o1 = d.off[i + 1];
        double s[3] = {0.0, 0.0, 0.0};
        if (sub == 0) { s[0] = d.zp[3 * i]; s[1] = d.zp[3 * i + 1]; s[2] = d.zp[3 * i + 2]; }
        for (int64_t o = o0; o < o1; o += 4) {
            const double2 *y = reinterpret_cast<const double2 *>(d.Y + 18 * (int64_t)d.pos_of[o]);
            const double2 *sc = reinterpret_cast<const double2 *>(step_c + 6 * (int64_t)d.cam[o]);
            const double2 c0 = sc[0], c1 = sc[1], c2 = sc[2];
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const double2 y0 = y[3 * m], y1 = y[3 * m + 1], y2 = y[3 * m + 2];
                s[m] += y0.x * c0.x + y0.y * c0.y + y1.x * c1.x + y1.y * c1.y + y2.x * c2.x + y2.y * c2.y;
            }
        }
#pragma unroll
        for (int e = 0; e < 3; ++e) s[e] = v_quad_sum(s[e]);
        double L[6], sp[3], st[3];
#pragma unroll
        for (int e = 0; e < 6; ++e) L[e] = d.Lp[6 * i + e];
        chol3_bwd(L, s, st);
#pragma unroll
        for (int e = 0; e < 3; ++e) { st[e] = -st[e]; sp[e] = d.sc_pt[3 * i + e]; }
        const double X[3] = {Xp[3 * i], Xp[3 * i + 1], Xp[3 * i + 2]};
        if (sub == 0) {
#pragma unroll
            for (int e = 0; e < 3; ++e) d.step_p[3 * i + e] = st[e];
            double pl[4], Jl[3];
#pragma unroll
            for (int e = 0; e < 4; ++e) pl[e] = d.plane[4 * i + e];
            const double rp = plane_eval(X, pl, d.inv_sig_pl, Jl);
            const double m = Jl[0] * sp[0] * st[0] + Jl[1] * sp[1] * st[1] + Jl[2] * sp[2] * st[2];
            mc -= m * (rp + 0.5 * m);
        }
        for (int64_t o = o0; o < o1; o += 4) {
            const int64_t cam = d.cam[o];
            const double *sc = step_c + 6 * cam;
            const double *scl = d.sc_cam + 6 * cam;
            double w[6];
#pragma unroll
            for (int e = 0; e < 6; ++e) w[e] = cam == d.fixed_cam ? 0.0 : scl[e] * sc[e]; // the constant camera: zero Jacobian columns
            const double2 qa = *reinterpret_cast<const double2 *>(qc + 4 * cam), qb = *reinterpret_cast<const double2 *>(qc + 4 * cam + 2);
            const double q[4] = {qa.x, qa.y, qb.x, qb.y}, t[3] = {tc[3 * cam], tc[3 * cam + 1], tc[3 * cam + 2]};
            const double2 uv = *reinterpret_cast<const double2 *>(d.uv + 2 * o);
            double rr[2], J[12], P[6];
            reproj_eval<true>(q, t, X, uv.x, uv.y, d.intr, d.inv_sig_px, rr, J, P);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                double m = 0.0;
#pragma unroll
                for (int e = 0; e < 6; ++e) m += J[6 * k + e] * w[e];
#pragma unroll
                for (int e = 0; e < 3; ++e) m += P[3 * k + e] * sp[e] * st[e];
                mc -= m * (rr[k] + 0.5 * m);
            }
        }
    }
    const double tot = v_block_sum(mc, red);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

// candidate = x (+) step (unscaled): threads [0,M) cameras, [M, M+Ta) landmarks.  part[2*blk] = |cand - x|^2,
// part[2*blk+1] = |x|^2 over the free parameters (ambient, as Ceres' step_norm / x_norm).
__global__ __launch_bounds__(256) void vis_apply_kernel(VisDev d, const double *__restrict__ step_c, const double *__restrict__ qc,
                                                        const double *__restrict__ tc, const double *__restrict__ Xp,
                                                        double *__restrict__ qc2, double *__restrict__ tc2,
                                                        double *__restrict__ Xp2, double *__restrict__ part)
{
    __shared__ double red[4];
    const int64_t gid = blockIdx.x * (int64_t)256 + threadIdx.x;
    double dn = 0.0, xn = 0.0;
    if (gid < d.M) {
        double q[4], t[3], dl[6], q2[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) q[e] = qc[4 * gid + e];
#pragma unroll
        for (int e = 0; e < 3; ++e) t[e] = tc[3 * gid + e];
#pragma unroll
        for (int e = 0; e < 6; ++e) dl[e] = step_c[6 * gid + e] * d.sc_cam[6 * gid + e];
        if (gid == d.fixed_cam) {
#pragma unroll
            for (int e = 0; e < 4; ++e) qc2[4 * gid + e] = q[e];
#pragma unroll
            for (int e = 0; e < 3; ++e) tc2[3 * gid + e] = t[e];
        } else {
            quat_plus(q, dl, q2);
            double dc = 0.0, xc = 0.0;
#pragma unroll
            for (int e = 0; e < 4; ++e) { qc2[4 * gid + e] = q2[e]; dc += (q2[e] - q[e]) * (q2[e] - q[e]); xc += q[e] * q[e]; }
#pragma unroll
            for (int e = 0; e < 3; ++e) { tc2[3 * gid + e] = t[e] + dl[3 + e]; dc += dl[3 + e] * dl[3 + e]; xc += t[e] * t[e]; }
            if (d.count_cams) { dn = dc; xn = xc; } // the cameras are replicated over the ranks: one of them counts their norms
        }
    } else if (gid < d.M + d.Ta) {
        const int64_t i = gid - d.M;
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const double x = Xp[3 * i + e], dl = d.step_p[3 * i + e] * d.sc_pt[3 * i + e];
            Xp2[3 * i + e] = x + dl;
            dn += dl * dl;
            xn += x * x;
        }
    }
    const double t0 = v_block_sum(dn, red);
    const double t1 = v_block_sum(xn, red);
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = t0; part[2 * blockIdx.x + 1] = t1; }
}

// out[0] = sum part[0..n), out[1] = sum of odd entries if stride 2 (deterministic, single workgroup)
__global__ __launch_bounds__(1024) void vis_reduce_kernel(const double *__restrict__ part, int64_t n, int stride, double *__restrict__ out)
{
    __shared__ double red[16];
    for (int k = 0; k < stride; ++k) {
        double s = 0.0;
        for (int64_t i = threadIdx.x; i < n; i += 1024) s += part[stride * i + k];
        s = v_wave_sum(s);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int i = 0; i < 16; ++i) t += red[i];
            out[k] = t;
        }
        __syncthreads();
    }
}

// camera-major copy of the pixel observations (set-up, once)
__global__ void vis_gather_uv_kernel(VisDev d, double *__restrict__ uv_cm)
{
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= d.O) return;
    const int64_t o = d.csc_f[t];
    uv_cm[2 * t] = d.uv[2 * o];
    uv_cm[2 * t + 1] = d.uv[2 * o + 1];
}

// End of an LM iteration on one rank, ONE workgroup: the three per-workgroup partial lists (model cost change | step and
// parameter norms | cost at the trial point) summed in a fixed order, and everything the host's accept / reject logic reads
// written straight into its pinned buffer: host[0..4] = scal[0..4], host[8] = gradient max (bits), host[9] = solver status.
// Replaces three single-workgroup reductions and three device-to-host copies (six stream operations of ~5 us each).
__global__ __launch_bounds__(1024) void vis_finish_kernel(const double *__restrict__ part, int64_t nb_back, int64_t nb_apply,
                                                          int64_t nb_res, double *__restrict__ scal,
                                                          const unsigned long long *__restrict__ gmax, const int *__restrict__ status,
                                                          double *__restrict__ host)
{
    __shared__ double red[4][16];
    const double *pb = part, *pa = part + nb_back, *pr = part + nb_back + 2 * nb_apply;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int64_t i = threadIdx.x; i < nb_back; i += 1024) s0 += pb[i];
    for (int64_t i = threadIdx.x; i < nb_apply; i += 1024) { s1 += pa[2 * i]; s2 += pa[2 * i + 1]; }
    for (int64_t i = threadIdx.x; i < nb_res; i += 1024) s3 += pr[i];
    s0 = v_wave_sum(s0); s1 = v_wave_sum(s1); s2 = v_wave_sum(s2); s3 = v_wave_sum(s3);
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
        red[0][w] = s0; red[1][w] = s1; red[2][w] = s2; red[3][w] = s3;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t[4] = {0.0, 0.0, 0.0, 0.0};
        for (int k = 0; k < 4; ++k)
            for (int w = 0; w < 16; ++w) t[k] += red[k][w];
        scal[2] = t[0]; scal[3] = t[1]; scal[4] = t[2]; scal[1] = t[3];
        host[0] = scal[0]; host[1] = t[3]; host[2] = t[0]; host[3] = t[1]; host[4] = t[2];
        host[8] = __longlong_as_double((long long)gmax[0]);
        int st = status[0];
        double stw = 0.0;
        memcpy(&stw, &st, sizeof st);
        host[9] = stw;
    }
}

// ---------------------------------------------------------------------------------------------- launchers
static inline unsigned nblk(int64_t n, int b) { return (unsigned)((n + b - 1) / b > 0 ? (n + b - 1) / b : 1); }
// vis_back_kernel keeps one group of 64 landmarks per workgroup (1 / 2 / 4 groups: 0.315 / 0.326 / 0.320 ms per LM iteration: its two
// rounds of gathers are what it waits for, and a loop puts them one after the other)
static inline unsigned vis_back_grid(int64_t Ta) { return nblk(4 * Ta, 256); }
// workgroups of the quad-per-landmark kernels: four groups of 64 landmarks each (measured at 125 k landmarks: 1 / 4 / 8 / 16 groups
// per workgroup -> 0.334 / 0.323 / 0.335 / 0.378 ms per LM iteration; vis_point_kernel 29 -> 19 us with four)
static inline unsigned vis_quad_grid(int64_t Ta)
{
    const unsigned n = nblk(4 * Ta, 256);
    return (n + 3) / 4;
}

void vis_launch_residuals(const VisDev &d, bool jac, const double *qc, const double *tc, const double *Xp, double *part,
                          double *cost_out, hipStream_t s)
{
    const unsigned nb = nblk(d.O + d.Ta, 256);
    if (jac) hipLaunchKernelGGL(vis_residual_kernel<true>, dim3(nb), dim3(256), 0, s, d, qc, tc, Xp, part);
    else hipLaunchKernelGGL(vis_residual_kernel<false>, dim3(nb), dim3(256), 0, s, d, qc, tc, Xp, part);
    if (cost_out) hipLaunchKernelGGL(vis_reduce_kernel, dim3(1), dim3(1024), 0, s, part, (int64_t)nb, 1, cost_out);
}

// back-substitution, candidate point and its cost with ONE closing kernel (vis_finish_kernel) instead of a reduction each
void vis_launch_step_and_trial(const VisDev &d, const double *step_c, const double *qc, const double *tc, const double *Xp, double *qc2,
                               double *tc2, double *Xp2, double *part, double *scal, const unsigned long long *gmax, const int *status,
                               double *host_pin, hipStream_t s)
{
    const unsigned nb_back = vis_back_grid(d.Ta), nb_apply = nblk(d.M + d.Ta, 256), nb_res = nblk(d.O + d.Ta, 256);
    double *pa = part + nb_back, *pr = pa + 2 * (int64_t)nb_apply;
    hipLaunchKernelGGL(vis_back_kernel, dim3(nb_back), dim3(256), 0, s, d, step_c, qc, tc, Xp, part);
    hipLaunchKernelGGL(vis_apply_kernel, dim3(nb_apply), dim3(256), 0, s, d, step_c, qc, tc, Xp, qc2, tc2, Xp2, pa);
    hipLaunchKernelGGL(vis_residual_kernel<false>, dim3(nb_res), dim3(256), 0, s, d, qc2, tc2, Xp2, pr);
    hipLaunchKernelGGL(vis_finish_kernel, dim3(1), dim3(1024), 0, s, part, (int64_t)nb_back, (int64_t)nb_apply, (int64_t)nb_res, scal,
                       gmax, status, host_pin);
}

void vis_launch_colnorms(const VisDev &d, hipStream_t s)
{
    hipLaunchKernelGGL(vis_colnorm_pt_kernel, dim3(nblk(d.Ta, 256)), dim3(256), 0, s, d);
    hipLaunchKernelGGL(vis_colnorm_cam_kernel, dim3((unsigned)d.M), dim3(64), 0, s, d);
}

void vis_launch_colsums(const VisDev &d, hipStream_t s)
{
    hipLaunchKernelGGL(vis_colnorm_pt_kernel, dim3(nblk(d.Ta, 256)), dim3(256), 0, s, d);
    hipLaunchKernelGGL(vis_colsum_cam_kernel, dim3((unsigned)d.M), dim3(64), 0, s, d);
}
void vis_launch_colnorm_finish(const VisDev &d, hipStream_t s)
{
    hipLaunchKernelGGL(vis_colnorm_cam_finish_kernel, dim3(nblk(6 * (int64_t)d.M, 256)), dim3(256), 0, s, d);
}
void vis_launch_cam_finish(const VisDev &d, double radius, double min_diag, double max_diag, double *Hblk, const double *qc,
                           unsigned long long *gmax, hipStream_t s)
{
    hipLaunchKernelGGL(vis_cam_finish_kernel, dim3(nblk(d.M, 64)), dim3(64), 0, s, d, radius, min_diag, max_diag, Hblk, qc, gmax);
}

void vis_launch_gather_uv(const VisDev &d, double *uv_cm, hipStream_t s)
{
    if (d.O > 0) hipLaunchKernelGGL(vis_gather_uv_kernel, dim3(nblk(d.O, 256)), dim3(256), 0, s, d, uv_cm);
}

void vis_launch_reduced_system(const VisDev &d, const PairDev &pd, const double *qc, const double *tc, const double *Xp, double radius, double min_diag, double max_diag, double *Hblk,
                               int64_t hblk_doubles, double *g, unsigned long long *gmax, bool zero_first, hipStream_t s)
{
    if (zero_first) hipMemsetAsync(Hblk, 0, (size_t)hblk_doubles * sizeof(double), s);
    hipMemsetAsync(gmax, 0, sizeof(unsigned long long), s);
    hipLaunchKernelGGL(vis_point_kernel, dim3(vis_quad_grid(d.Ta)), dim3(256), 0, s, d, qc, tc, Xp, radius, min_diag, max_diag, gmax);
    // one wavefront per (camera, slice) while a slice is short: its 39 sums cost one 64-lane reduction per WAVEFRONT, which at ~250
    // observations per camera was most of the kernel with four wavefronts of one observation per lane each
    const int64_t per_slice = d.O / ((int64_t)d.M * d.S > 0 ? (int64_t)d.M * d.S : 1);
    if (per_slice <= 1024) hipLaunchKernelGGL(vis_cam_kernel, dim3((unsigned)((d.M * d.S + 3) / 4)), dim3(256), 0, s, d, qc, tc, Xp, 1);
    else hipLaunchKernelGGL(vis_cam_kernel, dim3((unsigned)(d.M * d.S)), dim3(256), 0, s, d, qc, tc, Xp, 0);
    hipLaunchKernelGGL(vis_cam_reduce_kernel, dim3(nblk(d.M, 64)), dim3(64), 0, s, d, radius, min_diag, max_diag, Hblk, g, qc, gmax);
    launch_pairs(pd, Hblk, s);
}

void vis_launch_back(const VisDev &d, const double *step_c, const double *qc, const double *tc, const double *Xp, double *part,
                     double *model_out, hipStream_t s)
{
    const unsigned nb = vis_back_grid(d.Ta);
    hipLaunchKernelGGL(vis_back_kernel, dim3(nb), dim3(256), 0, s, d, step_c, qc, tc, Xp, part);
    hipLaunchKernelGGL(vis_reduce_kernel, dim3(1), dim3(1024), 0, s, part, (int64_t)nb, 1, model_out);
}

void vis_launch_apply(const VisDev &d, const double *step_c, const double *qc, const double *tc, const double *Xp, double *qc2,
                      double *tc2, double *Xp2, double *part, double *norms_out, hipStream_t s)
{
    const unsigned nb = nblk(d.M + d.Ta, 256);
    hipLaunchKernelGGL(vis_apply_kernel, dim3(nb), dim3(256), 0, s, d, step_c, qc, tc, Xp, qc2, tc2, Xp2, part);
    hipLaunchKernelGGL(vis_reduce_kernel, dim3(1), dim3(1024), 0, s, part, (int64_t)nb, 2, norms_out);
}

} // namespace lvba
