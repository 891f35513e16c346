// tests/host_emul.cpp -- TEST-ONLY: runs the per-lane device math of global-lvba_amd/csrc/balm_math.h on
// the CPU (g++), sequentially, with the same factor -> voxel -> pair decomposition as balm_eval_kernel.
// Lets `pytest -m "not gpu"` check the E - Y Y^T formulation and the 3x3 Jacobi eigen-solver against the
// oracle without a GPU.  Never built into liblvba_hip.so; not a fallback.
#include <stdint.h>
#include <string.h>
#include <vector>
#include "../global-lvba_amd/csrc/balm_math.h"

using namespace lvba;

extern "C" int emul_cost(int n_poses, int64_t V, const int64_t *voff, const int32_t *pidx,
                         const double *clusters /*[F][10]*/, const double *poses, double *cost_sum)
{
    double tot = 0;
    for (int64_t a = 0; a < V; ++a) {
        double S[10] = {0};
        for (int64_t f = voff[a]; f < voff[a + 1]; ++f) {
            const double *x = poses + 12 * (int64_t)pidx[f];
            double t[10];
            transform_cluster(clusters + 10 * f, x, x + 9, t);
            for (int e = 0; e < 10; ++e) S[e] += t[e];
        }
        tot += voxel_lambda_min(S);
    }
    *cost_sum = tot;
    return 0;
}

// H: dense [6N x 6N] symmetric (both triangles filled), g [6N], cost_sum.
extern "C" int emul_eval(int n_poses, int64_t V, const int64_t *voff, const int32_t *pidx,
                         const double *clusters, const double *poses, double *H, double *g, double *cost_sum)
{
    const int64_t n = 6 * (int64_t)n_poses;
    memset(H, 0, sizeof(double) * n * n);
    memset(g, 0, sizeof(double) * n);
    double tot = 0;
    std::vector<double> Y;
    for (int64_t a = 0; a < V; ++a) {
        const int64_t f0 = voff[a], k = voff[a + 1] - f0;
        double S[10] = {0};
        for (int64_t f = f0; f < f0 + k; ++f) {
            const double *x = poses + 12 * (int64_t)pidx[f];
            double t[10];
            transform_cluster(clusters + 10 * f, x, x + 9, t);
            for (int e = 0; e < 10; ++e) S[e] += t[e];
        }
        VoxRec vr;
        tot += voxel_finish(S, vr);
        Y.assign(18 * k, 0.0);
        for (int64_t q = 0; q < k; ++q) {
            const int64_t f = f0 + q;
            const int I = pidx[f];
            const double *x = poses + 12 * (int64_t)I;
            double D[21], gi[6];
            factor_derivs(clusters + 10 * f, x, x + 9, vr, &Y[18 * q], D, gi);
            for (int e = 0; e < 6; ++e) g[6 * I + e] += gi[e];
            for (int c = 0; c < 6; ++c)
                for (int r = c; r < 6; ++r) {
                    H[(6 * I + r) * n + 6 * I + c] += D[dlow(r, c)];
                    if (r != c) H[(6 * I + c) * n + 6 * I + r] += D[dlow(r, c)];
                }
        }
        for (int64_t qi = 0; qi < k; ++qi)
            for (int64_t qj = qi + 1; qj < k; ++qj) {
                const int I = pidx[f0 + qi], J = pidx[f0 + qj];
                const double *Yi = &Y[18 * qi], *Yj = &Y[18 * qj];
                for (int r = 0; r < 6; ++r)
                    for (int c = 0; c < 6; ++c) {
                        const double v = -(Yi[r] * Yj[c] + Yi[6 + r] * Yj[6 + c] + Yi[12 + r] * Yj[12 + c]);
                        H[(6 * I + r) * n + 6 * J + c] += v;
                        H[(6 * J + c) * n + 6 * I + r] += v;
                    }
            }
    }
    *cost_sum = tot;
    return 0;
}

extern "C" int emul_retract(int n_poses, const double *poses, const double *dx, double *out)
{
    for (int j = 0; j < n_poses; ++j) retract_pose(poses + 12 * j, dx + 6 * j, out + 12 * j);
    return 0;
}

extern "C" int emul_eig3(const double *C6, double *lam, double *U)
{
    eig3<true>(C6, lam, U);
    return 0;
}
// the direct decomposition the LM kernels use (Newton for lam0, cross products, one 2 x 2 rotation); lam0 alone: U == NULL
extern "C" int emul_eig3_planar(const double *C6, double *lam, double *U)
{
    if (U) eig3_planar<true>(C6, lam, U);
    else eig3_planar<false>(C6, lam, nullptr);
    return 0;
}

// ---- visual stage: hand-derived Jacobians of global-lvba_amd/csrc/visual_math.h --------------------------------
#include "../global-lvba_amd/csrc/visual_math.h"
extern "C" int emul_reproj(const double *q, const double *t, const double *X, const double *uv, const double *intr,
                           double sigma, double *r, double *Jc, double *Jp)
{
    return reproj_eval<true>(q, t, X, uv[0], uv[1], intr, 1.0 / sigma, r, Jc, Jp) ? 1 : 0;
}
extern "C" double emul_plane(const double *X, const double *pl, double sigma, double *J) { return plane_eval(X, pl, 1.0 / sigma, J); }
extern "C" void emul_quat_plus(const double *a, const double *d, double *out) { quat_plus(a, d, out); }
