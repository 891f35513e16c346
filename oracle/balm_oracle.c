/* CPU oracle (plain C, fp64, OpenMP) for the BALM LiDAR bundle-adjustment hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  The product path (liblvba_hip.so) never links it.
 *
 * Pinned against the reference's own BALM headers compiled with the Eigen / PCL stand-ins of oracle/shim
 * (oracle/_ref/libbalm_ref.so; tests/test_ref_pin.py, tests/golden/ref_balm.npz) -- see oracle/balm_oracle.py's
 * header -- and by finite differences, the numpy twin and autograd.
 *
 * Restates, with the reference's own formulation (Auk / umumT / per-block corrections):
 *   PointCluster::transform            include/BALM/tools.hpp:450-456
 *   Exp / hat                          include/BALM/tools.hpp:62-77,105-112
 *   VOX_HESS::acc_evaluate2            include/BALM/bavoxel.hpp:68-174
 *   VOX_HESS::evaluate_only_residual   include/BALM/bavoxel.hpp:176-203
 *   BALM2::divide_thread               include/BALM/bavoxel.hpp:597-639  (16 slices, serial sum)
 *   BALM2::damping_iter                include/BALM/bavoxel.hpp:662-767
 * Eigen pieces restated: 3x3 SelfAdjointEigenSolver -> cyclic Jacobi (ascending eigenvalues;
 * results are eigenvector-sign invariant); SimplicialLDLT -> unpivoted LDL^T of the lower
 * triangle (dense or LAPACK-style lower band storage).
 *
 * Packed problem format: see include/lvba_hip.h.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

#define THD_NUM 16 /* bavoxel.hpp:25 */

/* ------------------------------------------------------------------ small dense helpers */
static void mat3_mul(const double *A, const double *B, double *C) /* row-major 3x3 */
{
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
static void mat3_mulT(const double *A, const double *B, double *C) /* A * B^T */
{
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            C[3 * i + j] = A[3 * i] * B[3 * j] + A[3 * i + 1] * B[3 * j + 1] + A[3 * i + 2] * B[3 * j + 2];
}
static void mat3_vec(const double *A, const double *x, double *y)
{
    for (int i = 0; i < 3; i++) y[i] = A[3 * i] * x[0] + A[3 * i + 1] * x[1] + A[3 * i + 2] * x[2];
}
static void mat3T_vec(const double *A, const double *x, double *y)
{
    for (int i = 0; i < 3; i++) y[i] = A[i] * x[0] + A[3 + i] * x[1] + A[6 + i] * x[2];
}
static void hat3(const double *v, double *M) /* tools.hpp:105-112 */
{
    M[0] = 0; M[1] = -v[2]; M[2] = v[1];
    M[3] = v[2]; M[4] = 0; M[5] = -v[0];
    M[6] = -v[1]; M[7] = v[0]; M[8] = 0;
}
static void exp_so3(const double *w, double *R) /* tools.hpp:62-77 */
{
    double th = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (th >= 1e-11) {
        double k[3] = {w[0] / th, w[1] / th, w[2] / th}, K[9], KK[9];
        hat3(k, K);
        mat3_mul(K, K, KK);
        double s = sin(th), c = 1.0 - cos(th);
        for (int i = 0; i < 9; i++) R[i] = I[i] + s * K[i] + c * KK[i];
    } else
        memcpy(R, I, sizeof I);
}

/* cyclic Jacobi for a symmetric 3x3 (row-major); eigenvalues ascending, U columns = vectors */
static void eigh3(const double *Cin, double *lam, double *U)
{
    double a[9];
    memcpy(a, Cin, sizeof a);
    double V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int sweep = 0; sweep < 50; sweep++) {
        double off = fabs(a[1]) + fabs(a[2]) + fabs(a[5]);
        if (off == 0.0) break;
        static const int PQ[3][2] = {{0, 1}, {0, 2}, {1, 2}};
        for (int e = 0; e < 3; e++) {
            int p = PQ[e][0], q = PQ[e][1];
            double apq = a[3 * p + q];
            if (apq == 0.0) continue;
            double app = a[3 * p + p], aqq = a[3 * q + q];
            double theta = (aqq - app) / (2.0 * apq);
            double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
            for (int k = 0; k < 3; k++) { /* A <- A J */
                double akp = a[3 * k + p], akq = a[3 * k + q];
                a[3 * k + p] = c * akp - s * akq;
                a[3 * k + q] = s * akp + c * akq;
            }
            for (int k = 0; k < 3; k++) { /* A <- J^T A */
                double apk = a[3 * p + k], aqk = a[3 * q + k];
                a[3 * p + k] = c * apk - s * aqk;
                a[3 * q + k] = s * apk + c * aqk;
            }
            a[3 * p + q] = a[3 * q + p] = 0.0;
            for (int k = 0; k < 3; k++) {
                double vkp = V[3 * k + p], vkq = V[3 * k + q];
                V[3 * k + p] = c * vkp - s * vkq;
                V[3 * k + q] = s * vkp + c * vkq;
            }
        }
    }
    int idx[3] = {0, 1, 2};
    double d[3] = {a[0], a[4], a[8]};
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2 - i; j++)
            if (d[idx[j]] > d[idx[j + 1]]) { int t = idx[j]; idx[j] = idx[j + 1]; idx[j + 1] = t; }
    for (int m = 0; m < 3; m++) {
        lam[m] = d[idx[m]];
        for (int k = 0; k < 3; k++) U[3 * k + m] = V[3 * k + idx[m]];
    }
}

/* ------------------------------------------------------------------ cluster algebra */
typedef struct { double P[9], v[3], n; } Clu;

static void unpack_cluster(const double *c, Clu *o)
{
    o->P[0] = c[0]; o->P[1] = c[1]; o->P[2] = c[2];
    o->P[3] = c[1]; o->P[4] = c[3]; o->P[5] = c[4];
    o->P[6] = c[2]; o->P[7] = c[4]; o->P[8] = c[5];
    o->v[0] = c[6]; o->v[1] = c[7]; o->v[2] = c[8];
    o->n = c[9];
}
/* PointCluster::transform, tools.hpp:450-456 */
static void clu_transform(const Clu *s, const double *R, const double *p, Clu *o)
{
    double Rv[3], RP[9], RPRt[9];
    mat3_vec(R, s->v, Rv);
    mat3_mul(R, s->P, RP);
    mat3_mulT(RP, R, RPRt);
    o->n = s->n;
    for (int i = 0; i < 3; i++) o->v[i] = Rv[i] + s->n * p[i];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            o->P[3 * i + j] = RPRt[3 * i + j] + Rv[i] * p[j] + p[i] * Rv[j] + s->n * p[i] * p[j];
}

/* merged covariance eigen-decomposition of one voxel; returns NN */
static double voxel_eig(const int64_t *voff, const int32_t *pidx, const double *clusters,
                        const double *poses, int64_t a, double *lam, double *U, double *vbar)
{
    Clu sig, c, t;
    memset(&sig, 0, sizeof sig);
    for (int64_t f = voff[a]; f < voff[a + 1]; f++) {
        const double *x = poses + 12 * (int64_t)pidx[f];
        unpack_cluster(clusters + 10 * f, &c);
        clu_transform(&c, x, x + 9, &t);
        for (int i = 0; i < 9; i++) sig.P[i] += t.P[i];
        for (int i = 0; i < 3; i++) sig.v[i] += t.v[i];
        sig.n += t.n;
    }
    double C[9];
    for (int i = 0; i < 3; i++) vbar[i] = sig.v[i] / sig.n;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) C[3 * i + j] = sig.P[3 * i + j] / sig.n - vbar[i] * vbar[j];
    eigh3(C, lam, U);
    return sig.n;
}

/* ------------------------------------------------------------------ block sinks */
typedef struct Sink {
    void (*add)(struct Sink *, int i, int j, const double *Hb /* 6x6 row-major */);
    double *H; int64_t ld;               /* dense col-major */
    int64_t cap, cnt; int64_t *keys; double *vals; int nposes;   /* hash of upper blocks */
} Sink;

static void dense_add(Sink *s, int i, int j, const double *Hb)
{
    for (int r = 0; r < 6; r++)
        for (int c = 0; c < 6; c++) s->H[(6 * (int64_t)i + r) + (6 * (int64_t)j + c) * s->ld] += Hb[6 * r + c];
}
static void hash_grow(Sink *s);
static int64_t hash_slot(Sink *s, int64_t key)
{
    uint64_t h = (uint64_t)key * 0x9E3779B97F4A7C15ull;
    int64_t m = s->cap - 1, p = (int64_t)(h >> 20) & m;
    while (s->keys[p] != -1 && s->keys[p] != key) p = (p + 1) & m;
    return p;
}
static void hash_add(Sink *s, int i, int j, const double *Hb)
{
    if (2 * (s->cnt + 1) > s->cap) hash_grow(s);
    int64_t key = (int64_t)i * s->nposes + j, p = hash_slot(s, key);
    if (s->keys[p] == -1) { s->keys[p] = key; s->cnt++; memset(s->vals + 36 * p, 0, 36 * sizeof(double)); }
    double *v = s->vals + 36 * p;
    for (int e = 0; e < 36; e++) v[e] += Hb[e];
}
static void hash_init(Sink *s, int nposes, int64_t cap)
{
    memset(s, 0, sizeof *s);
    s->add = hash_add; s->nposes = nposes; s->cap = cap;
    s->keys = malloc(cap * sizeof(int64_t));
    s->vals = malloc(cap * 36 * sizeof(double));
    for (int64_t i = 0; i < cap; i++) s->keys[i] = -1;
}
static void hash_grow(Sink *s)
{
    Sink n;
    hash_init(&n, s->nposes, s->cap * 2);
    for (int64_t i = 0; i < s->cap; i++)
        if (s->keys[i] != -1) {
            int64_t p = hash_slot(&n, s->keys[i]);
            n.keys[p] = s->keys[i]; n.cnt++;
            memcpy(n.vals + 36 * p, s->vals + 36 * i, 36 * sizeof(double));
        }
    free(s->keys); free(s->vals);
    *s = n;
}

/* ------------------------------------------------------------------ a4: acc_evaluate2 */
/* bavoxel.hpp:68-174 for voxels [head,end): upper blocks (i<=j) into sink, JacT, residual.
 * kmax_hint bounds the scratch; voxels may have any number of observers. */
static void acc_evaluate2(const int64_t *voff, const int32_t *pidx, const double *clusters,
                          const double *poses, int64_t head, int64_t end, Sink *sink, double *JacT,
                          double *residual)
{
    *residual = 0;
    int64_t kcap = 64;
    double *Auk = malloc(kcap * 18 * sizeof(double));      /* 3x6 row-major */
    double *viRiTuk = malloc(kcap * 3 * sizeof(double));
    for (int64_t a = head; a < end; a++) {
        int64_t f0 = voff[a], k = voff[a + 1] - f0;
        if (k > kcap) {
            kcap = 2 * k;
            Auk = realloc(Auk, kcap * 18 * sizeof(double));
            viRiTuk = realloc(viRiTuk, kcap * 3 * sizeof(double));
        }
        double lam[3], U[9], vBar[3];
        double NN = voxel_eig(voff, pidx, clusters, poses, a, lam, U, vBar); /* :90-103 */
        double u[3][3];
        for (int m = 0; m < 3; m++) for (int r = 0; r < 3; r++) u[m][r] = U[3 * r + m];
        const double *uk = u[0];
        double ukukT[9], umumT[9] = {0};
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) ukukT[3 * r + c] = uk[r] * uk[c];
        for (int m = 1; m < 3; m++) { /* :107-110 */
            double w = 2.0 / (lam[0] - lam[m]);
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) umumT[3 * r + c] += w * u[m][r] * u[m][c];
        }
        for (int64_t x = 0; x < k; x++) { /* :112-149 */
            int64_t f = f0 + x;
            int i = pidx[f];
            const double *Ri = poses + 12 * (int64_t)i, *pi = Ri + 9;
            Clu ci;
            unpack_cluster(clusters + 10 * f, &ci);
            double ni = ci.n, vihat[9], RiTuk[3], RiTukhat[9], PiRiTuk[3], ti_v[3];
            hat3(ci.v, vihat);
            mat3T_vec(Ri, uk, RiTuk);
            hat3(RiTuk, RiTukhat);
            mat3_vec(ci.P, RiTuk, PiRiTuk);
            double *w = viRiTuk + 3 * x;
            mat3_vec(vihat, RiTuk, w);
            for (int r = 0; r < 3; r++) ti_v[r] = pi[r] - vBar[r];
            double ukTti_v = uk[0] * ti_v[0] + uk[1] * ti_v[1] + uk[2] * ti_v[2];
            double combo1[9], combo2[3], Rv[3], hp[9];
            hat3(PiRiTuk, hp);
            for (int e = 0; e < 9; e++) combo1[e] = hp[e] + vihat[e] * ukTti_v;
            mat3_vec(Ri, ci.v, Rv);
            for (int r = 0; r < 3; r++) combo2[r] = Rv[r] + ni * ti_v[r];
            double RP[9], M[9], MH[9], Rc1[9];
            mat3_mul(Ri, ci.P, RP);
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) M[3 * r + c] = RP[3 * r + c] + ti_v[r] * ci.v[c];
            mat3_mul(M, RiTukhat, MH);
            mat3_mul(Ri, combo1, Rc1);
            double c2u = combo2[0] * uk[0] + combo2[1] * uk[1] + combo2[2] * uk[2];
            double *A = Auk + 18 * x;
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) {
                    A[6 * r + c] = (MH[3 * r + c] - Rc1[3 * r + c]) / NN;
                    A[6 * r + 3 + c] = (combo2[r] * uk[c] + (r == c ? c2u : 0.0)) / NN;
                }
            double jjt[6];
            for (int c = 0; c < 6; c++) jjt[c] = A[c] * uk[0] + A[6 + c] * uk[1] + A[12 + c] * uk[2];
            for (int c = 0; c < 6; c++) JacT[6 * (int64_t)i + c] += jjt[c];
            /* Hb = A^T umumT A */
            double WA[18], Hb[36];
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 6; c++)
                    WA[6 * r + c] = umumT[3 * r] * A[c] + umumT[3 * r + 1] * A[6 + c] + umumT[3 * r + 2] * A[12 + c];
            for (int r = 0; r < 6; r++)
                for (int c = 0; c < 6; c++) Hb[6 * r + c] = A[r] * WA[c] + A[6 + r] * WA[6 + c] + A[12 + r] * WA[12 + c];
            double T1[9], T2[9], hj[9];
            mat3_mul(RiTukhat, ci.P, T1);
            for (int e = 0; e < 9; e++) T1[e] = combo1[e] - T1[e];
            mat3_mul(T1, RiTukhat, T2);
            hat3(jjt, hj);
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) {
                    Hb[6 * r + c] += 2.0 / NN * T2[3 * r + c] - 2.0 / NN / NN * w[r] * w[c] - 0.5 * hj[3 * r + c];
                    double HRt = 2.0 / NN * (1.0 - ni / NN) * w[r] * uk[c];
                    Hb[6 * r + 3 + c] += HRt;
                    Hb[6 * (3 + c) + r] += HRt;
                    Hb[6 * (3 + r) + 3 + c] += 2.0 / NN * (ni - ni * ni / NN) * ukukT[3 * r + c];
                }
            sink->add(sink, i, i, Hb);
        }
        for (int64_t x = 0; x + 1 < k; x++) { /* :151-167 */
            int i = pidx[f0 + x];
            double ni = clusters[10 * (f0 + x) + 9];
            const double *Ai = Auk + 18 * x, *wi = viRiTuk + 3 * x;
            double WtAi[18]; /* (umumT Ai) is reused: Hb = Ai^T umumT Aj = (umumT Ai)^T Aj (umumT symmetric) */
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 6; c++)
                    WtAi[6 * r + c] = umumT[3 * r] * Ai[c] + umumT[3 * r + 1] * Ai[6 + c] + umumT[3 * r + 2] * Ai[12 + c];
            for (int64_t y = x + 1; y < k; y++) {
                int j = pidx[f0 + y];
                double nj = clusters[10 * (f0 + y) + 9];
                const double *Aj = Auk + 18 * y, *wj = viRiTuk + 3 * y;
                double Hb[36];
                for (int r = 0; r < 6; r++)
                    for (int c = 0; c < 6; c++)
                        Hb[6 * r + c] = WtAi[r] * Aj[c] + WtAi[6 + r] * Aj[6 + c] + WtAi[12 + r] * Aj[12 + c];
                for (int r = 0; r < 3; r++)
                    for (int c = 0; c < 3; c++) {
                        Hb[6 * r + c] += -2.0 / NN / NN * wi[r] * wj[c];
                        Hb[6 * r + 3 + c] += -2.0 * nj / NN / NN * wi[r] * uk[c];
                        Hb[6 * (3 + r) + c] += -2.0 * ni / NN / NN * uk[r] * wj[c];
                        Hb[6 * (3 + r) + 3 + c] += -2.0 * ni * nj / NN / NN * ukukT[3 * r + c];
                    }
                sink->add(sink, i, j, Hb);
            }
        }
        *residual += lam[0]; /* :168 */
    }
    free(Auk);
    free(viRiTuk);
}

static void slices(int64_t g_size, int *t_out, int64_t *head, int64_t *end) /* bavoxel.hpp:614-624 */
{
    int t = g_size < THD_NUM ? 1 : THD_NUM;
    double part = 1.0 * (double)g_size / t;
    for (int i = 0; i < t; i++) { head[i] = (int64_t)(int)(part * i); end[i] = (int64_t)(int)(part * (i + 1)); }
    *t_out = t;
}

/* ------------------------------------------------------------------ exported API */

/* a5/a7: sum of lambda_min over all voxels (not averaged); reference is single-threaded
 * (bavoxel.hpp:176-203); nthreads>1 only changes the summation grouping. */
int bo_cost(int n_poses, int64_t V, const int64_t *voff, const int32_t *pidx, const double *clusters,
            const double *poses, int nthreads, double *cost_sum)
{
    (void)n_poses;
    double tot = 0;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) reduction(+ : tot) schedule(static)
    for (int64_t a = 0; a < V; a++) {
        double lam[3], U[9], vb[3];
        voxel_eig(voff, pidx, clusters, poses, a, lam, U, vb);
        tot += lam[0];
    }
    *cost_sum = tot;
    return 0;
}

/* per-voxel eigenvalues (ascending), [V][3] */
int bo_voxel_lambdas(int64_t V, const int64_t *voff, const int32_t *pidx, const double *clusters,
                     const double *poses, double *lam_out)
{
#pragma omp parallel for schedule(static)
    for (int64_t a = 0; a < V; a++) {
        double U[9], vb[3];
        voxel_eig(voff, pidx, clusters, poses, a, lam_out + 3 * a, U, vb);
    }
    return 0;
}

/* a6: divide_thread with the reference's memory scheme: 16 thread-local dense (6N)^2
 * Hessians summed serially in thread order, then mirrored.  H col-major [6N x 6N], g [6N],
 * cost_avg = residual / V.  Small/medium N only (16 * (6N)^2 * 8 bytes). */
int bo_eval_dense(int n_poses, int64_t V, const int64_t *voff, const int32_t *pidx,
                  const double *clusters, const double *poses, double *H, double *g, double *cost_avg)
{
    int64_t n = 6 * (int64_t)n_poses;
    int T;
    int64_t head[THD_NUM], end[THD_NUM];
    slices(V, &T, head, end);
    double *Ht[THD_NUM], *gt[THD_NUM], rt[THD_NUM];
#pragma omp parallel for num_threads(T) schedule(static, 1)
    for (int t = 0; t < T; t++) {
        Ht[t] = calloc((size_t)(n * n), sizeof(double));
        gt[t] = calloc((size_t)n, sizeof(double));
        Sink s;
        memset(&s, 0, sizeof s);
        s.add = dense_add; s.H = Ht[t]; s.ld = n;
        acc_evaluate2(voff, pidx, clusters, poses, head[t], end[t], &s, gt[t], &rt[t]);
    }
    memset(H, 0, (size_t)(n * n) * sizeof(double));
    memset(g, 0, (size_t)n * sizeof(double));
    double residual = 0;
    for (int t = 0; t < T; t++) { /* :626-633 */
        for (int64_t e = 0; e < n * n; e++) H[e] += Ht[t][e];
        for (int64_t e = 0; e < n; e++) g[e] += gt[t][e];
        residual += rt[t];
        free(Ht[t]); free(gt[t]);
    }
    for (int64_t bi = 1; bi < n_poses; bi++) /* :171-173 mirror strictly-upper blocks */
        for (int64_t bj = 0; bj < bi; bj++)
            for (int r = 0; r < 6; r++)
                for (int c = 0; c < 6; c++)
                    H[(6 * bi + r) + (6 * bj + c) * n] = H[(6 * bj + c) + (6 * bi + r) * n];
    *cost_avg = residual / (double)V;
    return 0;
}

/* BASELINE.md variant (D), what the reference does with the dense Hessian between divide_thread and the solver
 * (bavoxel.hpp:692-703): D = diag(Hess) as a dense matrix, HessuD = Hess + u D as a third one, then a scan of all
 * (6N)^2 entries, row by row over the column-major matrix, pushing the non-zero ones onto a triplet list
 * (Eigen::Triplet<double>: two ints and a double; std::vector growth by doubling).  H: col-major [n x n].
 * Returns the number of triplets (-1: out of memory). */
int64_t bo_dense_to_triplets(int64_t n, const double *H, double u)
{
    double *D = calloc((size_t)(n * n), sizeof(double));
    double *HuD = malloc((size_t)(n * n) * sizeof(double));
    if (!D || !HuD) { free(D); free(HuD); return -1; }
    for (int64_t a = 0; a < n; a++) D[a + a * n] = H[a + a * n];          /* :692 */
    for (int64_t e = 0; e < n * n; e++) HuD[e] = H[e] + u * D[e];          /* :693 */
    typedef struct { int r, c; double v; } Trip;
    int64_t cap = 1, cnt = 0;
    Trip *tl = malloc(sizeof(Trip));
    for (int64_t a = 0; a < n && tl; a++)                                  /* :697-703 */
        for (int64_t b = 0; b < n; b++) {
            const double v = HuD[a + b * n];
            if (v != 0) {
                if (cnt == cap) {
                    Trip *t2 = realloc(tl, (size_t)(2 * cap) * sizeof(Trip));
                    if (!t2) { free(tl); tl = NULL; break; }
                    tl = t2; cap *= 2;
                }
                tl[cnt].r = (int)a; tl[cnt].c = (int)b; tl[cnt].v = v; cnt++;
            }
        }
    free(D); free(HuD);
    if (!tl) return -1;
    free(tl);
    return cnt;
}

/* a6, sparse-honest variant (BASELINE.md variant S): same math and slicing, thread-local
 * hash maps of upper 6x6 blocks merged in thread order.  Outputs the merged block list
 * (bi<=bj, 6x6 row-major) if blocks != NULL and cap is large enough; *nblocks always set. */
int bo_eval_sparse(int n_poses, int64_t V, const int64_t *voff, const int32_t *pidx,
                   const double *clusters, const double *poses, int nthreads, int64_t cap, int32_t *bi,
                   int32_t *bj, double *blocks, int64_t *nblocks, double *g, double *cost_avg)
{
    int64_t n = 6 * (int64_t)n_poses;
    int T;
    int64_t head[THD_NUM], end[THD_NUM];
    slices(V, &T, head, end);
    Sink st[THD_NUM];
    double *gt[THD_NUM], rt[THD_NUM];
    if (nthreads < 1) nthreads = T;
#pragma omp parallel for num_threads(nthreads < T ? nthreads : T) schedule(static, 1)
    for (int t = 0; t < T; t++) {
        hash_init(&st[t], n_poses, 1 << 12);
        gt[t] = calloc((size_t)n, sizeof(double));
        acc_evaluate2(voff, pidx, clusters, poses, head[t], end[t], &st[t], gt[t], &rt[t]);
    }
    memset(g, 0, (size_t)n * sizeof(double));
    double residual = 0;
    Sink tot;
    hash_init(&tot, n_poses, 1 << 14);
    for (int t = 0; t < T; t++) {
        for (int64_t p = 0; p < st[t].cap; p++)
            if (st[t].keys[p] != -1)
                hash_add(&tot, (int)(st[t].keys[p] / n_poses), (int)(st[t].keys[p] % n_poses), st[t].vals + 36 * p);
        for (int64_t e = 0; e < n; e++) g[e] += gt[t][e];
        residual += rt[t];
        free(st[t].keys); free(st[t].vals); free(gt[t]);
    }
    *nblocks = tot.cnt;
    int rc = 0;
    if (blocks) {
        if (tot.cnt > cap) rc = -1;
        else {
            int64_t o = 0;
            for (int64_t p = 0; p < tot.cap; p++)
                if (tot.keys[p] != -1) {
                    bi[o] = (int32_t)(tot.keys[p] / n_poses);
                    bj[o] = (int32_t)(tot.keys[p] % n_poses);
                    memcpy(blocks + 36 * o, tot.vals + 36 * p, 36 * sizeof(double));
                    o++;
                }
        }
    }
    free(tot.keys); free(tot.vals);
    *cost_avg = residual / (double)V;
    return rc;
}

/* SimplicialLDLT stand-in: unpivoted LDL^T of the LOWER triangle of A (col-major, lda=n,
 * overwritten), then solves A x = b.  Right-looking, column at a time, OpenMP over the
 * trailing columns. */
int bo_ldlt_solve_dense(int64_t n, double *A, const double *b, double *x, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    double *w = malloc((size_t)n * sizeof(double));
    for (int64_t j = 0; j < n; j++) {
        double d = A[j + j * n];
        if (d == 0.0 || !isfinite(d)) { free(w); return 1; }
        for (int64_t r = j + 1; r < n; r++) { w[r] = A[r + j * n]; A[r + j * n] = w[r] / d; }
#pragma omp parallel for num_threads(nthreads) schedule(static) if (n - j > 256)
        for (int64_t c = j + 1; c < n; c++) {
            double lc = A[c + j * n];
            if (lc == 0.0) continue;
            double *col = A + c * n;
            for (int64_t r = c; r < n; r++) col[r] -= w[r] * lc;
        }
    }
    for (int64_t i = 0; i < n; i++) x[i] = b[i];
    for (int64_t j = 0; j < n; j++) { double xj = x[j]; for (int64_t r = j + 1; r < n; r++) x[r] -= A[r + j * n] * xj; }
    for (int64_t j = 0; j < n; j++) x[j] /= A[j + j * n];
    for (int64_t j = n - 1; j >= 0; j--) { double s = x[j]; for (int64_t r = j + 1; r < n; r++) s -= A[r + j * n] * x[r]; x[j] = s; }
    free(w);
    return 0;
}

/* Same for LAPACK-style lower band storage AB[(r-c) + c*ldab], 0 <= r-c <= bw, ldab >= bw+1. */
int bo_ldlt_solve_band(int64_t n, int64_t bw, double *AB, int64_t ldab, const double *b, double *x, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
#define BA(r, c) AB[((r) - (c)) + (c) * ldab]
    double *w = malloc((size_t)(bw + 1) * sizeof(double));
    for (int64_t j = 0; j < n; j++) {
        double d = BA(j, j);
        if (d == 0.0 || !isfinite(d)) { free(w); return 1; }
        int64_t rmax = j + bw < n - 1 ? j + bw : n - 1;
        for (int64_t r = j + 1; r <= rmax; r++) { w[r - j] = BA(r, j); BA(r, j) = w[r - j] / d; }
#pragma omp parallel for num_threads(nthreads) schedule(static) if (rmax - j > 256)
        for (int64_t c = j + 1; c <= rmax; c++) {
            double lc = BA(c, j);
            if (lc == 0.0) continue;
            for (int64_t r = c; r <= rmax; r++) BA(r, c) -= w[r - j] * lc;
        }
    }
    for (int64_t i = 0; i < n; i++) x[i] = b[i];
    for (int64_t j = 0; j < n; j++) {
        double xj = x[j];
        int64_t rmax = j + bw < n - 1 ? j + bw : n - 1;
        for (int64_t r = j + 1; r <= rmax; r++) x[r] -= BA(r, j) * xj;
    }
    for (int64_t j = 0; j < n; j++) x[j] /= BA(j, j);
    for (int64_t j = n - 1; j >= 0; j--) {
        double s = x[j];
        int64_t rmax = j + bw < n - 1 ? j + bw : n - 1;
        for (int64_t r = j + 1; r <= rmax; r++) s -= BA(r, j) * x[r];
        x[j] = s;
    }
#undef BA
    free(w);
    return 0;
}

/* bavoxel.hpp:722-727 */
int bo_retract(int n_poses, const double *poses, const double *dx, double *out)
{
    for (int j = 0; j < n_poses; j++) {
        const double *R = poses + 12 * j, *p = R + 9;
        double E[9];
        exp_so3(dx + 6 * j, E);
        mat3_mul(R, E, out + 12 * j);
        for (int r = 0; r < 3; r++) out[12 * j + 9 + r] = p[r] + dx[6 * j + 3 + r];
    }
    return 0;
}

/* a8: BALM2::damping_iter (bavoxel.hpp:662-767) with dense H (small/medium N).
 * trace rows: [it, residual1, residual2, u, v, q, q1, accepted, evaluated] (9 doubles). */
int bo_damping_iter(int n_poses, int64_t V, const int64_t *voff, const int32_t *pidx,
                    const double *clusters, double *poses, int max_iter, double u, double v,
                    double rel_tol, double *trace, int *n_trace)
{
    int64_t n = 6 * (int64_t)n_poses;
    double *Hess = malloc((size_t)(n * n) * sizeof(double)), *HuD = malloc((size_t)(n * n) * sizeof(double));
    double *JacT = malloc(n * sizeof(double)), *rhs = malloc(n * sizeof(double)), *dxi = malloc(n * sizeof(double));
    double *xt = malloc((size_t)n_poses * 12 * sizeof(double));
    double residual1 = 0, residual2 = 0, q;
    int is_calc_hess = 1, rows = 0, rc = 0;
    int nth = omp_get_max_threads();
    for (int it = 0; it < max_iter; it++) {
        int evaluated = is_calc_hess;
        if (is_calc_hess) bo_eval_dense(n_poses, V, voff, pidx, clusters, poses, Hess, JacT, &residual1);
        memcpy(HuD, Hess, (size_t)(n * n) * sizeof(double));
        for (int64_t a = 0; a < n; a++) { HuD[a + a * n] += u * Hess[a + a * n]; rhs[a] = -JacT[a]; }
        if (bo_ldlt_solve_dense(n, HuD, rhs, dxi, nth)) { rc = 1; break; }
        bo_retract(n_poses, poses, dxi, xt);
        double q1 = 0;
        for (int64_t a = 0; a < n; a++) q1 += dxi[a] * (u * Hess[a + a * n] * dxi[a] - JacT[a]);
        q1 *= 0.5;
        double c2;
        bo_cost(n_poses, V, voff, pidx, clusters, xt, 1, &c2);
        residual2 = c2 / (double)V;
        q1 /= (double)V;
        q = residual1 - residual2;
        double *row = trace + 9 * rows++;
        row[0] = it; row[1] = residual1; row[2] = residual2; row[3] = u; row[4] = v; row[5] = q; row[6] = q1;
        row[7] = q > 0; row[8] = evaluated;
        if (q > 0) {
            memcpy(poses, xt, (size_t)n_poses * 12 * sizeof(double));
            q = q / q1;
            v = 2;
            q = 1 - pow(2 * q - 1, 3);
            u *= (q < (1.0 / 3.0) ? (1.0 / 3.0) : q);
            is_calc_hess = 1;
        } else {
            u = u * v;
            v = 2 * v;
            is_calc_hess = 0;
        }
        if (fabs(residual1 - residual2) / residual1 < rel_tol) break;
    }
    *n_trace = rows;
    free(Hess); free(HuD); free(JacT); free(rhs); free(dxi); free(xt);
    return rc;
}

/* ------------------------------------------------------------------ a8 at the BASELINE.json sizes
 * BALM2::damping_iter (bavoxel.hpp:662-767) for systems whose dense (6N)^2 Hessian is out of reach (C3: 12 000^2):
 * the same loop as bo_damping_iter with the "sparse-honest" evaluation (bo_eval_sparse: thread-local hash maps of pose
 * blocks merged in thread order) and the unpivoted LDL^T in lower BAND storage under a caller-given pose order
 * (iperm[caller pose] = position; any permutation is valid -- it only changes the rounding of the factorisation).
 * The hash maps see the blocks at (pidx[x], pidx[y]) for x < y inside a voxel, whichever is larger.
 * times[3] (may be NULL) accumulates wall seconds of {evaluation, damped solve, cost-only pass}; eval_threads is passed to
 * bo_eval_sparse (the reference runs 16 std::threads), the cost pass is single-threaded as in the reference (:176-203,731),
 * the band solve uses solve_threads.
 * Returns 0, 1 (zero / non-finite pivot) or -1 (the permutation's bandwidth exceeds bw_blocks). */
int bo_damping_iter_band(int n_poses, int64_t V, const int64_t *voff, const int32_t *pidx,
                         const double *clusters, double *poses, const int32_t *iperm, int bw_blocks,
                         int max_iter, double u, double v, double rel_tol, int eval_threads,
                         int solve_threads, double *trace, int *n_trace, double *times)
{
    int64_t n = 6 * (int64_t)n_poses;
    int64_t bw = 6 * (int64_t)bw_blocks + 5;
    if (bw > n - 1) bw = n - 1;
    int64_t ldab = bw + 1;
    double *Hb = malloc((size_t)(ldab * n) * sizeof(double)), *HuD = malloc((size_t)(ldab * n) * sizeof(double));
    double *JacT = malloc(n * sizeof(double)), *gperm = malloc(n * sizeof(double)), *rhs = malloc(n * sizeof(double));
    double *dxp = malloc(n * sizeof(double)), *dxi = malloc(n * sizeof(double));
    double *xt = malloc((size_t)n_poses * 12 * sizeof(double));
    double residual1 = 0, residual2 = 0, q;
    int is_calc_hess = 1, rows = 0, rc = 0;
    int64_t cap = 0;
    int32_t *bi = NULL, *bj = NULL;
    double *blocks = NULL;
    if (times) times[0] = times[1] = times[2] = 0.0;
    for (int it = 0; it < max_iter && rc == 0; it++) {
        int evaluated = is_calc_hess;
        if (is_calc_hess) {
            double t0 = omp_get_wtime();
            int64_t nb = 0;
            bo_eval_sparse(n_poses, V, voff, pidx, clusters, poses, eval_threads, 0, NULL, NULL, NULL, &nb, JacT, &residual1);
            if (nb > cap) {
                cap = nb + nb / 8;
                bi = realloc(bi, cap * sizeof(int32_t)); bj = realloc(bj, cap * sizeof(int32_t));
                blocks = realloc(blocks, (size_t)cap * 36 * sizeof(double));
            }
            bo_eval_sparse(n_poses, V, voff, pidx, clusters, poses, eval_threads, cap, bi, bj, blocks, &nb, JacT, &residual1);
            memset(Hb, 0, (size_t)(ldab * n) * sizeof(double));
            for (int64_t b = 0; b < nb && rc == 0; b++) {
                const int64_t I = iperm[bi[b]], J = iperm[bj[b]];
                const double *B = blocks + 36 * b;
                if ((I > J ? I - J : J - I) > bw_blocks) { rc = -1; break; }
                for (int r = 0; r < 6; r++)
                    for (int c = 0; c < 6; c++) {
                        const int64_t R = 6 * I + r, C = 6 * J + c;
                        if (I == J) { if (r >= c) Hb[(R - C) + C * ldab] += B[6 * r + c]; }
                        else if (R > C) Hb[(R - C) + C * ldab] += B[6 * r + c];
                        else Hb[(C - R) + R * ldab] += B[6 * r + c];
                    }
            }
            for (int64_t a = 0; a < n; a++) gperm[6 * (int64_t)iperm[a / 6] + a % 6] = JacT[a];
            if (times) times[0] += 0.5 * (omp_get_wtime() - t0); /* the list is produced twice (count, then fill): one pass counted */
            if (rc) break;
        }
        double t1 = omp_get_wtime();
        memcpy(HuD, Hb, (size_t)(ldab * n) * sizeof(double));
        for (int64_t a = 0; a < n; a++) { HuD[a * ldab] += u * Hb[a * ldab]; rhs[a] = -gperm[a]; }
        if (bo_ldlt_solve_band(n, bw, HuD, ldab, rhs, dxp, solve_threads)) { rc = 1; break; }
        if (times) times[1] += omp_get_wtime() - t1;
        for (int64_t a = 0; a < n; a++) dxi[a] = dxp[6 * (int64_t)iperm[a / 6] + a % 6];
        bo_retract(n_poses, poses, dxi, xt);
        double q1 = 0;
        for (int64_t a = 0; a < n; a++) q1 += dxp[a] * (u * Hb[a * ldab] * dxp[a] - gperm[a]);
        q1 *= 0.5;
        double c2, t2 = omp_get_wtime();
        bo_cost(n_poses, V, voff, pidx, clusters, xt, 1, &c2);
        if (times) times[2] += omp_get_wtime() - t2;
        residual2 = c2 / (double)V;
        q1 /= (double)V;
        q = residual1 - residual2;
        double *row = trace + 9 * rows++;
        row[0] = it; row[1] = residual1; row[2] = residual2; row[3] = u; row[4] = v; row[5] = q; row[6] = q1;
        row[7] = q > 0; row[8] = evaluated;
        if (q > 0) {
            memcpy(poses, xt, (size_t)n_poses * 12 * sizeof(double));
            q = q / q1;
            v = 2;
            q = 1 - pow(2 * q - 1, 3);
            u *= (q < (1.0 / 3.0) ? (1.0 / 3.0) : q);
            is_calc_hess = 1;
        } else {
            u = u * v;
            v = 2 * v;
            is_calc_hess = 0;
        }
        if (fabs(residual1 - residual2) / residual1 < rel_tol) break;
    }
    *n_trace = rows;
    free(Hb); free(HuD); free(JacT); free(gperm); free(rhs); free(dxp); free(dxi); free(xt);
    free(bi); free(bj); free(blocks);
    return rc;
}
