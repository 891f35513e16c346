// tools/step_microbench.hip -- standalone timing of ONE launch of the LDL^T step kernel (development tool, not product): the
// factorisation workgroups alone, the trailing-update workgroups alone (one tile per workgroup / persistent), and both.
// Includes ldlt.hip directly.  usage: step_microbench [n=12000] [bw=2597]
#include "../global-lvba_amd/csrc/ldlt.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace lvba;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <class F> static float time_us(hipStream_t s, int reps, F f)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(a, s));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps * 1e3f;
}


__global__ void xcc_kernel(int *out) { if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15; }

int main(int argc, char **argv)
{
    {
        int *d, h[64]; CK(hipMalloc((void **)&d, 64 * 4));
        hipLaunchKernelGGL(xcc_kernel, dim3(64), dim3(256), 0, 0, d);
        CK(hipMemcpy(h, d, 64 * 4, hipMemcpyDeviceToHost));
        printf("XCC of blocks 0..31:"); for (int i = 0; i < 32; ++i) printf(" %d", h[i]); printf("\n");
    }
    const int64_t n = argc > 1 ? atoll(argv[1]) : 12000, bw = argc > 2 ? atoll(argv[2]) : 2597;
    const int64_t ldab = bw + LVBA_NB + 64;
    LdltMat M; M.n = n; M.ld = ldab - 1; M.bw = bw;
    const int64_t sA = ldab * (n + 1), sW = ldlt_ws_one(n, bw);
    const size_t adoubles = (size_t)(2 * sA + 65 * ldab);
    std::vector<double> hA(adoubles, 0.0);
    srand(1);
    for (int q = 0; q < 2; ++q)
        for (int64_t c = 0; c < n; ++c) {
            for (int64_t o = 1; o <= bw && c + o < n; ++o) hA[q * sA + o + c * ldab] = 0.02 * (rand() / (double)RAND_MAX - 0.5);
            hA[q * sA + c * ldab] = 2.0 + bw * 0.01;
        }
    CK(hipMalloc((void **)&M.a, adoubles * 8)); CK(hipMemcpy(M.a, hA.data(), adoubles * 8, hipMemcpyHostToDevice));
    double *work; CK(hipMalloc((void **)&work, (2 * sW + n + 64) * 8)); CK(hipMemset(work, 0, (2 * sW + n + 64) * 8));
    const int64_t nsteps = ldlt_num_panels(n), ldz = ldz_for(n, bw);
    double *Gall = work, *dvec = Gall + nsteps * 4096, *b = dvec + n, *bacc = b + n;
    double *Zbuf[4] = {bacc + n, bacc + n + ldz * LVBA_NB, bacc + n + 2 * ldz * LVBA_NB, bacc + n + 3 * ldz * LVBA_NB};
    int *status; CK(hipMalloc((void **)&status, 4)); CK(hipMemset(status, 0, 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    struct Geo { int64_t k, w0, rend, T; int nbe; };
    auto geom = [&](int64_t st) { Geo q; q.k = st * 64; q.nbe = 64; q.w0 = q.k + 64; q.rend = q.k + 64 + bw; q.T = (q.rend - q.w0 + 63) / 64; return q; };
    const Geo qe = geom(10), qo = geom(11), q2 = geom(12);
    const int64_t Tb = qo.T - 1, total = Tb * (Tb + 1) / 2;
    printf("n=%lld bw=%lld T=%lld bulk tiles of one panel=%lld (x2 problems)\n", (long long)n, (long long)bw, (long long)qo.T, (long long)total);
    // big: 128 x 64 update tiles of the tile columns holding tiles [t0, t1); else one 64 x 64 tile per workgroup
    auto launch = [&](bool big, int T2, int64_t t0, int64_t t1, bool pair, int ny) {
        const int nx = 0;
        int64_t ca = t0, cb = t1, nbu = t1 - t0;
        if (big && t1 > t0) {
            ca = 0; cb = Tb;
            while (ca < Tb && col_start(ca, Tb) < t0) ++ca;
            while (cb > ca && col_start(cb, Tb) > t1) --cb;
            nbu = 0;
            for (int64_t c = ca; c < cb; ++c) nbu += pair_col_items(c, Tb);
        }
        if (big)
            hipLaunchKernelGGL(ldlt_step_kernel<true>, dim3((unsigned)((T2 + nx + nbu) * ny)), dim3(256), 0, s, M, q2.k, q2.nbe, q2.w0, q2.rend, T2, Gall + 12 * 4096,
                               dvec, Zbuf[0], b, status, qo.k, qo.nbe, qo.w0, qo.rend, (const double *)Zbuf[3], ldz, sA, sW, qe.k, qe.nbe, qe.w0, qe.rend,
                               pair ? (const double *)Zbuf[2] : (const double *)nullptr, ca, cb, ny);
        else
            hipLaunchKernelGGL(ldlt_step_kernel<false>, dim3((unsigned)((T2 + nx + nbu) * ny)), dim3(256), 0, s, M, q2.k, q2.nbe, q2.w0, q2.rend, T2, Gall + 12 * 4096,
                               dvec, Zbuf[0], b, status, qo.k, qo.nbe, qo.w0, qo.rend, (const double *)Zbuf[3], ldz, sA, sW, qe.k, qe.nbe, qe.w0, qe.rend,
                               pair ? (const double *)Zbuf[2] : (const double *)nullptr, ca, cb, ny);
    };
    const int T2 = (int)q2.T;
    int64_t cs = 1;
    while (cs < Tb && 2 * col_start(cs, Tb) < total) ++cs;
    const int64_t t_half = col_start(cs, Tb);
    if (argc > 3) { // counter mode (under rocprofv3 --pmc): five launches of ONE configuration, update workgroups only
        const int mode = atoi(argv[3]); // 0: rank 64, 2 problems; 1: rank 128 first half, 2 problems; 2: rank 128 first half, 1 problem
        for (int i = 0; i < 5; ++i) launch(true, 0, 0, mode == 0 ? total : t_half, mode != 0, mode == 2 ? 1 : 2);
        CK(hipStreamSynchronize(s));
        return 0;
    }
    for (int ny = 1; ny <= 2; ++ny) {
        printf("---- %d problem(s) per launch\n", ny);
        printf("factorisation workgroups alone (%d)                     %7.2f us\n", T2 * ny, time_us(s, 200, [&] { launch(false, T2, 0, 0, false, ny); }));
        for (int pair = 0; pair < 3; ++pair) { // 0: rank 64, all tiles; 1 / 2: rank 128, first / second half of the tile columns
            const int64_t t0 = pair == 2 ? t_half : 0, t1 = pair == 1 ? t_half : total;
            const double gf = (double)(t1 - t0) * ny * (pair ? 2 : 1) * 2.0 * 64 * 64 * 64 * 1e-9;
            const char *nm = pair == 0 ? "rank-64  " : pair == 1 ? "rank-128a" : "rank-128b";
            for (int big = 0; big < 2; ++big) {
                float t = time_us(s, 100, [&] { launch(big, 0, t0, t1, pair, ny); });
                printf("%s update alone, %s tiles     %7.2f us  %5.1f TFLOP/s\n", nm, big ? "128 x 64" : " 64 x 64", t, gf / t * 1e3);
                t = time_us(s, 100, [&] { launch(big, T2, t0, t1, pair, ny); });
                printf("%s factorisation + %s tiles   %7.2f us\n", nm, big ? "128 x 64" : " 64 x 64", t);
            }
        }
    }
    return 0;
}
