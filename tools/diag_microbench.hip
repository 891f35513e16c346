// tools/diag_microbench.hip -- the 64 x 64 diagonal-block factorisation of the band LDL^T (csrc/ldlt_diag.h) on its own:
// correctness against a host LDL^T (d, and A (G D G^T) = I) on random symmetric indefinite blocks, full and short (nbe < 64),
// and its duration in shader clock ticks (s_memtime) with the load / factor / store phases apart.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -o tools/diag_microbench tools/diag_microbench.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>
#include "../global-lvba_amd/csrc/lvba_internal.h"

namespace lvba {
typedef double d4 __attribute__((ext_vector_type(4)));
#define LVBA_TS 80
#define LVBA_X_SENTINEL 0x7ff4dead5eed0001ULL
#ifdef LVBA_DIAG_HEADER // (an older form of the header, for A / B runs on the same box)
#include LVBA_DIAG_HEADER
#else
#include "../global-lvba_amd/csrc/ldlt_diag.h"
#endif
#ifndef LVBA_K1B_DVS
#define LVBA_K1B_DVS (64 * LVBA_W1S + 256 + 16 * LVBA_Z1S) // round 5's layout
#endif

__global__ __launch_bounds__(256, 2) void diag_stamp_kernel(LdltMat M, int nbe, double *__restrict__ G, double *__restrict__ dvec,
                                                            int *__restrict__ status, long long *__restrict__ stamps, int64_t sA)
{
    __shared__ double lds[2 * 64 * LVBA_TS];
    static_assert(LVBA_K1B_LDS <= 2 * 64 * LVBA_TS, "LDS budget");
    M.a += sA * blockIdx.x; G += 4096 * blockIdx.x; dvec += 64 * blockIdx.x;
    const long long t0 = __builtin_readcyclecounter();
    diag_blocked_load(lds, M, 0, nbe);
    const long long t1 = __builtin_readcyclecounter();
    diag_blocked_factor(lds, nbe, status);
    const long long t2 = __builtin_readcyclecounter();
    const double *W = lds, *dvs = lds + LVBA_K1B_DVS;
    const int tid = threadIdx.x;
    if (tid < nbe) dvec[tid] = dvs[tid];
    for (int e = tid; e < 4096; e += 256) G[e] = W[(e & 63) * LVBA_W1S + 64 + (e >> 6)];
    __syncthreads();
    const long long t3 = __builtin_readcyclecounter();
    if (tid == 0) { stamps[4 * blockIdx.x] = t1 - t0; stamps[4 * blockIdx.x + 1] = t2 - t1; stamps[4 * blockIdx.x + 2] = t3 - t2; }
}
} // namespace lvba

static double check(const std::vector<double> &A, const double *G, const double *d, int nbe, double *derr)
{
    // host LDL^T of the leading nbe x nbe block (unpivoted), d compared; then A (G D G^T) = I on that block
    const int n = nbe;
    std::vector<double> L(A.begin(), A.end()), dh(n);
    for (int j = 0; j < n; ++j) {
        double dj = L[j + 64 * j];
        for (int k = 0; k < j; ++k) dj -= L[j + 64 * k] * L[j + 64 * k] * dh[k];
        dh[j] = dj;
        for (int i = j + 1; i < n; ++i) {
            double v = L[i + 64 * j];
            for (int k = 0; k < j; ++k) v -= L[i + 64 * k] * L[j + 64 * k] * dh[k];
            L[i + 64 * j] = v / dj;
        }
    }
    double de = 0.0;
    for (int j = 0; j < n; ++j) de = std::max(de, std::fabs(d[j] - dh[j]) / std::fabs(dh[j]));
    *derr = de;
    // Ainv = G D G^T with G[m][c] row-major (m = row of the inverse factor, c = column)
    std::vector<double> Ai(64 * 64, 0.0);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0.0;
            for (int c = 0; c < n; ++c) s += G[i * 64 + c] * d[c] * G[j * 64 + c];
            Ai[i + 64 * j] = s;
        }
    double worst = 0.0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0.0;
            for (int k = 0; k < n; ++k) {
                const double a = k >= i ? A[k + 64 * i] : A[i + 64 * k]; // symmetric: the lower triangle is stored
                s += a * Ai[k + 64 * j];
            }
            worst = std::max(worst, std::fabs(s - (i == j ? 1.0 : 0.0)));
        }
    return worst;
}

int main(int argc, char **argv)
{
    const int nblk = argc > 1 ? atoi(argv[1]) : 256;
    int rc = 0;
    for (int nbe : {64, 40, 17, 64}) {
        std::vector<double> A((size_t)nblk * 4096);
        srand(1234 + nbe);
        for (int b = 0; b < nblk; ++b) {
            double *a = A.data() + (size_t)b * 4096;
            // diagonally dominant-ish symmetric matrix with mixed-sign pivots (the exact Hessian may be indefinite)
            for (int j = 0; j < 64; ++j)
                for (int i = j; i < 64; ++i) {
                    const double r = rand() / (double)RAND_MAX - 0.5;
                    a[i + 64 * j] = (i == j) ? ((b & 1) && (j % 7 == 3) ? -1.0 : 1.0) * (20.0 + 10.0 * r) : r;
                    if (i != j) a[j + 64 * i] = 0.0;
                }
        }
        double *dA, *dG, *dd;
        int *ds;
        long long *dst;
        hipMalloc(&dA, A.size() * 8); hipMalloc(&dG, (size_t)nblk * 4096 * 8); hipMalloc(&dd, (size_t)nblk * 64 * 8);
        hipMalloc(&ds, 4); hipMalloc(&dst, (size_t)nblk * 32);
        hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
        hipMemset(ds, 0, 4);
        lvba::LdltMat M{dA, 64, 64, 63, 0};
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        float ms = 0.f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(lvba::diag_stamp_kernel, dim3(nblk), dim3(256), 0, 0, M, nbe, dG, dd, ds, dst, (int64_t)4096);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        std::vector<double> G((size_t)nblk * 4096), d((size_t)nblk * 64);
        std::vector<long long> st((size_t)nblk * 4);
        int status = 0;
        hipMemcpy(G.data(), dG, G.size() * 8, hipMemcpyDeviceToHost);
        hipMemcpy(d.data(), dd, d.size() * 8, hipMemcpyDeviceToHost);
        hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost);
        hipMemcpy(&status, ds, 4, hipMemcpyDeviceToHost);
        double worst = 0.0, dworst = 0.0;
        for (int b = 0; b < std::min(nblk, 16); ++b) {
            double de;
            const std::vector<double> Ab(A.begin() + (size_t)b * 4096, A.begin() + (size_t)(b + 1) * 4096);
            worst = std::max(worst, check(Ab, G.data() + (size_t)b * 4096, d.data() + (size_t)b * 64, nbe, &de));
            dworst = std::max(dworst, de);
        }
        long long c_load = 0, c_fac = 0, c_store = 0;
        for (int b = 0; b < nblk; ++b) { c_load += st[4 * b]; c_fac += st[4 * b + 1]; c_store += st[4 * b + 2]; }
        const bool ok = worst < 1e-11 && dworst < 1e-12 && status == 0;
        if (!ok) rc = 1;
        printf("nbe %2d  %d blocks: |A Ainv - I| %.2e  d rel %.2e  status %d  %s   ticks (mean per block): load %lld | factor %lld | store %lld   launch %.1f us\n",
               nbe, nblk, worst, dworst, status, ok ? "ok" : "FAILED", c_load / nblk, c_fac / nblk, c_store / nblk, ms * 1e3);
        hipFree(dA); hipFree(dG); hipFree(dd); hipFree(ds); hipFree(dst);
    }
    return rc;
}
