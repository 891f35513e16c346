// tools/bcr_microbench.hip -- standalone timing of the block-cyclic-reduction kernels (development tool, not product).
// Includes bcr.hip directly.  usage: bcr_microbench [n_cams=2000] [band_blocks=3]
#define LVBA_BCR_TIMING
#include "../global-lvba_amd/csrc/bcr.hip"
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
    return 1e3f * ms / reps;
}

int main(int argc, char **argv)
{
    const int M = argc > 1 ? atoi(argv[1]) : 2000, Bb = argc > 2 ? atoi(argv[2]) : 3;
    const int64_t Bb1 = Bb + 1, n = 6 * (int64_t)M;
    std::vector<double> H((size_t)M * Bb1 * 36, 0.0), g(n, 1.0);
    srand(3);
    for (int64_t J = 0; J < M; ++J)
        for (int dI = 0; dI <= Bb && J + dI < M; ++dI)
            for (int e = 0; e < 36; ++e) {
                const int c = e / 6, r = e % 6;
                double v = 0.05 * (rand() / (double)RAND_MAX - 0.5);
                if (dI == 0) v = r == c ? 4.0 : (r > c ? v : 0.0);
                H[(J * Bb1 + dI) * 36 + e] = v;
            }
    double *dH, *dg, *du, *dx, *work; int *status;
    CK(hipMalloc((void **)&dH, H.size() * 8)); CK(hipMemcpy(dH, H.data(), H.size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc((void **)&dg, n * 8)); CK(hipMemcpy(dg, g.data(), n * 8, hipMemcpyHostToDevice));
    CK(hipMalloc((void **)&du, 8)); CK(hipMemset(du, 0, 8));
    CK(hipMalloc((void **)&dx, n * 8));
    CK(hipMalloc((void **)&work, bcr_workspace_doubles(M, Bb) * 8)); CK(hipMemset(work, 0, bcr_workspace_doubles(M, Bb) * 8));
    CK(hipMalloc((void **)&status, 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    printf("cameras %d, band blocks %d, applicable %d\n", M, Bb, (int)bcr_applicable(M, Bb));
    float t = time_us(s, 50, [&] { bcr_solve(dH, Bb, M, dg, du, dx, work, status, s); });
    printf("whole solve, eager launches   %8.1f us\n", t);
    hipGraph_t gph; hipGraphExec_t ex;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    bcr_solve(dH, Bb, M, dg, du, dx, work, status, s);
    CK(hipStreamEndCapture(s, &gph)); CK(hipGraphInstantiate(&ex, gph, nullptr, nullptr, 0));
    t = time_us(s, 50, [&] { CK(hipGraphLaunch(ex, s)); });
    printf("whole solve, one graph        %8.1f us\n", t);
    int st_host = 0; CK(hipMemcpy(&st_host, status, 4, hipMemcpyDeviceToHost));
    std::vector<double> x(n); CK(hipMemcpy(x.data(), dx, n * 8, hipMemcpyDeviceToHost));
    printf("status %d  x[0] %.6g x[n/2] %.6g\n", st_host, x[0], x[n / 2]);
    // single kernels of the 32-scalar form
    BcrDev p; p.k = 5; p.nb = (M + 4) / 5; p.M = M; p.Bb = Bb;
    const int64_t m2 = (int64_t)p.nb * 1024, m1 = (int64_t)p.nb * 32;
    p.D = work; p.L = p.D + m2; p.T1 = p.L + m2; p.T2 = p.T1 + m2; p.L2 = p.T2 + m2; p.rhs = p.L2 + m2; p.t = p.rhs + m1; p.x = p.t + m1;
    for (int grid : {200, 50, 1}) {
        t = time_us(s, 200, [&] { hipLaunchKernelGGL(bcr_level_kernel, dim3(grid), dim3(256), 0, s, p, 1, p.L, p.L2, 0, status, dx); });
        unsigned long long c[16]; CK(hipMemcpyFromSymbol(c, HIP_SYMBOL(g_bcr_clk), sizeof c));
        printf("level kernel x%-4d %7.2f us   cycles: loads %llu | gauss-jordan %llu (rows/cols to LDS %llu, 4x4 inverse %llu, operands %llu, MFMA + rows %llu) | products 1 %llu | products 2 + stores %llu | total %llu\n", grid, t,
               c[1] - c[0], c[2] - c[1], c[8], c[9], c[10], c[11], c[3] - c[2], c[4] - c[3], c[4] - c[0]);
    }
    t = time_us(s, 200, [&] { hipLaunchKernelGGL(bcr_A_kernel<32>, dim3(100), dim3(256), 0, s, p, 1, status); });
    printf("A kernel x100     %7.2f us\n", t);
    t = time_us(s, 200, [&] { hipLaunchKernelGGL(bcr_B_kernel<32>, dim3(100), dim3(256), 0, s, p, 1); });
    printf("B kernel x100     %7.2f us\n", t);
    t = time_us(s, 200, [&] { hipLaunchKernelGGL(bcr_back_kernel<32>, dim3(100), dim3(256), 0, s, p, 1); });
    printf("back kernel x100  %7.2f us\n", t);
    t = time_us(s, 200, [&] { hipLaunchKernelGGL(bcr_assemble_kernel<32>, dim3(p.nb), dim3(256), 0, s, p, dH, dg, du); });
    printf("assemble          %7.2f us\n", t);
    t = time_us(s, 200, [&] { hipLaunchKernelGGL(bcr_scatter_kernel<32>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, dx); });
    printf("scatter           %7.2f us\n", t);
    return 0;
}
