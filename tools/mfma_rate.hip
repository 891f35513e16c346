// tools/mfma_rate.hip -- how fast does ONE workgroup issue v_mfma_f64_16x16x4_f64?  (development tool)
// The look-ahead LDL^T's chain workgroup is alone on its CU, one wavefront per SIMD: its 64 x 64 x 64 products ran at ~120 cycles
// per MFMA in tools/solver_microbench, twice the 64 cycles the 78.6 TFLOP/s peak implies.  This measures the instruction alone:
// NACC independent accumulators per wavefront, WPS wavefronts per SIMD, operands in registers.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int NACC>
__global__ void rate_kernel(double *out, unsigned long long *clk, int iters)
{
    d4 acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a) acc[a] = (d4){0.0, 0.0, 0.0, 0.0};
    double x = 1.0 + threadIdx.x * 1e-3, y = 0.5 - threadIdx.x * 1e-4;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[a], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0.0;
#pragma unroll
    for (int a = 0; a < NACC; ++a) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int NACC> static int run(int threads, int blocks, double *out, unsigned long long *clk)
{
    const int iters = 512;
    hipLaunchKernelGGL(rate_kernel<NACC>, dim3(blocks), dim3(threads), 0, 0, out, clk, iters);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(rate_kernel<NACC>, dim3(blocks), dim3(threads), 0, 0, out, clk, iters);
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    unsigned long long c; CK(hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost));
    const double n = (double)iters * NACC;
    printf("accumulators %d, %d waves per SIMD, %3d workgroups: %6.1f clock ticks per MFMA per wave, %6.1f ns per MFMA per wave (launch %.1f us)\n", NACC,
           threads / 256, blocks, c / n, (ms * 1e6 - 3000) / n, ms * 1e3);
    return 0;
}

int main()
{
    double *out; unsigned long long *clk;
    CK(hipMalloc((void **)&out, 1024 * 512 * 8)); CK(hipMalloc((void **)&clk, 1024 * 8));
    for (int threads : {256, 512})
        for (int blocks : {1, 256}) {
            run<1>(threads, blocks, out, clk);
            run<2>(threads, blocks, out, clk);
            run<4>(threads, blocks, out, clk);
            run<8>(threads, blocks, out, clk);
        }
    return 0;
}
