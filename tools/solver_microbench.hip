// tools/solver_microbench.hip -- standalone timing of the LDL^T kernels (development tool, not product).
// Includes ldlt.hip directly.  usage: solver_microbench [n=12000] [bw=2813]
#define LVBA_K1_TIMING
#ifndef LVBA_MB_DB
#define LVBA_MB_DB 2
#endif
#include "../global-lvba_amd/csrc/ldlt.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace lvba;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <class F> static float time_ms(hipStream_t s, int reps, F f)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(a, s));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main(int argc, char **argv)
{
    const int64_t n = argc > 1 ? atoll(argv[1]) : 12000, bw = argc > 2 ? atoll(argv[2]) : 2813;
    const int64_t ldab = bw + LVBA_NB + 64;
    LdltMat A; A.n = n; A.ld = ldab - 1; A.bw = bw;
    std::vector<double> hA((size_t)ldab * n + ldab, 0.0);
    srand(1);
    for (int64_t c = 0; c < n; ++c) {
        for (int64_t o = 1; o <= bw && c + o < n; ++o) hA[o + c * ldab] = 0.02 * (rand() / (double)RAND_MAX - 0.5);
        hA[c * ldab] = 2.0 + bw * 0.01;
    }
    CK(hipMalloc((void **)&A.a, hA.size() * 8)); CK(hipMemcpy(A.a, hA.data(), hA.size() * 8, hipMemcpyHostToDevice));
    const int64_t nsteps = ldlt_num_panels(n);
    double *work; CK(hipMalloc((void **)&work, ldlt_workspace_doubles(n, bw) * 8)); CK(hipMemset(work, 0, ldlt_workspace_doubles(n, bw) * 8));
    double *Gall = work, *dvec = Gall + nsteps * 4096, *b = dvec + n, *bacc = b + n;
    const int64_t ldz = ldz_for(n, bw);
    double *Zws = bacc + n;
    int *status; CK(hipMalloc((void **)&status, 4)); CK(hipMemset(status, 0, 4));
    double *x; CK(hipMalloc((void **)&x, n * 8));
    hipStream_t s; CK(hipStreamCreate(&s));
    const int64_t k = 64 * 10, w0 = k + 64, rend = (k + 64 + bw < n) ? k + 64 + bw : n, T = (rend - w0 + 63) / 64;
    printf("n=%lld bw=%lld panels=%lld T=%lld\n", (long long)n, (long long)bw, (long long)nsteps, (long long)T);
    float t;
    t = time_ms(s, 200, [&] { hipLaunchKernelGGL(ldlt_diag_blocked_kernel, dim3(1), dim3(256), 0, s, A, k, 64, Gall, dvec, status, (int64_t)0, (int64_t)0, (double *)nullptr, (int64_t)0); });
    {
        unsigned long long c[16]; CK(hipMemcpyFromSymbol(c, HIP_SYMBOL(g_k1b_clk), sizeof c));
        printf("K1 blocked %7.2f us   cycles: load %llu |", t * 1e3, c[1] - c[0]);
        for (int q = 0; q < 4; ++q) printf(" diag %llu panel %llu update %llu |", c[2 + 3 * q] - c[1 + 3 * q], c[3 + 3 * q] - c[2 + 3 * q], c[4 + 3 * q] - c[3 + 3 * q]);
        printf(" store %llu  total %llu\n", c[14] - c[13], c[14] - c[0]);
    }
    t = time_ms(s, 200, [&] { hipLaunchKernelGGL(ldlt_diagpanel_kernel, dim3((unsigned)T), dim3(256), 0, s, A, k, 64, w0, rend, Gall, dvec, Zws, ldz, b, status, 0, 0); });
    printf("diag+panel %7.2f us  (%lld tiles)\n", t * 1e3, (long long)T);
    t = time_ms(s, 100, [&] { hipLaunchKernelGGL(ldlt_update_kernel, dim3((unsigned)(T * (T + 1) / 2)), dim3(256), 0, s, A, k, 64, w0, rend, Zws, ldz, 0, 0, 0); });
    printf("K3 update %8.2f us  (%lld tiles, %.1f TFLOP/s)\n", t * 1e3, (long long)(T * (T + 1) / 2), T * (T + 1) / 2 * 2.0 * 64 * 64 * 64 / (t * 1e-3) / 1e12);
    t = time_ms(s, 200, [&] { hipLaunchKernelGGL(ldlt_back_kernel, dim3((unsigned)((bw + 255) / 256)), dim3(256), 0, s, A, k + bw, 64, Gall, dvec, b, bacc, x, k); });
    printf("back      %8.2f us\n", t * 1e3);
    t = time_ms(s, 200, [&] { hipLaunchKernelGGL(ldlt_diagpanel_kernel, dim3((unsigned)T), dim3(256), 0, s, A, k, 64, w0, rend, Gall, dvec, Zws, ldz, b, status, 0, 0);
                              hipLaunchKernelGGL(ldlt_update_kernel, dim3((unsigned)(T * (T + 1) / 2)), dim3(256), 0, s, A, k, 64, w0, rend, Zws, ldz, 0, 0, 0); });
    printf("diag+panel, update chain %8.2f us\n", t * 1e3);
    // ---- the look-ahead launch (ldlt_lookahead.h) in pieces, at this geometry: panel p = 10 with its predecessor
    {
        auto geo = [&](int64_t st) {
            PanelGeo g; g.k = 64 * st; g.nbe = 64; g.w0 = g.k + 64; g.rend = (g.w0 + bw < n) ? g.w0 + bw : n; g.T = (int)((g.rend - g.w0 + 63) / 64);
            return g;
        };
        double *side; CK(hipMalloc((void **)&side, 4 * 4096 * 8)); CK(hipMemset(side, 0, 4 * 4096 * 8));
        double *Z2; CK(hipMalloc((void **)&Z2, 4 * ldz * 64 * 8)); CK(hipMemset(Z2, 0, 4 * ldz * 64 * 8));
        Step2Args a{};
        a.skip_a = a.skip_b = -1;
        a.M = A; a.sA = 0; a.sW = 0; a.ldz = ldz; a.nprob = 1; a.roles = 1; a.has_q = 1; a.do_diag = 1; a.nbe_next = 64;
        a.p = geo(10); a.q = geo(9); a.rend_next = geo(11).rend;
        a.side_r = side; a.side_w = side + 4096; a.Gp = Gall + 10 * 4096; a.Gn = Gall + 11 * 4096; a.dvec = dvec; a.b = b;
        a.dq_r = side + 2 * 4096; a.dq_w = side + 3 * 4096; // (LVBA_MB_NODQ: the chain multiplies panel q's share itself, as before)
#ifdef LVBA_MB_NODQ
        a.dq_r = nullptr; a.dq_w = nullptr;
#endif
        a.Zp = Z2; a.Zq = Z2 + ldz * 64; a.status = status;
        const int Tfull = a.p.T;
        a.p.T = 1; // the chain workgroup alone
        t = time_ms(s, 200, [&] { hipLaunchKernelGGL((ldlt_step2_kernel<true, LVBA_MB_DB>), dim3(1), dim3(256), 0, s, a); });
        printf("look-ahead: chain role alone            %7.2f us\n", t * 1e3);
        a.has_q = 0;
        t = time_ms(s, 200, [&] { hipLaunchKernelGGL((ldlt_step2_kernel<true, LVBA_MB_DB>), dim3(1), dim3(256), 0, s, a); });
        printf("look-ahead: chain role, no predecessor  %7.2f us\n", t * 1e3);
        a.has_q = 1; a.do_diag = 0;
        t = time_ms(s, 200, [&] { hipLaunchKernelGGL((ldlt_step2_kernel<true, LVBA_MB_DB>), dim3(1), dim3(256), 0, s, a); });
        printf("look-ahead: chain role without the diagonal factorisation %7.2f us\n", t * 1e3);
        a.do_diag = 1; a.p.T = Tfull;
        t = time_ms(s, 200, [&] { hipLaunchKernelGGL((ldlt_step2_kernel<true, LVBA_MB_DB>), dim3((unsigned)Tfull), dim3(256), 0, s, a); });
        printf("look-ahead: chain + %d row roles         %7.2f us\n", Tfull - 1, t * 1e3);
        // + the pair job of the steady state: panels (8, 9) as a rank-128 update, first half of the tile columns
        a.njobs = 1;
        BulkJob &J = a.job[0];
        J.o = geo(9); J.e = geo(8); J.Zo = Z2 + ldz * 64; J.Ze = Z2 + 2 * ldz * 64; J.pair = 1;
        const int64_t Tb = J.o.T - 1;
        int64_t tot = 0, part = 0, cs = 1;
        for (int64_t c = 1; c < Tb; ++c) tot += pair_col_items(c, Tb);
        while (cs < Tb && (cs < 3 || 2 * part < tot)) part += pair_col_items(cs++, Tb);
        J.ca = 1; J.cb = cs; J.nwg = LVBA_MB_DB == 3 ? sq_job_items(1, cs, Tb) : part;
        t = time_ms(s, 200, [&] { hipLaunchKernelGGL((ldlt_step2_kernel<true, LVBA_MB_DB>), dim3((unsigned)(Tfull + J.nwg)), dim3(256), 0, s, a); });
        printf("look-ahead: roles + first half of a pair job (%lld tiles of 128 x 64) %7.2f us\n", (long long)J.nwg, t * 1e3);
        a.roles = 0;
        t = time_ms(s, 200, [&] { hipLaunchKernelGGL((ldlt_step2_kernel<true, LVBA_MB_DB>), dim3((unsigned)J.nwg), dim3(256), 0, s, a); });
        printf("look-ahead: that job alone              %7.2f us\n", t * 1e3);
    }
    // ---- the steady state of config C3's two-ended phase: TWO problems per launch (2 chain + 2 x 40 row workgroups + the first
    // half of a pair job for both: ~410 tiles of 128 x 64), with the chain workgroup's own clock, and the stagger experiment
    {
        const int64_t sA = ldab * (n + 1), sW = ldlt_workspace_doubles(n, bw) / 2 - 64;
        LdltMat A2 = A;
        CK(hipMalloc((void **)&A2.a, (2 * hA.size() + 65 * ldab) * 8));
        CK(hipMemcpy(A2.a, hA.data(), hA.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(A2.a + sA, hA.data(), std::min<size_t>(hA.size(), (size_t)sA) * 8, hipMemcpyHostToDevice));
        double *w2; CK(hipMalloc((void **)&w2, (2 * sW + 4 * 4096 + 8 * ldz * 64) * 8)); CK(hipMemset(w2, 0, (2 * sW + 4 * 4096 + 8 * ldz * 64) * 8));
        unsigned long long *dbg; CK(hipMalloc((void **)&dbg, 8 * 2048)); CK(hipMemset(dbg, 0, 8 * 2048));
        auto geo = [&](int64_t st) {
            PanelGeo g; g.k = 64 * st; g.nbe = 64; g.w0 = g.k + 64; g.rend = (g.w0 + bw < n) ? g.w0 + bw : n; g.T = (int)((g.rend - g.w0 + 63) / 64);
            return g;
        };
        // problem 0's arrays inside w2: G [16 panels] | d, b | Z x 4 | side x 2 ; problem 1 at + sW (sW only has to be larger)
        double *G0 = w2, *d0 = G0 + 16 * 4096, *b0 = d0 + n, *Z0 = b0 + n, *side0 = Z0 + 4 * ldz * 64;
        if (side0 + 4 * 4096 > w2 + sW) { printf("workspace layout too small\n"); return 1; }
        Step2Args a{};
        a.skip_a = a.skip_b = -1;
        a.M = A2; a.sA = sA; a.sW = sW; a.ldz = ldz; a.nprob = 2; a.roles = 1; a.has_q = 1; a.do_diag = 1; a.nbe_next = 64;
        a.p = geo(10); a.q = geo(9); a.rend_next = geo(11).rend;
        a.side_r = side0; a.side_w = side0 + 4096; a.Gp = G0 + 10 * 4096; a.Gn = G0 + 11 * 4096; a.dvec = d0; a.b = b0;
        a.dq_r = side0 + 2 * 4096; a.dq_w = side0 + 3 * 4096;
#ifdef LVBA_MB_NODQ
        a.dq_r = nullptr; a.dq_w = nullptr;
#endif
        a.Zp = Z0; a.Zq = Z0 + ldz * 64; a.status = status; a.dbg = dbg;
        const int T = a.p.T;
        auto chain_us = [&]() { unsigned long long c[2]; CK(hipMemcpy(c, dbg, 16, hipMemcpyDeviceToHost)); return (double)(c[1] - c[0]); };
        t = time_ms(s, 200, [&] { hipLaunchKernelGGL((ldlt_step2_kernel<true, LVBA_MB_DB>), dim3((unsigned)(2 * T)), dim3(256), 0, s, a); });
        printf("2 problems: roles alone (%d workgroups)        %7.2f us   chain workgroup %6.0f cycles\n", 2 * T, t * 1e3, chain_us());
        {
            unsigned long long c[10]; CK(hipMemcpy(c, dbg + 520, sizeof c, hipMemcpyDeviceToHost));
            const char *nm[9] = {"first loads + stage", "product q", "stage A, G (+ block load issue)", "product L", "put L, Z + stores", "product p",
                                 "b, W build", "diagonal factorisation", "G store"};
            printf("   chain role phases (cycles):");
            for (int k = 0; k < 9; ++k) printf(" %s %llu |", nm[k], c[k + 1] - c[k]);
            printf(" total %llu\n", c[9] - c[0]);
        }
        a.njobs = 1;
        BulkJob &J = a.job[0];
        J.o = geo(9); J.e = geo(8); J.Zo = Z0 + ldz * 64; J.Ze = Z0 + 2 * ldz * 64; J.pair = 1;
        const int64_t Tb = J.o.T - 1;
        int64_t tot = 0, part = 0, cs = 1;
        for (int64_t c = 1; c < Tb; ++c) tot += pair_col_items(c, Tb);
        while (cs < Tb && (cs < 3 || 2 * part < tot)) part += pair_col_items(cs++, Tb);
        J.ca = 1; J.cb = cs; J.nwg = LVBA_MB_DB == 3 ? sq_job_items(1, cs, Tb) : part;
        const double job_flops = 2.0 * part * 2.0 * 128 * 64 * 128; // (the 128 x 64 count: what lies above the diagonal is not work)
        const unsigned nall = (unsigned)(2 * (T + J.nwg));
        t = time_ms(s, 200, [&] { hipLaunchKernelGGL((ldlt_step2_kernel<true, LVBA_MB_DB>), dim3(nall), dim3(256), 0, s, a); });
        printf("2 problems: roles + first half of the pair job (%u workgroups) %7.2f us   chain workgroup %6.0f cycles\n", nall, t * 1e3, chain_us());
        { // who shares a CU with the two chain workgroups (blocks 0 and 1)?  HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]; XCC_ID [3:0]
            std::vector<unsigned long long> h(8 + nall);
            CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
            auto cu_of = [&](unsigned b) { const unsigned long long v = h[8 + b]; return (unsigned)(((v >> 32) & 15) << 16 | (v & 0xFF00)); };
            for (unsigned c = 0; c < 2; ++c) {
                printf("   chain block %u on xcc %llu hw_id 0x%llx; same CU:", c, (h[8 + c] >> 32) & 15, h[8 + c] & 0xFFFF);
                for (unsigned b = 0; b < nall; ++b) if (b != c && cu_of(b) == cu_of(c)) printf(" %u", b);
                printf("\n");
            }
            std::vector<unsigned> seen;
            for (unsigned b = 0; b < nall; ++b) { bool f = false; for (unsigned q : seen) f = f || q == cu_of(b); if (!f) seen.push_back(cu_of(b)); }
            printf("   distinct CUs used by the %u workgroups: %zu; blocks 0..15 on xcc:", nall, seen.size());
            for (unsigned b = 0; b < 16; ++b) printf(" %llu", (h[8 + b] >> 32) & 15);
            printf("\n");
            for (int pr : {256, 258, 264}) {
                a.skip_a = pr; a.skip_b = pr + 1;
                t = time_ms(s, 200, [&] { hipLaunchKernelGGL((ldlt_step2_kernel<true, LVBA_MB_DB>), dim3(nall), dim3(256), 0, s, a); });
                printf("   blocks %d, %d return at once: %7.2f us   chain workgroup %6.0f cycles\n", pr, pr + 1, t * 1e3, chain_us());
            }
            a.skip_a = a.skip_b = -1;
        }
        for (int from : {256}) // the second dispatch round only / every bulk workgroup of problem 1 ... (blockIdx >= from)
            for (int sn : {8, 24}) {
                a.stagger_from = from; a.stagger_n = sn;
                t = time_ms(s, 200, [&] { hipLaunchKernelGGL((ldlt_step2_kernel<true, LVBA_MB_DB>), dim3(nall), dim3(256), 0, s, a); });
                printf("   stagger: blockIdx >= %3d start %4.1f us late: %7.2f us   chain workgroup %6.0f cycles\n", from, sn * 0.43, t * 1e3, chain_us());
            }
        a.stagger_n = 0;
        a.roles = 0;
        t = time_ms(s, 200, [&] { hipLaunchKernelGGL((ldlt_step2_kernel<true, LVBA_MB_DB>), dim3((unsigned)(2 * J.nwg)), dim3(256), 0, s, a); });
        printf("2 problems: that job alone (%lld tiles)          %7.2f us  (%.1f TFLOP/s)\n", (long long)(2 * J.nwg), t * 1e3,
               job_flops / (t * 1e-3) / 1e12);
        {   // where does a bulk tile's time go?  stamps of one workgroup (block 7: first round; block 300: second round, sharing its CU)
            const char *nm[10] = {"first loads + stage", "products 0", "stage 1", "products 1", "stage 2 (+ C loads issued)", "products 2", "stage 3",
                                  "products 3", "", "C wait + stores"};
            for (int blk : {7, 300}) {
                CK(hipMemcpyToSymbol(HIP_SYMBOL(g_bulk_stamp_block), &blk, sizeof blk));
                hipLaunchKernelGGL((ldlt_step2_kernel<true, LVBA_MB_DB>), dim3((unsigned)(2 * J.nwg)), dim3(256), 0, s, a);
                CK(hipStreamSynchronize(s));
                unsigned long long c[16]; CK(hipMemcpyFromSymbol(c, HIP_SYMBOL(g_bulk_clk), sizeof c));
                printf("   bulk tile of block %3d (cycles from its start; two buffers: 1 prologue, 2..5 after products 0..3, 8 loop end, 10 end):", blk);
                for (int k = 1; k <= 10; ++k) printf(" [%d] %lld", k, (long long)(c[k] - c[0]));
                printf("\n");
                (void)nm;
            }
            int off = -1; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_bulk_stamp_block), &off, sizeof off));
        }
        a.fence_probe = 1; // what every bulk tile would pay in a flag-driven form without kernel boundaries
        t = time_ms(s, 200, [&] { hipLaunchKernelGGL((ldlt_step2_kernel<true, LVBA_MB_DB>), dim3((unsigned)(2 * J.nwg)), dim3(256), 0, s, a); });
        printf("2 problems: that job alone, every tile with poll + acquire fence before and release fence + atomic after: %7.2f us\n", t * 1e3);
        a.roles = 1;
        t = time_ms(s, 200, [&] { hipLaunchKernelGGL((ldlt_step2_kernel<true, LVBA_MB_DB>), dim3(nall), dim3(256), 0, s, a); });
        printf("2 problems: roles + that job, bulk tiles fenced the same way: %7.2f us   chain workgroup %6.0f cycles\n", t * 1e3, chain_us());
        a.fence_probe = 0;
        t = time_ms(s, 200, [&] { hipLaunchKernelGGL((ldlt_step2_kernel<true, LVBA_MB_DB>), dim3(nall), dim3(256), 0, s, a); });
        printf("2 problems: roles + that job, no fences:                      %7.2f us   chain workgroup %6.0f cycles\n", t * 1e3, chain_us());
        a.roles = 0;
        J.dbg_same = 1;
        t = time_ms(s, 200, [&] { hipLaunchKernelGGL((ldlt_step2_kernel<true, LVBA_MB_DB>), dim3((unsigned)(2 * J.nwg)), dim3(256), 0, s, a); });
        printf("2 problems: that job with every workgroup on the SAME tile (operands and C from L2): %7.2f us\n", t * 1e3);
        J.dbg_same = 0;
        for (unsigned nb : {128u, 256u, 384u, 512u}) { // fewer tiles: how does the launch scale?
            t = time_ms(s, 200, [&] { hipLaunchKernelGGL((ldlt_step2_kernel<true, LVBA_MB_DB>), dim3(nb), dim3(256), 0, s, a); });
            printf("   first %3u workgroups of the job only: %7.2f us\n", nb, t * 1e3);
        }
    }
    // empty-kernel launch chain for reference
    t = time_ms(s, 200, [&] { hipLaunchKernelGGL(ldlt_prepare_kernel, dim3(1), dim3(64), 0, s, A, work, 0, 0, work, work, work, (unsigned long long *)x, 0ULL, LdltTwist{0, 0, 0, 0, nullptr}, (const int32_t *)nullptr, 0, status); });
    printf("tiny kernel back-to-back %8.2f us\n", t * 1e3);
    return 0;
}
