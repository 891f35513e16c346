// ldlt_nd.h -- the damped solve for pose graphs that are NOT a narrow band: one level of nested dissection (included by ldlt.hip
// only, inside namespace lvba; the partition comes from nd_plan.h, host-only).
//
// The reference factorises whatever sparsity pattern arrives (Eigen::SimplicialLDLT + AMD ordering, include/BALM/bavoxel.hpp:
// 696-710).  The band LDL^T of this file family covers trajectories whose co-visibility graph folds into a band; two shapes it
// does not cover are (a) a graph with a HUB -- a place crossed many times: a clique of poses hanging on the ring, which no band
// ordering keeps narrow -- and (b) a band that is long compared with its width (10 000 poses, n / bw = 24) on SEVERAL GPUs, where
// the band factorisation is a serial chain two ranks can split at best.  Both are the same structure:
//
//        [ A_1            E_1^T ]      A_a   the ARCS: what is left of the graph when the separator poses are taken out, each a
//        [      ...        ...  ]            band matrix of its own (independent of the other arcs),
//        [           A_P  E_P^T ]      S     the SEPARATOR poses (the hub / chunks of the long band) with their fill,
//        [ E_1  ...  E_P    S   ]      E_a   the coupling of arc a to the separator poses it touches (s_a columns).
//
//   per arc a (its own stream; on several ranks: its owner)
//        A_a = L D L^T, c = L^-1 b_a          ldlt_solve(..., LDLT_FACTOR): the look-ahead band factorisation, unchanged
//        Y   = L^-1 E_a^T                     riding in the factorisation's launches one panel behind (ldlt_lookahead.h: FwdPassenger),
//                                             fp64 MFMA: the s_a border columns are right-hand sides
//        S_a = Y^T D^-1 Y,  g_a = Y^T D^-1 c  nd_schur_kernel / nd_gs_kernel
//   separator   S' = S + u diag(S) - sum_a S_a,  g' = g_S + sum_a g_a  (summed in arc order: deterministic; all-reduced over
//               the ranks), solved by ldlt_solve again (band or dense, both ends at once, two ranks) -> x_S
//   per arc a   b_a -= E_a^T x_S (in the factorisation's own convention: nd_zstep_kernel), ldlt_solve(..., LDLT_BACKWARD) -> x_a
//
// Storage: B (the rows of E_a^T as the forward substitution leaves them) and Y are [n_a][ldb] ROW-major -- a tile is read along
// the separator columns, which is the contiguous direction for every consumer (operand tiles [m][x] of tile_product, the rank
// updates, the matrix-vector products).  The block store of the separator system has the layout of the Hessian store
// (block (I, J) at (J (Bb + 1) + I - J) 36), so that ldlt_solve's fill kernels read it as they read the Hessian.
#pragma once

// (NdArc / NdSys: lvba_internal.h)

// B[6 (J - p0) + c][6 q + r] = H[6 I + r][6 J + c], I = ps + sep[q]  (block (I, J) of the full lower block store, stride N)
__global__ void nd_border_fill_kernel(const double *__restrict__ Hblk, int64_t N, int32_t p0, int32_t Na, int32_t ps,
                                      const int32_t *__restrict__ sep, int32_t nsep, double *__restrict__ B, int64_t ldb)
{
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= (int64_t)Na * nsep * 36) return;
    const int el = (int)(e % 36);
    const int64_t blk = e / 36;
    const int32_t q = (int32_t)(blk % nsep), jj = (int32_t)(blk / nsep);
    const int c = el / 6, r = el - 6 * c;
    const int64_t J = p0 + jj, I = (int64_t)ps + sep[q];
    B[(6 * (int64_t)jj + c) * ldb + 6 * q + r] = Hblk[(J * N + (I - J)) * 36 + el];
}

// wv[k + c] = sum_m G_p[m][c] b[k + m]  (one workgroup per panel): D^-1 L^-1 b, what the separator's right-hand side needs
// (and rd = 1 / d for nd_schur_kernel: with sixteen fp64 divisions per thread and row chunk it took 4.32 instead of 4.10 ms at n = 29 000)
__global__ __launch_bounds__(256) void nd_w_kernel(const double *__restrict__ Gall, const double *__restrict__ b, const double *__restrict__ dvec,
                                                   int64_t n, double *__restrict__ wv, double *__restrict__ rd)
{
    __shared__ double red[4][64];
    const int64_t k = 64 * (int64_t)blockIdx.x;
    const int nbe = (int)((n - k) < 64 ? (n - k) : 64);
    const double *G = Gall + 4096 * (int64_t)blockIdx.x;
    const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
    double s = 0.0;
#pragma unroll
    for (int m = 16 * q; m < 16 * q + 16; ++m) s += m < nbe ? G[m * 64 + c] * b[k + m] : 0.0;
    red[q][c] = s;
    __syncthreads();
    if (threadIdx.x < nbe) {
        wv[k + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        rd[k + threadIdx.x] = 1.0 / dvec[k + threadIdx.x];
    }
}

// gpart[slice][jj] = sum over the slice's rows of Y[r][jj] wv[r]   (grid: ldb / 64 x ND_GS_SLICES)
__global__ __launch_bounds__(256) void nd_gs_kernel(const double *__restrict__ Y, const double *__restrict__ wv, int64_t n, int64_t ldb,
                                                    double *__restrict__ gpart)
{
    __shared__ double red[4][64];
    const int jj = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int64_t j = 64 * (int64_t)blockIdx.x + jj;
    const int64_t per = (n + ND_GS_SLICES - 1) / ND_GS_SLICES, ra = per * blockIdx.y, rb = (ra + per < n) ? ra + per : n;
    double s = 0.0;
    for (int64_t r = ra + q; r < rb; r += 4) s += Y[r * ldb + j] * wv[r];
    red[q][jj] = s;
    __syncthreads();
    if (threadIdx.x < 64) gpart[(int64_t)blockIdx.y * ldb + j] = red[0][jj] + red[1][jj] + red[2][jj] + red[3][jj];
}

// Sa(i, j) = sum_r Y[r][i] Y[r][j] rd_r (rd = 1 / d: nd_w_kernel) for the lower 64 x 64 tiles (ti >= tj), column j at Sa + j ldb
__global__ __launch_bounds__(256, 2) void nd_schur_kernel(const double *__restrict__ Y, const double *__restrict__ rd, int64_t n,
                                                          int64_t ldb, double *__restrict__ Sa)
{
    __shared__ double lds[LVBA_K3_LDS];
    double *Ls = lds, *Zs = lds + 64 * LVBA_TS;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, row = tid & 63, i = lane & 15, kk = lane >> 4;
    int64_t ti = (int64_t)((sqrt(8.0 * (double)blockIdx.x + 1.0) - 1.0) * 0.5);
    while (ti * (ti + 1) / 2 > (int64_t)blockIdx.x) --ti;
    while ((ti + 1) * (ti + 2) / 2 <= (int64_t)blockIdx.x) ++ti;
    const int64_t tj = blockIdx.x - ti * (ti + 1) / 2;
    const int64_t i0 = 64 * ti, j0 = 64 * tj;
    d4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
    // fetch() only loads -- from a clamped row, unconditionally: rows past the end are masked and the 1 / d factor is applied when the
    // chunk is staged, one product later.  (With the mask and the factor applied where the values were loaded, the compiler waited
    // for every row's loads in turn: sixteen memory round trips per chunk, 4.10 instead of 3.30 ms at n = 29 000, s = 1 818.)
    double yi[16], yj[16], rv[16];
    auto fetch = [&](int64_t rc) {
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int64_t r = rc + w + 4 * it, rr = r < n ? r : n - 1;
            yi[it] = Y[rr * ldb + i0 + row];
            yj[it] = Y[rr * ldb + j0 + row];
            rv[it] = rd[rr];
        }
    };
    fetch(0);
    for (int64_t rc = 0; rc < n; rc += 64) {
        if (rc) __syncthreads(); // everybody has left the tiles of the chunk before
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const bool ok = rc + w + 4 * it < n;
            yi[it] = ok ? yi[it] : 0.0;
            yj[it] = ok ? yj[it] * rv[it] : 0.0;
        }
        stage_tile(Ls, yi, w, row);
        stage_tile(Zs, yj, w, row);
        __syncthreads();
        if (rc + 64 < n) fetch(rc + 64); // the next chunk travels beside this chunk's product
        tile_product(Ls, Zs, w, i, kk, acc); // (i = 16 t + i, j = 16 w + kk + 4 reg)
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int64_t j = j0 + 16 * w + kk + 4 * reg;
#pragma unroll
        for (int t = 0; t < 4; ++t) Sa[j * ldb + i0 + 16 * t + i] = acc[t][reg];
    }
}
// (A 128 x 128 tile per workgroup, 64 x 64 per wavefront -- twice the arithmetic per LDS read -- was measured in round 5 and is
// SLOWER: 1.59 against 0.73 ms at n = 5 100, s = 1 818.  The border has a few dozen tile columns: 120 tiles of 128 leave half the
// chip idle where 435 tiles of 64 fill it.  Cutting the arc's rows into slices so that 128 x 128 tiles fill the chip again (partial
// sums added in slice order) did not help either: 5.0 against 4.3 ms at n = 29 000 -- 312 registers, one wavefront per SIMD, and the
// row chunks' loads are not hidden.  This kernel requests Y ~ 30 times (12.9 GB at that size, 3.1 TB/s); a blocked tile order that lets
// an XCD's 55 tiles share 11 - 26 of the 29 column panels instead of all of them changed nothing (4.46 against 4.32 ms): the tiles march
// through Y's rows together, a 64-row slab of Y is under 1 MB, and the panels come out of L2 either way.)

// The separator system's block store and gradient from the Hessian store: S(I, J) + u diag, as ldlt_solve's fill expects it
// (the damping is applied HERE: the solve runs with u = 0, because the Schur complements must not be damped).  include = 0:
// zeros (a rank other than 0 of a multi-rank job: the sum over the ranks is the system)
__global__ void nd_sep_build_kernel(const double *__restrict__ Hblk, int64_t N, int32_t ps, int32_t Ns, int32_t BbS,
                                    const double *__restrict__ u_dev, const double *__restrict__ g, int include,
                                    double *__restrict__ Sblk, double *__restrict__ gS)
{
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)Ns * (BbS + 1) * 36;
    if (e < 6 * (int64_t)Ns) gS[e] = include ? g[6 * (int64_t)ps + e] : 0.0;
    if (e >= total) return;
    const int el = (int)(e % 36);
    const int64_t slot = e / 36, J = slot / (BbS + 1), d = slot - J * (BbS + 1);
    double v = 0.0;
    if (include && J + d < Ns) {
        v = Hblk[(((int64_t)ps + J) * N + d) * 36 + el];
        if (d == 0 && el / 6 == el % 6) v += u_dev[0] * v;
    }
    Sblk[e] = v;
}

// Sblk -= S_a, gS += g_a for one arc (launched arc after arc on the main stream: the sums have a fixed order)
__global__ void nd_sep_sub_kernel(const double *__restrict__ Sa, const double *__restrict__ gpart, int64_t ldb,
                                  const int32_t *__restrict__ sep, int32_t nsep, int32_t BbS, double *__restrict__ Sblk,
                                  double *__restrict__ gS)
{
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e < 6 * (int64_t)nsep) {
        double s = 0.0;
        for (int sl = 0; sl < ND_GS_SLICES; ++sl) s += gpart[(int64_t)sl * ldb + e];
        gS[6 * (int64_t)sep[e / 6] + e % 6] += s;
    }
    const int64_t nb = (int64_t)nsep * (nsep + 1) / 2;
    if (e >= nb * 36) return;
    const int el = (int)(e % 36);
    const int64_t b = e / 36;
    int64_t qi = (int64_t)((sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
    while (qi * (qi + 1) / 2 > b) --qi;
    while ((qi + 1) * (qi + 2) / 2 <= b) ++qi;
    const int64_t qj = b - qi * (qi + 1) / 2; // qi >= qj, sep ascending: I >= J
    const int c = el / 6, r = el - 6 * c;
    const int64_t I = sep[qi], J = sep[qj];
    int64_t si = 6 * qi + r, sj = 6 * qj + c;
    if (si < sj) { const int64_t t = si; si = sj; sj = t; } // (diagonal blocks: the upper half by symmetry -- only lower tiles exist)
    Sblk[(J * (BbS + 1) + (I - J)) * 36 + el] -= Sa[sj * ldb + si];
}

// b[r] -= sum_jj B[r][jj] xS[6 sep[jj / 6] + jj % 6]   (one wavefront per row)
__global__ __launch_bounds__(256) void nd_zstep_kernel(const double *__restrict__ B, int64_t n, int64_t ldb, const double *__restrict__ xS,
                                                       const int32_t *__restrict__ sep, int32_t nsep, double *__restrict__ b)
{
    const int lane = threadIdx.x & 63;
    const int64_t r = 4 * (int64_t)blockIdx.x + (threadIdx.x >> 6);
    if (r >= n) return;
    double s = 0.0;
    for (int64_t jj = lane; jj < 6 * (int64_t)nsep; jj += 64) s += B[r * ldb + jj] * xS[6 * (int64_t)sep[jj / 6] + jj % 6];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) b[r] -= s;
}

__global__ void nd_status_kernel(const int *__restrict__ st, int n, int *__restrict__ out)
{
    int m = 0;
    for (int a = 0; a < n; ++a) m = st[a] > m ? st[a] : m;
    out[0] = m;
}
__global__ void nd_zero_kernel(double *__restrict__ x, int64_t n)
{
    const int64_t a = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (a < n) x[a] = 0.0;
}

// ------------------------------------------------------------------------------------------- host
int32_t nd_solve(NdSys &nd, const double *Hblk, int32_t N, const double *g, const double *u_dev, double *x, int *status,
                 hipStream_t s, const LdltDist *dist)
{
    const int rank = dist ? dist->rank : 0;
    const int P = (int)nd.arcs.size();
    const int64_t Bb1 = N; // the Hessian store of a dissected system is the full lower block triangle
    // an error return must not leave arc streams running beside what the caller enqueues on s next (retract, the next evaluation):
    // s waits for every owned arc's stream first
    auto bail = [&](int32_t rc) {
        for (int a = 0; a < P; ++a) {
            NdArc &A = nd.arcs[(size_t)a];
            if (A.owner != rank) continue;
            hipEventRecord(A.done, A.stream);
            hipStreamWaitEvent(s, A.done, 0);
        }
        (void)hipGetLastError();
        return rc;
    };
    hipEventRecord(nd.start, s);
    // ---- the arcs, each on a stream of its own: the factorisation, with the forward substitution of the border's columns riding in
    // its launches one panel behind (ldlt_lookahead.h: FwdPassenger -- a step launch is a few dozen workgroups on a serial chain, the
    // passengers a few hundred: they share the chip, and the host issues ONE launch per panel), then Y^T D^-1 Y
    for (int a = 0; a < P; ++a) {
        NdArc &A = nd.arcs[(size_t)a];
        if (A.owner != rank) continue;
        hipStream_t as = A.stream;
        hipStreamWaitEvent(as, nd.start, 0);
        LdltBorder border{A.B, A.Y, A.ldb};
        if (A.nsep > 0) {
            const int64_t cnt = (int64_t)A.Na * A.nsep * 36;
            hipLaunchKernelGGL(nd_border_fill_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, as, Hblk, (int64_t)N, A.p0, A.Na,
                               nd.ps, A.sep, A.nsep, A.B, A.ldb);
        }
        const int32_t rc = ldlt_solve(A.A, Hblk + (int64_t)A.p0 * Bb1 * 36, (int)(Bb1 - 1), A.Na, g + 6 * (int64_t)A.p0, u_dev,
                                      x + 6 * (int64_t)A.p0, A.work, A.status, as, nullptr, nullptr, LDLT_FACTOR,
                                      A.nsep > 0 ? &border : nullptr);
        if (rc != LVBA_OK) return bail(rc);
        if (A.nsep > 0) {
            const double *Gall = ldlt_work_G(A.work), *dvec = ldlt_work_d(A.n, A.work);
            const unsigned nct = (unsigned)(A.ldb / 64);
            const unsigned np = (unsigned)((A.n + LVBA_NB - 1) / LVBA_NB);
            hipLaunchKernelGGL(nd_w_kernel, dim3(np), dim3(256), 0, as, Gall, (const double *)ldlt_work_b(A.n, A.work), dvec, A.n, A.wv, A.wv + A.n);
            hipLaunchKernelGGL(nd_gs_kernel, dim3(nct, ND_GS_SLICES), dim3(256), 0, as, (const double *)A.Y, (const double *)A.wv, A.n,
                               A.ldb, A.gpart);
            hipLaunchKernelGGL(nd_schur_kernel, dim3(nct * (nct + 1) / 2), dim3(256), 0, as, (const double *)A.Y, (const double *)(A.wv + A.n), A.n, A.ldb,
                               A.Sa);
        }
        hipEventRecord(A.done, as);
    }
    // ---- the separator system: S + u diag(S) from the store, minus the arcs' Schur complements in arc order
    const int64_t sep_doubles = (int64_t)nd.Ns * (nd.BbS + 1) * 36;
    {
        const int64_t cnt = std::max<int64_t>(sep_doubles, 6 * (int64_t)nd.Ns);
        hipLaunchKernelGGL(nd_sep_build_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, s, Hblk, (int64_t)N, nd.ps, nd.Ns, nd.BbS,
                           u_dev, g, rank == 0 ? 1 : 0, nd.Sblk, nd.Sblk + sep_doubles);
    }
    for (int a = 0; a < P; ++a) {
        NdArc &A = nd.arcs[(size_t)a];
        if (A.owner != rank) continue;
        hipStreamWaitEvent(s, A.done, 0);
        if (A.nsep == 0) continue;
        const int64_t cnt = std::max<int64_t>((int64_t)A.nsep * (A.nsep + 1) / 2 * 36, 6 * (int64_t)A.nsep);
        hipLaunchKernelGGL(nd_sep_sub_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, s, (const double *)A.Sa,
                           (const double *)A.gpart, A.ldb, A.sep, A.nsep, nd.BbS, nd.Sblk, nd.Sblk + sep_doubles);
    }
    if (dist && dist->n_ranks >= 2 && dist->allreduce_sum(dist->ctx, nd.Sblk, (size_t)(sep_doubles + 6 * (int64_t)nd.Ns))) return bail(LVBA_ERR_DIST);
    double *xS = x + 6 * (int64_t)nd.ps;
    {
        const int32_t rc = ldlt_solve(nd.AS, nd.Sblk, nd.BbS, nd.Ns, nd.Sblk + sep_doubles, nd.d_zero, xS, nd.workS, nd.statusS, s, dist, nullptr,
                                      LDLT_ALL);
        if (rc != LVBA_OK) return bail(rc);
    }
    // ---- back into the arcs
    hipEventRecord(nd.mid, s);
    for (int a = 0; a < P; ++a) {
        NdArc &A = nd.arcs[(size_t)a];
        double *xa = x + 6 * (int64_t)A.p0;
        if (A.owner != rank) { // another rank's arc: zeros here, its solution arrives with the sum below
            hipLaunchKernelGGL(nd_zero_kernel, dim3((unsigned)((A.n + 255) / 256)), dim3(256), 0, s, xa, A.n);
            continue;
        }
        hipStream_t as = A.stream;
        hipStreamWaitEvent(as, nd.mid, 0);
        if (A.nsep > 0)
            hipLaunchKernelGGL(nd_zstep_kernel, dim3((unsigned)((A.n + 3) / 4)), dim3(256), 0, as, (const double *)A.B, A.n, A.ldb,
                               (const double *)xS, A.sep, A.nsep, ldlt_work_b(A.n, A.work));
        const int32_t rc = ldlt_solve(A.A, Hblk + (int64_t)A.p0 * Bb1 * 36, (int)(Bb1 - 1), A.Na, g + 6 * (int64_t)A.p0, u_dev, xa, A.work,
                                      A.status, as, nullptr, nullptr, LDLT_BACKWARD);
        if (rc != LVBA_OK) return bail(rc);
        hipEventRecord(A.done, as);
    }
    for (int a = 0; a < P; ++a)
        if (nd.arcs[(size_t)a].owner == rank) hipStreamWaitEvent(s, nd.arcs[(size_t)a].done, 0);
    hipLaunchKernelGGL(nd_status_kernel, dim3(1), dim3(1), 0, s, (const int *)nd.d_stat, P + 1, status);
    if (hipGetLastError() != hipSuccess) return LVBA_ERR_DEVICE; // (one check behind the whole launch sequence, as elsewhere in the library)
    if (dist && dist->n_ranks >= 2) {
        if (dist->allreduce_sum(dist->ctx, x, (size_t)(6 * (int64_t)nd.ps))) return LVBA_ERR_DIST; // (x_S is the same on every rank already)
        if (dist->allreduce_max_i32(dist->ctx, status)) return LVBA_ERR_DIST;
    }
    return LVBA_OK;
}
