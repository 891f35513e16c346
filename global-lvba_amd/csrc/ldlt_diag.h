// ldlt_diag.h -- LDL^T of one 64 x 64 diagonal block together with G = L11^-T D^-1 (included by ldlt.hip only, inside namespace
// lvba): the serial pivot chain of the band factorisation.
#pragma once

// ---------------------------------------------------------------------------------------------- K1
// The pivot reciprocal is v_rcp_f64 + 2 Newton steps instead of an IEEE division.
__device__ __forceinline__ double fast_rcp(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    r = fma(r, fma(-d, r, 1.0), r);
    r = fma(r, fma(-d, r, 1.0), r);
    return r;
}
#define LVBA_PIN(x) asm volatile("" : "+v"(x)) // keep the value computed HERE (LLVM otherwise sinks it to its first use)

// ------------------------------------------------------------------------------------------ K1, blocked
// LDL^T of the 64x64 diagonal block together with G = L11^-T D^-1 (an identity appended as 64 extra ROWS and carried
// through the same elimination).  The 64x64 block and the 64 appended identity rows live in LDS (W[128][64]); per 16-column
// block step
//   diag   wave 0 alone: the serial pivot chain.  ALL 80 rows the 16 columns still matter for ride in its registers: every
//          16-lane DPP row holds the 16 diagonal rows (a[16], four identical copies), and next to them (g[16]) one of the four
//          groups of 16 other rows -- this block's identity rows, the block rows below, the identity rows of finished blocks.
//          A pivot's column entries reach the other lanes through `row_newbcast` DPP moves (one 64-bit instruction; no SGPR round
//          trip, no wait states to pad), pivots are taken in PAIRS: the two reciprocals 1/d_J and 1/d_{J+1} = d_J / (d_J C - B^2)
//          are formed side by side, so the chain of dependent operations is paid once per two columns.  Arithmetically this is
//          the scalar elimination (same L, same D up to rounding): no 2x2 pivoting.  Rows of the diagonal block that are
//          already finished are not masked: what they accumulate lies above the diagonal and is never read.
//          With the other rows in the chain the old panel step (X = A G11 on the matrix pipe + a barrier) is gone.
//   update 4 waves: trailing 64 x (48 - 16 s) block -= X (X D)^T, fp64 MFMA; the column tile the next diag step needs first,
//          then (beside that diag step, which wave 0 runs) the rest.
// History: row-per-lane with LDS publishes 22 us per block; one wavefront's registers + v_readlane broadcasts + a panel product
// 11.8 us (25.4 k cycles: 294 per pivot -- two readlanes, a wait state and an FMA per entry, the Newton steps behind them);
// the symmetric sweep operator 22 us (a barrier per pivot); this form: see DESIGN.md 5.2.
#define LVBA_W1S 130 // column stride of W (doubles)
#define LVBA_Z1S 50  // column stride of the Z^T tile (doubles)
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
// lane L of every 16-lane row to all lanes of that row (v_mov_b64_dpp row_newbcast)
template <int L>
__device__ __forceinline__ double bcast16(double v)
{
    return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + L, 0xF, 0xF, true); // (bound_ctrl: no lane keeps `old`, so no register is set up for it)
}
template <int J, int C>
__device__ __forceinline__ void k1p_col(double (&a)[16], double (&g)[16], double ua1, double u2, double la1, double la2, double lg1,
                                        double lg2)
{
    const double u1c = bcast16<C>(ua1), u2c = bcast16<C>(u2);
    a[C] = fma(-la2, u2c, fma(-la1, u1c, a[C]));
    g[C] = fma(-lg2, u2c, fma(-lg1, u1c, g[C]));
}
template <int J, int... Cs>
__device__ __forceinline__ void k1p_cols(std::integer_sequence<int, Cs...>, double (&a)[16], double (&g)[16], double ua1, double u2,
                                         double la1, double la2, double lg1, double lg2)
{
    (k1p_col<J, J + 2 + Cs>(a, g, ua1, u2, la1, la2, lg1, lg2), ...);
}
// columns J and J + 1 (J even).  dd[c] = d_c on every lane.
template <int J>
__device__ __forceinline__ void k1p_pair(double (&a)[16], double (&g)[16], double (&dd)[16])
{
    const double ua1 = a[J];
    const double P = bcast16<J>(ua1), B = bcast16<J + 1>(ua1), Cc = bcast16<J + 1>(a[J + 1]);
    const double r1 = fast_rcp(P);
    const double det = fma(P, Cc, -(B * B));
    const double r2 = P * fast_rcp(det);
    const double la1 = ua1 * r1;
    const double u2 = fma(-la1, B, a[J + 1]);
    const double la2 = u2 * r2;
    const double lg1 = g[J] * r1;
    const double ug2 = fma(-lg1, B, g[J + 1]);
    const double lg2 = ug2 * r2;
    a[J + 1] = u2;
    g[J] = lg1;
    g[J + 1] = lg2;
    dd[J] = P;
    dd[J + 1] = bcast16<J + 1>(u2);
    k1p_cols<J>(std::make_integer_sequence<int, 14 - J>{}, a, g, ua1, u2, la1, la2, lg1, lg2);
}
template <int... Ps>
__device__ __forceinline__ void k1p_pairs(std::integer_sequence<int, Ps...>, double (&a)[16], double (&g)[16], double (&dd)[16])
{
    (k1p_pair<2 * Ps>(a, g, dd), ...);
}

#define LVBA_K1B_LDS (64 * LVBA_W1S + 2 * 16 * LVBA_Z1S + 64) // doubles
#define LVBA_K1B_DVS (64 * LVBA_W1S + 2 * 16 * LVBA_Z1S)      // where diag_blocked_factor leaves d[64]
// diag_blocked_load: the 64x64 block at (k, k) into W (lower triangle; identity below row nbe) with the identity appended.
// diag_blocked_factor: the factorisation of what W holds (the look-ahead kernel fills W itself, from the registers its updates
// of the block end in).  Leaves d in lds[LVBA_K1B_DVS ..] and G[m][c] in W[c * LVBA_W1S + 64 + m]; ends on a __syncthreads().
__device__ __forceinline__ void diag_blocked_load(double *lds, LdltMat M, int64_t k, int nbe)
{
    double *W = lds;                      // (row, col) at col * LVBA_W1S + row; rows 64..127 = the appended identity
    const int tid = threadIdx.x;
    {
        // all 16 loads of a lane are issued before the first one is waited for (one memory latency instead of a chain of
        // load -> LDS store pairs)
        double vv[16];
        const int row = tid & 63;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int col = (tid >> 6) + 4 * it;
            double v = 0.0;
            if (row < nbe) {
                if (col <= row) v = M.a[(k + row) + (k + col) * M.ld];
            } else if (col == row)
                v = 1.0;
            vv[it] = v;
        }
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int col = (tid >> 6) + 4 * it;
            W[col * LVBA_W1S + row] = vv[it];
            W[col * LVBA_W1S + 64 + row] = (row == col) ? 1.0 : 0.0;
        }
    }
    __syncthreads();
}
__device__ __forceinline__ void diag_blocked_factor(double *lds, int nbe, int *__restrict__ status)
{
    double *W = lds;
    double *Zt = W + 64 * LVBA_W1S;       // two buffers [j][block row relative to c0 + 16] = X * d (block step s uses buffer s & 1)
    double *dvs = lds + LVBA_K1B_DVS;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int i15 = lane & 15, kk = lane >> 4;
    // ---- diag step of the 16 columns at c0: wavefront 0 only, no barrier inside
    auto diag_step = [&](int c0, double *Ztw) {
        const int nbr = 48 - c0;          // block rows below this block
        const int t = 16 * (kk - 1) + i15; // row groups 1..3: the 48 rows below / left over, as the update's row tiles have them
        const int R = kk == 0 ? 64 + c0 + i15 : (t < nbr ? c0 + 16 + t : 64 + t - nbr);
        double a[16], g[16], dd[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            a[c] = W[(c0 + c) * LVBA_W1S + c0 + i15];
            g[c] = W[(c0 + c) * LVBA_W1S + R];
        }
        k1p_pairs(std::make_integer_sequence<int, 8>{}, a, g, dd);
#pragma unroll
        for (int c = 0; c < 16; ++c) W[(c0 + c) * LVBA_W1S + R] = g[c];
        if (kk > 0 && t < nbr) {
#pragma unroll
            for (int c = 0; c < 16; ++c) Ztw[c * LVBA_Z1S + t] = g[c] * dd[c];
        }
        if (kk == 0) {
            double dl = 0.0;
#pragma unroll
            for (int c = 0; c < 16; ++c) dl = (c == i15) ? dd[c] : dl;
            dvs[c0 + i15] = dl;
            if (c0 + i15 < nbe && (!(dl != 0.0) || !isfinite(dl))) status[0] = 1;
        }
    };
    if (w == 0) diag_step(0, Zt);
    __syncthreads();
    for (int s = 0; s < 4; ++s) {
        const int c0 = 16 * s;
        const int nb_rows = 48 - c0; // block rows still to come
        const double *Ztr = Zt + (s & 1) * 16 * LVBA_Z1S;
        // row tile of this wave in the update: waves 0..2 -> the 48 rows below / left over (block rows, then identity rows of
        // finished blocks), wave 3 -> the identity rows of this block
        const int base = (w < 3) ? ((16 * w < nb_rows) ? c0 + 16 + 16 * w : 64 + 16 * w - nb_rows) : 64 + c0;
        // ---- update: C[base + i][c0 + 16 + 16 ct + n] -= sum_j X[base + i][c0 + j] * Z[16 ct + n][j].  A block-row tile
        // only needs its lower part (ct <= its own index); identity-row tiles need every column tile.
        const int ct_end = (w < 3 && 16 * w < nb_rows) ? w + 1 : nb_rows / 16;
        auto tile = [&](int ct) {
            d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double av = W[(c0 + 4 * q + kk) * LVBA_W1S + base + i15];
                const double bv = Ztr[(4 * q + kk) * LVBA_Z1S + 16 * ct + i15];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) W[(c0 + 16 + 16 * ct + i15) * LVBA_W1S + base + kk + 4 * r] -= acc[r];
        };
        // the column tile the next diag step reads (the next 16 columns, every row), then -- beside that diag step, wave 0's --
        // the rest: nobody else touches those 16 columns any more, and the step writes its Z^T into the other buffer
        if (nb_rows > 0) tile(0);
        __syncthreads();
        if (w == 0) {
            if (nb_rows > 0) diag_step(c0 + 16, Zt + ((s + 1) & 1) * 16 * LVBA_Z1S);
        } else {
            for (int ct = 1; ct < ct_end; ++ct) tile(ct);
        }
        __syncthreads();
    }
}
__device__ __forceinline__ void diag_blocked_body(double *lds, LdltMat M, int64_t k, int nbe, int *__restrict__ status)
{
    diag_blocked_load(lds, M, k, nbe);
    diag_blocked_factor(lds, nbe, status);
}

// blockIdx.y: the problem of a two-ended factorisation.  blockIdx.x = 1 (look-ahead schedule, first launch of a phase): the
// side copy of the tile below the diagonal block, A(rows w0 .., columns k ..) as [m][row], masked like load_panel_tile.
__global__ __launch_bounds__(256) void ldlt_diag_blocked_kernel(LdltMat M, int64_t k, int nbe, double *__restrict__ G,
                                                               double *__restrict__ dvec, int *__restrict__ status,
                                                               int64_t sA, int64_t sW, double *__restrict__ side, int64_t rend)
{
    __shared__ double lds[LVBA_K1B_LDS];
    if (blockIdx.y) { M.a += sA; G += sW; dvec += sW; if (side) side += sW; } // the second problem of a two-ended factorisation
    if (blockIdx.x == 1) {
        const int row = threadIdx.x & 63, w = threadIdx.x >> 6;
        const int64_t r = k + nbe + row;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int m = w + 4 * it;
            side[m * 64 + row] = (r < rend && m < nbe) ? M.a[r + (k + m) * M.ld] : 0.0;
        }
        return;
    }
    diag_blocked_body(lds, M, k, nbe, status);
    const double *W = lds, *dvs = lds + LVBA_K1B_DVS;
    const int tid = threadIdx.x;
    if (tid < nbe) dvec[k + tid] = dvs[tid];
    for (int e = tid; e < 4096; e += 256) { // G[m][c], row-major
        const int c = e & 63, m = e >> 6;
        G[e] = W[c * LVBA_W1S + 64 + m];
    }
}
