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
// through the same elimination), organised so that the serial chain only ever spans a 16x16 block held in ONE wavefront's
// registers (an earlier row-per-lane form published every column through LDS and cost 22 us per block; this one 14 us): the 64x64 block and the 64 appended identity rows live in LDS
// (W[128][64]); per 16-column block step
//   diag   wave 0: lanes 0..15 hold the block's rows, lanes 16..31 the matching identity rows; 16 compile-time steps, the
//          pivot row reaches the other lanes through v_readlane (no LDS round trip, no barrier); yields d, the block's
//          G rows and G11 = L11^-T D11^-1;
//   panel  3 waves: the 48 rows below / left over (block rows still to come + identity rows of finished blocks) times G11,
//          fp64 MFMA 16x16x4 -- a triangular solve turned into a product, as everywhere else in this file;
//   update 4 waves: trailing 64 x (48 - 16 s) block -= X (X D)^T, fp64 MFMA.
// 64 pivots still follow one another, but each costs ~(16 - j) readlane+FMA pairs instead of an LDS publish / flag /
// read-back of a 64-entry column.
// (Round 3 tried the opposite extreme -- the whole block by the symmetric sweep operator, every thread 16 entries of its column
// in registers, ONE barrier and one 64-double pivot row per pivot, which also yields A11^-1 and turns the panel into a block
// LDL^T step.  Measured with tools/solver_microbench: 22.0 us against 13.3 us for this form -- a workgroup barrier + LDS round
// trip per pivot is ~700 cycles, the in-wavefront chain here ~290 per pivot.  Withdrawn.)
#define LVBA_W1S 130 // column stride of W (doubles)
#define LVBA_Z1S 50  // column stride of the Z^T tile (doubles)
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
template <int J>
__device__ __forceinline__ void k1b_step(double (&a)[16], int lane, double &rd)
{
    const bool done = lane < 16 && lane <= J; // finished block rows: l = 0 leaves them untouched
    const double u = a[J];
    const double l = done ? 0.0 : u * rd;
    a[J] = done ? u : l;
    if constexpr (J + 1 < 16) {
        a[J + 1] = fma(-l, readlane_f64(u, J + 1), a[J + 1]);
        // next pivot: start its reciprocal now, refine it after the rest of the row (the FMAs below do not depend on it
        // and fill the latency of v_rcp_f64 and of the readlanes)
        const double pn = readlane_f64(a[J + 1], J + 1);
        double r = __builtin_amdgcn_rcp(pn);
        LVBA_PIN(r);
#pragma unroll
        for (int c = J + 2; c < 16; ++c) {
            a[c] = fma(-l, readlane_f64(u, c), a[c]);
            LVBA_PIN(a[c]);
        }
        r = fma(r, fma(-pn, r, 1.0), r);
        r = fma(r, fma(-pn, r, 1.0), r);
        rd = r;
    }
}
template <int... Js>
__device__ __forceinline__ void k1b_steps(std::integer_sequence<int, Js...>, double (&a)[16], int lane, double rd)
{
    (k1b_step<Js>(a, lane, rd), ...);
}

#define LVBA_K1B_LDS (64 * LVBA_W1S + 256 + 16 * LVBA_Z1S + 64) // doubles
// diag_blocked_load: the 64x64 block at (k, k) into W (lower triangle; identity below row nbe) with the identity appended.
// diag_blocked_factor: the factorisation of what W holds (the look-ahead kernel fills W itself, from the registers its updates
// of the block end in).  Leaves d in dvs[64] and G[m][c] in W[c * LVBA_W1S + 64 + m]; ends on a __syncthreads().
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
    double *G11s = W + 64 * LVBA_W1S;     // [m][c]
    double *Zt = G11s + 256;              // [j][block row relative to c0 + 16] = X * d
    double *dvs = Zt + 16 * LVBA_Z1S;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int i15 = lane & 15, kk = lane >> 4;
    // ---- diag step of the 16 columns at c0: wavefront 0 only, no barrier inside
    auto diag_step = [&](int c0) {
        const int r = lane < 16 ? c0 + lane : 64 + c0 + (lane & 15);
        double a[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) a[c] = (lane < 32) ? W[(c0 + c) * LVBA_W1S + r] : 0.0;
        k1b_steps(std::make_integer_sequence<int, 16>{}, a, lane, fast_rcp(readlane_f64(a[0], 0)));
        if (lane < 16) {
            double dl = 0.0;
#pragma unroll
            for (int c = 0; c < 16; ++c) dl = (c == lane) ? a[c] : dl;
            dvs[c0 + lane] = dl;
            if (c0 + lane < nbe && (!(dl != 0.0) || !isfinite(dl))) status[0] = 1;
        } else if (lane < 32) {
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                W[(c0 + c) * LVBA_W1S + r] = a[c];
                G11s[(lane - 16) * 16 + c] = a[c];
            }
        }
    };
    if (w == 0) diag_step(0);
    __syncthreads();
    for (int s = 0; s < 4; ++s) {
        const int c0 = 16 * s;
        const int nb_rows = 48 - c0; // block rows still to come
        // row tile of this wave in the panel / update steps: waves 0..2 -> the 48 panel rows, wave 3 -> the identity
        // rows of this block (their X is what the diag step wrote)
        const int base = (w < 3) ? ((16 * w < nb_rows) ? c0 + 16 + 16 * w : 64 + 16 * w - nb_rows) : 64 + c0;
        if (w < 3) { // ---- panel: X = A * G11
            d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double av = W[(c0 + 4 * q + kk) * LVBA_W1S + base + i15];
                const double bv = G11s[(4 * q + kk) * 16 + i15];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
            }
            // acc[r] = X[base + kk + 4r][c0 + i15]
            const double dj = dvs[c0 + i15];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                W[(c0 + i15) * LVBA_W1S + base + kk + 4 * r] = acc[r];
                if (16 * w < nb_rows) Zt[i15 * LVBA_Z1S + 16 * w + kk + 4 * r] = acc[r] * dj;
            }
        }
        __syncthreads();
        // ---- update: C[base + i][c0 + 16 + 16 ct + n] -= sum_j X[base + i][c0 + j] * Z[16 ct + n][j].  A block-row tile
        // only needs its lower part (ct <= its own index); identity-row tiles need every column tile.
        const int ct_end = (w < 3 && 16 * w < nb_rows) ? w + 1 : nb_rows / 16;
        for (int ct = 0; ct < ct_end; ++ct) {
            d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double av = W[(c0 + 4 * q + kk) * LVBA_W1S + base + i15];
                const double bv = Zt[(4 * q + kk) * LVBA_Z1S + 16 * ct + i15];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) W[(c0 + 16 + 16 * ct + i15) * LVBA_W1S + base + kk + 4 * r] -= acc[r];
        }
        // look-ahead: wavefront 0's tile was the next diagonal block (rows c0+16.., column tile 0), which nobody else
        // touches -- its pivot chain runs while the other wavefronts finish their update tiles
        if (w == 0 && nb_rows > 0) diag_step(c0 + 16);
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
    const double *W = lds, *dvs = lds + 64 * LVBA_W1S + 256 + 16 * LVBA_Z1S;
    const int tid = threadIdx.x;
    if (tid < nbe) dvec[k + tid] = dvs[tid];
    for (int e = tid; e < 4096; e += 256) { // G[m][c], row-major
        const int c = e & 63, m = e >> 6;
        G[e] = W[c * LVBA_W1S + 64 + m];
    }
}
