// ldlt_tiles.h -- the trailing update C -= L Z^T of the band LDL^T as fp64-MFMA tiles (included by ldlt.hip only, inside namespace
// lvba): 64 x 64 tiles with 64-bit addressing (matrices of 4 GB and more) and the 128 x 64 tile of the look-ahead launches.
#pragma once

// ---------------------------------------------------------------------------------------------- K3
// every lower tile (ti >= tj) of the window.  At one 64x64 tile per workgroup the kernel moves 128 KB (C in and out, L, Z)
// per 0.52 MFLOP = 4 flop/B: it runs at the HBM bound (~5 TB/s -> ~20 TFLOP/s), not at the MFMA bound.
#define LVBA_K3_LDS (2 * 64 * LVBA_TS) // doubles
// column-major enumeration of the lower tiles of a Tb x Tb triangle: column tj holds the Tb - tj tiles ti = tj .. Tb-1, columns
// one after the other (bidx -> (ti >= tj)).  A launch can then take a RANGE of tile columns -- the ones the next
// factorisations need first.
__host__ __device__ __forceinline__ int64_t col_start(int64_t tj, int64_t Tb) { return tj * Tb - tj * (tj - 1) / 2; }
__device__ __forceinline__ void col_decode(int64_t bidx, int64_t Tb, int64_t &ti, int64_t &tj)
{
    const double bq = (double)(2 * Tb + 1);
    tj = (int64_t)((bq - sqrt(bq * bq - 8.0 * (double)bidx)) * 0.5);
    if (tj < 0) tj = 0;
    if (tj > Tb - 1) tj = Tb - 1;
    while (tj > 0 && col_start(tj, Tb) > bidx) --tj;
    while (tj + 1 < Tb && col_start(tj + 1, Tb) <= bidx) ++tj;
    ti = tj + (bidx - col_start(tj, Tb));
}
__device__ __forceinline__ void update_tile(double *lds, LdltMat M, int64_t k, int nbe, int64_t w0, int64_t rend,
                                            const double *__restrict__ Zws, int64_t ldz, int64_t ti, int64_t tj)
{
    double *Ls = lds;                // [m][row of tile ti]
    double *Zs = lds + 64 * LVBA_TS; // [m][row of tile tj] (= column of the updated tile)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t r0 = w0 + 64 * ti, c0 = w0 + 64 * tj;
    const int row = tid & 63;
    const int i = lane & 15, kk = lane >> 4;
    double lv[16], zv[16];
    {
        const int64_t r = r0 + row, c = c0 + row;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int m = w + 4 * it;
            lv[it] = (r < rend && m < nbe) ? M.a[r + (k + m) * M.ld] : 0.0;
            zv[it] = (c < rend && m < nbe) ? Zws[(c - w0) + m * ldz] : 0.0;
        }
    }
    // prefetch the C tile entries this lane updates: c = c0+16w+kk+4reg, r = r0+16t+i
    double cv[16];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int64_t c = c0 + 16 * w + kk + 4 * reg, r = r0 + 16 * t + i;
            cv[4 * t + reg] = (r < rend && c < rend && r >= c) ? M.a[r + c * M.ld] : 0.0;
        }
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int m = w + 4 * it;
        Ls[m * LVBA_TS + row] = lv[it];
        Zs[m * LVBA_TS + row] = zv[it];
    }
    __syncthreads();
    d4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int k0 = 0; k0 < 64; k0 += 4) {
        const double a = Zs[(k0 + kk) * LVBA_TS + 16 * w + i];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const double bv = Ls[(k0 + kk) * LVBA_TS + 16 * t + i];
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv, acc[t], 0, 0, 0);
        }
    }
    // acc[t][reg] = sum_m Z[c = 16w+kk+4reg][m] * L[r = 16t+i][m]
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int64_t c = c0 + 16 * w + kk + 4 * reg, r = r0 + 16 * t + i;
            if (r < rend && c < rend && r >= c) M.a[r + c * M.ld] = cv[4 * t + reg] - acc[t][reg];
        }
}
// The same tile with the contributions of TWO consecutive panels (e, then o = e + 1) in one pass: C is read and written once
// per 128 columns instead of once per 64 (rank-128 update; the end phase of the two-ended factorisation is bound by exactly that
// traffic).  (ti, tj) are tile coordinates in panel o's window; panel e's window starts one tile earlier and ends one tile
// earlier, so its rows / columns >= rend_e contribute nothing.  L and Z of panel o are prefetched while panel e's products run.
__device__ __forceinline__ void update_tile2(double *lds, LdltMat M, int64_t k, int nbe, int64_t w0, int64_t rend,
                                             const double *__restrict__ Zws, int64_t ke, int nbe_e, int64_t w0e, int64_t rend_e,
                                             const double *__restrict__ Zwe, int64_t ldz, int64_t ti, int64_t tj)
{
    double *Ls = lds;                // [m][row of tile ti]
    double *Zs = lds + 64 * LVBA_TS; // [m][row of tile tj] (= column of the updated tile)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t r0 = w0 + 64 * ti, c0 = w0 + 64 * tj;
    const int row = tid & 63;
    const int i = lane & 15, kk = lane >> 4;
    const int64_t rr = r0 + row, cc = c0 + row;
    double lv[16], zv[16], lv2[16], zv2[16];
#pragma unroll
    for (int it = 0; it < 16; ++it) { // panel e first
        const int m = w + 4 * it;
        lv[it] = (rr < rend_e && m < nbe_e) ? M.a[rr + (ke + m) * M.ld] : 0.0;
        zv[it] = (cc < rend_e && m < nbe_e) ? Zwe[(cc - w0e) + m * ldz] : 0.0;
    }
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int m = w + 4 * it;
        lv2[it] = (rr < rend && m < nbe) ? M.a[rr + (k + m) * M.ld] : 0.0;
        zv2[it] = (cc < rend && m < nbe) ? Zws[(cc - w0) + m * ldz] : 0.0;
    }
    double cv[16];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int64_t c = c0 + 16 * w + kk + 4 * reg, r = r0 + 16 * t + i;
            cv[4 * t + reg] = (r < rend && c < rend && r >= c) ? M.a[r + c * M.ld] : 0.0;
        }
    d4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        if (pass) __syncthreads(); // everybody is done reading panel e's tiles
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int m = w + 4 * it;
            Ls[m * LVBA_TS + row] = pass ? lv2[it] : lv[it];
            Zs[m * LVBA_TS + row] = pass ? zv2[it] : zv[it];
        }
        __syncthreads();
#pragma unroll 4
        for (int k0 = 0; k0 < 64; k0 += 4) {
            const double a = Zs[(k0 + kk) * LVBA_TS + 16 * w + i];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const double bv = Ls[(k0 + kk) * LVBA_TS + 16 * t + i];
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv, acc[t], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int64_t c = c0 + 16 * w + kk + 4 * reg, r = r0 + 16 * t + i;
            if (r < rend && c < rend && r >= c) M.a[r + c * M.ld] = cv[4 * t + reg] - acc[t][reg];
        }
}

// ------------------------------------------------------------------------------------ K3, 128 x 64 tiles
// The bulk of the trailing update as 128 x 64 tiles (two tile rows of one tile column), K in chunks of 32 columns, for one
// panel (K = 64) or a pair of panels (K = 128, panel e's columns first).  Measured on the 64 x 64 form (2 problems x 414 tiles
// of a paired update, tools/step_microbench): L / Z operand loads 6 us (L2-bound), C load + store 8 us (HBM-bound), MFMA + LDS
// 15 us -- and 34 us in total, because every workgroup did them one after the other and the two workgroups of a CU in step.  Here
//   * while the products of a chunk run from LDS, the next chunk's operands are on their way into registers, and the C entries
//     are fetched beside the last chunk's products;
//   * a wavefront owns 32 rows x 64 columns: 6 LDS operand reads per 8 MFMAs (5 per 4 before), and the Z rows are fetched once
//     per 128 rows of L;
//   * operands move 16 bytes per lane (two rows of a column), C entries as the MFMA layout has them;
//   * a paired update of both problems is ~410 workgroups: one round of the 2 x 256 slots the factorisation's LDS leaves.
struct PanelRef { int64_t k, w0, rend; int nbe; const double *Z; };
typedef __attribute__((address_space(1))) double gdouble;
typedef __attribute__((address_space(1))) char gchar;
#define LVBA_TL 144 // LDS stride of an L chunk column (128 rows), 16 mod 32 like LVBA_TS
#define LVBA_K3B_LDS (32 * LVBA_TL + 32 * LVBA_TS) // doubles
// column-major enumeration of the 128 x 64 tiles of tile columns [ca, cb) of a Tb x Tb lower triangle: column tj holds the row
// pairs (tj + 2u, tj + 2u + 1), u < (Tb - tj + 1) / 2 (the last pair of a column may be a single tile row)
__host__ __device__ __forceinline__ int64_t pair_col_items(int64_t tj, int64_t Tb) { return (Tb - tj + 1) / 2; }
__device__ __forceinline__ bool pair_decode(int64_t j, int64_t ca, int64_t cb, int64_t Tb, int64_t &R0, int64_t &tj)
{
    for (int64_t c = ca; c < cb; ++c) {
        const int64_t n = pair_col_items(c, Tb);
        if (j < n) { tj = c; R0 = c + 2 * j; return true; }
        j -= n;
    }
    return false;
}
// Buffer addressing (resource + 32-bit lane offset + 32-bit scalar offset): the 60 loads / 32 stores of a tile then need three
// lane-offset registers between them; as flat 64-bit pointers their addresses alone filled > 100 VGPRs and spilled.
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
#define LVBA_BUF_WORD3 0x00020000 // raw buffer, 32-bit data format (gfx90a / gfx94x / gfx950)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_of(const double *p)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(p), 0, 0xFFFFFFF0u, LVBA_BUF_WORD3); // no range to check against
}
__device__ __forceinline__ double buf_ld(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    const v2u a = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return __hiloint2double(a.y, a.x);
}
__device__ __forceinline__ double2 buf_ld2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    const v4u a = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_double2(__hiloint2double(a.y, a.x), __hiloint2double(a.w, a.z));
}
__device__ __forceinline__ void buf_st(__amdgpu_buffer_rsrc_t r, double v, unsigned voff, unsigned soff)
{
    const v2u a = {(unsigned)__double2loint(v), (unsigned)__double2hiint(v)};
    __builtin_amdgcn_raw_buffer_store_b64(a, r, voff, soff, 0);
}
#ifdef LVBA_STAMPS
#define LVBA_BSTAMP(m_) do { if (bst && threadIdx.x == 0) bst[m_] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define LVBA_BSTAMP(m_) do { } while (0)
#endif
template <int nch> // K chunks of 32: 2 = one panel, 4 = a pair (pe, then po)
__device__ __forceinline__ void bulk_tile_128(double *lds, LdltMat M, const PanelRef po, const PanelRef pe, int64_t ldz64, int64_t R0,
                                              int64_t tj, unsigned long long *bst = nullptr)
{
    LVBA_BSTAMP(0);
    // a chunk buffer: Ls[m][row 0..127] at its start, Zs[m][row 0..63] behind it (+ 32 * LVBA_TL), m = column of the chunk
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned w = (unsigned)__builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, kk = lane >> 4;
    const int64_t r0 = po.w0 + 64 * (R0 + 1), c0 = po.w0 + 64 * (tj + 1);
    const bool two = r0 + 64 < po.rend; // the second tile row exists (else wavefronts 2, 3 have nothing to multiply)
    // all offsets below are BYTES in 32 bits: one problem's band storage is < 2^32 bytes (ld * n * 8 = 256 MB at C3)
    const unsigned ld = (unsigned)M.ld, ldz = (unsigned)ldz64;
    const __amdgpu_buffer_rsrc_t rA = buf_of(M.a), rZo = buf_of(po.Z), rZe = buf_of(nch == 4 ? pe.Z : po.Z);
    // operand fetch: L chunk = 128 rows x 32 columns, lane -> rows 2 lane, 2 lane + 1 of column w + 4 it (it < 8);
    //                Z chunk =  64 rows x 32 columns, lane -> rows 2 (lane & 31), + 1 of column 2 (w + 4 it) + (lane >> 5) (it < 4)
    // Two register sets: the chunk being staged and the next one in flight (L: [0..15], Z: [16..23]); set A also takes the C
    // entries ([0..31]) once the last even chunk has left it -- as arrays of their own the compiler gives them registers of
    // their own and spills.
    double xa[32], xb[24];
    const int lrow = 2 * lane, zrow = 2 * (lane & 31), zc = lane >> 5;
    const unsigned lvoff = 8u * (unsigned)lrow, zvoff = 8u * ((unsigned)zrow + (unsigned)zc * ldz);
    auto fetch = [&](int ch, double *xs) {
        const bool use_e = nch == 4 && ch < 2;
        const unsigned qk = (unsigned)(use_e ? pe.k : po.k), zr = (unsigned)(c0 - (use_e ? pe.w0 : po.w0));
        const unsigned m0 = 32u * (unsigned)(ch & 1);
        const unsigned lsoff = 8u * ((unsigned)r0 + (qk + m0 + w) * ld), zsoff = 8u * (zr + (m0 + 2u * w) * ldz);
        // unconditional: what lies outside the window is masked when it is stored to LDS (the band storage's columns overlap
        // their neighbours', and block_system.hip leaves slack behind the last one)
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const double2 v = buf_ld2(rA, lvoff, lsoff + (unsigned)it * (32u * ld));
            xs[2 * it] = v.x; xs[2 * it + 1] = v.y;
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const double2 v = buf_ld2(use_e ? rZe : rZo, zvoff, zsoff + (unsigned)it * (64u * ldz));
            xs[16 + 2 * it] = v.x; xs[17 + 2 * it] = v.y;
        }
    };
    auto stage = [&](int ch, double *xs) { // registers -> LDS, masking rows / columns outside the panel's window
        double *Ls = lds, *Zs = Ls + 32 * LVBA_TL;
        const bool use_e = nch == 4 && ch < 2;
        const int64_t qrend = use_e ? pe.rend : po.rend;
        const int qnbe = use_e ? pe.nbe : po.nbe;
        const int m0 = 32 * (ch & 1);
        if (!(r0 + 128 <= qrend && c0 + 64 <= qrend && qnbe == 64)) { // edge tiles only (wave-uniform)
            const bool l0 = r0 + lrow < qrend, l1 = r0 + lrow + 1 < qrend;
            const bool z0 = c0 + zrow < qrend, z1 = c0 + zrow + 1 < qrend;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const bool mok = m0 + (int)w + 4 * it < qnbe;
                xs[2 * it] = (l0 && mok) ? xs[2 * it] : 0.0;
                xs[2 * it + 1] = (l1 && mok) ? xs[2 * it + 1] : 0.0;
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const bool mok = m0 + 2 * ((int)w + 4 * it) + zc < qnbe;
                xs[16 + 2 * it] = (z0 && mok) ? xs[16 + 2 * it] : 0.0;
                xs[17 + 2 * it] = (z1 && mok) ? xs[17 + 2 * it] : 0.0;
            }
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) *reinterpret_cast<double2 *>(Ls + (w + 4 * it) * LVBA_TL + lrow) = make_double2(xs[2 * it], xs[2 * it + 1]);
#pragma unroll
        for (int it = 0; it < 4; ++it) *reinterpret_cast<double2 *>(Zs + (2 * (w + 4 * it) + zc) * LVBA_TS + zrow) = make_double2(xs[16 + 2 * it], xs[17 + 2 * it]);
    };
    // wavefront w: rows 32 w .. 32 w + 31 (two 16-row blocks tl) x 64 columns (four 16-column blocks cq);
    // acc[tl][cq][reg] <-> row r0 + 32 w + 16 tl + i, column c0 + 16 cq + kk + 4 reg
    d4 acc[2][4];
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
        for (int cq = 0; cq < 4; ++cq) acc[tl][cq] = (d4){0.0, 0.0, 0.0, 0.0};
    const bool busy = two || w < 2;
    auto products = [&](int ch) { // the chunk in LDS.  The operands of step k0 + 4 are read before the MFMAs of step k0 are issued: a
                                  // wavefront issues in order, and reads placed after them only start when the matrix pipe is draining
        if (!busy) return;
        // (an offset the compiler cannot see through: the sixteen LDS addresses of a chunk's k-steps are then formed per chunk instead of
        // being kept in registers across the whole tile -- with them the kernel needed 258 registers, and the reload of a spilled one,
        // placed behind the prefetch of the next chunk, made the first products of every tile wait for that prefetch to arrive:
        // C3 solve 4.10 -> 4.03 ms on the same box)
        int lo = 0;
        asm volatile("" : "+v"(lo));
        const double *Ls = lds + lo, *Zs = Ls + 32 * LVBA_TL;
        double a[2][4], bv[2][2];
        auto rd = [&](int k0, int q) {
#pragma unroll
            for (int cq = 0; cq < 4; ++cq) a[q][cq] = Zs[(k0 + kk) * LVBA_TS + 16 * cq + i];
#pragma unroll
            for (int tl = 0; tl < 2; ++tl) bv[q][tl] = Ls[(k0 + kk) * LVBA_TL + 32 * w + 16 * tl + i];
        };
        rd(0, 0);
#pragma unroll
        for (int k0 = 0; k0 < 32; k0 += 4) {
            const int q = (k0 >> 2) & 1;
            if (k0 + 4 < 32) rd(k0 + 4, q ^ 1);
            __builtin_amdgcn_sched_barrier(0); // keep the reads AHEAD of the MFMAs (LLVM's scheduler sinks them to their first use otherwise)
#pragma unroll
            for (int tl = 0; tl < 2; ++tl)
#pragma unroll
                for (int cq = 0; cq < 4; ++cq) acc[tl][cq] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q][cq], bv[q][tl], acc[tl][cq], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    const unsigned cvoff = 8u * ((unsigned)i + (unsigned)kk * ld);                 // lane part of a C entry's offset
    const unsigned csoff = 8u * ((unsigned)r0 + 32u * w + (unsigned)c0 * ld);      // + 128 tl, + 8 (16 cq + 4 reg) ld
    // Chunk c is staged from its register set (even chunks: A, odd: B) and the set is refilled at once with chunk c + 2, which
    // then has the products of two chunks to arrive in.  After the last even chunk, set A takes the C entries instead.
    // (fully unrolled: inside a loop the compiler's wait counts at the back edge drain every load in flight)
    auto load_c = [&]() { // Unmasked: entries outside the window or above the diagonal are read (inside the allocation, see
                          // block_system.hip) but never stored
#pragma unroll
        for (int cq = 0; cq < 4; ++cq)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const unsigned so = csoff + 8u * (unsigned)(16 * cq + 4 * reg) * ld;
#pragma unroll
                for (int tl = 0; tl < 2; ++tl) xa[16 * tl + 4 * cq + reg] = buf_ld(rA, cvoff + 128u * tl, so);
            }
    };
    fetch(0, xa);
    fetch(1, xb);
    LVBA_BSTAMP(1);
#pragma unroll
    for (int ch = 0; ch < nch; ch += 2) {
        if (ch > 0) __syncthreads(); // everybody is done with chunk ch - 1 in LDS
        stage(ch, xa);
        __syncthreads();
        LVBA_BSTAMP(2 + 2 * ch);
        if (ch + 2 < nch) fetch(ch + 2, xa);
        else if (busy) load_c();
        products(ch);
        __syncthreads();
        LVBA_BSTAMP(3 + 2 * ch);
        stage(ch + 1, xb);
        __syncthreads();
        LVBA_BSTAMP(4 + 2 * ch);
        if (ch + 3 < nch) fetch(ch + 3, xb);
        products(ch + 1);
        LVBA_BSTAMP(5 + 2 * ch);
    }
    if (busy) {
        const bool inner = r0 + 128 <= po.rend && r0 > c0; // whole tile inside the window and below the diagonal
        if (inner) { // (a branch of its own: with the test inside the loops every store sat behind ~15 instructions of mask logic)
#pragma unroll
            for (int cq = 0; cq < 4; ++cq)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const unsigned so = csoff + 8u * (unsigned)(16 * cq + 4 * reg) * ld;
#pragma unroll
                    for (int tl = 0; tl < 2; ++tl) buf_st(rA, xa[16 * tl + 4 * cq + reg] - acc[tl][cq][reg], cvoff + 128u * tl, so);
                }
        } else { // edge tiles (the diagonal, the window's last rows): 32-bit tile-local limits -- with the 64-bit tests spelled out per
                 // entry these were the slowest workgroups of every two-ended launch (27.5 us against a median of 21: stamps, round 6)
            const int rlim = (int)(po.rend - r0 < 128 ? po.rend - r0 : 128), clim = (int)(po.rend - c0 < 64 ? po.rend - c0 : 64);
            const int dd = (int)(c0 - r0); // <= 0; entry (rl, cl) lies in the lower triangle iff rl >= cl + dd
            const int rl0 = 32 * (int)w + i;
#pragma unroll
            for (int cq = 0; cq < 4; ++cq)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const unsigned so = csoff + 8u * (unsigned)(16 * cq + 4 * reg) * ld;
                    const int cl = 16 * cq + kk + 4 * reg;
#pragma unroll
                    for (int tl = 0; tl < 2; ++tl) {
                        const int rl = rl0 + 16 * tl;
                        if (rl < rlim && cl < clim && rl >= cl + dd)
                            buf_st(rA, xa[16 * tl + 4 * cq + reg] - acc[tl][cq][reg], cvoff + 128u * tl, so);
                    }
                }
        }
    }
    LVBA_BSTAMP(10);
}
