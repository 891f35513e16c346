// pair_lists.hip -- device construction of the block-major pair lists of the pair pass.
//
// Every voxel with k factors contributes k(k-1)/2 off-diagonal products -Y_x Y_y^T; the pair pass wants them grouped by
// destination block (I, J) of the pose-block matrix, blocks visited in 8 x 8 tiles, pairs of a block in voxel order.  That
// is one stable sort of Q = sum k(k-1)/2 records -- 2.3e7 at C3, and growing like k^2 for a global problem whose anchors
// overlap heavily -- so it runs here: one thread per pair writes (key, value), rocPRIM sorts on the minimal number of key
// bits, a run-length encode yields the non-empty blocks and their list lengths.  The host keeps the O(#blocks) tail
// (work-item cutting) only.  Replaces three O(Q) host loops that took 0.24 s of a 0.5 s set-up at C3.
//
// key   = tile(J/8, I/8) << 6 | (J%8) << 3 | (I%8)   -- ascending key == (tile, block slot) order of the former host sort
// value = (pos_x, pos_y) of the two factors in the pose-major Y array, swapped so that x belongs to the larger block index
#include <algorithm>
#include <cstring>
#include <cstdint>
#include <rocprim/rocprim.hpp>
#include "host_arena.h"
#include "lvba_common.h"
#include "mempool.h"
#include "pair_lists.h"

namespace lvba {

namespace {

__global__ __launch_bounds__(256) void pair_gen_kernel(const int64_t *__restrict__ voff, const int64_t *__restrict__ poff,
                                                       const int32_t *__restrict__ blk_of, const int32_t *__restrict__ pos_of,
                                                       int64_t G, int64_t Q, int64_t tiles_per_row, int64_t window_groups,
                                                       unsigned block_bits, uint64_t *__restrict__ keys,
                                                       uint64_t *__restrict__ vals)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Q) return;
    // group of pair q: last a with poff[a] <= q
    int64_t lo = 0, hi = G;
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (poff[mid] <= q) lo = mid; else hi = mid;
    }
    const int64_t f0 = voff[lo], k = voff[lo + 1] - f0, t = q - poff[lo];
    // (x, y), x < y, in row-major order of the strict upper triangle: row x starts at x(2k - x - 1)/2
    const double b = (double)(2 * k - 1);
    int64_t x = (int64_t)((b - sqrt(b * b - 8.0 * (double)t)) * 0.5);
    if (x < 0) x = 0;
    while (x > 0 && x * (2 * k - x - 1) / 2 > t) --x;
    while ((x + 1) * (2 * k - x - 2) / 2 <= t) ++x;
    const int64_t y = x + 1 + (t - x * (2 * k - x - 1) / 2);
    int32_t I = blk_of[f0 + x], J = blk_of[f0 + y];
    uint32_t px = (uint32_t)pos_of[f0 + x], py = (uint32_t)pos_of[f0 + y];
    if (I < J) {
        const int32_t ti = I; I = J; J = ti;
        const uint32_t tp = px; px = py; py = tp;
    }
    const uint64_t tile = (uint64_t)(J >> 3) * (uint64_t)tiles_per_row + (uint64_t)(I >> 3);
    uint64_t key = tile << 6 | (uint64_t)(J & 7) << 3 | (uint64_t)(I & 7);
    if (window_groups > 0) key |= (uint64_t)(lo / window_groups) << block_bits;
    keys[q] = key;
    vals[q] = (uint64_t)px | (uint64_t)py << 32;
}

__global__ void csc_key_kernel(int64_t F, const int32_t *__restrict__ pidx, const int32_t *__restrict__ iperm,
                               uint32_t *__restrict__ key, uint32_t *__restrict__ val, int32_t *__restrict__ blk_of)
{
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const int32_t b = iperm[pidx[f]];
    key[f] = (uint32_t)b;
    val[f] = (uint32_t)f;
    blk_of[f] = b;
}
__global__ void csc_tables_kernel(int64_t F, int64_t G, const uint32_t *__restrict__ val_s, const int64_t *__restrict__ voff,
                                  int32_t *__restrict__ csc_f, int32_t *__restrict__ group_of_pos, int32_t *__restrict__ pos_of)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= F) return;
    const int64_t f = val_s[t];
    csc_f[t] = (int32_t)f;
    pos_of[f] = (int32_t)t;
    int64_t lo = 0, hi = G; // group of factor f: last a with voff[a] <= f (empty groups are never the last)
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (voff[mid] <= f) lo = mid; else hi = mid;
    }
    group_of_pos[t] = (int32_t)lo;
}
__global__ void csc_off_kernel(int32_t N, int64_t F, const uint32_t *__restrict__ key_s, int64_t *__restrict__ csc_off)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > N) return;
    int64_t lo = 0, hi = F; // first sorted position whose key is >= i
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)key_s[mid] < i) lo = mid + 1; else hi = mid;
    }
    csc_off[i] = lo;
}

__global__ void adj_kernel(const int64_t *__restrict__ voff, const int64_t *__restrict__ poff, const int32_t *__restrict__ pidx,
                           int64_t G, int64_t Q, int32_t N, uint8_t *__restrict__ adj)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Q) return;
    int64_t lo = 0, hi = G;
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (poff[mid] <= q) lo = mid; else hi = mid;
    }
    const int64_t f0 = voff[lo], k = voff[lo + 1] - f0, t = q - poff[lo];
    const double b = (double)(2 * k - 1);
    int64_t x = (int64_t)((b - sqrt(b * b - 8.0 * (double)t)) * 0.5);
    if (x < 0) x = 0;
    while (x > 0 && x * (2 * k - x - 1) / 2 > t) --x;
    while ((x + 1) * (2 * k - x - 2) / 2 <= t) ++x;
    const int64_t y = x + 1 + (t - x * (2 * k - x - 1) / 2);
    const int32_t i = pidx[f0 + x], j = pidx[f0 + y];
    adj[(size_t)i * N + j] = 1; // every writer stores the same value: no atomics needed
    adj[(size_t)j * N + i] = 1;
}

// pairs of piece r move from old_off[r] .. to new_off[r] .. (new_off ascending): one thread per pair
__global__ void pair_permute_kernel(int64_t Q, int64_t n_pieces, const int64_t *__restrict__ new_off, const int64_t *__restrict__ old_off,
                                    const uint64_t *__restrict__ src, uint64_t *__restrict__ dst)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Q) return;
    int64_t lo = 0, hi = n_pieces; // last piece with new_off <= q
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (new_off[mid] <= q) lo = mid; else hi = mid;
    }
    dst[q] = src[old_off[lo] + (q - new_off[lo])];
}

} // namespace

// Co-visibility (byte adjacency, N x N, caller pose indices) of the local factors: one thread per pair of observers of a
// voxel.  h_adj [N*N] is overwritten.
int32_t adjacency_build(hipStream_t s, int64_t G, const int64_t *h_voff, int64_t F, const int32_t *h_pidx, int32_t N, int64_t Q,
                        uint8_t *h_adj)
{
    const size_t nn = (size_t)N * N;
    DevBuf d_adj(s), d_voff(s), d_poff(s), d_pidx(s);
    HIPCHK(d_adj.alloc(nn));
    HIPCHK(hipMemsetAsync(d_adj.p, 0, nn, s));
    if (Q > 0) {
        lvba::hvec<int64_t> poff((size_t)G + 1, 0);
        for (int64_t a = 0; a < G; ++a) {
            const int64_t k = h_voff[a + 1] - h_voff[a];
            poff[a + 1] = poff[a] + k * (k - 1) / 2;
        }
        if (poff[G] != Q) return LVBA_ERR_STATE;
        HIPCHK(d_voff.alloc(8 * ((size_t)G + 1))); HIPCHK(d_poff.alloc(8 * ((size_t)G + 1))); HIPCHK(d_pidx.alloc(4 * (size_t)F));
        HIPCHK(lvba::copy_h2d(d_voff.p, h_voff, 8 * ((size_t)G + 1)));
        HIPCHK(lvba::copy_h2d(d_poff.p, poff.data(), 8 * ((size_t)G + 1)));
        HIPCHK(lvba::copy_h2d(d_pidx.p, h_pidx, 4 * (size_t)F)); // (pieces below the pinning threshold: mempool.h)
        adj_kernel<<<(unsigned)((Q + 255) / 256), 256, 0, s>>>((const int64_t *)d_voff.p, (const int64_t *)d_poff.p,
                                                               (const int32_t *)d_pidx.p, G, Q, N, (uint8_t *)d_adj.p);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(s)); // poff is a local
    }
    HIPCHK(hipMemcpyAsync(h_adj, d_adj.p, nn, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return LVBA_OK;
}

// Pose-major (CSC) view of the factors: position t of the pose-major order holds factor csc_f[t] of group
// group_of_pos[t]; pos_of is its inverse; csc_off [N+1] the pose segments.  One stable sort of (pose block, factor) --
// factors of a pose stay in voxel order, exactly what the former host loop over the groups produced.  d_blk_of [F] receives
// the solver-order pose block of every factor (input of pair_lists_build).
int32_t csc_build(hipStream_t s, int64_t G, const int64_t *h_voff, int64_t F, const int32_t *h_pidx, int32_t N,
                  const int32_t *h_iperm, int32_t *d_csc_f, int32_t *d_group_of_pos, int32_t *d_pos_of, int64_t *d_csc_off,
                  int32_t *d_blk_of)
{
    DevBuf d_pidx(s), d_iperm(s), d_voff(s), key(s), val(s), key_s(s), val_s(s), tmp(s);
    HIPCHK(d_iperm.alloc(4 * (size_t)std::max(N, 1)));
    HIPCHK(hipMemcpyAsync(d_iperm.p, h_iperm, 4 * (size_t)N, hipMemcpyHostToDevice, s));
    if (F > 0) {
        HIPCHK(d_pidx.alloc(4 * (size_t)F)); HIPCHK(d_voff.alloc(8 * ((size_t)G + 1)));
        HIPCHK(key.alloc(4 * (size_t)F)); HIPCHK(val.alloc(4 * (size_t)F)); HIPCHK(key_s.alloc(4 * (size_t)F)); HIPCHK(val_s.alloc(4 * (size_t)F));
        HIPCHK(lvba::copy_h2d(d_pidx.p, h_pidx, 4 * (size_t)F)); // (pieces below the pinning threshold: mempool.h)
        HIPCHK(lvba::copy_h2d(d_voff.p, h_voff, 8 * ((size_t)G + 1)));
        csc_key_kernel<<<(unsigned)((F + 255) / 256), 256, 0, s>>>(F, (const int32_t *)d_pidx.p, (const int32_t *)d_iperm.p,
                                                                   (uint32_t *)key.p, (uint32_t *)val.p, d_blk_of);
        HIPCHK(hipGetLastError());
        unsigned end_bit = 1;
        while (end_bit < 32 && ((uint32_t)N >> end_bit) != 0) ++end_bit;
        size_t bytes = 0;
        HIPCHK(rocprim::radix_sort_pairs(nullptr, bytes, (const uint32_t *)key.p, (uint32_t *)key_s.p, (const uint32_t *)val.p,
                                         (uint32_t *)val_s.p, (size_t)F, 0u, end_bit, s));
        HIPCHK(tmp.alloc(bytes));
        HIPCHK(rocprim::radix_sort_pairs(tmp.p, bytes, (const uint32_t *)key.p, (uint32_t *)key_s.p, (const uint32_t *)val.p,
                                         (uint32_t *)val_s.p, (size_t)F, 0u, end_bit, s));
        csc_tables_kernel<<<(unsigned)((F + 255) / 256), 256, 0, s>>>(F, G, (const uint32_t *)val_s.p, (const int64_t *)d_voff.p, d_csc_f,
                                                                      d_group_of_pos, d_pos_of);
        HIPCHK(hipGetLastError());
    }
    csc_off_kernel<<<(unsigned)((N + 1 + 255) / 256), 256, 0, s>>>(N, F, (const uint32_t *)key_s.p, d_csc_off);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));
    return LVBA_OK;
}

int32_t pair_lists_build(hipStream_t s, int64_t G, const int64_t *h_voff, int64_t F, const int32_t *d_blk_of,
                         const int32_t *d_pos_of, int32_t N, int32_t Bb1, int64_t Q, int64_t window_groups, int64_t cut, int2 *d_pairs,
                         lvba::hvec<int64_t> &blk_slot, lvba::hvec<int64_t> &blk_off)
{
    blk_slot.clear();
    blk_off.assign(1, 0);
    if (Q <= 0) return LVBA_OK;
    lvba::hvec<int64_t> poff((size_t)G + 1, 0);
    for (int64_t a = 0; a < G; ++a) {
        const int64_t k = h_voff[a + 1] - h_voff[a];
        poff[a + 1] = poff[a] + k * (k - 1) / 2;
    }
    if (poff[G] != Q) return LVBA_ERR_STATE;
    const int64_t tiles_per_row = N / 8 + 1;
    const uint64_t max_key = ((uint64_t)tiles_per_row * (uint64_t)tiles_per_row) << 6;
    unsigned end_bit = 1;
    while (end_bit < 64 && (max_key >> end_bit) != 0) ++end_bit;
    const unsigned block_bits = end_bit;
    if (window_groups > 0) {
        const uint64_t n_win = (uint64_t)((G + window_groups - 1) / window_groups);
        unsigned wb = 1;
        while (wb < 40 && (n_win >> wb) != 0) ++wb;
        end_bit += wb;
        if (end_bit > 64) return LVBA_ERR_UNSUPPORTED;
    }
    const uint64_t block_mask = (block_bits >= 64) ? ~0ull : (((uint64_t)1 << block_bits) - 1);

    DevBuf d_voff(s), d_poff(s), k_in(s), k_out(s), v_in(s), uniq(s), cnt(s), nruns(s), tmp(s);
    HIPCHK(d_voff.alloc((size_t)(G + 1) * 8));
    HIPCHK(d_poff.alloc((size_t)(G + 1) * 8));
    HIPCHK(k_in.alloc((size_t)Q * 8));
    HIPCHK(k_out.alloc((size_t)Q * 8));
    HIPCHK(v_in.alloc((size_t)Q * 8));
    HIPCHK(lvba::copy_h2d(d_voff.p, h_voff, (size_t)(G + 1) * 8));
    HIPCHK(lvba::copy_h2d(d_poff.p, poff.data(), (size_t)(G + 1) * 8));
    pair_gen_kernel<<<(unsigned)((Q + 255) / 256), 256, 0, s>>>((const int64_t *)d_voff.p, (const int64_t *)d_poff.p,
                                                                d_blk_of, d_pos_of, G, Q, tiles_per_row, window_groups,
                                                                block_bits, (uint64_t *)k_in.p, (uint64_t *)v_in.p);
    HIPCHK(hipGetLastError());
    {
        size_t bytes = 0;
        HIPCHK(rocprim::radix_sort_pairs(nullptr, bytes, (const uint64_t *)k_in.p, (uint64_t *)k_out.p, (const uint64_t *)v_in.p,
                                         (uint64_t *)d_pairs, (size_t)Q, 0u, end_bit, s));
        HIPCHK(tmp.alloc(bytes));
        HIPCHK(rocprim::radix_sort_pairs(tmp.p, bytes, (const uint64_t *)k_in.p, (uint64_t *)k_out.p, (const uint64_t *)v_in.p,
                                         (uint64_t *)d_pairs, (size_t)Q, 0u, end_bit, s));
    }
    // non-empty blocks and their list lengths; the number of runs is bounded by the band profile N * Bb1 and by Q
    const int64_t max_runs = window_groups > 0 ? Q : std::min<int64_t>(Q, (int64_t)N * Bb1);
    HIPCHK(uniq.alloc((size_t)max_runs * 8));
    HIPCHK(cnt.alloc((size_t)max_runs * 4));
    HIPCHK(nruns.alloc(8));
    {
        size_t bytes = 0;
        HIPCHK(rocprim::run_length_encode(nullptr, bytes, (const uint64_t *)k_out.p, (size_t)Q, (uint64_t *)uniq.p,
                                          (uint32_t *)cnt.p, (uint64_t *)nruns.p, s));
        DevBuf tmp2(s);
        HIPCHK(tmp2.alloc(bytes));
        HIPCHK(rocprim::run_length_encode(tmp2.p, bytes, (const uint64_t *)k_out.p, (size_t)Q, (uint64_t *)uniq.p,
                                          (uint32_t *)cnt.p, (uint64_t *)nruns.p, s));
        HIPCHK(hipStreamSynchronize(s));
    }
    uint64_t n_runs = 0;
    HIPCHK(lvba::copy_d2h(&n_runs, nruns.p, 8));
    lvba::hvec<uint64_t> h_uniq((size_t)n_runs);
    lvba::hvec<uint32_t> h_cnt((size_t)n_runs);
    if (n_runs) {
        HIPCHK(lvba::copy_d2h(h_uniq.data(), uniq.p, (size_t)n_runs * 8));
        HIPCHK(lvba::copy_d2h(h_cnt.data(), cnt.p, (size_t)n_runs * 4));
    }
    auto slot_of_key = [&](uint64_t k64) {
        const uint64_t key = k64 & block_mask, tile = key >> 6;
        const int64_t J = (int64_t)(tile / (uint64_t)tiles_per_row) * 8 + (int64_t)((key >> 3) & 7);
        const int64_t I = (int64_t)(tile % (uint64_t)tiles_per_row) * 8 + (int64_t)(key & 7);
        return J * Bb1 + (I - J);
    };
    if (window_groups > 0 && cut > 0) {
        // Work items of the column-per-lane pair kernel: a wavefront walks ten consecutive items in lock-step and runs as long
        // as the longest of them.  So the runs are cut into pieces of <= cut pairs HERE and the pieces of a window are laid out
        // by length, longest first (stable: equal lengths stay in (tile, block) order): the ten items of a wavefront then have
        // about the same length (C3: ~11 pairs on average against ~25 for the longest of ten in (tile, block) order).  The pairs
        // move with their pieces (one gather of the sorted array).
        struct Piece { int64_t win, old_off, slot; int32_t len; };
        lvba::hvec<Piece> pc;
        pc.reserve((size_t)n_runs + (size_t)(Q / cut) + 1);
        int64_t o = 0;
        for (size_t r = 0; r < (size_t)n_runs; ++r) {
            const int64_t win = (int64_t)(h_uniq[r] >> block_bits), slot = slot_of_key(h_uniq[r]);
            for (int64_t q = 0; q < (int64_t)h_cnt[r]; q += cut)
                pc.push_back(Piece{win, o + q, slot, (int32_t)std::min<int64_t>(cut, (int64_t)h_cnt[r] - q)});
            o += (int64_t)h_cnt[r];
        }
        if (o != Q) return LVBA_ERR_STATE;
        std::stable_sort(pc.begin(), pc.end(), [](const Piece &a, const Piece &b) { return a.win != b.win ? a.win < b.win : a.len > b.len; });
        const size_t np = pc.size();
        lvba::hvec<int64_t> new_off(np + 1), old_off(np);
        blk_slot.resize(np);
        new_off[0] = 0;
        for (size_t i = 0; i < np; ++i) {
            blk_slot[i] = pc[i].slot;
            old_off[i] = pc[i].old_off;
            new_off[i + 1] = new_off[i] + pc[i].len;
        }
        blk_off = new_off;
        DevBuf d_new(s), d_old(s);
        HIPCHK(d_new.alloc((np + 1) * 8));
        HIPCHK(d_old.alloc(np * 8));
        HIPCHK(lvba::copy_h2d(d_new.p, new_off.data(), (np + 1) * 8));
        HIPCHK(lvba::copy_h2d(d_old.p, old_off.data(), np * 8));
        // v_in (the unsorted values) is free: gather into it, then back into the caller's array
        pair_permute_kernel<<<(unsigned)((Q + 255) / 256), 256, 0, s>>>(Q, (int64_t)np, (const int64_t *)d_new.p, (const int64_t *)d_old.p,
                                                                        (const uint64_t *)d_pairs, (uint64_t *)v_in.p);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(d_pairs, v_in.p, (size_t)Q * 8, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipStreamSynchronize(s));
        return LVBA_OK;
    }
    blk_slot.resize((size_t)n_runs);
    blk_off.resize((size_t)n_runs + 1);
    for (size_t r = 0; r < (size_t)n_runs; ++r) {
        blk_slot[r] = slot_of_key(h_uniq[r]);
        blk_off[r + 1] = blk_off[r] + (int64_t)h_cnt[r];
    }
    if (blk_off[n_runs] != Q) return LVBA_ERR_STATE;
    return LVBA_OK;
}

} // namespace lvba
