// host_tables.h -- host-side work partitioning of the evaluation kernels (header-only, no HIP: unit-tested on the CPU by
// tests/host_tables_check.cpp).
#pragma once
#include "host_arena.h"
#include <algorithm>
#include <cstdint>
#include <vector>

namespace lvba {

// Chunks of the voxel-major kernels: consecutive voxels with at most max_factors factors and max_voxels voxels per chunk (one
// workgroup each: lane = factor, then lane = voxel).  A voxel with more than max_factors observers sits alone in its chunk
// and is merged in tiles by the kernels.  voxel_off [n_voxels + 1]; chunk_v0 receives the first voxel of every chunk plus
// n_voxels; Q = sum k (k - 1) / 2.  Returns -1, or the index of the first voxel with fewer than two factors.
// breaks (optional, ascending voxel indices, n_breaks of them): a chunk never straddles one -- the voxel groups of a grouped
// refinement (lvba_balm_set_groups) sum their chunks' costs separately.
inline int64_t chunk_voxels(int64_t n_voxels, const int64_t *voxel_off, int max_factors, int max_voxels,
                            lvba::hvec<int64_t> &chunk_v0, int64_t &Q, const int64_t *breaks = nullptr, int64_t n_breaks = 0)
{
    chunk_v0.assign(1, 0);
    int64_t nf = 0, nv = 0, nb = 0;
    Q = 0;
    for (int64_t a = 0; a < n_voxels; ++a) {
        const int64_t k = voxel_off[a + 1] - voxel_off[a];
        if (k < 2) return a;
        Q += k * (k - 1) / 2;
        while (nb < n_breaks && breaks[nb] < a) ++nb;
        if (nb < n_breaks && breaks[nb] == a && nv > 0) { chunk_v0.push_back(a); nf = 0; nv = 0; }
        if (k > max_factors) {
            if (nv > 0) chunk_v0.push_back(a);
            chunk_v0.push_back(a + 1);
            nf = 0; nv = 0;
            continue;
        }
        if (nf + k > max_factors || nv == max_voxels) { chunk_v0.push_back(a); nf = 0; nv = 0; }
        nf += k; nv += 1;
    }
    if (chunk_v0.back() != n_voxels || chunk_v0.size() == 1) chunk_v0.push_back(n_voxels); // (an empty problem keeps one empty chunk)
    return -1;
}

// Work items of the pair pass.  One 16-lane group per block is right when there are many blocks (C3: 4e5 blocks of ~60
// pairs); with few blocks and long lists (window BA: 190 blocks x 2000 pairs) it leaves the chip empty, so lists longer
// than `cut` pairs become several items whose partial blocks are summed afterwards (balm_pair_reduce_kernel).
//   item i covers pairs [item_off[i], item_off[i+1]); item_dst[i] >= 0: the block slot it writes;
//   item_dst[i] < 0: partial block -(1 + item_dst[i]); block m of the cut ones sums partials [multi_off[m], multi_off[m+1])
//   into slot multi_slot[m].
inline int64_t pair_cut_length(int64_t Q)
{
    const int64_t cut = (Q / 4096 + 15) / 16 * 16;
    return std::max<int64_t>(64, std::min<int64_t>(cut, 512));
}
inline void cut_pair_items(const lvba::hvec<int64_t> &blk_slot, const lvba::hvec<int64_t> &blk_off, int64_t Q,
                           lvba::hvec<int64_t> &item_off, lvba::hvec<int64_t> &item_dst, lvba::hvec<int64_t> &multi_off,
                           lvba::hvec<int64_t> &multi_slot, int64_t &n_partial)
{
    item_off.assign(1, 0);
    item_dst.clear();
    multi_off.assign(1, 0);
    multi_slot.clear();
    n_partial = 0;
    const int64_t cut = pair_cut_length(Q);
    for (size_t bi = 0; bi < blk_slot.size(); ++bi) {
        const int64_t q0 = blk_off[bi], q1 = blk_off[bi + 1];
        if (q1 - q0 <= cut) {
            item_off.push_back(q1);
            item_dst.push_back(blk_slot[bi]);
        } else {
            for (int64_t q = q0; q < q1; q += cut) {
                item_off.push_back(std::min(q + cut, q1));
                item_dst.push_back(-(1 + n_partial++));
            }
            multi_off.push_back(n_partial);
            multi_slot.push_back(blk_slot[bi]);
        }
    }
}

// Work items of the lane-per-item pair kernel (balm_pair_lane_kernel): runs whose block slots may REPEAT (pair lists grouped by
// (voxel window, block): a block then appears in one run per window) and runs longer than `cut` pairs are cut; every piece is
// one item.  A block with a single item is written straight into the store; a block with several is summed from partial
// blocks in item order (= voxel window order: deterministic).  Partial indices follow the ITEM order, so that consecutive
// items store to consecutive partial blocks; the lists of a summed block's partials are multi_idx[multi_off[m] ..
// multi_off[m+1]).  run_slot [R], run_off [R+1]; n_slots bounds the slot values.
inline void group_pair_items(const lvba::hvec<int64_t> &run_slot, const lvba::hvec<int64_t> &run_off, int64_t cut, int64_t n_slots,
                             lvba::hvec<int64_t> &item_off, lvba::hvec<int64_t> &item_dst, lvba::hvec<int64_t> &multi_off,
                             lvba::hvec<int64_t> &multi_slot, lvba::hvec<int64_t> &multi_idx, int64_t &n_partial)
{
    item_off.assign(1, 0);
    item_dst.clear();
    multi_off.assign(1, 0);
    multi_slot.clear();
    multi_idx.clear();
    n_partial = 0;
    if (cut < 1) cut = 1;
    lvba::hvec<int32_t> cnt((size_t)n_slots, 0);
    lvba::hvec<int64_t> item_slot;
    for (size_t r = 0; r < run_slot.size(); ++r)
        for (int64_t q = run_off[r]; q < run_off[r + 1]; q += cut) {
            item_off.push_back(std::min(q + cut, run_off[r + 1]));
            item_slot.push_back(run_slot[r]);
            cnt[(size_t)run_slot[r]]++;
        }
    lvba::hvec<int64_t> base((size_t)n_slots, -1);
    int64_t tot = 0;
    for (int64_t sl = 0; sl < n_slots; ++sl)
        if (cnt[(size_t)sl] > 1) {
            base[(size_t)sl] = tot;
            tot += cnt[(size_t)sl];
            multi_slot.push_back(sl);
            multi_off.push_back(tot);
        }
    multi_idx.resize((size_t)tot);
    item_dst.resize(item_slot.size());
    for (size_t i = 0; i < item_slot.size(); ++i) {
        const int64_t sl = item_slot[i];
        if (base[(size_t)sl] < 0) { item_dst[i] = sl; continue; }
        multi_idx[(size_t)base[(size_t)sl]++] = n_partial;
        item_dst[i] = -(1 + n_partial++);
    }
}

} // namespace lvba
