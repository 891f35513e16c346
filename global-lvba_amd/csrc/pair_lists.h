// pair_lists.h -- device construction of the block-major pair lists (pair_lists.hip).
#pragma once
#include "host_arena.h"
#include <hip/hip_runtime.h>
#include <cstdint>
#include <vector>

namespace lvba {

// voff [G+1]: factor ranges of the voxels; blk_of [F]: solver-order pose block of every factor (device); d_pos_of [F]:
// position of every factor in the pose-major Y array (device).  Writes the Q sorted (pos_x, pos_y) records to d_pairs
// (device, caller-allocated) and returns the non-empty block slots J * Bb1 + (I - J) in tile order with their list
// offsets.  window_groups > 0: the lists are grouped by (window of `window_groups` consecutive voxels, block) instead, so
// that the pairs processed at about the same time draw on one window's Y records (an L2-sized slice of every pose's
// segment); a block then appears in several runs (blk_slot repeats) and is summed from partial blocks.  With cut > 0 as well
// the runs come back cut into pieces of <= cut pairs, the pieces of a window ordered by length (longest first).  Synchronises
// the stream.
// Byte co-visibility matrix [N*N] of the local factors (caller pose indices), written to the host array h_adj.
int32_t adjacency_build(hipStream_t s, int64_t G, const int64_t *h_voff, int64_t F, const int32_t *h_pidx, int32_t N, int64_t Q,
                        uint8_t *h_adj);
// Pose-major tables (pair_lists.hip): csc_f, group_of_pos, pos_of [F], csc_off [N+1] and blk_of [F], all device arrays.
int32_t csc_build(hipStream_t s, int64_t G, const int64_t *h_voff, int64_t F, const int32_t *h_pidx, int32_t N,
                  const int32_t *h_iperm, int32_t *d_csc_f, int32_t *d_group_of_pos, int32_t *d_pos_of, int64_t *d_csc_off,
                  int32_t *d_blk_of);
int32_t pair_lists_build(hipStream_t s, int64_t G, const int64_t *h_voff, int64_t F, const int32_t *d_blk_of,
                         const int32_t *d_pos_of, int32_t N, int32_t Bb1, int64_t Q, int64_t window_groups, int64_t cut, int2 *d_pairs,
                         lvba::hvec<int64_t> &blk_slot, lvba::hvec<int64_t> &blk_off);

} // namespace lvba
