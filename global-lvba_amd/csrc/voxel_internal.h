// voxel_internal.h -- pieces shared by the voxel front-end (voxelize.hip) and the window-BA driver (window_ba.hip):
// the device-resident scan set, rocPRIM wrappers on the caching pool, small utilities.
#pragma once
#include "host_arena.h"
#include <cstring>
#include <cstdint>
#include <rocprim/rocprim.hpp>
#include <chrono>
#include <vector>
#include "lvba_common.h"
#include "mempool.h"
#include "key_pack.h"

struct lvba_scans_s {
    int device = 0;
    int n_frames = 0;
    lvba::hvec<int64_t> frame_off; // [n_frames+1]
    float *d_pts = nullptr;         // [P][3]
    int64_t *d_frame_off = nullptr;
};

struct lvba_voxmap_s;
// the admitted voxels' clusters of a map, [n_factors][10] on the device (voxelize.hip; for the window driver's joint problem)
const double *lvba_voxmap_clusters(const lvba_voxmap_s *h);
// lvba_voxmap_build_scans on the caller's stream (the map does not own it)
struct lvba_scans_s;
int32_t lvba_voxmap_build_scans_on(lvba_scans_s *sc, int32_t frame_begin, int32_t n_frames, const double *poses,
                                   const lvba_voxel_opts *opts, hipStream_t stream, lvba_voxmap_s **out);

// ONE map for all windows of window_size frames in [frame_begin, frame_begin + n_frames) -- a root is (window, key) --, and the
// per-window views into it: what lvba_voxmap_build_scans_on would give for that window (same voxels in the same order, bit for
// bit), owning nothing; destroy the views before the joint map.  voxelize.hip.
int32_t lvba_voxmap_build_scans_joint(lvba_scans_s *sc, int32_t frame_begin, int32_t n_frames, int32_t window_size, const double *poses,
                                      const lvba_voxel_opts *opts, hipStream_t stream, lvba_voxmap_s **out);
int32_t lvba_voxmap_window_view(lvba_voxmap_s *joint, int32_t w, lvba_voxmap_s **out);
int32_t lvba_voxmap_window_range(lvba_voxmap_s *joint, int32_t w, int64_t *v0, int64_t *v1, int64_t *f0, int64_t *f1);

struct lvba_balm_s;
namespace lvba {
// lvba_balm_create_dev without the argument checks and the voxel re-layout pass (lvba_api.hip; for arrays this library made itself)
int32_t balm_create_dev_trusted(int32_t n_poses, int64_t n_voxels, const int64_t *voxel_off, const int32_t *pose_idx,
                                const double *d_clusters, int32_t device, lvba_balm_s **out);

inline double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
inline unsigned grid_for(int64_t n, int bs) { return (unsigned)((n + bs - 1) / bs); }

template <class K>
inline int32_t sort_pairs(hipStream_t s, const K *kin, K *kout, const uint32_t *vin, uint32_t *vout, size_t n,
                   unsigned end_bit)
{
    size_t bytes = 0;
    HIPCHK(rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, n, 0u, end_bit, s));
    DevBuf tmp(s);
    HIPCHK(tmp.alloc(bytes));
    HIPCHK(rocprim::radix_sort_pairs(tmp.p, bytes, kin, kout, vin, vout, n, 0u, end_bit, s));
    HIPCHK(hipStreamSynchronize(s));
    return LVBA_OK;
}
template <class T>
inline int32_t scan_incl(hipStream_t s, const T *in, T *out, size_t n)
{
    size_t bytes = 0;
    HIPCHK(rocprim::inclusive_scan(nullptr, bytes, in, out, n, rocprim::plus<T>(), s));
    DevBuf tmp(s);
    HIPCHK(tmp.alloc(bytes));
    HIPCHK(rocprim::inclusive_scan(tmp.p, bytes, in, out, n, rocprim::plus<T>(), s));
    HIPCHK(hipStreamSynchronize(s));
    return LVBA_OK;
}
template <class T>
inline int32_t scan_excl(hipStream_t s, const T *in, T *out, size_t n)
{
    size_t bytes = 0;
    HIPCHK(rocprim::exclusive_scan(nullptr, bytes, in, out, T(0), n, rocprim::plus<T>(), s));
    DevBuf tmp(s);
    HIPCHK(tmp.alloc(bytes));
    HIPCHK(rocprim::exclusive_scan(tmp.p, bytes, in, out, T(0), n, rocprim::plus<T>(), s));
    HIPCHK(hipStreamSynchronize(s));
    return LVBA_OK;
}


// ---- voxel keys shared by the plane map (voxelize.hip) and the depth grid map (fusion.hip) --------------------------
constexpr int KEY_BIAS = 1 << 20; // key components must lie in [-2^20, 2^20)
// (int64)(float)(p / vs), minus one for negatives (cut_voxel, bavoxel.hpp:809-815; the same rule at src/lvba_system.cpp:1289-1293
// and :1539-1544)
__device__ __forceinline__ bool root_key_of(const double pw[3], double vs, int64_t k[3])
{
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float loc = (float)(pw[j] / vs);
        if (loc < 0) loc = (float)((double)loc - 1.0);
        ok = ok && (fabsf(loc) < (float)KEY_BIAS); // also false for NaN / inf
        k[j] = ok ? (int64_t)loc : 0;
    }
    return ok;
}
__device__ __forceinline__ uint64_t pack_key(const int64_t k[3])
{
    return ((uint64_t)(k[0] + KEY_BIAS) << 42) | ((uint64_t)(k[1] + KEY_BIAS) << 21) | (uint64_t)(k[2] + KEY_BIAS);
}

// ---- sorting on the bits that vary --------------------------------------------------------------------------------
// A packed key is 3 x 21 bits, but the points of a map span a few hundred voxels per axis: the radix sort by root key (and
// by anchor leaf, window_ba.hip) spent eight passes over 64-bit keys on 20-30 bits of information (1.8 of the 3.3 ms of a 16 M
// point map).  The kernel that makes the keys also reduces the RANGE of the three (biased) components -- per wavefront, and a
// wavefront touches the global range only where it widens it: after the first few, none does --, the host reads it with the
// error flag it waits for anyway, and the keys are re-packed as  (x - x0) << (by + bz) | (y - y0) << bz | (z - z0):
// the same lexicographic order, hence the same stable sort, in 32 bits whenever bx + by + bz <= 32.
// (KeyPack, key_pack_of, key_compress, key_expand: key_pack.h -- plain arithmetic, also compiled by tests/key_pack_check.cpp)
// max over the 64 lanes of a wavefront, result in lane 63: four row shifts and two row broadcasts on the DPP path of the
// vector ALU (as ds_bpermute shuffles the six reductions cost the key kernel 0.2 ms per 16 M points)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i32(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xf, false); }
__device__ __forceinline__ int wave_max_to_lane63(int v) // v >= 0
{
    constexpr int id = 0; // what a lane without a source sees
    auto op = [](int a, int b) { return max(a, b); };
    v = op(v, dpp_i32<0x111, 0xf>(id, v)); // row_shr:1
    v = op(v, dpp_i32<0x112, 0xf>(id, v)); // row_shr:2
    v = op(v, dpp_i32<0x114, 0xf>(id, v)); // row_shr:4
    v = op(v, dpp_i32<0x118, 0xf>(id, v)); // row_shr:8   -> lane 15 of every row of 16: the row
    v = op(v, dpp_i32<0x142, 0xa>(id, v)); // row_bcast:15 -> rows 1 and 3 take in rows 0 and 2
    v = op(v, dpp_i32<0x143, 0xc>(id, v)); // row_bcast:31 -> rows 2 and 3 take in rows 0 + 1
    return v;
}
// every lane of the wavefront calls this (lanes without a point: valid = false); kb = biased components.  The wavefront's six
// maxima go to its own slot of `partial` ([wavefronts of the launch][6]) -- no atomics, nothing read back: conditional atomics on
// the global range cost 0.25 ms per 16 M points in the round trip of the read alone (and 16 ms unconditionally) --,
// key_range_reduce_kernel folds the slots into rng[6] (zeroed by the caller).
__device__ __forceinline__ void key_range_update(int *__restrict__ partial, const int kb[3], bool valid)
{
    int v[6];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        v[j] = wave_max_to_lane63(valid ? KEY_MAXC - kb[j] : 0);
        v[3 + j] = wave_max_to_lane63(valid ? kb[j] : 0);
    }
    if ((threadIdx.x & 63) == 63) {
        int *slot = partial + 6 * ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
#pragma unroll
        for (int j = 0; j < 6; ++j) slot[j] = v[j];
    }
}
inline int64_t key_range_slots(int64_t n_points, int block) { return (n_points + block - 1) / block * (block / 64); } // wavefronts
// (at most 64 workgroups, one atomic per workgroup and component: 1 024 wavefronts x 6 atomics on six addresses took 70 us --
// same-address atomics retire one per ~11 ns)
static __global__ __launch_bounds__(256) void key_range_reduce_kernel(int64_t n_slots, const int *__restrict__ partial, int *__restrict__ rng)
{
    __shared__ int part[4][6];
    int v[6] = {0, 0, 0, 0, 0, 0};
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_slots; w += (int64_t)gridDim.x * blockDim.x)
#pragma unroll
        for (int j = 0; j < 6; ++j) v[j] = max(v[j], partial[6 * w + j]);
#pragma unroll
    for (int j = 0; j < 6; ++j) v[j] = wave_max_to_lane63(v[j]);
    if ((threadIdx.x & 63) == 63)
#pragma unroll
        for (int j = 0; j < 6; ++j) part[threadIdx.x >> 6][j] = v[j];
    __syncthreads();
    if (threadIdx.x < 6) {
        const int j = threadIdx.x;
        const int m = max(max(part[0][j], part[1][j]), max(part[2][j], part[3][j]));
        if (m > 0) atomicMax(rng + j, m);
    }
}
inline unsigned key_range_reduce_grid(int64_t n_slots) { return (unsigned)std::min<int64_t>(64, (n_slots + 255) / 256); }
template <class K>
__global__ void key_compress_kernel(int64_t n, const uint64_t *key, const KeyPack kp, K *out /* may be key */)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = key_compress<K>(key[i], kp);
}

} // namespace lvba
