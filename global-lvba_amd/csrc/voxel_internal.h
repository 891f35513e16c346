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

inline int32_t sort_pairs(hipStream_t s, const uint64_t *kin, uint64_t *kout, const uint32_t *vin, uint32_t *vout, size_t n,
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

} // namespace lvba
