// voxelize.hip -- the voxel front-end that BUILDS the LiDAR-BA problem, on the device.
//
// Replaces the per-point hash-map insertion and recursive octree of the reference
//   cut_voxel                           include/BALM/bavoxel.hpp:799-836
//   OCTO_TREE_NODE::recut / cut_func    include/BALM/bavoxel.hpp:391-464 (+ judge_eigen :335-352)
//   OCTO_TREE_NODE::tras_opt            include/BALM/bavoxel.hpp:466-474 -> VOX_HESS::push_voxel :45-54
//   findCorrespondPoint + plane lookup  include/BALM/bavoxel.hpp:320-333, src/lvba_system.cpp:1531-1565
// by a sort-based formulation (no pointers, no hash map):
//   1. one lane per point: world transform, root voxel key (with the reference's fp32 quotient / "-1 if negative" /
//      truncation) and the two octant codes the point WOULD take if its root and child were split (the fp32 centre
//      arithmetic of cut_func is closed-form given the key, so no tree needs to exist yet);
//   2. three stable radix sorts (rocPRIM) on (root | frame | octant prefix) make every (node, frame) PointCluster of
//      every layer a contiguous run IN CLOUD ORDER, so one lane summing its run reproduces PointCluster::push
//      bit for bit (products of fp32 values are exact in fp64; the order of the sums is the reference's);
//   3. one wave per root voxel walks the (at most 1 + 8 + 64) nodes: merged world-frame covariance, eigen test,
//      PLANE / split / drop, admission (>= 2 observing frames) and the CSR emission that lvba_balm_create takes.
// The octree states stay on the device as two words per root for the landmark -> plane lookup of the visual stage.
#include <cstring>
#include <cstdint>
#include <rocprim/rocprim.hpp>
#include <vector>
#include <new>
#include "lvba_common.h"
#include "balm_math.h"

using namespace lvba;

namespace {

constexpr int KEY_BIAS = 1 << 20; // root key components must lie in [-2^20, 2^20)
enum : int { ST_NONE = 0, ST_DROP = 1, ST_PLANE = 2, ST_SPLIT = 3 };

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8); }
    template <class T> T *as() const { return (T *)p; }
    void *release() { void *q = p; p = nullptr; return q; }
};

// ---- shared per-point arithmetic (cut_voxel :809-815, root centre :826-829, cut_func :368-381) ------------------
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
__device__ __forceinline__ void octants_of(const double pw[3], const int64_t k[3], double vs, int &o1, int &o2)
{
    const float quater = (float)(vs / 4.0);
    o1 = 0; o2 = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float c0 = (float)((0.5 + (double)k[j]) * vs);
        const int b1 = pw[j] > (double)c0 ? 1 : 0;
        const float c1 = c0 + (float)(2 * b1 - 1) * quater;
        const int b2 = pw[j] > (double)c1 ? 1 : 0;
        o1 |= b1 << (2 - j);
        o2 |= b2 << (2 - j);
    }
}
__device__ __forceinline__ uint64_t pack_key(const int64_t k[3])
{
    return ((uint64_t)(k[0] + KEY_BIAS) << 42) | ((uint64_t)(k[1] + KEY_BIAS) << 21) | (uint64_t)(k[2] + KEY_BIAS);
}

// ---- 1. keys ----------------------------------------------------------------------------------------------------
__global__ void vox_key_kernel(int64_t P, const float *__restrict__ pts, const int64_t *__restrict__ frame_off,
                               int n_frames, const double *__restrict__ poses, double vs,
                               uint64_t *__restrict__ key, uint32_t *__restrict__ sec, uint32_t *__restrict__ idx,
                               int *__restrict__ err)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    int lo = 0, hi = n_frames; // frame f with frame_off[f] <= i < frame_off[f+1]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (frame_off[mid] <= i) lo = mid; else hi = mid;
    }
    const double *T = poses + 12 * (int64_t)lo;
    const double p0 = pts[3 * i], p1 = pts[3 * i + 1], p2 = pts[3 * i + 2];
    double pw[3];
    pw[0] = T[0] * p0 + T[1] * p1 + T[2] * p2 + T[9];
    pw[1] = T[3] * p0 + T[4] * p1 + T[5] * p2 + T[10];
    pw[2] = T[6] * p0 + T[7] * p1 + T[8] * p2 + T[11];
    int64_t k[3];
    if (!root_key_of(pw, vs, k)) { *err = 1; k[0] = k[1] = k[2] = 0; }
    int o1, o2;
    octants_of(pw, k, vs, o1, o2);
    key[i] = pack_key(k);
    sec[i] = ((uint32_t)lo << 6) | (uint32_t)(o1 << 3) | (uint32_t)o2;
    idx[i] = (uint32_t)i;
}

// head[i] = 1 where the sorted key changes
__global__ void vox_heads_kernel(int64_t n, const uint64_t *__restrict__ key, uint32_t *__restrict__ head)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    head[i] = (i == 0 || key[i] != key[i - 1]) ? 1u : 0u;
}
// segment tables from heads + their inclusive scan
__global__ void vox_segs_kernel(int64_t n, const uint64_t *__restrict__ key, const uint32_t *__restrict__ head,
                                const uint32_t *__restrict__ incl, uint64_t *__restrict__ seg_key,
                                uint32_t *__restrict__ seg_start)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (head[i]) {
        seg_key[incl[i] - 1] = key[i];
        seg_start[incl[i] - 1] = (uint32_t)i;
    }
    if (i == n - 1) seg_start[incl[i]] = (uint32_t)n;
}
// root id of every ORIGINAL point
__global__ void vox_rid_kernel(int64_t n, const uint32_t *__restrict__ incl, const uint32_t *__restrict__ idx0,
                               uint32_t *__restrict__ rid)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rid[idx0[i]] = incl[i] - 1;
}
// composite key (root id | frame | octant prefix of `level`): sorted==1 takes points in root-sorted order
__global__ void vox_comp_kernel(int64_t n, const uint32_t *__restrict__ rid, const uint32_t *__restrict__ sec,
                                const uint32_t *__restrict__ order, int shift, uint32_t mask,
                                uint64_t *__restrict__ comp, uint32_t *__restrict__ idx_out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t src = order ? order[i] : (uint32_t)i;
    comp[i] = ((uint64_t)rid[src] << shift) | (uint64_t)(sec[src] & mask);
    if (idx_out) idx_out[i] = (uint32_t)i;
}

// ---- 2. PointCluster of every (node, frame) run, summed in cloud order (tools.hpp:428-433) ---------------------------
__global__ void vox_cluster_kernel(int64_t nseg, const uint32_t *__restrict__ seg_start,
                                   const uint32_t *__restrict__ order, const float *__restrict__ pts,
                                   double *__restrict__ cl)
{
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nseg) return;
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, v0 = 0, v1 = 0, v2 = 0;
    const uint32_t b = seg_start[s], e = seg_start[s + 1];
    for (uint32_t i = b; i < e; ++i) {
        const float *q = pts + 3 * (int64_t)order[i];
        const double x = q[0], y = q[1], z = q[2];
        c0 += x * x; c1 += x * y; c2 += x * z; c3 += y * y; c4 += y * z; c5 += z * z;
        v0 += x; v1 += y; v2 += z;
    }
    double *o = cl + 10 * s;
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3; o[4] = c4; o[5] = c5; o[6] = v0; o[7] = v1; o[8] = v2;
    o[9] = (double)(e - b);
}

// first segment of every root in a level's table (tables are sorted by root id first)
__global__ void vox_rootrange_kernel(int64_t R, int64_t nseg, const uint64_t *__restrict__ seg_key, int shift,
                                     uint32_t *__restrict__ rs)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > R) return;
    int64_t lo = 0, hi = nseg; // first s with (seg_key[s] >> shift) >= r
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)(seg_key[mid] >> shift) < r) lo = mid + 1; else hi = mid;
    }
    rs[r] = (uint32_t)lo;
}

// ---- 3. per-root octree decisions and emission ---------------------------------------------------------------------
struct NodeArgs {
    int64_t R;
    const uint64_t *seg_key[3];
    const double *seg_cl[3];
    const uint32_t *rs[3];
    const double *poses;
    float ratio[3];
    int min_ps;
    uint32_t fmask; // frame field of a segment key: (key >> 6) & fmask
    // count mode outputs
    int32_t *n_plane, *n_vox;
    int64_t *n_fac;
    uint64_t *mask;
    uint32_t *rootinfo; // state0 | split1 << 8
    // write mode inputs (exclusive scans of the above) and outputs
    const int32_t *plane_first, *vox_first;
    const int64_t *fac_first;
    double *plane;      // [n_planes][6] centre, direct
    int64_t *vox_off;   // [V+1]
    int32_t *pose_idx;  // [F]
    double *clusters;   // [F][10]
    int32_t *vox_label; // [V][2] root id, layer | o1 << 4 | o2 << 8
};

// merged world-frame statistics of node (level, code) of root r, frames in ascending order (judge_eigen :337-344)
__device__ __forceinline__ int node_stats(const NodeArgs &a, int level, uint32_t code, int64_t r, double *S)
{
#pragma unroll
    for (int k = 0; k < 10; ++k) S[k] = 0.0;
    int nf = 0;
    const uint64_t *sk = a.seg_key[level];
    for (uint32_t s = a.rs[level][r], e = a.rs[level][r + 1]; s < e; ++s) {
        const uint64_t key = sk[s];
        if ((uint32_t)(key & 63u) != code) continue;
        const uint32_t f = (uint32_t)(key >> 6) & a.fmask;
        double T[10];
        transform_cluster(a.seg_cl[level] + 10 * (int64_t)s, a.poses + 12 * (int64_t)f, a.poses + 12 * (int64_t)f + 9, T);
#pragma unroll
        for (int k = 0; k < 10; ++k) S[k] += T[k];
        ++nf;
    }
    return nf;
}
// recut's decision for one node (:396-427); also returns the plane (centre, direct) of judge_eigen
__device__ __forceinline__ int node_decide(const double *S, int nf, int min_ps, float ratio, bool last_layer,
                                           double *plane)
{
    if (nf == 0) return ST_NONE;
    if (S[9] < (double)min_ps) return ST_DROP;
    double C[6], vb[3], lam[3], U[9];
    voxel_cov(S, C, vb);
    eig3<true>(C, lam, U);
    if (plane) {
        plane[0] = vb[0]; plane[1] = vb[1]; plane[2] = vb[2];
        plane[3] = U[0]; plane[4] = U[3]; plane[5] = U[6];
    }
    if (lam[0] / lam[2] > (double)ratio) return last_layer ? ST_DROP : ST_SPLIT;
    return ST_PLANE;
}

template <bool WRITE>
__global__ __launch_bounds__(64) void vox_node_kernel(NodeArgs a)
{
    const int64_t r = blockIdx.x;
    const int t = threadIdx.x, o1 = t >> 3, o2 = t & 7;
    __shared__ int s_state0, s_state1[8], s_nf[64];
    double S[10], plane[6];
    if (t == 0) {
        const int nf = node_stats(a, 0, 0u, r, S);
        s_state0 = node_decide(S, nf, a.min_ps, a.ratio[0], false, nullptr);
    }
    __syncthreads();
    const int st0 = s_state0;
    if (t < 8) {
        int st1 = ST_NONE;
        if (st0 == ST_SPLIT) {
            const int nf = node_stats(a, 1, (uint32_t)(t << 3), r, S);
            st1 = node_decide(S, nf, a.min_ps, a.ratio[1], false, nullptr);
        }
        s_state1[t] = st1;
    }
    __syncthreads();
    const int st1 = s_state1[o1];
    // the node this lane emits (lanes in path order: root -> lane 0, child o1 -> lane 8*o1, grandchild -> its own lane)
    int level = -1;
    uint32_t code = 0;
    if (st0 == ST_PLANE) { if (t == 0) level = 0; }
    else if (st0 == ST_SPLIT) {
        if (st1 == ST_PLANE) { if (o2 == 0) { level = 1; code = (uint32_t)(o1 << 3); } }
        else if (st1 == ST_SPLIT) { level = 2; code = (uint32_t)t; }
    }
    int nf = 0, st = ST_NONE;
    if (level >= 0) {
        nf = node_stats(a, level, code, r, S);
        st = node_decide(S, nf, a.min_ps, a.ratio[level], level == 2, plane);
    }
    const bool is_plane = st == ST_PLANE;
    const bool admitted = is_plane && nf >= 2;
    const uint64_t pmask = __ballot(is_plane);
    const uint64_t amask = __ballot(admitted);
    const uint64_t lt = t == 0 ? 0ull : (~0ull >> (64 - t));
    s_nf[t] = admitted ? nf : 0;
    __syncthreads();
    if (!WRITE) {
        if (t == 0) {
            int64_t fsum = 0;
            for (int k = 0; k < 64; ++k) fsum += s_nf[k];
            uint32_t split1 = 0;
            for (int k = 0; k < 8; ++k) split1 |= (s_state1[k] == ST_SPLIT ? 1u : 0u) << k;
            a.n_plane[r] = __popcll(pmask);
            a.n_vox[r] = __popcll(amask);
            a.n_fac[r] = fsum;
            a.mask[r] = pmask;
            a.rootinfo[r] = (uint32_t)st0 | (split1 << 8);
        }
        return;
    }
    if (is_plane) {
        double *o = a.plane + 6 * ((int64_t)a.plane_first[r] + __popcll(pmask & lt));
#pragma unroll
        for (int k = 0; k < 6; ++k) o[k] = plane[k];
    }
    if (admitted) {
        int64_t foff = a.fac_first[r];
        for (int k = 0; k < t; ++k) foff += s_nf[k];
        const int64_t v = (int64_t)a.vox_first[r] + __popcll(amask & lt);
        a.vox_off[v] = foff;
        a.vox_label[2 * v] = (int32_t)r;
        a.vox_label[2 * v + 1] = level | ((level >= 1 ? o1 : 0) << 4) | ((level == 2 ? o2 : 0) << 8);
        const uint64_t *sk = a.seg_key[level];
        for (uint32_t s = a.rs[level][r], e = a.rs[level][r + 1]; s < e; ++s) {
            const uint64_t key = sk[s];
            if ((uint32_t)(key & 63u) != code) continue;
            a.pose_idx[foff] = (int32_t)((uint32_t)(key >> 6) & a.fmask);
            const double *c = a.seg_cl[level] + 10 * (int64_t)s;
#pragma unroll
            for (int k = 0; k < 10; ++k) a.clusters[10 * foff + k] = c[k];
            ++foff;
        }
    }
}

// ---- landmark -> plane lookup (src/lvba_system.cpp:1531-1565) ------------------------------------------------------
__global__ void vox_lookup_kernel(int64_t n, const double *__restrict__ X, double vs, int64_t R,
                                  const uint64_t *__restrict__ root_key, const uint64_t *__restrict__ mask,
                                  const uint32_t *__restrict__ rootinfo, const int32_t *__restrict__ plane_first,
                                  const double *__restrict__ plane, double *__restrict__ out, uint8_t *__restrict__ valid)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double *o = out + 4 * i;
    o[0] = o[1] = o[2] = o[3] = 0.0;
    valid[i] = 0;
    const double pw[3] = {X[3 * i], X[3 * i + 1], X[3 * i + 2]};
    if (!(isfinite(pw[0]) && isfinite(pw[1]) && isfinite(pw[2]))) return;
    int64_t k[3];
    if (!root_key_of(pw, vs, k)) return;
    const uint64_t key = pack_key(k);
    int64_t lo = 0, hi = R;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (root_key[mid] < key) lo = mid + 1; else hi = mid;
    }
    if (lo >= R || root_key[lo] != key) return;
    const uint32_t info = rootinfo[lo];
    const int st0 = info & 0xff;
    int o1, o2;
    octants_of(pw, k, vs, o1, o2);
    int lane;
    if (st0 == ST_PLANE) lane = 0;
    else if (st0 == ST_SPLIT) lane = ((info >> (8 + o1)) & 1u) ? (o1 << 3 | o2) : (o1 << 3);
    else return;
    const uint64_t m = mask[lo];
    if (!((m >> lane) & 1ull)) return;
    const double *pl = plane + 6 * ((int64_t)plane_first[lo] + __popcll(lane ? (m & (~0ull >> (64 - lane))) : 0ull));
    const double nn = sqrt(pl[3] * pl[3] + pl[4] * pl[4] + pl[5] * pl[5]);
    if (!(isfinite(nn) && nn >= 1e-6 && isfinite(pl[0]) && isfinite(pl[1]) && isfinite(pl[2]))) return;
    const double n0 = pl[3] / nn, n1 = pl[4] / nn, n2 = pl[5] / nn;
    o[0] = n0; o[1] = n1; o[2] = n2;
    o[3] = -(n0 * pl[0] + n1 * pl[1] + n2 * pl[2]);
    valid[i] = 1;
}

inline unsigned grid_for(int64_t n, int bs) { return (unsigned)((n + bs - 1) / bs); }
inline int bits_for(uint64_t n) { int b = 1; while ((n >> b) != 0) ++b; return b; } // bits to hold values < n (n >= 1)

} // namespace

struct lvba_voxmap_s {
    int device = 0;
    hipStream_t stream = nullptr;
    int n_frames = 0;
    lvba_voxel_opts opts{};
    lvba_voxmap_info_t info{};
    // device-resident map
    uint64_t *d_root_key = nullptr, *d_mask = nullptr;
    uint32_t *d_rootinfo = nullptr;
    int32_t *d_plane_first = nullptr;
    double *d_plane = nullptr;
    // device-resident problem (CSR of admitted voxels)
    int64_t *d_vox_off = nullptr;
    int32_t *d_pose_idx = nullptr, *d_vox_label = nullptr;
    double *d_clusters = nullptr;
};

extern "C" void lvba_voxel_default_opts(lvba_voxel_opts *o)
{
    if (!o) return;
    o->voxel_size = 1.0;                                   // cut_voxel's default (bavoxel.hpp:801)
    o->eigen_ratio[0] = 0.3f; o->eigen_ratio[1] = 0.1f;    // include/dataset_io.h:77 (stage 1)
    o->eigen_ratio[2] = 0.06f; o->eigen_ratio[3] = 0.03f;
    o->min_points = 15;                                    // bavoxel.hpp:24
    o->layer_limit = 2;                                    // bavoxel.hpp:13
}

extern "C" int32_t lvba_voxmap_destroy(lvba_voxmap_t h)
{
    if (!h) return LVBA_OK;
    (void)hipSetDevice(h->device);
    void *ptrs[] = {h->d_root_key, h->d_mask, h->d_rootinfo, h->d_plane_first, h->d_plane,
                    h->d_vox_off, h->d_pose_idx, h->d_vox_label, h->d_clusters};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return LVBA_OK;
}

namespace {

template <class K>
int32_t sort_pairs(hipStream_t s, const K *kin, K *kout, const uint32_t *vin, uint32_t *vout, size_t n, unsigned end_bit)
{
    size_t bytes = 0;
    HIPCHK(rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, n, 0u, end_bit, s));
    DevBuf tmp;
    HIPCHK(tmp.alloc(bytes));
    HIPCHK(rocprim::radix_sort_pairs(tmp.p, bytes, kin, kout, vin, vout, n, 0u, end_bit, s));
    HIPCHK(hipStreamSynchronize(s));
    return LVBA_OK;
}
template <class T>
int32_t scan_incl(hipStream_t s, const T *in, T *out, size_t n)
{
    size_t bytes = 0;
    HIPCHK(rocprim::inclusive_scan(nullptr, bytes, in, out, n, rocprim::plus<T>(), s));
    DevBuf tmp;
    HIPCHK(tmp.alloc(bytes));
    HIPCHK(rocprim::inclusive_scan(tmp.p, bytes, in, out, n, rocprim::plus<T>(), s));
    HIPCHK(hipStreamSynchronize(s));
    return LVBA_OK;
}
template <class T>
int32_t scan_excl(hipStream_t s, const T *in, T *out, size_t n)
{
    size_t bytes = 0;
    HIPCHK(rocprim::exclusive_scan(nullptr, bytes, in, out, T(0), n, rocprim::plus<T>(), s));
    DevBuf tmp;
    HIPCHK(tmp.alloc(bytes));
    HIPCHK(rocprim::exclusive_scan(tmp.p, bytes, in, out, T(0), n, rocprim::plus<T>(), s));
    HIPCHK(hipStreamSynchronize(s));
    return LVBA_OK;
}

// heads -> scan -> segment table of a sorted key array; returns the number of segments
int32_t segment(hipStream_t s, int64_t n, const uint64_t *key_sorted, DevBuf &incl, DevBuf &seg_key, DevBuf &seg_start,
                int64_t *nseg)
{
    DevBuf head;
    HIPCHK(head.alloc(4 * n));
    HIPCHK(incl.alloc(4 * n));
    vox_heads_kernel<<<grid_for(n, 256), 256, 0, s>>>(n, key_sorted, head.as<uint32_t>());
    TRY(scan_incl<uint32_t>(s, head.as<uint32_t>(), incl.as<uint32_t>(), (size_t)n));
    uint32_t last = 0;
    HIPCHK(hipMemcpy(&last, incl.as<uint32_t>() + (n - 1), 4, hipMemcpyDeviceToHost));
    *nseg = last;
    HIPCHK(seg_key.alloc(8 * (size_t)last));
    HIPCHK(seg_start.alloc(4 * ((size_t)last + 1)));
    vox_segs_kernel<<<grid_for(n, 256), 256, 0, s>>>(n, key_sorted, head.as<uint32_t>(), incl.as<uint32_t>(),
                                                     seg_key.as<uint64_t>(), seg_start.as<uint32_t>());
    HIPCHK(hipGetLastError());
    return LVBA_OK;
}

int32_t voxmap_build_impl(lvba_voxmap_s *h, const void *const *frame_points, const int64_t *frame_count,
                          int32_t stride_bytes, const double *poses)
{
    const int nfr = h->n_frames;
    hipStream_t s = h->stream;
    std::vector<int64_t> foff(nfr + 1, 0);
    for (int f = 0; f < nfr; ++f) {
        if (frame_count[f] < 0 || (frame_count[f] > 0 && !frame_points[f]))
            return lvba_fail(LVBA_ERR_ARG, "frame %d: bad point count / null cloud", f);
        foff[f + 1] = foff[f] + frame_count[f];
    }
    const int64_t P = foff[nfr];
    if (P >= (int64_t)1 << 31) return lvba_fail(LVBA_ERR_UNSUPPORTED, "more than 2^31 points in one map (%lld)", (long long)P);
    h->info.n_points = P;
    if (P == 0) return LVBA_OK;

    // -- upload (xyz packed out of the caller's point stride, e.g. sizeof(pcl::PointXYZINormal))
    DevBuf pts, d_foff, d_poses;
    HIPCHK(pts.alloc(12 * (size_t)P));
    HIPCHK(d_foff.alloc(8 * (nfr + 1)));
    HIPCHK(d_poses.alloc(96 * (size_t)nfr));
    for (int f = 0; f < nfr; ++f)
        if (frame_count[f] > 0)
            HIPCHK(hipMemcpy2DAsync(pts.as<float>() + 3 * foff[f], 12, frame_points[f], (size_t)stride_bytes, 12,
                                    (size_t)frame_count[f], hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d_foff.p, foff.data(), 8 * (nfr + 1), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d_poses.p, poses, 96 * (size_t)nfr, hipMemcpyHostToDevice, s));

    // -- 1. keys, root sort (stable: cloud order inside a root == (frame, index) order)
    DevBuf key, sec, idx, key_s, idx0, d_err;
    HIPCHK(key.alloc(8 * P)); HIPCHK(sec.alloc(4 * P)); HIPCHK(idx.alloc(4 * P));
    HIPCHK(key_s.alloc(8 * P)); HIPCHK(idx0.alloc(4 * P)); HIPCHK(d_err.alloc(4));
    HIPCHK(hipMemsetAsync(d_err.p, 0, 4, s));
    vox_key_kernel<<<grid_for(P, 256), 256, 0, s>>>(P, pts.as<float>(), d_foff.as<int64_t>(), nfr, d_poses.as<double>(),
                                                    h->opts.voxel_size, key.as<uint64_t>(), sec.as<uint32_t>(),
                                                    idx.as<uint32_t>(), d_err.as<int>());
    HIPCHK(hipGetLastError());
    int err = 0;
    HIPCHK(hipMemcpyAsync(&err, d_err.p, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (err) return lvba_fail(LVBA_ERR_ARG, "a point is non-finite or its voxel key exceeds +-2^20 after the pose transform");
    TRY(sort_pairs<uint64_t>(s, key.as<uint64_t>(), key_s.as<uint64_t>(), idx.as<uint32_t>(), idx0.as<uint32_t>(), (size_t)P, 63));

    int64_t R = 0;
    DevBuf incl0, root_key, root_start, rid;
    TRY(segment(s, P, key_s.as<uint64_t>(), incl0, root_key, root_start, &R));
    HIPCHK(rid.alloc(4 * P));
    vox_rid_kernel<<<grid_for(P, 256), 256, 0, s>>>(P, incl0.as<uint32_t>(), idx0.as<uint32_t>(), rid.as<uint32_t>());
    HIPCHK(hipGetLastError());
    h->info.n_roots = R;

    // -- 2. per-level (node, frame) cluster tables
    const int fbits = bits_for((uint64_t)nfr), rbits = bits_for((uint64_t)R);
    const int shift = fbits + 6;
    if (fbits > 26 || shift + rbits > 64) return lvba_fail(LVBA_ERR_UNSUPPORTED, "frames x roots exceed the 64-bit sort key");
    DevBuf seg_key[3], seg_start[3], seg_cl[3], rs[3];
    int64_t nseg[3] = {0, 0, 0};
    {
        DevBuf comp, comp_s, idxL, iota;
        HIPCHK(comp.alloc(8 * P)); HIPCHK(comp_s.alloc(8 * P)); HIPCHK(idxL.alloc(4 * P)); HIPCHK(iota.alloc(4 * P));
        for (int L = 0; L < 3; ++L) {
            const uint32_t mask = L == 0 ? ~63u : (L == 1 ? ~7u : ~0u);
            const uint64_t *sorted_keys;
            const uint32_t *order;
            if (L == 0) { // the root sort already is the (root, frame, index) order
                vox_comp_kernel<<<grid_for(P, 256), 256, 0, s>>>(P, rid.as<uint32_t>(), sec.as<uint32_t>(), idx0.as<uint32_t>(),
                                                                 shift, mask, comp_s.as<uint64_t>(), nullptr);
                sorted_keys = comp_s.as<uint64_t>();
                order = idx0.as<uint32_t>();
            } else {
                vox_comp_kernel<<<grid_for(P, 256), 256, 0, s>>>(P, rid.as<uint32_t>(), sec.as<uint32_t>(), nullptr, shift, mask,
                                                                 comp.as<uint64_t>(), iota.as<uint32_t>());
                TRY(sort_pairs<uint64_t>(s, comp.as<uint64_t>(), comp_s.as<uint64_t>(), iota.as<uint32_t>(), idxL.as<uint32_t>(),
                                         (size_t)P, (unsigned)(shift + rbits)));
                sorted_keys = comp_s.as<uint64_t>();
                order = idxL.as<uint32_t>();
            }
            HIPCHK(hipGetLastError());
            DevBuf incl;
            TRY(segment(s, P, sorted_keys, incl, seg_key[L], seg_start[L], &nseg[L]));
            HIPCHK(seg_cl[L].alloc(80 * (size_t)nseg[L]));
            vox_cluster_kernel<<<grid_for(nseg[L], 64), 64, 0, s>>>(nseg[L], seg_start[L].as<uint32_t>(), order, pts.as<float>(),
                                                                    seg_cl[L].as<double>());
            HIPCHK(rs[L].alloc(4 * ((size_t)R + 1)));
            vox_rootrange_kernel<<<grid_for(R + 1, 256), 256, 0, s>>>(R, nseg[L], seg_key[L].as<uint64_t>(), shift, rs[L].as<uint32_t>());
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(s));
        }
    }

    // -- 3. octree decisions: count, scan, write
    DevBuf n_plane, n_vox, n_fac, mask, rootinfo, plane_first, vox_first, fac_first;
    HIPCHK(n_plane.alloc(4 * (R + 1))); HIPCHK(n_vox.alloc(4 * (R + 1))); HIPCHK(n_fac.alloc(8 * (R + 1)));
    HIPCHK(mask.alloc(8 * R)); HIPCHK(rootinfo.alloc(4 * R));
    HIPCHK(plane_first.alloc(4 * (R + 1))); HIPCHK(vox_first.alloc(4 * (R + 1))); HIPCHK(fac_first.alloc(8 * (R + 1)));
    HIPCHK(hipMemsetAsync(n_plane.p, 0, 4 * (R + 1), s));
    HIPCHK(hipMemsetAsync(n_vox.p, 0, 4 * (R + 1), s));
    HIPCHK(hipMemsetAsync(n_fac.p, 0, 8 * (R + 1), s));
    NodeArgs a{};
    a.R = R;
    for (int L = 0; L < 3; ++L) {
        a.seg_key[L] = seg_key[L].as<uint64_t>();
        a.seg_cl[L] = seg_cl[L].as<double>();
        a.rs[L] = rs[L].as<uint32_t>();
        a.ratio[L] = h->opts.eigen_ratio[L];
    }
    a.poses = d_poses.as<double>();
    a.min_ps = h->opts.min_points;
    a.fmask = (1u << fbits) - 1u;
    a.n_plane = n_plane.as<int32_t>(); a.n_vox = n_vox.as<int32_t>(); a.n_fac = n_fac.as<int64_t>();
    a.mask = mask.as<uint64_t>(); a.rootinfo = rootinfo.as<uint32_t>();
    vox_node_kernel<false><<<(unsigned)R, 64, 0, s>>>(a);
    HIPCHK(hipGetLastError());
    TRY(scan_excl<int32_t>(s, n_plane.as<int32_t>(), plane_first.as<int32_t>(), (size_t)R + 1));
    TRY(scan_excl<int32_t>(s, n_vox.as<int32_t>(), vox_first.as<int32_t>(), (size_t)R + 1));
    TRY(scan_excl<int64_t>(s, n_fac.as<int64_t>(), fac_first.as<int64_t>(), (size_t)R + 1));
    int32_t n_planes = 0, V = 0;
    int64_t F = 0;
    HIPCHK(hipMemcpy(&n_planes, plane_first.as<int32_t>() + R, 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(&V, vox_first.as<int32_t>() + R, 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(&F, fac_first.as<int64_t>() + R, 8, hipMemcpyDeviceToHost));
    h->info.n_planes = n_planes; h->info.n_voxels = V; h->info.n_factors = F;

    DevBuf plane, vox_off, pose_idx, clusters, vox_label;
    HIPCHK(plane.alloc(48 * (size_t)n_planes)); HIPCHK(vox_off.alloc(8 * ((size_t)V + 1)));
    HIPCHK(pose_idx.alloc(4 * (size_t)F)); HIPCHK(clusters.alloc(80 * (size_t)F)); HIPCHK(vox_label.alloc(8 * (size_t)V));
    a.plane_first = plane_first.as<int32_t>(); a.vox_first = vox_first.as<int32_t>(); a.fac_first = fac_first.as<int64_t>();
    a.plane = plane.as<double>(); a.vox_off = vox_off.as<int64_t>(); a.pose_idx = pose_idx.as<int32_t>();
    a.clusters = clusters.as<double>(); a.vox_label = vox_label.as<int32_t>();
    vox_node_kernel<true><<<(unsigned)R, 64, 0, s>>>(a);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(vox_off.as<int64_t>() + V, &F, 8, hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));

    h->d_root_key = (uint64_t *)root_key.release();
    h->d_mask = (uint64_t *)mask.release();
    h->d_rootinfo = (uint32_t *)rootinfo.release();
    h->d_plane_first = (int32_t *)plane_first.release();
    h->d_plane = (double *)plane.release();
    h->d_vox_off = (int64_t *)vox_off.release();
    h->d_pose_idx = (int32_t *)pose_idx.release();
    h->d_vox_label = (int32_t *)vox_label.release();
    h->d_clusters = (double *)clusters.release();
    return LVBA_OK;
}

} // namespace

extern "C" int32_t lvba_voxmap_build(int32_t device, int32_t n_frames, const void *const *frame_points,
                                     const int64_t *frame_count, int32_t point_stride_bytes, const double *poses,
                                     const lvba_voxel_opts *opts, lvba_voxmap_t *out)
{
    if (!out) return lvba_fail(LVBA_ERR_ARG, "out is null");
    *out = nullptr;
    if (n_frames < 1 || !frame_points || !frame_count || !poses)
        return lvba_fail(LVBA_ERR_ARG, "n_frames < 1 or a null argument");
    if (point_stride_bytes < 12 || point_stride_bytes % 4)
        return lvba_fail(LVBA_ERR_ARG, "point_stride_bytes must be a multiple of 4 and >= 12 (got %d)", point_stride_bytes);
    lvba_voxel_opts o;
    lvba_voxel_default_opts(&o);
    if (opts) o = *opts;
    if (!(o.voxel_size > 0.0) || o.min_points < 1) return lvba_fail(LVBA_ERR_ARG, "voxel_size must be > 0 and min_points >= 1");
    if (o.layer_limit != 2) return lvba_fail(LVBA_ERR_UNSUPPORTED, "layer_limit is fixed at 2 (bavoxel.hpp:13), got %d", o.layer_limit);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return lvba_fail(LVBA_ERR_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return lvba_fail(LVBA_ERR_ARG, "device %d out of range [0,%d)", device, ndev);
    HIPCHK(hipSetDevice(device));
    lvba_voxmap_s *h = new (std::nothrow) lvba_voxmap_s();
    if (!h) return lvba_fail(LVBA_ERR_NOMEM, "host allocation failed");
    h->device = device;
    h->n_frames = n_frames;
    h->opts = o;
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
        delete h;
        return lvba_fail(LVBA_ERR_DEVICE, "hipStreamCreate failed");
    }
    const int32_t rc = voxmap_build_impl(h, frame_points, frame_count, point_stride_bytes, poses);
    if (rc != LVBA_OK) { lvba_voxmap_destroy(h); return rc; }
    *out = h;
    return LVBA_OK;
}

extern "C" int32_t lvba_voxmap_info(lvba_voxmap_t h, lvba_voxmap_info_t *info)
{
    if (!h || !info) return lvba_fail(LVBA_ERR_ARG, "null argument");
    *info = h->info;
    return LVBA_OK;
}

extern "C" int32_t lvba_voxmap_export(lvba_voxmap_t h, int64_t *voxel_off, int32_t *pose_idx, double *clusters,
                                      int64_t *voxel_key)
{
    if (!h) return lvba_fail(LVBA_ERR_ARG, "null handle");
    HIPCHK(hipSetDevice(h->device));
    const int64_t V = h->info.n_voxels, F = h->info.n_factors;
    if (voxel_off) {
        if (V > 0) HIPCHK(hipMemcpy(voxel_off, h->d_vox_off, 8 * (V + 1), hipMemcpyDeviceToHost));
        else voxel_off[0] = 0;
    }
    if (pose_idx && F > 0) HIPCHK(hipMemcpy(pose_idx, h->d_pose_idx, 4 * F, hipMemcpyDeviceToHost));
    if (clusters && F > 0) HIPCHK(hipMemcpy(clusters, h->d_clusters, 80 * F, hipMemcpyDeviceToHost));
    if (voxel_key && V > 0) {
        std::vector<int32_t> label(2 * V);
        std::vector<uint64_t> rk(h->info.n_roots);
        HIPCHK(hipMemcpy(label.data(), h->d_vox_label, 8 * V, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(rk.data(), h->d_root_key, 8 * h->info.n_roots, hipMemcpyDeviceToHost));
        for (int64_t v = 0; v < V; ++v) {
            const uint64_t k = rk[label[2 * v]];
            voxel_key[4 * v + 0] = (int64_t)(k >> 42) - KEY_BIAS;
            voxel_key[4 * v + 1] = (int64_t)((k >> 21) & 0x1fffff) - KEY_BIAS;
            voxel_key[4 * v + 2] = (int64_t)(k & 0x1fffff) - KEY_BIAS;
            voxel_key[4 * v + 3] = label[2 * v + 1];
        }
    }
    return LVBA_OK;
}

extern "C" int32_t lvba_voxmap_to_balm(lvba_voxmap_t h, lvba_balm_t *out)
{
    if (!h || !out) return lvba_fail(LVBA_ERR_ARG, "null argument");
    *out = nullptr;
    const int64_t V = h->info.n_voxels, F = h->info.n_factors;
    if (V == 0) return lvba_fail(LVBA_ERR_ARG, "the map holds no admitted plane voxel (nothing to optimise)");
    std::vector<int64_t> off(V + 1);
    std::vector<int32_t> idx(F);
    std::vector<double> cl(10 * F);
    TRY(lvba_voxmap_export(h, off.data(), idx.data(), cl.data(), nullptr));
    return lvba_balm_create(h->n_frames, V, off.data(), idx.data(), cl.data(), h->device, out);
}

extern "C" int32_t lvba_voxmap_find_planes(lvba_voxmap_t h, int64_t n, const double *X, double *plane, uint8_t *valid)
{
    if (!h || n < 0 || (n > 0 && (!X || !plane || !valid))) return lvba_fail(LVBA_ERR_ARG, "null argument");
    if (n == 0) return LVBA_OK;
    if (h->info.n_roots == 0) {
        memset(plane, 0, 32 * (size_t)n);
        memset(valid, 0, (size_t)n);
        return LVBA_OK;
    }
    HIPCHK(hipSetDevice(h->device));
    DevBuf dX, dpl, dval;
    HIPCHK(dX.alloc(24 * (size_t)n)); HIPCHK(dpl.alloc(32 * (size_t)n)); HIPCHK(dval.alloc((size_t)n));
    HIPCHK(hipMemcpyAsync(dX.p, X, 24 * (size_t)n, hipMemcpyHostToDevice, h->stream));
    vox_lookup_kernel<<<grid_for(n, 256), 256, 0, h->stream>>>(n, dX.as<double>(), h->opts.voxel_size, h->info.n_roots,
                                                               h->d_root_key, h->d_mask, h->d_rootinfo, h->d_plane_first,
                                                               h->d_plane, dpl.as<double>(), dval.as<uint8_t>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(plane, dpl.p, 32 * (size_t)n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(valid, dval.p, (size_t)n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return LVBA_OK;
}
