// voxelize.hip -- the voxel front-end that BUILDS the LiDAR-BA problem, on the device.
//
// Replaces the per-point hash-map insertion and recursive octree of the reference
//   cut_voxel                           include/BALM/bavoxel.hpp:799-836
//   OCTO_TREE_NODE::recut / cut_func    include/BALM/bavoxel.hpp:391-464 (+ judge_eigen :335-352)
//   OCTO_TREE_NODE::tras_opt            include/BALM/bavoxel.hpp:466-474 -> VOX_HESS::push_voxel :45-54
//   findCorrespondPoint + plane lookup  include/BALM/bavoxel.hpp:320-333, src/lvba_system.cpp:1531-1565
// by a sort-based formulation (no pointers, no hash map):
//   1. one lane per point: world transform, root voxel key (with the reference's fp32 quotient / "-1 if negative" /
//      truncation) and the two octant codes the point WOULD take if its root and then its child were split (the fp32
//      centre arithmetic of cut_func is closed-form given the key, so no tree has to exist yet); the point becomes a
//      16-byte record (x, y, z, frame << 6 | o1 << 3 | o2);
//   2. ONE stable radix sort (rocPRIM) by root key; records are gathered into that order, so every root's points are
//      contiguous, frame-major and in cloud order inside a frame; heads + scans give the root table and the table of
//      (root, frame) segments;
//   3. root layer: one lane per segment sums its records (PointCluster::push in cloud order -- products of fp32 values
//      are exact in fp64 and the order of the sums is the reference's, so the cluster is bit-identical); one lane per
//      root merges its segments in the world frame (judge_eigen's covMat) and decides PLANE / split / drop;
//   4. only for roots that split: one wave per segment, lane t = grandchild t = (o1, o2): every lane sees every record
//      (coalesced 16-byte loads, v_readlane broadcast) and advances its child's and grandchild's cluster in cloud
//      order; one wave per split root then merges and decides the 8 + 64 nodes (min_ps, eigen ratio, layer limit);
//   5. ballots / scans fix the emission order (root key, then octant path); the admitted nodes' clusters are copied
//      out as the CSR that lvba_balm_create takes.
// The octree states stay on the device as two words per root for the landmark -> plane lookup of the visual stage.
#include <cstring>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <new>
#include <chrono>
#include <thread>
#include <algorithm>
#include "host_arena.h"
#include "lvba_common.h"
#include "balm_math.h"
#include "mempool.h"
#include "lvba_internal.h"
#include "voxel_internal.h"

using namespace lvba;

namespace {

enum : int { ST_NONE = 0, ST_DROP = 1, ST_PLANE = 2, ST_SPLIT = 3 };

// ---- shared per-point arithmetic (cut_voxel :809-815, root centre :826-829, cut_func :368-381) ------------------
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

// ---- 1. keys + records --------------------------------------------------------------------------------------------
// frame_off: the window's slice of the scan set's frame offsets; pts already points at the window's first point
__global__ void vox_key_kernel(int64_t P, const float *__restrict__ pts, const int64_t *__restrict__ frame_off,
                               int n_frames, const double *__restrict__ poses, double vs,
                               uint64_t *__restrict__ key, float4 *__restrict__ rec, uint32_t *__restrict__ idx,
                               int *__restrict__ err, int *__restrict__ range_partial /* voxel_internal.h: key_range_update */)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int kb[3] = {0, 0, 0};
    if (i < P) {
    const int64_t base = frame_off[0];
    int lo = 0, hi = n_frames; // frame f with frame_off[f] <= base + i < frame_off[f+1]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (frame_off[mid] - base <= i) lo = mid; else hi = mid;
    }
    const double *T = poses + 12 * (int64_t)lo;
    const float fx = pts[3 * i], fy = pts[3 * i + 1], fz = pts[3 * i + 2];
    const double p0 = fx, p1 = fy, p2 = fz;
    double pw[3];
    pw[0] = T[0] * p0 + T[1] * p1 + T[2] * p2 + T[9];
    pw[1] = T[3] * p0 + T[4] * p1 + T[5] * p2 + T[10];
    pw[2] = T[6] * p0 + T[7] * p1 + T[8] * p2 + T[11];
    int64_t k[3];
    if (!root_key_of(pw, vs, k)) { *err = 1; k[0] = k[1] = k[2] = 0; }
    int o1, o2;
    octants_of(pw, k, vs, o1, o2);
    key[i] = pack_key(k);
    rec[i] = make_float4(fx, fy, fz, __int_as_float((int)(((uint32_t)lo << 6) | (uint32_t)(o1 << 3) | (uint32_t)o2)));
    idx[i] = (uint32_t)i;
#pragma unroll
    for (int j = 0; j < 3; ++j) kb[j] = (int)(k[j] + KEY_BIAS);
    }
    key_range_update(range_partial, kb, i < P); // (the sort below runs on the bits that vary)
}
// records into sorted order, the sorted keys back in their 3 x 21-bit form, and the head flags of the sorted sequence in the same
// pass: heads[i] = (first record of a root) << 32 | (first record of a (root, frame) segment) -- one 64-bit word, so that ONE
// inclusive scan numbers roots and segments together.  The frame of a record is found from its index in the window's points (a
// search over <= 2^25 frame offsets that sit in the scalar cache): the flags need no second gather.
// (ws > 0, joint map of several windows: a root is (window, key), window = frame / ws -- the window index sits above the key bits
// of the compressed key, so "the compressed key changed" covers it)
__device__ __forceinline__ int frame_of_point(const int64_t *__restrict__ frame_off, int n_frames, int64_t i)
{
    const int64_t base = frame_off[0];
    int lo = 0, hi = n_frames; // frame f with frame_off[f] <= base + i < frame_off[f+1]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (frame_off[mid] - base <= i) lo = mid; else hi = mid;
    }
    return lo;
}
template <class K>
__global__ void vox_gather_kernel(int64_t n, const float4 *__restrict__ rec, const uint32_t *__restrict__ order,
                                  float4 *__restrict__ out, const K *__restrict__ ckey_s, const KeyPack kp, uint64_t *__restrict__ key_s,
                                  const int64_t *__restrict__ frame_off, int n_frames, uint64_t *__restrict__ heads)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t oi = order[i];
    out[i] = rec[oi];
    const K c = ckey_s[i];
    bool hr = i == 0, hs = i == 0;
    if (i > 0) {
        hr = c != ckey_s[i - 1]; // (key_s is a buffer of its own: the neighbour's compressed key is still there)
        hs = hr || frame_of_point(frame_off, n_frames, oi) != frame_of_point(frame_off, n_frames, order[i - 1]);
    }
    heads[i] = ((uint64_t)(hr ? 1u : 0u) << 32) | (uint64_t)(hs ? 1u : 0u);
    K ck = c;
    if (kp.total < (int)(8 * sizeof(K))) ck &= (K)(((K)1 << kp.total) - 1); // (a joint map's window index sits above the key bits)
    key_s[i] = key_expand<K>(ck, kp);
}
// joint map: (window << total) | re-packed key -- the records of a window stay together, inside it the order is the key's
template <class K>
__global__ void key_compress_win_kernel(int64_t n, const uint64_t *key, const float4 *__restrict__ rec, const KeyPack kp, int ws,
                                        K *out /* may be key */)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const K win = (K)((__float_as_int(rec[i].w) >> 6) / ws);
    out[i] = (K)(win << kp.total) | key_compress<K>(key[i], kp);
}
// first admitted voxel / factor of every window that has roots (the others are filled in on the host)
__global__ void vox_win_first_kernel(int64_t R, const uint32_t *__restrict__ root_seg, const int32_t *__restrict__ seg_frame, int ws,
                                     const int32_t *__restrict__ vox_first, const int64_t *__restrict__ fac_first,
                                     int64_t *__restrict__ win_v0, int64_t *__restrict__ win_f0)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const int w = seg_frame[root_seg[r]] / ws;
    if (r == 0 || seg_frame[root_seg[r - 1]] / ws != w) { win_v0[w] = vox_first[r]; win_f0[w] = fac_first[r]; }
}
// The build sizes its next buffers and launches from counts the device has just formed.  Every such read used to be a
// synchronous copy of its own (sixteen blocking calls per map); now one single-workgroup kernel writes the words of up to six
// device locations into a slot of the process-wide pinned stage (mempool.h) and the host waits for the stream ONCE per decision
// point.  No slot to be had: the plain copies.
struct FetchArgs {
    const uint32_t *src[6];
    int words[6], n;
    uint32_t *dst;
};
__global__ void vox_fetch_kernel(FetchArgs a)
{
    int base = 0;
    for (int k = 0; k < a.n; ++k) {
        for (int i = threadIdx.x; i < a.words[k]; i += blockDim.x) a.dst[base + i] = a.src[k][i];
        base += a.words[k];
    }
}
struct Fetch {
    FetchArgs a{};
    void *host[6];
    void add(void *h, const void *d, size_t bytes) { host[a.n] = h; a.src[a.n] = (const uint32_t *)d; a.words[a.n] = (int)(bytes / 4); ++a.n; }
    hipError_t run(hipStream_t s)
    {
        size_t total = 0;
        for (int k = 0; k < a.n; ++k) total += 4 * (size_t)a.words[k];
        void *slot = lvba::HostStage::get().lock(total);
        if (!slot) {
            hipError_t e = hipStreamSynchronize(s);
            for (int k = 0; k < a.n && e == hipSuccess; ++k) e = lvba::copy_d2h(host[k], a.src[k], 4 * (size_t)a.words[k]);
            return e;
        }
        a.dst = (uint32_t *)slot;
        vox_fetch_kernel<<<1, 256, 0, s>>>(a);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        const char *q = (const char *)slot;
        for (int k = 0; k < a.n && e == hipSuccess; ++k) { memcpy(host[k], q, 4 * (size_t)a.words[k]); q += 4 * (size_t)a.words[k]; }
        lvba::HostStage::get().unlock(slot);
        return e;
    }
};
__global__ void vox_close_offsets_kernel(int64_t *__restrict__ vox_off_end, const int64_t *__restrict__ fac_total) { *vox_off_end = *fac_total; }
__global__ void vox_pose_rel_kernel(int64_t F, int32_t *__restrict__ pose_idx, int ws)
{
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f < F) pose_idx[f] %= ws;
}
// root table and (root, frame) segment table from the packed head flags (vox_gather_kernel) and their inclusive scan
__global__ void vox_tables_kernel(int64_t n, const uint64_t *__restrict__ key, const float4 *__restrict__ rec,
                                  const uint64_t *__restrict__ heads, const uint64_t *__restrict__ incl,
                                  uint64_t *__restrict__ root_key, uint32_t *__restrict__ root_seg,
                                  uint32_t *__restrict__ seg_start, uint32_t *__restrict__ seg_root,
                                  int32_t *__restrict__ seg_frame)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t hd = heads[i], in = incl[i];
    const uint32_t in_root = (uint32_t)(in >> 32), in_seg = (uint32_t)in;
    if (hd & 1u) {
        const uint32_t sg = in_seg - 1;
        seg_start[sg] = (uint32_t)i;
        seg_root[sg] = in_root - 1;
        seg_frame[sg] = __float_as_int(rec[i].w) >> 6;
        if (hd >> 32) {
            root_key[in_root - 1] = key[i];
            root_seg[in_root - 1] = sg;
        }
    }
    if (i == n - 1) {
        seg_start[in_seg] = (uint32_t)n;
        root_seg[in_root] = in_seg;
    }
}

// ---- 3. octree -------------------------------------------------------------------------------------------------
struct Acc { // one PointCluster in the making (tools.hpp:407-433)
    double c0, c1, c2, c3, c4, c5, v0, v1, v2;
    int n;
    __device__ __forceinline__ void clear() { c0 = c1 = c2 = c3 = c4 = c5 = v0 = v1 = v2 = 0.0; n = 0; }
    __device__ __forceinline__ void push(double x, double y, double z)
    {
        c0 += x * x; c1 += x * y; c2 += x * z; c3 += y * y; c4 += y * z; c5 += z * z;
        v0 += x; v1 += y; v2 += z;
        ++n;
    }
    __device__ __forceinline__ void store(double *o) const
    {
        o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3; o[4] = c4; o[5] = c5; o[6] = v0; o[7] = v1; o[8] = v2; o[9] = (double)n;
    }
};
// covMat += tmp.transform(sig_orig[f], x_buf[f])  (judge_eigen :337-344)
__device__ __forceinline__ void merge_world(const double *c, const double *T, double *S)
{
    double W[10];
    transform_cluster(c, T, T + 9, W);
#pragma unroll
    for (int k = 0; k < 10; ++k) S[k] += W[k];
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
    plane[0] = vb[0]; plane[1] = vb[1]; plane[2] = vb[2];
    plane[3] = U[0]; plane[4] = U[3]; plane[5] = U[6];
    if (lam[0] / lam[2] > (double)ratio) return last_layer ? ST_DROP : ST_SPLIT;
    return ST_PLANE;
}
__device__ __forceinline__ float lane_bcast(float v, int j)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j));
}

// 3a. root-level PointCluster of every (root, frame) segment, records in cloud order.  The sum is a serial chain by
// definition (that is what makes it bit-identical to PointCluster::push), so the work is arranged to keep the chain as
// short as the hardware allows:
//   segments below SEG_BIG records: one lane per segment (a wave's loop is bounded by SEG_BIG iterations);
//   larger segments: one wave per segment -- the 64 lanes form the nine terms (xx, xy, xz, yy, yz, zz, x, y, z) of 64
//   records in parallel and park them in LDS, then lanes 0..8 each run ONE of the nine chains over the 64 parked
//   values (an LDS read and one dependent v_add_f64 per record instead of ~25 instructions).
// Both also note which children / grandchildren the segment touches (m1: 8 bits by o1, m2: 64 bits by code) for the
// roots that will split.
constexpr uint32_t SEG_BIG = 48;
__global__ void vox_seg_small_kernel(int64_t NS, const uint32_t *__restrict__ seg_start, const float4 *__restrict__ rec,
                                     double *__restrict__ segcl, uint32_t *__restrict__ segm1, uint64_t *__restrict__ segm2)
{
    const int64_t sg = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (sg >= NS) return;
    const uint32_t b = seg_start[sg], e = seg_start[sg + 1];
    if (e - b >= SEG_BIG) return;
    Acc A;
    A.clear();
    uint32_t a1 = 0;
    uint64_t a2 = 0;
    for (uint32_t i = b; i < e; ++i) {
        const float4 q = rec[i];
        A.push((double)q.x, (double)q.y, (double)q.z);
        const int c = __float_as_int(q.w) & 63;
        a1 |= 1u << (c >> 3);
        a2 |= 1ull << c;
    }
    A.store(segcl + 10 * sg);
    segm1[sg] = a1;
    segm2[sg] = a2;
}
__global__ __launch_bounds__(64) void vox_seg_big_kernel(int64_t NS, const uint32_t *__restrict__ seg_start, const float4 *__restrict__ rec,
                                                         double *__restrict__ segcl, uint32_t *__restrict__ segm1,
                                                         uint64_t *__restrict__ segm2)
{
    // (one block per segment, most of which return at once.  Measured in round 5 and withdrawn: a few thousand persistent blocks
    // walking the segment table -- the octree phase 0.91 -> 1.27 ms: the big segments are serial chains, and a block that walks
    // fifty of them one after the other is what the launch then waits for --, and four segments per block, one wavefront each:
    // 0.91 -> 1.01 ms.)
    const int64_t sg = blockIdx.x;
    if (sg >= NS) return;
    const uint32_t b = seg_start[sg], e = seg_start[sg + 1];
    if (e - b < SEG_BIG) return;
    __shared__ double terms[64 * 9];
    __shared__ unsigned long long s_m2;
    __shared__ unsigned s_m1;
    const int t = threadIdx.x;
    if (t == 0) { s_m1 = 0u; s_m2 = 0ull; }
    double acc = 0.0;
    uint32_t a1 = 0;
    uint64_t a2 = 0;
    for (uint32_t base = b; base < e; base += 64) {
        const int cnt = (int)min(64u, e - base);
        if (t < cnt) {
            const float4 q = rec[base + t];
            const double x = q.x, y = q.y, z = q.z;
            double *o = terms + 9 * t;
            o[0] = x * x; o[1] = x * y; o[2] = x * z; o[3] = y * y; o[4] = y * z; o[5] = z * z; o[6] = x; o[7] = y; o[8] = z;
            const int c = __float_as_int(q.w) & 63;
            a1 |= 1u << (c >> 3);
            a2 |= 1ull << c;
        }
        __syncthreads();
        if (t < 9)
            for (int j = 0; j < cnt; ++j) acc += terms[9 * j + t];
        __syncthreads();
    }
    if (a1) atomicOr(&s_m1, a1);
    if (a2) atomicOr(&s_m2, (unsigned long long)a2);
    __syncthreads();
    if (t < 9) segcl[10 * sg + t] = acc;
    if (t == 9) {
        segcl[10 * sg + 9] = (double)(e - b);
        segm1[sg] = s_m1;
        segm2[sg] = s_m2;
    }
}

struct RootArgs {
    int64_t R;
    const uint32_t *root_seg; // [R+1]
    const int32_t *seg_frame;
    const double *segcl;
    const double *poses;
    float ratio[3];
    int min_ps;
    // per-root results shared by all later stages
    int32_t *n_plane, *n_vox;
    int64_t *n_fac;
    uint64_t *mask;      // lanes that hold a PLANE node
    uint32_t *rootinfo;  // state0 | split1 << 8
    double *plane0;      // [R][6] plane of the root node
    uint32_t *is_split;  // [R+1] 1 if the root splits (scanned into the split list)
    uint32_t *seg_split; // [NS+1] 1 for every segment of a splitting root
};
// 3b. root decision (recut at layer 0): one lane per root, its segments merged in frame order.
__global__ void vox_root_kernel(RootArgs a)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.R) return;
    const uint32_t s0 = a.root_seg[r], s1 = a.root_seg[r + 1];
    double S[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) S[k] = 0.0;
    for (uint32_t sg = s0; sg < s1; ++sg) merge_world(a.segcl + 10 * (int64_t)sg, a.poses + 12 * (int64_t)a.seg_frame[sg], S);
    double pl[6] = {0, 0, 0, 0, 0, 0};
    const int nf = (int)(s1 - s0);
    const int st = node_decide(S, nf, a.min_ps, a.ratio[0], false, pl);
    const bool plane = st == ST_PLANE, adm = plane && nf >= 2, split = st == ST_SPLIT;
    a.n_plane[r] = plane ? 1 : 0;
    a.n_vox[r] = adm ? 1 : 0;
    a.n_fac[r] = adm ? nf : 0;
    a.mask[r] = plane ? 1ull : 0ull;
    a.rootinfo[r] = (uint32_t)st;
#pragma unroll
    for (int k = 0; k < 6; ++k) a.plane0[6 * r + k] = pl[k];
    a.is_split[r] = split ? 1u : 0u;
    for (uint32_t sg = s0; sg < s1; ++sg) a.seg_split[sg] = split ? 1u : 0u;
}

// compaction of the split roots / their segments (flag + exclusive scan -> list)
__global__ void vox_compact_kernel(int64_t n, const uint32_t *__restrict__ flag, const uint32_t *__restrict__ excl,
                                   uint32_t *__restrict__ list)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && flag[i]) list[excl[i]] = (uint32_t)i;
}

// 3c. masks and node counts of the split segments, in split-list order
__global__ void vox_split_masks_kernel(int64_t NSS, const uint32_t *__restrict__ split_segs,
                                       const uint32_t *__restrict__ segm1, const uint64_t *__restrict__ segm2,
                                       uint32_t *__restrict__ m1, uint64_t *__restrict__ m2, uint32_t *__restrict__ cnt)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= NSS) { if (k == NSS) cnt[k] = 0; return; }
    const uint32_t sg = split_segs[k];
    const uint32_t a1 = segm1[sg];
    const uint64_t a2 = segm2[sg];
    m1[k] = a1;
    m2[k] = a2;
    cnt[k] = (uint32_t)(__popc(a1) + __popcll(a2));
}

// 3d. child and grandchild PointClusters of a split segment: one wave per segment, lane t = grandchild t = (o1, o2);
// every lane sees every record (v_readlane broadcast) and advances its two accumulators in cloud order.
// Stored at nodecl[base ..]: the children present (ascending o1), then the grandchildren present (ascending code).
__global__ __launch_bounds__(64) void vox_split_cluster_kernel(const uint32_t *__restrict__ split_segs,
                                                               const uint32_t *__restrict__ seg_start,
                                                               const float4 *__restrict__ rec,
                                                               const uint32_t *__restrict__ m1v,
                                                               const uint64_t *__restrict__ m2v,
                                                               const uint32_t *__restrict__ basev,
                                                               double *__restrict__ nodecl)
{
    const int64_t k = blockIdx.x;
    const int t = threadIdx.x, o1 = t >> 3, o2 = t & 7;
    const uint32_t sg = split_segs[k];
    const uint32_t b = seg_start[sg], e = seg_start[sg + 1];
    Acc A1, A2;
    A1.clear(); A2.clear();
    for (uint32_t base = b; base < e; base += 64) {
        const int cnt = (int)min(64u, e - base);
        const float4 q = t < cnt ? rec[base + t] : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < cnt; ++j) {
            const int c = __builtin_amdgcn_readlane(__float_as_int(q.w), j) & 63;
            const double x = lane_bcast(q.x, j), y = lane_bcast(q.y, j), z = lane_bcast(q.z, j);
            if ((c >> 3) == o1) A1.push(x, y, z);
            if (c == t) A2.push(x, y, z);
        }
    }
    const uint32_t m1 = m1v[k];
    const uint64_t m2 = m2v[k];
    const int64_t base = basev[k];
    if (o2 == 0 && ((m1 >> o1) & 1u)) A1.store(nodecl + 10 * (base + __popc(m1 & ((1u << o1) - 1u))));
    if ((m2 >> t) & 1ull) A2.store(nodecl + 10 * (base + __popc(m1) + __popcll(t ? (m2 & (~0ull >> (64 - t))) : 0ull)));
}

struct SplitArgs {
    const uint32_t *split_roots; // [NRS]
    const uint32_t *root_seg;    // [R+1]
    const uint32_t *seg_excl;    // [NS+1] exclusive scan of seg_split: position of a segment in the split list
    const int32_t *seg_frame;
    const uint32_t *m1;
    const uint64_t *m2;
    const uint32_t *base;
    const double *nodecl;
    const double *poses;
    float ratio[3];
    int min_ps;
    int32_t *n_plane, *n_vox;
    int64_t *n_fac;
    uint64_t *mask;
    uint32_t *rootinfo;
    unsigned *tmp_count; // PLANE nodes of split roots parked so far
    int32_t *tmp_base;   // [NRS] first parked slot of the split root
    double *tmp_plane;   // [][6]
    int32_t *tmp_nf;     // [] observing frames
    // emission (after the scans)
    const int32_t *plane_first, *vox_first;
    const int64_t *fac_first;
    double *plane;
    int64_t *vox_off;
    int32_t *pose_idx;
    double *clusters;
    int32_t *vox_label;
};
// 3e. children / grandchildren of a split root (recut at layers 1 and 2): one wave per split root, lane t = grandchild t.
__global__ __launch_bounds__(64) void vox_split_decide_kernel(SplitArgs a)
{
    const int64_t k = blockIdx.x;
    const int64_t r = a.split_roots[k];
    const int t = threadIdx.x, o1 = t >> 3, o2 = t & 7;
    const uint64_t lt = t == 0 ? 0ull : (~0ull >> (64 - t));
    __shared__ int s_nf[64];
    __shared__ int s_base;
    double S1[10], S2[10], pl[6], pl1[6];
#pragma unroll
    for (int q = 0; q < 10; ++q) S1[q] = S2[q] = 0.0;
    int nf1 = 0, nf2 = 0;
    for (uint32_t sg = a.root_seg[r], s1 = a.root_seg[r + 1]; sg < s1; ++sg) {
        const uint32_t j = a.seg_excl[sg];
        const uint32_t m1 = a.m1[j];
        const uint64_t m2 = a.m2[j];
        const int64_t base = a.base[j];
        const double *T = a.poses + 12 * (int64_t)a.seg_frame[sg];
        if ((m1 >> o1) & 1u) { merge_world(a.nodecl + 10 * (base + __popc(m1 & ((1u << o1) - 1u))), T, S1); ++nf1; }
        if ((m2 >> t) & 1ull) { merge_world(a.nodecl + 10 * (base + __popc(m1) + __popcll(m2 & lt)), T, S2); ++nf2; }
    }
    int level = -1, nf = 0;
    const int st1 = node_decide(S1, nf1, a.min_ps, a.ratio[1], false, pl1);
    if (st1 == ST_PLANE) {
        if (o2 == 0) {
            level = 1; nf = nf1;
#pragma unroll
            for (int q = 0; q < 6; ++q) pl[q] = pl1[q];
        }
    } else if (st1 == ST_SPLIT) {
        if (node_decide(S2, nf2, a.min_ps, a.ratio[2], true, pl) == ST_PLANE) { level = 2; nf = nf2; }
    }
    // lanes in path order: child o1 -> lane 8*o1, grandchild -> its own lane
    const bool is_plane = level >= 0;
    const bool admitted = is_plane && nf >= 2;
    const uint64_t pmask = __ballot(is_plane);
    const uint64_t amask = __ballot(admitted);
    const uint64_t smask = __ballot(st1 == ST_SPLIT && o2 == 0); // bit 8*o1
    s_nf[t] = admitted ? nf : 0;
    if (t == 0) s_base = pmask ? (int)atomicAdd(a.tmp_count, (unsigned)__popcll(pmask)) : 0;
    __syncthreads();
    if (is_plane) { // parked until the scans have fixed the final order
        const int64_t slot = (int64_t)s_base + __popcll(pmask & lt);
#pragma unroll
        for (int q = 0; q < 6; ++q) a.tmp_plane[6 * slot + q] = pl[q];
        a.tmp_nf[slot] = nf;
    }
    if (t == 0) {
        int64_t fsum = 0;
        for (int q = 0; q < 64; ++q) fsum += s_nf[q];
        uint32_t split1 = 0;
        for (int q = 0; q < 8; ++q) split1 |= (uint32_t)((smask >> (8 * q)) & 1ull) << q;
        a.n_plane[r] = __popcll(pmask);
        a.n_vox[r] = __popcll(amask);
        a.n_fac[r] = fsum;
        a.mask[r] = pmask;
        a.rootinfo[r] = (uint32_t)ST_SPLIT | (split1 << 8);
        a.tmp_base[k] = s_base;
    }
}
// 3f. emission for a split root (tras_opt :466-474 -> push_voxel :45-54): planes into their final slots, admitted
// nodes' per-frame clusters copied in frame order.
__global__ __launch_bounds__(64) void vox_split_emit_kernel(SplitArgs a)
{
    const int64_t k = blockIdx.x;
    const int64_t r = a.split_roots[k];
    const uint64_t pmask = a.mask[r];
    if (pmask == 0ull) return;
    const int t = threadIdx.x, o1 = t >> 3, o2 = t & 7;
    __shared__ int s_nf[64];
    const uint32_t info = a.rootinfo[r];
    const uint64_t lt = t == 0 ? 0ull : (~0ull >> (64 - t));
    const bool is_plane = (pmask >> t) & 1ull;
    const int level = ((info >> (8 + o1)) & 1u) ? 2 : 1;
    int nf = 0;
    if (is_plane) {
        const int64_t slot = (int64_t)a.tmp_base[k] + __popcll(pmask & lt);
        nf = a.tmp_nf[slot];
        double *o = a.plane + 6 * ((int64_t)a.plane_first[r] + __popcll(pmask & lt));
#pragma unroll
        for (int q = 0; q < 6; ++q) o[q] = a.tmp_plane[6 * slot + q];
    }
    const bool admitted = is_plane && nf >= 2;
    const uint64_t amask = __ballot(admitted);
    if (amask == 0ull) return;
    s_nf[t] = admitted ? nf : 0;
    __syncthreads();
    if (!admitted) return;
    int64_t foff = a.fac_first[r];
    for (int q = 0; q < t; ++q) foff += s_nf[q];
    const int64_t v = (int64_t)a.vox_first[r] + __popcll(amask & lt);
    a.vox_off[v] = foff;
    a.vox_label[2 * v] = (int32_t)r;
    a.vox_label[2 * v + 1] = level | (o1 << 4) | ((level == 2 ? o2 : 0) << 8);
    for (uint32_t sg = a.root_seg[r], s1 = a.root_seg[r + 1]; sg < s1; ++sg) {
        const uint32_t j = a.seg_excl[sg];
        const uint32_t m1 = a.m1[j];
        const uint64_t m2 = a.m2[j];
        int64_t src = -1;
        if (level == 1) { if ((m1 >> o1) & 1u) src = (int64_t)a.base[j] + __popc(m1 & ((1u << o1) - 1u)); }
        else if ((m2 >> t) & 1ull) src = (int64_t)a.base[j] + __popc(m1) + __popcll(m2 & lt);
        if (src < 0) continue;
        a.pose_idx[foff] = a.seg_frame[sg];
#pragma unroll
        for (int q = 0; q < 10; ++q) a.clusters[10 * foff + q] = a.nodecl[10 * src + q];
        ++foff;
    }
}
// 3g. emission for roots that are planes themselves: one lane per segment.
__global__ void vox_root_emit_kernel(int64_t NS, const uint32_t *__restrict__ seg_root, const uint32_t *__restrict__ root_seg,
                                     const int32_t *__restrict__ seg_frame, const double *__restrict__ segcl,
                                     const uint32_t *__restrict__ rootinfo, const int32_t *__restrict__ n_vox,
                                     const double *__restrict__ plane0, const int32_t *__restrict__ plane_first,
                                     const int32_t *__restrict__ vox_first, const int64_t *__restrict__ fac_first,
                                     double *__restrict__ plane, int64_t *__restrict__ vox_off, int32_t *__restrict__ pose_idx,
                                     double *__restrict__ clusters, int32_t *__restrict__ vox_label)
{
    const int64_t sg = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (sg >= NS) return;
    const uint32_t r = seg_root[sg];
    if ((rootinfo[r] & 0xff) != ST_PLANE) return;
    const uint32_t first = root_seg[r];
    if (sg == first) {
#pragma unroll
        for (int q = 0; q < 6; ++q) plane[6 * (int64_t)plane_first[r] + q] = plane0[6 * (int64_t)r + q];
    }
    if (!n_vox[r]) return;
    const int64_t foff = fac_first[r] + (sg - first);
    if (sg == first) {
        const int64_t v = vox_first[r];
        vox_off[v] = foff;
        vox_label[2 * v] = (int32_t)r;
        vox_label[2 * v + 1] = 0;
    }
    pose_idx[foff] = seg_frame[sg];
#pragma unroll
    for (int q = 0; q < 10; ++q) clusters[10 * foff + q] = segcl[10 * sg + q];
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

} // namespace

struct lvba_voxmap_s {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = true; // false: the caller's stream (the window driver's worker streams; a stream costs ~0.6 ms to destroy)
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
    // A JOINT map of several windows (lvba_voxmap_build_scans_joint: roots are (window, key); the admitted voxels of window w are
    // the range win_v0[w] .. win_v0[w + 1], its factors win_f0[w] .. win_f0[w + 1], pose indices relative to the window's first
    // frame) and the per-window VIEWS into one (lvba_voxmap_window_view: they own nothing).  Neither has a key lookup: the root
    // table of a joint map holds a key once per window.
    int window_size = 0, n_windows = 1;
    lvba::hvec<int64_t> win_v0, win_f0;
    bool is_view = false;
};

const double *lvba_voxmap_clusters(const lvba_voxmap_s *h) { return h ? h->d_clusters : nullptr; }

extern "C" int64_t lvba_release_cached_memory(void)
{
    lvba::StreamCache::get().release_all();
    return (int64_t)DevicePool::get().release() + (int64_t)lvba::HostArena::get().release() + (int64_t)lvba::PinnedCache::get().release_all();
}

extern "C" void lvba_voxel_default_opts(lvba_voxel_opts *o)
{
    if (!o) return;
    o->voxel_size = 1.0;                                   // cut_voxel's default (bavoxel.hpp:801)
    o->eigen_ratio[0] = 0.3f; o->eigen_ratio[1] = 0.1f;    // include/dataset_io.h:77 (stage 1)
    o->eigen_ratio[2] = 0.06f; o->eigen_ratio[3] = 0.03f;
    o->min_points = 15;                                    // bavoxel.hpp:24
    o->layer_limit = 2;                                    // bavoxel.hpp:13
}

extern "C" int32_t lvba_scans_destroy(lvba_scans_t sc)
{
    if (!sc) return LVBA_OK;
    (void)hipSetDevice(sc->device);
    if (sc->d_pts) (void)hipFree(sc->d_pts);
    if (sc->d_frame_off) (void)hipFree(sc->d_frame_off);
    delete sc;
    return LVBA_OK;
}

extern "C" int32_t lvba_scans_create(int32_t device, int32_t n_frames, const void *const *frame_points,
                                     const int64_t *frame_count, int32_t point_stride_bytes, lvba_scans_t *out)
{
    if (!out) return lvba_fail(LVBA_ERR_ARG, "out is null");
    *out = nullptr;
    if (n_frames < 1 || !frame_points || !frame_count) return lvba_fail(LVBA_ERR_ARG, "n_frames < 1 or a null argument");
    if (point_stride_bytes < 12 || point_stride_bytes % 4)
        return lvba_fail(LVBA_ERR_ARG, "point_stride_bytes must be a multiple of 4 and >= 12 (got %d)", point_stride_bytes);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return lvba_fail(LVBA_ERR_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return lvba_fail(LVBA_ERR_ARG, "device %d out of range [0,%d)", device, ndev);
    HIPCHK(hipSetDevice(device));
    lvba_scans_s *sc = new (std::nothrow) lvba_scans_s();
    if (!sc) return lvba_fail(LVBA_ERR_NOMEM, "host allocation failed");
    sc->device = device;
    sc->n_frames = n_frames;
    sc->frame_off.assign(n_frames + 1, 0);
    for (int f = 0; f < n_frames; ++f) {
        if (frame_count[f] < 0 || (frame_count[f] > 0 && !frame_points[f])) {
            delete sc;
            return lvba_fail(LVBA_ERR_ARG, "frame %d: bad point count / null cloud", f);
        }
        sc->frame_off[f + 1] = sc->frame_off[f] + frame_count[f];
    }
    const int64_t P = sc->frame_off[n_frames];
    auto fail = [&](hipError_t e, const char *what) {
        lvba_scans_destroy(sc);
        return lvba_fail(e == hipErrorOutOfMemory ? LVBA_ERR_NOMEM : LVBA_ERR_DEVICE, "%s: %s", what, hipGetErrorString(e));
    };
    hipError_t e;
    if ((e = hipMalloc((void **)&sc->d_pts, P ? 12 * (size_t)P : 8)) != hipSuccess) return fail(e, "hipMalloc(points)");
    if ((e = hipMalloc((void **)&sc->d_frame_off, 8 * ((size_t)n_frames + 1))) != hipSuccess) return fail(e, "hipMalloc(frame_off)");
    // xyz packed out of the caller's point stride (e.g. sizeof(pcl::PointXYZINormal) = 48).  hipMemcpy2D of 12-byte rows out of
    // pageable memory ran at 3.3 GB/s (57.8 ms for 16 M points: 18 x the map build it feeds).  Now the host packs the
    // coordinates itself -- a few threads, each a slice of a chunk of <= 1 M points -- into one of two pinned buffers, and the
    // chunk travels as ONE contiguous asynchronous copy while the next is being packed (hipMemcpy2D remains the fall-back when no
    // pinned memory is to be had).
    bool packed_path = point_stride_bytes != 12 && P > 0;
    if (packed_path) {
        // Every worker thread owns a contiguous run of the points (across frames), two pinned slots of kSub points and a stream:
        // pack a slot, send it, pack the other one while it travels -- no thread waits for another.  (Round 3's form spawned its
        // packing threads anew for every frame-sized chunk and allocated / freed its 24 MB of pinned memory per call: 17-20 ms
        // for 16 M points at a 48-byte stride; this form 9-13 ms with 4-8 threads, 11 with two, 23 with one: the packing is the
        // bound up to two threads, the link after that.)  The pinned block and the streams come from caches (mempool.h).
        const int64_t kSub = (int64_t)1 << 17; // points per slot: 1.5 MB
        unsigned nthr = std::thread::hardware_concurrency();
        nthr = std::max(1u, std::min(nthr ? nthr : 1u, 8u));
        nthr = (unsigned)std::max<int64_t>(1, std::min<int64_t>(nthr, P / 65536));
        const size_t slot_bytes = 12 * (size_t)kSub;
        void *pin = nullptr;
        if (lvba::PinnedCache::get().acquire(&pin, 2 * slot_bytes * nthr) != hipSuccess) {
            (void)hipGetLastError();
            packed_path = false; // no pinned memory to be had: the plain path below
        }
        if (packed_path) {
            std::vector<hipError_t> errs(nthr, hipSuccess);
            const int64_t *foff = sc->frame_off.data();
            auto work = [&](unsigned t) {
                hipError_t &er = errs[t];
                hipStream_t us = nullptr;
                hipEvent_t ev[2] = {nullptr, nullptr};
                if ((er = hipSetDevice(device)) != hipSuccess) return;
                if ((er = lvba::StreamCache::get().acquire(&us)) != hipSuccess) return;
                for (int k = 0; k < 2 && er == hipSuccess; ++k) er = hipEventCreateWithFlags(&ev[k], hipEventDisableTiming);
                const int64_t g0 = P * (int64_t)t / nthr, g1 = P * (int64_t)(t + 1) / nthr;
                int f = (int)(std::upper_bound(foff, foff + n_frames + 1, g0) - foff) - 1;
                f = std::max(0, std::min(f, n_frames - 1));
                bool used[2] = {false, false};
                int k = 0;
                for (int64_t g = g0; g < g1 && er == hipSuccess; g += kSub, k ^= 1) {
                    const int64_t gend = std::min(g + kSub, g1);
                    if (used[k] && (er = hipEventSynchronize(ev[k])) != hipSuccess) break; // its last copy has left the slot
                    float *dst = reinterpret_cast<float *>(static_cast<char *>(pin) + (2 * (size_t)t + (size_t)k) * slot_bytes);
                    for (int64_t i = g; i < gend;) {
                        while (foff[f + 1] <= i) ++f; // (frames without points)
                        const int64_t fe = std::min(gend, foff[f + 1]);
                        const char *src = static_cast<const char *>(frame_points[f]) + (size_t)(i - foff[f]) * (size_t)point_stride_bytes;
                        for (; i < fe; ++i, src += point_stride_bytes, dst += 3) memcpy(dst, src, 12);
                    }
                    er = hipMemcpyAsync(sc->d_pts + 3 * g, static_cast<char *>(pin) + (2 * (size_t)t + (size_t)k) * slot_bytes,
                                        12 * (size_t)(gend - g), hipMemcpyHostToDevice, us);
                    if (er == hipSuccess) er = hipEventRecord(ev[k], us);
                    used[k] = true;
                }
                const hipError_t es = hipStreamSynchronize(us);
                if (er == hipSuccess) er = es;
                for (int q = 0; q < 2; ++q)
                    if (ev[q]) (void)hipEventDestroy(ev[q]);
                lvba::StreamCache::get().release(us);
            };
            {
                std::vector<std::thread> th;
                for (unsigned t = 1; t < nthr; ++t) th.emplace_back(work, t);
                work(0);
                for (auto &q : th) q.join();
            }
            e = hipSuccess;
            for (hipError_t q : errs)
                if (q != hipSuccess) e = q;
            (void)hipSetDevice(device);
        }
        lvba::PinnedCache::get().release(pin);
        if (packed_path && e != hipSuccess) return fail(e, "packed upload (points)");
    }
    if (!packed_path)
        for (int f = 0; f < n_frames; ++f)
            if (frame_count[f] > 0) {
                float *dst = sc->d_pts + 3 * sc->frame_off[f];
                e = point_stride_bytes == 12
                        ? lvba::copy_h2d(dst, frame_points[f], 12 * (size_t)frame_count[f])
                        : hipMemcpy2D(dst, 12, frame_points[f], (size_t)point_stride_bytes, 12, (size_t)frame_count[f],
                                      hipMemcpyHostToDevice);
                if (e != hipSuccess) return fail(e, "hipMemcpy(points)");
            }
    if ((e = lvba::copy_h2d(sc->d_frame_off, sc->frame_off.data(), 8 * ((size_t)n_frames + 1))) != hipSuccess)
        return fail(e, "hipMemcpy(frame_off)");
    *out = sc;
    return LVBA_OK;
}

extern "C" int32_t lvba_voxmap_destroy(lvba_voxmap_t h)
{
    if (!h) return LVBA_OK;
    if (h->is_view) { delete h; return LVBA_OK; } // a window view of a joint map: the arrays are the joint map's
    (void)hipSetDevice(h->device);
    void *ptrs[] = {h->d_root_key, h->d_mask, h->d_rootinfo, h->d_plane_first, h->d_plane,
                    h->d_vox_off, h->d_pose_idx, h->d_vox_label, h->d_clusters};
    for (void *p : ptrs) DevicePool::get().free(p);
    if (h->stream && h->owns_stream) lvba::StreamCache::get().release(h->stream);
    delete h;
    return LVBA_OK;
}

namespace {

// window_size > 0: a JOINT map of the windows [frame_begin + k ws, frame_begin + (k + 1) ws) -- one sort, one pass of every kernel
// for all of them; every window's part is what its own map would be (same records in the same order, root by root).
int32_t voxmap_build_impl(lvba_voxmap_s *h, const lvba_scans_s *sc, int frame_begin, const double *poses, int window_size = 0)
{
    const int nfr = h->n_frames;
    const int ws = window_size > 0 && window_size < nfr ? window_size : 0;
    const int n_win = ws ? (nfr + ws - 1) / ws : 1;
    const int wbits = n_win > 1 ? 32 - __builtin_clz((unsigned)(n_win - 1)) : 0;
    h->window_size = ws; h->n_windows = n_win;
    hipStream_t s = h->stream;
    const int64_t p_begin = sc->frame_off[frame_begin];
    const int64_t P = sc->frame_off[frame_begin + nfr] - p_begin;
    if (P >= (int64_t)1 << 31) return lvba_fail(LVBA_ERR_UNSUPPORTED, "more than 2^31 points in one map (%lld)", (long long)P);
    if (nfr >= (1 << 25)) return lvba_fail(LVBA_ERR_UNSUPPORTED, "more than 2^25 frames in one map");
    h->info.n_points = P;
    if (P == 0) return LVBA_OK;
    const float *pts = sc->d_pts + 3 * p_begin;
    double t0 = now_ms();

    DevBuf d_poses(s);
    HIPCHK(d_poses.alloc(96 * (size_t)nfr));
    // (small transfers to / from the caller's PAGEABLE memory go through the synchronous copy: the asynchronous one pins the
    // pages on the fly, which was measured at up to 24 ms for 30 KB)
    HIPCHK(lvba::copy_h2d(d_poses.p, poses, 96 * (size_t)nfr));

    // -- 1./2. keys + records, root sort (stable: inside a root the records stay frame-major, cloud order inside a frame),
    //          root table and (root, frame) segment table
    int64_t R = 0, NS = 0;
    DevBuf rec_s(s), root_key(s), root_seg(s), seg_start(s), seg_root(s), seg_frame(s);
    {
        DevBuf key(s), rec(s), idx(s), key_s(s), idx0(s), d_err(s);
        HIPCHK(key.alloc(8 * P)); HIPCHK(rec.alloc(16 * P)); HIPCHK(idx.alloc(4 * P));
        HIPCHK(key_s.alloc(8 * P)); HIPCHK(idx0.alloc(4 * P)); HIPCHK(rec_s.alloc(16 * P)); HIPCHK(d_err.alloc(28));
        int h_err[7] = {0}; // [0] error flag, [1..6] range of the biased key components
        DevBuf d_part(s);
        const int64_t n_slots = key_range_slots(P, 256);
        HIPCHK(d_part.alloc(24 * (size_t)n_slots));
        HIPCHK(hipMemsetAsync(d_err.p, 0, 28, s));
        vox_key_kernel<<<grid_for(P, 256), 256, 0, s>>>(P, pts, sc->d_frame_off + frame_begin, nfr, d_poses.as<double>(),
                                                        h->opts.voxel_size, key.as<uint64_t>(), rec.as<float4>(),
                                                        idx.as<uint32_t>(), d_err.as<int>(), d_part.as<int>());
        HIPCHK(hipGetLastError());
        key_range_reduce_kernel<<<key_range_reduce_grid(n_slots), 256, 0, s>>>(n_slots, d_part.as<int>(), d_err.as<int>() + 1);
        HIPCHK(hipGetLastError());
        { Fetch f; f.add(h_err, d_err.p, 28); HIPCHK(f.run(s)); }
        if (h_err[0]) return lvba_fail(LVBA_ERR_ARG, "a point is non-finite or its voxel key exceeds +-2^20 after the pose transform");
        h->info.key_ms = now_ms() - t0; t0 = now_ms();
        const KeyPack kp = key_pack_of(h_err + 1);
        // (the gather also writes the head flags of the sorted sequence: vox_gather_kernel)
        DevBuf heads(s), incl(s);
        HIPCHK(heads.alloc(8 * P)); HIPCHK(incl.alloc(8 * P));
        const int64_t *foff = sc->d_frame_off + frame_begin;
        uint64_t *keys_sorted = key_s.as<uint64_t>(); // the sorted keys in their 3 x 21-bit form
        const unsigned sort_bits = (unsigned)(kp.total + (ws ? wbits : 0));
        if (ws && sort_bits > 64) return lvba_fail(LVBA_ERR_UNSUPPORTED, "joint map: %d key bits + %d window bits", kp.total, wbits);
        DevBuf k32(s), k32s(s); // (live until the block ends: handing them back earlier would cost a stream round trip)
        if (sort_bits <= 32) {
            HIPCHK(k32.alloc(4 * P)); HIPCHK(k32s.alloc(4 * P));
            if (ws) key_compress_win_kernel<uint32_t><<<grid_for(P, 256), 256, 0, s>>>(P, key.as<uint64_t>(), rec.as<float4>(), kp, ws, k32.as<uint32_t>());
            else key_compress_kernel<uint32_t><<<grid_for(P, 256), 256, 0, s>>>(P, key.as<uint64_t>(), kp, k32.as<uint32_t>());
            HIPCHK(hipGetLastError());
            TRY(sort_pairs(s, k32.as<uint32_t>(), k32s.as<uint32_t>(), idx.as<uint32_t>(), idx0.as<uint32_t>(), (size_t)P, sort_bits));
            vox_gather_kernel<uint32_t><<<grid_for(P, 256), 256, 0, s>>>(P, rec.as<float4>(), idx0.as<uint32_t>(), rec_s.as<float4>(), k32s.as<uint32_t>(), kp,
                                                                         keys_sorted, foff, nfr, heads.as<uint64_t>());
            HIPCHK(hipGetLastError());
        } else { // wide maps: 64-bit keys, still only the bits that vary (compressed in place; expanded into the unsorted keys' buffer)
            if (ws) key_compress_win_kernel<uint64_t><<<grid_for(P, 256), 256, 0, s>>>(P, key.as<uint64_t>(), rec.as<float4>(), kp, ws, key.as<uint64_t>());
            else key_compress_kernel<uint64_t><<<grid_for(P, 256), 256, 0, s>>>(P, key.as<uint64_t>(), kp, key.as<uint64_t>());
            HIPCHK(hipGetLastError());
            TRY(sort_pairs(s, key.as<uint64_t>(), key_s.as<uint64_t>(), idx.as<uint32_t>(), idx0.as<uint32_t>(), (size_t)P, sort_bits));
            keys_sorted = key.as<uint64_t>();
            vox_gather_kernel<uint64_t><<<grid_for(P, 256), 256, 0, s>>>(P, rec.as<float4>(), idx0.as<uint32_t>(), rec_s.as<float4>(), key_s.as<uint64_t>(), kp,
                                                                         keys_sorted, foff, nfr, heads.as<uint64_t>());
            HIPCHK(hipGetLastError());
        }
        // ONE inclusive scan numbers roots (high word) and (root, frame) segments (low word)
        TRY(scan_incl<uint64_t>(s, heads.as<uint64_t>(), incl.as<uint64_t>(), (size_t)P));
        uint64_t last = 0;
        { Fetch f; f.add(&last, incl.as<uint64_t>() + (P - 1), 8); HIPCHK(f.run(s)); }
        R = (int64_t)(last >> 32); NS = (int64_t)(last & 0xFFFFFFFFull);
        HIPCHK(root_key.alloc(8 * (size_t)R)); HIPCHK(root_seg.alloc(4 * ((size_t)R + 1)));
        HIPCHK(seg_start.alloc(4 * ((size_t)NS + 1))); HIPCHK(seg_root.alloc(4 * (size_t)NS)); HIPCHK(seg_frame.alloc(4 * (size_t)NS));
        vox_tables_kernel<<<grid_for(P, 256), 256, 0, s>>>(P, keys_sorted, rec_s.as<float4>(), heads.as<uint64_t>(), incl.as<uint64_t>(),
                                                           root_key.as<uint64_t>(), root_seg.as<uint32_t>(), seg_start.as<uint32_t>(),
                                                           seg_root.as<uint32_t>(), seg_frame.as<int32_t>());
        HIPCHK(hipGetLastError());
    } // (the work buffers' destructors wait for the stream before the pool takes them back)
    h->info.n_roots = R;
    h->info.sort_ms = now_ms() - t0; t0 = now_ms();

    // -- 3. octree: root clusters and decisions for every root, children / grandchildren for the roots that split
    DevBuf segm1(s), segm2(s);
    HIPCHK(segm1.alloc(4 * (size_t)NS)); HIPCHK(segm2.alloc(8 * (size_t)NS));
    DevBuf segcl(s), n_plane(s), n_vox(s), n_fac(s), mask(s), rootinfo(s), plane0(s), is_split(s), seg_split(s), split_excl(s), seg_excl(s);
    HIPCHK(segcl.alloc(80 * (size_t)NS));
    HIPCHK(n_plane.alloc(4 * (R + 1))); HIPCHK(n_vox.alloc(4 * (R + 1))); HIPCHK(n_fac.alloc(8 * (R + 1)));
    HIPCHK(mask.alloc(8 * R)); HIPCHK(rootinfo.alloc(4 * R)); HIPCHK(plane0.alloc(48 * R));
    HIPCHK(is_split.alloc(4 * (R + 1))); HIPCHK(seg_split.alloc(4 * (NS + 1)));
    HIPCHK(split_excl.alloc(4 * (R + 1))); HIPCHK(seg_excl.alloc(4 * (NS + 1)));
    HIPCHK(hipMemsetAsync(n_plane.as<int32_t>() + R, 0, 4, s));
    HIPCHK(hipMemsetAsync(n_vox.as<int32_t>() + R, 0, 4, s));
    HIPCHK(hipMemsetAsync(n_fac.as<int64_t>() + R, 0, 8, s));
    HIPCHK(hipMemsetAsync(is_split.as<uint32_t>() + R, 0, 4, s));
    HIPCHK(hipMemsetAsync(seg_split.as<uint32_t>() + NS, 0, 4, s));
    vox_seg_small_kernel<<<grid_for(NS, 64), 64, 0, s>>>(NS, seg_start.as<uint32_t>(), rec_s.as<float4>(), segcl.as<double>(),
                                                         segm1.as<uint32_t>(), segm2.as<uint64_t>());
    vox_seg_big_kernel<<<(unsigned)NS, 64, 0, s>>>(NS, seg_start.as<uint32_t>(), rec_s.as<float4>(), segcl.as<double>(),
                                                   segm1.as<uint32_t>(), segm2.as<uint64_t>());
    HIPCHK(hipGetLastError());
    RootArgs ra{};
    ra.R = R; ra.root_seg = root_seg.as<uint32_t>(); ra.seg_frame = seg_frame.as<int32_t>(); ra.segcl = segcl.as<double>();
    ra.poses = d_poses.as<double>();
    for (int L = 0; L < 3; ++L) ra.ratio[L] = h->opts.eigen_ratio[L];
    ra.min_ps = h->opts.min_points;
    ra.n_plane = n_plane.as<int32_t>(); ra.n_vox = n_vox.as<int32_t>(); ra.n_fac = n_fac.as<int64_t>();
    ra.mask = mask.as<uint64_t>(); ra.rootinfo = rootinfo.as<uint32_t>(); ra.plane0 = plane0.as<double>();
    ra.is_split = is_split.as<uint32_t>(); ra.seg_split = seg_split.as<uint32_t>();
    vox_root_kernel<<<grid_for(R, 64), 64, 0, s>>>(ra);
    HIPCHK(hipGetLastError());
    TRY(scan_excl<uint32_t>(s, is_split.as<uint32_t>(), split_excl.as<uint32_t>(), (size_t)R + 1));
    TRY(scan_excl<uint32_t>(s, seg_split.as<uint32_t>(), seg_excl.as<uint32_t>(), (size_t)NS + 1));
    uint32_t NRS = 0, NSS = 0;
    { Fetch f; f.add(&NRS, split_excl.as<uint32_t>() + R, 4); f.add(&NSS, seg_excl.as<uint32_t>() + NS, 4); HIPCHK(f.run(s)); }

    DevBuf split_roots(s), split_segs(s), m1(s), m2(s), cnt(s), base(s), nodecl(s), tmp_count(s), tmp_base(s), tmp_plane(s), tmp_nf(s);
    SplitArgs sa{};
    if (NRS > 0) {
        HIPCHK(split_roots.alloc(4 * (size_t)NRS)); HIPCHK(split_segs.alloc(4 * (size_t)NSS));
        vox_compact_kernel<<<grid_for(R, 256), 256, 0, s>>>(R, is_split.as<uint32_t>(), split_excl.as<uint32_t>(), split_roots.as<uint32_t>());
        vox_compact_kernel<<<grid_for(NS, 256), 256, 0, s>>>(NS, seg_split.as<uint32_t>(), seg_excl.as<uint32_t>(), split_segs.as<uint32_t>());
        HIPCHK(m1.alloc(4 * (size_t)NSS)); HIPCHK(m2.alloc(8 * (size_t)NSS)); HIPCHK(cnt.alloc(4 * ((size_t)NSS + 1)));
        HIPCHK(base.alloc(4 * ((size_t)NSS + 1)));
        vox_split_masks_kernel<<<grid_for((int64_t)NSS + 1, 256), 256, 0, s>>>(NSS, split_segs.as<uint32_t>(), segm1.as<uint32_t>(),
                                                                               segm2.as<uint64_t>(), m1.as<uint32_t>(), m2.as<uint64_t>(),
                                                                               cnt.as<uint32_t>());
        HIPCHK(hipGetLastError());
        TRY(scan_excl<uint32_t>(s, cnt.as<uint32_t>(), base.as<uint32_t>(), (size_t)NSS + 1));
        uint32_t NN = 0;
        { Fetch f; f.add(&NN, base.as<uint32_t>() + NSS, 4); HIPCHK(f.run(s)); }
        HIPCHK(nodecl.alloc(80 * (size_t)NN));
        vox_split_cluster_kernel<<<NSS, 64, 0, s>>>(split_segs.as<uint32_t>(), seg_start.as<uint32_t>(), rec_s.as<float4>(),
                                                    m1.as<uint32_t>(), m2.as<uint64_t>(), base.as<uint32_t>(), nodecl.as<double>());
        HIPCHK(hipGetLastError());
        const size_t tmp_cap = (size_t)(P / h->opts.min_points) + 1; // a PLANE node holds >= min_points points of its own
        HIPCHK(tmp_count.alloc(4)); HIPCHK(tmp_base.alloc(4 * (size_t)NRS));
        HIPCHK(tmp_plane.alloc(48 * tmp_cap)); HIPCHK(tmp_nf.alloc(4 * tmp_cap));
        HIPCHK(hipMemsetAsync(tmp_count.p, 0, 4, s));
        sa.split_roots = split_roots.as<uint32_t>(); sa.root_seg = root_seg.as<uint32_t>(); sa.seg_excl = seg_excl.as<uint32_t>();
        sa.seg_frame = seg_frame.as<int32_t>(); sa.m1 = m1.as<uint32_t>(); sa.m2 = m2.as<uint64_t>(); sa.base = base.as<uint32_t>();
        sa.nodecl = nodecl.as<double>(); sa.poses = d_poses.as<double>();
        for (int L = 0; L < 3; ++L) sa.ratio[L] = h->opts.eigen_ratio[L];
        sa.min_ps = h->opts.min_points;
        sa.n_plane = n_plane.as<int32_t>(); sa.n_vox = n_vox.as<int32_t>(); sa.n_fac = n_fac.as<int64_t>();
        sa.mask = mask.as<uint64_t>(); sa.rootinfo = rootinfo.as<uint32_t>();
        sa.tmp_count = tmp_count.as<unsigned>(); sa.tmp_base = tmp_base.as<int32_t>();
        sa.tmp_plane = tmp_plane.as<double>(); sa.tmp_nf = tmp_nf.as<int32_t>();
        vox_split_decide_kernel<<<NRS, 64, 0, s>>>(sa);
        HIPCHK(hipGetLastError());
    }
    DevBuf plane_first(s), vox_first(s), fac_first(s);
    HIPCHK(plane_first.alloc(4 * (R + 1))); HIPCHK(vox_first.alloc(4 * (R + 1))); HIPCHK(fac_first.alloc(8 * (R + 1)));
    TRY(scan_excl<int32_t>(s, n_plane.as<int32_t>(), plane_first.as<int32_t>(), (size_t)R + 1));
    TRY(scan_excl<int32_t>(s, n_vox.as<int32_t>(), vox_first.as<int32_t>(), (size_t)R + 1));
    TRY(scan_excl<int64_t>(s, n_fac.as<int64_t>(), fac_first.as<int64_t>(), (size_t)R + 1));
    int32_t n_planes = 0, V = 0;
    int64_t F = 0;
    Fetch fc;
    fc.add(&n_planes, plane_first.as<int32_t>() + R, 4);
    fc.add(&V, vox_first.as<int32_t>() + R, 4);
    fc.add(&F, fac_first.as<int64_t>() + R, 8);
    DevBuf wv(s), wf(s);
    if (ws) { // where every window's voxels and factors begin
        HIPCHK(wv.alloc(8 * (size_t)n_win)); HIPCHK(wf.alloc(8 * (size_t)n_win));
        HIPCHK(hipMemsetAsync(wv.p, 0xFF, 8 * (size_t)n_win, s)); HIPCHK(hipMemsetAsync(wf.p, 0xFF, 8 * (size_t)n_win, s));
        vox_win_first_kernel<<<grid_for(R, 256), 256, 0, s>>>(R, root_seg.as<uint32_t>(), seg_frame.as<int32_t>(), ws, vox_first.as<int32_t>(),
                                                              fac_first.as<int64_t>(), wv.as<int64_t>(), wf.as<int64_t>());
        HIPCHK(hipGetLastError());
        h->win_v0.assign((size_t)n_win + 1, 0); h->win_f0.assign((size_t)n_win + 1, 0);
        fc.add(h->win_v0.data(), wv.p, 8 * (size_t)n_win);
        fc.add(h->win_f0.data(), wf.p, 8 * (size_t)n_win);
    }
    HIPCHK(fc.run(s)); // the totals and the windows' first voxels / factors in one round trip
    h->info.n_planes = n_planes; h->info.n_voxels = V; h->info.n_factors = F;
    if (ws) {
        h->win_v0[(size_t)n_win] = V; h->win_f0[(size_t)n_win] = F;
        for (int k = n_win - 1; k >= 0; --k) // a window without roots begins (and ends) where the next one begins
            if (h->win_v0[(size_t)k] < 0) { h->win_v0[(size_t)k] = h->win_v0[(size_t)k + 1]; h->win_f0[(size_t)k] = h->win_f0[(size_t)k + 1]; }
    }
    h->info.count_ms = now_ms() - t0; t0 = now_ms();

    // -- emission
    DevBuf plane(s), vox_off(s), pose_idx(s), clusters(s), vox_label(s);
    HIPCHK(plane.alloc(48 * (size_t)n_planes)); HIPCHK(vox_off.alloc(8 * ((size_t)V + 1)));
    HIPCHK(pose_idx.alloc(4 * (size_t)F)); HIPCHK(clusters.alloc(80 * (size_t)F)); HIPCHK(vox_label.alloc(8 * (size_t)V));
    vox_root_emit_kernel<<<grid_for(NS, 256), 256, 0, s>>>(NS, seg_root.as<uint32_t>(), root_seg.as<uint32_t>(), seg_frame.as<int32_t>(),
                                                           segcl.as<double>(), rootinfo.as<uint32_t>(), n_vox.as<int32_t>(),
                                                           plane0.as<double>(), plane_first.as<int32_t>(), vox_first.as<int32_t>(),
                                                           fac_first.as<int64_t>(), plane.as<double>(), vox_off.as<int64_t>(),
                                                           pose_idx.as<int32_t>(), clusters.as<double>(), vox_label.as<int32_t>());
    HIPCHK(hipGetLastError());
    if (NRS > 0) {
        sa.plane_first = plane_first.as<int32_t>(); sa.vox_first = vox_first.as<int32_t>(); sa.fac_first = fac_first.as<int64_t>();
        sa.plane = plane.as<double>(); sa.vox_off = vox_off.as<int64_t>(); sa.pose_idx = pose_idx.as<int32_t>();
        sa.clusters = clusters.as<double>(); sa.vox_label = vox_label.as<int32_t>();
        vox_split_emit_kernel<<<NRS, 64, 0, s>>>(sa);
        HIPCHK(hipGetLastError());
    }
    if (ws && F > 0) { // pose indices relative to the window's first frame, as the window's own map would have them
        vox_pose_rel_kernel<<<grid_for(F, 256), 256, 0, s>>>(F, pose_idx.as<int32_t>(), ws);
        HIPCHK(hipGetLastError());
    }
    vox_close_offsets_kernel<<<1, 1, 0, s>>>(vox_off.as<int64_t>() + V, fac_first.as<int64_t>() + R);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));
    h->info.write_ms = now_ms() - t0;

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

// stream != nullptr: the map works on the caller's stream and does not own it (the caller keeps it alive as long as the map)
static int32_t build_scans_common(lvba_scans_t sc, int32_t frame_begin, int32_t n_frames, const double *poses,
                                  const lvba_voxel_opts *opts, hipStream_t stream, int window_size, lvba_voxmap_t *out);
int32_t lvba_voxmap_build_scans_on(lvba_scans_t sc, int32_t frame_begin, int32_t n_frames, const double *poses,
                                   const lvba_voxel_opts *opts, hipStream_t stream, lvba_voxmap_t *out)
{
    return build_scans_common(sc, frame_begin, n_frames, poses, opts, stream, 0, out);
}
// One map for the windows of window_size frames that make up [frame_begin, frame_begin + n_frames) (window_ba.hip); the windows'
// own maps are views into it (lvba_voxmap_window_view).
int32_t lvba_voxmap_build_scans_joint(lvba_scans_t sc, int32_t frame_begin, int32_t n_frames, int32_t window_size, const double *poses,
                                      const lvba_voxel_opts *opts, hipStream_t stream, lvba_voxmap_t *out)
{
    if (window_size < 1) return lvba_fail(LVBA_ERR_ARG, "window_size must be >= 1");
    return build_scans_common(sc, frame_begin, n_frames, poses, opts, stream, window_size, out);
}
// where window w's admitted voxels and factors sit in the joint map's arrays
int32_t lvba_voxmap_window_range(lvba_voxmap_t joint, int32_t w, int64_t *v0, int64_t *v1, int64_t *f0, int64_t *f1)
{
    if (!joint || joint->is_view || w < 0 || w >= joint->n_windows || !v0 || !v1 || !f0 || !f1) return lvba_fail(LVBA_ERR_ARG, "bad argument");
    if (joint->window_size == 0) { *v0 = 0; *v1 = joint->info.n_voxels; *f0 = 0; *f1 = joint->info.n_factors; return LVBA_OK; }
    *v0 = joint->win_v0[(size_t)w]; *v1 = joint->win_v0[(size_t)w + 1];
    *f0 = joint->win_f0[(size_t)w]; *f1 = joint->win_f0[(size_t)w + 1];
    return LVBA_OK;
}
int32_t lvba_voxmap_window_view(lvba_voxmap_t joint, int32_t w, lvba_voxmap_t *out)
{
    if (!joint || !out) return lvba_fail(LVBA_ERR_ARG, "null argument");
    *out = nullptr;
    if (joint->is_view || w < 0 || w >= joint->n_windows) return lvba_fail(LVBA_ERR_ARG, "window %d of %d", w, joint->n_windows);
    lvba_voxmap_s *v = new (std::nothrow) lvba_voxmap_s();
    if (!v) return lvba_fail(LVBA_ERR_NOMEM, "host allocation failed");
    v->device = joint->device; v->stream = joint->stream; v->owns_stream = false; v->opts = joint->opts; v->is_view = true;
    v->info = joint->info;
    if (joint->window_size == 0) { // a single window: the map itself
        v->n_frames = joint->n_frames;
        v->d_vox_off = joint->d_vox_off; v->d_pose_idx = joint->d_pose_idx; v->d_clusters = joint->d_clusters; v->d_vox_label = joint->d_vox_label;
    } else {
        const int64_t v0 = joint->win_v0[(size_t)w], v1 = joint->win_v0[(size_t)w + 1], f0 = joint->win_f0[(size_t)w], f1 = joint->win_f0[(size_t)w + 1];
        v->n_frames = std::min(joint->window_size, joint->n_frames - w * joint->window_size);
        v->info.n_voxels = v1 - v0; v->info.n_factors = f1 - f0;
        v->d_vox_off = joint->d_vox_off + v0; v->d_pose_idx = joint->d_pose_idx + f0;
        v->d_clusters = joint->d_clusters + 10 * f0; v->d_vox_label = joint->d_vox_label + 2 * v0;
    }
    v->d_root_key = joint->d_root_key; // (labels index the joint map's root table)
    v->window_size = joint->window_size; v->n_windows = joint->n_windows;
    *out = v;
    return LVBA_OK;
}
static int32_t build_scans_common(lvba_scans_t sc, int32_t frame_begin, int32_t n_frames, const double *poses,
                                  const lvba_voxel_opts *opts, hipStream_t stream, int window_size, lvba_voxmap_t *out)
{
    if (!out) return lvba_fail(LVBA_ERR_ARG, "out is null");
    *out = nullptr;
    if (!sc || !poses) return lvba_fail(LVBA_ERR_ARG, "null argument");
    if (frame_begin < 0 || n_frames < 1 || frame_begin + n_frames > sc->n_frames)
        return lvba_fail(LVBA_ERR_ARG, "frame range [%d, %d) outside the %d uploaded frames", frame_begin,
                         frame_begin + n_frames, sc->n_frames);
    lvba_voxel_opts o;
    lvba_voxel_default_opts(&o);
    if (opts) o = *opts;
    if (!(o.voxel_size > 0.0) || o.min_points < 1) return lvba_fail(LVBA_ERR_ARG, "voxel_size must be > 0 and min_points >= 1");
    if (o.layer_limit != 2) return lvba_fail(LVBA_ERR_UNSUPPORTED, "layer_limit is fixed at 2 (bavoxel.hpp:13), got %d", o.layer_limit);
    HIPCHK(hipSetDevice(sc->device));
    lvba_voxmap_s *h = new (std::nothrow) lvba_voxmap_s();
    if (!h) return lvba_fail(LVBA_ERR_NOMEM, "host allocation failed");
    h->device = sc->device;
    h->n_frames = n_frames;
    h->opts = o;
    if (stream) {
        h->stream = stream;
        h->owns_stream = false;
    } else if (lvba::StreamCache::get().acquire(&h->stream) != hipSuccess) {
        delete h;
        return lvba_fail(LVBA_ERR_DEVICE, "hipStreamCreate failed");
    }
    const int32_t rc = voxmap_build_impl(h, sc, frame_begin, poses, window_size);
    if (rc != LVBA_OK) { lvba_voxmap_destroy(h); return rc; }
    *out = h;
    return LVBA_OK;
}

extern "C" int32_t lvba_voxmap_build_scans(lvba_scans_t sc, int32_t frame_begin, int32_t n_frames, const double *poses,
                                           const lvba_voxel_opts *opts, lvba_voxmap_t *out)
{
    return lvba_voxmap_build_scans_on(sc, frame_begin, n_frames, poses, opts, nullptr, out);
}

extern "C" int32_t lvba_voxmap_build(int32_t device, int32_t n_frames, const void *const *frame_points,
                                     const int64_t *frame_count, int32_t point_stride_bytes, const double *poses,
                                     const lvba_voxel_opts *opts, lvba_voxmap_t *out)
{
    if (!out) return lvba_fail(LVBA_ERR_ARG, "out is null");
    *out = nullptr;
    if (!poses) return lvba_fail(LVBA_ERR_ARG, "poses is null");
    const double t0 = now_ms();
    lvba_scans_t sc = nullptr;
    TRY(lvba_scans_create(device, n_frames, frame_points, frame_count, point_stride_bytes, &sc));
    const double up = now_ms() - t0;
    const int32_t rc = lvba_voxmap_build_scans(sc, 0, n_frames, poses, opts, out);
    lvba_scans_destroy(sc);
    if (rc == LVBA_OK) (*out)->info.upload_ms = up;
    return rc;
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
        if (V > 0) HIPCHK(lvba::copy_d2h(voxel_off, h->d_vox_off, 8 * (V + 1)));
        else voxel_off[0] = 0;
    }
    if (pose_idx && F > 0) HIPCHK(lvba::copy_d2h(pose_idx, h->d_pose_idx, 4 * F));
    if (clusters && F > 0) HIPCHK(lvba::copy_d2h(clusters, h->d_clusters, 80 * F));
    if (voxel_key && V > 0) {
        lvba::hvec<int32_t> label(2 * V);
        lvba::hvec<uint64_t> rk(h->info.n_roots);
        HIPCHK(lvba::copy_d2h(label.data(), h->d_vox_label, 8 * V));
        HIPCHK(lvba::copy_d2h(rk.data(), h->d_root_key, 8 * h->info.n_roots));
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
    lvba::hvec<int64_t> off(V + 1);
    lvba::hvec<int32_t> idx(F);
    TRY(lvba_voxmap_export(h, off.data(), idx.data(), nullptr, nullptr)); // CSR structure to the host, clusters stay in HBM
    return lvba_balm_create_dev(h->n_frames, V, off.data(), idx.data(), h->d_clusters, h->device, out);
}

extern "C" int32_t lvba_voxmap_find_planes(lvba_voxmap_t h, int64_t n, const double *X, double *plane, uint8_t *valid)
{
    if (!h || n < 0 || (n > 0 && (!X || !plane || !valid))) return lvba_fail(LVBA_ERR_ARG, "null argument");
    if (n == 0) return LVBA_OK;
    if (h->is_view || h->window_size > 0)
        return lvba_fail(LVBA_ERR_UNSUPPORTED, "a joint map of several windows (or a view into one) has no key lookup");
    if (h->info.n_roots == 0) {
        memset(plane, 0, 32 * (size_t)n);
        memset(valid, 0, (size_t)n);
        return LVBA_OK;
    }
    HIPCHK(hipSetDevice(h->device));
    DevBuf dX(h->stream), dpl(h->stream), dval(h->stream);
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
