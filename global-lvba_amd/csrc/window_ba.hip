// window_ba.hip -- the window-BA stage that turns raw scans + odometry into anchor frames, on the device.
//
// Replaces LvbaSystem::runWindowBA (reference src/lvba_system.cpp:204-310): per window of `window_size` frames
//   cut_voxel / recut / tras_opt at the odometry poses   :247-257   -> lvba_voxmap_build_scans + lvba_voxmap_to_balm
//   skip if fewer than 3 plane voxels per frame           :258-262
//   BALM2::damping_iter                                   :264       -> lvba_balm_refine
//   re-alignment to the odometry pose of the first frame  :268-279   (12-double algebra, host)
//   relative poses to the anchor, merge of the clouds     :284-299   -> wba_merge_kernel (fp32 write-back as pl_transform,
//                                                                       include/BALM/tools.hpp:385-395)
//   down_sampling_voxel2                                  tools.hpp:300-359 -> key + stable sort + first-minimum per voxel
// The raw clouds never leave HBM between the stages; the anchor clouds are born there (a new lvba_scans_t) and feed the
// global stages' lvba_voxmap_build_scans directly.
// down_sampling_voxel2 emits survivors in unordered_map order (unspecified); here: sorted by voxel key (x, y, z).
#include <atomic>
#include <functional>
#include <string>
#include <thread>
#include "host_arena.h"
#include "voxel_internal.h"
#include "lvba_internal.h"
#include "block_system.h"

using namespace lvba;

namespace {

constexpr int KEY_BIAS = 1 << 20;

// merged cloud of one window: every point moved into the anchor frame with its frame's relative pose and rounded to fp32
// (pl_transform); plus the leaf-voxel key and squared distance to the voxel centre of down_sampling_voxel2.
__global__ void wba_merge_kernel(int64_t P, const float *__restrict__ pts, const int64_t *__restrict__ frame_off, int n_frames,
                                 const double *__restrict__ rel, double leaf, float *__restrict__ out,
                                 uint64_t *__restrict__ key, double *__restrict__ d2, uint32_t *__restrict__ idx,
                                 int *__restrict__ err, int *__restrict__ range_partial /* voxel_internal.h: key_range_update */)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int kb[3] = {0, 0, 0};
    if (i < P) {
    const int64_t base = frame_off[0];
    int lo = 0, hi = n_frames;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (frame_off[mid] - base <= i) lo = mid; else hi = mid;
    }
    const double *T = rel + 12 * (int64_t)lo;
    const double p0 = pts[3 * i], p1 = pts[3 * i + 1], p2 = pts[3 * i + 2];
    const float q[3] = {(float)(T[0] * p0 + T[1] * p1 + T[2] * p2 + T[9]), (float)(T[3] * p0 + T[4] * p1 + T[5] * p2 + T[10]),
                        (float)(T[6] * p0 + T[7] * p1 + T[8] * p2 + T[11])};
    out[3 * i] = q[0]; out[3 * i + 1] = q[1]; out[3 * i + 2] = q[2];
    if (key) {
    int64_t k[3];
    double dd = 0.0;
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float loc = (float)((double)q[j] / leaf);
        if (loc < 0.f) loc -= 1.f;
        ok = ok && (fabsf(loc) < (float)KEY_BIAS);
        k[j] = ok ? (int64_t)loc : 0;
        const double c = ((double)k[j] + 0.5) * leaf;
        const double d = (double)q[j] - c;
        dd = __dadd_rn(dd, __dmul_rn(d, d)); // dx*dx + dy*dy + dz*dz, left to right, no contraction
    }
    if (!ok) *err = 1;
    key[i] = ((uint64_t)(k[0] + KEY_BIAS) << 42) | ((uint64_t)(k[1] + KEY_BIAS) << 21) | (uint64_t)(k[2] + KEY_BIAS);
    d2[i] = dd;
    idx[i] = (uint32_t)i;
#pragma unroll
    for (int j = 0; j < 3; ++j) kb[j] = (int)(k[j] + KEY_BIAS);
    }
    }
    if (key) key_range_update(range_partial, kb, i < P); // (the sort runs on the bits that vary)
}
// after the stable sort by key (K: the re-packed keys, equal exactly where the leaf keys are): the run leader picks the first
// minimum of d2 in merged order
template <class K>
__global__ void wba_pick_kernel(int64_t P, const K *__restrict__ key_s, const uint32_t *__restrict__ order,
                                const double *__restrict__ d2, uint32_t *__restrict__ flag, uint32_t *__restrict__ pick)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const bool head = i == 0 || key_s[i] != key_s[i - 1];
    flag[i] = head ? 1u : 0u;
    if (!head) return;
    const K k = key_s[i];
    uint32_t best = order[i];
    double bd = d2[best];
    for (int64_t j = i + 1; j < P && key_s[j] == k; ++j) {
        const uint32_t c = order[j];
        const double d = d2[c];
        if (d < bd) { bd = d; best = c; }
    }
    pick[i] = best;
}
__global__ void wba_compact_kernel(int64_t P, const uint32_t *__restrict__ flag, const uint32_t *__restrict__ excl,
                                   const uint32_t *__restrict__ pick, const float *__restrict__ merged, float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P || !flag[i]) return;
    const uint32_t s = pick[i], o = excl[i];
    out[3 * (int64_t)o] = merged[3 * (int64_t)s];
    out[3 * (int64_t)o + 1] = merged[3 * (int64_t)s + 1];
    out[3 * (int64_t)o + 2] = merged[3 * (int64_t)s + 2];
}

// joint form of the merge stage (all windows in one pass): (code of the point's window << total) | re-packed leaf key -- the
// points of a window stay together, inside it the order is the leaf key's, as in the window's own sort.  code[window]: its rank
// among the windows that produce an anchor cloud; the others' points get the code above all of those and sort to the end.
template <class K>
__global__ void wba_compress_win_kernel(int64_t P, const uint64_t *key, const int64_t *__restrict__ frame_off, int n_frames, int ws,
                                        const uint32_t *__restrict__ code, const KeyPack kp, K *out /* may be key */)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const int64_t base = frame_off[0];
    int lo = 0, hi = n_frames;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (frame_off[mid] - base <= i) lo = mid; else hi = mid;
    }
    const uint64_t c = code[lo / ws];
    out[i] = (K)((c << kp.total) | key_compress<uint64_t>(key[i], kp)); // (in 64 bits, narrowed afterwards: key_pack.h)
}
// anchor points of every window: the scan of the leader flags read at the windows' bounds in the sorted sequence
__global__ void wba_bounds_kernel(int n, const int64_t *__restrict__ at, const uint32_t *__restrict__ excl, uint32_t *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = excl[at[i]];
}

inline void mat3_mul(const double *A, const double *B, double *C)
{
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) C[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
}
inline void mat3_mulT(const double *A, const double *B, double *C) // A * B^T
{
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) C[3 * r + c] = A[3 * r] * B[3 * c] + A[3 * r + 1] * B[3 * c + 1] + A[3 * r + 2] * B[3 * c + 2];
}
inline void mat3T_mul(const double *A, const double *B, double *C) // A^T * B
{
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) C[3 * r + c] = A[r] * B[c] + A[3 + r] * B[3 + c] + A[6 + r] * B[6 + c];
}

} // namespace

extern "C" void lvba_window_default_opts(lvba_window_opts *o)
{
    if (!o) return;
    o->window_size = 10;     // include/dataset_io.h:71
    o->use_rel = 1;          // config.yaml window_ba.use_window_ba_rel
    o->merge_only = 0;
    o->lm_mode = 0;
    o->anchor_leaf = 0.1;    // include/dataset_io.h:72
    lvba_voxel_default_opts(&o->voxel);
    o->voxel.voxel_size = 0.5; // stage1_root_voxel_size_, include/dataset_io.h:76
    lvba_balm_default_opts(&o->lm);
}

extern "C" int32_t lvba_scans_info(lvba_scans_t sc, int32_t *n_frames, int64_t *frame_count)
{
    if (!sc) return lvba_fail(LVBA_ERR_ARG, "null handle");
    if (n_frames) *n_frames = sc->n_frames;
    if (frame_count)
        for (int f = 0; f < sc->n_frames; ++f) frame_count[f] = sc->frame_off[f + 1] - sc->frame_off[f];
    return LVBA_OK;
}

extern "C" int32_t lvba_scans_download(lvba_scans_t sc, int32_t frame, float *xyz)
{
    if (!sc || !xyz) return lvba_fail(LVBA_ERR_ARG, "null argument");
    if (frame < 0 || frame >= sc->n_frames) return lvba_fail(LVBA_ERR_ARG, "frame %d out of range [0,%d)", frame, sc->n_frames);
    HIPCHK(hipSetDevice(sc->device));
    const int64_t n = sc->frame_off[frame + 1] - sc->frame_off[frame];
    if (n > 0) HIPCHK(lvba::copy_d2h(xyz, sc->d_pts + 3 * sc->frame_off[frame], 12 * (size_t)n));
    return LVBA_OK;
}

extern "C" int32_t lvba_window_ba(lvba_scans_t sc, const double *poses, const lvba_window_opts *opts, double *window_poses,
                                  double *rel_poses, int32_t *anchor_index, double *anchor_poses, int32_t *n_anchors,
                                  lvba_scans_t *anchor_scans, lvba_window_info *win_info)
{
    if (anchor_scans) *anchor_scans = nullptr;
    if (!sc || !poses || !rel_poses || !anchor_index || !anchor_poses || !n_anchors || !anchor_scans)
        return lvba_fail(LVBA_ERR_ARG, "null argument");
    lvba_window_opts o;
    lvba_window_default_opts(&o);
    if (opts) o = *opts;
    if (o.window_size < 1) return lvba_fail(LVBA_ERR_ARG, "window_size must be >= 1");
    HIPCHK(hipSetDevice(sc->device));
    const int n = sc->n_frames, w = o.window_size;
    hipStream_t s = nullptr;
    HIPCHK(lvba::StreamCache::get().acquire(&s));
    struct StreamGuard { hipStream_t s; ~StreamGuard() { lvba::StreamCache::get().release(s); } } sguard{s};

    static const double I12[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
    for (int i = 0; i < n; ++i) { // rel_poses_to_anchor_.assign(total, IMUST()), anchor_index -1 (:338-339)
        memcpy(rel_poses + 12 * i, I12, sizeof I12);
        anchor_index[i] = -1;
    }
    if (window_poses) memcpy(window_poses, poses, 96 * (size_t)n);

    struct AnchorCloud { float *d; int64_t n; };
    lvba::hvec<AnchorCloud> clouds;
    auto free_clouds = [&]() { for (auto &c : clouds) DevicePool::get().free(c.d); clouds.clear(); };
    // One window = map -> problem -> LM -> anchor cloud; windows are independent (src/lvba_system.cpp:232-302 runs them one
    // after the other).  Three stages:
    //   1. voxel map of every window (a few host threads, each with its own stream: a single map build leaves the GPU idle
    //      most of the time -- launch and synchronisation latency);
    //   2. the LM refinements of ALL windows in lock-step as one grouped problem (lvba_balm_set_groups /
    //      lvba_balm_refine_groups: one evaluation, one band factorisation with a damping value per window, one cost pass per
    //      iteration for all windows; every window keeps its own LM state).  lm_mode = 1, a single
    //      window, or a broken pivot in the joint factorisation: one window at a time, as before;
    //   3. alignment, relative poses, anchor merge + down-sampling: all windows in one pass (stage_finish_joint), or per window
    //      on the host threads again.
    // Results are assembled in window order below.
    struct WinResult { int32_t rc = LVBA_OK; std::string err; lvba_window_info info{}; lvba::hvec<double> x, rel; float *d_out = nullptr; int64_t n_out = 0;
                       lvba_voxmap_t map = nullptr; bool refined = false; };
    const int n_win = (n + w - 1) / w;
    lvba::hvec<WinResult> results((size_t)n_win);
    auto win_range = [&](int wi, int &start, int &cw) { start = wi * w; cw = std::min(w, n - start); };
    lvba_voxmap_t joint_map = nullptr; // stage 1 as ONE map of all windows (stage_map_joint); the windows' maps are views into it
    // ---- stage 1: the voxel map at the odometry poses (:247-257) and the skip rule (:258-262)
    auto stage_map = [&](int wi, hipStream_t ws, WinResult &R) -> int32_t {
        int start, cw;
        win_range(wi, start, cw);
        lvba_window_info &info = R.info;
        info = lvba_window_info{};
        info.start = start; info.n_frames = cw; info.anchor = -1;
        const double *x_odom = poses + 12 * (int64_t)start;
        R.x.assign(x_odom, x_odom + 12 * (size_t)cw);
        if (o.merge_only) return LVBA_OK;
        const double tw = now_ms();
        int32_t rc = lvba_voxmap_build_scans_on(sc, start, cw, x_odom, &o.voxel, ws, &R.map); // on the worker's stream, which outlives the map
        if (rc != LVBA_OK) return rc;
        lvba_voxmap_info_t mi;
        lvba_voxmap_info(R.map, &mi);
        info.n_voxels = mi.n_voxels; info.n_factors = mi.n_factors;
        info.map_ms = now_ms() - tw;
        if (mi.n_voxels < 3 * (int64_t)cw) { // :258-262
            info.skipped = 1;
            lvba_voxmap_destroy(R.map);
            R.map = nullptr;
        }
        return LVBA_OK;
    };
    // ---- stage 2, one window at a time: damping_iter on the window's own problem (:264)
    auto stage_lm_single = [&](int wi, hipStream_t, WinResult &R) -> int32_t {
        if (!R.map || R.refined) return LVBA_OK;
        lvba_window_info &info = R.info;
        double tw = now_ms();
        lvba_balm_t b = nullptr;
        int32_t rc = lvba_voxmap_to_balm(R.map, &b);
        lvba_voxmap_destroy(R.map);
        R.map = nullptr;
        if (rc != LVBA_OK) return rc;
        lvba::hvec<lvba_lm_trace> trace((size_t)std::max(1, o.lm.max_iter));
        int32_t nt = 0;
        lvba_balm_info_t bi;
        lvba_balm_info(b, &bi); // forces the one-off problem set-up (ordering, pair lists) so that it is timed apart
        info.setup_ms = now_ms() - tw;
        rc = lvba_balm_refine(b, R.x.data(), &o.lm, trace.data(), &nt);
        lvba_balm_destroy(b);
        if (rc < 0) return rc;
        info.lm_status = rc; info.n_iter = nt;
        if (nt > 0) {
            info.cost_first = trace[0].residual1;
            info.cost_last = trace[nt - 1].accepted ? trace[nt - 1].residual2 : trace[nt - 1].residual1;
        }
        info.solve_ms = now_ms() - tw;
        R.refined = true;
        (void)wi;
        return LVBA_OK;
    };
    // ---- stage 2, all windows at once.  Returns LVBA_OK with `done` = false when the windows have to go one by one.
    auto stage_lm_batched = [&](bool &done) -> int32_t {
        done = false;
        lvba::hvec<int> live;
        for (int wi = 0; wi < n_win; ++wi)
            if (results[(size_t)wi].map) live.push_back(wi);
        if (live.size() < 2) return LVBA_OK;
        const double t0 = now_ms();
        const int G = (int)live.size();
        lvba::hvec<int32_t> pose_off((size_t)G + 1, 0);
        lvba::hvec<int64_t> vox_off((size_t)G + 1, 0), fac_off((size_t)G + 1, 0);
        for (int k = 0; k < G; ++k) {
            const WinResult &R = results[(size_t)live[(size_t)k]];
            pose_off[(size_t)k + 1] = pose_off[(size_t)k] + R.info.n_frames;
            vox_off[(size_t)k + 1] = vox_off[(size_t)k] + R.info.n_voxels;
            fac_off[(size_t)k + 1] = fac_off[(size_t)k] + R.info.n_factors;
        }
        const int64_t V = vox_off[(size_t)G], F = fac_off[(size_t)G];
        if (F >= ((int64_t)1 << 31)) return LVBA_OK; // too large for one handle: one by one
        lvba::hvec<int64_t> off((size_t)V + 1, 0);
        lvba::hvec<int32_t> idx((size_t)F);
        lvba::hvec<double> x(12 * (size_t)pose_off[(size_t)G]);
        DevBuf d_clu(s);
        HIPCHK(d_clu.alloc(80 * (size_t)F));
        lvba::hvec<int64_t> joff; // with a joint map: its whole CSR structure in ONE pair of copies instead of two per window
        lvba::hvec<int32_t> jidx;
        if (joint_map) {
            lvba_voxmap_info_t ji;
            lvba_voxmap_info(joint_map, &ji);
            joff.resize((size_t)ji.n_voxels + 1); jidx.resize((size_t)std::max<int64_t>(ji.n_factors, 1));
            TRY(lvba_voxmap_export(joint_map, joff.data(), jidx.data(), nullptr, nullptr));
        }
        for (int k = 0; k < G; ++k) {
            const WinResult &R = results[(size_t)live[(size_t)k]];
            const int64_t v0 = vox_off[(size_t)k], f0 = fac_off[(size_t)k], nv = R.info.n_voxels, nf = R.info.n_factors;
            lvba::hvec<int64_t> o1((size_t)nv + 1);
            if (joint_map) {
                int64_t jv0, jv1, jf0, jf1;
                TRY(lvba_voxmap_window_range(joint_map, live[(size_t)k], &jv0, &jv1, &jf0, &jf1));
                memcpy(o1.data(), joff.data() + jv0, 8 * ((size_t)nv + 1));
                memcpy(idx.data() + f0, jidx.data() + jf0, 4 * (size_t)nf);
            } else
                TRY(lvba_voxmap_export(R.map, o1.data(), idx.data() + f0, nullptr, nullptr)); // CSR structure to the host, clusters stay in HBM
            for (int64_t a = 0; a <= nv; ++a) off[(size_t)(v0 + a)] = f0 + (o1[(size_t)a] - o1[0]);
            for (int64_t f = f0; f < f0 + nf; ++f) idx[(size_t)f] += pose_off[(size_t)k];
            HIPCHK(hipMemcpyAsync(d_clu.as<double>() + 10 * f0, lvba_voxmap_clusters(R.map), 80 * (size_t)nf, hipMemcpyDeviceToDevice, s));
            memcpy(x.data() + 12 * (size_t)pose_off[(size_t)k], R.x.data(), 96 * (size_t)R.info.n_frames);
        }
        HIPCHK(hipStreamSynchronize(s));
        const bool tm = timing_on("window");
        double tk = now_ms();
        auto mk = [&](const char *what) { if (tm) { const double t = now_ms(); fprintf(stderr, "[window_ba LM] %-16s %.3f ms\n", what, t - tk); tk = t; } };
        if (tm) fprintf(stderr, "[window_ba LM] %-16s %.3f ms\n", "export + concat", tk - t0);
        lvba_balm_t b = nullptr;
        TRY(balm_create_dev_trusted(pose_off[(size_t)G], V, off.data(), idx.data(), d_clu.as<double>(), sc->device, &b));
        struct Guard { lvba_balm_t b; ~Guard() { if (b) lvba_balm_destroy(b); } } guard{b};
        mk("create");
        TRY(lvba_balm_set_groups(b, G, pose_off.data(), vox_off.data()));
        mk("set_groups");
        lvba_balm_info_t bi;
        TRY(lvba_balm_info(b, &bi)); // the one-off set-up, timed apart
        mk("set-up");
        const double t1 = now_ms();
        lvba::hvec<int32_t> n_iter((size_t)G), status((size_t)G);
        lvba::hvec<double> first((size_t)G), last((size_t)G);
        const int32_t rc = lvba_balm_refine_groups(b, x.data(), &o.lm, n_iter.data(), status.data(), first.data(), last.data());
        if (rc == LVBA_NUM_FACTORIZATION) return LVBA_OK; // the windows are not independent in a broken factorisation: one by one
        if (rc < 0) return rc;
        const double t2 = now_ms();
        mk("refine_groups");
        for (int k = 0; k < G; ++k) {
            WinResult &R = results[(size_t)live[(size_t)k]];
            memcpy(R.x.data(), x.data() + 12 * (size_t)pose_off[(size_t)k], 96 * (size_t)R.info.n_frames);
            R.info.n_iter = n_iter[(size_t)k]; R.info.lm_status = status[(size_t)k];
            R.info.cost_first = first[(size_t)k]; R.info.cost_last = last[(size_t)k];
            R.info.setup_ms = (t1 - t0) / G; R.info.solve_ms = (t2 - t0) / G; // the joint problem's times, shared out evenly
            R.refined = true;
            lvba_voxmap_destroy(R.map);
            R.map = nullptr;
        }
        mk("results, maps freed");
        lvba_balm_destroy(guard.b);
        guard.b = nullptr;
        mk("handle destroyed");
        done = true;
        return LVBA_OK;
    };
    // ---- stage 3: alignment (:268-279), relative poses (:284-299), merge + down_sampling_voxel2 on the device
    auto align_window = [&](int wi, WinResult &R) {
        int start, cw;
        win_range(wi, start, cw);
        const double *x_odom = poses + 12 * (int64_t)start;
        lvba::hvec<double> &x = R.x;
        lvba::hvec<double> &rel = R.rel;
        rel.assign(12 * (size_t)cw, 0.0);
        const double *Ro0 = x_odom, *po0 = x_odom + 9;
        double R_align[9], p_align[3] = {0, 0, 0};
        if (o.use_rel) {
            mat3_mulT(Ro0, x.data(), R_align);
            for (int r = 0; r < 3; ++r)
                p_align[r] = po0[r] - (R_align[3 * r] * x[9] + R_align[3 * r + 1] * x[10] + R_align[3 * r + 2] * x[11]);
        }
        for (int j = 0; j < cw; ++j) {
            double Ra[9], pa[3];
            if (o.use_rel) {
                const double *Rj = x.data() + 12 * j, *pj = Rj + 9;
                mat3_mul(R_align, Rj, Ra);
                for (int r = 0; r < 3; ++r)
                    pa[r] = R_align[3 * r] * pj[0] + R_align[3 * r + 1] * pj[1] + R_align[3 * r + 2] * pj[2] + p_align[r];
            } else {
                memcpy(Ra, x_odom + 12 * j, 72);
                memcpy(pa, x_odom + 12 * j + 9, 24);
            }
            double *rj = rel.data() + 12 * j;
            mat3T_mul(Ro0, Ra, rj);
            const double d[3] = {pa[0] - po0[0], pa[1] - po0[1], pa[2] - po0[2]};
            for (int r = 0; r < 3; ++r) rj[9 + r] = Ro0[r] * d[0] + Ro0[3 + r] * d[1] + Ro0[6 + r] * d[2];
        }
    };
    auto stage_finish = [&](int wi, hipStream_t s, WinResult &R) -> int32_t {
        int start, cw;
        win_range(wi, start, cw);
        lvba_window_info &info = R.info;
        if (info.skipped) return LVBA_OK;
        double tw = now_ms();
        align_window(wi, R);
        lvba::hvec<double> &rel = R.rel;
        // merge + down_sampling_voxel2 on the device
        const int64_t p_begin = sc->frame_off[start], P = sc->frame_off[start + cw] - p_begin;
        const bool down = o.anchor_leaf >= 0.001 && P > 0; // tools.hpp:303
        float *d_out = nullptr;
        int64_t n_out = P;
        if (P > 0) {
            DevBuf d_rel(s), merged(s), key(s), d2(s), idx(s), d_err(s);
            HIPCHK(d_rel.alloc(96 * (size_t)cw)); HIPCHK(merged.alloc(12 * (size_t)P)); HIPCHK(d_err.alloc(28));
            HIPCHK(lvba::copy_h2d(d_rel.p, rel.data(), 96 * (size_t)cw)); // (pageable source: synchronous copy, voxelize.hip)
            int h_err[7] = {0}; // [0] error flag, [1..6] range of the biased key components
            DevBuf d_part(s);
            const int64_t n_slots = key_range_slots(P, 256);
            if (down) HIPCHK(d_part.alloc(24 * (size_t)n_slots));
            HIPCHK(hipMemsetAsync(d_err.p, 0, 28, s));
            if (down) { HIPCHK(key.alloc(8 * (size_t)P)); HIPCHK(d2.alloc(8 * (size_t)P)); HIPCHK(idx.alloc(4 * (size_t)P)); }
            wba_merge_kernel<<<grid_for(P, 256), 256, 0, s>>>(P, sc->d_pts + 3 * p_begin, sc->d_frame_off + start, cw, d_rel.as<double>(),
                                                              o.anchor_leaf, merged.as<float>(), down ? key.as<uint64_t>() : nullptr,
                                                              d2.as<double>(), idx.as<uint32_t>(), d_err.as<int>(), d_part.as<int>());
            HIPCHK(hipGetLastError());
            if (down) {
                key_range_reduce_kernel<<<key_range_reduce_grid(n_slots), 256, 0, s>>>(n_slots, d_part.as<int>(), d_err.as<int>() + 1);
                HIPCHK(hipGetLastError());
            }
            if (!down) {
                HIPCHK(hipStreamSynchronize(s));
                d_out = (float *)merged.release();
            } else {
                HIPCHK(hipStreamSynchronize(s));
                HIPCHK(lvba::copy_d2h(h_err, d_err.p, 28));
                if (h_err[0]) { return lvba_fail(LVBA_ERR_ARG, "window %d: a merged point is non-finite or outside +-2^20 anchor leaves", wi); }
                DevBuf key_s(s), order(s), flag(s), excl(s), pick(s);
                HIPCHK(key_s.alloc(8 * (size_t)P)); HIPCHK(order.alloc(4 * (size_t)P)); HIPCHK(flag.alloc(4 * ((size_t)P + 1)));
                HIPCHK(excl.alloc(4 * ((size_t)P + 1))); HIPCHK(pick.alloc(4 * (size_t)P));
                // the sort runs on the bits of the leaf key that vary (voxel_internal.h); run leaders only compare sorted keys for
                // equality, so the re-packed ones serve as they are
                const KeyPack kp = key_pack_of(h_err + 1);
                if (kp.total <= 32) {
                    DevBuf k32(s);
                    HIPCHK(k32.alloc(4 * (size_t)P));
                    key_compress_kernel<uint32_t><<<grid_for(P, 256), 256, 0, s>>>(P, key.as<uint64_t>(), kp, k32.as<uint32_t>());
                    HIPCHK(hipGetLastError());
                    TRY(sort_pairs(s, k32.as<uint32_t>(), key_s.as<uint32_t>(), idx.as<uint32_t>(), order.as<uint32_t>(), (size_t)P, (unsigned)kp.total));
                    wba_pick_kernel<uint32_t><<<grid_for(P, 256), 256, 0, s>>>(P, key_s.as<uint32_t>(), order.as<uint32_t>(), d2.as<double>(),
                                                                               flag.as<uint32_t>(), pick.as<uint32_t>());
                } else {
                    key_compress_kernel<uint64_t><<<grid_for(P, 256), 256, 0, s>>>(P, key.as<uint64_t>(), kp, key.as<uint64_t>());
                    HIPCHK(hipGetLastError());
                    const unsigned bits = (unsigned)kp.total;
                    TRY(sort_pairs(s, key.as<uint64_t>(), key_s.as<uint64_t>(), idx.as<uint32_t>(), order.as<uint32_t>(), (size_t)P, bits));
                    wba_pick_kernel<uint64_t><<<grid_for(P, 256), 256, 0, s>>>(P, key_s.as<uint64_t>(), order.as<uint32_t>(), d2.as<double>(),
                                                                               flag.as<uint32_t>(), pick.as<uint32_t>());
                }
                HIPCHK(hipGetLastError());
                HIPCHK(hipMemsetAsync(flag.as<uint32_t>() + P, 0, 4, s));
                TRY(scan_excl<uint32_t>(s, flag.as<uint32_t>(), excl.as<uint32_t>(), (size_t)P + 1));
                uint32_t cnt = 0;
                HIPCHK(lvba::copy_d2h(&cnt, excl.as<uint32_t>() + P, 4));
                n_out = cnt;
                void *raw = nullptr;
                HIPCHK(DevicePool::get().alloc(&raw, 12 * (size_t)std::max<int64_t>(n_out, 1)));
                d_out = (float *)raw;
                wba_compact_kernel<<<grid_for(P, 256), 256, 0, s>>>(P, flag.as<uint32_t>(), excl.as<uint32_t>(), pick.as<uint32_t>(),
                                                                    merged.as<float>(), d_out);
                HIPCHK(hipGetLastError());
                HIPCHK(hipStreamSynchronize(s));
            }
        }
        info.merge_ms = now_ms() - tw;
        info.n_anchor_points = n_out;
        R.d_out = d_out; R.n_out = n_out;
        return LVBA_OK;
    };
    // a stage over all windows, on a small pool of host threads (LVBA_WINDOW_THREADS, default 4; 1 = in the calling thread), each
    // with a stream of its own; the streams live until the call returns (the maps of stage 1 work on them)
    int n_thr = 4;
    if (const char *e = getenv("LVBA_WINDOW_THREADS")) n_thr = atoi(e);
    n_thr = std::max(1, std::min(n_thr, n_win));
    lvba::hvec<hipStream_t> wstreams;
    struct StreamsGuard { lvba::hvec<hipStream_t> &v; ~StreamsGuard() { for (hipStream_t q : v) if (q) lvba::StreamCache::get().release(q); } } wguard{wstreams};
    if (n_thr > 1) {
        wstreams.assign((size_t)n_thr, nullptr);
        for (auto &q : wstreams)
            if (lvba::StreamCache::get().acquire(&q) != hipSuccess) { (void)hipGetLastError(); q = nullptr; }
    }
    auto run_stage = [&](const std::function<int32_t(int, hipStream_t, WinResult &)> &stage) {
        std::atomic<int> next{0};
        lvba::hvec<char> visited((size_t)n_win, 0);
        auto worker = [&](hipStream_t ws) {
            for (int wi = next.fetch_add(1); wi < n_win; wi = next.fetch_add(1)) {
                WinResult &R = results[(size_t)wi];
                visited[(size_t)wi] = 1;
                if (R.rc < 0) continue;
                R.rc = stage(wi, ws, R);
                if (R.rc < 0) R.err = lvba_last_error();
            }
        };
        if (n_thr == 1) {
            worker(s);
            return;
        }
        struct Inhibit { Inhibit() { bs_graph_inhibit(+1); } ~Inhibit() { bs_graph_inhibit(-1); } } inhibit;
        lvba::hvec<std::thread> pool;
        for (int t = 0; t < n_thr; ++t)
            pool.emplace_back([&, t]() {
                if (!wstreams[(size_t)t] || hipSetDevice(sc->device) != hipSuccess) return;
                worker(wstreams[(size_t)t]);
                (void)hipStreamSynchronize(wstreams[(size_t)t]);
            });
        for (auto &th : pool) th.join();
        for (int wi = 0; wi < n_win; ++wi) // a thread that could not get a stream leaves its windows untouched
            if (!visited[(size_t)wi] && results[(size_t)wi].rc >= 0) {
                WinResult &R = results[(size_t)wi];
                R.rc = stage(wi, s, R);
                if (R.rc < 0) R.err = lvba_last_error();
            }
    };
    auto free_maps = [&]() {
        for (auto &q : results) if (q.map) { lvba_voxmap_destroy(q.map); q.map = nullptr; }
        if (joint_map) { lvba_voxmap_destroy(joint_map); joint_map = nullptr; }
    };
    // ---- stage 1 for all windows at once: a root voxel is (window, key), so one sort and one pass of every kernel of the map build
    // serve every window (the per-window builds are dozens of dependent launches and a dozen host round trips EACH); a window's
    // part of the joint map is bit for bit what its own build gives (tests/test_gpu_window.py).  LVBA_WINDOW_JOINT_MAP=0: off.
    // The joint build holds keys, records, indices and sort temporaries of ALL frames at once (~90 bytes per point, where a
    // per-window build needs one window's worth): it is only tried when that fits the device's free memory with room to spare --
    // a long sequence goes window by window instead of running into hipErrorOutOfMemory first.
    auto stage_map_joint = [&]() -> bool {
        static const bool on = [] { const char *e = getenv("LVBA_WINDOW_JOINT_MAP"); return !(e && !strcmp(e, "0")); }();
        if (!on || o.merge_only || n_win < 2) return false;
        {
            size_t free_b = 0, total_b = 0;
            const int64_t P_all = sc->frame_off[(size_t)n] - sc->frame_off[0];
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return false; }
            if ((double)P_all * 96.0 > 0.5 * (double)free_b) {
                if (timing_on("window"))
                    fprintf(stderr, "[window_ba] joint voxel map skipped: %lld points x ~96 B against %.1f GB free -> one map per window\n",
                            (long long)P_all, (double)free_b / 1e9);
                return false;
            }
        }
        const double tw = now_ms();
        if (lvba_voxmap_build_scans_joint(sc, 0, n, w, poses, &o.voxel, s, &joint_map) != LVBA_OK) {
            joint_map = nullptr; // (too many key + window bits, a bad point, ...: the per-window builds say what it is)
            return false;
        }
        const double per = (now_ms() - tw) / n_win;
        for (int wi = 0; wi < n_win; ++wi) {
            WinResult &R = results[(size_t)wi];
            int start, cw;
            win_range(wi, start, cw);
            R.info = lvba_window_info{};
            R.info.start = start; R.info.n_frames = cw; R.info.anchor = -1;
            R.x.assign(poses + 12 * (int64_t)start, poses + 12 * (int64_t)(start + cw));
            R.rc = lvba_voxmap_window_view(joint_map, wi, &R.map);
            if (R.rc != LVBA_OK) { R.err = lvba_last_error(); continue; }
            lvba_voxmap_info_t mi;
            lvba_voxmap_info(R.map, &mi);
            R.info.n_voxels = mi.n_voxels; R.info.n_factors = mi.n_factors;
            R.info.map_ms = per;
            if (mi.n_voxels < 3 * (int64_t)cw) { // :258-262
                R.info.skipped = 1;
                lvba_voxmap_destroy(R.map);
                R.map = nullptr;
            }
        }
        return true;
    };
    // ---- stage 3 for all windows at once, as stage 1: the points of every window moved by their frames' relative poses in one
    // launch, ONE sort by (window, leaf key), one pick / scan / compaction -- straight into the anchor scan set's point array, whose
    // frames are the windows' runs of the compacted sequence (their bounds in the SORTED sequence are the windows' point counts:
    // known on the host).  A window's anchor cloud is bit for bit what its own pass gives: the same fp32 points, the same leaf keys,
    // the same order inside a leaf (the sort is stable and a window's points keep their order), the leaves in key order.  Per window
    // the stage was seventeen launches and five waits on the host.  Same switch and the same memory rule as the joint map.
    float *joint_pts = nullptr; // [anchor points of all windows][3], allocated like a scan set's d_pts
    auto stage_finish_joint = [&]() -> bool {
        static const bool on = [] { const char *e = getenv("LVBA_WINDOW_JOINT_MAP"); return !(e && !strcmp(e, "0")); }();
        const int64_t p_begin = sc->frame_off[0], P = sc->frame_off[(size_t)n] - p_begin;
        if (!on || n_win < 2 || !(o.anchor_leaf >= 0.001) || P <= 0 || P >= ((int64_t)1 << 32)) return false;
        {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return false; }
            if ((double)P * 96.0 > 0.5 * (double)free_b) return false;
        }
        const double tw = now_ms();
        // codes, relative poses of all frames, the windows' bounds in the sorted sequence
        lvba::hvec<uint32_t> code((size_t)n_win);
        lvba::hvec<int64_t> at;
        lvba::hvec<double> rel_all(12 * (size_t)n);
        int G = 0;
        for (int wi = 0; wi < n_win; ++wi)
            if (!results[(size_t)wi].info.skipped) ++G;
        if (G == 0) return false;
        at.push_back(0);
        for (int wi = 0, g = 0; wi < n_win; ++wi) {
            WinResult &R = results[(size_t)wi];
            int start, cw;
            win_range(wi, start, cw);
            if (R.info.skipped) {
                code[(size_t)wi] = (uint32_t)G;
                for (int j = 0; j < cw; ++j) memcpy(rel_all.data() + 12 * (size_t)(start + j), I12, sizeof I12);
                continue;
            }
            code[(size_t)wi] = (uint32_t)g++;
            align_window(wi, R);
            memcpy(rel_all.data() + 12 * (size_t)start, R.rel.data(), 96 * (size_t)cw);
            at.push_back(at.back() + (sc->frame_off[(size_t)(start + cw)] - sc->frame_off[(size_t)start]));
        }
        const int64_t P_live = at.back();
        const int cbits = G < n_win ? 32 - __builtin_clz((unsigned)G) : (G > 1 ? 32 - __builtin_clz((unsigned)(G - 1)) : 0);
        const bool jt = timing_on("window");
        double tj = now_ms();
        auto jm = [&](const char *what) { // (LVBA_TIMING: waits for the stream at every mark)
            if (!jt) return;
            (void)hipStreamSynchronize(s);
            const double t = now_ms();
            fprintf(stderr, "[window_ba merge] %-18s %.3f ms\n", what, t - tj);
            tj = t;
        };
        auto body = [&]() -> int32_t {
            jm("align (host)");
            DevBuf d_rel(s), d_code(s), d_at(s), d_cnt(s), merged(s), key(s), d2(s), idx(s), d_err(s), d_part(s);
            HIPCHK(d_rel.alloc(96 * (size_t)n)); HIPCHK(d_code.alloc(4 * (size_t)n_win)); HIPCHK(d_at.alloc(8 * ((size_t)G + 1)));
            HIPCHK(d_cnt.alloc(4 * ((size_t)G + 1))); HIPCHK(merged.alloc(12 * (size_t)P)); HIPCHK(d_err.alloc(28));
            HIPCHK(key.alloc(8 * (size_t)P)); HIPCHK(d2.alloc(8 * (size_t)P)); HIPCHK(idx.alloc(4 * (size_t)P));
            const int64_t n_slots = key_range_slots(P, 256);
            HIPCHK(d_part.alloc(24 * (size_t)n_slots));
            HIPCHK(lvba::copy_h2d(d_rel.p, rel_all.data(), 96 * (size_t)n)); // (pageable sources: synchronous copies)
            HIPCHK(lvba::copy_h2d(d_code.p, code.data(), 4 * (size_t)n_win));
            HIPCHK(lvba::copy_h2d(d_at.p, at.data(), 8 * ((size_t)G + 1)));
            HIPCHK(hipMemsetAsync(d_err.p, 0, 28, s));
            jm("alloc + tables");
            wba_merge_kernel<<<grid_for(P, 256), 256, 0, s>>>(P, sc->d_pts + 3 * p_begin, sc->d_frame_off, n, d_rel.as<double>(), o.anchor_leaf,
                                                              merged.as<float>(), key.as<uint64_t>(), d2.as<double>(), idx.as<uint32_t>(),
                                                              d_err.as<int>(), d_part.as<int>());
            HIPCHK(hipGetLastError());
            key_range_reduce_kernel<<<key_range_reduce_grid(n_slots), 256, 0, s>>>(n_slots, d_part.as<int>(), d_err.as<int>() + 1);
            HIPCHK(hipGetLastError());
            int h_err[7] = {0};
            HIPCHK(hipStreamSynchronize(s));
            HIPCHK(lvba::copy_d2h(h_err, d_err.p, 28));
            if (h_err[0]) return LVBA_ERR_ARG; // (the per-window passes name the window)
            jm("merge + range");
            const KeyPack kp = key_pack_of(h_err + 1);
            const unsigned bits = (unsigned)(kp.total + cbits);
            if (bits > 64) return LVBA_ERR_UNSUPPORTED;
            DevBuf key_s(s), order(s), flag(s), excl(s), pick(s);
            HIPCHK(key_s.alloc(8 * (size_t)P)); HIPCHK(order.alloc(4 * (size_t)P)); HIPCHK(flag.alloc(4 * ((size_t)P_live + 1)));
            HIPCHK(excl.alloc(4 * ((size_t)P_live + 1))); HIPCHK(pick.alloc(4 * (size_t)std::max<int64_t>(P_live, 1)));
            if (bits <= 32) {
                DevBuf k32(s);
                HIPCHK(k32.alloc(4 * (size_t)P));
                wba_compress_win_kernel<uint32_t><<<grid_for(P, 256), 256, 0, s>>>(P, key.as<uint64_t>(), sc->d_frame_off, n, w, d_code.as<uint32_t>(),
                                                                                   kp, k32.as<uint32_t>());
                HIPCHK(hipGetLastError());
                TRY(sort_pairs(s, k32.as<uint32_t>(), key_s.as<uint32_t>(), idx.as<uint32_t>(), order.as<uint32_t>(), (size_t)P, bits));
                if (P_live > 0)
                    wba_pick_kernel<uint32_t><<<grid_for(P_live, 256), 256, 0, s>>>(P_live, key_s.as<uint32_t>(), order.as<uint32_t>(), d2.as<double>(),
                                                                                    flag.as<uint32_t>(), pick.as<uint32_t>());
            } else {
                wba_compress_win_kernel<uint64_t><<<grid_for(P, 256), 256, 0, s>>>(P, key.as<uint64_t>(), sc->d_frame_off, n, w, d_code.as<uint32_t>(),
                                                                                   kp, key.as<uint64_t>());
                HIPCHK(hipGetLastError());
                TRY(sort_pairs(s, key.as<uint64_t>(), key_s.as<uint64_t>(), idx.as<uint32_t>(), order.as<uint32_t>(), (size_t)P, bits));
                if (P_live > 0)
                    wba_pick_kernel<uint64_t><<<grid_for(P_live, 256), 256, 0, s>>>(P_live, key_s.as<uint64_t>(), order.as<uint32_t>(), d2.as<double>(),
                                                                                    flag.as<uint32_t>(), pick.as<uint32_t>());
            }
            HIPCHK(hipGetLastError());
            jm("sort + pick");
            HIPCHK(hipMemsetAsync(flag.as<uint32_t>() + P_live, 0, 4, s));
            TRY(scan_excl<uint32_t>(s, flag.as<uint32_t>(), excl.as<uint32_t>(), (size_t)P_live + 1));
            wba_bounds_kernel<<<grid_for(G + 1, 256), 256, 0, s>>>(G + 1, d_at.as<int64_t>(), excl.as<uint32_t>(), d_cnt.as<uint32_t>());
            HIPCHK(hipGetLastError());
            lvba::hvec<uint32_t> cnt((size_t)G + 1);
            HIPCHK(hipStreamSynchronize(s));
            HIPCHK(lvba::copy_d2h(cnt.data(), d_cnt.p, 4 * ((size_t)G + 1)));
            const int64_t n_all = cnt[(size_t)G];
            jm("scan + counts");
            hipError_t e = hipMalloc((void **)&joint_pts, n_all ? 12 * (size_t)n_all : 8);
            if (e != hipSuccess) { (void)hipGetLastError(); joint_pts = nullptr; return LVBA_ERR_NOMEM; }
            jm("hipMalloc (cloud)");
            if (P_live > 0) {
                wba_compact_kernel<<<grid_for(P_live, 256), 256, 0, s>>>(P_live, flag.as<uint32_t>(), excl.as<uint32_t>(), pick.as<uint32_t>(),
                                                                         merged.as<float>(), joint_pts);
                HIPCHK(hipGetLastError());
            }
            HIPCHK(hipStreamSynchronize(s));
            jm("compact");
            const double per = (now_ms() - tw) / G;
            for (int wi = 0; wi < n_win; ++wi) {
                WinResult &R = results[(size_t)wi];
                if (R.info.skipped) continue;
                const uint32_t g = code[(size_t)wi];
                R.d_out = nullptr;
                R.n_out = (int64_t)cnt[(size_t)g + 1] - (int64_t)cnt[(size_t)g];
                R.info.n_anchor_points = R.n_out;
                R.info.merge_ms = per;
            }
            return LVBA_OK;
        };
        const int32_t rc = body();
        if (rc != LVBA_OK) {
            if (joint_pts) { (void)hipFree(joint_pts); joint_pts = nullptr; }
            if (timing_on("window")) fprintf(stderr, "[window_ba] joint merge not taken (rc %d): one pass per window\n", rc);
            return false;
        }
        return true;
    };
    const bool timing = timing_on("window"); // stage times of the whole call to stderr
    double tmark = now_ms();
    auto mark = [&](const char *what) {
        if (!timing) return;
        const double t = now_ms();
        fprintf(stderr, "[window_ba] %-18s %.3f ms\n", what, t - tmark);
        tmark = t;
    };
    if (!stage_map_joint()) run_stage(stage_map);
    mark("voxel maps");
    bool any_failed = false;
    for (auto &q : results) any_failed = any_failed || q.rc < 0;
    if (!any_failed && !o.merge_only) {
        const bool batch = o.lm_mode == 0;
        bool done = false;
        if (batch) {
            const int32_t rc = stage_lm_batched(done);
            if (rc < 0) { free_maps(); return rc; }
        }
        if (!done) run_stage(stage_lm_single);
        for (auto &q : results) any_failed = any_failed || q.rc < 0;
        mark(done ? "LM, all windows" : "LM, one by one");
    }
    if (!any_failed && !stage_finish_joint()) run_stage(stage_finish);
    free_maps();
    mark("align + merge");
    for (int wi = 0; wi < n_win; ++wi) { // assemble in window order
        WinResult &R = results[(size_t)wi];
        if (R.rc < 0) {
            for (auto &q : results) if (q.d_out) DevicePool::get().free(q.d_out);
            clouds.clear();
            if (joint_pts) (void)hipFree(joint_pts);
            return lvba_fail(R.rc, "%s", R.err.c_str());
        }
    }
    for (int wi = 0; wi < n_win; ++wi) {
        WinResult &R = results[(size_t)wi];
        const int start = wi * w, cw = std::min(w, n - start);
        if (!R.info.skipped) {
            if (window_poses) memcpy(window_poses + 12 * (int64_t)start, R.x.data(), 96 * (size_t)cw);
            memcpy(rel_poses + 12 * (int64_t)start, R.rel.data(), 96 * (size_t)cw);
            for (int j = 0; j < cw; ++j) anchor_index[start + j] = (int32_t)clouds.size();
            R.info.anchor = (int32_t)clouds.size();
            memcpy(anchor_poses + 12 * clouds.size(), poses + 12 * (int64_t)start, 96);
            clouds.push_back({R.d_out, R.n_out});
        }
        if (win_info) win_info[wi] = R.info;
    }
    // the anchor clouds as a scan set of their own
    lvba_scans_s *out = new (std::nothrow) lvba_scans_s();
    if (!out) { free_clouds(); if (joint_pts) (void)hipFree(joint_pts); return lvba_fail(LVBA_ERR_NOMEM, "host allocation failed"); }
    out->device = sc->device;
    out->n_frames = (int)clouds.size();
    out->frame_off.assign(clouds.size() + 1, 0);
    for (size_t a = 0; a < clouds.size(); ++a) out->frame_off[a + 1] = out->frame_off[a] + clouds[a].n;
    const int64_t PT = out->frame_off.back();
    hipError_t e = hipSuccess;
    if (joint_pts) out->d_pts = joint_pts; // (stage_finish_joint: the windows' clouds already lie one after the other)
    else e = hipMalloc((void **)&out->d_pts, PT ? 12 * (size_t)PT : 8);
    if (e == hipSuccess) e = hipMalloc((void **)&out->d_frame_off, 8 * (clouds.size() + 1));
    for (size_t a = 0; a < clouds.size() && e == hipSuccess && !joint_pts; ++a)
        if (clouds[a].n > 0)
            e = hipMemcpy(out->d_pts + 3 * out->frame_off[a], clouds[a].d, 12 * (size_t)clouds[a].n, hipMemcpyDeviceToDevice);
    if (e == hipSuccess)
        e = lvba::copy_h2d(out->d_frame_off, out->frame_off.data(), 8 * (clouds.size() + 1));
    free_clouds();
    if (e != hipSuccess) {
        lvba_scans_destroy(out);
        return lvba_fail(e == hipErrorOutOfMemory ? LVBA_ERR_NOMEM : LVBA_ERR_DEVICE, "anchor scan set: %s", hipGetErrorString(e));
    }
    *n_anchors = out->n_frames;
    *anchor_scans = out;
    mark("anchor scan set");
    return LVBA_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// The window stage over SEVERAL GPUs of one node.  The windows are independent problems (the reference solves them one after
// the other, src/lvba_system.cpp:232-302), so this is where the GPUs of a node all have work: device k takes a contiguous run
// of whole windows (lvba_window_split), a host thread per device runs lvba_window_ba on that device's scan set, the results are
// put together in window order and the anchor clouds are gathered on the first device (one device-to-device copy per share).
// A single LM refinement, by contrast, is bound by the serial chain of its band factorisation.
extern "C" int32_t lvba_window_split(int32_t n_frames, int32_t window_size, int32_t n_shares, int32_t *frame_begin)
{
    if (n_frames < 0 || window_size < 1 || n_shares < 1 || !frame_begin) return lvba_fail(LVBA_ERR_ARG, "bad argument");
    const int64_t n_win = ((int64_t)n_frames + window_size - 1) / window_size;
    for (int k = 0; k <= n_shares; ++k) { // the thread split of bavoxel.hpp:621-624, on windows
        const int64_t wb = n_win * k / n_shares;
        frame_begin[k] = (int32_t)std::min<int64_t>(n_frames, wb * window_size);
    }
    return LVBA_OK;
}

extern "C" int32_t lvba_window_ba_multi(int32_t n_shares, const lvba_scans_t *scans, const double *poses, const lvba_window_opts *opts,
                                        double *window_poses, double *rel_poses, int32_t *anchor_index, double *anchor_poses,
                                        int32_t *n_anchors, lvba_scans_t *anchor_scans, lvba_window_info *win_info)
{
    if (anchor_scans) *anchor_scans = nullptr;
    if (n_shares < 1 || !scans || !poses || !rel_poses || !anchor_index || !anchor_poses || !n_anchors || !anchor_scans)
        return lvba_fail(LVBA_ERR_ARG, "null argument");
    lvba_window_opts o;
    lvba_window_default_opts(&o);
    if (opts) o = *opts;
    if (o.window_size < 1) return lvba_fail(LVBA_ERR_ARG, "window_size must be >= 1");
    std::vector<int64_t> fb((size_t)n_shares + 1, 0), wb((size_t)n_shares + 1, 0);
    for (int k = 0; k < n_shares; ++k) {
        if (!scans[k]) return lvba_fail(LVBA_ERR_ARG, "share %d: null scan set", k);
        if (k + 1 < n_shares && scans[k]->n_frames % o.window_size)
            return lvba_fail(LVBA_ERR_ARG, "share %d holds %d frames: every share but the last must hold whole windows of %d (lvba_window_split)",
                             k, scans[k]->n_frames, o.window_size);
        fb[(size_t)k + 1] = fb[(size_t)k] + scans[k]->n_frames;
        wb[(size_t)k + 1] = wb[(size_t)k] + (scans[k]->n_frames + o.window_size - 1) / o.window_size;
    }
    struct Share { int32_t rc = LVBA_OK, na = 0; std::string err; lvba_scans_t anchors = nullptr; std::vector<double> ap; };
    std::vector<Share> sh((size_t)n_shares);
    // several host threads drive the device(s) from here on: no solve graph is captured meanwhile (with HIP 7.0 a capture in one
    // thread is invalidated by allocations and synchronous copies in another, block_system.hip)
    struct Inhibit { Inhibit() { bs_graph_inhibit(+1); } ~Inhibit() { bs_graph_inhibit(-1); } } inhibit;
    auto run = [&](int k) {
        Share &S = sh[(size_t)k];
        const int64_t nf = scans[k]->n_frames, nw = wb[(size_t)k + 1] - wb[(size_t)k];
        S.ap.assign(12 * (size_t)std::max<int64_t>(nw, 1), 0.0);
        S.rc = lvba_window_ba(scans[k], poses + 12 * fb[(size_t)k], &o, window_poses ? window_poses + 12 * fb[(size_t)k] : nullptr,
                              rel_poses + 12 * fb[(size_t)k], anchor_index + fb[(size_t)k], S.ap.data(), &S.na, &S.anchors,
                              win_info ? win_info + wb[(size_t)k] : nullptr);
        if (S.rc < 0) S.err = lvba_last_error();
        (void)nf;
    };
    {
        std::vector<std::thread> th;
        for (int k = 1; k < n_shares; ++k) th.emplace_back(run, k);
        run(0);
        for (auto &t : th) t.join();
    }
    auto drop = [&]() { for (auto &S : sh) if (S.anchors) { lvba_scans_destroy(S.anchors); S.anchors = nullptr; } };
    for (int k = 0; k < n_shares; ++k)
        if (sh[(size_t)k].rc < 0) {
            const int32_t rc = sh[(size_t)k].rc;
            const std::string msg = sh[(size_t)k].err;
            drop();
            return lvba_fail(rc, "share %d: %s", k, msg.c_str());
        }
    // window order: anchors of share k come after those of the shares before it
    int32_t base = 0;
    for (int k = 0; k < n_shares; ++k) {
        const Share &S = sh[(size_t)k];
        for (int64_t f = fb[(size_t)k]; f < fb[(size_t)k + 1]; ++f)
            if (anchor_index[f] >= 0) anchor_index[f] += base;
        if (win_info)
            for (int64_t w = wb[(size_t)k]; w < wb[(size_t)k + 1]; ++w) {
                win_info[w].start += (int32_t)fb[(size_t)k];
                if (win_info[w].anchor >= 0) win_info[w].anchor += base;
            }
        memcpy(anchor_poses + 12 * (size_t)base, S.ap.data(), 96 * (size_t)S.na);
        base += S.na;
    }
    // the anchor clouds as ONE scan set on the first share's device
    lvba_scans_s *out = new (std::nothrow) lvba_scans_s();
    if (!out) { drop(); return lvba_fail(LVBA_ERR_NOMEM, "host allocation failed"); }
    out->device = scans[0]->device;
    out->n_frames = base;
    out->frame_off.assign((size_t)base + 1, 0);
    {
        size_t a = 0;
        for (int k = 0; k < n_shares; ++k)
            for (int f = 0; f < sh[(size_t)k].na; ++f, ++a)
                out->frame_off[a + 1] = out->frame_off[a] + (sh[(size_t)k].anchors->frame_off[(size_t)f + 1] - sh[(size_t)k].anchors->frame_off[(size_t)f]);
    }
    hipError_t e = hipSetDevice(out->device);
    const int64_t PT = out->frame_off.back();
    if (e == hipSuccess) e = hipMalloc((void **)&out->d_pts, PT ? 12 * (size_t)PT : 8);
    if (e == hipSuccess) e = hipMalloc((void **)&out->d_frame_off, 8 * ((size_t)base + 1));
    {
        size_t a = 0;
        for (int k = 0; k < n_shares && e == hipSuccess; ++k) {
            const lvba_scans_s *A = sh[(size_t)k].anchors;
            const int64_t np = A->frame_off[(size_t)sh[(size_t)k].na];
            if (np > 0) e = hipMemcpy(out->d_pts + 3 * out->frame_off[a], A->d_pts, 12 * (size_t)np, hipMemcpyDefault); // (across devices: UVA)
            a += (size_t)sh[(size_t)k].na;
        }
    }
    if (e == hipSuccess) e = lvba::copy_h2d(out->d_frame_off, out->frame_off.data(), 8 * ((size_t)base + 1));
    drop();
    if (e != hipSuccess) {
        lvba_scans_destroy(out);
        return lvba_fail(e == hipErrorOutOfMemory ? LVBA_ERR_NOMEM : LVBA_ERR_DEVICE, "anchor scan set: %s", hipGetErrorString(e));
    }
    *n_anchors = base;
    *anchor_scans = out;
    return LVBA_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// LvbaSystem::runLidarBA (src/lvba_system.cpp:312-410) without the ROS/visualisation calls: window BA -> anchors, then the
// global stages (stage 1 optional, stage 2) each re-cutting the anchor clouds at the current anchor poses with that stage's
// voxel size / eigen ratios and running damping_iter over all anchors, then every frame's pose = anchor o rel (:393-404).
extern "C" void lvba_lidar_ba_default_opts(lvba_lidar_ba_opts *o)
{
    if (!o) return;
    lvba_window_default_opts(&o->window);
    o->window_enable = 1;                                        // include/dataset_io.h:70
    o->stage1_enable = 1;                                        // include/dataset_io.h:75
    o->stage_voxel_size[0] = 0.5; o->stage_voxel_size[1] = 0.5;  // include/dataset_io.h:76,79
    const float r1[4] = {0.3f, 0.1f, 0.06f, 0.03f}, r2[4] = {0.08f, 0.08f, 0.08f, 0.08f}; // :77,:80
    for (int k = 0; k < 4; ++k) { o->stage_eigen_ratio[0][k] = r1[k]; o->stage_eigen_ratio[1][k] = r2[k]; }
    lvba_balm_default_opts(&o->lm);
}

// n_shares == 1: the whole sequence on scs[0]'s device; > 1: the window stage over the shares (lvba_window_ba_multi), the global
// stages -- single problems over all anchors -- on the first share's device, where the anchor clouds are gathered
static int32_t lidar_ba_impl(int32_t n_shares, const lvba_scans_t *scs, const double *poses_in, const lvba_lidar_ba_opts *opts,
                             double *poses_out, lvba_lidar_ba_report *rep)
{
    if (n_shares < 1 || !scs || !scs[0] || !poses_in || !poses_out) return lvba_fail(LVBA_ERR_ARG, "null argument");
    lvba_lidar_ba_opts o;
    lvba_lidar_ba_default_opts(&o);
    if (opts) o = *opts;
    lvba_scans_t sc = scs[0];
    int n = 0;
    for (int k = 0; k < n_shares; ++k) {
        if (!scs[k]) return lvba_fail(LVBA_ERR_ARG, "share %d: null scan set", k);
        n += scs[k]->n_frames;
    }
    if (n_shares > 1 && !o.window_enable)
        return lvba_fail(LVBA_ERR_ARG, "several shares need the window stage (window_enable = 0 cuts the RAW scans in the global stages: one device)");
    lvba_lidar_ba_report r{};
    r.n_frames = n;
    lvba::hvec<double> rel(12 * (size_t)n), anchor_poses;
    lvba::hvec<int32_t> aidx((size_t)n);
    lvba_scans_t anchors = nullptr;
    int32_t na = 0;
    double t0 = now_ms();
    if (o.window_enable) {
        const int nw = (n + o.window.window_size - 1) / std::max(1, o.window.window_size);
        anchor_poses.resize(12 * (size_t)std::max(nw, 1));
        lvba::hvec<lvba_window_info> wi((size_t)std::max(nw, 1));
        if (n_shares == 1) TRY(lvba_window_ba(sc, poses_in, &o.window, nullptr, rel.data(), aidx.data(), anchor_poses.data(), &na, &anchors, wi.data()));
        else TRY(lvba_window_ba_multi(n_shares, scs, poses_in, &o.window, nullptr, rel.data(), aidx.data(), anchor_poses.data(), &na, &anchors, wi.data()));
        r.n_windows = nw;
        for (int k = 0; k < nw; ++k) r.n_windows_skipped += wi[k].skipped;
    } else { // :221-229: every frame is its own anchor
        static const double I12[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
        anchor_poses.assign(poses_in, poses_in + 12 * (size_t)n);
        for (int i = 0; i < n; ++i) { memcpy(rel.data() + 12 * i, I12, sizeof I12); aidx[i] = i; }
        na = n;
    }
    r.n_anchors = na;
    r.window_ms = now_ms() - t0;
    struct AnchorGuard { lvba_scans_t a; ~AnchorGuard() { lvba_scans_destroy(a); } } guard{anchors};
    lvba_scans_t cut = o.window_enable ? anchors : sc;
    if (na > 0)
        for (int idx = o.stage1_enable ? 0 : 1; idx < 2; ++idx) {
            t0 = now_ms();
            lvba_voxel_opts vo = o.window.voxel;
            vo.voxel_size = o.stage_voxel_size[idx];
            for (int k = 0; k < 4; ++k) vo.eigen_ratio[k] = o.stage_eigen_ratio[idx][k];
            lvba_voxmap_t map = nullptr;
            TRY(lvba_voxmap_build_scans(cut, 0, na, anchor_poses.data(), &vo, &map));
            lvba_voxmap_info_t mi;
            lvba_voxmap_info(map, &mi);
            r.stage_voxels[idx] = mi.n_voxels; r.stage_factors[idx] = mi.n_factors; r.stage_ran[idx] = 1;
            if (mi.n_voxels == 0) {
                // nothing admitted at this stage's voxel size: upstream, damping_iter over an empty VOX_HESS averages 0 / 0
                // (bavoxel.hpp:634-635), every step is rejected on the NaN cost and the poses come out as they went in
                // (src/lvba_system.cpp:386) -- the stage is a no-op, the next stage and the anchor / rel composition still run
                lvba_voxmap_destroy(map);
                r.stage_ran[idx] = 0;
                r.stage_ms[idx] = now_ms() - t0;
                continue;
            }
            lvba_balm_t b = nullptr;
            int32_t rc = lvba_voxmap_to_balm(map, &b);
            lvba_voxmap_destroy(map);
            if (rc != LVBA_OK) return rc;
            lvba::hvec<lvba_lm_trace> trace((size_t)std::max(1, o.lm.max_iter));
            int32_t nt = 0;
            rc = lvba_balm_refine(b, anchor_poses.data(), &o.lm, trace.data(), &nt);
            lvba_balm_destroy(b);
            if (rc < 0) return rc;
            r.stage_status[idx] = rc; r.stage_iters[idx] = nt;
            if (nt > 0) {
                r.stage_cost_first[idx] = trace[0].residual1;
                r.stage_cost_last[idx] = trace[nt - 1].accepted ? trace[nt - 1].residual2 : trace[nt - 1].residual1;
            }
            r.stage_ms[idx] = now_ms() - t0;
        }
    memcpy(poses_out, poses_in, 96 * (size_t)n); // optimized_x_buf_ = x_buf_full (:393)
    for (int i = 0; i < n; ++i) {
        const int a = aidx[i];
        if (a < 0 || a >= na) continue;
        const double *A = anchor_poses.data() + 12 * (size_t)a, *L = rel.data() + 12 * (size_t)i;
        double *O = poses_out + 12 * (size_t)i;
        mat3_mul(A, L, O);
        for (int q = 0; q < 3; ++q) O[9 + q] = A[3 * q] * L[9] + A[3 * q + 1] * L[10] + A[3 * q + 2] * L[11] + A[9 + q];
    }
    if (rep) *rep = r;
    return LVBA_OK;
}

extern "C" int32_t lvba_lidar_ba(lvba_scans_t sc, const double *poses_in, const lvba_lidar_ba_opts *opts, double *poses_out,
                                 lvba_lidar_ba_report *rep)
{
    return lidar_ba_impl(1, &sc, poses_in, opts, poses_out, rep);
}

extern "C" int32_t lvba_lidar_ba_multi(int32_t n_shares, const lvba_scans_t *scans, const double *poses_in,
                                       const lvba_lidar_ba_opts *opts, double *poses_out, lvba_lidar_ba_report *rep)
{
    return lidar_ba_impl(n_shares, scans, poses_in, opts, poses_out, rep);
}
