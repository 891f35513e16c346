// C++ restatement of the reference's voxel front-end, data structures included (std::unordered_map of octree roots,
// per-node point vectors, recursion).  TEST INFRASTRUCTURE ONLY: second, structurally faithful oracle next to
// oracle/voxel_oracle.py and the CPU baseline that bench/tools time beside the GPU front-end.
//   cut_voxel                        include/BALM/bavoxel.hpp:799-836
//   OCTO_TREE_NODE::cut_func/recut   include/BALM/bavoxel.hpp:357-464 (judge_eigen :335-352)
//   OCTO_TREE_NODE::tras_opt         include/BALM/bavoxel.hpp:466-474 -> VOX_HESS::push_voxel :45-54
//   VOXEL_LOC + its std::hash        include/BALM/tools.hpp:29-60
// Pinned against the reference's own headers compiled with the Eigen / PCL stand-ins of oracle/shim (tests/test_ref_pin.py:
// bit-identical clusters) and by the hand-built cases of tests/test_voxel_oracle.py through the Python twin, which this
// file must match bit for bit.
// Eigen::SelfAdjointEigenSolver is replaced by a cyclic Jacobi iteration (results differ at rounding level only).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <unordered_map>
#include <vector>

namespace {

struct VoxelLoc {
    int64_t x, y, z;
    bool operator==(const VoxelLoc &o) const { return x == o.x && y == o.y && z == o.z; }
    bool operator<(const VoxelLoc &o) const { return x != o.x ? x < o.x : (y != o.y ? y < o.y : z < o.z); }
};
struct VoxelHash { // tools.hpp:50-58, HASH_P 116101, MAX_N 10000000000
    size_t operator()(const VoxelLoc &s) const
    {
        const size_t P = 116101, M = 10000000000ull;
        return (((std::hash<int64_t>()(s.z) * P) % M + std::hash<int64_t>()(s.y)) * P) % M + std::hash<int64_t>()(s.x);
    }
};
struct V3 { double x, y, z; };
struct Cluster { // PointCluster, tools.hpp:407-466 (P symmetric, kept as 6)
    double P[6] = {0, 0, 0, 0, 0, 0}, v[3] = {0, 0, 0};
    int N = 0;
    void push(const V3 &p)
    {
        P[0] += p.x * p.x; P[1] += p.x * p.y; P[2] += p.x * p.z; P[3] += p.y * p.y; P[4] += p.y * p.z; P[5] += p.z * p.z;
        v[0] += p.x; v[1] += p.y; v[2] += p.z;
        N++;
    }
};

void eigh3(const double C[6], double lam[3], double U[9])
{
    double a[9] = {C[0], C[1], C[2], C[1], C[3], C[4], C[2], C[4], C[5]};
    double V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    static const int PQ[3][2] = {{0, 1}, {0, 2}, {1, 2}};
    for (int sweep = 0; sweep < 50; sweep++) {
        if (std::fabs(a[1]) + std::fabs(a[2]) + std::fabs(a[5]) == 0.0) break;
        for (auto &pq : PQ) {
            const int p = pq[0], q = pq[1];
            const double apq = a[3 * p + q];
            if (apq == 0.0) continue;
            const double theta = (a[3 * q + q] - a[3 * p + p]) / (2.0 * apq);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
            const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
            for (int k = 0; k < 3; k++) { const double x = a[3 * k + p], y = a[3 * k + q]; a[3 * k + p] = c * x - s * y; a[3 * k + q] = s * x + c * y; }
            for (int k = 0; k < 3; k++) { const double x = a[3 * p + k], y = a[3 * q + k]; a[3 * p + k] = c * x - s * y; a[3 * q + k] = s * x + c * y; }
            a[3 * p + q] = a[3 * q + p] = 0.0;
            for (int k = 0; k < 3; k++) { const double x = V[3 * k + p], y = V[3 * k + q]; V[3 * k + p] = c * x - s * y; V[3 * k + q] = s * x + c * y; }
        }
    }
    int idx[3] = {0, 1, 2};
    const double d[3] = {a[0], a[4], a[8]};
    std::sort(idx, idx + 3, [&](int i, int j) { return d[i] < d[j]; });
    for (int m = 0; m < 3; m++) {
        lam[m] = d[idx[m]];
        for (int k = 0; k < 3; k++) U[3 * k + m] = V[3 * k + idx[m]];
    }
}

struct Ctx {
    int win;
    const double *poses; // [win][12]
    float ratio[4];
    int min_ps;
};

struct Node {
    int layer = 0, state = 0; // 0 UNKNOWN, 1 MID_NODE, 2 PLANE
    float center[3] = {0, 0, 0}, quater = 0;
    std::vector<std::vector<V3>> vec_orig;
    std::vector<Cluster> sig_orig;
    Node *leaves[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    double plane_c[3] = {0, 0, 0}, plane_n[3] = {0, 0, 0};
    explicit Node(int win) : vec_orig(win), sig_orig(win) {}
    ~Node() { for (auto *l : leaves) delete l; }

    bool judge_eigen(const Ctx &cx) // bavoxel.hpp:335-352
    {
        double S[6] = {0, 0, 0, 0, 0, 0}, sv[3] = {0, 0, 0}, sn = 0;
        for (int i = 0; i < cx.win; i++) {
            const Cluster &c = sig_orig[i];
            if (c.N <= 0) continue;
            const double *R = cx.poses + 12 * i, *p = R + 9; // PointCluster::transform, tools.hpp:450-456
            const double Pm[9] = {c.P[0], c.P[1], c.P[2], c.P[1], c.P[3], c.P[4], c.P[2], c.P[4], c.P[5]};
            double RP[9], RPR[9], Rv[3];
            for (int r = 0; r < 3; r++) {
                Rv[r] = R[3 * r] * c.v[0] + R[3 * r + 1] * c.v[1] + R[3 * r + 2] * c.v[2];
                for (int k = 0; k < 3; k++) RP[3 * r + k] = R[3 * r] * Pm[k] + R[3 * r + 1] * Pm[3 + k] + R[3 * r + 2] * Pm[6 + k];
            }
            for (int r = 0; r < 3; r++)
                for (int k = 0; k < 3; k++) RPR[3 * r + k] = RP[3 * r] * R[3 * k] + RP[3 * r + 1] * R[3 * k + 1] + RP[3 * r + 2] * R[3 * k + 2];
            const int ij[6][2] = {{0, 0}, {0, 1}, {0, 2}, {1, 1}, {1, 2}, {2, 2}};
            for (int e = 0; e < 6; e++) {
                const int a = ij[e][0], b = ij[e][1];
                S[e] += RPR[3 * a + b] + Rv[a] * p[b] + p[a] * Rv[b] + c.N * p[a] * p[b];
            }
            for (int r = 0; r < 3; r++) sv[r] += Rv[r] + c.N * p[r];
            sn += c.N;
        }
        const double cen[3] = {sv[0] / sn, sv[1] / sn, sv[2] / sn};
        const double C[6] = {S[0] / sn - cen[0] * cen[0], S[1] / sn - cen[0] * cen[1], S[2] / sn - cen[0] * cen[2],
                             S[3] / sn - cen[1] * cen[1], S[4] / sn - cen[1] * cen[2], S[5] / sn - cen[2] * cen[2]};
        double lam[3], U[9];
        eigh3(C, lam, U);
        for (int r = 0; r < 3; r++) { plane_c[r] = cen[r]; plane_n[r] = U[3 * r]; }
        return !(lam[0] / lam[2] > (double)cx.ratio[layer]);
    }
    void cut_func(const Ctx &cx, int ci) // bavoxel.hpp:357-389
    {
        const double *R = cx.poses + 12 * ci, *t = R + 9;
        for (const V3 &p : vec_orig[ci]) {
            const double w[3] = {R[0] * p.x + R[1] * p.y + R[2] * p.z + t[0], R[3] * p.x + R[4] * p.y + R[5] * p.z + t[1],
                                 R[6] * p.x + R[7] * p.y + R[8] * p.z + t[2]};
            int xyz[3] = {0, 0, 0};
            for (int k = 0; k < 3; k++)
                if (w[k] > center[k]) xyz[k] = 1;
            const int leaf = 4 * xyz[0] + 2 * xyz[1] + xyz[2];
            if (!leaves[leaf]) {
                Node *n = new Node(cx.win);
                for (int k = 0; k < 3; k++) n->center[k] = center[k] + (2 * xyz[k] - 1) * quater;
                n->quater = quater / 2.0;
                n->layer = layer + 1;
                leaves[leaf] = n;
            }
            leaves[leaf]->vec_orig[ci].push_back(p);
            leaves[leaf]->sig_orig[ci].push(p);
        }
        std::vector<V3>().swap(vec_orig[ci]);
    }
    void recut(const Ctx &cx) // bavoxel.hpp:391-464
    {
        if (state == 0) {
            int point_size = 0;
            for (int i = 0; i < cx.win; i++) point_size += sig_orig[i].N;
            if (point_size < cx.min_ps) { state = 1; return; }
            if (judge_eigen(cx)) { state = 2; return; }
            if (layer == 2) { state = 1; return; }
            for (int i = 0; i < cx.win; i++) cut_func(cx, i);
            state = 3; // split (the reference leaves octo_state UNKNOWN here; the flag only stops a second cut)
        }
        for (auto *l : leaves)
            if (l) l->recut(cx);
    }
};

struct Result {
    std::vector<int64_t> off{0};
    std::vector<int32_t> idx;
    std::vector<double> clu;
    std::vector<int64_t> key; // [V][4]
    int64_t n_roots = 0, n_planes = 0;
};

void tras_opt(const Node *n, const Ctx &cx, const VoxelLoc &k, int path, Result &out) // bavoxel.hpp:466-474, :45-54
{
    if (n->state == 2) {
        out.n_planes++;
        int nz = 0;
        for (int i = 0; i < cx.win; i++) nz += n->sig_orig[i].N != 0;
        if (nz < 2) return;
        for (int i = 0; i < cx.win; i++) {
            const Cluster &c = n->sig_orig[i];
            if (c.N == 0) continue;
            out.idx.push_back(i);
            out.clu.insert(out.clu.end(), c.P, c.P + 6);
            out.clu.insert(out.clu.end(), c.v, c.v + 3);
            out.clu.push_back((double)c.N);
        }
        out.off.push_back((int64_t)out.idx.size());
        const int64_t kk[4] = {k.x, k.y, k.z, path};
        out.key.insert(out.key.end(), kk, kk + 4);
    } else
        for (int o = 0; o < 8; o++)
            if (n->leaves[o]) {
                const int child_path = n->layer == 0 ? (1 | (o << 4)) : (2 | (path & 0xf0) | (o << 8));
                tras_opt(n->leaves[o], cx, k, child_path, out);
            }
}

} // namespace

extern "C" {

// Builds the map of `win` frames (points: packed fp32 xyz, frame f = points[frame_off[f] .. frame_off[f+1])) and
// returns an opaque result; voxels sorted by (root key, octant path) with path = layer | o1 << 4 | o2 << 8.
void *vo_build(int win, const int64_t *frame_off, const float *pts, const double *poses, double voxel_size,
               const float ratio[4], int min_ps)
{
    Ctx cx{win, poses, {ratio[0], ratio[1], ratio[2], ratio[3]}, min_ps};
    std::unordered_map<VoxelLoc, Node *, VoxelHash> surf_map;
    for (int f = 0; f < win; f++) { // cut_voxel, bavoxel.hpp:799-836
        const double *R = poses + 12 * f, *t = R + 9;
        for (int64_t i = frame_off[f]; i < frame_off[f + 1]; i++) {
            const V3 p{pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
            const double w[3] = {R[0] * p.x + R[1] * p.y + R[2] * p.z + t[0], R[3] * p.x + R[4] * p.y + R[5] * p.z + t[1],
                                 R[6] * p.x + R[7] * p.y + R[8] * p.z + t[2]};
            float loc[3];
            for (int j = 0; j < 3; j++) {
                loc[j] = w[j] / voxel_size;
                if (loc[j] < 0) loc[j] -= 1.0;
            }
            const VoxelLoc key{(int64_t)loc[0], (int64_t)loc[1], (int64_t)loc[2]};
            auto it = surf_map.find(key);
            if (it == surf_map.end()) {
                Node *n = new Node(win);
                n->center[0] = (0.5 + key.x) * voxel_size;
                n->center[1] = (0.5 + key.y) * voxel_size;
                n->center[2] = (0.5 + key.z) * voxel_size;
                n->quater = voxel_size / 4.0;
                it = surf_map.emplace(key, n).first;
            }
            it->second->vec_orig[f].push_back(p);
            it->second->sig_orig[f].push(p);
        }
    }
    for (auto &kv : surf_map) kv.second->recut(cx);
    std::vector<VoxelLoc> keys;
    keys.reserve(surf_map.size());
    for (auto &kv : surf_map) keys.push_back(kv.first);
    std::sort(keys.begin(), keys.end());
    Result *res = new Result();
    res->n_roots = (int64_t)keys.size();
    for (const VoxelLoc &k : keys) tras_opt(surf_map[k], cx, k, 0, *res);
    for (auto &kv : surf_map) delete kv.second;
    return res;
}
void vo_sizes(void *h, int64_t *n_roots, int64_t *n_planes, int64_t *n_voxels, int64_t *n_factors)
{
    const Result *r = (const Result *)h;
    *n_roots = r->n_roots; *n_planes = r->n_planes;
    *n_voxels = (int64_t)r->off.size() - 1; *n_factors = (int64_t)r->idx.size();
}
void vo_export(void *h, int64_t *off, int32_t *idx, double *clu, int64_t *key)
{
    const Result *r = (const Result *)h;
    std::memcpy(off, r->off.data(), r->off.size() * 8);
    if (!r->idx.empty()) {
        std::memcpy(idx, r->idx.data(), r->idx.size() * 4);
        std::memcpy(clu, r->clu.data(), r->clu.size() * 8);
        std::memcpy(key, r->key.data(), r->key.size() * 8);
    }
}
void vo_free(void *h) { delete (Result *)h; }
}
