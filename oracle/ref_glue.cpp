// ref_glue.cpp -- TEST INFRASTRUCTURE ONLY: C entry points around the REFERENCE'S OWN BALM code.
//
// This file includes include/BALM/bavoxel.hpp (and through it tools.hpp) from where they lie under /root/reference --
// nothing of them is copied into this repository -- and is compiled by `make -C oracle ref` into
// oracle/_ref/libbalm_ref.so against the Eigen / PCL stand-ins of oracle/shim/ (Eigen and PCL are not installed here; see
// oracle/shim/lvba_eigen_standin.h for what the stand-in supplies in Eigen's place: storage, products, the 3x3 symmetric
// eigen-solver and the LDL^T).  The functions below only move data in and out of the reference's types and repeat the few
// lines of its call sites that string the pieces together (cited at each function).  tests/test_ref_pin.py uses the library
// to pin oracle/balm_oracle.{py,c}, oracle/voxel_oracle.{py,cpp} and oracle/window_oracle.py against the reference.
#include <array>
#include <cctype>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>

#include "BALM/bavoxel.hpp"

namespace {

// packed cluster: P00 P01 P02 P11 P12 P22 v0 v1 v2 N
void unpack_cluster(const double *c, PointCluster &pc)
{
    pc.P << c[0], c[1], c[2], c[1], c[3], c[4], c[2], c[4], c[5];
    pc.v << c[6], c[7], c[8];
    pc.N = (int)c[9];
}
void pack_cluster(const PointCluster &pc, double *c)
{
    c[0] = pc.P(0, 0); c[1] = pc.P(0, 1); c[2] = pc.P(0, 2); c[3] = pc.P(1, 1); c[4] = pc.P(1, 2); c[5] = pc.P(2, 2);
    c[6] = pc.v[0]; c[7] = pc.v[1]; c[8] = pc.v[2];
    c[9] = (double)pc.N;
}
// pose: R row-major (9) then p (3)
void unpack_poses(int win, const double *x, std::vector<IMUST> &xs)
{
    xs.assign((size_t)win, IMUST());
    for (int i = 0; i < win; ++i) {
        const double *p = x + 12 * i;
        xs[i].R << p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8];
        xs[i].p << p[9], p[10], p[11];
    }
}
void pack_poses(const std::vector<IMUST> &xs, double *x)
{
    for (size_t i = 0; i < xs.size(); ++i) {
        double *p = x + 12 * i;
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) p[3 * r + c] = xs[i].R(r, c);
        for (int r = 0; r < 3; ++r) p[9 + r] = xs[i].p[r];
    }
}

struct Problem {
    int win;
    std::vector<std::vector<PointCluster>> sig; // owned clusters, one vector<PointCluster>(win) per voxel
    VOX_HESS vox;
    Problem(int w, int64_t V, const double *clusters) : win(w), sig((size_t)V), vox(w)
    {
        for (int64_t a = 0; a < V; ++a) {
            sig[a].resize((size_t)w);
            for (int i = 0; i < w; ++i) unpack_cluster(clusters + ((size_t)a * w + i) * 10, sig[a][i]);
        }
        for (int64_t a = 0; a < V; ++a) vox.push_voxel(&sig[a], nullptr); // bavoxel.hpp:45-54 decides admission
    }
};

struct PlaneOut {
    int64_t key[3];
    int64_t path; // layer << 6 | o1 << 3 | o2
    std::vector<double> clusters; // [win][10]
    double center[3], direct[3], value[3];
};

struct Map {
    int win = 0;
    std::unordered_map<VOXEL_LOC, OCTO_TREE_ROOT *> surf_map;
    std::vector<PlaneOut> planes;
    int64_t n_admitted = 0;
    ~Map()
    {
        for (auto &kv : surf_map) delete kv.second;
    }
};

void walk(const VOXEL_LOC &key, OCTO_TREE_NODE *n, int64_t path, int win, std::vector<PlaneOut> &out)
{
    if (n->octo_state == PLANE) {
        PlaneOut p;
        p.key[0] = key.x; p.key[1] = key.y; p.key[2] = key.z;
        p.path = ((int64_t)n->layer << 6) | path;
        p.clusters.resize((size_t)win * 10);
        for (int i = 0; i < win; ++i) pack_cluster(n->sig_orig[i], &p.clusters[(size_t)i * 10]);
        for (int r = 0; r < 3; ++r) { p.center[r] = n->center[r]; p.direct[r] = n->direct[r]; p.value[r] = n->value_vector[r]; }
        out.push_back(p);
        return;
    }
    for (int o = 0; o < 8; ++o)
        if (n->leaves[o] != nullptr) walk(key, n->leaves[o], n->layer == 0 ? (int64_t)o << 3 : path | o, win, out);
}

} // namespace

extern "C" {

// VOX_HESS::acc_evaluate2 over all admitted voxels (bavoxel.hpp:68-174).  H: [6w][6w] (symmetric), g: [6w].
// Returns the number of admitted voxels.
int64_t ref_acc_evaluate2(int win, int64_t V, const double *clusters, const double *poses, double *H, double *g, double *residual)
{
    Problem pr(win, V, clusters);
    std::vector<IMUST> xs;
    unpack_poses(win, poses, xs);
    Eigen::MatrixXd Hess(6 * win, 6 * win);
    Eigen::VectorXd JacT(6 * win);
    double res = 0;
    pr.vox.acc_evaluate2(xs, 0, (int)pr.vox.plvec_voxels.size(), Hess, JacT, res);
    for (int c = 0; c < 6 * win; ++c)
        for (int r = 0; r < 6 * win; ++r) H[(size_t)r * 6 * win + c] = Hess(r, c);
    for (int r = 0; r < 6 * win; ++r) g[r] = JacT[r];
    *residual = res;
    return (int64_t)pr.vox.plvec_voxels.size();
}

// BALM2::divide_thread (bavoxel.hpp:597-639): the 16-thread split and the AVG_THR division.
double ref_divide_thread(int win, int64_t V, const double *clusters, const double *poses, double *H, double *g)
{
    Problem pr(win, V, clusters);
    std::vector<IMUST> xs, x_ab((size_t)win);
    unpack_poses(win, poses, xs);
    BALM2 opt(win);
    Eigen::MatrixXd Hess(6 * win, 6 * win);
    Eigen::VectorXd JacT(6 * win);
    const double r = opt.divide_thread(xs, pr.vox, x_ab, Hess, JacT);
    for (int c = 0; c < 6 * win; ++c)
        for (int rr = 0; rr < 6 * win; ++rr) H[(size_t)rr * 6 * win + c] = Hess(rr, c);
    for (int rr = 0; rr < 6 * win; ++rr) g[rr] = JacT[rr];
    return r;
}

// BALM2::only_residual (bavoxel.hpp:641-648) -> VOX_HESS::evaluate_only_residual (:176-203)
double ref_only_residual(int win, int64_t V, const double *clusters, const double *poses, int is_avg)
{
    Problem pr(win, V, clusters);
    std::vector<IMUST> xs, x_ab((size_t)win);
    unpack_poses(win, poses, xs);
    BALM2 opt(win);
    return opt.only_residual(xs, pr.vox, x_ab, is_avg != 0);
}

// BALM2::damping_iter (bavoxel.hpp:662-767): poses in, refined poses out.
int64_t ref_damping_iter(int win, int64_t V, const double *clusters, double *poses)
{
    Problem pr(win, V, clusters);
    std::vector<IMUST> xs;
    unpack_poses(win, poses, xs);
    BALM2 opt(win);
    opt.damping_iter(xs, pr.vox);
    pack_poses(xs, poses);
    return (int64_t)pr.vox.plvec_voxels.size();
}

// Exp / Log / PointCluster::transform (tools.hpp:62-77, 98-103, 450-464)
void ref_exp(const double *w, double *R)
{
    const Eigen::Matrix3d M = Exp(Eigen::Vector3d(w[0], w[1], w[2]));
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) R[3 * r + c] = M(r, c);
}
void ref_transform_cluster(const double *cluster, const double *pose, double *out)
{
    PointCluster in, o;
    unpack_cluster(cluster, in);
    std::vector<IMUST> xs;
    unpack_poses(1, pose, xs);
    o.transform(in, xs[0]);
    pack_cluster(o, out);
}

// The map build of one window / stage as the reference's call sites do it (src/lvba_system.cpp:247-257, 366-377):
// cut_voxel for every frame, then recut + tras_opt for every root.  pts: xyz per point; poses [win][12].
void *ref_map_build(int win, const int64_t *frame_off, const float *pts, const double *poses, double voxel_size,
                    const float *eigen_ratio4)
{
    Map *m = new Map;
    m->win = win;
    std::array<float, 4> er = {eigen_ratio4[0], eigen_ratio4[1], eigen_ratio4[2], eigen_ratio4[3]};
    set_eigen_ratio_array(er);
    std::vector<IMUST> xs;
    unpack_poses(win, poses, xs);
    for (int j = 0; j < win; ++j) {
        pcl::PointCloud<PointType> cloud;
        cloud.reserve((size_t)(frame_off[j + 1] - frame_off[j]));
        for (int64_t i = frame_off[j]; i < frame_off[j + 1]; ++i) {
            PointType p;
            p.x = pts[3 * i]; p.y = pts[3 * i + 1]; p.z = pts[3 * i + 2];
            cloud.push_back(p);
        }
        cut_voxel(m->surf_map, cloud, xs[j], j, win, voxel_size, eigen_ratio4[0]);
    }
    VOX_HESS vox(win);
    for (auto iter = m->surf_map.begin(); iter != m->surf_map.end(); ++iter) {
        iter->second->recut(xs);
        iter->second->tras_opt(vox);
    }
    m->n_admitted = (int64_t)vox.plvec_voxels.size();
    for (auto &kv : m->surf_map) walk(kv.first, kv.second, 0, win, m->planes);
    return m;
}
void ref_map_sizes(void *h, int64_t *n_roots, int64_t *n_planes, int64_t *n_admitted)
{
    Map *m = (Map *)h;
    *n_roots = (int64_t)m->surf_map.size();
    *n_planes = (int64_t)m->planes.size();
    *n_admitted = m->n_admitted;
}
// keys [P][4] (x, y, z, path), clusters [P][win][10], geo [P][9] (center, direct, eigenvalues); map iteration order
void ref_map_export(void *h, int64_t *keys, double *clusters, double *geo)
{
    Map *m = (Map *)h;
    for (size_t a = 0; a < m->planes.size(); ++a) {
        const PlaneOut &p = m->planes[a];
        keys[4 * a] = p.key[0]; keys[4 * a + 1] = p.key[1]; keys[4 * a + 2] = p.key[2]; keys[4 * a + 3] = p.path;
        std::memcpy(clusters + a * (size_t)m->win * 10, p.clusters.data(), sizeof(double) * (size_t)m->win * 10);
        for (int r = 0; r < 3; ++r) { geo[9 * a + r] = p.center[r]; geo[9 * a + 3 + r] = p.direct[r]; geo[9 * a + 6 + r] = p.value[r]; }
    }
}
// landmark -> plane (src/lvba_system.cpp:1531-1565 around OCTO_TREE_NODE::findCorrespondPoint, bavoxel.hpp:320-333).
// out [n][4] = (n, d), zeros when there is no plane.
void ref_map_find_planes(void *h, int64_t n, const double *X, double surf_voxel_size, double *out)
{
    Map *m = (Map *)h;
    for (int64_t pi = 0; pi < n; ++pi) {
        double *o = out + 4 * pi;
        o[0] = o[1] = o[2] = o[3] = 0.0;
        Eigen::Vector3d Xp(X[3 * pi], X[3 * pi + 1], X[3 * pi + 2]);
        if (!std::isfinite(Xp[0]) || !std::isfinite(Xp[1]) || !std::isfinite(Xp[2])) continue;
        float loc_xyz[3];
        for (int j = 0; j < 3; ++j) {
            loc_xyz[j] = Xp[j] / surf_voxel_size;
            if (loc_xyz[j] < 0) loc_xyz[j] -= 1.0f;
        }
        VOXEL_LOC key((int64_t)loc_xyz[0], (int64_t)loc_xyz[1], (int64_t)loc_xyz[2]);
        auto it = m->surf_map.find(key);
        if (it == m->surf_map.end() || it->second == nullptr) continue;
        OCTO_TREE_NODE *node = it->second->findCorrespondPoint(Xp);
        if (node == nullptr || node->octo_state != PLANE) continue;
        if (node->direct.norm() < 1e-6) continue;
        Eigen::Vector3d nrm = node->direct;
        nrm.normalize();
        o[0] = nrm[0]; o[1] = nrm[1]; o[2] = nrm[2];
        o[3] = -nrm.dot(node->center);
    }
}
void ref_map_free(void *h) { delete (Map *)h; }

// down_sampling_voxel2 (tools.hpp:259-318) on xyz points; returns the number kept, their xyz in out (map order).
int64_t ref_down_sampling_voxel2(int64_t n, const float *pts, double leaf, float *out)
{
    pcl::PointCloud<PointType> cloud;
    cloud.reserve((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        PointType p;
        p.x = pts[3 * i]; p.y = pts[3 * i + 1]; p.z = pts[3 * i + 2];
        cloud.push_back(p);
    }
    down_sampling_voxel2(cloud, leaf);
    for (size_t i = 0; i < cloud.size(); ++i) { out[3 * i] = cloud[i].x; out[3 * i + 1] = cloud[i].y; out[3 * i + 2] = cloud[i].z; }
    return (int64_t)cloud.size();
}

// pl_transform (tools.hpp:333-343): fp32 points through a double pose, written back as float
void ref_pl_transform(int64_t n, float *pts, const double *pose)
{
    pcl::PointCloud<PointType> cloud;
    for (int64_t i = 0; i < n; ++i) {
        PointType p;
        p.x = pts[3 * i]; p.y = pts[3 * i + 1]; p.z = pts[3 * i + 2];
        cloud.push_back(p);
    }
    std::vector<IMUST> xs;
    unpack_poses(1, pose, xs);
    pl_transform(cloud, xs[0]);
    for (int64_t i = 0; i < n; ++i) { pts[3 * i] = cloud[i].x; pts[3 * i + 1] = cloud[i].y; pts[3 * i + 2] = cloud[i].z; }
}

} // extern "C"
